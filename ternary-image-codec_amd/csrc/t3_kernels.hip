// t3_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Word27 path.  No MFMA anywhere: the
// work is byte/trit permutation plus GF(3)-linear small-field arithmetic and the bound is HBM.
//
//   K1 pack_pixels_kernel     pixels -> raw Word27                       (encode_raw_pixels_to_words OLD:723-734)
//   K5 unpack_words_kernel    raw Word27 -> pixels                       (decode_raw_words_to_pixels OLD:735-747)
//   K2 encode_kernel<FE>      pixels|raw words -> coded band-serial body (encode_profile_from_raw OLD:1043-1169)
//      beacon_kernel          sparse beacon insertion pass               (OLD:1118-1141)
//      rs_encode_blocks_kernel  block-level RSCodec::encode_block        (OLD:517-535)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t3_device.h"
#include "t3_rs_core.h"

namespace t3 {

extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const DevDiv& d) { return d.d <= 1 ? n : (__umulhi(n, d.mul) >> d.sh); }

// ---------------------------------------------------------------------------------------------------------
// symbol construction
// ---------------------------------------------------------------------------------------------------------
// Components as the reference's i2tr sees them: v % 3^w of the uint32 cast (no clamping, OLD:675-682,697-702).
__device__ __forceinline__ uint32_t red_y(uint32_t y16) { return y16 % 243u; }
__device__ __forceinline__ uint32_t red_c(uint32_t c16) { return (uint32_t)((int32_t)(int16_t)c16 + 40) % 81u; }

// 3 pixels = 39 trits = 13 symbols (trit t of the stream = trit t%13 of pixel t/13; Y:5, Cb+40:4, Cr+40:4).
// Every symbol is a div/mod-by-power-of-3 splice of at most two components — no per-trit work.
__device__ __forceinline__ void px3_to_sym13(const uint32_t* c /*9 reduced comps*/, uint32_t* s /*13*/) {
    const uint32_t Y0 = c[0], B0 = c[1], R0 = c[2], Y1 = c[3], B1 = c[4], R1 = c[5], Y2 = c[6], B2 = c[7], R2 = c[8];
    s[0]  = Y0 % 27u;
    s[1]  = Y0 / 27u + 9u * (B0 % 3u);
    s[2]  = B0 / 3u;
    s[3]  = R0 % 27u;
    s[4]  = R0 / 27u + 3u * (Y1 % 9u);
    s[5]  = Y1 / 9u;
    s[6]  = B1 % 27u;
    s[7]  = B1 / 27u + 3u * (R1 % 9u);
    s[8]  = R1 / 9u + 9u * (Y2 % 3u);
    s[9]  = (Y2 / 3u) % 27u;
    s[10] = Y2 / 81u + 3u * (B2 % 9u);
    s[11] = B2 / 9u + 9u * (R2 % 3u);
    s[12] = R2 / 3u;
}

// 3 raw words (27 canonical symbols, trit 26 of each dropped, OLD:1065-1076) = 78 trits = 26 symbols.
__device__ __forceinline__ void w3_to_sym26(const uint32_t* c /*27 symbols %27*/, uint32_t* s /*26*/) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = c[i];
    s[8] = c[8] % 9u + 9u * (c[9] % 3u);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[9 + i] = c[9 + i] / 3u + 9u * (c[10 + i] % 3u);
    s[17] = (c[17] / 3u) % 3u + 3u * (c[18] % 9u);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[18 + i] = c[18 + i] / 9u + 3u * (c[19 + i] % 9u);
}

// ---------------------------------------------------------------------------------------------------------
// K1 / K5 : RAW packer (2 pixels <-> 9 symbols).  One lane = 4 words = 8 pixels = 48 B <-> 36 B.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px2_to_word(const uint32_t* c /*6 reduced comps*/, uint32_t* s /*9*/) {
    // T[0..4]=Ya T[5..8]=Cba T[9..12]=Cra T[13..17]=Yb T[18..21]=Cbb T[22..25]=Crb T[26]=0 (OLD:693-705)
    const uint32_t Y0 = c[0], B0 = c[1], R0 = c[2], Y1 = c[3], B1 = c[4], R1 = c[5];
    s[0] = Y0 % 27u; s[1] = Y0 / 27u + 9u * (B0 % 3u); s[2] = B0 / 3u; s[3] = R0 % 27u;
    s[4] = R0 / 27u + 3u * (Y1 % 9u); s[5] = Y1 / 9u; s[6] = B1 % 27u; s[7] = B1 / 27u + 3u * (R1 % 9u);
    s[8] = R1 / 9u;
}

__global__ __launch_bounds__(256) void pack_pixels_kernel(const uint16_t* __restrict__ px, uint64_t n_px, uint8_t* __restrict__ words, uint64_t n_words) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t c[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint64_t p = 2 * w + i;
        if (p < n_px) { c[3 * i] = red_y(px[3 * p]); c[3 * i + 1] = red_c(px[3 * p + 1]); c[3 * i + 2] = red_c(px[3 * p + 2]); }
        else { c[3 * i] = 0; c[3 * i + 1] = 40; c[3 * i + 2] = 40; }     // PixelYCbCrQuant{} pad (OLD:730)
    }
    uint32_t s[9]; px2_to_word(c, s);
#pragma unroll
    for (int i = 0; i < 9; ++i) words[9 * w + i] = (uint8_t)s[i];
}

__global__ __launch_bounds__(256) void unpack_words_kernel(const uint8_t* __restrict__ words, uint64_t n_words, uint16_t* __restrict__ px) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = words[9 * w + i] % 27u;            // unpack3 reduces each digit (OLD:28-31)
    // inverse of px2_to_word; trit 26 (= c[8]/9) is ignored (OLD:716-721)
    const uint32_t Y0 = c[0] + 27u * (c[1] % 9u);
    const uint32_t B0 = c[1] / 9u + 3u * c[2];
    const uint32_t R0 = c[3] + 27u * (c[4] % 3u);
    const uint32_t Y1 = c[4] / 3u + 9u * c[5];
    const uint32_t B1 = c[6] + 27u * (c[7] % 3u);
    const uint32_t R1 = c[7] / 3u + 9u * (c[8] % 9u);
    uint16_t* o = px + 6 * w;
    o[0] = (uint16_t)Y0; o[1] = (uint16_t)(int16_t)((int)B0 - 40); o[2] = (uint16_t)(int16_t)((int)R0 - 40);
    o[3] = (uint16_t)Y1; o[4] = (uint16_t)(int16_t)((int)B1 - 40); o[5] = (uint16_t)(int16_t)((int)R1 - 40);
}

// ---------------------------------------------------------------------------------------------------------
// K2 : fused encode
// ---------------------------------------------------------------------------------------------------------
// SWAR reduction mod 3 of five 6-bit fields (each <= 63) to {0,1,2}: 4 == 1 (mod 3) so fold the high bits down.
__device__ __forceinline__ uint32_t mod3x5(uint32_t x) {
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x0F3CF3CFu);   // <= 3 + 15
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x030C30C3u);   // <= 3 + 3   (x <= 15 -> x>>2 <= 3)
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x01041041u);   // <= 3
    const uint32_t t = x & (x >> 1) & 0x01041041u;      // fields equal to 3
    return x - (t | (t << 1));
}

template <int R> struct LutGeo;
template <> struct LutGeo<2> { static constexpr int SLAB = 384, NDW = 3, SCR_SHIFT = 16; };
template <> struct LutGeo<4> { static constexpr int SLAB = 512, NDW = 3, SCR_SHIFT = 0; };
template <> struct LutGeo<6> { static constexpr int SLAB = 512, NDW = 4, SCR_SHIFT = 18; };
template <> struct LutGeo<8> { static constexpr int SLAB = 768, NDW = 5, SCR_SHIFT = 0; };

// One RS block: K data symbols read at stride 9 from the stream-ordered LDS symbol buffer, parity through the
// per-position LUT (two or three conflict-free ds_read_b64 per symbol), scrambling folded into the same reads.
//   sym_addr : LDS byte address of the block's first data symbol
//   lut      : LDS byte address of the band's LUT
//   c0       : scrambler cycle phase of the block's first body symbol ((i0 - 2) mod 6)
//   first    : block starts at body symbol 0 (the two pre-period states apply)
//   o[7]     : 26 output bytes, little-endian packed (o[6] holds 2)
template <int R>
__device__ __forceinline__ void encode_block(uint32_t sym_addr, uint32_t lut, uint32_t c0, bool first, const EncArgs& a, uint32_t* o) {
    constexpr int K = 26 - R;
    using G = LutGeo<R>;
    // scrambler state per residue class of the position (cycle is 6-periodic)
    uint32_t st[6], sh[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        st[q] = (a.cyc24 >> (2u * (c0 + q))) & 3u;
        sh[q] = (uint32_t)G::SCR_SHIFT + 5u * (st[q] >> 1);   // T1 at SCR_SHIFT, T2 at SCR_SHIFT+5; unused when st==0
    }
    uint32_t acc[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 7; ++i) o[i] = 0;
    uint32_t d01[2] = {0, 0};
#pragma unroll
    for (int p = 0; p < K; ++p) {
        const uint32_t d = lds[sym_addr + 9 * p];
        if (p < 2) d01[p] = d;
        const uint32_t e = lut + (uint32_t)(p * G::SLAB) + d * 8u;
        const uint2 A = *(const uint2*)(lds + e);
        acc[0] += A.x; acc[1] += A.y;
        uint32_t scr;
        if constexpr (R == 2) {
            const uint32_t B = *(const uint32_t*)(lds + lut + (uint32_t)(p * G::SLAB) + 256u + d * 4u);
            acc[2] += B; scr = B;
        } else {
            const uint2 B = *(const uint2*)(lds + e + 256u);
            acc[2] += B.x;
            if constexpr (R == 4) scr = B.y;
            if constexpr (R == 6) { acc[3] += B.y; scr = B.y; }
            if constexpr (R == 8) {
                acc[3] += B.y;
                const uint2 C = *(const uint2*)(lds + e + 512u);
                acc[4] += C.x; scr = C.y;
            }
        }
        const uint32_t v = (scr >> sh[p % 6]) & 31u;
        const uint32_t outp = st[p % 6] == 0 ? d : v;
        o[p >> 2] |= outp << (8 * (p & 3));
    }
    // parity symbols get their own scrambler states: add them to the trit fields before the mod-3 fold
    constexpr int NMAIN = R < 5 ? R : 5;
    uint32_t cm = 0;
#pragma unroll
    for (int j = 0; j < NMAIN; ++j) cm |= st[(K + j) % 6] << (6 * j);
    const uint32_t x0 = mod3x5(acc[0] + cm), x1 = mod3x5(acc[1] + cm), x2 = mod3x5(acc[2] + cm);
    const uint32_t S = x0 + 3u * x1 + 9u * x2;           // five parity symbols in 6-bit fields
    uint32_t par[8];
#pragma unroll
    for (int j = 0; j < NMAIN; ++j) par[j] = (S >> (6 * j)) & 63u;
    if constexpr (R == 6) {
        const uint32_t x3 = mod3x5(acc[3] + st[(K + 5) % 6] * 0x1041u);
        par[5] = (x3 & 63u) + 3u * ((x3 >> 6) & 63u) + 9u * ((x3 >> 12) & 63u);
    }
    if constexpr (R == 8) {
        const uint32_t s5 = st[(K + 5) % 6], s6 = st[(K + 6) % 6], s7 = st[(K + 7) % 6];
        const uint32_t x3 = mod3x5(acc[3] + s5 * 0x1041u + s6 * 0x1040000u);
        const uint32_t x4 = mod3x5(acc[4] + s6 + s7 * 0x41040u);
        par[5] = (x3 & 63u) + 3u * ((x3 >> 6) & 63u) + 9u * ((x3 >> 12) & 63u);
        par[6] = ((x3 >> 18) & 63u) + 3u * ((x3 >> 24) & 63u) + 9u * (x4 & 63u);
        par[7] = ((x4 >> 6) & 63u) + 3u * ((x4 >> 12) & 63u) + 9u * ((x4 >> 18) & 63u);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) { const int p = K + j; o[p >> 2] |= par[j] << (8 * (p & 3)); }
    if (first) {
        // body symbols 0 and 1 see the pre-period states (exact whatever the seed; OLD:81-87)
        const uint32_t pre[2] = {a.pre0, a.pre1};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t d = d01[p], s = pre[p];
            const uint32_t t0 = (d % 3u + s) % 3u, t1 = ((d / 3u) % 3u + s) % 3u, t2 = (d / 9u + s) % 3u;
            o[0] = (o[0] & ~(0xFFu << (8 * p))) | ((t0 + 3u * t1 + 9u * t2) << (8 * p));
        }
    }
}

// 2-D boustrophedon position map (an involution inside each row segment; OLD:750-780)
__device__ __forceinline__ uint32_t il_perm(uint32_t u, const EncArgs& a) {
    const uint32_t chunk = fdiv(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdiv(rem, a.div_w), c = rem - r * a.il_w;
    const uint32_t rowlen = min(a.il_w, take - r * a.il_w);
    return base + r * a.il_w + ((r & 1u) ? rowlen - 1u - c : c);
}
__device__ __forceinline__ uint32_t il_row_start(uint32_t u, const EncArgs& a) {
    const uint32_t chunk = fdiv(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    return base + fdiv(rem, a.div_w) * a.il_w;
}
__device__ __forceinline__ uint32_t il_row_end(uint32_t u, const EncArgs& a) {   // one past the last symbol of u's row segment
    const uint32_t chunk = fdiv(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdiv(rem, a.div_w);
    return base + r * a.il_w + min(a.il_w, take - r * a.il_w);
}

template <int FE>
__global__ __launch_bounds__(1024) void encode_kernel(const EncArgs a) {
    constexpr uint32_t GS = FE == FE_PIXELS ? kGroupSyms : kGroupSymsW;      // symbols per lane group
    constexpr uint32_t GB = FE == FE_PIXELS ? 72u : 108u;                     // input bytes per lane group
    constexpr uint32_t UB = FE == FE_PIXELS ? 6u : 9u;                        // bytes per input unit
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t TS = 9u * a.Lq;

    // LUT images -> LDS once per (persistent) workgroup
    for (uint32_t i = tid * 16u; i < a.lut_bytes; i += nthr * 16u)
        *(uint4*)(lds + i) = *(const uint4*)((const uint8_t*)a.lut_img + i);

    if (blockIdx.x == 0 && a.frame_out) {                                    // header symbols + zero tail (OLD:1159-1167)
        if (tid < a.hdr_syms) a.frame_out[tid] = a.hdr[tid];
        if (tid < a.pad_bytes) a.frame_out[a.out_syms + tid] = 0;
    }
    __syncthreads();

    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint32_t S0 = tile * TS;
        // ---------------- phase 1: input -> stream-ordered symbols in LDS ----------------
        uint32_t u_lo = S0, u_hi = S0 + TS;                                   // pre-interleave symbols this tile needs
        if (a.il_on) {
            for (uint32_t i = tid * 16u; i < TS; i += nthr * 16u) *(uint4*)(lds + a.sym_off + i) = make_uint4(0, 0, 0, 0);
            const uint32_t hi = min(S0 + TS, a.n_sym);
            if (S0 < hi) { u_lo = il_row_start(S0, a); u_hi = il_row_end(hi - 1u, a); } else u_hi = u_lo;
            __syncthreads();
        }
        const uint32_t g_lo = u_lo / GS, g_hi = (u_hi + GS - 1u) / GS;
        for (uint32_t gc = g_lo; gc < g_hi; gc += a.stage_groups) {
            const uint32_t gc_hi = min(g_hi, gc + a.stage_groups);
            const uint64_t b0 = ((uint64_t)gc * GB) & ~15ull, b1 = (uint64_t)gc_hi * GB;
            const uint64_t real = a.n_units * UB;
            const uint32_t n_chunks = (uint32_t)((b1 - b0 + 15u) >> 4);
            for (uint32_t i = tid; i < n_chunks; i += nthr) {                  // coalesced 16-B row loads
                const uint64_t o = b0 + 16ull * i;
                uint4 v;
                if (o + 16u <= real) v = *(const uint4*)(a.in + o);
                else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    if constexpr (FE == FE_PIXELS) {
#pragma unroll
                        for (int h = 0; h < 8; ++h) {                             // shorts past the data: pad pixel, then zero-trit pixels
                            const uint64_t n = (o >> 1) + h, px = n / 3u; const uint32_t comp = (uint32_t)(n - 3u * px);
                            uint32_t val;
                            if (px < a.n_units) val = *(const uint16_t*)(a.in + 2u * n);
                            else if (px < a.n_units_pad) val = 0u;
                            else val = comp == 0 ? 0u : 0xFFD8u;                  // -40 -> Cb+40 = 0
                            w[h >> 1] |= val << (16 * (h & 1));
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < 16; ++h) { const uint64_t n = o + h; if (n < real) w[h >> 2] |= (uint32_t)a.in[n] << (8 * (h & 3)); }
                    }
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
                *(uint4*)(lds + a.stage_off + 16u * i) = v;
            }
            __syncthreads();
            for (uint32_t g = gc + tid; g < gc_hi; g += nthr) {
                const uint32_t src = a.stage_off + (uint32_t)((uint64_t)g * GB - b0);
                uint32_t s[GS];
                if constexpr (FE == FE_PIXELS) {
                    uint32_t raw[18];
#pragma unroll
                    for (int i = 0; i < 9; ++i) { const uint2 t = *(const uint2*)(lds + src + 8 * i); raw[2 * i] = t.x; raw[2 * i + 1] = t.y; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t c[9];
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            const int n = 9 * q + i; const uint32_t h = (raw[n >> 1] >> (16 * (n & 1))) & 0xFFFFu;
                            c[i] = (i % 3 == 0) ? red_y(h) : red_c(h);
                        }
                        px3_to_sym13(c, s + 13 * q);
                    }
                } else {
                    uint32_t raw[27];
#pragma unroll
                    for (int i = 0; i < 27; ++i) raw[i] = *(const uint32_t*)(lds + src + 4 * i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t c[27];
#pragma unroll
                        for (int i = 0; i < 27; ++i) { const int n = 27 * q + i; c[i] = ((raw[n >> 2] >> (8 * (n & 3))) & 0xFFu) % 27u; }
                        w3_to_sym26(c, s + 26 * q);
                    }
                }
                const uint32_t u0 = g * GS;
                if (!a.il_on && u0 >= S0 && u0 + GS <= S0 + TS) {               // whole group lands in the tile: dword stores
                    const uint32_t dst = a.sym_off + (u0 - S0);
#pragma unroll
                    for (uint32_t i = 0; i < GS / 4; ++i)
                        *(uint32_t*)(lds + dst + 4 * i) = s[4 * i] | s[4 * i + 1] << 8 | s[4 * i + 2] << 16 | s[4 * i + 3] << 24;
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < GS; ++i) {
                        uint32_t u = u0 + i;
                        if (a.il_on) { if (u >= a.n_sym) continue; u = il_perm(u, a); }
                        if (u >= S0 && u < S0 + TS) lds[a.sym_off + (u - S0)] = (uint8_t)s[i];
                    }
                }
            }
            __syncthreads();
        }

        // ---------------- phase 2: one lane = two consecutive RS blocks of one band ----------------
        if (wave < a.n_waves) {
            const uint32_t b = a.wave_band[wave], t = a.wave_pair0[wave] + lane;
            const uint32_t k = a.band_k[b], nbt = a.band_nb_tile[b];
            const uint32_t m0 = 2u * t, mg0 = tile * nbt + m0;
            if (m0 < nbt && mg0 < a.band_blocks[b]) {
                const uint32_t lut = a.band_lut_off[b];
                const uint32_t sa = a.sym_off + b + 9u * k * m0;
                const uint32_t cA = (a.band_boff6[b] + 2u * (mg0 % 3u)) % 6u, cB = (cA + 2u) % 6u;   // 26 == 2 (mod 6)
                const bool first = (a.band_body_off[b] == 0) && (mg0 == 0);
                uint32_t oA[7], oB[7];
                switch (k) {
                    case 24: encode_block<2>(sa, lut, cA, first, a, oA); encode_block<2>(sa + 9u * 24u, lut, cB, false, a, oB); break;
                    case 22: encode_block<4>(sa, lut, cA, first, a, oA); encode_block<4>(sa + 9u * 22u, lut, cB, false, a, oB); break;
                    case 20: encode_block<6>(sa, lut, cA, first, a, oA); encode_block<6>(sa + 9u * 20u, lut, cB, false, a, oB); break;
                    default: encode_block<8>(sa, lut, cA, first, a, oA); encode_block<8>(sa + 9u * 18u, lut, cB, false, a, oB); break;
                }
                uint32_t w13[13];
#pragma unroll
                for (int i = 0; i < 6; ++i) w13[i] = oA[i];
                w13[6] = (oA[6] & 0xFFFFu) | (oB[0] << 16);
#pragma unroll
                for (int i = 0; i < 6; ++i) w13[7 + i] = (oB[i] >> 16) | (oB[i + 1] << 16);
                // staging image is placed so that LDS address == global address (mod 16): the copy-out is then
                // aligned on both sides.  Global run starts are even, so dword stores are either aligned or off by 2.
                const uint64_t gaddr = (uint64_t)(uintptr_t)a.body_out + a.band_body_off[b] + 26ull * ((uint64_t)tile * nbt);
                const uint32_t shift = (uint32_t)gaddr & 15u;
                const uint32_t dst = a.band_out_off[b] + shift + 52u * t;
                if ((shift & 2u) == 0) {
#pragma unroll
                    for (int i = 0; i < 13; ++i) *(uint32_t*)(lds + dst + 4 * i) = w13[i];
                } else {
                    *(uint16_t*)(lds + dst) = (uint16_t)w13[0];
#pragma unroll
                    for (int i = 0; i < 12; ++i) *(uint32_t*)(lds + dst + 2 + 4 * i) = (w13[i] >> 16) | (w13[i + 1] << 16);
                    *(uint16_t*)(lds + dst + 50) = (uint16_t)(w13[12] >> 16);
                }
            }
        }
        __syncthreads();

        // ---------------- phase 3: coalesced copy-out, one band run per wave ----------------
        const uint32_t nwv = nthr >> 6;
        for (uint32_t b = wave; b < 9; b += nwv) {
            const uint32_t nbt = a.band_nb_tile[b];
            const uint64_t first_blk = (uint64_t)tile * nbt;
            if (first_blk >= a.band_blocks[b]) continue;
            const uint32_t nvalid = (uint32_t)min((uint64_t)nbt, (uint64_t)a.band_blocks[b] - first_blk);
            const uint32_t Rb = 26u * nvalid;
            uint8_t* g = a.body_out + a.band_body_off[b] + 26ull * first_blk;
            const uint32_t shift = (uint32_t)(uintptr_t)g & 15u;
            const uint32_t src = a.band_out_off[b] + shift;
            const uint32_t head = min(Rb, (16u - shift) & 15u);
            if (2u * lane < head) *(uint16_t*)(g + 2u * lane) = *(const uint16_t*)(lds + src + 2u * lane);
            const uint32_t nmain = (Rb - head) >> 4;
            for (uint32_t i = lane; i < nmain; i += 64u)
                *(uint4*)(g + head + 16u * i) = *(const uint4*)(lds + src + head + 16u * i);
            const uint32_t done = head + 16u * nmain, tail = Rb - done;
            if (2u * lane < tail) *(uint16_t*)(g + done + 2u * lane) = *(const uint16_t*)(lds + src + done + 2u * lane);
        }
        __syncthreads();
    }
}

template __global__ void encode_kernel<FE_PIXELS>(const EncArgs);
template __global__ void encode_kernel<FE_WORDS>(const EncArgs);

// ---------------------------------------------------------------------------------------------------------
// beacon insertion pass (OLD:1118-1141): gather, one lane per framed byte
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beacon_kernel(const BeaconArgs a) {
    const uint64_t q0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q0 == 0) {
        for (uint32_t i = 0; i < a.hdr_syms; ++i) a.frame_out[i] = a.hdr[i];
        for (uint32_t i = 0; i < a.pad_bytes; ++i) a.frame_out[a.hdr_syms + a.framed_syms + i] = 0;
    }
    for (uint64_t q = q0; q < a.framed_syms; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = q / 9u; const uint32_t slot = (uint32_t)(q - 9u * w);
        const bool hitw = (w % a.period) == 0;
        uint8_t v;
        if (hitw && slot == a.slot) v = (uint8_t)a.sym;
        else {
            uint64_t nb = 0;                                   // beacons strictly before q
            if (a.slot < 9u) nb = (w + a.period - 1u) / a.period + ((hitw && a.slot < slot) ? 1u : 0u);
            const uint64_t kq = q - nb;
            v = kq < a.body_syms ? a.body[kq] : 0;
        }
        a.frame_out[a.hdr_syms + q] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// block-level RS encode (RSCodec::encode_block OLD:517-535): out = data || data * P
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rs_encode_blocks_kernel(const uint8_t* __restrict__ data, uint64_t n_blocks, int k,
                                                               const uint8_t* __restrict__ P /*k x r*/, const RsTables* __restrict__ tab,
                                                               uint8_t* __restrict__ code) {
    __shared__ uint8_t sP[24 * 8]; __shared__ RsTables sT;
    const int r = 26 - k;
    for (int i = threadIdx.x; i < k * r; i += blockDim.x) sP[i] = P[i];
    for (int i = threadIdx.x; i < (int)sizeof(RsTables); i += blockDim.x) ((uint8_t*)&sT)[i] = ((const uint8_t*)tab)[i];
    __syncthreads();
    const uint64_t blk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks) return;
    uint8_t par[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint8_t* d = data + blk * k; uint8_t* o = code + blk * 26;
    for (int i = 0; i < k; ++i) {
        const uint8_t di = d[i] % 27u; o[i] = d[i];
        for (int j = 0; j < r; ++j) par[j] = sT.add[par[j] * 27 + sT.mul[di * 27 + sP[i * r + j]]];
    }
    for (int j = 0; j < r; ++j) o[k + j] = par[j];
}

}  // namespace t3
