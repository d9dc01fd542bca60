// t3_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Word27 encode path.  Byte/trit permutation plus GF(3)-linear
// small-field arithmetic; the RS parity of the frame kernels runs on v_mfma_i32_32x32x32_i8 (phase2_mfma below); bound: HBM.
//
//   K1 pack_pixels_kernel     pixels -> raw Word27                       (encode_raw_pixels_to_words OLD:723-734)
//   K5 unpack_words_kernel    raw Word27 -> pixels                       (decode_raw_words_to_pixels OLD:735-747)
//   K2 encode_kernel<FE>      pixels|raw words -> coded band-serial body (encode_profile_from_raw OLD:1043-1169)
//      beacon_kernel          sparse beacon insertion pass               (OLD:1118-1141)
//      rs_encode_blocks_kernel  block-level RSCodec::encode_block        (OLD:517-535)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "t3_device.h"
#include "t3_rs_core.h"

namespace t3 {

extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
// LDS access by absolute byte address.  The kernels own the whole LDS allocation (no static __shared__), so the dynamic
// array starts at address 0; going through `lds + x` instead makes the compiler add that (link-time) zero to every address.
#define T3_LDS_PTR(T, a) ((const __attribute__((address_space(3))) T*)(uintptr_t)(a))
#define T3_LDS_WPTR(T, a) ((__attribute__((address_space(3))) T*)(uintptr_t)(a))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint32_t lds_u8(uint32_t a)  { return *T3_LDS_PTR(uint8_t, a); }
__device__ __forceinline__ uint32_t lds_u32(uint32_t a) { return *T3_LDS_PTR(uint32_t, a); }

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const DevDiv& d) { return d.d <= 1 ? n : (__umulhi(n, d.mul) >> d.sh); }
// ... for divisors known to be >= 2 (2-D geometry: the host takes rows of one symbol, where the map is the identity, as 1-D).  No test on d:
// hoisted out of the tile loop that test lived in a register pair the kernels did not have, was parked in a VGPR, spilled, and its reload
// (scratch_load + s_waitcnt vmcnt(0)) drained the next tile's prefetch in the middle of phase 1.
__device__ __forceinline__ uint32_t fdiv2(uint32_t n, const DevDiv& d) { return __umulhi(n, d.mul) >> d.sh; }

// ---------------------------------------------------------------------------------------------------------
// small-integer division by powers of three with full-rate 24-bit multiplies (ranges checked in tests/test_host_logic.py)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t div3(uint32_t x)  { return __umul24(x, 171u) >> 9; }    // x < 512
__device__ __forceinline__ uint32_t div9(uint32_t x)  { return __umul24(x, 228u) >> 11; }   // x < 512
__device__ __forceinline__ uint32_t div27(uint32_t x) { return __umul24(x, 152u) >> 12; }   // x < 512
__device__ __forceinline__ uint32_t div81(uint32_t x) { return __umul24(x, 405u) >> 15; }   // x < 885
__device__ __forceinline__ uint32_t mod3(uint32_t x)  { return x - 3u * div3(x); }
__device__ __forceinline__ uint32_t mod9(uint32_t x)  { return x - 9u * div9(x); }
__device__ __forceinline__ uint32_t mod27(uint32_t x) { return x - 27u * div27(x); }

// Components as the reference's i2tr sees them: v % 3^w of the uint32 cast (no clamping, OLD:675-682,697-702).
// 16-bit operands: floor(x/d) = floor((x + 0.5) * fl(1/d)) exactly for x < 65536 (the +0.5 keeps the product
// >= 0.5/d away from every integer, far more than the float rounding error).
__device__ __forceinline__ uint32_t red_y(uint32_t y16) {
    const uint32_t q = (uint32_t)(((float)y16 + 0.5f) * (1.0f / 243.0f));
    return y16 - __umul24(q, 243u);
}
__device__ __forceinline__ uint32_t red_c(uint32_t c16) {
    const int32_t v = (int32_t)(int16_t)c16 + 40;
    // negative v: (2^32 + v) % 81 = (v + 49 + 81*405) % 81, and v + 32854 > 0 for every int16
    const uint32_t x = v < 0 ? (uint32_t)(v + 32854) : (uint32_t)v;
    const uint32_t q = (uint32_t)(((float)x + 0.5f) * (1.0f / 81.0f));
    return x - __umul24(q, 81u);
}

// 3 pixels = 39 trits = 13 symbols (trit t of the stream = trit t%13 of pixel t/13; Y:5, Cb+40:4, Cr+40:4).
// Every symbol is a div/mod-by-power-of-3 splice of at most two components — no per-trit work.
__device__ __forceinline__ void px3_to_sym13(const uint32_t* c /*9 reduced comps*/, uint32_t* s /*13*/) {
    const uint32_t Y0 = c[0], B0 = c[1], R0 = c[2], Y1 = c[3], B1 = c[4], R1 = c[5], Y2 = c[6], B2 = c[7], R2 = c[8];
    uint32_t q;
    q = div27(Y0); s[0] = Y0 - 27u * q;            s[1] = q + 9u * mod3(B0);
    s[2] = div3(B0);
    q = div27(R0); s[3] = R0 - 27u * q;            s[4] = q + 3u * mod9(Y1);
    s[5] = div9(Y1);
    q = div27(B1); s[6] = B1 - 27u * q;            s[7] = q + 3u * mod9(R1);
    s[8] = div9(R1) + 9u * mod3(Y2);
    s[9] = mod27(div3(Y2));
    s[10] = div81(Y2) + 3u * mod9(B2);
    s[11] = div9(B2) + 9u * mod3(R2);
    s[12] = div3(R2);
}

// 3 raw words (27 canonical symbols, trit 26 of each dropped, OLD:1065-1076) = 78 trits = 26 symbols.
__device__ __forceinline__ void w3_to_sym26(const uint32_t* c /*27 symbols < 27*/, uint32_t* s /*26*/) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = c[i];
    s[8] = mod9(c[8]) + 9u * mod3(c[9]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[9 + i] = div3(c[9 + i]) + 9u * mod3(c[10 + i]);
    s[17] = mod3(div3(c[17])) + 3u * mod9(c[18]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[18 + i] = div9(c[18 + i]) + 3u * mod9(c[19 + i]);
}

// ---------------------------------------------------------------------------------------------------------
// K1 / K5 : RAW packer (2 pixels <-> 9 symbols), one lane per word
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void px2_to_word(const uint32_t* c /*6 reduced comps*/, uint32_t* s /*9*/) {
    // T[0..4]=Ya T[5..8]=Cba T[9..12]=Cra T[13..17]=Yb T[18..21]=Cbb T[22..25]=Crb T[26]=0 (OLD:693-705)
    const uint32_t Y0 = c[0], B0 = c[1], R0 = c[2], Y1 = c[3], B1 = c[4], R1 = c[5];
    s[0] = mod27(Y0); s[1] = div27(Y0) + 9u * mod3(B0); s[2] = div3(B0); s[3] = mod27(R0);
    s[4] = div27(R0) + 3u * mod9(Y1); s[5] = div9(Y1); s[6] = mod27(B1); s[7] = div27(B1) + 3u * mod9(R1);
    s[8] = div9(R1);
}

// One lane = four words: 48 bytes of pixels in (three 16-byte loads), 36 bytes out (nine dwords); the frame's last lanes and
// unaligned buffers go element by element.
__device__ __forceinline__ void pack_one(const uint16_t* h /*6 components; nullptr-free*/, uint32_t n_valid_px, uint32_t* s) {
    uint32_t c[6];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if ((uint32_t)i < n_valid_px) { c[3 * i] = red_y(h[3 * i]); c[3 * i + 1] = red_c(h[3 * i + 1]); c[3 * i + 2] = red_c(h[3 * i + 2]); }
        else { c[3 * i] = 0; c[3 * i + 1] = 40; c[3 * i + 2] = 40; }     // PixelYCbCrQuant{} pad (OLD:730)
    }
    px2_to_word(c, s);
}
__global__ __launch_bounds__(256) void pack_pixels_kernel(const uint16_t* __restrict__ px, uint64_t n_px, uint8_t* __restrict__ words, uint64_t n_words) {
    const uint64_t w0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (w0 >= n_words) return;
    if (2 * w0 + 8 <= n_px && w0 + 4 <= n_words && (((uintptr_t)px | (uintptr_t)words) & 15u) == 0) {
        uint32_t in[12];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const uint4 v = *(const uint4*)(px + 6 * w0 + 8 * k); in[4 * k] = v.x; in[4 * k + 1] = v.y; in[4 * k + 2] = v.z; in[4 * k + 3] = v.w; }
        uint32_t o[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            uint16_t h[6];
#pragma unroll
            for (uint32_t i = 0; i < 6; ++i) { const uint32_t e = 6u * q + i; h[i] = (uint16_t)(in[e >> 1] >> (16u * (e & 1u))); }
            uint32_t sy[9]; pack_one(h, 2, sy);
#pragma unroll
            for (uint32_t i = 0; i < 9; ++i) { const uint32_t b = 9u * q + i; o[b >> 2] |= (sy[i] & 0xFFu) << (8u * (b & 3u)); }
        }
        uint32_t* d = (uint32_t*)(words + 9 * w0);
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] = o[k];
        return;
    }
    for (uint64_t w = w0; w < min(w0 + 4, n_words); ++w) {
        uint16_t h[6] = {0, 0, 0, 0, 0, 0};
        const uint32_t nv = (uint32_t)min((uint64_t)2, n_px > 2 * w ? n_px - 2 * w : 0ull);
        for (uint32_t i = 0; i < 3u * nv; ++i) h[i] = px[6 * w + i];
        uint32_t sy[9]; pack_one(h, nv, sy);
        for (int i = 0; i < 9; ++i) words[9 * w + i] = (uint8_t)sy[i];
    }
}

__device__ __forceinline__ void unpack_one(const uint32_t* craw, uint16_t* o) {
    uint32_t c[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) c[i] = mod27(craw[i]);                     // unpack3 reduces each digit (OLD:28-31)
    // inverse of px2_to_word; trit 26 (= c[8]/9) is ignored (OLD:716-721)
    const uint32_t Y0 = c[0] + 27u * mod9(c[1]);
    const uint32_t B0 = div9(c[1]) + 3u * c[2];
    const uint32_t R0 = c[3] + 27u * mod3(c[4]);
    const uint32_t Y1 = div3(c[4]) + 9u * c[5];
    const uint32_t B1 = c[6] + 27u * mod3(c[7]);
    const uint32_t R1 = div3(c[7]) + 9u * mod9(c[8]);
    o[0] = (uint16_t)Y0; o[1] = (uint16_t)(int16_t)((int)B0 - 40); o[2] = (uint16_t)(int16_t)((int)R0 - 40);
    o[3] = (uint16_t)Y1; o[4] = (uint16_t)(int16_t)((int)B1 - 40); o[5] = (uint16_t)(int16_t)((int)R1 - 40);
}
// One lane = four words: nine dwords in, three 16-byte stores of pixels out.
__global__ __launch_bounds__(256) void unpack_words_kernel(const uint8_t* __restrict__ words, uint64_t n_words, uint16_t* __restrict__ px) {
    const uint64_t w0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (w0 >= n_words) return;
    if (w0 + 4 <= n_words && (((uintptr_t)px | (uintptr_t)words) & 15u) == 0) {
        uint32_t in[9];
        const uint32_t* sdw = (const uint32_t*)(words + 9 * w0);
#pragma unroll
        for (int k = 0; k < 9; ++k) in[k] = sdw[k];
        uint32_t o[12];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            uint32_t c[9];
#pragma unroll
            for (uint32_t i = 0; i < 9; ++i) { const uint32_t b = 9u * q + i; c[i] = (in[b >> 2] >> (8u * (b & 3u))) & 0xFFu; }
            uint16_t h[6]; unpack_one(c, h);
#pragma unroll
            for (uint32_t i = 0; i < 3; ++i) o[3u * q + i] = (uint32_t)h[2 * i] | (uint32_t)h[2 * i + 1] << 16;
        }
        uint4* d = (uint4*)(px + 6 * w0);
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
        return;
    }
    for (uint64_t w = w0; w < min(w0 + 4, n_words); ++w) {
        uint32_t c[9];
        for (int i = 0; i < 9; ++i) c[i] = words[9 * w + i];
        uint16_t h[6]; unpack_one(c, h);
        for (int i = 0; i < 6; ++i) px[6 * w + i] = h[i];
    }
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

#ifndef T3_ENC_CHUNK
#define T3_ENC_CHUNK 5      // data symbols whose LUT reads may be in flight together
#endif
template <int R> struct LutGeo;     // must match lut_geom() in t3_host.cpp
template <> struct LutGeo<2> { static constexpr uint32_t SLAB = 1024, VOFF = 256; };
template <> struct LutGeo<4> { static constexpr uint32_t SLAB = 1024, VOFF = 256; };
template <> struct LutGeo<6> { static constexpr uint32_t SLAB = 1024, VOFF = 256; };
template <> struct LutGeo<8> { static constexpr uint32_t SLAB = 1280, VOFF = 512; };

struct Blk26 { uint32_t w[7]; };   // 26 output bytes, little-endian packed (w[6] holds 2)
struct BandRow { uint32_t k, nbt, blocks, lut_off, pad_, boff6; uint64_t body_off; };   // 32 B, LDS header row b
__device__ __forceinline__ BandRow band_row(uint32_t b) { return *(const BandRow*)(lds + 32u * b); }
__device__ __forceinline__ uint32_t band_first(uint32_t b) { return *(const uint32_t*)(lds + 288u + 4u * b); }

__device__ __forceinline__ uint32_t add13(uint32_t d, uint32_t s) {   // d + (s,s,s) trit-wise (scramble_symbol OLD:81-87)
    const uint32_t q1 = div3(d), q2 = div9(d);
    uint32_t t0 = d - 3u * q1 + s, t1 = q1 - 3u * q2 + s, t2 = q2 + s;
    t0 -= t0 >= 3u ? 3u : 0u; t1 -= t1 >= 3u ? 3u : 0u; t2 -= t2 >= 3u ? 3u : 0u;
    return t0 + 3u * t1 + 9u * t2;
}

// One RS block.  The stream-ordered LDS symbol buffer holds symbols PRE-SCALED by 8 (= the byte offset of the symbol's
// 8-byte LUT entry), so a data symbol costs: one ds_read_u8, then per table one ds_read_b64/b32 whose address is that
// byte plus an immediate — no address arithmetic for the fixed tables.  Parity contributions arrive as 6-bit SWAR trit
// fields (5 per dword; k*2+2 <= 50 < 64, no carries) and are folded mod 3 once per block.  The table that carries the
// last accumulator dword exists in three variants, one per scrambler state, and holds the scrambled image of the symbol
// in its top byte: choosing the variant (one add of a per-lane class base) scrambles, one v_perm_b32 places the byte.
//   sym_addr : LDS byte address of the block's first data symbol;  lut: LDS byte address of the band's LUT
//   c0       : scrambler cycle phase of the block's first body symbol ((i0 - 2) mod 6)
//   first    : block starts at body symbol 0 (the two pre-period states apply)
template <int R, bool FIXED_LUT>
__device__ __forceinline__ Blk26 encode_block(uint32_t sym_addr, uint32_t lut_rt, uint32_t c0, bool first, const EncArgs& a) {
    constexpr uint32_t K = 26 - R;
    using G = LutGeo<R>;
    const uint32_t lut = FIXED_LUT ? (uint32_t)kLdsHdr : lut_rt;       // single-k launches: LUT sits right behind the header
    uint32_t st[6], vb[6];                          // scrambler state per residue class of the position (6-periodic)
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        st[q] = (a.cyc24 >> (2u * (c0 + q))) & 3u;
        vb[q] = (FIXED_LUT ? 0u : lut) + (st[q] << 8);                   // variant tables are 256 B apart
    }
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0;
    Blk26 o;
#pragma unroll
    for (int i = 0; i < 7; ++i) o.w[i] = 0;
    uint32_t d0 = 0, d1 = 0;
#pragma unroll
    for (uint32_t p = 0; p < K; ++p) {
        const uint32_t d8 = lds[sym_addr + 9u * p];
        if (p == 0) d0 = d8;
        if (p == 1) d1 = d8;
        const uint32_t fa = FIXED_LUT ? d8 : d8 + lut;                   // fixed tables: byte offset + immediate
        const uint2 A = *(const uint2*)(lds + fa + (FIXED_LUT ? lut : 0u) + p * G::SLAB);
        acc0 += A.x; acc1 += A.y;
        if constexpr (R == 8) { const uint2 B = *(const uint2*)(lds + fa + (FIXED_LUT ? lut : 0u) + p * G::SLAB + 256u); acc2 += B.x; acc3 += B.y; }
        const uint32_t va = d8 + vb[p % 6];
        uint32_t img;
        if constexpr (R == 6) {
            const uint2 V = *(const uint2*)(lds + va + (FIXED_LUT ? lut : 0u) + p * G::SLAB + G::VOFF);
            acc2 += V.x; acc3 += V.y; img = V.y;
        } else {
            const uint32_t V = *(const uint32_t*)(lds + va + (FIXED_LUT ? lut : 0u) + p * G::SLAB + G::VOFF);
            if constexpr (R == 8) acc4 += V; else acc2 += V;
            img = V;
        }
        // result byte (p&3) <- top byte of img, other bytes kept
        constexpr uint32_t sel[4] = {0x03020107u, 0x03020700u, 0x03070100u, 0x07020100u};
        o.w[p >> 2] = __builtin_amdgcn_perm(img, o.w[p >> 2], sel[p & 3]);
        if (p % T3_ENC_CHUNK == T3_ENC_CHUNK - 1) {
            // bound the LUT reads in flight: without this the compiler issues all reads first and sinks the adds (spills)
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(o.w[p >> 2]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // parity symbols get their own scrambler states: add them to the trit fields before the mod-3 fold
    constexpr int NMAIN = R < 5 ? R : 5;
    uint32_t cm = 0;
#pragma unroll
    for (int j = 0; j < NMAIN; ++j) cm |= st[(K + j) % 6] << (6 * j);
    const uint32_t x0 = mod3x5(acc0 + cm), x1 = mod3x5(acc1 + cm), x2 = mod3x5(acc2 + cm);
    const uint32_t S = x0 + 3u * x1 + 9u * x2;           // five parity symbols in 6-bit fields
    uint32_t par[8];
#pragma unroll
    for (int j = 0; j < NMAIN; ++j) par[j] = (S >> (6 * j)) & 63u;
    if constexpr (R == 6) {
        const uint32_t x3 = mod3x5(acc3 + st[(K + 5) % 6] * 0x1041u);
        par[5] = (x3 & 63u) + 3u * ((x3 >> 6) & 63u) + 9u * ((x3 >> 12) & 63u);
    }
    if constexpr (R == 8) {
        const uint32_t s5 = st[(K + 5) % 6], s6 = st[(K + 6) % 6], s7 = st[(K + 7) % 6];
        const uint32_t x3 = mod3x5(acc3 + s5 * 0x1041u + s6 * 0x1040000u);
        const uint32_t x4 = mod3x5(acc4 + s6 + s7 * 0x41040u);
        par[5] = (x3 & 63u) + 3u * ((x3 >> 6) & 63u) + 9u * ((x3 >> 12) & 63u);
        par[6] = ((x3 >> 18) & 63u) + 3u * ((x3 >> 24) & 63u) + 9u * (x4 & 63u);
        par[7] = ((x4 >> 6) & 63u) + 3u * ((x4 >> 12) & 63u) + 9u * ((x4 >> 18) & 63u);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) { const int p = K + j; o.w[p >> 2] |= par[j] << (8 * (p & 3)); }
    if (first)   // body symbols 0 and 1 see the pre-period states (exact whatever the seed; OLD:81-87)
        o.w[0] = (o.w[0] & 0xFFFF0000u) | add13(d0 >> 3, a.pre0) | (add13(d1 >> 3, a.pre1) << 8);
    return o;
}

// 2-D boustrophedon position map (an involution inside each row segment; OLD:750-780)
__device__ __forceinline__ uint32_t il_perm(uint32_t u, const EncArgs& a) {
    const uint32_t chunk = fdiv2(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdiv2(rem, a.div_w), c = rem - r * a.il_w;
    const uint32_t rowlen = min(a.il_w, take - r * a.il_w);
    return base + r * a.il_w + ((r & 1u) ? rowlen - 1u - c : c);
}
// The same map for a run of consecutive positions: one pair of divisions at the start, then a few compares per step
struct IlCursor {
    uint32_t base, take, rw, c, rowlen, odd;                 // chunk start, chunk size, row start in the chunk, column, row length, row parity
    __device__ __forceinline__ void init(uint32_t u, const EncArgs& a) {
        const uint32_t chunk = fdiv2(u, a.div_A); base = chunk * a.il_A;
        const uint32_t rem = u - base; take = min(a.il_A, a.n_sym - base);
        const uint32_t r = fdiv2(rem, a.div_w); rw = r * a.il_w; c = rem - rw; odd = r & 1u;
        rowlen = min(a.il_w, take - rw);
    }
    __device__ __forceinline__ uint32_t get() const { return base + rw + (odd ? rowlen - 1u - c : c); }
    __device__ __forceinline__ void next(const EncArgs& a, const uint32_t step = 1u) {     // step 4: rows that are multiples of 4
        c += step;
        if (c >= rowlen) {
            c = 0; rw += a.il_w; odd ^= 1u;
            if (rw >= take) { base += a.il_A; take = min(a.il_A, a.n_sym - base); rw = 0; odd = 0; }
            rowlen = min(a.il_w, take - rw);
        }
    }
};
__device__ __forceinline__ uint32_t il_row_start(uint32_t u, const EncArgs& a) {
    const uint32_t chunk = fdiv2(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    return base + fdiv2(rem, a.div_w) * a.il_w;
}
__device__ __forceinline__ uint32_t il_row_end(uint32_t u, const EncArgs& a) {   // one past the last symbol of u's row segment
    const uint32_t chunk = fdiv2(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdiv2(rem, a.div_w);
    return base + r * a.il_w + min(a.il_w, take - r * a.il_w);
}

// Row segment of position u: start, length, parity (odd rows are reversed)
__device__ __forceinline__ void il_row(uint32_t u, const EncArgs& a, uint32_t& rl, uint32_t& rn, uint32_t& odd) {
    const uint32_t chunk = fdiv2(u, a.div_A), base = chunk * a.il_A, rem = u - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdiv2(rem, a.div_w);
    rl = base + r * a.il_w; rn = min(a.il_w, take - r * a.il_w); odd = r & 1u;
}
// The pre-interleave symbols that land in the post-interleave tile [S0, S0 + TS): the map is an involution inside every row
// segment, so whole rows of the tile come from themselves and only the tile's partial first / last row comes from the mirrored
// piece of that row -- at most three runs of consecutive pre-interleave positions (ascending, adjacent ones merged), TS symbols
// in all, whatever the row width.  Positions past the end of the stream map to themselves.
struct IlRuns { uint32_t lo[3], hi[3], plo[3], n; };      // plo: post-interleave position of the run's lowest-placed symbol (its symbols occupy [plo, plo + hi - lo) of the tile)
__device__ __forceinline__ IlRuns il_runs(uint32_t S0, uint32_t TS, const EncArgs& a) {
    IlRuns R; R.n = 0; R.lo[0] = R.lo[1] = R.lo[2] = 0; R.hi[0] = R.hi[1] = R.hi[2] = 0; R.plo[0] = R.plo[1] = R.plo[2] = 0;
    auto push = [&](uint32_t lo, uint32_t hi, uint32_t plo) {                  // (no dynamic indexing: the runs stay in registers)
        if (lo >= hi) return;
        // adjacent runs are merged when their places are adjacent too (identity-placed neighbours; a mirrored piece never is)
        if (R.n == 0u) { R.lo[0] = lo; R.hi[0] = hi; R.plo[0] = plo; R.n = 1u; }
        else if (R.n == 1u) { if (R.hi[0] == lo && R.plo[0] + (R.hi[0] - R.lo[0]) == plo) R.hi[0] = hi; else { R.lo[1] = lo; R.hi[1] = hi; R.plo[1] = plo; R.n = 2u; } }
        else if (R.n == 2u) { if (R.hi[1] == lo && R.plo[1] + (R.hi[1] - R.lo[1]) == plo) R.hi[1] = hi; else { R.lo[2] = lo; R.hi[2] = hi; R.plo[2] = plo; R.n = 3u; } }
        else if (R.hi[2] == lo) R.hi[2] = hi;
    };
    const uint32_t E = min(S0 + TS, a.n_sym);
    if (S0 < E) {
        uint32_t rl0, rn0, od0, rl1, rn1, od1;
        il_row(S0, a, rl0, rn0, od0); il_row(E - 1u, a, rl1, rn1, od1);
        if (rl0 == rl1) push(od0 ? rl0 + rn0 - (E - rl0) : S0, od0 ? rl0 + rn0 - (S0 - rl0) : E, S0);
        else {
            const uint32_t he = rl0 + rn0;
            push(od0 ? rl0 : S0, od0 ? he - (S0 - rl0) : he, S0);
            push(he, rl1, he);
            push(od1 ? rl1 + rn1 - (E - rl1) : rl1, od1 ? rl1 + rn1 : E, rl1);
        }
    }
    push(max(S0, a.n_sym), S0 + TS, max(S0, a.n_sym));
    return R;
}

// Phase 2 for one lane: encode block m of band b and store its 26 bytes straight to the band's run in global memory.
// A block starts 2-byte aligned (header and band offsets are even), so it is exactly six aligned dwords plus one short —
// at the front when the block starts at 2 (mod 4), at the back otherwise: 7 stores per lane, no overlap with the
// neighbour blocks, no LDS staging.  Consecutive lanes hold consecutive blocks, so a wave's seven store instructions
// cover one contiguous 1664-byte run (13 cache lines).
template <int R, bool FIXED_LUT>
__device__ __forceinline__ bool phase2_band(const EncArgs& a, uint32_t symb, uint32_t tile, uint32_t b, uint32_t m, uint32_t nbt) {
    constexpr uint32_t K = 26 - R;
    const uint32_t mg = tile * nbt + m;
    const BandRow r = band_row(b);
    if (!(m < nbt && mg < r.blocks)) return false;
    const uint32_t c0 = (r.boff6 + 2u * (mg % 3u)) % 6u;                             // 26 == 2 (mod 6)
    const Blk26 o = encode_block<R, FIXED_LUT>(symb + b + 9u * K * m, r.lut_off, c0, r.body_off == 0 && mg == 0, a);
    uint8_t* G = a.body_out + r.body_off + 26ull * mg;
    const bool al = ((uint32_t)(uintptr_t)G & 2u) == 0;
    uint32_t* base = (uint32_t*)(G + (al ? 0 : 2));                                    // six aligned dwords
#pragma unroll
    for (int i = 0; i < 6; ++i) base[i] = al ? o.w[i] : ((o.w[i] >> 16) | (o.w[i + 1] << 16));
    *(uint16_t*)(G + (al ? 24 : 0)) = (uint16_t)(al ? o.w[6] : o.w[0]);                // and the remaining short
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Phase 2 on the matrix cores (single-k launches).  RS parity is GF(3)-linear in the data trits: per block a (3r x 3k)
// matrix-vector product mod 3.  One v_mfma_i32_32x32x32_i8 chain (3 K-steps) does it for 32 blocks: the B operand is the
// data, one dword per symbol = its three trits as bytes (byte 3 carries the scrambled symbol and meets a zero matrix
// column), fetched from a 27-entry LDS table per scrambler state; lane (n, h) supplies positions 8s + 4h + d of block n in
// K-step s.  The A operand (the matrix, host-built in the instruction's lane order) sits in 12 VGPRs.  The accumulators
// come back as: lane (n, h) holds the three trit sums of parity symbols h r/2 .. h r/2 + r/2 - 1 of block n; adding the
// scrambler state and folding mod 3 is three byte-table reads per symbol.  A wave does two sets of 32 blocks.
// Output: lane (n, h) owns bytes [8s + 4h, +4) of the block for s = 0..2 (plus bytes 24, 25 for h = 1): three dword
// stores at 2-byte alignment and one short.  Returns the number of global store instructions issued (wave-uniform).
// ---------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct __attribute__((packed, aligned(2))) U32a2 { uint32_t v; };
struct __attribute__((packed, aligned(2))) U128a2 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) U128a1 { uint32_t v[4]; };
constexpr uint32_t kMfmaModOff = 3 * 4096, kMfmaScr = 336;     // must match t3_host.hpp; scrambler dwords sit in the LDS header

// How the (up to) two sets of a call map to blocks: set s covers items item0[s] .. item0[s] + 31 of a run of n_items blocks
// dealt linearly over bands that share k: item -> (band index item / nb, block item % nb); band_tab = LDS address of the
// index -> band bytes (UEP groups) or ~0 for the identity (one k on all nine bands).
struct P2Map { uint32_t item0[2]; uint32_t n_items, nb; DevDiv div_nb; uint32_t band_tab, scr_off; };

template <int R, bool GRP, bool BCN, bool REGEO = false>      // BCN: beacon insertion fused into the stores; GRP: UEP group call (one set, band table, the group's scrambler dwords); else one k on all nine bands (two sets); REGEO: see load()
__device__ __forceinline__ uint32_t phase2_mfma(const EncArgs& a, uint32_t symb, uint32_t tile, uint32_t lane, const v4i (&Afr)[3], const P2Map& M) {
    constexpr uint32_t K = 26 - R, H = R / 2;
    constexpr uint32_t TB = GRP ? kLdsHdrUep : kLdsHdr, MB = TB + kMfmaModOff;
    const uint32_t n = lane & 31u, h = lane >> 5;
    const uint32_t tb3 = __builtin_amdgcn_readfirstlane((tile * M.nb) % 3u);
    // A wave does two sets of 32 blocks.  The table reads of BOTH sets are issued before either set's MFMA chain, so the
    // second set's two dependent LDS round trips hide under the first set's chain and epilogue.
    struct Set { v4i Bv[3]; uint32_t W[3]; uint32_t c0, mg; uint64_t goff; bool valid, first; uint32_t dd0, dd1; };
    auto load = [&](uint32_t set, Set& s) {
        uint32_t nn = n;
        if constexpr (REGEO) asm volatile("" : "+v"(nn));                     // the wave's items never change: hoisted out of the tile loop, both sets' block geometry was kept in (spilled)
                                                                              // registers by the kernels that are over the 80-VGPR budget (reloads + vmcnt(0) in every tile): those recompute it
        const uint32_t item = M.item0[set] + nn;                              // blocks are dealt linearly across the bands (item0 huge: no set)
        const uint32_t bi = min(fdiv(item, M.div_nb), 8u), m = item - bi * M.nb;
        uint32_t b = bi;
        if constexpr (GRP) b = lds_u8(M.band_tab + bi);
        const BandRow r = band_row(b);
        s.mg = tile * M.nb + m;
        s.valid = item < M.n_items && s.mg < r.blocks;                        // lanes without a block run along (reads stay inside LDS) and store nothing
        s.goff = r.body_off + 26ull * s.mg;
        s.first = r.body_off == 0 && s.mg == 0 && h == 0;
        // scrambler phase of the block's first symbol: (boff6 + 2 (mg mod 3)) mod 6 (26 == 2 mod 6), without wide multiplies
        uint32_t m3 = tb3 + m - 3u * ((m * 683u) >> 11); m3 -= m3 >= 3u ? 3u : 0u;   // m < 2048
        uint32_t c0 = r.boff6 + 2u * m3; c0 -= c0 >= 6u ? 6u : 0u;
        s.c0 = c0;
        uint32_t c0h = c0 + 4u * h; c0h -= c0h >= 6u ? 6u : 0u;               // ... of this lane's first position 4h
        const uint32_t cycs = a.cyc24 >> (2u * c0h);
        uint32_t vb[6];                                                       // table base per position class: state (4 KiB apart), own bank copy
#pragma unroll
        for (uint32_t q = 0; q < 6; ++q) vb[q] = (((cycs >> (2u * q)) & 3u) << 12) | (TB + 4u * n);
        const uint32_t sa = symb + b + 9u * K * m + 36u * h;
        s.dd0 = 0; s.dd1 = 0;
#pragma unroll
        for (uint32_t st = 0; st < 3; ++st) {
            uint32_t x[4];
#pragma unroll
            for (uint32_t d = 0; d < 4; ++d) {
                const uint32_t d4 = lds_u8(sa + 72u * st + 9u * d);           // positions >= k read neighbouring bytes: they meet zero matrix columns
                if (set == 0 && st == 0 && d == 0) s.dd0 = d4;                // (body symbols 0 and 1 sit in set 0 of wave 0)
                if (set == 0 && st == 0 && d == 1) s.dd1 = d4;
                x[d] = lds_u32((d4 << 5) + vb[(8u * st + d) % 6u]);
                s.Bv[st][d] = (int)x[d];
            }
            const uint32_t t01 = __builtin_amdgcn_perm(x[1], x[0], 0x0c0c0703u), t23 = __builtin_amdgcn_perm(x[3], x[2], 0x07030c0cu);
            s.W[st] = t01 | t23;                                               // the four scrambled symbols
        }
    };
    auto finish = [&](Set& s) -> uint32_t {
        uint32_t c0K = s.c0 + (K % 6u); c0K -= c0K >= 6u ? 6u : 0u;           // scrambler phase of the first parity symbol
        if constexpr (R >= 4) {     // the states of the parity symbols ride in unused positions of the upper half (see mfma_scr_pos)
            const u32x2 sd = *T3_LDS_PTR(u32x2, (GRP ? M.scr_off : (uint32_t)kMfmaScr) + 8u * c0K);
            if constexpr (R == 4) s.Bv[2][2] = h ? (int)sd.x : s.Bv[2][2];
            else { s.Bv[2][0] = h ? (int)sd.x : s.Bv[2][0]; s.Bv[2][1] = h ? (int)sd.y : s.Bv[2][1]; }
        }
        v16i acc = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};   // bias: trit sums in [-60, 62] -> table index
#ifndef T3_ABL_NO_MFMA
#pragma unroll
        for (uint32_t st = 0; st < 3; ++st) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Afr[st], s.Bv[st], acc, 0, 0, 0);
#else
        acc[0] += s.Bv[0][0] & 1; acc[3] += s.Bv[1][1] & 1; acc[6] += s.Bv[2][2] & 1; acc[1] += (s.Bv[0][1] ^ s.Bv[0][2] ^ s.Bv[0][3] ^ s.Bv[1][0] ^ s.Bv[1][2] ^ s.Bv[1][3] ^ s.Bv[2][0] ^ s.Bv[2][1] ^ s.Bv[2][3]) & 1;
#endif
        // parity symbols of this lane: h H + jj; mod-3 fold and 3^t weight by byte tables (one bank per dword: conflict-free)
        uint32_t Pown = 0;
#pragma unroll
        for (uint32_t jj = 0; jj < H; ++jj) {
            uint32_t x0 = (uint32_t)acc[3 * jj], x1 = (uint32_t)acc[3 * jj + 1], x2 = (uint32_t)acc[3 * jj + 2];
            if constexpr (R == 2) {                                              // k = 24: no free position, add the state here
                const uint32_t stt = (a.cyc24 >> (2u * (c0K + h))) & 3u;
                x0 += stt; x1 += stt; x2 += stt;
            }
            const uint32_t sym = lds_u8(MB + x0) + lds_u8(MB + 128u + x1) + lds_u8(MB + 256u + x2);
            Pown |= sym << (8u * jj);
        }
        const uint32_t Plo = __builtin_amdgcn_permlane32_swap(Pown, Pown, false, false)[0];   // upper half: the h = 0 partner's parities
        uint32_t W2 = s.W[2], tail;
        if constexpr (R == 6)      { W2 = h ? ((Plo & 0x00FFFFFFu) | (Pown << 24)) : W2; tail = Pown >> 8; }
        else if constexpr (R == 8) { W2 = h ? ((Plo >> 16) | (Pown << 16)) : ((W2 & 0xFFFFu) | (Pown << 16)); tail = Pown >> 16; }
        else if constexpr (R == 4) { W2 = h ? ((W2 & 0xFFFFu) | (Plo << 16)) : W2; tail = Pown; }
        else                       { tail = (Plo & 0xFFu) | (Pown << 8); }
        uint32_t W0 = s.W[0];
        if (s.first)                                            // body symbols 0 and 1 see the pre-period states (exact whatever the seed; OLD:81-87)
            W0 = (W0 & 0xFFFF0000u) | add13(s.dd0 >> 2, a.pre0) | (add13(s.dd1 >> 2, a.pre1) << 8);
        // Lane (n, h) holds bytes [8s + 4h, +4) of block n.  Two half-wave exchanges give the lower lane bytes 0..15 and the
        // upper lane bytes 10..25 of the block: ONE 16-byte store per lane covers the 26 bytes (bytes 10..15 are written by
        // both, with the same values), instead of three scattered dwords and a short -- an eighth of the cache-line requests.
        // v_permlane32_swap(a, b): a's upper half-wave <-> b's lower half-wave.  swap(W0, W1) and swap(W1, W2) leave in every lane the
        // four dwords of its 16-byte run in order -- h=0: bytes 0..3, 4..7, 8..11, 12..15; h=1: 8..11, 12..15, 16..19, 20..23 --
        // and one v_perm per dword with a per-lane selector takes them as they are (h=0) or shifted by two bytes (h=1: bytes 10..25).
        const auto q01 = __builtin_amdgcn_permlane32_swap(W0, s.W[1], false, false);
        const auto q23 = __builtin_amdgcn_permlane32_swap(s.W[1], W2, false, false);
        const uint32_t selE = h ? 0x05040302u : 0x03020100u;                      // v_perm(S0, S1): 0..3 = bytes of S1, 4..7 = bytes of S0
        U128a2 E;
        E.v[0] = __builtin_amdgcn_perm(q01[1], q01[0], selE);
        E.v[1] = __builtin_amdgcn_perm(q23[0], q01[1], selE);
        E.v[2] = __builtin_amdgcn_perm(q23[1], q23[0], selE);
        E.v[3] = __builtin_amdgcn_perm(tail, q23[1], selE);                        // h=1: bytes 22..25
        if constexpr (BCN) {
            // The run's 16 body bytes start at body offset g0; nb0 beacons lie in front of it in the framed stream and the next one
            // comes after c more body bytes.  c < 16: it falls inside the run, whose bytes from c on move up by one (the 17th
            // byte goes out on its own); c == bcn_pb with a beacon directly in front of the run (only a block's first run can
            // have no run before it that holds that beacon): this lane writes it.  bcn_pb >= 17: one beacon per run at most.
            const uint32_t g0 = (uint32_t)s.goff + 10u * h;                       // body symbols are 31-bit (plan_layout)
            uint32_t nb0 = 0, c = a.bcn_slot - g0;
            if (g0 >= a.bcn_slot) { const uint32_t u = g0 - a.bcn_slot, j = fdiv2(u, a.bcn_div); nb0 = j + 1u; c = a.bcn_pb - (u - j * a.bcn_pb); }
            const bool inside = c < 16u, pre = nb0 != 0u && c == a.bcn_pb;
            const uint32_t dc = inside ? c >> 2 : 4u, bc = c & 3u;
            const uint32_t Ed = dc == 0u ? E.v[0] : dc == 1u ? E.v[1] : dc == 2u ? E.v[2] : E.v[3];
            const uint32_t sel = bc == 0u ? 0x02010004u : bc == 1u ? 0x02010400u : bc == 2u ? 0x02040100u : 0x04020100u;
            const uint32_t Mx = __builtin_amdgcn_perm(a.bcn_sym, Ed, sel);        // low bc bytes, the beacon, the rest one byte up
            U128a1 F;
            F.v[0] = dc == 0u ? Mx : E.v[0];
#pragma unroll
            for (uint32_t i = 1; i < 4; ++i) F.v[i] = i < dc ? E.v[i] : i == dc ? Mx : __builtin_amdgcn_alignbyte(E.v[i], E.v[i - 1], 3u);
            uint8_t* dst = a.body_out + (s.goff + 10u * h + nb0);
            const bool extra = s.valid && (inside || pre);
            if (s.valid) *(U128a1*)dst = F;                                         // any byte alignment
            const bool any_extra = __builtin_amdgcn_ballot_w64(extra) != 0;
            if (any_extra) { if (extra) *(inside ? dst + 16 : dst - 1) = (uint8_t)(inside ? E.v[3] >> 24 : a.bcn_sym); }
            return (__builtin_amdgcn_ballot_w64(s.valid) != 0 ? 1u : 0u) + (any_extra ? 1u : 0u);
        }
#ifdef T3_ABL_NO_STORE
        if (s.valid && a.n_tiles == 0xFFFFFFFFu)
#else
        if (s.valid)
#endif
            *(U128a2*)(a.body_out + s.goff + 10u * h) = E;                         // 2-byte aligned (measured: as fast as 16-byte aligned)
#ifndef T3_ABL_NO_STORE
        return __builtin_amdgcn_ballot_w64(s.valid) != 0 ? 1u : 0u;               // a store with no active lane is branched over
#else
        return 0u;
#endif
    };
    Set s0;
    load(0, s0);
    if constexpr (GRP) return finish(s0);                                    // UEP path: one set per call
    Set s1;
    load(1, s1);
    uint32_t issued = finish(s0);
    issued += finish(s1);
    return issued;
}

// Stage the input bytes of lane groups [g_lo, g_hi) into the stage buffer at LDS offset `stage`: image byte x = input
// byte b0 + x with b0 = 16-aligned start of group g_lo.  Whole 1-KiB pieces inside the real data go by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, completes behind vmcnt, so the next tile's input streams in under
// this tile's compute); pieces that touch the end of the data are synthesised (pad pixel OLD:730, then zero trits).
#ifndef T3_DMA_AUX
#define T3_DMA_AUX 3   // cache policy of the input LDS-DMA: sc0 | nt (the input is read once; measured 2-3 % over the default policy, profiles/r02/notes.md)
#endif
template <int FE>
__device__ __forceinline__ void stage_input(const EncArgs& a, uint32_t stage, uint32_t g_lo, uint32_t g_hi, uint32_t lane, uint32_t wave, uint32_t nwv) {
    constexpr uint32_t GB = FE == FE_PIXELS ? kGroupBytes : FE == FE_RGB ? kGroupBytesRgb : kGroupBytesW, UB = FE == FE_PIXELS ? 6u : FE == FE_RGB ? 3u : 9u;
    const uint64_t b0 = ((uint64_t)g_lo * GB) & ~15ull, b1 = (uint64_t)g_hi * GB, real = a.n_units * UB;
    const uint32_t n_chunks = (uint32_t)((b1 - b0 + 15u) >> 4);
    for (uint32_t c0 = __builtin_amdgcn_readfirstlane(wave) * 64u; c0 < n_chunks; c0 += nwv * 64u) {
        const uint64_t o = b0 + 16ull * (c0 + lane);
        if (b0 + 16ull * (c0 + 64u) <= real) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t*)(a.in + o),
                                             (__attribute__((address_space(3))) uint32_t*)(lds + stage + 16u * c0), 16, 0, T3_DMA_AUX);
        } else if (c0 + lane < n_chunks) {
            uint32_t w[4] = {0, 0, 0, 0};
            if (o + 16u <= real) { const uint4 v = *(const uint4*)(a.in + o); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
            else if constexpr (FE == FE_PIXELS) {
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    const uint64_t n = (o >> 1) + h, px = n / 3u; const uint32_t comp = (uint32_t)(n - 3u * px);
                    uint32_t val;
                    if (px < a.n_units) val = *(const uint16_t*)(a.in + 2u * n);
                    else if (px < a.n_units_pad) val = 0u;
                    else val = comp == 0 ? 0u : 0xFFD8u;                          // -40 -> Cb+40 = 0
                    w[h >> 1] |= val << (16 * (h & 1));
                }
            } else {                                                              // raw words, RGB: bytes past the end read as zero (RGB black = the pad pixel)
#pragma unroll
                for (int h = 0; h < 16; ++h) { const uint64_t n = o + h; if (n < real) w[h >> 2] |= (uint32_t)a.in[n] << (8 * (h & 3)); }
            }
            *(uint4*)(lds + stage + 16u * (c0 + lane)) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Phase 1: lane groups [g_lo, g_hi) of the stage buffer -> stream-ordered symbols [S0, S0+TS) in LDS.
template <int FE, bool IL, int SH>     // SH: symbols are stored pre-scaled by 2^SH (the byte offset of their table entry)
__device__ __forceinline__ void convert_groups(const EncArgs& a, uint32_t stage, uint32_t g_base, uint32_t g_lo, uint32_t g_hi,
                                               uint32_t S0, uint32_t TS, uint32_t tid, uint32_t nthr) {
    constexpr uint32_t GS = FE == FE_PIXELS ? kGroupSyms : kGroupSymsW, GB = FE == FE_PIXELS ? kGroupBytes : kGroupBytesW;
    constexpr uint32_t QS = GS / 2;                                        // 3 px -> 13 symbols, 3 words -> 26 symbols
    constexpr uint32_t EM = FE == FE_PIXELS ? 2u : 4u;                     // bytes per LDS store on the fast path (26 g is 2-aligned)
    const uint64_t b0 = ((uint64_t)g_base * GB) & ~15ull;
    for (uint32_t g0 = g_lo; g0 < g_hi; g0 += nthr) {                      // wave-uniform trip count (ballots inside)
        if (g0 + (tid & ~63u) >= g_hi) break;                                // a wave without a live lane has nothing to convert (the ballots are per wave)
        const uint32_t g = g0 + tid; const bool live = g < g_hi;
        const uint32_t src = stage + (uint32_t)((uint64_t)(live ? g : g_lo) * GB - b0);
        const uint32_t u0 = g * GS;
        const bool whole = !IL && u0 >= S0 && u0 + GS <= S0 + TS;         // whole group lands in the tile: wide stores
        const uint32_t dst = a.sym_off + (u0 - S0);
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            uint32_t sq[QS];
            if constexpr (FE == FE_PIXELS) {
                uint32_t h[9], c[9]; bool bad = false;
#pragma unroll
                for (uint32_t i = 0; i < 9; ++i) {
                    h[i] = *(const uint16_t*)(lds + src + 18u * q + 2u * i);
                    c[i] = (i % 3 == 0) ? h[i] : ((h[i] + 40u) & 0xFFFFu);
                    bad |= c[i] >= ((i % 3 == 0) ? 243u : 81u);
                }
                if (__builtin_amdgcn_ballot_w64(bad) != 0) {                // out-of-range quantised values: exact general reduction
#pragma unroll
                    for (uint32_t i = 0; i < 9; ++i) c[i] = (i % 3 == 0) ? red_y(h[i]) : red_c(h[i]);
                }
                px3_to_sym13(c, sq);
            } else {
                uint32_t c[27]; bool bad = false;
                // the half group's 27 bytes as 14 halfwords (groups are 2-byte aligned; q = 1 starts on an odd byte)
                uint32_t hb[28];
#pragma unroll
                for (uint32_t i = 0; i < 14; ++i) { const uint32_t h = *(const uint16_t*)(lds + src + 26u * q + 2u * i); hb[2 * i] = h & 0xFFu; hb[2 * i + 1] = h >> 8; }
#pragma unroll
                for (uint32_t i = 0; i < 27; ++i) { c[i] = hb[q + i]; bad |= c[i] >= 27u; }
                if (__builtin_amdgcn_ballot_w64(bad) != 0) {
#pragma unroll
                    for (uint32_t i = 0; i < 27; ++i) c[i] = mod27(c[i]);
                }
                w3_to_sym26(c, sq);
            }
            if (live && whole) {
#pragma unroll
                for (uint32_t i = 0; i < QS; ++i) {
                    const uint32_t n = q * QS + i;
                    acc |= sq[i] << ((uint32_t)SH + 8u * (n % EM));            // stored pre-scaled (table entry offset)
                    if (n % EM == EM - 1u) {
                        if constexpr (EM == 2u) *(uint16_t*)(lds + dst + (n - 1u)) = (uint16_t)acc; else *(uint32_t*)(lds + dst + (n - 3u)) = acc;
                        acc = 0;
                    }
                }
            } else if (!IL && live) {
                // group straddles a tile edge (tile edges are multiples of 4, group starts are even): same wide stores, predicated
#pragma unroll
                for (uint32_t i = 0; i < QS; ++i) {
                    const uint32_t n = q * QS + i;
                    acc |= sq[i] << ((uint32_t)SH + 8u * (n % EM));
                    if (n % EM == EM - 1u) {
                        const uint32_t u = u0 + n - (EM - 1u);
                        if (u >= S0 && u + EM <= S0 + TS) {
                            if constexpr (EM == 2u) *(uint16_t*)(lds + dst + (n - 1u)) = (uint16_t)acc; else *(uint32_t*)(lds + dst + (n - 3u)) = acc;
                        }
                        acc = 0;
                    }
                }
            } else if (live) {
                IlCursor cur;
                if constexpr (IL) { if (u0 + q * QS < a.n_sym) cur.init(u0 + q * QS, a); }
#pragma unroll
                for (uint32_t i = 0; i < QS; ++i) {
                    uint32_t u = u0 + q * QS + i;
                    if constexpr (IL) { if (u >= a.n_sym) continue; u = cur.get(); cur.next(a); }
                    if (u >= S0 && u < S0 + TS) lds[a.sym_off + (u - S0)] = (uint8_t)(sq[i] << SH);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Phase 1 for pixels, packed: one lane converts TWO pixel triples at once with 16-bit packed VALU ops (v_pk_*): component
// i of triple A sits in the low half of a register, of triple B = A+2 in the high half (ds_read_u16 + ds_read_u16_d16_hi),
// every product stays below 2^16.  Both triples of a lane have the same parity, and a wave handles one parity only, so the
// byte pairing of the 13 output symbols (offset 13t is odd for odd t) is wave-uniform: 6 b16 stores + 1 b8 store per triple.
// ---------------------------------------------------------------------------------------------------------
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk_d3(u16x2 x)  { return (x * (uint16_t)171) >> (uint16_t)9; }    // x < 512 (products < 2^16 for x <= 383)
__device__ __forceinline__ u16x2 pk_d9(u16x2 x)  { return (x * (uint16_t)228) >> (uint16_t)11; }   // x <= 287
__device__ __forceinline__ u16x2 pk_d27(u16x2 x) { return (x * (uint16_t)152) >> (uint16_t)12; }   // x <= 431
__device__ __forceinline__ uint32_t pk_bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

// 9 reduced components (Y < 243, C < 81) of a triple pair -> 13 symbol pairs, each already multiplied by SC
template <int SC>
__device__ __forceinline__ void px3x2_to_sym13x8(const u16x2* c, u16x2* s) {
    const u16x2 Y0 = c[0], B0 = c[1], R0 = c[2], Y1 = c[3], B1 = c[4], R1 = c[5], Y2 = c[6], B2 = c[7], R2 = c[8];
    const uint16_t k8 = SC, k24 = 3 * SC, k72 = 9 * SC, k216 = 27 * SC;
    u16x2 q, t;
    q = pk_d27(Y0); s[0] = Y0 * k8 - q * k216;            t = pk_d3(B0);  s[1] = q * k8 + (B0 - t * (uint16_t)3) * k72;  s[2] = t * k8;
    q = pk_d27(R0); s[3] = R0 * k8 - q * k216;            t = pk_d9(Y1);  s[4] = q * k8 + (Y1 - t * (uint16_t)9) * k24;  s[5] = t * k8;
    q = pk_d27(B1); s[6] = B1 * k8 - q * k216;            t = pk_d9(R1);  s[7] = q * k8 + (R1 - t * (uint16_t)9) * k24;
    const u16x2 y3 = pk_d3(Y2), y9 = pk_d9(Y2), y81 = pk_d9(y9);                                  // Y2/3, Y2/9, Y2/81
    s[8] = t * k8 + (Y2 - y3 * (uint16_t)3) * k72;
    s[9] = y3 * k8 - y81 * k216;                                                                   // (Y2/3) % 27 = Y2/3 - 27 (Y2/81)
    t = pk_d9(B2);  s[10] = y81 * k8 + (B2 - t * (uint16_t)9) * k24;
    q = pk_d3(R2);  s[11] = t * k8 + (R2 - q * (uint16_t)3) * k72;  s[12] = q * k8;
}

// Convert the pixel triples that cover stream symbols [S0, S0+TS) from the stage buffer (image byte x = input byte b0 + x).
// A lane takes FOUR consecutive triples (12 pixels, 72 input bytes -> 52 symbols): its input is nine aligned 8-byte reads, its
// output thirteen aligned dwords.  Triples 0 and 2 share registers as low/high halves, so do 1 and 3 (same parity each).
// Triples are written whole: the symbol buffer has kSymFront bytes of slack in front and 64 behind, which take the symbols of
// the first/last triples that belong to the neighbouring tiles (and of the up to three triples past the tile's last one).
// Symbols [u_lo, u_hi) are produced (1-D: the tile itself; pipelined 2-D: the row segments it overlaps); symbol u lands at
// LDS byte sym_off + (u - u_lo).
// FE_RGB: the io_image.hpp bridge for one pixel (rgb_to_ycbcr :47-57, quantize_ycbcr :69-78): every float product and sum rounded on
// its own (no contraction), std::lround of a non-negative value = trunc(x + 0.5) exactly (x + 0.5 is exact or rounds inside
// the integer's unit interval), Y quantised in float (242 Y / 255 is never within 1/510 of a tie, the float error is 1e-5),
// chroma through the 256-byte table of its quantiser.  Returns the reduced components Y < 243, Cb + 40, Cr + 40 <= 80.
__device__ __forceinline__ void rgb_px_to_comps(const uint32_t r8, const uint32_t g8, const uint32_t b8, const uint32_t qt, uint32_t& Y, uint32_t& B, uint32_t& R) {
    const float r = (float)r8, g = (float)g8, b = (float)b8;
    const float y = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
    const float cb = __fadd_rn(__fadd_rn(__fsub_rn(__fmul_rn(-0.168736f, r), __fmul_rn(0.331264f, g)), __fmul_rn(0.5f, b)), 128.0f);
    const float cr = __fadd_rn(__fsub_rn(__fsub_rn(__fmul_rn(0.5f, r), __fmul_rn(0.418688f, g)), __fmul_rn(0.081312f, b)), 128.0f);
    const float Yi = fminf(truncf(__fadd_rn(y, 0.5f)), 255.0f);
    Y = (uint32_t)__fmaf_rn(Yi, 242.0f / 255.0f, 0.5f);
    const uint32_t Cb = min((uint32_t)__fadd_rn(cb, 0.5f), 255u), Cr = min((uint32_t)__fadd_rn(cr, 0.5f), 255u);
    B = lds_u8(qt + Cb); R = lds_u8(qt + Cr);
}

// One run of consecutive pixel triples for phase 1: triples [t_base, t_end) in lane units of four (t_base a multiple of 4); triple t
// reads its input at LDS address src0 + t * (18 | 9) (pixels | RGB)
struct P1Run { uint32_t t_base, t_end, n_units, src0, lo, hi, dst0; };   // lo, hi, dst0: run-placed 2-D flow (symbol u of [lo, hi) goes to LDS address dst0 + u)
template <int FE>
__device__ __forceinline__ P1Run p1_run(uint32_t u_lo, uint32_t u_hi, uint32_t stage) {    // symbols [u_lo, u_hi), their input staged at `stage` (see stage_input)
    constexpr uint32_t GBf = FE == FE_PIXELS ? kGroupBytes : kGroupBytesRgb, TB = FE == FE_PIXELS ? 18u : 9u;
    P1Run r; r.t_base = (u_lo / 13u) & ~3u; r.t_end = (u_hi + 12u) / 13u; r.n_units = (r.t_end - r.t_base + 3u) / 4u;
    const uint64_t b0 = ((uint64_t)(r.t_base / 2u) * GBf) & ~15ull;                       // 16-aligned start of the first lane group (two triples each)
    r.src0 = stage - (uint32_t)b0;                                                       // (wraps; src0 + t * TB does not)
    r.lo = u_lo; r.hi = u_hi; r.dst0 = 0;
    (void)TB;
    return r;
}

// IL: the symbols go to their post-interleave places in the tile [S0, S0 + TS) (what falls outside belongs to another tile);
// else symbol u goes to sym_off + (u - S0).
// placed (IL only, wave-uniform): the run-placed flow -- a run's symbols go, in pre-interleave order, to the place the run occupies in the
// tile (P1Run::dst0; rows, chunks and tile edges are multiples of 4 there, so aligned dwords stay aligned dwords) and the caller
// reverses the odd rows' pieces in place afterwards.
template <int SC, int FE, bool IL>
__device__ __forceinline__ void convert_pixels_packed(const EncArgs& a, const P1Run r0, const P1Run r1, const P1Run r2, uint32_t S0, uint32_t TS,
                                                      uint32_t lane, uint32_t wave, uint32_t nwv, const bool placed = false) {
    constexpr uint32_t TB = FE == FE_PIXELS ? 18u : 9u;
    const uint32_t nw1 = min(a.p1_wpp, nwv);                                      // waves that convert (planner: just enough lanes)
    if (wave >= nw1) return;
    const uint32_t n0 = r0.n_units, n01 = IL ? n0 + r1.n_units : n0, n_all = IL ? n01 + r2.n_units : n0;      // (separate values, not an array: selects, no private memory)
    for (uint32_t e0 = wave * 64u; e0 < n_all; e0 += nw1 * 64u) {
        const uint32_t e = e0 + lane;
        const uint32_t ri = !IL ? 0u : e < n0 ? 0u : e < n01 ? 1u : 2u;            // (1-D: one run)
        const uint32_t t_base = ri == 0u ? r0.t_base : ri == 1u ? r1.t_base : r2.t_base;
        const uint32_t t_end = ri == 0u ? r0.t_end : ri == 1u ? r1.t_end : r2.t_end;
        const uint32_t src0 = ri == 0u ? r0.src0 : ri == 1u ? r1.src0 : r2.src0;
        const uint32_t t = t_base + 4u * (e - (ri == 0u ? 0u : ri == 1u ? n0 : n01));
        const bool live = t < t_end;
        u16x2 sA[13], sB[13];
        if constexpr (FE == FE_RGB) {
            // 12 pixels = 36 bytes = nine aligned dwords; pixel p = bytes 3p .. 3p + 2
            const uint32_t src = src0 + (live ? t : t_base) * TB;
            uint32_t D[9];
#pragma unroll
            for (uint32_t i = 0; i < 9; ++i) D[i] = lds_u32(src + 4u * i);
            // pixels past the padded end of the frame are zero TRITS (Cb + 40 = 0), which no RGB value encodes: only the frame's last lanes
            const uint64_t px0 = 3ull * t;
            const bool tail = __builtin_amdgcn_ballot_w64(live && px0 + 12u > a.n_units_pad) != 0;
            auto byte = [&](uint32_t k) -> uint32_t { return (D[k >> 2] >> (8u * (k & 3u))) & 0xFFu; };
#pragma unroll
            for (uint32_t pair = 0; pair < 2; ++pair) {                           // pair 0 = triples (0, 2), pair 1 = triples (1, 3): six pixels at a time (register budget)
                u16x2 c[9];
#pragma unroll
                for (uint32_t m = 0; m < 3; ++m) {
                    const uint32_t pa = 3u * pair + m, pb = pa + 6u;              // pixel of the pair's first / second triple
                    uint32_t Ya, Ba, Ra, Yb, Bb, Rb;
                    rgb_px_to_comps(byte(3u * pa), byte(3u * pa + 1u), byte(3u * pa + 2u), a.qt_off, Ya, Ba, Ra);
                    rgb_px_to_comps(byte(3u * pb), byte(3u * pb + 1u), byte(3u * pb + 2u), a.qt_off, Yb, Bb, Rb);
                    if (tail) {
                        if (px0 + pa >= a.n_units_pad) { Ya = 0; Ba = 0; Ra = 0; }
                        if (px0 + pb >= a.n_units_pad) { Yb = 0; Bb = 0; Rb = 0; }
                    }
                    c[3 * m] = u16x2{(uint16_t)Ya, (uint16_t)Yb}; c[3 * m + 1] = u16x2{(uint16_t)Ba, (uint16_t)Bb}; c[3 * m + 2] = u16x2{(uint16_t)Ra, (uint16_t)Rb};
                }
                px3x2_to_sym13x8<SC>(c, pair ? sB : sA);
            }
        } else {
        const uint32_t src = src0 + (live ? t : t_base) * TB;                             // 8-byte aligned
        uint32_t D[18];
#pragma unroll
        for (uint32_t i = 0; i < 9; ++i) { const u32x2 v = *T3_LDS_PTR(u32x2, src + 8u * i); D[2 * i] = v.x; D[2 * i + 1] = v.y; }
        // halves whose triple lies past the tile's last one hold stale bytes: keep them out of the range check
        const uint32_t liveA = t + 2u < t_end ? 0xFFFFFFFFu : 0x0000FFFFu, liveB = (t + 1u < t_end ? 0x0000FFFFu : 0u) | (t + 3u < t_end ? 0xFFFF0000u : 0u);
#pragma unroll
        for (uint32_t pair = 0; pair < 2; ++pair) {                               // pair 0 = triples (0, 2), pair 1 = triples (1, 3)
            u16x2 h[9], c[9];
            u16x2 mxY = {0, 0}, mxC = {0, 0};                                    // range check on the maxima: one comparison per kind instead of one per component
#pragma unroll
            for (uint32_t i = 0; i < 9; ++i) {
                const uint32_t u = 9u * pair + i;                                 // 16-bit index of the low-half component; the high half sits 18 further
                const uint32_t w = __builtin_amdgcn_perm(D[9u + u / 2u], D[u / 2u], (u & 1u) ? 0x07060302u : 0x05040100u);
                h[i] = __builtin_bit_cast(u16x2, w);
                c[i] = (i % 3 == 0) ? h[i] : h[i] + (uint16_t)40;
                if (i % 3 == 0) mxY = __builtin_elementwise_max(mxY, c[i]); else mxC = __builtin_elementwise_max(mxC, c[i]);
            }
            const u16x2 over = __builtin_elementwise_sub_sat(mxY, (u16x2)((uint16_t)242)) | __builtin_elementwise_sub_sat(mxC, (u16x2)((uint16_t)80));
            if (__builtin_amdgcn_ballot_w64(live && (pk_bits(over) & (pair ? liveB : liveA)) != 0u) != 0) {   // out-of-range quantised values: exact general reduction
#pragma unroll
                for (uint32_t i = 0; i < 9; ++i) {
                    const uint32_t lo = h[i].x, hi = h[i].y;
                    c[i] = (i % 3 == 0) ? u16x2{(uint16_t)red_y(lo), (uint16_t)red_y(hi)} : u16x2{(uint16_t)red_c(lo), (uint16_t)red_c(hi)};
                }
            }
            px3x2_to_sym13x8<SC>(c, pair ? sB : sA);
        }
        }
        // 16-bit pieces of the 52 output bytes (low half: first triple of the pair, high half: second):
        //   E_j = (s_2j, s_2j+1) of triples 0/2;  O_j = (s_2j+1, s_2j+2) of triples 1/3;  X = (s_12 of 0/2, s_0 of 1/3)
        uint32_t E[6], O[6];
#pragma unroll
        for (uint32_t j = 0; j < 6; ++j) {
            E[j] = pk_bits(sA[2 * j]) | (pk_bits(sA[2 * j + 1]) << 8);
            O[j] = pk_bits(sB[2 * j + 1]) | (pk_bits(sB[2 * j + 2]) << 8);
        }
        const uint32_t X = pk_bits(sA[12]) | (pk_bits(sB[0]) << 8);
        constexpr uint32_t LL = 0x05040100u, HH = 0x07060302u, LH = 0x07060100u;   // v_perm(S0, S1): result = (S1.lo|S0.lo), (S1.hi|S0.hi), (S1.lo|S0.hi)
        uint32_t o[13];
        o[0] = __builtin_amdgcn_perm(E[1], E[0], LL);  o[1] = __builtin_amdgcn_perm(E[3], E[2], LL);  o[2] = __builtin_amdgcn_perm(E[5], E[4], LL);
        o[3] = __builtin_amdgcn_perm(O[0], X, LL);     o[4] = __builtin_amdgcn_perm(O[2], O[1], LL);  o[5] = __builtin_amdgcn_perm(O[4], O[3], LL);
        o[6] = __builtin_amdgcn_perm(E[0], O[5], LH);
        o[7] = __builtin_amdgcn_perm(E[2], E[1], HH);  o[8] = __builtin_amdgcn_perm(E[4], E[3], HH);  o[9] = __builtin_amdgcn_perm(X, E[5], HH);
        o[10] = __builtin_amdgcn_perm(O[1], O[0], HH); o[11] = __builtin_amdgcn_perm(O[3], O[2], HH); o[12] = __builtin_amdgcn_perm(O[5], O[4], HH);
        if constexpr (!IL) {
            if (live) {
                const uint32_t dst = a.sym_off + 13u * t - S0;                    // dword aligned; may sit below sym_off (front slack)
#pragma unroll
                for (uint32_t j = 0; j < 6; ++j) *T3_LDS_WPTR(u32x2a4, dst + 8u * j) = u32x2a4{o[2 * j], o[2 * j + 1]};
                *T3_LDS_WPTR(uint32_t, dst + 48u) = o[12];
            }
        } else if (placed) {
            const uint32_t lo = ri == 0u ? r0.lo : ri == 1u ? r1.lo : r2.lo, hi = ri == 0u ? r0.hi : ri == 1u ? r1.hi : r2.hi;
            const uint32_t u0 = 13u * t, dst = (ri == 0u ? r0.dst0 : ri == 1u ? r1.dst0 : r2.dst0) + u0;     // dword aligned
            const bool inside = live && u0 >= lo && u0 + 52u <= hi;
            if (inside) {
#pragma unroll
                for (uint32_t j = 0; j < 6; ++j) *T3_LDS_WPTR(u32x2a4, dst + 8u * j) = u32x2a4{o[2 * j], o[2 * j + 1]};
                *T3_LDS_WPTR(uint32_t, dst + 48u) = o[12];
            }
            if (__builtin_amdgcn_ballot_w64(live && !inside) != 0) {              // the few lane units a run's ends cut through: dword by dword
                if (live && !inside) {
#pragma unroll
                    for (uint32_t j = 0; j < 13; ++j) { const uint32_t u = u0 + 4u * j; if (u >= lo && u < hi) *T3_LDS_WPTR(uint32_t, dst + 4u * j) = o[j]; }
                }
            }
        } else if (live) {
            // 2-D boustrophedon folded into the stores (OLD:750-780): with rows, chunks and the tile start multiples of 4 an aligned
            // dword of the stream stays an aligned dword -- in place in even rows, byte-reversed at the mirrored column in odd
            // rows; any other geometry (and the stream's last, shorter row) goes symbol by symbol
            const uint32_t u0 = 13u * t, lim = 13u * t_end;                       // u0: a multiple of 4; symbols from lim on were never converted
            const bool rows4 = (a.il_w & 3u) == 0u && ((a.il_A & 3u) == 0u || a.il_A >= a.n_sym) && (S0 & 3u) == 0u && (TS & 3u) == 0u;
            IlCursor cur;
            if (u0 < a.n_sym) cur.init(u0, a);
            if (rows4) {
#pragma unroll
                for (uint32_t j = 0; j < 13; ++j) {
                    const uint32_t u = u0 + 4u * j;
                    uint32_t v = u, val = o[j];
                    bool whole = u + 4u <= lim;
                    if (u < a.n_sym) {
                        whole = whole && u + 4u <= a.n_sym && cur.rowlen == a.il_w;
                        if (cur.odd) { v = cur.base + cur.rw + (a.il_w - 4u - cur.c); val = __builtin_bswap32(val); }
                        cur.next(a, 4u);
                    }
                    if (whole) { if (v - S0 < TS) *T3_LDS_WPTR(uint32_t, a.sym_off + (v - S0)) = val; }
                    else if (u < lim) {
#pragma unroll
                        for (uint32_t i = 0; i < 4; ++i) {
                            const uint32_t uu = u + i, vv = uu < a.n_sym ? il_perm(uu, a) : uu;
                            if (uu < lim && vv - S0 < TS) *T3_LDS_WPTR(uint8_t, a.sym_off + (vv - S0)) = (uint8_t)(o[j] >> (8u * i));
                        }
                    }
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < 13; ++j) {                              // (unrolled: o[] must stay in registers)
#pragma unroll
                    for (uint32_t i = 0; i < 4; ++i) {
                        const uint32_t u = u0 + 4u * j + i;
                        uint32_t v = u;
                        if (u < a.n_sym) { v = cur.get(); cur.next(a); }
                        if (u < lim && v - S0 < TS) *T3_LDS_WPTR(uint8_t, a.sym_off + (v - S0)) = (uint8_t)(o[j] >> (8u * i));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Phase 1 for raw words, packed (1-D; the reference's own entry encode_profile_from_raw OLD:1043, regroup OLD:1051-1082): 3 words = 27
// symbol bytes of which trit 26 of every word is dropped = 78 trits = 26 symbols.  One lane = FOUR word triples = 108 input bytes =
// 27 aligned dwords -> 104 symbols = 26 aligned dwords (round 2 gave a lane one group of 6 words through 28 16-bit LDS reads and a
// division per symbol: 0.111 ms per 8K frame against 0.086 for pixels).  With c[0..26] the triple's bytes:
//   s[m] = c[m]                                  m < 8            (word 0 as it is)
//   s[m] = H(c[m]) + W Lo(c[m+1])                m = 8 .. 25      (word 1 shifted by one trit, word 2 by two)
//     m = 8: c%9 + 9 (c'%3);  9..16: c/3 + 9 (c'%3);  17: (c/3)%3 + 3 (c'%9);  18..25: c/9 + 3 (c'%9)
// computed on pairs (c[8+2i], c[9+2i]) in the 16-bit halves of a register (v_pk_*): H and Lo are element-wise, the pairing of
// Lo(c[m+1]) with H(c[m]) is one v_alignbit per output pair.  Triples are written whole (slack either side of the symbol buffer:
// kSymSlackW); the stage buffer holds real input for all four triples of every live lane (stage_tile rounds up), so stale bytes
// never enter the range check.  Symbols are stored pre-scaled by 2^SH like the pixel converter's.
// ---------------------------------------------------------------------------------------------------------
struct W1Run { uint32_t t_base, t_end, n_units, src0; };      // word triples [t_base, t_end), lane units of four; triple t reads its 27 bytes at src0 + 27 t
__device__ __forceinline__ W1Run w1_run(uint32_t u_lo, uint32_t u_hi, uint32_t stage) {
    W1Run r; r.t_base = (u_lo / 26u) & ~3u; r.t_end = (u_hi + 25u) / 26u; r.n_units = (r.t_end - r.t_base + 3u) / 4u;
    const uint64_t b0 = ((uint64_t)(r.t_base / 2u) * kGroupBytesW) & ~15ull;              // 16-aligned start of the first lane group (two triples each): stage_input
    r.src0 = stage - (uint32_t)b0;                                                         // (wraps; src0 + 27 t does not)
    return r;
}
// A lane unit's two halves (two triples each) go to two different waves -- even converting waves take half 0, odd ones half 1, so the
// byte selectors stay compile-time constants -- which halves the length of phase 1, a serial section of the tile (a wave alone issues a
// vector instruction every four to five cycles).
template <int SH, uint32_t HALF>
__device__ __forceinline__ void convert_words_half(const EncArgs& a, const W1Run r, uint32_t S0, uint32_t lane, uint32_t uw, uint32_t nuw) {
    for (uint32_t e0 = uw * 64u; e0 < r.n_units; e0 += nuw * 64u) {
        const uint32_t e = e0 + lane;
        const uint32_t t = r.t_base + 4u * e;
        const bool live = t < r.t_end;
        const uint32_t src = r.src0 + 27u * (live ? t : r.t_base);                          // dword aligned
        const uint32_t dst = a.sym_off + 26u * t - S0;                                       // dword aligned; may sit below sym_off (front slack)
        {                                                                                    // two triples = 54 bytes in, 52 symbols = 13 dwords out
            constexpr uint32_t half = HALF;
            constexpr uint32_t kDw = 14;
            const uint32_t w0b = 54u * half, d0 = w0b >> 2, B0 = w0b & 3u;                    // the half's bytes start B0 bytes into its first dword
            uint32_t D[kDw];
#pragma unroll
            for (uint32_t i = 0; i < kDw / 2; ++i) { const u32x2a4 v = *T3_LDS_PTR(u32x2a4, src + 4u * d0 + 8u * i); D[2 * i] = v.x; D[2 * i + 1] = v.y; }
            // any byte >= 27?  (b + 101) sets bit 7 exactly for b in 27..154, a byte >= 155 has it set already (a carry only adds set bits).
            // The 56 bytes read hold the half's 54 and two of its neighbours: real input as well (stage_tile stages whole lanes).
            uint32_t hi = 0;
#pragma unroll
            for (uint32_t i = 0; i < kDw; ++i) hi |= D[i] | (D[i] + 0x65656565u);
            if (__builtin_amdgcn_ballot_w64(live && (hi & 0x80808080u) != 0u) != 0) {       // non-canonical symbols: unpack3 reduces every digit (OLD:28-31)
#pragma unroll
                for (uint32_t i = 0; i < kDw; ++i) {
                    uint32_t o = 0;
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) { const uint32_t c = (D[i] >> (8u * q)) & 0xFFu; o |= mod27(c) << (8u * q); }   // c < 256 < 512
                    D[i] = o;
                }
            }
            uint32_t R[2][7];
#pragma unroll
            for (uint32_t jj = 0; jj < 2; ++jj) {
                const uint32_t B = B0 + 27u * jj;                                            // first byte of the triple in D
                // byte k of D as the low / high half of a pair (v_perm(S0, S1): selector 0..3 = bytes of S1, 4..7 = of S0, 0x0c = zero)
                auto pair = [&](uint32_t k1, uint32_t k2) -> u16x2 {
                    const uint32_t sel = (k1 & 3u) | 0x0c00u | ((4u + (k2 & 3u)) << 16) | 0x0c000000u;
                    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(D[k2 >> 2], D[k1 >> 2], sel));
                };
                u16x2 P[10], Lo[10], H[9];
#pragma unroll
                for (uint32_t m = 0; m < 10; ++m) P[m] = m < 9u ? pair(B + 8u + 2u * m, B + 9u + 2u * m) : __builtin_bit_cast(u16x2, (D[(B + 26u) >> 2] >> (8u * ((B + 26u) & 3u))) & 0xFFu);
                u16x2 q3[5], q9a, q9b;
#pragma unroll
                for (uint32_t m = 0; m < 5; ++m) { q3[m] = pk_d3(P[m]); Lo[m] = P[m] - q3[m] * (uint16_t)3; }     // c % 3 (the low half of Lo[0] is not used)
                q9a = pk_d9(P[0]); q9b = pk_d9(P[4]);
                {   // H[0] = (c8 % 9, c9 / 3)
                    const u16x2 r9 = P[0] - q9a * (uint16_t)9;
                    H[0] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(pk_bits(q3[0]), pk_bits(r9), 0x07060100u));
                }
#pragma unroll
                for (uint32_t m = 1; m < 4; ++m) H[m] = q3[m];
                H[4] = q3[4] - q9b * u16x2{0, 3};                                            // (c16 / 3, (c17 / 3) % 3)
#pragma unroll
                for (uint32_t m = 5; m < 10; ++m) { const u16x2 q9 = pk_d9(P[m]); if (m < 9u) H[m] = q9; Lo[m] = P[m] - q9 * (uint16_t)9; }   // c / 9, c % 9
                uint32_t O[9];
#pragma unroll
                for (uint32_t m = 0; m < 9; ++m) {
                    const u16x2 nx = __builtin_bit_cast(u16x2, __builtin_amdgcn_alignbit(pk_bits(Lo[m + 1]), pk_bits(Lo[m]), 16u));   // (Lo of c[9+2m], Lo of c[10+2m])
                    const u16x2 Wv = m < 4u ? u16x2{9, 9} : m == 4u ? u16x2{9, 3} : u16x2{3, 3};
                    O[m] = pk_bits(H[m] + nx * Wv);
                }
                // bytes: s0..s7 = c0..c7, then the nine pairs
                auto dw = [&](uint32_t k) -> uint32_t { return (k & 3u) == 0u ? D[k >> 2] : __builtin_amdgcn_alignbyte(D[(k >> 2) + 1u], D[k >> 2], k & 3u); };
                R[jj][0] = dw(B); R[jj][1] = dw(B + 4u);
#pragma unroll
                for (uint32_t i = 0; i < 4; ++i) R[jj][2 + i] = __builtin_amdgcn_perm(O[2 * i + 1], O[2 * i], 0x06040200u);
                R[jj][6] = __builtin_amdgcn_perm(0u, O[8], 0x0c0c0200u);
#pragma unroll
                for (uint32_t i = 0; i < 7; ++i) R[jj][i] <<= (uint32_t)SH;                   // table-entry offsets (symbols <= 26: no carry between bytes)
            }
            // 52 bytes: the first triple's 26, then the second's at byte 26 (two bytes into dword 6)
            uint32_t o[13];
#pragma unroll
            for (uint32_t i = 0; i < 6; ++i) o[i] = R[0][i];
            o[6] = R[0][6] | (R[1][0] << 16);
#pragma unroll
            for (uint32_t i = 0; i < 6; ++i) o[7 + i] = __builtin_amdgcn_alignbit(R[1][i + 1], R[1][i], 16u);
            if (live) {
                const uint32_t d = dst + 52u * half;
#pragma unroll
                for (uint32_t i = 0; i < 6; ++i) *T3_LDS_WPTR(u32x2a4, d + 8u * i) = u32x2a4{o[2 * i], o[2 * i + 1]};
                *T3_LDS_WPTR(uint32_t, d + 48u) = o[12];
            }
        }
    }
}
template <int SH>
__device__ __forceinline__ void convert_words_packed(const EncArgs& a, const W1Run r, uint32_t S0, uint32_t lane, uint32_t wave, uint32_t nwv) {
    const uint32_t nuw = min(a.p1_wpp, nwv / 2u);                                            // waves per half (planner: just enough lane units)
    if (wave >= 2u * nuw) return;
    if (wave & 1u) convert_words_half<SH, 1>(a, r, S0, lane, wave >> 1, nuw); else convert_words_half<SH, 0>(a, r, S0, lane, wave >> 1, nuw);
}

// workgroup barrier that waits for this wave's LDS traffic only: unlike __syncthreads() it leaves the LDS-DMA prefetch
// of the next tile (and the previous tile's global stores) in flight
__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and the one at the top of a tile, which also drains vmcnt: the prefetched input has landed for every wave
__device__ __forceinline__ void barrier_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Top-of-tile barrier: vmcnt completes in order and the LDS-DMA prefetch of this tile was issued BEFORE the previous tile's
// `younger` global stores, so waiting until at most `younger` operations are outstanding is exactly "the input has landed"
// without also waiting for those stores to be acknowledged.
__device__ __forceinline__ void barrier_input(uint32_t younger) {
#ifdef T3_NO_COUNTED_VMCNT
    younger = 0;
#endif
    switch (younger) {
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
        default: barrier_all(); break;
    }
}

// RSEL = 26-k when every band of the launch shares one k (the common case: no dead code paths, fewer registers,
// 640-thread bound so that two workgroups share a CU); RSEL = 0 handles mixed k with a wave-uniform switch.
// IL: 0 = 1-D; 1 = 2-D, whole rows + permutation pass (raw words: the row-by-row flow); 2 = 2-D, runs + permuting stores (wide rows)
template <int FE, int IL, int RSEL, bool BCN>
__device__ __forceinline__ void encode_body(const EncArgs& a) {
    constexpr uint32_t GS = fe_px(FE) ? kGroupSyms : kGroupSymsW;      // symbols per lane group
    constexpr uint32_t GBf = FE == FE_PIXELS ? kGroupBytes : FE == FE_RGB ? kGroupBytesRgb : kGroupBytesW;
    constexpr int SH = RSEL != 0 ? 2 : 3;                                     // symbol pre-scale: 4-byte T entries (MFMA) / 8-byte LUT entries
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
    const uint32_t TS = 9u * a.Lq;

    // per-band geometry and wave roles -> LDS header (kernel arguments must not be indexed dynamically: that would
    // force a private copy of the whole argument block); LUT images -> LDS once per (persistent) workgroup
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) {
            BandRow r; r.k = a.band_k[b]; r.nbt = a.band_nb_tile[b]; r.blocks = a.band_blocks[b]; r.lut_off = a.band_lut_off[b];
            r.pad_ = 0; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b];
            *(BandRow*)(lds + 32 * b) = r;
        }
#pragma unroll
        for (int b = 0; b < 10; ++b) *(uint32_t*)(lds + 288 + 4 * b) = a.band_first[b];
#pragma unroll
        for (int i = 0; i < 12; ++i) *(uint32_t*)(lds + 336 + 4 * i) = a.scr[i];
        if constexpr (RSEL == 1) {                                            // UEP group records and the set table
#pragma unroll
            for (int gi = 0; gi < kMaxGrp; ++gi) {
                uint32_t* gp = (uint32_t*)(lds + kHdrGrp + kHdrGrpStride * gi);
                gp[0] = a.grp[gi].nb; gp[1] = a.grp[gi].div_nb.mul; gp[2] = a.grp[gi].div_nb.sh; gp[3] = a.grp[gi].div_nb.d;
                gp[4] = a.grp[gi].n_items; gp[5] = a.grp[gi].r; gp[9] = a.grp[gi].afrag_off;
#pragma unroll
                for (int q = 0; q < 3; ++q) gp[6 + q] = (uint32_t)a.grp[gi].bands[4 * q] | (uint32_t)a.grp[gi].bands[4 * q + 1] << 8 | (uint32_t)a.grp[gi].bands[4 * q + 2] << 16 | (uint32_t)a.grp[gi].bands[4 * q + 3] << 24;
#pragma unroll
                for (int q = 0; q < 12; ++q) gp[12 + q] = a.grp[gi].scr[q];
            }
#pragma unroll
            for (int q = 0; q < kMaxSets; ++q) *(uint32_t*)(lds + kHdrSets + 4 * q) = a.set_tab[q];
        }
    }
    for (uint32_t i = tid * 16u; i < a.lut_bytes; i += nthr * 16u)
        *(uint4*)(lds + (RSEL == 1 ? kLdsHdrUep : kLdsHdr) + i) = *(const uint4*)((const uint8_t*)a.lut_img + i);

    if constexpr (FE == FE_RGB) { if (tid < 64u) *(uint32_t*)(lds + a.qt_off + 4u * tid) = ((const uint32_t*)a.qt)[tid]; }

    if (blockIdx.x == 0 && a.frame_out) {                                    // header symbols + zero tail (OLD:1159-1167)
        if (tid == 0) {                                                      // constant indices only (see above)
#pragma unroll
            for (uint32_t i = 0; i < 96; ++i) if (i < a.hdr_syms) a.frame_out[i] = a.hdr[i];
        }
        if (tid < a.pad_bytes) a.frame_out[a.out_syms + tid] = 0;
        if constexpr (BCN) { if (tid < a.bcn_tail_len) a.frame_out[a.bcn_tail_off + tid] = (uint8_t)(a.bcn_tail_vals >> (8u * tid)); }   // after the last body byte
    }

    v4i Afr[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};                   // single-k kernels: the parity matrix lives in 12 VGPRs
    if constexpr (RSEL > 1) {
#pragma unroll
        for (int s = 0; s < 3; ++s) Afr[s] = ((const v4i*)a.afrag)[s * 64 + lane];
        // Let these loads land here, with a wait the compiler's counter tracking sees (the builtin, not inline asm): otherwise it
        // protects their first use inside the tile loop with s_waitcnt vmcnt(0) in front of the first MFMA of every tile, which also
        // waits for the next tile's input prefetch and the previous tile's stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);                                     // vmcnt(0) only (gfx9 encoding: vm[3:0] | exp << 4 | lgkm << 8 | vm[5:4] << 14)
    }

#ifdef T3_STAMPS   // diagnostic build: per-phase cycle sums of wave 0 (never in the product build)
    uint64_t st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime(), st_t0 = st_prev, st_rt0 = __builtin_amdgcn_s_memrealtime();
#define T3_STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define T3_STAMP(i) do { } while (0)
#endif

    // first lane group whose input a tile starting at stream symbol S needs (pixels: the packed converter starts at a
    // multiple of 4 triples = 2 groups)
    auto first_group = [](uint32_t S) -> uint32_t { return fe_px(FE) ? ((S / 13u) & ~3u) / 2u : ((S / 26u) & ~3u) / 2u; };   // (raw words: four word triples = two groups per lane)
    // one past the last lane group: raw words stage whole lanes (four triples), so that no lane of the packed converter meets stale bytes
    auto end_group = [](uint32_t S) -> uint32_t { return fe_px(FE) ? (S + GS - 1u) / GS : ((((S + 25u) / 26u) + 3u) & ~3u) / 2u; };
    // pipelined flow (input prefetch, packed converter): always in 1-D; in 2-D for pixel / RGB input (il_async != 0):
    //   il_async == 1 (rows up to 512 symbols): the tile's input covers the whole row segments it overlaps; phase 1 leaves the symbols in
    //     PRE-interleave order and a permutation pass by all waves moves them, in post-interleave order, into the stage buffer the
    //     tile's input has just been consumed from (measured: cheaper than permuting in the three converting waves' stores);
    //   il_async == 2 (wider rows): the tile's pre-interleave symbols are up to three runs (il_runs) staged one behind the other and
    //     phase 1 stores every symbol at its post-interleave place -- no row is staged whole, any width.
    // (Raw words in 2-D keep the row-by-row flow below.)
    constexpr bool fast = !IL || fe_px(FE);                                    // (the host sets a.il_async == IL for pixel / RGB input, 0 for raw words)
    // the runs of a tile and where each one's input sits in a stage buffer: run i at kRunPitch-rounded offsets (an LDS-DMA piece is
    // a whole KiB, so a run's last piece may reach up to 1008 bytes past its end)
    struct TileIn { uint32_t lo[3], hi[3], off[3], plo[3], n; };
    auto tile_in = [&](uint32_t S) -> TileIn {
        TileIn T; T.n = 1; T.lo[0] = S; T.hi[0] = S + TS; T.off[0] = 0; T.lo[1] = T.lo[2] = T.hi[1] = T.hi[2] = 0; T.off[1] = T.off[2] = 0; T.plo[0] = S; T.plo[1] = T.plo[2] = 0;
        if constexpr (IL == 1 && fe_px(FE)) {                                  // narrow rows: the whole row segments the tile overlaps, one run
            if (S < a.n_sym) { T.lo[0] = il_row_start(S, a); T.hi[0] = max(il_row_end(min(S + TS, a.n_sym) - 1u, a), S + TS); }
            return T;
        }
        if constexpr (IL == 2 && fe_px(FE)) {
            const IlRuns R = il_runs(S, TS, a);
            T.n = R.n; uint32_t off = 0;
#pragma unroll
            for (uint32_t i = 0; i < 3; ++i) {
                T.lo[i] = R.lo[i]; T.hi[i] = R.hi[i]; T.off[i] = off; T.plo[i] = R.plo[i];
                const uint32_t bytes = (uint32_t)((uint64_t)end_group(R.hi[i]) * GBf - (((uint64_t)first_group(R.lo[i]) * GBf) & ~15ull));
                if (i < R.n) off += (bytes + 1023u + 16u) & ~1023u;
            }
        }
        return T;
    };
    auto stage_tile = [&](const TileIn& T, uint32_t stage, uint32_t w, uint32_t nw) {
#pragma unroll
        for (uint32_t i = 0; i < 3; ++i) if (i < T.n) stage_input<FE>(a, stage + T.off[i], first_group(T.lo[i]), end_group(T.hi[i]), lane, w, nw);
    };
    if constexpr (fast) {                                                    // prologue: first tile's input
        if (blockIdx.x < a.n_tiles) stage_tile(tile_in(blockIdx.x * TS), a.stage_off, wave, nwv);
    }
    uint32_t younger = 0;                                                    // VMEM ops this wave issued after its last prefetch
    // Tiles are handed out dynamically: the three workgroups of a CU progress at different speeds (oldest wave first),
    // up to 1.6x apart.  The first tile is blockIdx.x, every further one a ticket.  One counter serves ~11 ns per draw
    // (memory-side atomic), too slow for 14k tiles, so workgroups and tiles are split into n_classes classes by index
    // modulo n_classes, each with its own counter (class == XCD under round-robin dispatch, but nothing relies on it).
    const bool dyn = fast && a.tile_ctr != nullptr;
    const uint32_t NC = a.n_classes, cls = blockIdx.x % NC;
    uint32_t* const ctr = a.tile_ctr + 64u * cls;                              // one counter per class, 256 B apart
    const uint32_t wgc = (gridDim.x - cls + NC - 1u) / NC;                     // workgroups in this class
    // a class's tiles are cls + NC j: j < wgc first tiles (= blockIdx), wgc <= j < 2 wgc second tiles (static too), then tickets
    // the input of tile i+1 is requested at the top of tile i into the other stage buffer, by the waves that phase 1 (pixels)
    // leaves idle: issuing the LDS-DMA costs ~400 cycles per KiB piece and would otherwise sit between the two phases
    const uint32_t w0 = fe_px(FE) ? min(a.p1_wpp, nwv - 1u) : min(2u * min(a.p1_wpp, nwv / 2u), nwv - 1u);   // waves that convert (planner: just enough lanes of four triples; raw words: two waves per lane unit)
    // Tickets are drawn by lane 0 of the LAST wave: the compiler turns the atomic into its wave-aggregated form, which reads the
    // result back at once (s_waitcnt vmcnt(0): the atomic's round trip plus the acknowledgement of the wave's stores of the
    // previous tile).  On thread 0 that stall sat in front of phase 1's conversion, on the critical path of every tile; the
    // last wave has no conversion work (pixels), and is taken off prefetch duty so that the wait does not cover a DMA either.
    const bool excl = dyn && w0 + 1u < nwv;                                    // the drawing wave issues no prefetch
    const uint32_t n_pf = nwv - w0 - (excl ? 1u : 0u);
    // The roles rotate from tile to tile: a "virtual" wave index vw = wave - rot (mod nwv) decides who converts (vw < w0),
    // who prefetches and who draws (vw = nwv - 1), and rot advances by w0 per tile.  Waves sit on SIMD (wave mod 4) for the
    // whole kernel; with fixed roles the conversion -- more than half of the kernel's VALU work -- always ran on the same
    // three SIMDs of a CU and those bounded the tile rate.  A ticket is drawn at the top of a tile and names the tile two after
    // it (its input is requested at the top of the next tile): the drawing wave reads the atomic back at once anyway, so
    // holding the ticket for one more tile only made the workgroups commit a tile earlier than needed (longer tail).
    uint32_t par = 0, rot = 0;
    for (uint32_t tile = blockIdx.x, nxt = dyn ? cls + NC * (wgc + blockIdx.x / NC) : blockIdx.x + gridDim.x, nn = 0; tile < a.n_tiles;
         tile = nxt, nxt = nn, par ^= 1u, rot = (rot + w0 >= nwv ? rot + w0 - nwv : rot + w0)) {
        const uint32_t vw = fast ? (wave >= rot ? wave - rot : wave + nwv - rot) : wave;
        const bool drawer = dyn && lane == 0u && vw == nwv - 1u;
        const uint32_t S0 = tile * TS;
        const uint32_t stage = a.stage_off + (fast ? par * a.stage_stride : 0u);
        uint32_t symb = a.sym_off;                                            // where phase 2 finds the tile's symbols
        // ---------------- phase 1: input -> stream-ordered symbols in LDS ----------------
        if constexpr (fast) {
            barrier_input(younger);                                           // this tile's input has landed, everyone left phase 2
            T3_STAMP(0);
            if (drawer) *(uint32_t*)(lds + 328) = cls + NC * (2u * wgc + atomicAdd(ctr, 1u));   // the tile after the next one (read after the barrier below)
#ifndef T3_ABL_NO_PREFETCH
            if (nxt < a.n_tiles && vw >= w0 && vw - w0 < n_pf) stage_tile(tile_in(nxt * TS), a.stage_off + (par ^ 1u) * a.stage_stride, vw - w0, n_pf);
#endif
            T3_STAMP(4);
            uint32_t u_lo = S0, u_hi = S0 + TS;
            bool placed = false;                                              // IL == 2: this tile goes through the run-placed flow
#ifndef T3_ABL_NO_P1
            if constexpr (fe_px(FE)) {
                const TileIn T = tile_in(S0);
                u_hi = T.hi[0];
                P1Run r0 = p1_run<FE>(T.lo[0], T.hi[0], stage + T.off[0]), r1 = p1_run<FE>(T.lo[1], T.hi[1], stage + T.off[1]), r2 = p1_run<FE>(T.lo[2], T.hi[2], stage + T.off[2]);
                if (T.n < 1u) r0.n_units = 0; if (T.n < 2u) r1.n_units = 0; if (T.n < 3u) r2.n_units = 0;
                u_lo = T.lo[0];
                if constexpr (IL == 2) {
                    // run-placed flow (round 3): whole rows of multiples of 4 symbols, the tile inside the stream; the stream's last tiles
                    // (ragged last row, padding) and other geometries keep the cursor flow
                    placed = (a.il_w & 3u) == 0u && ((a.il_A & 3u) == 0u || a.il_A >= a.n_sym) && (S0 & 3u) == 0u && (TS & 3u) == 0u && S0 + TS <= a.n_sym
                             && il_row_end(S0 + TS - 1u, a) - il_row_start(S0 + TS - 1u, a) == a.il_w;
                    r0.dst0 = a.sym_off + (T.plo[0] - S0) - T.lo[0]; r1.dst0 = a.sym_off + (T.plo[1] - S0) - T.lo[1]; r2.dst0 = a.sym_off + (T.plo[2] - S0) - T.lo[2];   // (wraps; dst0 + u does not)
                }
                if constexpr (IL == 1) convert_pixels_packed<(1 << SH), FE, false>(a, r0, r1, r2, u_lo, TS, lane, vw, nwv);   // pre-interleave order; the pass below moves them
                else convert_pixels_packed<(1 << SH), FE, IL == 2>(a, r0, r1, r2, S0, TS, lane, vw, nwv, placed);
            } else convert_words_packed<SH>(a, w1_run(S0, S0 + TS, stage), S0, lane, vw, nwv);
#endif
            T3_STAMP(5);                                                      // (diagnostic) this wave's conversion
            barrier_lds();                                                    // symbols complete
            T3_STAMP(1);
            nn = dyn ? __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + 328)) : nxt + gridDim.x;   // the tile after the next one
            if (IL == 2 && placed) {
                // the odd rows' pieces inside the tile, reversed in place (dword pairs, bytes swapped): the tile's first row from S0, whole
                // rows, the last row up to the tile's end -- one lane = the dwords i and n - 1 - i of a piece
                const uint32_t E = S0 + TS;
                uint32_t rl0, rn0, od0; il_row(S0, a, rl0, rn0, od0);
                const uint32_t he = min(rl0 + rn0, E);                            // end of the first row's piece
                const uint32_t w8 = (a.il_w + 7u) >> 3;                           // lane tasks of a whole row
                const uint32_t n0 = od0 ? (he - S0 + 7u) >> 3 : 0u;
                const uint32_t rows = (E - he + a.il_w - 1u) / a.il_w;            // further rows the tile touches (the last one maybe in part)
                for (uint32_t t = tid; t < n0 + rows * w8; t += nthr) {
                    uint32_t pa, len, i;                                          // piece start (post position), length, dword index
                    if (t < n0) { pa = S0; len = he - S0; i = t; }
                    else {
                        const uint32_t ri = (t - n0) / w8; i = (t - n0) - ri * w8;
                        pa = he + ri * a.il_w; len = min(a.il_w, E - pa);
                        const uint32_t chunk = fdiv2(pa, a.div_A), r = fdiv2(pa - chunk * a.il_A, a.div_w);
                        if (!(r & 1u)) continue;
                    }
                    const uint32_t nd = len >> 2, j = nd - 1u - i;
                    if (i > j || i >= nd) continue;
                    const uint32_t ad = a.sym_off + (pa - S0);
                    const uint32_t x = lds_u32(ad + 4u * i), y = lds_u32(ad + 4u * j);
                    *T3_LDS_WPTR(uint32_t, ad + 4u * i) = __builtin_bswap32(y);
                    if (i != j) *T3_LDS_WPTR(uint32_t, ad + 4u * j) = __builtin_bswap32(x);
                }
                barrier_lds();
            }
            if (IL == 1 && (a.il_w & 15u) == 0u) {
                // Rows of whole 16-byte granules (round 3): the interleave maps every row of the chunk grid onto itself -- even rows stay,
                // odd rows are mirrored -- and the symbol buffer holds whole rows (tile_in), so the odd rows are reversed IN PLACE: a lane swaps
                // the 16-byte granules g and G - 1 - g of a row, bytes reversed; phase 2 then reads the tile at its offset inside the
                // first row.  About a hundred lane tasks per tile; the pass below (every symbol moved into the consumed stage buffer, two
                // divisions per granule) took 3.5 k of a tile's 12 k cycles (stamp build, profiles/r03/notes.md).
                const uint32_t G = a.il_w >> 4, G2 = (G + 1u) >> 1, n_rows = (u_hi - u_lo + a.il_w - 1u) / a.il_w;
                for (uint32_t t = tid; t < n_rows * G2; t += nthr) {
                    const uint32_t ri = t / G2, g = t - ri * G2, p0 = u_lo + ri * a.il_w;
                    if (p0 >= a.n_sym) continue;                                    // padding past the stream's end: identity
                    const uint32_t chunk = fdiv2(p0, a.div_A), base = chunk * a.il_A, r = fdiv2(p0 - base, a.div_w);
                    if (!(r & 1u)) continue;
                    const uint32_t take = min(a.il_A, a.n_sym - base), rowlen = min(a.il_w, take - r * a.il_w);
                    const uint32_t ra = a.sym_off + (p0 - u_lo);                    // 16-byte aligned: rows start at multiples of 16 from u_lo
                    if (rowlen == a.il_w) {
                        const uint32_t g1 = G - 1u - g;
                        const u32x4 x = *T3_LDS_PTR(u32x4, ra + 16u * g), y = *T3_LDS_PTR(u32x4, ra + 16u * g1);
                        *T3_LDS_WPTR(u32x4, ra + 16u * g) = u32x4{__builtin_bswap32(y.w), __builtin_bswap32(y.z), __builtin_bswap32(y.y), __builtin_bswap32(y.x)};
                        if (g1 != g) *T3_LDS_WPTR(u32x4, ra + 16u * g1) = u32x4{__builtin_bswap32(x.w), __builtin_bswap32(x.z), __builtin_bswap32(x.y), __builtin_bswap32(x.x)};
                    } else if (g == 0u) {                                           // the stream's last, short row: one lane, byte by byte
                        for (uint32_t i = 0; 2u * i + 1u < rowlen; ++i) {
                            const uint32_t lo = lds_u8(ra + i), hi = lds_u8(ra + rowlen - 1u - i);
                            *T3_LDS_WPTR(uint8_t, ra + i) = (uint8_t)hi; *T3_LDS_WPTR(uint8_t, ra + rowlen - 1u - i) = (uint8_t)lo;
                        }
                    }
                }
                barrier_lds();
                T3_STAMP(3);
                symb = a.sym_off + (S0 - u_lo);
            } else if constexpr (IL == 1) {
                // permutation pass: post-interleave position v of the tile <- pre-interleave symbol il_perm(v) (an involution);
                // one lane = 4 consecutive positions = one dword of the image phase 2 reads
                // Rows of the chunk grid map onto themselves, and with rows that are multiples of 4 symbols (tile edges and chunk sizes
                // are too) an aligned dword of a row stays an aligned dword: copied in even rows, byte-reversed from the mirrored
                // column in odd rows.  Anything else (other widths, the stream's last short row, the padding past the stream's
                // end) walks the cursor symbol by symbol.
                const bool rows4 = (a.il_w & 3u) == 0u && ((a.il_A & 3u) == 0u || a.il_A >= a.n_sym) && (S0 & 3u) == 0u;
                auto one_dword = [&](uint32_t v) {                                 // image dword at tile offset v - S0 (a multiple of 4)
                    const uint32_t dst = stage + (v - S0);
                    uint32_t w4 = 0;
                    bool done = false;
                    if (rows4 && v + 4u <= a.n_sym) {
                        const uint32_t chunk = fdiv2(v, a.div_A), base = chunk * a.il_A, rem = v - base, take = min(a.il_A, a.n_sym - base);
                        const uint32_t r = fdiv2(rem, a.div_w), rw = r * a.il_w, c = rem - rw, rowlen = min(a.il_w, take - rw);
                        if (!(r & 1u)) { w4 = lds_u32(a.sym_off + (v - u_lo)); done = true; }
                        else if (rowlen == a.il_w) { w4 = __builtin_bswap32(lds_u32(a.sym_off + (base + rw + (a.il_w - 4u - c) - u_lo))); done = true; }
                    }
                    if (!done) {
                        IlCursor cur;
                        if (v < a.n_sym) cur.init(v, a);
#pragma unroll
                        for (uint32_t q = 0; q < 4; ++q, ++v) {
                            uint32_t u = v;
                            if (v < a.n_sym) { u = cur.get(); cur.next(a); }
                            w4 |= lds_u8(a.sym_off + (u - u_lo)) << (8u * q);
                        }
                    }
                    *T3_LDS_WPTR(uint32_t, dst) = w4;
                };
                if (rows4 && (a.il_w & 15u) == 0u && ((a.il_A & 15u) == 0u || a.il_A >= a.n_sym)) {
                    // rows are multiples of 16 symbols: one lane = one 16-byte granule of the stream (aligned in stream coordinates, so it
                    // lies inside one row): one row computation per 16 symbols; the tile's ragged ends go dword by dword
                    const uint32_t g0 = S0 & ~15u;
                    for (uint32_t q = tid; g0 + 16u * q < S0 + TS; q += nthr) {
                        const uint32_t v = g0 + 16u * q;
                        if (v >= S0 && v + 16u <= S0 + TS && v + 16u <= a.n_sym) {
                            const uint32_t chunk = fdiv2(v, a.div_A), base = chunk * a.il_A, rem = v - base, take = min(a.il_A, a.n_sym - base);
                            const uint32_t r = fdiv2(rem, a.div_w), rw = r * a.il_w, c = rem - rw, rowlen = min(a.il_w, take - rw);
                            if (!(r & 1u) || rowlen == a.il_w) {
                                const uint32_t src = (r & 1u) ? base + rw + (a.il_w - 16u - c) : v;
                                const u32x4 x = *T3_LDS_PTR(u32x4, a.sym_off + (src - u_lo));           // 16-byte aligned: rows start at multiples of 16 from u_lo
                                const uint32_t dst = stage + (v - S0);                                  // only 4-byte aligned (tile edges are multiples of 4)
                                if (r & 1u) {
                                    *T3_LDS_WPTR(u32x2a4, dst) = u32x2a4{__builtin_bswap32(x.w), __builtin_bswap32(x.z)};
                                    *T3_LDS_WPTR(u32x2a4, dst + 8u) = u32x2a4{__builtin_bswap32(x.y), __builtin_bswap32(x.x)};
                                } else {
                                    *T3_LDS_WPTR(u32x2a4, dst) = u32x2a4{x.x, x.y};
                                    *T3_LDS_WPTR(u32x2a4, dst + 8u) = u32x2a4{x.z, x.w};
                                }
                                continue;
                            }
                        }
                        for (uint32_t d = 0; d < 4u; ++d) { const uint32_t vd = v + 4u * d; if (vd >= S0 && vd < S0 + TS) one_dword(vd); }
                    }
                } else {
                    for (uint32_t g = tid; 4u * g < TS; g += nthr) one_dword(S0 + 4u * g);
                }
                barrier_lds();
                T3_STAMP(3);
                symb = stage;
            }
        } else {
            nn = nxt + gridDim.x;
            __syncthreads();                                                  // everyone left phase 2 of the previous tile
            for (uint32_t i = tid * 16u; i < TS; i += nthr * 16u) *(uint4*)(lds + a.sym_off + i) = make_uint4(0, 0, 0, 0);
            uint32_t u_lo = S0, u_hi = S0;                                   // pre-interleave symbols this tile needs: whole row segments
            const uint32_t hi = min(S0 + TS, a.n_sym);
            if (S0 < hi) { u_lo = il_row_start(S0, a); u_hi = il_row_end(hi - 1u, a); }
            const uint32_t g_lo = u_lo / GS, g_hi = (u_hi + GS - 1u) / GS;
            __syncthreads();
            for (uint32_t gc = g_lo; gc < g_hi; gc += a.stage_groups) {
                const uint32_t gc_hi = min(g_hi, gc + a.stage_groups);
                stage_input<FE>(a, stage, gc, gc_hi, lane, wave, nwv);
                __syncthreads();
                T3_STAMP(0);
                convert_groups<FE, true, SH>(a, stage, gc, gc, gc_hi, S0, TS, tid, nthr);
                __syncthreads();
                T3_STAMP(1);
            }
        }

        // ---------------- phase 2: one lane = one RS block; a wave stays inside one band; stores go straight to HBM ----------------
        younger = 0;
#ifndef T3_ABL_NO_P2
        {
            const uint32_t item = tid;                                        // one lane = one block, dealt linearly across the bands
            bool did = false;
            if constexpr (RSEL > 1) {                                          // one k on all nine bands: both sets of the wave in one call
                P2Map M; M.item0[0] = wave * 64u; M.item0[1] = wave * 64u + 32u; M.n_items = a.n_items; M.nb = a.nb_uniform; M.div_nb = a.div_nb;
                M.band_tab = ~0u; M.scr_off = kMfmaScr;
                // REGEO (recompute the block geometry per tile instead of keeping it across the tile loop): chosen per instantiation from the
                // register allocator's result (profiles/kernel_resources.py: spilled VGPRs with / without) -- RS(26,20) from pixels / raw words
                // fits the 80-VGPR budget as it is, most others stop spilling with it, a few spill less without it
                constexpr bool regeo = FE == FE_RGB ? (RSEL != 6 || IL == 2) : RSEL == 6 ? false
                    : (FE == FE_PIXELS && IL == 0 && RSEL == 8 && !BCN) ? false
                    : (FE == FE_PIXELS && IL == 2 && (RSEL == 2 || (RSEL == 4 && BCN))) ? false
                    : (FE == FE_WORDS && IL == 1 && (RSEL == 2 || RSEL == 8)) ? false : true;
                younger = phase2_mfma<RSEL, false, BCN, regeo>(a, symb, tile, lane, Afr, M);
            } else if constexpr (RSEL == 1) {                                  // UEP: a set lies inside one group of bands that share k
                younger = 0;
#pragma unroll
                for (uint32_t q = 0; q < 2; ++q) {
                    const uint32_t set = wave + 8u * q;                         // sets go round the eight waves
                    if (set >= a.n_sets) continue;
                    const uint32_t st = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kHdrSets + 4u * set));
                    const uint32_t gb = kHdrGrp + kHdrGrpStride * (st & 0xFFu);
                    P2Map M; M.item0[0] = st >> 8; M.item0[1] = 0xFFFF0000u;
                    M.nb = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb)); M.div_nb.mul = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 4));
                    M.div_nb.sh = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 8)); M.div_nb.d = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 12));
                    M.n_items = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 16));
                    const uint32_t rr = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 20)), ao = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + gb + 36));
                    M.band_tab = gb + 24u; M.scr_off = gb + 48u;
                    v4i Ag[3];
#pragma unroll
                    for (int s = 0; s < 3; ++s) Ag[s] = *T3_LDS_PTR(v4i, ao + 16u * (s * 64u + lane));
                    switch (rr) {
                        case 2: younger += phase2_mfma<2, true, BCN>(a, symb, tile, lane, Ag, M); break;
                        case 4: younger += phase2_mfma<4, true, BCN>(a, symb, tile, lane, Ag, M); break;
                        case 6: younger += phase2_mfma<6, true, BCN>(a, symb, tile, lane, Ag, M); break;
                        default: younger += phase2_mfma<8, true, BCN>(a, symb, tile, lane, Ag, M); break;
                    }
                }
            } else if (item < a.n_items) {
                uint32_t b = 0;
#pragma unroll
                for (uint32_t q = 1; q < 9; ++q) if (item >= band_first(q)) b = q;
                const uint32_t m = item - band_first(b), nbt = band_row(b).nbt;
                switch (band_row(b).k) {                                         // lanes of one wave may sit in two bands (mixed k: divergent)
                    case 24: did = phase2_band<2, false>(a, symb, tile, b, m, nbt); break;
                    case 22: did = phase2_band<4, false>(a, symb, tile, b, m, nbt); break;
                    case 20: did = phase2_band<6, false>(a, symb, tile, b, m, nbt); break;
                    default: did = phase2_band<8, false>(a, symb, tile, b, m, nbt); break;
                }
            }
            // single-k kernels count the store instructions issued after the prefetch (see phase2_mfma); the mixed kernel
            // does not (younger = 0 over-waits, which is safe)
            (void)did;
        }
#endif
        T3_STAMP(2);
    }
    if (dyn && lane == 0u && wave == nwv - 1u) {                                   // (not `tid`: it would stay live, or spilled, across the whole tile loop)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (every draw was read back by its wave right away)
        if (atomicAdd(a.tile_ctr + 64u * NC, 1u) == gridDim.x - 1u) {          // ... and so has everyone else's: re-arm for the next launch
            for (uint32_t c = 0; c <= NC; ++c) __hip_atomic_store(a.tile_ctr + 64u * c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef T3_STAMPS
    if (tid == 0 && a.dbg) {
        uint64_t* d = a.dbg + 16ull * blockIdx.x;
        d[8] = __builtin_amdgcn_s_getreg(31 << 11 | 4); d[9] = __builtin_amdgcn_s_getreg(31 << 11 | 20);   // HW_ID, XCC_ID
        d[0] = st_acc[0]; d[1] = st_acc[1]; d[2] = st_acc[2] + st_acc[3]; d[3] = st_rt0; d[10] = st_acc[3];
        d[4] = __builtin_amdgcn_s_memtime() - st_t0; d[5] = __builtin_amdgcn_s_memrealtime() - st_rt0; d[6] = st_acc[4]; d[7] = st_acc[5];
    }
#endif
}

#ifndef T3_ENC_WAVES_PER_EU
#define T3_ENC_WAVES_PER_EU 6   // <= 80 VGPRs: three 8-wave workgroups per CU
#endif
template <int FE, int IL, int RSEL, bool BCN>
__global__ __launch_bounds__(512, T3_ENC_WAVES_PER_EU) void encode_kernel_k(const EncArgs a) { encode_body<FE, IL, RSEL, BCN>(a); }
template <int FE, int IL, bool BCN>
__global__ __launch_bounds__(512, T3_ENC_WAVES_PER_EU) void encode_kernel_uep(const EncArgs a) { encode_body<FE, IL, 1, BCN>(a); }   // UEP on the matrix cores
template <int FE, int IL>
__global__ __launch_bounds__(1024) void encode_kernel_mixed(const EncArgs a) { encode_body<FE, IL, 0, false>(a); }

#define T3_INST_KB(FE, IL, BCN) \
    template __global__ void encode_kernel_k<FE, IL, 2, BCN>(const EncArgs); template __global__ void encode_kernel_k<FE, IL, 4, BCN>(const EncArgs); \
    template __global__ void encode_kernel_k<FE, IL, 6, BCN>(const EncArgs); template __global__ void encode_kernel_k<FE, IL, 8, BCN>(const EncArgs); \
    template __global__ void encode_kernel_uep<FE, IL, BCN>(const EncArgs);
#define T3_INST_K(FE, IL) T3_INST_KB(FE, IL, false) T3_INST_KB(FE, IL, true) template __global__ void encode_kernel_mixed<FE, IL>(const EncArgs);
T3_INST_K(FE_PIXELS, 0) T3_INST_K(FE_PIXELS, 1) T3_INST_K(FE_PIXELS, 2) T3_INST_K(FE_WORDS, 0) T3_INST_K(FE_WORDS, 1) T3_INST_K(FE_RGB, 0) T3_INST_K(FE_RGB, 1) T3_INST_K(FE_RGB, 2)

// ---------------------------------------------------------------------------------------------------------
// beacon insertion pass (OLD:1118-1141): framed[q] = beacon symbol at slot `slot` of every period-th word, else
// body[q - #beacons before q], zero past the body.  One lane = one 16-byte granule of the output frame: between two
// beacons that is a shifted copy (aligned dword loads + funnel shifts, one 16-byte store); a granule that holds a beacon
// or touches the header / the end of the body goes byte by byte.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void beacon_kernel(const BeaconArgs a) {
    const uint64_t g0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g0 == 0) {
        for (uint32_t i = 0; i < a.hdr_syms; ++i) a.frame_out[i] = a.hdr[i];
        for (uint32_t i = 0; i < a.pad_bytes; ++i) a.frame_out[a.hdr_syms + a.framed_syms + i] = 0;
    }
    const uint64_t cyc = 9ull * a.period;
    const uint64_t end = a.hdr_syms + a.framed_syms;
    const bool small = end < (1ull << 32) && cyc < (1ull << 32);                  // 32-bit divisions where the frame allows
    auto before = [&](uint64_t q) -> uint64_t {                                    // beacons strictly before q
        if (a.slot >= 9u || q <= a.slot) return 0u;
        return small ? (uint64_t)(((uint32_t)q - a.slot - 1u) / (uint32_t)cyc + 1u) : (q - a.slot - 1u) / cyc + 1u;
    };
    for (uint64_t g = g0; 16 * g < end; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t B0 = 16 * g;
        if (B0 >= a.hdr_syms && B0 + 16 <= end && ((uintptr_t)a.frame_out & 15u) == 0) {
            const uint64_t q = B0 - a.hdr_syms, nb = before(q);
            if (nb == before(q + 16) && q - nb + 20 <= a.body_syms) {
                const uint8_t* src = a.body + (q - nb);
                const uint32_t sh = ((uint32_t)(uintptr_t)src & 3u) * 8u;
                const uint32_t* p = (const uint32_t*)((uintptr_t)src & ~(uintptr_t)3);
                uint32_t dw[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) dw[i] = p[i];
                *(uint4*)(a.frame_out + B0) = make_uint4(__builtin_amdgcn_alignbit(dw[1], dw[0], sh), __builtin_amdgcn_alignbit(dw[2], dw[1], sh),
                                                          __builtin_amdgcn_alignbit(dw[3], dw[2], sh), __builtin_amdgcn_alignbit(dw[4], dw[3], sh));
                continue;
            }
        }
        // byte by byte, no division per byte: the beacons are at slot + j cyc, `nb` of them lie before q
        const uint64_t Bs = max(B0, (uint64_t)a.hdr_syms), Be = min(B0 + 16, end);
        uint64_t nb = before(Bs - a.hdr_syms), qb = a.slot < 9u ? a.slot + nb * cyc : ~0ull;
        for (uint64_t B = Bs; B < Be; ++B) {
            const uint64_t q = B - a.hdr_syms;
            uint8_t v;
            if (q == qb) { v = (uint8_t)a.sym; ++nb; qb += cyc; }
            else { const uint64_t kq = q - nb; v = kq < a.body_syms ? a.body[kq] : 0; }
            a.frame_out[B] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// standalone 2-D boustrophedon (interleave2D_boustrophedon / deinterleave2D_boustrophedon OLD:750-813): out[u] = in[perm(u)].
// The map is an involution inside every row segment (odd rows of a chunk reversed, the stream's ragged last rows within
// their own length), so one kernel serves both directions.  API completeness: the frame kernels carry the map fused.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interleave_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n, uint32_t w, uint32_t A, DevDiv div_A, DevDiv div_w) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const uint32_t chunk = fdiv(u, div_A), base = chunk * A, rem = u - base, take = min(A, n - base);
    const uint32_t r = fdiv(rem, div_w), c = rem - r * w, rowlen = min(w, take - r * w);
    out[u] = in[base + r * w + ((r & 1u) ? rowlen - 1u - c : c)];
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
