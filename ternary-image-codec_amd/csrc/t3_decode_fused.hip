// t3_decode_fused.hip — fused FIXED-mode ("v6c") decoder for gfx950: K3 syndromes + K4 Berlekamp–Massey / Chien / Forney +
// K5 symbols -> pixels|words in ONE launch, tiled like the encoder (BASELINE config 5: decode with injected trit errors).
// Replaces, for mode=FIXED / uniform k / 1-D / no beacon: descramble (OLD:938-947), per-band RS decode (OLD:963-991,
// decode_block OLD:546-662 with the Forney sign fixed), the i%9 re-merge the reference omits, symbols -> 26-trit words
// (OLD:1022-1040) and unpack_two_pixels (OLD:706-722).  Everything else goes through the generic kernels in t3_decode.hip.
//
// One lane = one RS block, blocks dealt linearly over the eight waves.  Per tile:
//   D1  7 aligned dword loads per lane (blocks are 26 B, 2-byte aligned) -> 26 symbols in registers
//   D2  descramble through an 81-byte LDS table (result pre-scaled by 8), syndromes through a per-position LUT:
//       6-bit SWAR trit fields, two conflict-free ds_read_b64 per symbol, one mod-3 fold per block
//   D3  lanes with non-zero syndromes: Berlekamp-Massey (x*B kept shifted, fixed 8-coefficient registers) on a fused
//       multiply-add table (a + x y, 27^3 bytes in LDS), degree test, Chien search by table (root mask per locator,
//       global memory), Forney
//   D4  data symbols -> stream order in LDS (byte 9(mk+p)+b), <=t corrected bytes patched in place
//   D5  pixels: one lane = four triples, 13 aligned dwords of symbols -> 12 pixels with packed 16-bit ops -> 72 bytes stored
//       straight to memory;  raw words: 26 symbols -> 3 words, staged in LDS, copied out with 16-byte coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"

namespace t3 {

extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
// LDS access by absolute byte address in the hot loops: the kernel owns the whole LDS allocation (no static __shared__), so
// the dynamic array starts at 0; `lds[x]` would make the compiler add that link-time zero to every address (t3_kernels.hip).
#define T3_LP(T, a) ((__attribute__((address_space(3))) T*)(uintptr_t)(a))
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t l8(uint32_t a) { return *T3_LP(const uint8_t, a); }

namespace {
constexpr uint32_t MUL = kFxTab, ADD = kFxTab + 729, SUB = kFxTab + 1458, INV = kFxTab + 2187, NEG = INV + 27, EXP = NEG + 27, DSC = EXP + 26;
static_assert(DSC == kFxTab + offsetof(FxTables, descr), "LDS table map");

__device__ __forceinline__ uint32_t gfm(uint32_t a, uint32_t b) { return l8(MUL + a * 27u + b); }
__device__ __forceinline__ uint32_t gfa(uint32_t a, uint32_t b) { return l8(ADD + a * 27u + b); }
__device__ __forceinline__ uint32_t gfs(uint32_t a, uint32_t b) { return l8(SUB + a * 27u + b); }

__device__ __forceinline__ uint32_t mod3x5(uint32_t x) {          // five 6-bit fields (<= 63) -> {0,1,2}
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x0F3CF3CFu);
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x030C30C3u);
    x = (x & 0x030C30C3u) + ((x >> 2) & 0x01041041u);
    const uint32_t t = x & (x >> 1) & 0x01041041u;
    return x - (t | (t << 1));
}
__device__ __forceinline__ uint32_t d3(uint32_t x)  { return __umul24(x, 171u) >> 9; }
__device__ __forceinline__ uint32_t d9(uint32_t x)  { return __umul24(x, 228u) >> 11; }
__device__ __forceinline__ uint32_t d27(uint32_t x) { return __umul24(x, 152u) >> 12; }

struct Fix { uint32_t np; uint32_t pos[4]; uint32_t mag[4]; };   // up to t = 4 corrections (RS(26,18))

// decode_block after the syndromes (OLD:567-659), FIXED flavour, for R syndromes and T = R/2.  Polynomials live in fixed
// zero-padded registers, which is equivalent to the reference's growing vectors (only coefficient VALUES matter once the
// Horner loops start at the true degree).  Returns false for an uncorrectable block.
template <int R>
__device__ __forceinline__ bool fx_correct(const uint32_t* S, Fix& fx, const uint32_t* __restrict__ root_tbl, uint32_t FMA) {
    auto fma = [FMA](uint32_t acc, uint32_t x, uint32_t y) -> uint32_t { return l8(FMA + (x * 27u + y) * 27u + acc); };   // acc + x y
    constexpr int T = R / 2, NP = R + 2;
    // sigma, and x^m * B as the reference's xmdB.  B = (sigma before the last length change) / (its discrepancy): the division is
    // kept as the scalar `binv` and applied to the discrepancy instead (d binv) * x^m * sigma_old -- one product per
    // iteration instead of one per coefficient, the same field elements in the end
    uint32_t sg[NP], bx[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { sg[i] = 0; bx[i] = 0; }
    sg[0] = 1; bx[1] = 1;
    uint32_t L = 0, binv = 1;
#pragma unroll
    for (int n = 0; n < R; ++n) {
        uint32_t d = S[n];
#pragma unroll
        for (int i = 1; i <= n; ++i) d = fma(d, sg[i], S[n - i]);           // sigma[i] = 0 beyond L: same sum as OLD:572
        const bool upd = d != 0 && 2u * L <= (uint32_t)n;
        const uint32_t nc = l8(NEG + gfm(d, binv));                          // -(d / d_old)
        uint32_t old[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            old[i] = sg[i];
            if (i <= n + 1) sg[i] = fma(old[i], nc, bx[i]);                  // sigma - d B x^m; d == 0: unchanged (OLD:573-587)
        }
        if (upd) { L = (uint32_t)n + 1u - L; binv = l8(INV + d); }           // B <- T / delta (OLD:590-592)
#pragma unroll
        for (int i = NP - 1; i >= 1; --i) bx[i] = upd ? old[i - 1] : bx[i - 1];  // next x^m * B (unscaled)
        bx[0] = 0;
    }
    uint32_t deg = 0;
#pragma unroll
    for (int i = 1; i < NP; ++i) if (sg[i] != 0) deg = (uint32_t)i;
    fx.np = 0;
#ifdef T3_ABL_DEC_BM_ONLY
    fx.np = 1; fx.pos[0] = deg; fx.mag[0] = sg[1]; return true;
#endif
    if (deg > (uint32_t)T) return false;                                      // then #roots > t or #roots != deg (OLD:624 + FIXED rule)
    // Chien (OLD:611-623), tabulated: sigma_0 = 1, so the T higher coefficients index the mask of the positions i with
    // sigma(alpha^-i) = 0 (built on the host from the same Horner evaluation, t3_api_decode.cpp)
    uint32_t ridx = sg[T];
#pragma unroll
    for (int q = T - 1; q >= 1; --q) ridx = ridx * 27u + sg[q];
    const uint32_t roots = root_tbl[ridx];
    const uint32_t np = (uint32_t)__popc(roots);
    if (np != deg) return false;
    // Omega = S(x) sigma(x) mod x^R (OLD:606-610), sigma' in characteristic 3 (OLD:625-641): sigma1 + 2 sigma2 x (+ 4th, 5th for R=8)
    uint32_t Om[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        uint32_t acc = S[q];                                                   // j = 0 term, sigma0 = 1
#pragma unroll
        for (int j = 1; j <= T; ++j) if (j <= q) acc = fma(acc, S[q - j], sg[j]);
        Om[q] = acc;
    }
    uint32_t r = roots;
#pragma unroll
    for (int e = 0; e < T; ++e) {
        if ((uint32_t)e < np) {
            const uint32_t p = (uint32_t)__ffs((int)r) - 1u; r &= r - 1u;
            const uint32_t xi = l8(EXP + (p == 0 ? 0u : 26u - p));
            uint32_t num = Om[R - 1];
#pragma unroll
            for (int q = R - 2; q >= 0; --q) num = fma(Om[q], num, xi);
            uint32_t den = fma(sg[1], gfa(sg[2], sg[2]), xi);                  // sigma1 + 2 sigma2 x  (x^2 term of sigma' is 3 sigma3 = 0)
            if constexpr (T >= 4) den = fma(den, gfm(gfm(sg[4], xi), xi), xi);  // + 4 sigma4 x^3 = sigma4 x^3
            if (den == 0) return false;                                        // OLD:656
            fx.pos[e] = p; fx.mag[e] = gfm(l8(NEG + num), l8(INV + den));    // OLD:657; FIXED subtracts it
        }
    }
    fx.np = np;
    return true;
}

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pd3(u16x2 x)  { return (x * (uint16_t)171) >> (uint16_t)9; }
__device__ __forceinline__ u16x2 pd9(u16x2 x)  { return (x * (uint16_t)228) >> (uint16_t)11; }
__device__ __forceinline__ uint32_t bits(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }

struct Row { uint32_t blocks, boff6; uint64_t body_off; };
__device__ __forceinline__ Row row(uint32_t b) { return *(const Row*)(lds + 16u * b); }
}  // namespace

template <int R, bool TO_PIXELS>
__global__ __launch_bounds__(512, 4) void decode_fixed_kernel(const DecFxArgs a) {
    constexpr uint32_t K = 26 - R, SLAB = R == 8 ? 768u : 512u;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    // constants -> LDS
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) { Row r; r.blocks = a.band_blocks[b]; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b]; *(Row*)(lds + 16 * b) = r; }
    }
    for (uint32_t i = tid * 16u; i < (uint32_t)sizeof(FxTables); i += nthr * 16u) *(uint4*)(lds + kFxTab + i) = *(const uint4*)((const uint8_t*)a.tab + i);
    for (uint32_t i = tid * 16u; i < a.lut_bytes; i += nthr * 16u) *(uint4*)(lds + kFxLut + i) = *(const uint4*)((const uint8_t*)a.lut + i);
    for (uint32_t i = tid * 16u; i < 19696u; i += nthr * 16u) *(uint4*)(lds + a.fma_off + i) = *(const uint4*)(a.fma + i);
    __syncthreads();

    const uint32_t units_tile = TO_PIXELS ? (a.TS / 13u) * 3u : (a.TS / 26u) * 3u;       // pixels / words produced per tile
    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        // ---------------- D1-D4: one lane = one block of band `wave` ----------------
        {
            const uint32_t item = tid, b = min(item / a.nb, 8u), m = item - b * a.nb;   // one lane = one block, dealt linearly across the bands
            const Row rw = row(b);
            const uint64_t mg = (uint64_t)tile * a.nb + m;
            if (item < 9u * a.nb && mg < rw.blocks) {
                const uint8_t* g = a.in + a.hdr_syms + rw.body_off + 26ull * mg;
                const uint32_t sh = ((uint32_t)(uintptr_t)g & 3u) * 8u;
                uint32_t w[7];
                if (g + 28 <= a.in + a.in_bytes) {
                    const uint32_t* p = (const uint32_t*)((uintptr_t)g & ~(uintptr_t)3);
                    uint32_t dw[7];
#pragma unroll
                    for (int i = 0; i < 7; ++i) dw[i] = p[i];
#pragma unroll
                    for (int i = 0; i < 6; ++i) w[i] = __builtin_amdgcn_alignbit(dw[i + 1], dw[i], sh);
                    w[6] = dw[6] >> sh;
                } else {                                                       // last bytes of the stream: no over-read
#pragma unroll
                    for (int i = 0; i < 7; ++i) w[i] = 0;
#pragma unroll
                    for (int i = 0; i < 26; ++i) w[i >> 2] |= (uint32_t)g[i] << (8 * (i & 3));
                }
                // scrambler state per residue class of the position (6-periodic); the first two body symbols are special
                const uint32_t c0 = (rw.boff6 + 2u * (uint32_t)(mg % 3u)) % 6u;
                uint32_t dbase[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) dbase[q] = DSC + 32u * ((a.cyc24 >> (2u * (c0 + q))) & 3u);
                const bool first = rw.body_off == 0 && mg == 0;
                uint32_t c[26];
#pragma unroll
                for (int i = 0; i < 26; ++i) c[i] = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
                // any byte >= 27?  (b + 101) sets bit 7 exactly for b in 27..154, and a byte >= 155 has bit 7 set already
                uint32_t hi = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i) hi |= (w[i] | (w[i] + 0x65656565u));   // a carry out of a byte only ever adds set bits
                if (__builtin_amdgcn_ballot_w64((hi & 0x80808080u) != 0u) != 0) {
#pragma unroll
                    for (int i = 0; i < 26; ++i) c[i] -= 27u * d27(c[i]);           // unpack3 semantics for non-canonical bytes
                }
                uint32_t d8[26];
#pragma unroll
                for (int i = 0; i < 26; ++i) {
                    uint32_t base = dbase[i % 6];
                    if (i < 2 && first) base = DSC + 32u * (i == 0 ? a.pre0 : a.pre1);
                    d8[i] = l8(base + c[i]);                                      // descrambled symbol * 8
                }
                uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0;
#pragma unroll
#ifdef T3_ABL_DEC_NO_SYND
                for (uint32_t i = 0; i < 0; ++i) {
#else
                for (uint32_t i = 0; i < 26; ++i) {
#endif
                    const u32x2 A = *T3_LP(const u32x2, kFxLut + i * SLAB + d8[i]);
                    const u32x2 B = *T3_LP(const u32x2, kFxLut + i * SLAB + 256u + d8[i]);
                    acc0 += A.x; acc1 += A.y; acc2 += B.x; acc3 += B.y;
                    if constexpr (R == 8) acc4 += *T3_LP(const uint32_t, kFxLut + i * SLAB + 512u + d8[i]);
                    if (i % 9 == 8) { asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4)); __builtin_amdgcn_sched_barrier(0); }
                }
                const uint32_t x0 = mod3x5(acc0), x1 = mod3x5(acc1), x2 = mod3x5(acc2);
                const uint32_t Sm = x0 + 3u * x1 + 9u * x2;
                uint32_t S[R];
                constexpr int NMAIN = R < 5 ? R : 5;
#pragma unroll
                for (int j = 0; j < NMAIN; ++j) S[j] = (Sm >> (6 * j)) & 63u;
                uint32_t any = Sm & 0x3FFFFFFFu;
                if constexpr (R == 2) any = Sm & 0xFFFu;
                if constexpr (R == 4) any = Sm & 0xFFFFFFu;
                if constexpr (R >= 6) {
                    const uint32_t x3 = mod3x5(acc3);
                    S[5] = (x3 & 63u) + 3u * ((x3 >> 6) & 63u) + 9u * ((x3 >> 12) & 63u);
                    any |= S[5];
                    if constexpr (R == 8) {
                        const uint32_t x4 = mod3x5(acc4);
                        S[6] = ((x3 >> 18) & 63u) + 3u * ((x3 >> 24) & 63u) + 9u * (x4 & 63u);
                        S[7] = ((x4 >> 6) & 63u) + 3u * ((x4 >> 12) & 63u) + 9u * ((x4 >> 18) & 63u);
                        any |= S[6] | S[7];
                    }
                }
                // data symbols -> stream order (the zero padding of a band's last block is not stored)
                const uint32_t yb = a.y_off + b + 9u * K * m;
#pragma unroll
                for (uint32_t p = 0; p < K; ++p) *T3_LP(uint8_t, yb + 9u * p) = (uint8_t)(d8[p] >> 3);
#ifdef T3_ABL_DEC_NO_CORRECT
                if (any == 0x7FFFFFFFu) {
#else
                if (any != 0) {                                                   // OLD:562: all-zero syndromes -> nothing to do
#endif
                    Fix fx;
                    if (!fx_correct<R>(S, fx, a.roots, a.fma_off)) atomicAdd(a.fail, 1u);
                    else {
#pragma unroll
                        for (int e = 0; e < R / 2; ++e)
                            if ((uint32_t)e < fx.np && fx.pos[e] < K) { const uint32_t ad = yb + 9u * fx.pos[e]; *T3_LP(uint8_t, ad) = (uint8_t)gfs(l8(ad), fx.mag[e]); }
                    }
                }
            }
        }
        __syncthreads();
        // ---------------- D5: symbols -> output units, staged in LDS ----------------
        const uint64_t unit0 = (uint64_t)tile * units_tile;
        const uint32_t n_here = (uint32_t)min((uint64_t)units_tile, a.n_units > unit0 ? a.n_units - unit0 : 0ull);
        if constexpr (TO_PIXELS) {
            // One lane = four consecutive triples: 52 symbols (13 aligned dwords of Y) -> 12 pixels = 72 bytes, stored straight
            // to memory (lanes are consecutive, so a wave writes one contiguous run).  Triples 0/2 and 1/3 share registers as
            // 16-bit halves; inverse of the encoder's splice (unpack_two_pixels OLD:706-722) in packed arithmetic.
            const uint32_t ntr = a.TS / 13u;                                       // triples in the tile (a multiple of 4)
            for (uint32_t j = tid; 4u * j < ntr; j += nthr) {
                uint32_t D[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) D[i] = *T3_LP(const uint32_t, a.y_off + 52u * j + 4u * i);
                uint32_t o[18];
#pragma unroll
                for (uint32_t pair = 0; pair < 2; ++pair) {                         // pair 0 = triples (0, 2), pair 1 = triples (1, 3)
                    u16x2 sy[13];
#pragma unroll
                    for (uint32_t i = 0; i < 13; ++i) {
                        const uint32_t lo = 13u * pair + i, hi = lo + 26u;          // byte offsets of the two symbols
                        // v_perm(S0, S1): selector bytes 0..3 pick from S1, 4..7 from S0, 0x0c = zero
                        const uint32_t sel = (lo & 3u) | 0x0c00u | ((4u + (hi & 3u)) << 16) | 0x0c000000u;
                        sy[i] = __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(D[hi >> 2], D[lo >> 2], sel));
                    }
                    u16x2 q, t, Y0, B0, R0, Y1, B1, R1, Y2, B2, R2;
                    q = pd9(sy[1]);  Y0 = sy[0] + (sy[1] - q * (uint16_t)9) * (uint16_t)27;  B0 = q + sy[2] * (uint16_t)3;
                    q = pd3(sy[4]);  R0 = sy[3] + (sy[4] - q * (uint16_t)3) * (uint16_t)27;  Y1 = q + sy[5] * (uint16_t)9;
                    q = pd3(sy[7]);  B1 = sy[6] + (sy[7] - q * (uint16_t)3) * (uint16_t)27;
                    t = pd9(sy[8]);  R1 = q + (sy[8] - t * (uint16_t)9) * (uint16_t)9;
                    q = pd3(sy[10]); Y2 = t + sy[9] * (uint16_t)3 + (sy[10] - q * (uint16_t)3) * (uint16_t)81;
                    t = pd9(sy[11]); B2 = q + (sy[11] - t * (uint16_t)9) * (uint16_t)9;      R2 = t + sy[12] * (uint16_t)3;
                    const uint32_t c[9] = {bits(Y0), bits(B0 - (uint16_t)40), bits(R0 - (uint16_t)40), bits(Y1), bits(B1 - (uint16_t)40), bits(R1 - (uint16_t)40),
                                           bits(Y2), bits(B2 - (uint16_t)40), bits(R2 - (uint16_t)40)};
                    // 16-bit output index of component i: first triple of the pair 9 pair + i, second + 18
                    if (pair == 0) {
#pragma unroll
                        for (int d = 0; d < 4; ++d) { o[d] = __builtin_amdgcn_perm(c[2 * d + 1], c[2 * d], 0x05040100u); o[9 + d] = __builtin_amdgcn_perm(c[2 * d + 1], c[2 * d], 0x07060302u); }
                        o[4] = c[8] & 0xFFFFu; o[13] = c[8] >> 16;                  // low halves of dwords 4 and 13; pair 1 supplies the high halves
                    } else {
                        o[4] |= c[0] << 16; o[13] |= c[0] & 0xFFFF0000u;
#pragma unroll
                        for (int d = 0; d < 4; ++d) { o[5 + d] = __builtin_amdgcn_perm(c[2 * d + 2], c[2 * d + 1], 0x05040100u); o[14 + d] = __builtin_amdgcn_perm(c[2 * d + 2], c[2 * d + 1], 0x07060302u); }
                    }
                }
                uint8_t* g = (uint8_t*)a.out + (unit0 + 12ull * j) * 6u;            // 8-byte aligned
                if (12u * j + 12u <= n_here) {
                    typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(8)));
                    typedef uint32_t v2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                    for (int d = 0; d < 4; ++d) *(v4u*)(g + 16 * d) = v4u{o[4 * d], o[4 * d + 1], o[4 * d + 2], o[4 * d + 3]};
                    *(v2u*)(g + 64) = v2u{o[16], o[17]};
                } else {                                                            // the frame's last pixels: per 16-bit component
#pragma unroll
                    for (uint32_t h = 0; h < 36; ++h)
                        if (12u * j + h / 3u < n_here) *(uint16_t*)(g + 2u * h) = (uint16_t)(o[h >> 1] >> (16u * (h & 1u)));
                }
            }
        } else {
            const uint32_t ng = a.TS / 26u;                                        // groups of 26 symbols -> 3 words (OLD:1022-1040)
            for (uint32_t j = tid; j < ng; j += nthr) {
                const uint32_t ya = a.y_off + 26u * j;
                uint32_t s[26], o[27];
#pragma unroll
                for (int i = 0; i < 26; ++i) s[i] = lds[ya + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = s[i];
                o[8] = s[8] - 9u * d9(s[8]);                                        // trits 24,25 of word 0, trit 26 = 0
                uint32_t carry = d9(s[8]);                                          // trit 0 of word 1
#pragma unroll
                for (int i = 0; i < 8; ++i) { const uint32_t lo = s[9 + i] - 9u * d9(s[9 + i]); o[9 + i] = carry + 3u * lo; carry = d9(s[9 + i]); }
                { const uint32_t lo = s[17] - 3u * d3(s[17]); o[17] = carry + 3u * lo; }            // trits 24,25 of word 1 (trit 25 = digit 0 of s17)
                uint32_t car2 = d3(s[17]);                                          // trits 0,1 of word 2
#pragma unroll
                for (int i = 0; i < 8; ++i) { const uint32_t lo = s[18 + i] - 3u * d3(s[18 + i]); o[18 + i] = car2 + 9u * lo; car2 = d3(s[18 + i]); }
                o[26] = car2;                                                       // trits 24,25 of word 2
                const uint32_t oa = a.o_off + 27u * j;
#pragma unroll
                for (int i = 0; i < 27; ++i) lds[oa + i] = (uint8_t)o[i];
            }
        }
        if constexpr (!TO_PIXELS) __syncthreads();
        // coalesced copy-out of the tile's units (raw words; pixels were stored directly)
        if constexpr (!TO_PIXELS) {
            constexpr uint32_t UB = TO_PIXELS ? 6u : 9u;
            const uint32_t nbytes = n_here * UB;
            uint8_t* g = (uint8_t*)a.out + unit0 * UB;
            const uint32_t mis = (uint32_t)(uintptr_t)g & 15u;                       // words: tile starts are only 8-byte aligned
            const uint32_t head = min(nbytes, (16u - mis) & 15u);
            for (uint32_t i = tid; i < head; i += nthr) g[i] = lds[a.o_off + i];
            const uint32_t nmain = (nbytes - head) >> 4;
            if (head % 4u == 0) {
                for (uint32_t i = tid; i < nmain; i += nthr) {
                    const uint32_t* s4 = (const uint32_t*)(lds + a.o_off + head + 16u * i);
                    *(uint4*)(g + head + 16u * i) = make_uint4(s4[0], s4[1], s4[2], s4[3]);
                }
            } else {
                for (uint32_t i = tid; i < 16u * nmain; i += nthr) g[head + i] = lds[a.o_off + head + i];
            }
            for (uint32_t i = head + 16u * nmain + tid; i < nbytes; i += nthr) g[i] = lds[a.o_off + i];
        }
        __syncthreads();
    }
}

template __global__ void decode_fixed_kernel<2, true>(const DecFxArgs);  template __global__ void decode_fixed_kernel<2, false>(const DecFxArgs);
template __global__ void decode_fixed_kernel<4, true>(const DecFxArgs);  template __global__ void decode_fixed_kernel<4, false>(const DecFxArgs);
template __global__ void decode_fixed_kernel<6, true>(const DecFxArgs);  template __global__ void decode_fixed_kernel<6, false>(const DecFxArgs);
template __global__ void decode_fixed_kernel<8, true>(const DecFxArgs);  template __global__ void decode_fixed_kernel<8, false>(const DecFxArgs);

}  // namespace t3
