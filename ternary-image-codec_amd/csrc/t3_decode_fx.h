// t3_decode_fx.h — device helpers shared by the fused FIXED-mode decoders (t3_decode_fused.hip, t3_decode_stream.hip):
// LDS addressing, GF(27) table access, the mod-3 fold of the syndrome accumulators, the in-register corrector
// (Berlekamp-Massey / root table / Forney) and the packed symbol -> pixel conversion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// D1-D4 for one lane = one RS block (see t3_decode_fused.hip): 26 coded symbols at `body_off + 26 mg` of the band -> descramble ->
// syndromes through the LUT at LDS offset `lut` -> correction -> the K data symbols to LDS at yb, yb + 9, ...
struct FxCtx { const uint8_t* in; uint64_t in_bytes; uint32_t hdr_syms, cyc24, pre0, pre1; uint32_t* fail; const uint32_t* roots; uint32_t fma_off; };
template <int R>
__device__ __forceinline__ void fx_block(const FxCtx& cx, const Row rw, const uint64_t mg, const uint32_t lut, const uint32_t yb) {
    constexpr uint32_t K = 26 - R, SLAB = R == 8 ? 768u : 512u;
    const uint8_t* g = cx.in + cx.hdr_syms + rw.body_off + 26ull * mg;
    const uint32_t sh = ((uint32_t)(uintptr_t)g & 3u) * 8u;
    uint32_t w[7];
    if (g + 28 <= cx.in + cx.in_bytes) {
        const uint32_t* p = (const uint32_t*)((uintptr_t)g & ~(uintptr_t)3);
        uint32_t dw[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) dw[i] = __builtin_nontemporal_load(p + i);     // the coded stream is read once (as in t3_decode_fused.hip)
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
    for (int q = 0; q < 6; ++q) dbase[q] = DSC + 32u * ((cx.cyc24 >> (2u * (c0 + q))) & 3u);
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
        if (i < 2 && first) base = DSC + 32u * (i == 0 ? cx.pre0 : cx.pre1);
        d8[i] = l8(base + c[i]);                                      // descrambled symbol * 4 (byte offset of its LUT dword)
    }
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0;
#pragma unroll
#ifdef T3_ABL_DEC_NO_SYND
    for (uint32_t i = 0; i < 0; ++i) {
#else
    for (uint32_t i = 0; i < 26; ++i) {
#endif
        // four (five) 27-dword tables per position: 27 consecutive dwords sit in 27 different banks, so these gathers never conflict
        acc0 += *T3_LP(const uint32_t, lut + i * SLAB + d8[i]);
        acc1 += *T3_LP(const uint32_t, lut + i * SLAB + 128u + d8[i]);
        acc2 += *T3_LP(const uint32_t, lut + i * SLAB + 256u + d8[i]);
        if constexpr (R >= 6) acc3 += *T3_LP(const uint32_t, lut + i * SLAB + 384u + d8[i]);
        if constexpr (R == 8) acc4 += *T3_LP(const uint32_t, lut + i * SLAB + 512u + d8[i]);
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
#pragma unroll
    for (uint32_t p = 0; p < K; ++p) *T3_LP(uint8_t, yb + 9u * p) = (uint8_t)(d8[p] >> 2);
#ifdef T3_ABL_DEC_NO_CORRECT
    if (any == 0x7FFFFFFFu) {
#else
    if (any != 0) {                                                   // OLD:562: all-zero syndromes -> nothing to do
#endif
        Fix fx;
        if (!fx_correct<R>(S, fx, cx.roots, cx.fma_off)) atomicAdd(cx.fail, 1u);
        else {
#pragma unroll
            for (int e = 0; e < R / 2; ++e)
                if ((uint32_t)e < fx.np && fx.pos[e] < K) { const uint32_t ad = yb + 9u * fx.pos[e]; *T3_LP(uint8_t, ad) = (uint8_t)gfs(l8(ad), fx.mag[e]); }
        }
    }
}

// 52 symbols (13 aligned dwords) = four 13-symbol triples -> 12 pixels = 18 dwords.  Triples 0/2 and 1/3 share registers as
// 16-bit halves; inverse of the encoder's splice (unpack_two_pixels OLD:706-722) in packed arithmetic.
__device__ __forceinline__ void px12_from_syms(const uint32_t* D, uint32_t* o) {
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
}

// 26 symbols at LDS address ya -> three 26-trit words (27 bytes) at LDS address oa (OLD:1022-1040: trit 26 of a word = 0)
__device__ __forceinline__ void words3_from_syms(const uint32_t ya, const uint32_t oa) {
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
#pragma unroll
    for (int i = 0; i < 27; ++i) lds[oa + i] = (uint8_t)o[i];
}

// nbytes of LDS at `src` -> global g, 16-byte stores where the alignment allows
__device__ __forceinline__ void copy_out_lds(uint8_t* g, const uint32_t src, const uint32_t nbytes, const uint32_t tid, const uint32_t nthr) {
    const uint32_t mis = (uint32_t)(uintptr_t)g & 15u;
    const uint32_t head = min(nbytes, (16u - mis) & 15u);
    for (uint32_t i = tid; i < head; i += nthr) g[i] = lds[src + i];
    const uint32_t nmain = (nbytes - head) >> 4;
    if (((src + head) & 3u) == 0) {
        for (uint32_t i = tid; i < nmain; i += nthr) {
            const uint32_t* s4 = (const uint32_t*)(lds + src + head + 16u * i);
            *(uint4*)(g + head + 16u * i) = make_uint4(s4[0], s4[1], s4[2], s4[3]);
        }
    } else {
        for (uint32_t i = tid; i < 16u * nmain; i += nthr) g[head + i] = lds[src + head + i];
    }
    for (uint32_t i = head + 16u * nmain + tid; i < nbytes; i += nthr) g[i] = lds[src + i];
}
}  // namespace

}  // namespace t3
