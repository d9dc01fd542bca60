// t3_decode_fx2.h — second version of the fused FIXED decoder's block stages (decode_block OLD:546-662, descramble_symbol
// OLD:88-94), shared by t3_decode_fused.hip and t3_decode_stream.hip:
//   fx2_set      a wave decodes 32 blocks, two lanes per block: one 16-byte load per lane, one conflict-free T-table read per
//                symbol (descramble + trit expansion), syndromes by four v_mfma_i32_32x32x32_i8, data symbols -> stream order
//   fx2_single   one lane = one block: the single-error case in closed form (S_j = m alpha^{(j+1) p}: position and magnitude
//                from two logarithms, consistency of all r syndromes checked) -- exactly what Berlekamp-Massey + Chien +
//                Forney return for such syndromes
//   fx2_correct  one lane = one queued block: Berlekamp-Massey on T+1 coefficients (exact while L <= t, which every
//                correctable block satisfies), root table, Omega cut at deg sigma (exact when deg sigma = L = #roots); lanes
//                outside those conditions are handed to the full-length routine fx_correct of t3_decode_fx.h
#pragma once
#include "t3_decode_fx.h"
#include "t3_host.hpp"

namespace t3 {
namespace {

typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v16i_ __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(2))) U128a2_ { uint32_t v[4]; };

constexpr uint32_t SM = kFx2Small;     // small byte tables (27 entries: at most 7 dwords = 7 banks, conflict-free)
__device__ __forceinline__ uint32_t mod26(uint32_t u) { return min(u, u + 26u); }   // u = a - b as uint32, a, b < 26

// What a set leaves with the two lanes of a block: the r syndromes, h = 0 half (S_0 .. S_{r/2-1}, one byte each) and h = 1 half
struct Synd { uint32_t lo, hi; };

// geometry of one block seen by its two lanes
struct Blk { bool valid, first; uint32_t c0; const uint8_t* g; uint32_t yb; };

template <int R>
__device__ __forceinline__ Blk fx2_block_of(const uint32_t item, const uint32_t n_items, const uint32_t nb, const DevDiv& div_nb, const uint32_t tile,
                                            const uint8_t* body /* in + hdr_syms */, const uint32_t y_off) {
    constexpr uint32_t K = 26 - R;
    const uint32_t bi = min(__umulhi(item, div_nb.mul) >> div_nb.sh, 8u), m = item - bi * nb;     // nb >= 2
    const Row rw = row(bi);
    const uint64_t mg = (uint64_t)tile * nb + m;
    Blk b;
    b.valid = item < n_items && mg < rw.blocks;
    b.first = rw.body_off == 0 && mg == 0;
    b.c0 = (rw.boff6 + 2u * (uint32_t)(mg % 3u)) % 6u;                     // 26 == 2 (mod 6)
    b.g = body + rw.body_off + 26ull * mg;
    b.yb = y_off + bi + 9u * K * m;
    return b;
}

// One set: lane (n, h) = half h of block `item0 + n`.  `Lw` = the lane's 16 coded bytes (block bytes [10 h, 10 h + 16)), loaded by
// the caller (prefetched one tile ahead).
template <int R>
__device__ __forceinline__ Synd fx2_set(const Blk& b, const uint32_t (&Lw)[4], const uint32_t lane, const uint32_t af_off,
                                        const uint32_t cyc24, const uint32_t pre0, const uint32_t pre1) {
    constexpr uint32_t K = 26 - R, H = R / 2;
    const uint32_t n = lane & 31u, h = lane >> 5;
    // symbol q of this lane (position 13 h + q) = byte q of W: h = 1 starts at byte 3 of its load
    uint32_t W[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) W[i] = __builtin_amdgcn_alignbyte(Lw[i + 1], Lw[i], 3u * h);
    W[3] = Lw[3] >> (24u * h);
    // any byte >= 27 among the 13?  (b + 101) sets bit 7 exactly for b in 27..154, and a byte >= 155 has bit 7 set already
    uint32_t hi = (W[3] | (W[3] + 0x65u)) & 0x80u;
#pragma unroll
    for (int i = 0; i < 3; ++i) hi |= (W[i] | (W[i] + 0x65656565u)) & 0x80808080u;     // a carry out of a byte only ever adds set bits
    if (__builtin_amdgcn_ballot_w64(b.valid && hi != 0u) != 0) {                     // unpack3 semantics for non-canonical bytes (OLD:28-31)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t r = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const uint32_t c = (W[i] >> (8 * q)) & 0xFFu; r |= (c - 27u * d27(c)) << (8 * q); }
            W[i] = r;
        }
    }
    // scrambler state of position 13 h + q: cyc[(c0 + 13 h + q) mod 6], 13 == 1 (mod 6)
    uint32_t c0h = b.c0 + h; c0h -= c0h >= 6u ? 6u : 0u;
    const uint32_t cycs = cyc24 >> (2u * c0h);
    uint32_t vb[6];                                                               // T base per position class: state, own bank copy
#pragma unroll
    for (uint32_t q = 0; q < 6; ++q) vb[q] = ((cycs >> (2u * q)) & 3u) * (uint32_t)kSyndTState + ((uint32_t)kFx2T + 4u * n);
    const bool fst = b.first && h == 0;                                            // body symbols 0 and 1 see the pre-period states
    const uint32_t vb0 = fst ? pre0 * (uint32_t)kSyndTState + ((uint32_t)kFx2T + 4u * n) : vb[0];
    const uint32_t vb1 = fst ? pre1 * (uint32_t)kSyndTState + ((uint32_t)kFx2T + 4u * n) : vb[1];
    const uint32_t ya = b.yb + 117u * h;                                           // 9 * 13
    v16i_ acc = {81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81, 81};     // bias: trit sums in [-78, 78] -> [3, 159]
#pragma unroll
    for (uint32_t st = 0; st < 4; ++st) {
        v4i_ Bv = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t q = 4u * st + d;
            if (q >= 13u) continue;
            const uint32_t c = (W[st] >> (8u * d)) & 0xFFu;
            const uint32_t base = q == 0 ? vb0 : q == 1 ? vb1 : vb[q % 6u];
            const uint32_t x = *T3_LP(const uint32_t, (c << 7) + base);
            Bv[d] = (int)x;
            // data symbols -> stream order (byte 2 of the entry = the descrambled symbol): h = 0 holds positions 0..12, h = 1 13..25
            if (b.valid && (h == 0 || q + 13u < K)) *T3_LP(uint8_t, ya + 9u * q) = (uint8_t)(x >> 16);
        }
        const v4i_ Af = *T3_LP(const v4i_, af_off + 16u * (64u * st + lane));        // the syndrome matrix lives in LDS (80-VGPR budget)
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af, Bv, acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                                          // one K-step's table entries live at a time (80-VGPR budget)
    }
    uint32_t Pown = 0;
#pragma unroll
    for (uint32_t jj = 0; jj < H; ++jj) {
        uint32_t s = 0;
#pragma unroll
        for (uint32_t t = 0; t < 3; ++t) {
            const uint32_t x = (uint32_t)acc[3 * jj + t];
            const uint32_t r = x - 3u * (__umul24(x, 171u) >> 9);                    // x < 512
            s += r * (t == 0 ? 1u : t == 1 ? 3u : 9u);
        }
        Pown |= s << (8u * jj);
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(Pown, Pown, false, false);    // [0]: every lane sees the h = 0 half, [1]: the h = 1 half
    Synd r; r.lo = sw[0]; r.hi = sw[1];
    return r;
}

// Single-error closed form for the lane's own block.  Returns 0: not a single error (-> queue), 1: fixed (or only parity hit).
template <int R>
__device__ __forceinline__ uint32_t fx2_single(const Synd& sy, const uint32_t yb, const uint32_t FMA) {
    constexpr uint32_t K = 26 - R, H = R / 2;
    uint32_t lg[R]; bool ok = true;
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)R; ++j) {
        const uint32_t s = ((j < H ? sy.lo : sy.hi) >> (8u * (j % H))) & 0xFFu;
        lg[j] = l8(SM + kFx2LG + s);
        ok = ok && lg[j] != 0xFFu;
    }
    const uint32_t p = mod26(lg[1] - lg[0]);                                      // S_{j+1} / S_j = alpha^p
#pragma unroll
    for (uint32_t j = 1; j + 1 < (uint32_t)R; ++j) ok = ok && mod26(lg[j + 1] - lg[j]) == p;
    if (!ok) return 0u;
    if (p < K) {                                                                   // S_0 = m alpha^p
        const uint32_t m = l8(SM + kFx2EX + mod26(lg[0] - p));
        const uint32_t ad = yb + 9u * p;
        *T3_LP(uint8_t, ad) = (uint8_t)l8(FMA + (2u * 27u + m) * 27u + l8(ad));     // y - m = y + 2 m
    }
    return 1u;
}

// Berlekamp-Massey on T+1 coefficients + root table + short Omega.  Returns 0: corrected (fx filled), 1: uncorrectable,
// 2: outside the short routine's conditions (the caller runs fx_correct on this lane).
template <int R>
__device__ __forceinline__ uint32_t fx2_correct(const uint32_t* S, Fix& fx, const uint32_t* __restrict__ root_tbl, const uint32_t FMA) {
    auto fma = [FMA](uint32_t acc, uint32_t x, uint32_t y) -> uint32_t { return l8(FMA + (x * 27u + y) * 27u + acc); };   // acc + x y
    constexpr int T = R / 2;
    // sigma_1..T and x^m B as in fx_correct (B unscaled, the division kept in the scalar nbinv = -1 / d_old)
    uint32_t sg[T + 1], bx[T + 1];
#pragma unroll
    for (int i = 0; i <= T; ++i) { sg[i] = 0; bx[i] = 0; }
    sg[0] = 1; bx[1] = 1;
    uint32_t L = 0, nbinv = 2;                                                    // -1
    bool over = false;
#pragma unroll
    for (int n = 0; n < R; ++n) {
        uint32_t d = S[n];
#pragma unroll
        for (int i = 1; i <= T; ++i) if (i <= n) d = fma(d, sg[i], S[n - i]);
        const bool upd = d != 0 && 2u * L <= (uint32_t)n;
        const uint32_t nc = fma(0u, d, nbinv);                                     // -(d / d_old)
        uint32_t old[T + 1];
#pragma unroll
        for (int i = 1; i <= T; ++i) { old[i] = sg[i]; if (i <= n + 1) sg[i] = fma(old[i], nc, bx[i]); }
        old[0] = 1;
        if (upd) { L = (uint32_t)n + 1u - L; nbinv = l8(SM + kFx2NINV + d); over = over || L > (uint32_t)T; }
#pragma unroll
        for (int i = T; i >= 1; --i) bx[i] = upd ? old[i - 1] : bx[i - 1];
        bx[0] = 0;
    }
    uint32_t deg = 0;
#pragma unroll
    for (int i = 1; i <= T; ++i) if (sg[i] != 0) deg = (uint32_t)i;
    if (over || deg != L) return 2u;                                               // longer register, or deg sigma < L: full-length routine
    uint32_t ridx = sg[T];
#pragma unroll
    for (int q = T - 1; q >= 1; --q) ridx = ridx * 27u + sg[q];
    const uint32_t roots = root_tbl[ridx];
    const uint32_t np = (uint32_t)__popc(roots);
    if (np != deg) return 1u;
    // Omega = S sigma mod x^R has degree < deg sigma here (the register generates all R syndromes and has deg distinct roots)
    uint32_t Om[T];
#pragma unroll
    for (int q = 0; q < T; ++q) {
        uint32_t acc = S[q];
#pragma unroll
        for (int j = 1; j <= q; ++j) acc = fma(acc, S[q - j], sg[j]);
        Om[q] = acc;
    }
    uint32_t s22 = 0;                                                              // sigma' = sigma1 + 2 sigma2 x (+ sigma4 x^3)
    if constexpr (T >= 2) s22 = fma(sg[2], 1u, sg[2]);
    uint32_t r = roots;
#pragma unroll
    for (int e = 0; e < T; ++e) {
        if ((uint32_t)e < np) {
            const uint32_t p = (uint32_t)__ffs((int)r) - 1u; r &= r - 1u;
            const uint32_t xi = l8(SM + kFx2EX + (p == 0 ? 0u : 26u - p));
            uint32_t num = Om[T - 1];
#pragma unroll
            for (int q = T - 2; q >= 0; --q) num = fma(Om[q], num, xi);
            uint32_t den = fma(sg[1], s22, xi);
            if constexpr (T >= 4) den = fma(den, fma(0u, fma(0u, sg[4], xi), xi), xi);
            if (den == 0) return 1u;                                                // OLD:656
            fx.pos[e] = p; fx.mag[e] = fma(0u, l8(SM + kFx2NEG + num), l8(SM + kFx2INV + den));   // OLD:657; FIXED subtracts it
        }
    }
    fx.np = np;
    return 0u;
}

}  // namespace
}  // namespace t3
