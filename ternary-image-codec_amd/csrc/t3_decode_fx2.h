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

#ifndef T3_DEC_FOLD_VALU
#define T3_DEC_FOLD_VALU 0   // 1: mod-3 fold on the vector ALU instead of three byte-table reads per syndrome -- measured slower (0.153 -> 0.158 ms): the kernel is bound by vector issue
#endif

namespace t3 {
namespace {

typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v16i_ __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(2))) U128a2_ { uint32_t v[4]; };

// SMB + kFx2Small: LDS offset of the small byte tables (27 entries: at most 7 dwords = 7 banks, conflict-free)
__device__ __forceinline__ uint32_t mod26(uint32_t u) { return min(u, u + 26u); }   // u = a - b as uint32, a, b < 26

// What a set leaves with the two lanes of a block: the r syndromes, h = 0 half (S_0 .. S_{r/2-1}, one byte each) and h = 1 half
struct Synd { uint32_t lo, hi; };

// Geometry of one block seen by its two lanes, per set a lane works on: a constant word and a running byte offset.
//   Geo (constant): y_rel = band + 9 K m, where the block's symbols start in a tile's symbol buffer (m = block within the tile: also
//                   the queue tag) | cbase << 16, the scrambler phase (boff6 + 2 (m mod 3)) mod 6 of the block in tile 0 | has-an-item << 19
//   off (running):  byte offset of the block from the body's first symbol; a workgroup's next tile is nb * grid blocks further on in
//                   every band, so it advances by one wave-uniform constant per tile.
// Round 2 rebuilt all of this from (item, tile) for every set and every prefetch -- a table row read and ~20 vector instructions, four
// times per pass: an eighth of the kernel's vector instructions.
typedef uint32_t Geo;
struct Blk { bool valid, first; uint32_t c0, off, yb; };            // off: byte offset of the block from the body's first symbol
constexpr uint32_t kFx2Dummy = 384;                                   // 128 bytes of LDS that take the writes of lanes without a block

template <int R>
__device__ __forceinline__ Geo fx2_geo(const uint32_t item, const uint32_t n_items, const uint32_t nb, const DevDiv& div_nb, const uint32_t tile0, uint32_t& off0) {
    constexpr uint32_t K = 26 - R;
    const uint32_t bi = min(__umulhi(item, div_nb.mul) >> div_nb.sh, 8u), m = item - bi * nb;       // nb >= 2
    const uint32_t m3 = m - 3u * ((m * 683u) >> 11);                                                // m < 2048
    const Row rw = row(bi);
    uint32_t cb = rw.boff6 + 2u * m3; cb -= cb >= 6u ? 6u : 0u;                                     // 26 == 2 (mod 6)
    off0 = (uint32_t)rw.body_off + 26u * (tile0 * nb + m);                                          // the coded stream is shorter than 4 GiB (t3_api_decode.cpp)
    return (bi + 9u * K * m) | cb << 16 | (item < n_items ? 1u << 19 : 0u);
}
// A band's blocks end inside its last tile only (the bands of a uniform-k frame differ by at most one block): every other tile holds a
// block for every item, so the per-lane test against the band's block count -- a table row read -- runs for the last tile alone.
template <int R>
__device__ __forceinline__ bool fx2_has_block(const Geo g, const uint32_t tile, const uint32_t n_tiles, const uint32_t nb) {
    constexpr uint32_t K = 26 - R;
    bool v = (g >> 19) != 0u;
    if (tile + 1u >= n_tiles) {                                                                     // (wave-uniform)
        const uint32_t yr = g & 0xFFFFu, m = yr / (9u * K), bi = yr - 9u * K * m;
        v = v && tile * nb + m < row(bi).blocks;
    }
    return v;
}
// u2 = 2 ((tile nb) mod 3), wave-uniform
__device__ __forceinline__ Blk fx2_block(const Geo g, const uint32_t off, const bool valid, const uint32_t u2, const uint32_t y_off) {
    Blk b;
    b.valid = valid;
    b.first = off == 0u;                                                  // block 0 of band 0
    b.c0 = ((g >> 16) & 7u) + u2;                                        // scrambler phase of the block's first symbol, NOT reduced mod 6 (<= 9)
    b.off = off;
    b.yb = y_off + (g & 0xFFFFu);
    return b;
}

// One set: lane (n, h) = half h of block `item0 + n`.  `Lw` = the lane's 16 coded bytes (block bytes [10 h, 10 h + 16)), loaded by
// the caller (prefetched ahead).  The three mod-3 fold tables M_t[x] = 3^t ((x - 81) mod 3) are 160 B each.
// TCOP bank copies of the T table at LDS offset TBASE (32: conflict-free; 16: lanes n and n + 16 share a copy, two-way conflicts,
// half the space); MT: LDS offset of the fold tables (a compile-time constant, so that it rides in the instruction's offset field).
// RM >= R: the A operand at af_off is the evaluation matrix of RS(26, 26 - RM).  The roots of a code with fewer parity symbols are the
// first ones of a code with more (alpha^1 .. alpha^r), so one operand serves both codes of a two-code frame (t3_decode_uep.hip):
// RM syndromes are folded, the first R of them returned.
template <int R, uint32_t TCOP, uint32_t TBASE, uint32_t MT, uint32_t SMB = 0, int RM = R>
__device__ __forceinline__ Synd fx2_set(const Blk& b, const uint32_t (&Lw)[4], const uint32_t lane, const uint32_t af_off, const uint32_t pat_off) {
    constexpr uint32_t mt = MT;
    constexpr uint32_t K = 26 - R, H = RM / 2, HB = R / 2;
    static_assert(RM >= R, "matrix of the code with more parity");
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
    // Scrambler state of position 13 h + q: cyc[(c0 + 13 h + q) mod 6], 13 == 1 (mod 6).  The T table is state-major (entry 27 s + c), so
    // the state rides in the symbol byte itself: one 16-byte row of 27 s per position class start (host-built, t3_api_decode.cpp; row 6:
    // the stream's first block, whose symbols 0 and 1 see the pre-period states) is added to the 13 bytes -- c + 27 s <= 80, no carries --
    // instead of a table base per position class (round 2: twelve multiply-adds and four selects per set).  c0 comes un-reduced
    // (cbase + 2 ((tile nb) mod 3) <= 9), the row table simply repeats.
    const uint32_t prow = (b.first && h == 0u) ? 11u : b.c0 + h;                    // c0 + h <= 10 un-reduced: rows 6..10 repeat rows 0..4
    const v4i_ P = *T3_LP(const v4i_, pat_off + 16u * prow);
#pragma unroll
    for (int i = 0; i < 4; ++i) W[i] += (uint32_t)P[i];
    const uint32_t tb0 = 4u * (n % TCOP);                                          // own bank copy; entries are 4 TCOP bytes apart: disjoint bits
    // data symbols -> stream order (byte 2 of a T entry = the descrambled symbol): h = 0 holds positions 0..12, h = 1 13..25, of
    // which K..25 are parity; lanes without a block write into a dummy area instead of being masked off store by store
    const uint32_t ya = b.valid ? b.yb + 117u * h : SMB + kFx2Dummy;               // 9 * 13
    // bias 64 (the largest inline constant: the first MFMA takes it as its C operand, no register initialisation): trit sums in
    // [-78, 78] -> [-14, 142], and the fold tables are entered at index + 14
    v16i_ acc = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};
    // The table reads of K-step st + 1 are issued before the MFMA of step st (one step of entries in flight beside the one being
    // consumed: the LDS latency of a step hides under the previous step's MFMA), no further (80-VGPR budget).
    auto fetch = [&](const uint32_t st, v4i_& Bv) {
        Bv = v4i_{0, 0, 0, 0};
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) {
            const uint32_t q = 4u * st + d;
            if (q >= 13u) continue;
            const uint32_t c = __builtin_amdgcn_ubfe(W[st], 8u * d, 8u);              // 27 state + symbol
            Bv[d] = (int)*T3_LP(const uint32_t, TBASE + ((c * (4u * TCOP)) | tb0));
        }
    };
    auto emit = [&](const uint32_t st, const v4i_& Bv) {       // data symbols of the step -> stream order
#pragma unroll
        for (uint32_t d = 0; d < 4; ++d) { const uint32_t q = 4u * st + d; if (q < 13u && q + 13u < K) *T3_LP(uint8_t, ya + 9u * q) = (uint8_t)((uint32_t)Bv[d] >> 16); }   // both halves hold data here
        if (4u * st + 3u + 13u >= K) {                                              // positions >= K - 13 of the upper half are parity: one masked region per step
            if (h == 0) {
#pragma unroll
                for (uint32_t d = 0; d < 4; ++d) { const uint32_t q = 4u * st + d; if (q < 13u && q + 13u >= K) *T3_LP(uint8_t, ya + 9u * q) = (uint8_t)((uint32_t)Bv[d] >> 16); }
            }
        }
    };
    v4i_ Bq[2];                                                                   // double buffer by step parity (no copies)
    fetch(0, Bq[0]);
#pragma unroll
    for (uint32_t st = 0; st < 4; ++st) {
        if (st < 3u) fetch(st + 1u, Bq[(st + 1u) & 1u]);
        // the syndrome matrix lives in LDS (80-VGPR budget): three full steps, then step 3 of which only dword 0 (position 12) is used
        v4i_ Af = {0, 0, 0, 0};
        if (st < 3u) Af = *T3_LP(const v4i_, af_off + 16u * (64u * st + lane)); else Af[0] = *T3_LP(const int, af_off + 3072u + 4u * lane);
        emit(st, Bq[st & 1u]);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Af, Bq[st & 1u], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t Pown = 0;
#pragma unroll
    for (uint32_t jj = 0; jj < H; ++jj) {
#if T3_DEC_FOLD_VALU
        // mod-3 fold on the vector ALU (the LDS pipe is the busier one: three byte-table reads per syndrome become twelve integer
        // operations): a biased trit sum x in [-14, 142] has x + 14 = (trit sum) + 78 == trit sum (mod 3) and x + 14 < 512, so
        // q = ((x + 14) * 171) >> 9 = floor((x + 14) / 3); the three residues x_t + 14 - 3 q_t are weighted 1 / 3 / 9
        uint32_t u[3];
#pragma unroll
        for (uint32_t t = 0; t < 3; ++t) {
            const uint32_t x = (uint32_t)acc[3 * jj + t];
            const uint32_t q = __umul24(x + 14u, 171u) >> 9;                          // (one v_mad_u32_u24 + shift)
            u[t] = x - 3u * q;                                                        // residue - 14
        }
        const uint32_t s = u[0] + 3u * u[1] + 9u * u[2] + 14u * 13u;
#else
        // mod-3 fold and 3^t weight by byte tables
        const uint32_t s = l8(mt + 14u + (uint32_t)acc[3 * jj]) + l8(mt + 174u + (uint32_t)acc[3 * jj + 1]) + l8(mt + 334u + (uint32_t)acc[3 * jj + 2]);
#endif
        Pown |= s << (8u * jj);
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(Pown, Pown, false, false);    // [0]: every lane sees the h = 0 half, [1]: the h = 1 half
    Synd r; r.lo = sw[0]; r.hi = sw[1];
    if constexpr (RM != R) {                                                       // S_0 .. S_{R-1} of the RM folded: bytes of sw[0] | sw[1] << 8 H
        const uint64_t V = (uint64_t)sw[0] | (uint64_t)sw[1] << (8u * H);
        constexpr uint64_t M = (1ull << (8u * HB)) - 1ull;
        r.lo = (uint32_t)(V & M); r.hi = (uint32_t)((V >> (8u * HB)) & M);
    }
    return r;
}

// Single-error closed form for the lane's own block.  Returns 0: not a single error (-> queue), 1: fixed (or only parity hit).
template <int R, uint32_t SMB = 0>
__device__ __forceinline__ uint32_t fx2_single(const Synd& sy, const uint32_t yb, const uint32_t FMA) {
    constexpr uint32_t SM = SMB + kFx2Small;
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
// Table index of a + x y = FMA + 729 x + 27 y + a; multiplication commutes, so the operand that is known early carries the
// factor 729 (and the table base) and the other one the factor 27: one v_add3 per multiply-accumulate.
template <int R, uint32_t SMB = 0, bool WIDE = true>
__device__ __forceinline__ uint32_t fx2_correct(const uint32_t* S, Fix& fx, const uint32_t* __restrict__ root_tbl, const uint32_t FMA) {
    constexpr uint32_t SM = SMB + kFx2Small;
    constexpr int T = R / 2;
    auto tab = [](uint32_t idx) -> uint32_t { return l8(idx); };
    uint32_t Sx[R];                                                                // FMA + 729 S_j
#pragma unroll
    for (int j = 0; j < R; ++j) Sx[j] = __umul24(S[j], 729u) + FMA;
    // sigma_1..T (plain and x 27) and bs = -x^m B / d_old (x 27).  Round 3: the update vector is kept ALREADY divided by the
    // discrepancy it stems from, so sigma += d * bs is one table level behind d (round 2 kept B unscaled and first formed the scalar
    // -d / d_old: one more dependent read in each of the six iterations of a chain that is bound by its length); a length change
    // rescales the old sigma by -1 / d with T - 1 reads that nobody waits for.  The field elements are the same.
    uint32_t sg[T + 1], sg27[T + 1], bs27[T + 1];
#pragma unroll
    for (int i = 0; i <= T; ++i) { sg[i] = 0; sg27[i] = 0; bs27[i] = 0; }
    sg[0] = 1; sg27[0] = 27; bs27[1] = 27u * 2u;                                    // B = 1, d_old = 1: bs = -x
    uint32_t L = 0;
    bool over = false;
#pragma unroll
    for (int n = 0; n < R; ++n) {
        uint32_t d = S[n];
#pragma unroll
        for (int i = 1; i <= T; ++i) if (i <= n) d = tab(Sx[n - i] + sg27[i] + d);   // d += sigma_i S_{n-i}
        const bool upd = d != 0 && 2u * L <= (uint32_t)n;
        const uint32_t dx = __umul24(d, 729u) + FMA;
        uint32_t old27[T + 1];
#pragma unroll
        for (int i = 0; i <= T; ++i) old27[i] = sg27[i];
#pragma unroll
        for (int i = 1; i <= T; ++i) if (i <= n + 1) { sg[i] = tab(dx + bs27[i] + sg[i]); sg27[i] = 27u * sg[i]; }   // sigma + d bs
        // the next update vector: x bs, or after a length change -x sigma_old / d
        const uint32_t ninv = l8(SM + kFx2NINV + d), nix = __umul24(ninv, 729u) + FMA;
        uint32_t nb27[T + 1];
        nb27[0] = 0; nb27[1] = 27u * ninv;
#pragma unroll
        for (int i = 2; i <= T; ++i) nb27[i] = (i <= n + 1) ? 27u * tab(nix + old27[i - 1]) : 0u;                  // (sigma_old has degree <= n)
        if (upd) { L = (uint32_t)n + 1u - L; over = over || L > (uint32_t)T; }
#pragma unroll
        for (int i = T; i >= 1; --i) bs27[i] = upd ? nb27[i] : bs27[i - 1];
        bs27[0] = 0;
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
        for (int j = 1; j <= q; ++j) acc = tab(Sx[q - j] + sg27[j] + acc);
        Om[q] = acc;
    }
    uint32_t s22 = 0;                                                              // sigma' = sigma1 + 2 sigma2 x (+ sigma4 x^3)
    if constexpr (T >= 2) s22 = tab(FMA + 729u + sg27[2] + sg[2]);
    const uint32_t s22_27 = 27u * s22;
    if constexpr (WIDE) {
        // Forney for the (up to) T roots side by side: no branch per root and no early return inside, so the T chains of dependent table reads
        // (position -> x^-1 -> Omega(x^-1) -> value: seven levels) are issued interleaved instead of one after the other; a lane's unused
        // slots run on position 0 and are discarded (round 3: the correction is one long chain of LDS round trips, ~170 cycles each)
        uint32_t r = roots, bad = 0u;
        uint32_t pe[T], xix[T];
    #pragma unroll
        for (int e = 0; e < T; ++e) {
            const bool have = (uint32_t)e < np;
            const uint32_t p = have ? (uint32_t)__ffs((int)r) - 1u : 0u; r &= r - 1u;
            pe[e] = p;
            xix[e] = __umul24(l8(SM + kFx2EX + (p == 0 ? 0u : 26u - p)), 729u) + FMA;
        }
        uint32_t num[T], den[T];
    #pragma unroll
        for (int e = 0; e < T; ++e) { num[e] = Om[T - 1]; den[e] = tab(xix[e] + s22_27 + sg[1]); }
    #pragma unroll
        for (int q = T - 2; q >= 0; --q) {
    #pragma unroll
            for (int e = 0; e < T; ++e) num[e] = tab(xix[e] + 27u * num[e] + Om[q]);
        }
        if constexpr (T >= 4) {
    #pragma unroll
            for (int e = 0; e < T; ++e) { const uint32_t x2 = tab(xix[e] + sg27[4]); const uint32_t x3 = tab(xix[e] + 27u * x2); den[e] = tab(xix[e] + 27u * x3 + den[e]); }
        }
    #pragma unroll
        for (int e = 0; e < T; ++e) {
            const bool have = (uint32_t)e < np;
            bad |= (have && den[e] == 0u) ? 1u : 0u;                                    // OLD:656
            fx.pos[e] = pe[e]; fx.mag[e] = tab(FMA + 729u * l8(SM + kFx2NEG + num[e]) + 27u * l8(SM + kFx2INV + den[e]));   // OLD:657; FIXED subtracts it
        }
        if (bad) return 1u;
    } else {                                                                       // root by root (the two-code kernel of RS(26,20) + RS(26,22): the wide form costs it a spilled register inside the tile loop)
        uint32_t r = roots;
#pragma unroll
        for (int e = 0; e < T; ++e) {
            if ((uint32_t)e < np) {
                const uint32_t p = (uint32_t)__ffs((int)r) - 1u; r &= r - 1u;
                const uint32_t xi = l8(SM + kFx2EX + (p == 0 ? 0u : 26u - p));
                const uint32_t xix = __umul24(xi, 729u) + FMA;
                uint32_t num = Om[T - 1];
#pragma unroll
                for (int q = T - 2; q >= 0; --q) num = tab(xix + 27u * num + Om[q]);
                uint32_t den = tab(xix + s22_27 + sg[1]);
                if constexpr (T >= 4) { const uint32_t x2 = tab(xix + sg27[4]); const uint32_t x3 = tab(xix + 27u * x2); den = tab(xix + 27u * x3 + den); }
                if (den == 0) return 1u;                                                // OLD:656
                fx.pos[e] = p; fx.mag[e] = tab(FMA + 729u * l8(SM + kFx2NEG + num) + 27u * l8(SM + kFx2INV + den));   // OLD:657; FIXED subtracts it
            }
        }
    }
    fx.np = np;
    return 0u;
}

// The full-length routine (fx_correct of t3_decode_fx.h: sigma on R + 2 coefficients, every Omega coefficient) with all of its
// arithmetic on the multiply-accumulate table and the small byte tables, for the kernels that do not stage FxTables.
template <int R, uint32_t SMB = 0>
__device__ __forceinline__ bool fx2_correct_full(const uint32_t* S, Fix& fx, const uint32_t* __restrict__ root_tbl, const uint32_t FMA) {
    constexpr uint32_t SM = SMB + kFx2Small;
    auto fma = [FMA](uint32_t acc, uint32_t x, uint32_t y) -> uint32_t { return l8(FMA + (x * 27u + y) * 27u + acc); };   // acc + x y
    constexpr int T = R / 2, NP = R + 2;
    uint32_t sg[NP], bx[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) { sg[i] = 0; bx[i] = 0; }
    sg[0] = 1; bx[1] = 1;
    uint32_t L = 0, nbinv = 2;
#pragma unroll
    for (int n = 0; n < R; ++n) {
        uint32_t d = S[n];
#pragma unroll
        for (int i = 1; i <= n; ++i) d = fma(d, sg[i], S[n - i]);
        const bool upd = d != 0 && 2u * L <= (uint32_t)n;
        const uint32_t nc = fma(0u, d, nbinv);
        uint32_t old[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) { old[i] = sg[i]; if (i <= n + 1) sg[i] = fma(old[i], nc, bx[i]); }
        if (upd) { L = (uint32_t)n + 1u - L; nbinv = l8(SM + kFx2NINV + d); }
#pragma unroll
        for (int i = NP - 1; i >= 1; --i) bx[i] = upd ? old[i - 1] : bx[i - 1];
        bx[0] = 0;
    }
    uint32_t deg = 0;
#pragma unroll
    for (int i = 1; i < NP; ++i) if (sg[i] != 0) deg = (uint32_t)i;
    fx.np = 0;
    if (deg > (uint32_t)T) return false;
    uint32_t ridx = sg[T];
#pragma unroll
    for (int q = T - 1; q >= 1; --q) ridx = ridx * 27u + sg[q];
    const uint32_t roots = root_tbl[ridx];
    const uint32_t np = (uint32_t)__popc(roots);
    if (np != deg) return false;
    uint32_t Om[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        uint32_t acc = S[q];
#pragma unroll
        for (int j = 1; j <= T; ++j) if (j <= q) acc = fma(acc, S[q - j], sg[j]);
        Om[q] = acc;
    }
    uint32_t r = roots;
#pragma unroll
    for (int e = 0; e < T; ++e) {
        if ((uint32_t)e < np) {
            const uint32_t p = (uint32_t)__ffs((int)r) - 1u; r &= r - 1u;
            const uint32_t xi = l8(SM + kFx2EX + (p == 0 ? 0u : 26u - p));
            uint32_t num = Om[R - 1];
#pragma unroll
            for (int q = R - 2; q >= 0; --q) num = fma(Om[q], num, xi);
            uint32_t den = fma(sg[1], fma(sg[2], 1u, sg[2]), xi);
            if constexpr (T >= 4) den = fma(den, fma(0u, fma(0u, sg[4], xi), xi), xi);
            if (den == 0) return false;
            fx.pos[e] = p; fx.mag[e] = fma(0u, l8(SM + kFx2NEG + num), l8(SM + kFx2INV + den));
        }
    }
    fx.np = np;
    return true;
}

// One queued block: syndromes -> corrections patched into the symbol buffer at yb; returns false for an uncorrectable block.
template <int R, uint32_t SMB = 0, bool WIDE = true>
__device__ __forceinline__ bool fx2_fix_block(const uint32_t lo, const uint32_t hi, const uint32_t yb, const uint32_t* __restrict__ root_tbl, const uint32_t FMA) {
    constexpr uint32_t K = 26 - R, H = R / 2;
    uint32_t S[R];
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)R; ++j) S[j] = ((j < H ? lo : hi) >> (8u * (j % H))) & 0xFFu;
    Fix fx; fx.np = 0;
    uint32_t rc = fx2_correct<R, SMB, WIDE>(S, fx, root_tbl, FMA);
    if (rc == 2u) rc = fx2_correct_full<R, SMB>(S, fx, root_tbl, FMA) ? 0u : 1u;      // longer register than t: the full-length routine decides
    if (rc != 0u) return false;
    if constexpr (WIDE) {
        // the patches side by side as well: a slot that patches nothing (no root, or a parity position) works on the dummy bytes
        uint32_t ad[R / 2], yv[R / 2];
    #pragma unroll
        for (int q = 0; q < R / 2; ++q) { ad[q] = ((uint32_t)q < fx.np && fx.pos[q] < K) ? yb + 9u * fx.pos[q] : SMB + kFx2Dummy + 4u * (uint32_t)q; yv[q] = l8(ad[q]); }
    #pragma unroll
        for (int q = 0; q < R / 2; ++q) yv[q] = l8(FMA + (54u + fx.mag[q]) * 27u + min(yv[q], 26u));   // y - m = y + 2 m  (dummy bytes may hold anything)
    #pragma unroll
        for (int q = 0; q < R / 2; ++q) *T3_LP(uint8_t, ad[q]) = (uint8_t)yv[q];
    } else {
#pragma unroll
        for (int q = 0; q < R / 2; ++q)
            if ((uint32_t)q < fx.np && fx.pos[q] < K) { const uint32_t ad = yb + 9u * fx.pos[q]; *T3_LP(uint8_t, ad) = (uint8_t)l8(FMA + (54u + fx.mag[q]) * 27u + l8(ad)); }   // y - m = y + 2 m
    }
    return true;
}

// E1 for the 64 blocks two sets leave with a wave (lower half-wave: set A, upper: set B): single errors fixed in place, the other
// flagged blocks appended to the queue at q_off (syndromes, 8 bytes; block tags, 2 bytes, behind qcap entries); a block that
// finds the queue full is corrected on the spot.
template <int R, uint32_t SMB = 0, bool WIDE = true>
__device__ __forceinline__ void fx2_own_blocks(const uint32_t* __restrict__ roots, const uint32_t fma_off, uint32_t* fail, const Synd& sA, const Synd& sB, const Blk& bA, const Blk& bB, const uint32_t tag,
                                               const uint32_t lane, const uint32_t cnt_addr, const uint32_t q_off, const uint32_t qcap) {
    const uint32_t h = lane >> 5;
    Synd own; own.lo = h ? sB.lo : sA.lo; own.hi = h ? sB.hi : sA.hi;
    const bool valid = h ? bB.valid : bA.valid;
    const uint32_t yb = h ? bB.yb : bA.yb;
    bool flagged = valid && (own.lo | own.hi) != 0u;                                // OLD:562: all-zero syndromes -> nothing to do
    if (flagged) flagged = fx2_single<R, SMB>(own, yb, fma_off) == 0u;
    const uint64_t bal = __builtin_amdgcn_ballot_w64(flagged);
    if (bal != 0ull) {                                                              // wave-aggregated append: one LDS atomic per wave
        const uint32_t cnt = (uint32_t)__popcll(bal), first = (uint32_t)__builtin_ctzll(bal);
        uint32_t base = 0;
        // (the lane number is rebuilt here: as a value kept from the kernel's first lines it gets spilled, and its reload waits for every
        // load in flight)
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == first) base = __hip_atomic_fetch_add((uint32_t*)__builtin_assume_aligned(lds + cnt_addr, 4), cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = __builtin_amdgcn_readlane(base, (int)first);
        if (flagged) {
            const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
            if (__builtin_expect(slot < qcap, 1)) {
                *T3_LP(u32x2, q_off + 8u * slot) = u32x2{own.lo, own.hi};              // the r syndromes ...
                *T3_LP(uint16_t, q_off + 8u * qcap + 2u * slot) = (uint16_t)tag;       // ... and the block: where its symbols start in the tile's buffer
            } else if (!fx2_fix_block<R, SMB, WIDE>(own.lo, own.hi, yb, roots, fma_off)) atomicAdd(fail, 1u);   // queue full (cold)
        }
    }
}

// BM for queue entry e: the block's symbols are at y_off + tag, tag = band + 9 K (block within the tile)
template <int R, uint32_t SMB = 0, bool WIDE = true>
__device__ __forceinline__ void fx2_queue_entry(const uint32_t* __restrict__ roots, const uint32_t fma_off, uint32_t* fail, const uint32_t e, const uint32_t q_off, const uint32_t qcap, const uint32_t y_off) {
    const u32x2 sy = *T3_LP(const u32x2, q_off + 8u * e);
    const uint32_t tag = *T3_LP(const uint16_t, q_off + 8u * qcap + 2u * e);
    if (!fx2_fix_block<R, SMB, WIDE>(sy.x, sy.y, y_off + tag, roots, fma_off)) atomicAdd(fail, 1u);
}

}  // namespace
}  // namespace t3
