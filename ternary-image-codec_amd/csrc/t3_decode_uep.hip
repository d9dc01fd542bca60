// t3_decode_uep.hip — one-launch FIXED-mode ("v6c") decoder for the framings the uniform-k kernel (t3_decode_fused.hip) declines:
// per-band k (UEP, OLD:1089-1100; up to two codes per frame) and / or the 2-D boustrophedon interleave (deinterleave2D_boustrophedon
// OLD:781-813), pixels out.  Round 2 ran these through two kernels with the corrected symbols of the whole frame written to and read
// back from a stream-ordered scratch (2 x 144 MB per 8K frame: 1.84 x the algorithmic traffic, 0.29-0.31 ms for BASELINE configs[2]).
// Here a tile's symbols never leave LDS:
//   * producer / consumer waves, matrix-core syndromes, single errors in place, queue + Berlekamp-Massey: the block stages of
//     t3_decode_fx2.h, exactly as in decode_fixed_px_kernel.  Bands are grouped by k; a (wave, pass) pair of sets belongs to one group,
//     so a wave runs one code's routine at a time (wave-uniform switch on r); every group has its own correction queue, so the
//     consumers run full waves of one code.
//   * tile = 9 Lq stream symbols with Lq a common multiple of the k's -- NOT a whole number of pixels.  2-D: after the correction the
//     odd rows' pieces inside the tile are reversed in place (the map is an involution inside a row, so a row piece cut by the tile edge
//     turns into a run of consecutive PRE-interleave symbols at the same addresses); the tile then holds up to three runs of
//     consecutive pre-interleave symbols.  Pixel triples (13 symbols) that lie wholly inside a run are converted and stored from LDS
//     (one lane = four triples, funnel-shifted to the run's byte alignment).  The few symbols of the triples a run's ends cut through
//     (< 13 each side) go to a sparse scratch at their stream positions, and uep_edge_kernel -- one lane per run start -- converts those
//     triples afterwards: a few hundred bytes per tile instead of the whole frame.
// Restrictions (the host falls back to the two-kernel path): at most two codes; rows and chunk area multiples of 4 symbols in 2-D (the
// stream's last, shorter row may be anything); pixels out (no raw-word output, no fused RGB).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"
#include "t3_decode_fx.h"
#include "t3_decode_fx2.h"

namespace t3 {

namespace {
__device__ __forceinline__ void barrier_lds3() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef uint32_t v4u32u __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) V4a2u { v4u32u v; };
struct __attribute__((packed, aligned(2))) V2a2u { uint32_t v[2]; };
__device__ __forceinline__ v4u32u load16u(const uint8_t* p) { return __builtin_nontemporal_load(&((const V4a2u*)p)->v); }
__device__ __forceinline__ uint32_t mod3w(uint32_t x) { return x - 3u * (uint32_t)(((uint64_t)x * 0xAAAAAAABull) >> 33); }
__device__ __forceinline__ uint32_t udiv(uint32_t n, const DevDiv& d) { return __umulhi(n, d.mul) >> d.sh; }       // d >= 2

// LDS header words of this kernel (the block stages' own words: t3_decode.h kFx2*)
constexpr uint32_t kUepCnt = 144;                       // queue counters [buffer][group], 4 words (band rows end at 144)
struct GrpRec { uint32_t r, nb, n_items, af_off, q_rel, q_cap, div_mul, div_sh; uint8_t bands[12]; uint32_t pad_; };   // 48 B, LDS copy of DecUepArgs::Grp
static_assert(sizeof(GrpRec) == 48, "group record");

// Row geometry of post-interleave position v (v < n_sym): start, length, parity of its row (OLD:750-813: chunks of A = w h symbols,
// rows of w, the stream's ragged last chunk / row within their own length)
struct RowG { uint32_t start, len, odd; };
__device__ __forceinline__ RowG row_of(const DecUepArgs& a, uint32_t v) {
    const uint32_t chunk = udiv(v, a.div_A), base = chunk * a.il_A, rem = v - base, take = min(a.il_A, a.n_sym - base);
    const uint32_t r = udiv(rem, a.div_w), rw = r * a.il_w;
    RowG g; g.start = base + rw; g.len = min(a.il_w, take - rw); g.odd = r & 1u;
    return g;
}
// The runs of consecutive PRE-interleave symbols a tile holds once its odd row pieces are reversed in place: run i sits at tile offset
// at[i] (bytes from the tile's first symbol), covers pre-interleave positions [lo[i], lo[i] + len[i]).  1-D: one run.
struct Runs { uint32_t at[3], lo[3], len[3], n; };
__device__ __forceinline__ Runs tile_runs(const DecUepArgs& a, const uint32_t S0) {
    Runs R; R.n = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { R.at[i] = 0; R.lo[i] = 0; R.len[i] = 0; }
    const uint32_t E = min(S0 + a.TS, a.n_sym);                                     // (positions past the stream's end are padding, not pixels)
    if (S0 >= E) return R;
    auto push = [&](uint32_t at, uint32_t lo, uint32_t len) {                      // (no dynamic indexing: the runs stay in registers)
        if (!len) return;
        if (R.n == 0u) { R.at[0] = at; R.lo[0] = lo; R.len[0] = len; R.n = 1u; }
        else if (R.n == 1u) { if (R.lo[0] + R.len[0] == lo && R.at[0] + R.len[0] == at) R.len[0] += len; else { R.at[1] = at; R.lo[1] = lo; R.len[1] = len; R.n = 2u; } }
        else if (R.n == 2u) { if (R.lo[1] + R.len[1] == lo && R.at[1] + R.len[1] == at) R.len[1] += len; else { R.at[2] = at; R.lo[2] = lo; R.len[2] = len; R.n = 3u; } }
        else R.len[2] += len;                                                       // (cannot happen: at most three runs)
    };
    if (!a.il_on) { push(0u, S0, E - S0); return R; }
    const RowG r0 = row_of(a, S0), r1 = row_of(a, E - 1u);
    const uint32_t e0 = min(r0.start + r0.len, E);                                  // end of the first row's piece
    // a piece [p, q) of an odd row [s, s + L) holds, reversed, the pre-interleave positions [2 s + L - q, 2 s + L - p)
    push(0u, r0.odd ? 2u * r0.start + r0.len - e0 : S0, e0 - S0);
    if (e0 < E) {
        const uint32_t c0 = max(r1.start, e0);                                      // start of the last row's piece
        push(e0 - S0, e0, c0 - e0);                                                 // whole rows in between: onto themselves
        push(c0 - S0, r1.odd ? 2u * r1.start + r1.len - E : c0, E - c0);
    }
    return R;
}

// 13 symbols -> 3 pixels (unpack_two_pixels OLD:706-722 on the regrouped stream), scalar: the edge kernel's few triples
__device__ __forceinline__ void px3_from_syms(const uint32_t* s, uint16_t* o) {
    const uint32_t Y0 = s[0] + 27u * (s[1] - 9u * d9(s[1])), B0 = d9(s[1]) + 3u * s[2];
    const uint32_t R0 = s[3] + 27u * (s[4] - 3u * d3(s[4])), Y1 = d3(s[4]) + 9u * s[5];
    const uint32_t B1 = s[6] + 27u * (s[7] - 3u * d3(s[7]));
    const uint32_t R1 = d3(s[7]) + 9u * (s[8] - 9u * d9(s[8]));
    const uint32_t Y2 = d9(s[8]) + 3u * s[9] + 81u * (s[10] - 3u * d3(s[10]));
    const uint32_t B2 = d3(s[10]) + 9u * (s[11] - 9u * d9(s[11])), R2 = d9(s[11]) + 3u * s[12];
    o[0] = (uint16_t)Y0; o[1] = (uint16_t)(B0 - 40u); o[2] = (uint16_t)(R0 - 40u); o[3] = (uint16_t)Y1; o[4] = (uint16_t)(B1 - 40u); o[5] = (uint16_t)(R1 - 40u);
    o[6] = (uint16_t)Y2; o[7] = (uint16_t)(B2 - 40u); o[8] = (uint16_t)(R2 - 40u);
}

// producer work of one (wave, pass): two sets of one group
template <int R, int RM, bool WIDE>
__device__ __forceinline__ void uep_pass(const DecUepArgs& a, const Geo gA, const Geo gB, const uint32_t offA, const uint32_t offB, const bool vA, const bool vB, const bool haveB,
                                         const uint32_t u2, const uint32_t y_off, const uint32_t (&LA)[4], const uint32_t (&LB)[4], const uint32_t lane,
                                         const uint32_t af_off, const uint32_t* __restrict__ roots, const uint32_t cnt_addr, const uint32_t q_off, const uint32_t q_cap, uint32_t* const failp) {
    constexpr uint32_t TCOP = 16, TBASE = kFx2TPx, MT = kFx2ModPx;
    const uint32_t h = lane >> 5;
    const Blk bA = fx2_block(gA, offA, vA, u2, y_off), bB = fx2_block(gB, offB, vB, u2, y_off);
    const Synd sA = fx2_set<R, TCOP, TBASE, MT, 0, RM>(bA, LA, lane, af_off, a.pat_off);
    Synd sB; sB.lo = 0; sB.hi = 0;
    if (haveB) sB = fx2_set<R, TCOP, TBASE, MT, 0, RM>(bB, LB, lane, af_off, a.pat_off);
    fx2_own_blocks<R, 0, WIDE>(roots, a.fma_off, failp, sA, sB, bA, bB, (h ? gB : gA) & 0xFFFFu, lane, cnt_addr, q_off, q_cap);
}
}  // namespace

// RA / RB: r = 26 - k of group 0 / group 1 (the host orders the groups RA >= RB; RA == RB: one group)
template <int RA, int RB>
__global__ __launch_bounds__(512, 6) void decode_uep_px_kernel(const DecUepArgs a) {
    constexpr uint32_t NW = 4;                                                       // producer waves = consumer waves
    constexpr bool WIDE = !(RA == 6 && RB == 4);                                     // Forney chains side by side (t3_decode_fx2.h), except where it costs a spill
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nthr = blockDim.x;
    // ---- tickets (as decode_fixed_px_kernel) ----
    const uint32_t grid = gridDim.x;
    const bool dyn = a.tile_ctr != nullptr;
    const uint32_t NC = dyn ? a.n_classes : 1u, cls = blockIdx.x % NC;
    const uint32_t wgc = (grid - cls + NC - 1u) / NC;
    uint32_t* const ctr = a.tile_ctr + 64u * cls;
    if (tid == 0) *(uint32_t*)(lds + kFx2Next + 4u) = dyn ? cls + NC * (wgc + atomicAdd(ctr, 1u)) : blockIdx.x + grid;
    // header check and verdict words in this launch (streaming entry; as decode_fixed_px_kernel): failures counted in LDS, one global add per workgroup
    uint32_t* const failp = a.verdict ? (uint32_t*)(lds + kFx2FailWg) : a.fail;
    if (tid == 0) *(uint32_t*)(lds + kFx2FailWg) = 0u;
    if (a.verdict && blockIdx.x == 0u && wave == 2u * NW - 1u) {
        uint32_t want = 0;
#pragma unroll
        for (uint32_t q = 0; q < 24; ++q) want = lane == q ? a.hx[q] : want;
        bool mis = false;
        if (4u * lane < a.hdr_n) {
            const uint32_t nb = min(4u, a.hdr_n - 4u * lane), mask = nb >= 4u ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
            mis = ((((const uint32_t*)a.hdr_in)[lane] ^ want) & mask) != 0u;
        }
        const bool any = __builtin_amdgcn_ballot_w64(mis) != 0;
        if (lane == 0) a.verdict[0] = any ? 1u : 0u;
    }
    // ---- constants -> LDS ----
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) { Row r; r.blocks = a.band_blocks[b]; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b]; *(Row*)(lds + 16 * b) = r; }
#pragma unroll
        for (int i = 0; i < 4; ++i) *(uint32_t*)(lds + kUepCnt + 4 * i) = 0;
        *(uint32_t*)(lds + kFx2Sync) = 0; *(uint32_t*)(lds + kFx2Abort) = 0;
#pragma unroll
        for (int g = 0; g < kUepMaxGrp; ++g) {
            GrpRec r; r.r = a.grp[g].r; r.nb = a.grp[g].nb; r.n_items = a.grp[g].n_items; r.af_off = a.grp[g].af_off; r.q_rel = a.grp[g].q_rel; r.q_cap = a.grp[g].q_cap;
            r.div_mul = a.grp[g].div_nb.mul; r.div_sh = a.grp[g].div_nb.sh; r.pad_ = 0;
#pragma unroll
            for (int q = 0; q < 12; ++q) r.bands[q] = a.grp[g].bands[q];
            *(GrpRec*)(lds + a.rec_off + 48 * g) = r;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *(uint32_t*)(lds + a.rec_off + 96 + 4 * i) = a.pair_tab[i];
    }
    if (tid == 64u) {
#pragma unroll
        for (int i = 0; i < 48; ++i) *(uint32_t*)(lds + a.pat_off + 4 * i) = a.pat[i];
    }
    for (uint32_t i = tid * 16u; i < (uint32_t)kFx2SmallBytes; i += nthr * 16u) *(uint4*)(lds + kFx2Small + i) = *(const uint4*)(a.small + i);
    for (uint32_t i = tid * 16u; i < (uint32_t)kFx2ModBytes; i += nthr * 16u) *(uint4*)(lds + kFx2ModPx + i) = *(const uint4*)(a.small + kFx2SmallBytes + i);
    for (uint32_t i = tid * 16u; i < 3u * 27u * 4u * 16u; i += nthr * 16u) *(uint4*)(lds + kFx2TPx + i) = *(const uint4*)((const uint8_t*)a.ttab + i);
    for (uint32_t i = tid * 16u; i < 19696u; i += nthr * 16u) *(uint4*)(lds + a.fma_off + i) = *(const uint4*)(a.fma + i);
    for (uint32_t i = tid * 16u; i < 3072u; i += nthr * 16u) *(uint4*)(lds + a.grp[0].af_off + i) = *(const uint4*)((const uint8_t*)a.grp[0].afrag + i);   // group 0's A operand serves both codes
    if (tid < 64u) *(uint32_t*)(lds + a.grp[0].af_off + 3072u + 4u * tid) = a.grp[0].afrag[(3u * 64u + tid) * 4u];
    __syncthreads();
    const uint8_t* body = a.in + a.hdr_syms;
    uint32_t cur = blockIdx.x, nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u));

    if (wave < NW) {
        // ---------------- producers ----------------
        const uint32_t h = lane >> 5;
        // this wave's two pairs of sets (pass 0, pass 1): group and geometry of its lanes' blocks.  The pass loop below is NOT unrolled
        // (two codes x two passes of inlined block stages would be four copies of them: registers and instruction cache), so the per-pass
        // values are picked by selects.
        uint32_t pg0, pg1, pi0, pi1;                                                 // group (255: no pair), first item
        // three registers per pass (80-VGPR budget): X = byte offset of the block in tile 0 (< 2^28: host) | cbase << 28 | has-an-item << 31
        // for set A and for set B, and the two sets' y_rel as 16-bit halves of one word; Geo (t3_decode_fx2.h) is rebuilt from them at use
        uint32_t xa0, xb0, yy0, xa1, xb1, yy1;
        auto init = [&](const uint32_t p, uint32_t& pgp, uint32_t& pip, uint32_t& xa, uint32_t& xb, uint32_t& yy) {
            const uint32_t pt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + a.rec_off + 96u + 4u * (2u * wave + p)));
            pgp = pt & 0xFFu; pip = pt >> 8;                                         // pt == 0xFFFFFFFF: no pair (group 255)
            const bool idle = pgp >= a.n_grp;
            const uint32_t rb = a.rec_off + 48u * (idle ? 0u : pgp);
            const GrpRec G = *(const GrpRec*)(lds + rb);
            const uint32_t K = 26u - G.r;
            yy = 0;
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q) {
                const uint32_t item = (idle ? 0u : pip) + 32u * q + (lane & 31u);
                const uint32_t big = min(__umulhi(item, G.div_mul) >> G.div_sh, 11u), m = item - big * G.nb;     // band index within the group, block within the tile
                const uint32_t b = lds[rb + 32u + big];
                const uint32_t m3 = m - 3u * ((m * 683u) >> 11);
                const Row rw = row(min(b, 8u));
                uint32_t cb = rw.boff6 + 2u * m3; cb -= cb >= 6u ? 6u : 0u;
                const uint32_t x = ((uint32_t)rw.body_off + 26u * m) | cb << 28 | ((!idle && item < G.n_items) ? 1u << 31 : 0u);
                yy |= (b + 9u * K * m) << (16u * q);
                if (q == 0) xa = x; else xb = x;
            }
        };
        init(0, pg0, pi0, xa0, xb0, yy0); init(1, pg1, pi1, xa1, xb1, yy1);
        auto geo_of = [](const uint32_t x, const uint32_t yy, const uint32_t q) -> Geo { return ((yy >> (16u * q)) & 0xFFFFu) | (x >> 28) << 16; };   // y_rel | cbase << 16 | has << 19
        // a band's blocks end inside the last tiles only; there the lane's block is tested against the band's byte range
        auto has = [&](const Geo g, const uint32_t off, const uint32_t tile) -> bool {
            bool v = (g >> 19) != 0u;
            if (tile + 2u >= a.n_tiles) { const uint32_t yr = g & 0xFFFFu, b = yr - 9u * (yr / 9u); const Row rw = row(b); v = v && off < (uint32_t)rw.body_off + 26u * rw.blocks; }
            return v;
        };
        auto load_of = [&](const Geo g, const uint32_t off, const uint32_t tile) -> v4u32u { return load16u(body + (has(g, off, tile) ? off + 10u * h : 0u)); };
        auto nbp = [&](uint32_t pgp) -> uint32_t { return __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + a.rec_off + 48u * min(pgp, a.n_grp - 1u) + 4u)); };
        const uint32_t nb0 = nbp(pg0), nb1 = nbp(pg1);
        v4u32u PA = {0, 0, 0, 0}, PB = {0, 0, 0, 0};
        if (cur < a.n_tiles) { PA = load_of(geo_of(xa0, yy0, 0), (xa0 & 0x0FFFFFFFu) + cur * 26u * nb0, cur); PB = load_of(geo_of(xb0, yy0, 1), (xb0 & 0x0FFFFFFFu) + cur * 26u * nb0, cur); }
        for (uint32_t k = 0; cur < a.n_tiles; ++k) {
            const uint32_t tile = cur, buf = k & 1u;
            const uint32_t y_off = a.y_off + buf * a.y_stride;
            uint32_t raw; asm volatile("" : "=v"(raw));
#pragma unroll 1
            for (uint32_t pass = 0; pass < 2; ++pass) {
                uint32_t LA[4] = {PA[0], PA[1], PA[2], PA[3]}, LB[4] = {PB[0], PB[1], PB[2], PB[3]};
                asm volatile("" : "+v"(LA[0]), "+v"(LA[1]), "+v"(LA[2]), "+v"(LA[3]), "+v"(LB[0]), "+v"(LB[1]), "+v"(LB[2]), "+v"(LB[3]));   // (taken into registers before the draw: see decode_fixed_px_kernel)
                if (pass == 1u && tid == 0u && dyn) raw = atomicAdd(ctr, 1u);
                {   // the next pass's input
                    const uint32_t nt = pass == 0 ? tile : nxt, nbn = pass ? nb0 : nb1;
                    uint32_t xa = pass ? xa0 : xa1, xb = pass ? xb0 : xb1, yy = pass ? yy0 : yy1; asm volatile("" : "+v"(xa), "+v"(xb), "+v"(yy));
                    if (nt < a.n_tiles) { PA = load_of(geo_of(xa, yy, 0), (xa & 0x0FFFFFFFu) + nt * 26u * nbn, nt); PB = load_of(geo_of(xb, yy, 1), (xb & 0x0FFFFFFFu) + nt * 26u * nbn, nt); }
                }
                const uint32_t pgp = pass ? pg1 : pg0, pip = pass ? pi1 : pi0;
                if (pgp < a.n_grp) {                                                 // (wave-uniform) this wave has a pair in this pass
                    const uint32_t rb = a.rec_off + 48u * pgp, nbg = pass ? nb1 : nb0;
                    const uint32_t n_items = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb + 8u)), af_off = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb + 12u));
                    const uint32_t q_rel = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb + 16u)), q_cap = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb + 20u));
                    const uint32_t toff = tile * 26u * nbg, u2 = 2u * mod3w(tile * nbg);
                    uint32_t xa = pass ? xa1 : xa0, xb = pass ? xb1 : xb0, yy = pass ? yy1 : yy0; asm volatile("" : "+v"(xa), "+v"(xb), "+v"(yy));
                    const Geo gA = geo_of(xa, yy, 0), gB = geo_of(xb, yy, 1);
                    const uint32_t oA = (xa & 0x0FFFFFFFu) + toff, oB = (xb & 0x0FFFFFFFu) + toff;
                    const bool vA = has(gA, oA, tile), vB = has(gB, oB, tile), haveB = pip + 32u < n_items;
                    const uint32_t cnt_addr = kUepCnt + 4u * (2u * buf + pgp), q_off = a.q_off + buf * a.q_stride + q_rel;
                    // (one A operand, group 0's: the code with more parity -- its syndromes include the other code's)
                    if (pgp == 0u) uep_pass<RA, RA, WIDE>(a, gA, gB, oA, oB, vA, vB, haveB, u2, y_off, LA, LB, lane, af_off, a.grp[0].roots, cnt_addr, q_off, q_cap, failp);
                    else uep_pass<RB, RA, WIDE>(a, gA, gB, oA, oB, vA, vB, haveB, u2, y_off, LA, LB, lane, af_off, a.grp[1].roots, cnt_addr, q_off, q_cap, failp);
                }
            }
            if (tid == 0u) *(uint32_t*)(lds + kFx2Next + 4u * buf) = dyn ? cls + NC * (wgc + raw) : nxt + grid;
            barrier_lds3();
            cur = nxt; nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u * buf));
        }
        barrier_lds3();                                                             // the consumers' last interval
    } else {
        // ---------------- consumers: correction queues, [un-interleave], symbols -> pixels ----------------
        const uint32_t cw = wave - NW, ct = cw * 64u + lane;                        // consumer thread 0..255
        uint32_t prev = 0, rdv = 0;                                                  // rendezvous reached so far (x NW arrivals)
        uint32_t* const sync = (uint32_t*)__builtin_assume_aligned(lds + kFx2Sync, 4);
        auto rendezvous = [&]() {                                                    // every consumer wave's LDS writes are done before any wave goes on
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ++rdv;
            if (lane == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            uint32_t spins = 0;
            while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NW * rdv) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) { if (lane == 0) { *(uint32_t*)(lds + kFx2Abort) = 1u; atomicAdd(failp, 1u << 20); } break; }   // a bound, not a path
            }
        };
        for (uint32_t k = 0;; ++k) {
            if (k >= 1u) {
                const uint32_t tile = prev, buf = (k - 1u) & 1u;
                const uint32_t y_off = a.y_off + buf * a.y_stride, S0 = tile * a.TS;
                {   // both groups' queues at once: wave slots 0 .. nA - 1 take 64 entries of group 0 each, the following slots group 1 (the
                    // correction is one long dependent chain: two half-empty passes one after the other would cost two chain lengths)
                    const uint32_t rb1 = a.rec_off + 48u;
                    const uint32_t capA = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + a.rec_off + 20u)), capB = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb1 + 20u));
                    const uint32_t qA = a.q_off + buf * a.q_stride + __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + a.rec_off + 16u));
                    const uint32_t qB = a.q_off + buf * a.q_stride + __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + rb1 + 16u));
                    const uint32_t QA = min(__builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kUepCnt + 4u * (2u * buf))), capA);
                    const uint32_t QB = a.n_grp > 1u ? min(__builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kUepCnt + 4u * (2u * buf + 1u))), capB) : 0u;
                    const uint32_t nA = (QA + 63u) / 64u, nB = (QB + 63u) / 64u;
                    for (uint32_t sl = cw; sl < nA + nB; sl += NW) {
                        if (sl < nA) { const uint32_t e = 64u * sl + lane; if (e < QA) fx2_queue_entry<RA, 0, WIDE>(a.grp[0].roots, a.fma_off, failp, e, qA, capA, y_off); }
                        else { const uint32_t e = 64u * (sl - nA) + lane; if (e < QB) fx2_queue_entry<RB, 0, WIDE>(a.grp[1].roots, a.fma_off, failp, e, qB, capB, y_off); }
                    }
                }
                rendezvous();                                                       // every patch is in LDS
                if (ct < a.n_grp) *(uint32_t*)(lds + kUepCnt + 4u * (2u * buf + ct)) = 0;   // every consumer has read its Q; the producers touch these counters after the barrier
                if (a.il_on) {
                    // odd rows' pieces, reversed in place (dword granules: rows, chunks and the tile are multiples of 4 symbols)
                    const uint32_t E = min(S0 + a.TS, a.n_sym);
                    for (uint32_t d = ct; 4u * d < a.TS; d += 64u * NW) {
                        const uint32_t v = S0 + 4u * d;
                        if (v >= E) continue;
                        const RowG rg = row_of(a, v);
                        if (!rg.odd) continue;
                        const uint32_t lo = max(rg.start, S0), hi = min(rg.start + rg.len, E);
                        if ((hi - lo) & 3u) {                                            // the stream's last row, when it is not whole dwords: one lane, byte by byte
                            if (v == lo) for (uint32_t i = lo, j = hi - 1u; i < j; ++i, --j) { const uint32_t x = l8(y_off + (i - S0)); *T3_LP(uint8_t, y_off + (i - S0)) = (uint8_t)l8(y_off + (j - S0)); *T3_LP(uint8_t, y_off + (j - S0)) = (uint8_t)x; }
                            continue;
                        }
                        const uint32_t v2 = lo + hi - 4u - v;                            // the mirrored granule inside the piece
                        if (v2 < v) continue;                                            // (its partner does the swap)
                        const uint32_t x = *T3_LP(const uint32_t, y_off + (v - S0)), y = *T3_LP(const uint32_t, y_off + (v2 - S0));
                        *T3_LP(uint32_t, y_off + (v - S0)) = __builtin_bswap32(y);
                        if (v2 != v) *T3_LP(uint32_t, y_off + (v2 - S0)) = __builtin_bswap32(x);
                    }
                    rendezvous();
                }
                // whole pixel triples of the tile's runs: one lane = four triples (52 symbols at the run's byte alignment -> 72 bytes of pixels)
                const Runs R = tile_runs(a, S0);
                uint32_t j0[3], nu[3], j1[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) { j0[i] = (R.lo[i] + 12u) / 13u; j1[i] = (R.lo[i] + R.len[i]) / 13u; nu[i] = j1[i] > j0[i] ? (j1[i] - j0[i] + 3u) / 4u : 0u; }
                const uint32_t n01 = nu[0] + nu[1], n_all = n01 + nu[2];
                for (uint32_t e = ct; e < n_all; e += 64u * NW) {
                    const uint32_t ri = e < nu[0] ? 0u : e < n01 ? 1u : 2u;
                    const uint32_t eb = ri == 0u ? 0u : ri == 1u ? nu[0] : n01;
                    const uint32_t rj0 = ri == 0u ? j0[0] : ri == 1u ? j0[1] : j0[2], rj1 = ri == 0u ? j1[0] : ri == 1u ? j1[1] : j1[2];
                    const uint32_t rat = ri == 0u ? R.at[0] : ri == 1u ? R.at[1] : R.at[2], rlo = ri == 0u ? R.lo[0] : ri == 1u ? R.lo[1] : R.lo[2];
                    const uint32_t jt = rj0 + 4u * (e - eb), nt = min(4u, rj1 - jt);
                    const uint32_t ad = y_off + rat + (13u * jt - rlo), sh = ad & 3u, ab = ad & ~3u;
                    uint32_t Dw[14], D[13];
#pragma unroll
                    for (int i = 0; i < 14; ++i) Dw[i] = *T3_LP(const uint32_t, ab + 4u * i);
#pragma unroll
                    for (int i = 0; i < 13; ++i) D[i] = __builtin_amdgcn_alignbyte(Dw[i + 1], Dw[i], sh);
                    uint32_t o[18];
                    px12_from_syms(D, o);
                    const uint64_t px0 = 3ull * jt;
                    const uint32_t n_ok = (uint32_t)min((uint64_t)(3u * nt), a.n_units > px0 ? a.n_units - px0 : 0ull);
                    uint8_t* gp = (uint8_t*)a.out + px0 * 6u;                            // 2-byte aligned (18 bytes per triple)
                    if (n_ok == 12u) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { V4a2u w; w.v = v4u32u{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]}; *(V4a2u*)(gp + 16 * q) = w; }
                        V2a2u w2; w2.v[0] = o[16]; w2.v[1] = o[17]; *(V2a2u*)(gp + 64) = w2;
                    } else {
#pragma unroll
                        for (uint32_t hh = 0; hh < 36; ++hh) if (hh / 3u < n_ok) *(uint16_t*)(gp + 2u * hh) = (uint16_t)(o[hh >> 1] >> (16u * (hh & 1u)));
                    }
                }
                // the symbols of the triples the runs' ends cut through -> sparse scratch (uep_edge_kernel converts those triples)
                if (ct < 72u) {
                    const uint32_t ri = ct / 24u, q = ct - 24u * ri;                   // run, byte slot: 0..11 head, 12..23 tail
                    const uint32_t rat = ri == 0u ? R.at[0] : ri == 1u ? R.at[1] : R.at[2], rlo = ri == 0u ? R.lo[0] : ri == 1u ? R.lo[1] : R.lo[2], rlen = ri == 0u ? R.len[0] : ri == 1u ? R.len[1] : R.len[2];
                    const uint32_t rj0 = ri == 0u ? j0[0] : ri == 1u ? j0[1] : j0[2], rj1 = ri == 0u ? j1[0] : ri == 1u ? j1[1] : j1[2];
                    const uint32_t hi = rlo + rlen;
                    // head: [lo, min(hi, 13 j0)); tail: [max(lo, 13 j1), hi), the part of it the head does not already cover
                    const uint32_t head_end = min(hi, 13u * rj0), tail_lo = max(max(rlo, 13u * rj1), head_end);
                    const uint32_t p = q < 12u ? rlo + q : tail_lo + (q - 12u);
                    const bool in = ri < R.n && (q < 12u ? p < head_end : p < hi);
                    if (in) a.edge[p] = (uint8_t)l8(y_off + rat + (p - rlo));
                }
            }
            barrier_lds3();
            if (cur >= a.n_tiles) break;
            prev = cur; cur = nxt; nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u * (k & 1u)));
        }
    }
    if (dyn && tid == 0u) {
        if (a.verdict) { const uint32_t wgf = *(const uint32_t*)(lds + kFx2FailWg); if (wgf) atomicAdd(a.fail, wgf); }   // (every count precedes the loops' closing barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(a.tile_ctr + 64u * NC, 1u) == grid - 1u) {
            if (a.verdict) a.verdict[1] = atomicExch(a.fail, 0u);
            for (uint32_t c = 0; c <= NC; ++c) __hip_atomic_store(a.tile_ctr + 64u * c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// One lane per (tile, run): the pixel triple the run's first symbol cuts through, from the sparse scratch the tiles filled; plus the
// stream's last, incomplete triple (zero symbols past the end: the encoder's zero padding, OLD:1077-1081).
__global__ __launch_bounds__(256) void uep_edge_kernel(const DecUepArgs a) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t j;
    if (id < 3u * a.n_tiles) {
        const uint32_t tile = id / 3u, ri = id - 3u * tile;
        const Runs R = tile_runs(a, tile * a.TS);
        if (ri >= R.n) return;
        const uint32_t lo = ri == 0u ? R.lo[0] : ri == 1u ? R.lo[1] : R.lo[2];
        if (lo % 13u == 0u) return;
        j = lo / 13u;
    } else if (id == 3u * a.n_tiles) {
        if (a.n_sym % 13u == 0u) return;
        j = a.n_sym / 13u;
    } else return;
    uint32_t s[13];
#pragma unroll
    for (uint32_t q = 0; q < 13; ++q) { const uint32_t p = 13u * j + q; s[q] = p < a.n_sym ? a.edge[p] : 0u; }
    uint16_t o[9]; px3_from_syms(s, o);
    uint16_t* gp = (uint16_t*)a.out + 9ull * j;
#pragma unroll
    for (uint32_t p = 0; p < 3; ++p) if (3ull * j + p < a.n_units) { gp[3 * p] = o[3 * p]; gp[3 * p + 1] = o[3 * p + 1]; gp[3 * p + 2] = o[3 * p + 2]; }
}

#define T3_INST_UEP(A, B) template __global__ void decode_uep_px_kernel<A, B>(const DecUepArgs);
T3_INST_UEP(2, 2) T3_INST_UEP(4, 4) T3_INST_UEP(6, 6) T3_INST_UEP(8, 8) T3_INST_UEP(4, 2) T3_INST_UEP(6, 2) T3_INST_UEP(6, 4) T3_INST_UEP(8, 2) T3_INST_UEP(8, 4) T3_INST_UEP(8, 6)

}  // namespace t3
