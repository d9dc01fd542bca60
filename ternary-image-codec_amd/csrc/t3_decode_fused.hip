// t3_decode_fused.hip — fused FIXED-mode ("v6c") decoder for gfx950: K3 syndromes + K4 Berlekamp–Massey / Chien / Forney +
// K5 symbols -> pixels|words in ONE launch, tiled like the encoder (BASELINE config 5: decode with injected trit errors).
// Replaces, for mode=FIXED / uniform k / 1-D / no beacon: descramble (OLD:938-947), per-band RS decode (OLD:963-991,
// decode_block OLD:546-662 with the Forney sign fixed), the i%9 re-merge the reference omits, symbols -> 26-trit words
// (OLD:1022-1040) and unpack_two_pixels (OLD:706-722).  Everything else goes through t3_decode_stream.hip / t3_decode.hip.
//
// Stages of a tile (9 bands x nb blocks; block stages in t3_decode_fx2.h):
//   S   a set = 32 blocks, two lanes per block: one 16-byte load per lane, one T-table read per coded symbol (descramble +
//       trit expansion), syndromes on the matrix cores (78 x 3r GF(3) product as four v_mfma_i32_32x32x32_i8), mod-3 fold by
//       byte tables, data symbols -> stream order in LDS (byte 9(mk+p)+b)
//   E1  one lane = one block: non-zero syndromes that form a geometric progression are a single error, fixed in place; the
//       other flagged blocks are appended to an LDS queue, one wave-aggregated counter update per wave (ballot + prefix count)
//   BM  full waves drain the queue: Berlekamp-Massey on a fused multiply-add table (a + x y, 27^3 bytes in LDS), Chien search
//       by table (root mask per locator, global memory), Forney, <= t bytes patched
//   D5  pixels: one lane = four triples, 13 aligned dwords of symbols -> 12 pixels with packed 16-bit ops -> 72 bytes stored
//       straight to memory;  raw words: 26 symbols -> 3 words, staged in LDS, copied out with 16-byte coalesced stores.
//
// decode_fixed_px_kernel (pixels): every stage is a chain of dependent LDS reads (the correction alone is ~35 levels), so a
// workgroup that runs them one after the other spends most of a tile waiting.  Waves 0-3 (producers) therefore run S + E1 of tile
// k while waves 4-7 (consumers) run BM + D5 of tile k-1 on the other symbol buffer / queue: one workgroup barrier per tile, a
// four-wave rendezvous (LDS counter) between BM and D5.  The producers issue no stores and the consumers no streaming loads,
// so the input loads (issued one pass ahead) never wait behind store acknowledgements.
// decode_fixed_kernel (raw words): the same stages one after the other.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"
#include "t3_decode_fx.h"
#include "t3_decode_fx2.h"

namespace t3 {

namespace {
__device__ __forceinline__ void barrier_lds2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) V4a2 { v4u32 v; };
// the coded stream is read once: non-temporal loads (measured: clean stream 0.140 -> 0.131 ms, with errors 0.157 -> 0.154; non-temporal
// *stores* of the pixels cost 40 %: 0.140 -> 0.193)
__device__ __forceinline__ v4u32 load16(const uint8_t* p) { return __builtin_nontemporal_load(&((const V4a2*)p)->v); }

// A lane's 16 coded bytes, in flight.  BCN (beacon stripped in the loads, OLD:952-957): the run starts at framed offset
// g0 + (beacons in front of it); if the next beacon falls inside the run (after c < 16 body bytes) the run is 17 framed bytes long and
// x carries the 17th byte and c; run_bytes() closes the gap when the run is used.
template <bool BCN> struct Run { v4u32 w; };
template <> struct Run<true> { v4u32 w; uint32_t w4, x; };              // five aligned dwords that hold the (up to) 17 framed bytes; x = start byte | c << 8
template <bool BCN>
__device__ __forceinline__ Run<BCN> load_run(const DecFx2Args& a, const uint8_t* body, const uint32_t g0) {
    Run<BCN> r;
    if constexpr (!BCN) r.w = load16(body + g0);          // 2-byte aligned: as fast as aligned dwords (measured); odd addresses are not, hence:
    else {
        uint32_t nb0 = 0, c = a.bcn_slot - g0;
        if (g0 >= a.bcn_slot) { const uint32_t u = g0 - a.bcn_slot, j = __umulhi(u, a.bcn_div.mul) >> a.bcn_div.sh; nb0 = j + 1u; c = a.bcn_pb - (u - j * a.bcn_pb); }
        const uintptr_t p = (uintptr_t)(body + (g0 + nb0));
        const uint32_t* q = (const uint32_t*)(p & ~(uintptr_t)3);                    // aligned dwords (the stream starts 16-byte aligned: t3hip.h)
        r.w = __builtin_nontemporal_load((const v4u32*)q); r.w4 = 0;
        if (((uint32_t)p & 3u) != 0u || c < 16u) r.w4 = __builtin_nontemporal_load(q + 4);                          // (never a dword that lies wholly behind the run's last byte)
        r.x = ((uint32_t)p & 3u) | min(c, 16u) << 8;
    }
    return r;
}
template <bool BCN>
__device__ __forceinline__ void run_bytes(const Run<BCN>& r, uint32_t (&L)[4]) {
    if constexpr (!BCN) { L[0] = r.w[0]; L[1] = r.w[1]; L[2] = r.w[2]; L[3] = r.w[3]; }
    else {
        const uint32_t sh = r.x & 3u, c = r.x >> 8, dc = c >> 2, bc = c & 3u;        // dc == 4: no beacon in the run
        uint32_t F[5];
        F[0] = __builtin_amdgcn_alignbyte(r.w[1], r.w[0], sh); F[1] = __builtin_amdgcn_alignbyte(r.w[2], r.w[1], sh);
        F[2] = __builtin_amdgcn_alignbyte(r.w[3], r.w[2], sh); F[3] = __builtin_amdgcn_alignbyte(r.w4, r.w[3], sh);
        F[4] = r.w4 >> (8u * sh);                                                     // its low byte: the 17th framed byte
        const uint32_t D = dc == 0u ? F[0] : dc == 1u ? F[1] : dc == 2u ? F[2] : F[3], Dn = dc == 0u ? F[1] : dc == 1u ? F[2] : dc == 2u ? F[3] : F[4];
        const uint32_t sel = bc == 0u ? 0x04030201u : bc == 1u ? 0x04030200u : bc == 2u ? 0x04030100u : 0x04020100u;   // v_perm(S0, S1): 0..3 = S1, 4..7 = S0
        const uint32_t Mx = __builtin_amdgcn_perm(Dn, D, sel);                       // the dword the beacon sits in, without it
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) L[i] = i < dc ? F[i] : i == dc ? Mx : __builtin_amdgcn_alignbyte(F[i + 1], F[i], 1u);
    }
}

__device__ __forceinline__ uint32_t mod3u(uint32_t x) { return x - 3u * (uint32_t)(((uint64_t)x * 0xAAAAAAABull) >> 33); }

// constants -> LDS (both kernels): band rows, counters, byte tables, fold tables, T, multiply-accumulate table, A operand
template <uint32_t TCOP, uint32_t TBASE, uint32_t MT>
__device__ __forceinline__ void stage_tables(const DecFx2Args& a, const uint32_t tid, const uint32_t nthr) {
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) { Row r; r.blocks = a.band_blocks[b]; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b]; *(Row*)(lds + 16 * b) = r; }
        *(uint32_t*)(lds + kFx2Cnt) = 0; *(uint32_t*)(lds + kFx2Cnt + 4) = 0; *(uint32_t*)(lds + kFx2Sync) = 0; *(uint32_t*)(lds + kFx2Abort) = 0;
    }
    for (uint32_t i = tid * 16u; i < (uint32_t)kFx2SmallBytes; i += nthr * 16u) *(uint4*)(lds + kFx2Small + i) = *(const uint4*)(a.small + i);
    for (uint32_t i = tid * 16u; i < (uint32_t)kFx2ModBytes; i += nthr * 16u) *(uint4*)(lds + MT + i) = *(const uint4*)(a.small + kFx2SmallBytes + i);
    for (uint32_t i = tid * 16u; i < 3u * 27u * 4u * TCOP; i += nthr * 16u) *(uint4*)(lds + TBASE + i) = *(const uint4*)((const uint8_t*)a.ttab + i);
    for (uint32_t i = tid * 16u; i < 19696u; i += nthr * 16u) *(uint4*)(lds + a.fma_off + i) = *(const uint4*)(a.fma + i);
    for (uint32_t i = tid * 16u; i < 3072u; i += nthr * 16u) *(uint4*)(lds + a.af_off + i) = *(const uint4*)((const uint8_t*)a.afrag + i);
    if (tid < 64u) *(uint32_t*)(lds + a.af_off + 3072u + 4u * tid) = a.afrag[(3u * 64u + tid) * 4u];       // step 3: dword 0 of every lane
    if (tid == 64u) {                                                                // scrambler pattern rows (t3_decode_fx2.h, fx2_set): constant indices only
#pragma unroll
        for (int i = 0; i < 48; ++i) *(uint32_t*)(lds + a.pat_off + 4 * i) = a.pat[i];
    }
}

// D5 (pixels) for lane slot j of a tile: four triples = 52 symbols at y_off + 52 j -> 12 pixels = 72 bytes; RGB: the inverse
// io_image.hpp bridge fused in (dequantize_ycbcr :79-84 by table, ycbcr_to_rgb :57-66 with every float step rounded on its own,
// std::lround + clamp to 0..255 = min(trunc(x + 0.5) from zero up, 255)) -> 36 bytes
template <bool RGB>
__device__ __forceinline__ void fx2_pixels12(const DecFx2Args& a, const uint32_t j, const uint32_t y_off, const uint64_t unit0, const uint32_t n_here) {
    uint32_t D[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) D[i] = *T3_LP(const uint32_t, y_off + 52u * j + 4u * i);
    uint32_t o[18];
    px12_from_syms(D, o);
    if constexpr (RGB) {
        uint32_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t p = 0; p < 12; ++p) {
            auto comp = [&](uint32_t k) -> uint32_t { return (o[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu; };
            const uint32_t Yq = min(comp(3u * p), 242u);
            const int cbq = max(-40, min(40, (int)(int16_t)comp(3u * p + 1u))), crq = max(-40, min(40, (int)(int16_t)comp(3u * p + 2u)));
            const float y = (float)l8(a.dq_off + Yq);
            const float cb = __fsub_rn((float)l8(a.dq_off + 244u + (uint32_t)(cbq + 40)), 128.0f), cr = __fsub_rn((float)l8(a.dq_off + 244u + (uint32_t)(crq + 40)), 128.0f);
            const float r = __fadd_rn(y, __fmul_rn(1.402f, cr));
            const float g = __fsub_rn(__fsub_rn(y, __fmul_rn(0.344136f, cb)), __fmul_rn(0.714136f, cr));
            const float b = __fadd_rn(y, __fmul_rn(1.772f, cb));
            const uint32_t c3[3] = {min((uint32_t)__fadd_rn(r, 0.5f), 255u), min((uint32_t)__fadd_rn(g, 0.5f), 255u), min((uint32_t)__fadd_rn(b, 0.5f), 255u)};
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) { const uint32_t bi = 3u * p + k; w[bi >> 2] |= c3[k] << (8u * (bi & 3u)); }
        }
        uint8_t* g8 = (uint8_t*)a.out + (unit0 + 12ull * j) * 3u;                    // 4-byte aligned
        if (12u * j + 12u <= n_here) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(4)));
            *(v4u*)(g8) = v4u{w[0], w[1], w[2], w[3]}; *(v4u*)(g8 + 16) = v4u{w[4], w[5], w[6], w[7]}; *(uint32_t*)(g8 + 32) = w[8];
        } else {
#pragma unroll
            for (uint32_t bi = 0; bi < 36; ++bi) if (12u * j + bi / 3u < n_here) g8[bi] = (uint8_t)(w[bi >> 2] >> (8u * (bi & 3u)));
        }
    } else {
    uint8_t* g = (uint8_t*)a.out + (unit0 + 12ull * j) * 6u;                        // 8-byte aligned
    if (12u * j + 12u <= n_here) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(8)));
        typedef uint32_t v2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
        for (int d = 0; d < 4; ++d) *(v4u*)(g + 16 * d) = v4u{o[4 * d], o[4 * d + 1], o[4 * d + 2], o[4 * d + 3]};
        *(v2u*)(g + 64) = v2u{o[16], o[17]};
    } else {                                                                        // the frame's last pixels: per 16-bit component
#pragma unroll
        for (uint32_t hh = 0; hh < 36; ++hh)
            if (12u * j + hh / 3u < n_here) *(uint16_t*)(g + 2u * hh) = (uint16_t)(o[hh >> 1] >> (16u * (hh & 1u)));
    }
    }
}
}  // namespace

#ifndef T3_DEC_WAVES_PER_EU
#define T3_DEC_WAVES_PER_EU 6   // <= 80 VGPRs: three 8-wave workgroups per CU
#endif
#ifdef T3_DEC_STAMPS   // diagnostic build: per-phase cycle sums of waves 0 and 4 (never in the product build)
#define T3D_STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define T3D_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------------
// pixels out: producer / consumer waves
// ------------------------------------------------------------------------------------------------------------------
template <int R, bool RGB, bool BCN>
__global__ __launch_bounds__(T3_DEC_PX_THREADS, T3_DEC_WAVES_PER_EU) void decode_fixed_px_kernel(const DecFx2Args a) {
    constexpr uint32_t TCOP = T3_DEC_PX_TCOP, TBASE = kFx2TPx, MT = kFx2ModPx, QCAP = kFx2QCap;
    constexpr uint32_t NW = T3_DEC_PX_THREADS / 128;                                // producer waves = consumer waves
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tiles are handed out by tickets, as in the encoder (t3_kernels.hip): the workgroups of a CU progress at different speeds (a static
    // stride left the slowest workgroup 20 % behind the mean: stamp build, profiles/r03/notes.md).  Workgroup w starts with tile w; every
    // further tile is drawn from a counter -- one per class (index mod n_classes: a memory-side atomic serves ~11 ns per draw, too slow
    // for one counter and 15 k tiles).  The id of tile k + 2 is drawn by lane 0 of wave 0 during tile k and handed to all waves through
    // an LDS slot per barrier parity; an id >= n_tiles ends the workgroup.  a.tile_ctr == nullptr: static stride.
    const uint32_t grid = gridDim.x;
    const bool dyn = a.tile_ctr != nullptr;
    const uint32_t NC = dyn ? a.n_classes : 1u, cls = blockIdx.x % NC;
    const uint32_t wgc = (grid - cls + NC - 1u) / NC;                                // workgroups (= static first tiles) of this class
    uint32_t* const ctr = a.tile_ctr + 64u * cls;
    auto draw = [&]() -> uint32_t { return cls + NC * (wgc + atomicAdd(ctr, 1u)); };
    if (tid == 0) *(uint32_t*)(lds + kFx2Next + 4u) = dyn ? draw() : blockIdx.x + grid;   // tile 1 (slot of parity 1; read behind the barrier below)
    // verdict in this launch: uncorrectable blocks are counted in LDS and the workgroup adds its sum to the launch's counter once, in front of its
    // done count (global atomics from every wave would have to be waited for -- together with the wave's last pixel stores -- before that count)
    uint32_t* const failp = a.verdict ? (uint32_t*)(lds + kFx2FailWg) : a.fail;
    if (tid == 0) *(uint32_t*)(lds + kFx2FailWg) = 0u;
    stage_tables<TCOP, TBASE, MT>(a, tid, blockDim.x);
    if constexpr (RGB) { if (tid < 82u) *(uint32_t*)(lds + a.dq_off + 4u * tid) = ((const uint32_t*)a.dq)[tid]; }     // yd[244] | cd[84]
    __syncthreads();
#ifdef T3_DEC_STAMPS
    uint64_t st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime(), st_t0 = st_prev, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (a.verdict && blockIdx.x == 0u && wave == 2u * NW - 1u) {                      // the header check (hdr_compare_kernel's job), by a wave that starts idle
        uint32_t want = 0;
#pragma unroll
        for (uint32_t q = 0; q < 24; ++q) want = lane == q ? a.hx[q] : want;           // (kernel arguments are not indexed dynamically)
        bool mis = false;
        if (4u * lane < a.hdr_n) {
            const uint32_t nb = min(4u, a.hdr_n - 4u * lane), mask = nb >= 4u ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
            mis = ((((const uint32_t*)a.hdr_in)[lane] ^ want) & mask) != 0u;
        }
        const bool any = __builtin_amdgcn_ballot_w64(mis) != 0;
        if (lane == 0) a.verdict[0] = any ? 1u : 0u;
    }
    const uint8_t* body = a.in + a.hdr_syms;
    const uint32_t n_items = 9u * a.nb;
    const uint32_t units_tile = (a.TS / 13u) * 3u;                                  // pixels per tile
    uint32_t cur = blockIdx.x, nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u));   // this interval's tile, the next one's

    if (wave < NW) {
        // ---------------- producers: S + E1, two passes of two sets per tile and wave ----------------
        const uint32_t n = lane & 31u, h = lane >> 5;
        // one constant word and one byte offset (of the block in tile 0) per (pass, set) (t3_decode_fx2.h); the constant is made opaque
        // inside the loop, or the compiler unpacks all four ahead of it and spills the pieces (80-VGPR budget)
        Geo geo[2][2]; uint32_t off0[2][2];
        const uint32_t t_off = 26u * a.nb;                                          // from a tile to the next one, in every band
#pragma unroll
        for (uint32_t p = 0; p < 2; ++p) for (uint32_t q = 0; q < 2; ++q) geo[p][q] = fx2_geo<R>(wave * 128u + p * 64u + q * 32u + n, n_items, a.nb, a.div_nb, 0u, off0[p][q]);
        auto has = [&](uint32_t pass, uint32_t set, uint32_t tile) -> bool { Geo g = geo[pass][set]; asm volatile("" : "+v"(g)); return fx2_has_block<R>(g, tile, a.n_tiles, a.nb); };
        // lanes without a block read the first bytes of the body (always there) and ignore them
        auto run_of = [&](uint32_t pass, uint32_t set, uint32_t tile) -> Run<BCN> { return load_run<BCN>(a, body, has(pass, set, tile) ? off0[pass][set] + tile * t_off + 10u * h : 0u); };
        Run<BCN> PA, PB;                                                           // the next pass's two sets, in flight
        PA.w = v4u32{0, 0, 0, 0}; PB.w = PA.w; if constexpr (BCN) { PA.x = 16u << 8; PB.x = PA.x; PA.w4 = 0; PB.w4 = 0; }
        if (cur < a.n_tiles) { PA = run_of(0, 0, cur); PB = run_of(0, 1, cur); }
        for (uint32_t k = 0; cur < a.n_tiles; ++k) {
            const uint32_t tile = cur, buf = k & 1u;
            const uint32_t y_off = a.y_off + buf * a.y_stride, q_off = a.q_off + buf * a.q_stride;
            const uint32_t u2 = 2u * mod3u(tile * a.nb), toff = tile * t_off;
            uint32_t raw; asm volatile("" : "=v"(raw));                                 // the counter value of lane 0's draw (no merge with a default: a copy would wait for it)
#pragma unroll
            for (uint32_t pass = 0; pass < 2; ++pass) {
                uint32_t LA[4], LB[4];
                run_bytes<BCN>(PA, LA); run_bytes<BCN>(PB, LB);
                // the ticket for the tile after the next one: requested before this pass's loads, read after its work (the file is built
                // without the compiler's atomic optimiser, which would read the counter back at once)
                if (pass == 1u) {
                    // (the previous pass's loads are taken into registers first: vmcnt completes in order, and behind the conditional draw
                    // the compiler's conservative wait for them would cover the draw as well)
                    asm volatile("" : "+v"(LA[0]), "+v"(LA[1]), "+v"(LA[2]), "+v"(LA[3]), "+v"(LB[0]), "+v"(LB[1]), "+v"(LB[2]), "+v"(LB[3]));
                    if (tid == 0u && dyn) raw = atomicAdd(ctr, 1u);
                }
                {   // the next pass's input: in flight under this pass (the producers issue no stores, so it is waited for alone)
                    const uint32_t np = pass ^ 1u, nt = pass == 0 ? tile : nxt;
                    if (nt < a.n_tiles) { PA = run_of(np, 0, nt); PB = run_of(np, 1, nt); }
                }
                if (wave * 128u + pass * 64u < n_items) {                             // (wave-uniform) else: nothing left of the tile for this pass
                    Geo gA = geo[pass][0], gB = geo[pass][1]; asm volatile("" : "+v"(gA), "+v"(gB));
                    const Blk bA = fx2_block(gA, off0[pass][0] + toff, fx2_has_block<R>(gA, tile, a.n_tiles, a.nb), u2, y_off);
                    const Blk bB = fx2_block(gB, off0[pass][1] + toff, fx2_has_block<R>(gB, tile, a.n_tiles, a.nb), u2, y_off);
                    const Synd sA = fx2_set<R, TCOP, TBASE, MT>(bA, LA, lane, a.af_off, a.pat_off);
                    Synd sB; sB.lo = 0; sB.hi = 0;
                    if (wave * 128u + pass * 64u + 32u < n_items) sB = fx2_set<R, TCOP, TBASE, MT>(bB, LB, lane, a.af_off, a.pat_off);
                    fx2_own_blocks<R>(a.roots, a.fma_off, failp, sA, sB, bA, bB, (h ? gB : gA) & 0xFFFFu, lane, kFx2Cnt + 4u * buf, q_off, QCAP);
                }
            }
            if (tid == 0u) *(uint32_t*)(lds + kFx2Next + 4u * buf) = dyn ? cls + NC * (wgc + raw) : nxt + grid;
            T3D_STAMP(0);
            barrier_lds2();
            T3D_STAMP(1);
            cur = nxt; nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u * buf));
        }
        barrier_lds2();                                                             // the consumers' last interval
    } else {
        // ---------------- consumers: BM + D5 of the tile the producers finished in the previous interval ----------------
        const uint32_t cw = wave - NW;
#ifdef T3_DEC_CONS_PRIO
        __builtin_amdgcn_s_setprio(T3_DEC_CONS_PRIO);                               // the correction is one long dependent chain: let its steps issue first
#endif
        uint32_t prev = 0;
        for (uint32_t k = 0;; ++k) {
            if (k >= 1u) {
                const uint32_t tile = prev, buf = (k - 1u) & 1u;
                const uint32_t y_off = a.y_off + buf * a.y_stride, q_off = a.q_off + buf * a.q_stride;
                const uint32_t Q = min(*(const uint32_t*)(lds + kFx2Cnt + 4u * buf), QCAP);
                for (uint32_t e0 = cw * 64u; e0 < Q; e0 += 64u * NW) { const uint32_t e = e0 + lane; if (e < Q) fx2_queue_entry<R>(a.roots, a.fma_off, failp, e, q_off, QCAP, y_off); }
                T3D_STAMP(2);
                // rendezvous of the consumer waves: every patch is in LDS before any wave converts symbols
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                uint32_t* const sync = (uint32_t*)__builtin_assume_aligned(lds + kFx2Sync, 4);
                if (lane == 0) __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                {
                    uint32_t spins = 0;
                    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < NW * k) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { if (lane == 0) { *(uint32_t*)(lds + kFx2Abort) = 1u; atomicAdd(failp, 1u << 20); } break; }   // never seen; a bound, not a path
                    }
                }
                if (tid == 64u * NW) *(uint32_t*)(lds + kFx2Cnt + 4u * buf) = 0;         // every consumer has read Q; the producers touch this counter after the barrier
                T3D_STAMP(3);
                const uint64_t unit0 = (uint64_t)tile * units_tile;
                const uint32_t n_here = (uint32_t)min((uint64_t)units_tile, a.n_units > unit0 ? a.n_units - unit0 : 0ull);
                for (uint32_t j = cw * 64u + lane; 4u * j < a.TS / 13u; j += 64u * NW) fx2_pixels12<RGB>(a, j, y_off, unit0, n_here);
                T3D_STAMP(4);
            }
            barrier_lds2();
            T3D_STAMP(5);
            if (cur >= a.n_tiles) break;                                            // the producers had no tile in this interval: that was their closing barrier
            prev = cur; cur = nxt; nxt = __builtin_amdgcn_readfirstlane(*(const uint32_t*)(lds + kFx2Next + 4u * (k & 1u)));
        }
    }
    if (dyn && tid == 0u) {                                                          // re-arm the counters for the next launch on this stream: whoever finishes last
        if (a.verdict) { const uint32_t wgf = *(const uint32_t*)(lds + kFx2FailWg); if (wgf) atomicAdd(a.fail, wgf); }   // (every count precedes the loops' closing barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(a.tile_ctr + 64u * NC, 1u) == grid - 1u) {
            if (a.verdict) a.verdict[1] = atomicExch(a.fail, 0u);                    // uncorrectable blocks of the whole launch (every other workgroup's counts precede its done count)
            for (uint32_t c = 0; c <= NC; ++c) __hip_atomic_store(a.tile_ctr + 64u * c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef T3_DEC_STAMPS
    if ((tid == 0 || tid == 64u * NW) && a.dbg) {
        uint64_t* d = a.dbg + 16ull * blockIdx.x + (tid ? 8 : 0);
        for (int i = 0; i < 6; ++i) d[i] = st_acc[i];
        d[6] = __builtin_amdgcn_s_memtime() - st_t0; d[7] = __builtin_amdgcn_s_memrealtime() - st_rt0;
        if (tid == 0) { d[2] = st_rt0; d[3] = __builtin_amdgcn_s_memrealtime(); }     // producer slots 2, 3: start / end on the 100 MHz clock
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// raw words out: the phases one after the other
// ------------------------------------------------------------------------------------------------------------------
template <int R, bool BCN>
__global__ __launch_bounds__(512, T3_DEC_WAVES_PER_EU) void decode_fixed_kernel(const DecFx2Args a) {
    constexpr uint32_t TCOP = 32, TBASE = kFx2TSeq, MT = kFx2ModSeq;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    stage_tables<TCOP, TBASE, MT>(a, tid, nthr);
    __syncthreads();
    const uint8_t* body = a.in + a.hdr_syms;
    const uint32_t n_items = 9u * a.nb;
    const uint32_t units_tile = (a.TS / 26u) * 3u;                                  // words per tile
    const uint32_t n = lane & 31u, h = lane >> 5;
    uint32_t offA, offB;                                                           // running byte offsets of this wave's two blocks (t3_decode_fx2.h)
    const Geo gA0 = fx2_geo<R>(wave * 64u + n, n_items, a.nb, a.div_nb, blockIdx.x, offA), gB0 = fx2_geo<R>(wave * 64u + 32u + n, n_items, a.nb, a.div_nb, blockIdx.x, offB);
    const uint32_t d_off = 26u * a.nb * gridDim.x;
    Geo gA = gA0, gB = gB0;
    auto run_of = [&](const Geo g, const uint32_t off, const uint32_t tile) -> Run<BCN> { return load_run<BCN>(a, body, fx2_has_block<R>(g, tile, a.n_tiles, a.nb) ? off + 10u * h : 0u); };
    Run<BCN> PA, PB;                                                               // this wave's two sets of the current tile, prefetched
    PA.w = v4u32{0, 0, 0, 0}; PB.w = PA.w; if constexpr (BCN) { PA.x = 16u << 8; PB.x = PA.x; PA.w4 = 0; PB.w4 = 0; }
    if (blockIdx.x < a.n_tiles) { PA = run_of(gA, offA, blockIdx.x); PB = run_of(gB, offB, blockIdx.x); }
    uint32_t par = 0;
    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x, par ^= 1u) {
        const uint32_t u2 = 2u * mod3u(tile * a.nb);
        asm volatile("" : "+v"(gA), "+v"(gB));                                     // opaque: keeps the unpacked pieces out of loop-long registers
        const Blk bA = fx2_block(gA, offA, fx2_has_block<R>(gA, tile, a.n_tiles, a.nb), u2, a.y_off), bB = fx2_block(gB, offB, fx2_has_block<R>(gB, tile, a.n_tiles, a.nb), u2, a.y_off);
        uint32_t LA[4], LB[4];
        run_bytes<BCN>(PA, LA); run_bytes<BCN>(PB, LB);
        const Synd sA = fx2_set<R, TCOP, TBASE, MT>(bA, LA, lane, a.af_off, a.pat_off);
        const Synd sB = fx2_set<R, TCOP, TBASE, MT>(bB, LB, lane, a.af_off, a.pat_off);
        {   // the next tile's input, in flight under this tile's correction phase
            const uint32_t nt = tile + gridDim.x;
            offA += d_off; offB += d_off;
            if (nt < a.n_tiles) { PA = run_of(gA, offA, nt); PB = run_of(gB, offB, nt); }
        }
        fx2_own_blocks<R>(a.roots, a.fma_off, a.fail, sA, sB, bA, bB, (h ? gB0 : gA0) & 0xFFFFu, lane, kFx2Cnt + 4u * par, a.q_off, 512u);   // (bA / bB were built before the offsets advanced)
        barrier_lds2();
        {
            const uint32_t Q = *(const uint32_t*)(lds + kFx2Cnt + 4u * par);
            for (uint32_t e0 = wave * 64u; e0 < Q; e0 += nthr) { const uint32_t e = e0 + lane; if (e < Q) fx2_queue_entry<R>(a.roots, a.fma_off, a.fail, e, a.q_off, 512u, a.y_off); }
            if (tid == 0) *(uint32_t*)(lds + kFx2Cnt + 4u * (par ^ 1u)) = 0;        // the next tile's counter (nobody touches it in this phase)
        }
        barrier_lds2();
        // the loads are waited for BEFORE this tile's stores are issued (vmcnt completes in order, stores count too): they have had
        // the correction phase to land, and the wait does not cover the acknowledgement of stores issued a moment ago
        asm volatile("" : "+v"(PA.w), "+v"(PB.w));
        if constexpr (BCN) asm volatile("" : "+v"(PA.x), "+v"(PB.x), "+v"(PA.w4), "+v"(PB.w4));
        const uint64_t unit0 = (uint64_t)tile * units_tile;
        const uint32_t n_here = (uint32_t)min((uint64_t)units_tile, a.n_units > unit0 ? a.n_units - unit0 : 0ull);
        const uint32_t ng = a.TS / 26u;                                            // groups of 26 symbols -> 3 words (OLD:1022-1040)
        for (uint32_t j = tid; j < ng; j += nthr) words3_from_syms(a.y_off + 26u * j, a.o_off + 27u * j);
        __syncthreads();
        copy_out_lds((uint8_t*)a.out + unit0 * 9u, a.o_off, n_here * 9u, tid, nthr);   // tile starts are only 8-byte aligned
        barrier_lds2();
    }
}

#define T3_INST_DECB(R, BCN) template __global__ void decode_fixed_kernel<R, BCN>(const DecFx2Args); template __global__ void decode_fixed_px_kernel<R, false, BCN>(const DecFx2Args); \
    template __global__ void decode_fixed_px_kernel<R, true, BCN>(const DecFx2Args);
#define T3_INST_DEC(R) T3_INST_DECB(R, false) T3_INST_DECB(R, true)
T3_INST_DEC(2) T3_INST_DEC(4) T3_INST_DEC(6) T3_INST_DEC(8)

}  // namespace t3
