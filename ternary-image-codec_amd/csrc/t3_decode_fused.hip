// t3_decode_fused.hip — fused FIXED-mode ("v6c") decoder for gfx950: K3 syndromes + K4 Berlekamp–Massey / Chien / Forney +
// K5 symbols -> pixels|words in ONE launch, tiled like the encoder (BASELINE config 5: decode with injected trit errors).
// Replaces, for mode=FIXED / uniform k / 1-D / no beacon: descramble (OLD:938-947), per-band RS decode (OLD:963-991,
// decode_block OLD:546-662 with the Forney sign fixed), the i%9 re-merge the reference omits, symbols -> 26-trit words
// (OLD:1022-1040) and unpack_two_pixels (OLD:706-722).  Everything else goes through t3_decode_stream.hip / t3_decode.hip.
//
// Per tile (9 bands x nb blocks, eight waves):
//   S   a wave takes two sets of 32 blocks, two lanes per block (t3_decode_fx2.h): one 16-byte load per lane, one conflict-free
//       T-table read per coded symbol (descramble + trit expansion), syndromes on the matrix cores (78 x 3r GF(3) product as four
//       v_mfma_i32_32x32x32_i8), mod-3 fold in the VALU, data symbols -> stream order in LDS (byte 9(mk+p)+b)
//   E1  one lane = one block: non-zero syndromes that form a geometric progression are a single error, fixed in place
//       (two logarithm reads and one patch); the other flagged blocks are appended to an LDS queue, one wave-aggregated
//       counter update per wave (ballot + prefix count)
//   BM  after a barrier full waves drain the queue: Berlekamp-Massey on a fused multiply-add table (a + x y, 27^3 bytes in LDS),
//       Chien search by table (root mask per locator, global memory), Forney, <= t bytes patched
//   D5  pixels: one lane = four triples, 13 aligned dwords of symbols -> 12 pixels with packed 16-bit ops -> 72 bytes stored
//       straight to memory;  raw words: 26 symbols -> 3 words, staged in LDS, copied out with 16-byte coalesced stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"
#include "t3_decode_fx.h"
#include "t3_decode_fx2.h"

namespace t3 {

namespace {
__device__ __forceinline__ void barrier_lds2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// A tile's input is fetched one tile ahead into registers (plain loads, issued right after the previous tile's sets).  vmcnt
// completes in order and stores count too, so the loads are WAITED FOR before this tile's pixel stores are issued (an empty
// asm statement that names the registers makes the compiler place its s_waitcnt there): the wait then covers loads that have
// had the whole correction phase to land, not the acknowledgement of stores issued a moment ago.
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(2))) V4a2 { v4u32 v; };
__device__ __forceinline__ v4u32 load16(const uint8_t* p) { return ((const V4a2*)p)->v; }
}

#ifndef T3_DEC_WAVES_PER_EU
#define T3_DEC_WAVES_PER_EU 6   // <= 80 VGPRs: three 8-wave workgroups per CU
#endif
#ifdef T3_DEC_STAMPS   // diagnostic build: per-phase cycle sums of wave 0 (never in the product build)
#define T3D_STAMP(i) do { const uint64_t t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; } while (0)
#else
#define T3D_STAMP(i) do { } while (0)
#endif
template <int R, bool TO_PIXELS>
__global__ __launch_bounds__(512, T3_DEC_WAVES_PER_EU) void decode_fixed_kernel(const DecFx2Args a) {
    constexpr uint32_t K = 26 - R, H = R / 2;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // constants -> LDS
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) { Row r; r.blocks = a.band_blocks[b]; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b]; *(Row*)(lds + 16 * b) = r; }
        *(uint32_t*)(lds + kFx2Cnt) = 0; *(uint32_t*)(lds + kFx2Cnt + 4) = 0;
    }
    for (uint32_t i = tid * 16u; i < (uint32_t)kFx2SmallBytes; i += nthr * 16u) *(uint4*)(lds + kFx2Small + i) = *(const uint4*)(a.small + i);
    for (uint32_t i = tid * 16u; i < (uint32_t)sizeof(FxTables); i += nthr * 16u) *(uint4*)(lds + kFxTab + i) = *(const uint4*)((const uint8_t*)a.tab + i);
    for (uint32_t i = tid * 16u; i < (uint32_t)kSyndTBytes; i += nthr * 16u) *(uint4*)(lds + kFx2T + i) = *(const uint4*)((const uint8_t*)a.ttab + i);
    for (uint32_t i = tid * 16u; i < 19696u; i += nthr * 16u) *(uint4*)(lds + a.fma_off + i) = *(const uint4*)(a.fma + i);
    for (uint32_t i = tid * 16u; i < 4096u; i += nthr * 16u) *(uint4*)(lds + a.af_off + i) = *(const uint4*)((const uint8_t*)a.afrag + i);
    __syncthreads();
#ifdef T3_DEC_STAMPS
    uint64_t st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = __builtin_amdgcn_s_memtime(), st_t0 = st_prev, st_rt0 = __builtin_amdgcn_s_memrealtime();
#endif

    const uint8_t* body = a.in + a.hdr_syms;
    const uint32_t n_items = 9u * a.nb;
    const uint32_t units_tile = TO_PIXELS ? (a.TS / 13u) * 3u : (a.TS / 26u) * 3u;       // pixels / words produced per tile
    const uint32_t n = lane & 31u, h = lane >> 5;
    // lanes without a block read the first bytes of the body (always there) and ignore them
    auto src_of = [&](const Blk& b) -> const uint8_t* { return b.valid ? b.g + 10u * h : body; };
    v4u32 PA = {0, 0, 0, 0}, PB = {0, 0, 0, 0};                                    // this wave's two sets of the current tile, prefetched
    if (blockIdx.x < a.n_tiles) {
        PA = load16(src_of(fx2_block_of<R>(wave * 64u + n, n_items, a.nb, a.div_nb, blockIdx.x, body, a.y_off)));
        PB = load16(src_of(fx2_block_of<R>(wave * 64u + 32u + n, n_items, a.nb, a.div_nb, blockIdx.x, body, a.y_off)));
    }
    uint32_t par = 0;
    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x, par ^= 1u) {
        // ---------------- S: two sets of 32 blocks per wave ----------------
        const Blk bA = fx2_block_of<R>(wave * 64u + n, n_items, a.nb, a.div_nb, tile, body, a.y_off);
        const Blk bB = fx2_block_of<R>(wave * 64u + 32u + n, n_items, a.nb, a.div_nb, tile, body, a.y_off);
        T3D_STAMP(0);
        const uint32_t LA[4] = {PA[0], PA[1], PA[2], PA[3]}, LB[4] = {PB[0], PB[1], PB[2], PB[3]};
        const Synd sA = fx2_set<R>(bA, LA, lane, a.af_off, a.cyc24, a.pre0, a.pre1);
        const Synd sB = fx2_set<R>(bB, LB, lane, a.af_off, a.cyc24, a.pre0, a.pre1);
        {   // the next tile's input, in flight under this tile's correction phase
            const uint32_t nt = tile + gridDim.x;
            if (nt < a.n_tiles) {
                PA = load16(src_of(fx2_block_of<R>(wave * 64u + n, n_items, a.nb, a.div_nb, nt, body, a.y_off)));
                PB = load16(src_of(fx2_block_of<R>(wave * 64u + 32u + n, n_items, a.nb, a.div_nb, nt, body, a.y_off)));
            }
        }
        // ---------------- E1: one lane = one block (lower half-wave: set A, upper: set B) ----------------
        {
            Synd own; own.lo = h ? sB.lo : sA.lo; own.hi = h ? sB.hi : sA.hi;
            const bool valid = h ? bB.valid : bA.valid;
            const uint32_t yb = h ? bB.yb : bA.yb;
            bool flagged = valid && (own.lo | own.hi) != 0u;                        // OLD:562: all-zero syndromes -> nothing to do
            if (flagged) flagged = fx2_single<R>(own, yb, a.fma_off) == 0u;
            const uint64_t bal = __builtin_amdgcn_ballot_w64(flagged);
            if (bal != 0ull) {                                                      // wave-aggregated append: one LDS atomic per wave
                const uint32_t cnt = (uint32_t)__popcll(bal);
                uint32_t base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add((uint32_t*)__builtin_assume_aligned(lds + kFx2Cnt + 4u * par, 4), cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
                if (flagged) {
                    const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    *T3_LP(u32x2, a.q_off + 8u * slot) = u32x2{own.lo, own.hi};         // the r syndromes ...
                    *T3_LP(uint16_t, a.q_off + 4096u + 2u * slot) = (uint16_t)(wave * 64u + lane);   // ... and the block's item number (< 512)
                }
            }
        }
        T3D_STAMP(1);
        barrier_lds2();
        T3D_STAMP(2);
        // ---------------- BM: full waves drain the queue ----------------
        {
            const uint32_t Q = *(const uint32_t*)(lds + kFx2Cnt + 4u * par);
            for (uint32_t e0 = wave * 64u; e0 < Q; e0 += nthr) {
                const uint32_t e = e0 + lane;
                if (e < Q) {
                    const u32x2 sy = *T3_LP(const u32x2, a.q_off + 8u * e);
                    const uint32_t item = *T3_LP(const uint16_t, a.q_off + 4096u + 2u * e);
                    uint32_t S[R];
#pragma unroll
                    for (uint32_t j = 0; j < (uint32_t)R; ++j) S[j] = ((j < H ? sy.x : sy.y) >> (8u * (j % H))) & 0xFFu;
                    const uint32_t bi = min(__umulhi(item, a.div_nb.mul) >> a.div_nb.sh, 8u), m = item - bi * a.nb;
                    const uint32_t yb = a.y_off + bi + 9u * K * m;
                    Fix fx; fx.np = 0;
                    uint32_t rc = fx2_correct<R>(S, fx, a.roots, a.fma_off);
                    if (rc == 2u) rc = fx_correct<R>(S, fx, a.roots, a.fma_off) ? 0u : 1u;   // longer register than t: the full-length routine decides
                    if (rc != 0u) atomicAdd(a.fail, 1u);
                    else {
#pragma unroll
                        for (int q = 0; q < R / 2; ++q)
                            if ((uint32_t)q < fx.np && fx.pos[q] < K) { const uint32_t ad = yb + 9u * fx.pos[q]; *T3_LP(uint8_t, ad) = (uint8_t)l8(a.fma_off + (54u + fx.mag[q]) * 27u + l8(ad)); }
                    }
                }
            }
            if (tid == 0) *(uint32_t*)(lds + kFx2Cnt + 4u * (par ^ 1u)) = 0;        // the next tile's counter (nobody touches it in this phase)
        }
        T3D_STAMP(3);
        barrier_lds2();
        T3D_STAMP(2);
        asm volatile("" : "+v"(PA), "+v"(PB));                                     // the next tile's input has landed (see load16)
        // ---------------- D5: symbols -> output units ----------------
        const uint64_t unit0 = (uint64_t)tile * units_tile;
        const uint32_t n_here = (uint32_t)min((uint64_t)units_tile, a.n_units > unit0 ? a.n_units - unit0 : 0ull);
        if constexpr (TO_PIXELS) {
            // One lane = four consecutive triples: 52 symbols (13 aligned dwords of Y) -> 12 pixels = 72 bytes, stored straight
            // to memory (lanes are consecutive, so a wave writes one contiguous run).  Triples 0/2 and 1/3 share registers as
            // 16-bit halves; inverse of the encoder's splice (unpack_two_pixels OLD:706-722) in packed arithmetic.
            const uint32_t ntr = a.TS / 13u;                                       // triples in the tile (a multiple of 4, at most 4 * 512)
            const uint32_t j = tid;
            if (4u * j < ntr) {
                uint32_t D[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) D[i] = *T3_LP(const uint32_t, a.y_off + 52u * j + 4u * i);
                uint32_t o[18];
                px12_from_syms(D, o);
                uint8_t* g = (uint8_t*)a.out + (unit0 + 12ull * j) * 6u;            // 8-byte aligned
                const bool whole = 12u * j + 12u <= n_here;
                if (whole) {
                    typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(8)));
                    typedef uint32_t v2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                    for (int d = 0; d < 4; ++d) *(v4u*)(g + 16 * d) = v4u{o[4 * d], o[4 * d + 1], o[4 * d + 2], o[4 * d + 3]};
                    *(v2u*)(g + 64) = v2u{o[16], o[17]};
                } else {                                                            // the frame's last pixels: per 16-bit component
#pragma unroll
                    for (uint32_t hh = 0; hh < 36; ++hh)
                        if (12u * j + hh / 3u < n_here) *(uint16_t*)(g + 2u * hh) = (uint16_t)(o[hh >> 1] >> (16u * (hh & 1u)));
                }
            }
        } else {
            const uint32_t ng = a.TS / 26u;                                        // groups of 26 symbols -> 3 words (OLD:1022-1040)
            for (uint32_t j = tid; j < ng; j += nthr) words3_from_syms(a.y_off + 26u * j, a.o_off + 27u * j);
            __syncthreads();
            copy_out_lds((uint8_t*)a.out + unit0 * 9u, a.o_off, n_here * 9u, tid, nthr);   // tile starts are only 8-byte aligned
        }
        T3D_STAMP(4);
        barrier_lds2();
        T3D_STAMP(2);
    }
#ifdef T3_DEC_STAMPS
    if (tid == 0 && a.dbg) {
        uint64_t* d = a.dbg + 8ull * blockIdx.x;
        d[0] = st_acc[0]; d[1] = st_acc[1]; d[2] = st_acc[2]; d[3] = st_acc[3]; d[4] = st_acc[4];
        d[5] = __builtin_amdgcn_s_memtime() - st_t0; d[6] = __builtin_amdgcn_s_memrealtime() - st_rt0;
    }
#endif
}

template __global__ void decode_fixed_kernel<2, true>(const DecFx2Args);  template __global__ void decode_fixed_kernel<2, false>(const DecFx2Args);
template __global__ void decode_fixed_kernel<4, true>(const DecFx2Args);  template __global__ void decode_fixed_kernel<4, false>(const DecFx2Args);
template __global__ void decode_fixed_kernel<6, true>(const DecFx2Args);  template __global__ void decode_fixed_kernel<6, false>(const DecFx2Args);
template __global__ void decode_fixed_kernel<8, true>(const DecFx2Args);  template __global__ void decode_fixed_kernel<8, false>(const DecFx2Args);

}  // namespace t3
