// t3_decode_stream.hip — FIXED-mode ("v6c") decoder for the framings the fully fused kernel (t3_decode_fused.hip) does not
// take: per-band k (UEP, OLD:1089-1100), the 2-D boustrophedon interleave (deinterleave OLD:781-813) and the sparse
// beacon (OLD:952-957 / 1118-1141).  Three kernels:
//   debeacon_kernel      framed body -> band-serial body (beacon symbols dropped), only when the beacon is on
//   decode_stream_kernel D1-D4 of the fused decoder (fx_block: loads, descramble, syndrome LUT, Berlekamp-Massey / root table /
//                        Forney) with the bands grouped by k: a group owns whole waves of a tile, so a wave runs one code's
//                        routine; the corrected data symbols of a tile go to a stream-ordered scratch with 16-byte stores
//   emit_stream_kernel   a span of the stream -> LDS [through the de-interleave map: rows map onto themselves, so the span
//                        extended to whole rows is one contiguous read; 16-byte granules when rows are multiples of 16]
//                        -> pixels (one lane = four triples, as D5 of the fused kernel) or 26-trit words
// The scratch costs one extra write + read of the data symbols (2 x 144 MB per 8K frame) over the fused kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"
#include "t3_decode_fx.h"

namespace t3 {

namespace {
__device__ __forceinline__ uint32_t fdv(uint32_t n, const DevDiv& d) { return d.d <= 1 ? n : (__umulhi(n, d.mul) >> d.sh); }

constexpr uint32_t kStGrp = 160, kStGrpStride = 48;     // LDS header: 9 band rows (16 B), then the group records
struct GrpRec { uint32_t r, nb, n_items, wave0, lut_off, pad_; const uint32_t* roots; uint8_t bands[12]; uint32_t pad2_; };
static_assert(sizeof(GrpRec) == kStGrpStride && offsetof(GrpRec, roots) == 24 && offsetof(GrpRec, bands) == 32, "group record");
static_assert(kStGrp + kStMaxGrp * kStGrpStride <= kFxHdr, "LDS header");
}  // namespace

__global__ __launch_bounds__(256) void debeacon_kernel(const DebeaconArgs a) {
    // body symbol q sits in the framed stream at: group t = q / (9 period - 1); its first 8 symbols share a word with the
    // beacon (slot `slot`), the rest follow contiguously (the encoder's insertion, OLD:1118-1141, inverted): position
    // q + t + 1 for the contiguous part.  One lane = 16 body symbols: a shifted 16-byte copy (aligned dword loads + funnel
    // shifts) unless the run touches a group's first word.
    const uint32_t per = 9u * a.period - 1u;
    for (uint64_t q0 = 16ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x); q0 < a.body_syms; q0 += 16ull * gridDim.x * blockDim.x) {
        uint64_t t = q0 / per; uint32_t rem = (uint32_t)(q0 - t * per);
        if (rem >= 8u && rem + 16u <= per && q0 + 16 <= a.body_syms && q0 + t + 1u + 20u <= a.framed_bytes) {
            const uint8_t* src = a.framed + q0 + t + 1u;
            const uint32_t sh = ((uint32_t)(uintptr_t)src & 3u) * 8u;
            const uint32_t* p = (const uint32_t*)((uintptr_t)src & ~(uintptr_t)3);     // reads < 4 bytes past the run: checked against the buffer above
            uint32_t dw[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) dw[i] = p[i];
            *(uint4*)(a.body + q0) = make_uint4(__builtin_amdgcn_alignbit(dw[1], dw[0], sh), __builtin_amdgcn_alignbit(dw[2], dw[1], sh),
                                                __builtin_amdgcn_alignbit(dw[3], dw[2], sh), __builtin_amdgcn_alignbit(dw[4], dw[3], sh));
        } else {
            for (uint32_t i = 0; i < 16u && q0 + i < a.body_syms; ++i) {
                const uint64_t w0 = 9ull * t * a.period;
                const uint64_t pos = rem < 8u ? w0 + (rem < a.slot ? rem : rem + 1u) : w0 + 9u + (rem - 8u);
                a.body[q0 + i] = a.framed[pos];
                if (++rem == per) { rem = 0; ++t; }
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void decode_stream_kernel(const DecStArgs a) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63u, nwv = nthr >> 6;
    if (tid == 0) {
#pragma unroll
        for (int b = 0; b < 9; ++b) { Row r; r.blocks = a.band_blocks[b]; r.boff6 = a.band_boff6[b]; r.body_off = a.band_body_off[b]; *(Row*)(lds + 16 * b) = r; }
#pragma unroll
        for (int g = 0; g < kStMaxGrp; ++g) {
            GrpRec q; q.r = a.grp[g].r; q.nb = a.grp[g].nb; q.n_items = a.grp[g].n_items; q.wave0 = a.grp[g].wave0; q.lut_off = a.grp[g].lut_off; q.pad_ = 0; q.roots = a.grp[g].roots; q.pad2_ = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) q.bands[i] = a.grp[g].bands[i];
            *(GrpRec*)(lds + kStGrp + kStGrpStride * g) = q;
        }
    }
    for (uint32_t i = tid * 16u; i < (uint32_t)sizeof(FxTables); i += nthr * 16u) *(uint4*)(lds + kFxTab + i) = *(const uint4*)((const uint8_t*)a.tab + i);
#pragma unroll
    for (int g = 0; g < kStMaxGrp; ++g)
        if ((uint32_t)g < a.n_grp)
            for (uint32_t i = tid * 16u; i < a.grp[g].lut_bytes; i += nthr * 16u) *(uint4*)(lds + a.grp[g].lut_off + i) = *(const uint4*)((const uint8_t*)a.grp[g].lut + i);
    for (uint32_t i = tid * 16u; i < 19696u; i += nthr * 16u) *(uint4*)(lds + a.fma_off + i) = *(const uint4*)(a.fma + i);
    __syncthreads();

    for (uint32_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        for (uint32_t s = wave; s < a.n_slots; s += nwv) {
            uint32_t g = 0;
#pragma unroll
            for (uint32_t q = 1; q < (uint32_t)kStMaxGrp; ++q) if (q < a.n_grp && s >= a.grp[q].wave0) g = q;
            const uint32_t gb = kStGrp + kStGrpStride * g;                       // the group's record (wave-uniform)
            auto gw = [gb](uint32_t field) -> uint32_t { return __builtin_amdgcn_readfirstlane(*T3_LP(const uint32_t, gb + 4u * field)); };
            const uint32_t r = gw(0), nb = gw(1), n_items = gw(2), wave0 = gw(3), lut_off = gw(4);
            const uint32_t* roots = (const uint32_t*)(((uint64_t)gw(7) << 32) | gw(6));
            const uint32_t item = (s - wave0) * 64u + lane;
            if (item < n_items) {
                const uint32_t bi = item / nb, m = item - bi * nb;
                const uint32_t b = l8(gb + 32u + bi);
                const Row rw = row(b);
                const uint64_t mg = (uint64_t)tile * nb + m;
                if (mg < rw.blocks) {
                    const FxCtx cx{a.in, a.in_bytes, a.hdr_syms, a.cyc24, a.pre0, a.pre1, a.fail, roots, a.fma_off};
                    const uint32_t yb = a.y_off + b + 9u * (26u - r) * m;
                    switch (r) {
                        case 2: fx_block<2>(cx, rw, mg, lut_off, yb); break;
                        case 4: fx_block<4>(cx, rw, mg, lut_off, yb); break;
                        case 6: fx_block<6>(cx, rw, mg, lut_off, yb); break;
                        default: fx_block<8>(cx, rw, mg, lut_off, yb); break;
                    }
                }
            }
        }
        __syncthreads();
        const uint64_t s0 = (uint64_t)tile * a.TS;
        if (s0 < a.n_sym) copy_out_lds(a.ystream + s0, a.y_off, (uint32_t)min((uint64_t)a.TS, a.n_sym - s0), tid, nthr);
        __syncthreads();
    }
}

namespace {
// whole rows of the interleave grid that [v, v] touches: chunks of il_A symbols, rows of il_w, the last chunk ragged (OLD:781-813)
__device__ __forceinline__ void row_of(uint32_t v, const EmitStArgs& a, uint32_t& row_lo, uint32_t& row_len, uint32_t& odd) {
    const uint32_t chunk = fdv(v, a.div_A), base = chunk * a.il_A, rem = v - base;
    const uint32_t take = min(a.il_A, a.n_sym - base);
    const uint32_t r = fdv(rem, a.div_w);
    row_lo = base + r * a.il_w; row_len = min(a.il_w, take - r * a.il_w); odd = r & 1u;
}
}  // namespace

template <bool TO_PIXELS>
__global__ __launch_bounds__(512, 2) void emit_stream_kernel(const EmitStArgs a) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    const uint32_t units_step = TO_PIXELS ? (a.span / 13u) * 3u : (a.span / 26u) * 3u;
    // The de-interleave map is an involution inside every row segment (odd rows reversed, OLD:782-813): a step's pre-interleave symbols
    // [U0, U1) are gathered straight from their mirrored places in the corrected stream, whatever the row width -- nothing but the step
    // itself is read or staged.  The next step's granules (four per thread) are requested into registers before this step's conversion and
    // written to LDS after it, so the loads' latency lies under the conversion and its stores.
    constexpr uint32_t G = 4;                                                       // 16-byte granules per thread and step (span <= 16 * G * 512)
    const bool regs = (!a.il_on || a.il_fast) && a.span <= 16u * G * nthr;
    uint4 R[G]; uint32_t have = 0;
    auto fetch = [&](uint32_t st) {
        have = 0;
        const uint32_t U0 = st * a.span, U1 = (uint32_t)min((uint64_t)U0 + a.span, (uint64_t)a.n_sym);
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const uint32_t v = U0 + 16u * (tid + k * nthr);
            if (v >= U1) continue;
            if (!a.il_on) { R[k] = *(const uint4*)(a.ystream + v); have |= 1u << k; continue; }
            // rows and chunks are multiples of 16 symbols: a 16-byte granule stays inside one row; an odd row reverses it
            uint32_t rl, rn, od; row_of(v, a, rl, rn, od);
            if (!od) { R[k] = *(const uint4*)(a.ystream + v); have |= 1u << k; }
            else if (rn == a.il_w) {
                const uint4 q = *(const uint4*)(a.ystream + (rl + (a.il_w - 16u - (v - rl))));
                R[k] = make_uint4(__builtin_bswap32(q.w), __builtin_bswap32(q.z), __builtin_bswap32(q.y), __builtin_bswap32(q.x)); have |= 1u << k;
            }                                                                        // else: the stream's last, shorter row -- placed byte by byte in commit()
        }
    };
    if (regs && blockIdx.x < a.n_steps) fetch(blockIdx.x);
    for (uint32_t step = blockIdx.x; step < a.n_steps; step += gridDim.x) {
        const uint32_t U0 = step * a.span, U1 = (uint32_t)min((uint64_t)U0 + a.span, (uint64_t)a.n_sym);
        auto at = [&](uint32_t v) -> uint32_t { return a.sym_off + (v - U0); };     // LDS address of pre-interleave symbol v (U0: 16-byte aligned)
        if (regs) {
#pragma unroll
            for (uint32_t k = 0; k < G; ++k) {
                const uint32_t v = U0 + 16u * (tid + k * nthr);
                if (v >= U1) continue;
                if (have >> k & 1u) *(uint4*)(lds + at(v)) = R[k];
                else { uint32_t rl, rn, od; row_of(v, a, rl, rn, od); for (uint32_t i = 0; i < 16u && v + i < rl + rn; ++i) lds[at(v + i)] = a.ystream[rl + (rn - 1u - (v + i - rl))]; }
            }
        } else if (!a.il_on) {
            for (uint32_t v = U0 + 16u * tid; v < U1; v += 16u * nthr) *(uint4*)(lds + at(v)) = *(const uint4*)(a.ystream + v);
        } else {
            for (uint32_t v = U0 + tid; v < U1; v += nthr) {
                uint32_t rl, rn, od; row_of(v, a, rl, rn, od);
                lds[at(v)] = a.ystream[od ? rl + (rn - 1u - (v - rl)) : v];
            }
        }
        __syncthreads();
        if (regs && step + gridDim.x < a.n_steps) fetch(step + gridDim.x);
        const uint64_t unit0 = (uint64_t)step * units_step;
        const uint32_t n_here = (uint32_t)min((uint64_t)units_step, a.n_units > unit0 ? a.n_units - unit0 : 0ull);
        const uint32_t y0 = at(U0);                                             // 16-byte aligned: span and lo16 are
        if constexpr (TO_PIXELS) {
            for (uint32_t j = tid; 12u * j < n_here; j += nthr) {
                uint32_t D[13];
#pragma unroll
                for (int i = 0; i < 13; ++i) D[i] = *T3_LP(const uint32_t, y0 + 52u * j + 4u * i);
                uint32_t o[18];
                px12_from_syms(D, o);
                uint8_t* g = (uint8_t*)a.out + (unit0 + 12ull * j) * 6u;
                if (12u * j + 12u <= n_here) {
                    typedef uint32_t v4u __attribute__((ext_vector_type(4), aligned(8)));
                    typedef uint32_t v2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
                    for (int d = 0; d < 4; ++d) *(v4u*)(g + 16 * d) = v4u{o[4 * d], o[4 * d + 1], o[4 * d + 2], o[4 * d + 3]};
                    *(v2u*)(g + 64) = v2u{o[16], o[17]};
                } else {
#pragma unroll
                    for (uint32_t h = 0; h < 36; ++h)
                        if (12u * j + h / 3u < n_here) *(uint16_t*)(g + 2u * h) = (uint16_t)(o[h >> 1] >> (16u * (h & 1u)));
                }
            }
        } else {
            for (uint32_t j = tid; 3u * j < n_here; j += nthr) words3_from_syms(y0 + 26u * j, a.o_off + 27u * j);
            __syncthreads();
            copy_out_lds((uint8_t*)a.out + unit0 * 9u, a.o_off, n_here * 9u, tid, nthr);
        }
        __syncthreads();
    }
}

template __global__ void emit_stream_kernel<true>(const EmitStArgs);
template __global__ void emit_stream_kernel<false>(const EmitStArgs);

}  // namespace t3
