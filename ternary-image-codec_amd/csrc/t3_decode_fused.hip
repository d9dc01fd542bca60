// t3_decode_fused.hip — fused FIXED-mode ("v6c") decoder for gfx950: K3 syndromes + K4 Berlekamp–Massey / Chien / Forney +
// K5 symbols -> pixels|words in ONE launch, tiled like the encoder (BASELINE config 5: decode with injected trit errors).
// Replaces, for mode=FIXED / uniform k / 1-D / no beacon: descramble (OLD:938-947), per-band RS decode (OLD:963-991,
// decode_block OLD:546-662 with the Forney sign fixed), the i%9 re-merge the reference omits, symbols -> 26-trit words
// (OLD:1022-1040) and unpack_two_pixels (OLD:706-722).  Everything else goes through the generic kernels in t3_decode.hip.
//
// One lane = one RS block, blocks dealt linearly over the eight waves.  Per tile:
//   D1  7 aligned dword loads per lane (blocks are 26 B, 2-byte aligned) -> 26 symbols in registers
//   D2  descramble through an 81-byte LDS table (result pre-scaled by 4), syndromes through a per-position LUT:
//       6-bit SWAR trit fields in four 27-dword tables (27 consecutive dwords = 27 banks: conflict-free gathers), one mod-3
//       fold per block
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
#include "t3_decode_fx.h"

namespace t3 {


template <int R, bool TO_PIXELS>
__global__ __launch_bounds__(512, 4) void decode_fixed_kernel(const DecFxArgs a) {
    constexpr uint32_t K = 26 - R;
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
                const FxCtx c{a.in, a.in_bytes, a.hdr_syms, a.cyc24, a.pre0, a.pre1, a.fail, a.roots, a.fma_off};
                fx_block<R>(c, rw, mg, kFxLut, a.y_off + b + 9u * K * m);
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
                px12_from_syms(D, o);
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
