// t3_subword.hip — SURVEY §8 row f3: subword trit streams and the wire packings, as gather kernels for gfx950.
//   subword_extract_kernel   extract_subword_stream_from_words   OLD:834-844  (first N trits of every word, one byte per trit)
//   subword_build_kernel     build_words_from_subword_stream     OLD:845-859  (N trits per word, the rest = fill; last word zero-padded)
//   base243_pack_kernel      tpack::ut_to_base243                TPACK:28-38  (u32 trit count, then 5 trits per byte, LSD first)
//   base243_unpack_kernel    tpack::base243_to_ut                TPACK:40-50
//   mod27_bytes_kernel       tpack::words_to_bytes / bytes_to_words  TPACK:53-65  (every byte % 27)
// All are pure HBM streams (1 byte per trit on the trit side); one lane moves a dword-aligned group so that loads and
// stores are whole dwords wherever the stream lengths allow.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_subword.h"

namespace t3 {

namespace {
__device__ __forceinline__ uint32_t d3(uint32_t x) { return (x * 171u) >> 9; }     // x < 512
__device__ __forceinline__ uint32_t m27(uint32_t x) { return x - 27u * ((x * 2428u) >> 16); }   // x < 256: unpack3 reduces the digit (OLD:28-31)
}  // namespace

// A workgroup step takes 512 words (4608 bytes, 16-byte aligned in the stream): every lane expands dwords of four symbols
// into twelve trit bytes in LDS (27 bytes per word), then the N leading trits of each word leave as whole 16-byte stores
// (N * 512 is a multiple of 16): straight from the image when N = 27, gathered byte-wise otherwise.  Loads and stores are
// coalesced; the first version (one lane = four output trits read through the vector L1) reached 1.2 TB/s.
__global__ __launch_bounds__(256) void subword_extract_kernel(const uint8_t* __restrict__ words, uint64_t n_words, int N, uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t img[27 * 512 + 16];
    const uint32_t tid = threadIdx.x, uN = (uint32_t)N;
    const uint64_t n_steps = (n_words + 511) / 512;
    const bool fast_io = (((uintptr_t)words | (uintptr_t)out) & 15u) == 0;
    for (uint64_t step = blockIdx.x; step < n_steps; step += gridDim.x) {
        const uint64_t w0 = step * 512;
        const uint32_t nw = (uint32_t)min((uint64_t)512, n_words - w0);
        const uint8_t* src = words + 9 * w0;
        for (uint32_t q = tid; 4u * q < 9u * nw; q += 256u) {                  // dword q = symbols 4q .. 4q+3 of the step
            uint32_t v;
            if (fast_io && 4u * q + 4u <= 9u * nw) v = *(const uint32_t*)(src + 4u * q);
            else { v = 0; for (uint32_t i = 0; i < 4u && 4u * q + i < 9u * nw; ++i) v |= (uint32_t)src[4u * q + i] << (8u * i); }
            uint32_t o[3] = {0, 0, 0};
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t sy = m27((v >> (8u * i)) & 0xFFu), q1 = d3(sy), q2 = d3(q1);
                const uint32_t t3 = (sy - 3u * q1) | (q1 - 3u * q2) << 8 | q2 << 16;   // three trit bytes
                const uint32_t bit = 24u * i;                                            // position in the 96-bit group
                o[bit >> 5] |= t3 << (bit & 31u);
                if ((bit & 31u) > 8u) o[(bit >> 5) + 1u] |= t3 >> (32u - (bit & 31u));
            }
            *(uint32_t*)(img + 12u * q) = o[0]; *(uint32_t*)(img + 12u * q + 4u) = o[1]; *(uint32_t*)(img + 12u * q + 8u) = o[2];
        }
        __syncthreads();
        const uint32_t n_out = uN * nw;                                         // trits this step produces
        uint8_t* dst = out + (uint64_t)uN * w0;
        for (uint32_t u = tid; 16u * u < n_out; u += 256u) {
            const uint32_t t0 = 16u * u;
            uint32_t r[4];
            if (uN == 27u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = *(const uint32_t*)(img + t0 + 4u * k);
            } else {
                uint32_t w = t0 / uN, i = t0 - w * uN;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t x = 0;
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) { x |= (uint32_t)img[27u * w + i] << (8u * b); if (++i == uN) { i = 0; ++w; } }
                    r[k] = x;
                }
            }
            if (fast_io && t0 + 16u <= n_out) *(uint4*)(dst + t0) = make_uint4(r[0], r[1], r[2], r[3]);
            else for (uint32_t b = 0; b < 16u && t0 + b < n_out; ++b) dst[t0 + b] = (uint8_t)(r[b >> 2] >> (8u * (b & 3u)));
        }
        __syncthreads();
    }
}

// Inverse of the extraction, same staging: a workgroup step builds 512 words.  Their N * 512 trits (a multiple of 16 bytes)
// enter LDS with 16-byte loads; one lane then packs one output dword = four symbols (slots >= N of a word = fill, a slot past
// the end of the stream = 0: the last word is zero-padded, OLD:845-859) and stores it.  First version: one lane = one word,
// 27 byte loads and 9 byte stores, 1.9 TB/s.
__global__ __launch_bounds__(256) void subword_build_kernel(const uint8_t* __restrict__ trits, uint64_t n_trits, int N, uint32_t fill, uint8_t* __restrict__ words, uint64_t n_words) {
    __shared__ __attribute__((aligned(16))) uint8_t lin[27 * 512 + 16];
    const uint32_t tid = threadIdx.x, uN = (uint32_t)N;
    const uint64_t n_steps = (n_words + 511) / 512;
    const bool fast_io = (((uintptr_t)trits | (uintptr_t)words) & 15u) == 0;
    for (uint64_t step = blockIdx.x; step < n_steps; step += gridDim.x) {
        const uint64_t w0 = step * 512, t0 = w0 * uN;
        const uint32_t nw = (uint32_t)min((uint64_t)512, n_words - w0);
        const uint32_t n_in = (uint32_t)min((uint64_t)uN * nw, n_trits - t0);    // trits of the stream that belong to this step
        for (uint32_t u = tid; 16u * u < n_in; u += 256u) {
            if (fast_io && 16u * u + 16u <= n_in) *(uint4*)(lin + 16u * u) = *(const uint4*)(trits + t0 + 16u * u);
            else for (uint32_t b = 0; b < 16u && 16u * u + b < n_in; ++b) lin[16u * u + b] = trits[t0 + 16u * u + b];
        }
        __syncthreads();
        uint8_t* dst = words + 9 * w0;
        for (uint32_t q = tid; 4u * q < 9u * nw; q += 256u) {                  // output dword q = symbols 4q .. 4q+3 of the step
            uint32_t s = 4u * q, w = (s * 7282u) >> 16, k = s - 9u * w;        // s / 9 for s < 4608
            uint32_t v = 0;
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e) {
                uint32_t sym = 0, p = 1;
#pragma unroll
                for (uint32_t d = 0; d < 3; ++d) {
                    const uint32_t i = 3u * k + d, g = w * uN + i;
                    const uint32_t t = i < uN ? (g < n_in ? (uint32_t)lin[g] : 0u) : fill;
                    sym += p * t; p *= 3u;
                }
                v |= (sym & 0xFFu) << (8u * e);                               // pack3 OLD:24-27 (no reduction)
                if (++k == 9u) { k = 0; ++w; }
            }
            if (fast_io && 4u * q + 4u <= 9u * nw) *(uint32_t*)(dst + 4u * q) = v;
            else for (uint32_t e = 0; e < 4u && 4u * q + e < 9u * nw; ++e) dst[4u * q + e] = (uint8_t)(v >> (8u * e));
        }
        __syncthreads();
    }
}

// One lane = four output bytes = 20 trits = five aligned dwords of the stream; lane 0 also writes the 4-byte count header.
__global__ __launch_bounds__(256) void base243_pack_kernel(const uint8_t* __restrict__ trits, uint64_t n_trits, uint8_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) { const uint32_t total = (uint32_t)n_trits; out[0] = (uint8_t)total; out[1] = (uint8_t)(total >> 8); out[2] = (uint8_t)(total >> 16); out[3] = (uint8_t)(total >> 24); }
    const uint64_t t0 = 20 * j;
    if (t0 >= n_trits) return;
    const uint64_t n_bytes = (n_trits + 4) / 5;
    if (t0 + 20 <= n_trits && (((uintptr_t)trits | (uintptr_t)out) & 3u) == 0) {
        const uint32_t* p = (const uint32_t*)(trits + t0);
        uint32_t w[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) w[i] = p[i];
        uint32_t v = 0;
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            uint32_t x = 0, pw = 1;
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) { const uint32_t b = 5u * e + q; x += pw * ((w[b >> 2] >> (8u * (b & 3u))) & 0xFFu); pw *= 3u; }
            v |= (x & 0xFFu) << (8u * e);
        }
        *(uint32_t*)(out + 4 + 4 * j) = v;
    } else {
        for (uint64_t e = 4 * j; e < min(4 * j + 4, n_bytes); ++e) {
            uint32_t x = 0, pw = 1;
            for (int q = 0; q < 5; ++q) { x += pw * (5 * e + q < n_trits ? trits[5 * e + q] : 0u); pw *= 3u; }
            out[4 + e] = (uint8_t)x;
        }
    }
}

// One lane = four input bytes -> up to 20 trits = five aligned dwords (only the first `total` trits exist)
__global__ __launch_bounds__(256) void base243_unpack_kernel(const uint8_t* __restrict__ in, uint64_t n_bytes, uint64_t total, uint8_t* __restrict__ trits) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * j >= n_bytes || 20 * j >= total) return;
    if (4 * j + 4 <= n_bytes && 20 * j + 20 <= total && (((uintptr_t)in | (uintptr_t)trits) & 3u) == 0) {
        const uint32_t v4 = *(const uint32_t*)(in + 4 * j);
        uint32_t w[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t e = 0; e < 4; ++e) {
            uint32_t v = (v4 >> (8u * e)) & 0xFFu;
#pragma unroll
            for (uint32_t q = 0; q < 5; ++q) { const uint32_t d = d3(v), b = 5u * e + q; w[b >> 2] |= (v - 3u * d) << (8u * (b & 3u)); v = d; }
        }
        uint32_t* p = (uint32_t*)(trits + 20 * j);
#pragma unroll
        for (int i = 0; i < 5; ++i) p[i] = w[i];
    } else {
        for (uint64_t e = 4 * j; e < min(4 * j + 4, n_bytes); ++e) {
            uint32_t v = in[e];
            for (int q = 0; q < 5; ++q) { const uint32_t d = v / 3u; if (5 * e + q < total) trits[5 * e + q] = (uint8_t)(v - 3u * d); v = d; }
        }
    }
}

// out[i] = in[i] % 27, 16 bytes per lane where possible
__global__ __launch_bounds__(256) void mod27_bytes_kernel(const uint8_t* __restrict__ in, uint64_t n, uint8_t* __restrict__ out) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    if (i0 + 16 <= n && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        const uint4 q = *(const uint4*)(in + i0);
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t r = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) r |= m27((w[k] >> (8 * b)) & 0xFFu) << (8 * b);
            w[k] = r;
        }
        *(uint4*)(out + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint64_t i = i0; i < min(i0 + 16, n); ++i) out[i] = (uint8_t)m27(in[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// measurement aid: the part's streaming ceiling for a kernel that reads n_read bytes and writes n_write bytes (16 bytes per lane,
// four loads in flight per lane, persistent grid-stride) -- what profiles/copy_ceiling.py quotes beside the codec kernels
// ---------------------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ void stream_copy_body(const uint4* __restrict__ src, uint64_t n_read16, uint4* __restrict__ dst, uint64_t n_write16) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = n_read16 > n_write16 ? n_read16 : n_write16;
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    for (uint64_t i = i0; i < n; i += 4 * stride) {
        v4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint64_t j = i + k * stride; v[k] = j < n_read16 ? (NT ? __builtin_nontemporal_load((const v4*)src + j) : ((const v4*)src)[j]) : v4{0, 0, 0, 0}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t j = i + k * stride;
            if (j < n_write16) { if (NT) __builtin_nontemporal_store(v[k], (v4*)dst + j); else ((v4*)dst)[j] = v[k]; }
            else if (j < n_read16 && (v[k].x ^ v[k].y) == 0x9E3779B9u) ((v4*)dst)[0] = v[k];     // reads beyond the write volume stay live
        }
    }
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint64_t n_read16, uint4* __restrict__ dst, uint64_t n_write16, int nt) {
    if (nt) stream_copy_body<true>(src, n_read16, dst, n_write16); else stream_copy_body<false>(src, n_read16, dst, n_write16);
}

// ---------------------------------------------------------------------------------------------------------
// centred window copy (io_image.hpp:125-140 blit_center_rgb, :215-235 extract_center_q): dst is a dst_rows x dst_row_bytes byte
// image; rows [dy0, dy0 + n_rows) take bytes [dx0, dx0 + row_bytes) from source rows sy0 .. (source pitch src_row_bytes, read at
// byte sx0), every other destination byte is zero.  One aligned destination dword per lane; the source bytes are gathered one by
// one (rows of 3- and 6-byte elements start at any byte).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void center_window_kernel(const uint8_t* __restrict__ src, uint64_t src_row_bytes, uint32_t sy0, uint64_t sx0,
                                                            uint8_t* __restrict__ dst, uint64_t dst_row_bytes, uint64_t dst_bytes,
                                                            uint32_t dy0, uint32_t n_rows, uint64_t dx0, uint64_t row_bytes) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4u;
    for (uint64_t b0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u; b0 < dst_bytes; b0 += stride) {
        uint64_t y = b0 / dst_row_bytes, x = b0 - y * dst_row_bytes;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            if (b0 + i < dst_bytes && y >= dy0 && y - dy0 < n_rows && x >= dx0 && x - dx0 < row_bytes)
                v |= (uint32_t)src[(uint64_t)(sy0 + (uint32_t)(y - dy0)) * src_row_bytes + sx0 + (x - dx0)] << (8u * i);
            if (++x == dst_row_bytes) { x = 0; ++y; }
        }
        if (b0 + 4 <= dst_bytes) *(uint32_t*)(dst + b0) = v;
        else for (uint32_t i = 0; b0 + i < dst_bytes; ++i) dst[b0 + i] = (uint8_t)(v >> (8u * i));
    }
}

}  // namespace t3
