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

// One lane = one output dword = 4 consecutive trits of the stream; trit g is trit g % N of word g / N.  The symbol bytes
// it needs (each read by up to three trits) come through the vector L1; the stores are fully coalesced.  A launch
// covers < 2^32 trits (the launcher splits longer streams) so that one 32-bit division per lane locates the first trit.
__global__ __launch_bounds__(256) void subword_extract_kernel(const uint8_t* __restrict__ words, uint64_t n_words, int N, uint8_t* __restrict__ out) {
    const uint32_t n_trits = (uint32_t)(n_words * (uint64_t)N);
    const uint32_t g0 = 4u * (blockIdx.x * blockDim.x + threadIdx.x);
    if (g0 >= n_trits) return;
    uint32_t w = g0 / (uint32_t)N, i = g0 - w * (uint32_t)N;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q) {
        if (g0 + q < n_trits) {
            const uint32_t s3 = d3(i), r = i - 3u * s3;                       // symbol index, digit index
            const uint32_t sy = m27(words[(uint64_t)w * 9u + s3]);
            const uint32_t q1 = d3(sy), q2 = d3(q1);
            const uint32_t d = r == 0 ? sy - 3u * q1 : (r == 1 ? q1 - 3u * q2 : q2);
            v |= d << (8u * q);
        }
        if (++i == (uint32_t)N) { i = 0; ++w; }
    }
    if (g0 + 4u <= n_trits) *(uint32_t*)(out + g0) = v;
    else for (uint32_t q = 0; g0 + q < n_trits; ++q) out[g0 + q] = (uint8_t)(v >> (8u * q));
}

// One lane = one word: N trits of the stream (fewer for the last word: the rest of its first N slots is zero), slots N..26 = fill.
__global__ __launch_bounds__(256) void subword_build_kernel(const uint8_t* __restrict__ trits, uint64_t n_trits, int N, uint32_t fill, uint8_t* __restrict__ words, uint64_t n_words) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t base = w * (uint64_t)N;
    uint32_t t[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) t[i] = i < N ? (base + i < n_trits ? trits[base + i] : 0u) : fill;
#pragma unroll
    for (int s = 0; s < 9; ++s) words[w * 9 + s] = (uint8_t)(t[3 * s] + 3u * t[3 * s + 1] + 9u * t[3 * s + 2]);   // pack3 OLD:24-27 (no reduction)
}

// One lane = one output byte = 5 trits; lane 0 also writes the 4-byte count header.
__global__ __launch_bounds__(256) void base243_pack_kernel(const uint8_t* __restrict__ trits, uint64_t n_trits, uint8_t* __restrict__ out) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) { const uint32_t total = (uint32_t)n_trits; out[0] = (uint8_t)total; out[1] = (uint8_t)(total >> 8); out[2] = (uint8_t)(total >> 16); out[3] = (uint8_t)(total >> 24); }
    if (5 * j >= n_trits) return;
    uint32_t v = 0, p = 1;
#pragma unroll
    for (int q = 0; q < 5; ++q) { v += p * (5 * j + q < n_trits ? trits[5 * j + q] : 0u); p *= 3u; }
    out[4 + j] = (uint8_t)v;
}

// One lane = one input byte -> up to 5 trits (only the first `total` trits exist)
__global__ __launch_bounds__(256) void base243_unpack_kernel(const uint8_t* __restrict__ in, uint64_t n_bytes, uint64_t total, uint8_t* __restrict__ trits) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_bytes || 5 * j >= total) return;
    uint32_t v = in[j];
#pragma unroll
    for (int q = 0; q < 5; ++q) { const uint32_t d = v / 3u; if (5 * j + q < total) trits[5 * j + q] = (uint8_t)(v - 3u * d); v = d; }
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

}  // namespace t3
