// t3_rgb.hip — SURVEY §8 row f1, first version: the RGB8 <-> quantised YCbCr bridge as its own kernels (not yet fused into
// the encoder's phase 1).  rgb_to_quant_stream / quant_stream_to_rgb, old/include/io_image.hpp:47-90,156-195.
//
// Exactness: the reference evaluates float expressions left to right, every product and sum rounded to float on its own
// (x86-64 SSE, no fused multiply-add), then std::lround (half away from zero), then quantises in double.  Here the float
// steps are __fmul_rn/__fadd_rn/__fsub_rn (this file is also built with -ffp-contract=off), lround is trunc + an exact
// fraction test, and the double-precision quantisers are 256/256/243/81-entry tables computed on the host with the
// reference's own expressions (QuantTables, t3_api_rgb.cpp).
// PARITY UNPINNED against a reference build: neither version of io_image.hpp compiles (ImageU8 has no member `swap`,
// old/include/io_image.hpp:218); parity is against oracle/t3_oracle.c's restatement, exhaustively over all 2^24 RGB values.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t3_rgb.h"

namespace t3 {

namespace {
__device__ __forceinline__ int lround_f(float x) {                 // std::lround: nearest, ties away from zero
    const float t = truncf(x), f = __fsub_rn(x, t);               // exact
    return (int)t + (fabsf(f) >= 0.5f ? (x < 0.0f ? -1 : 1) : 0);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
}  // namespace

// One lane = 4 pixels: 12 bytes in (three aligned dwords) -> 24 bytes out (three aligned 8-byte stores).
__global__ __launch_bounds__(256) void rgb_to_quant_kernel(const uint8_t* __restrict__ rgb, uint64_t n_px, uint16_t* __restrict__ px, const QuantTables* __restrict__ tab) {
    __shared__ QuantTables T;
    for (uint32_t i = threadIdx.x; i < sizeof(QuantTables) / 4; i += blockDim.x) ((uint32_t*)&T)[i] = ((const uint32_t*)tab)[i];
    __syncthreads();
    const uint64_t p0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (p0 >= n_px) return;
    uint8_t in[12];
    const bool whole = p0 + 4 <= n_px && (((uintptr_t)rgb | (uintptr_t)px) & 7u) == 0;
    if (whole) { const uint32_t* s = (const uint32_t*)(rgb + 3 * p0); *(uint32_t*)(in) = s[0]; *(uint32_t*)(in + 4) = s[1]; *(uint32_t*)(in + 8) = s[2]; }
    else for (uint32_t i = 0; i < 12; ++i) in[i] = 3 * p0 + i < 3 * n_px ? rgb[3 * p0 + i] : 0;
    uint16_t o[12];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float r = (float)in[3 * q], g = (float)in[3 * q + 1], b = (float)in[3 * q + 2];
        const float y = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));                                   // io_image.hpp:50
        const float cb = __fadd_rn(__fadd_rn(__fsub_rn(__fmul_rn(-0.168736f, r), __fmul_rn(0.331264f, g)), __fmul_rn(0.5f, b)), 128.0f);          // :51
        const float cr = __fadd_rn(__fsub_rn(__fsub_rn(__fmul_rn(0.5f, r), __fmul_rn(0.418688f, g)), __fmul_rn(0.081312f, b)), 128.0f);           // :52
        const int Y = clampi(lround_f(y), 0, 255), Cb = clampi(lround_f(cb), 0, 255), Cr = clampi(lround_f(cr), 0, 255);
        o[3 * q] = T.yq[Y]; o[3 * q + 1] = (uint16_t)(int16_t)T.cq[Cb]; o[3 * q + 2] = (uint16_t)(int16_t)T.cq[Cr];                               // quantize_ycbcr :69-78
    }
    uint16_t* d = px + 3 * p0;
    if (whole) {
#pragma unroll
        for (int k = 0; k < 3; ++k) *(uint2*)(d + 4 * k) = make_uint2((uint32_t)o[4 * k] | (uint32_t)o[4 * k + 1] << 16, (uint32_t)o[4 * k + 2] | (uint32_t)o[4 * k + 3] << 16);
    } else for (uint32_t i = 0; i < 12 && 3 * p0 + i < 3 * n_px; ++i) d[i] = o[i];
}

// One lane = 4 pixels: 24 bytes in -> 12 bytes out.
__global__ __launch_bounds__(256) void quant_to_rgb_kernel(const uint16_t* __restrict__ px, uint64_t n_px, uint8_t* __restrict__ rgb, const QuantTables* __restrict__ tab) {
    __shared__ QuantTables T;
    for (uint32_t i = threadIdx.x; i < sizeof(QuantTables) / 4; i += blockDim.x) ((uint32_t*)&T)[i] = ((const uint32_t*)tab)[i];
    __syncthreads();
    const uint64_t p0 = 4 * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (p0 >= n_px) return;
    uint8_t o[12];
    uint16_t in[12];
    if (p0 + 4 <= n_px && ((uintptr_t)px & 7u) == 0) {                       // three aligned 8-byte loads
#pragma unroll
        for (int k = 0; k < 3; ++k) { const uint2 v = *(const uint2*)(px + 3 * p0 + 4 * k); in[4 * k] = (uint16_t)v.x; in[4 * k + 1] = (uint16_t)(v.x >> 16); in[4 * k + 2] = (uint16_t)v.y; in[4 * k + 3] = (uint16_t)(v.y >> 16); }
    } else for (uint32_t i = 0; i < 12; ++i) in[i] = 3 * p0 + i < 3 * n_px ? px[3 * p0 + i] : (uint16_t)0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (p0 + q >= n_px) { o[3 * q] = o[3 * q + 1] = o[3 * q + 2] = 0; continue; }
        const uint32_t Yq = in[3 * q]; const int Cbq = (int16_t)in[3 * q + 1], Crq = (int16_t)in[3 * q + 2];
        // dequantize_ycbcr :79-84 (tables cover the in-range values; beyond them the clamps decide)
        const int Y = Yq >= 242u ? 255 : T.yd[Yq];
        const int Cb = Cbq <= -40 ? 0 : (Cbq >= 40 ? 255 : T.cd[Cbq + 40]), Cr = Crq <= -40 ? 0 : (Crq >= 40 ? 255 : T.cd[Crq + 40]);
        const float y = (float)Y, cb = __fsub_rn((float)Cb, 128.0f), cr = __fsub_rn((float)Cr, 128.0f);                                           // ycbcr_to_rgb :57-66
        const float r = __fadd_rn(y, __fmul_rn(1.402f, cr));
        const float g = __fsub_rn(__fsub_rn(y, __fmul_rn(0.344136f, cb)), __fmul_rn(0.714136f, cr));
        const float b = __fadd_rn(y, __fmul_rn(1.772f, cb));
        o[3 * q] = (uint8_t)clampi(lround_f(r), 0, 255); o[3 * q + 1] = (uint8_t)clampi(lround_f(g), 0, 255); o[3 * q + 2] = (uint8_t)clampi(lround_f(b), 0, 255);
    }
    uint8_t* d = rgb + 3 * p0;
    if (p0 + 4 <= n_px && ((uintptr_t)rgb & 3u) == 0) {
        uint32_t* w = (uint32_t*)d;
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = (uint32_t)o[4 * k] | (uint32_t)o[4 * k + 1] << 8 | (uint32_t)o[4 * k + 2] << 16 | (uint32_t)o[4 * k + 3] << 24;
    } else for (uint32_t i = 0; i < 12 && 3 * p0 + i < 3 * n_px; ++i) d[i] = o[i];
}

}  // namespace t3
