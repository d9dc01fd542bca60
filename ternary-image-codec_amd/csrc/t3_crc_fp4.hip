// t3_crc_fp4.hip — the matrix-core CRC-32 (t3_crc_mfma.hip) on the block-scaled FP4 instruction of gfx950.
//
// Same algebra: the remainder of a 64-byte chunk is a 32 x 512 bit matrix times the chunk's bits, column n of the B operand = chunk n of
// the wave's current 2 KiB, the running remainder re-enters through an "append zero bytes" matrix.  v_mfma_scale_f32_32x32x64_f8f6f4
// with FP4 (e2m1) operands runs 64 K values per instruction in the cycles the i8 instruction needs for 32: a 2 KiB round is 8 data
// instructions + 1 feedback instead of 16 + 1.  Block scales 2^0 (e8m0 127); every product is 0 or exactly 1, a dot product is at most
// 512 + 16, exact in f32; the remainder bit is the parity of its integer value.
// Round 3: no table, no LDS in the loop.  An input dword becomes its four B dwords with five vector instructions: w & 0x11111111,
// w & 0x22222222, w & 0x44444444 and (w >> 1) & 0x44444444 -- a bit stays where it is inside its nibble and therefore reads as FP4 0.5, 1.0
// or 2.0; the matrix slice holds the reciprocal weight (2.0, 1.0, 0.5) in that K slot, so the product is 1.  (Round 2 read a 256-entry
// byte -> eight-nibble table, 32 bank copies: two vector instructions + one LDS read per BYTE, and the compiler's schedule put the LDS
// latency of every step in series with its matrix instruction: ~1,100 cycles per round and SIMD.)  Which K slot carries which bit is free
// (the hardware pairs position p of lane half kh in A with the same position in B): the host builds the slices in the order the
// kernel feeds bits (t3_api_decode.cpp, decode_init).
// Parity without a conversion: x = acc + 2^(23 - s) has the integer's bit 0 at mantissa bit s (the sum is exact: acc < 2^11); with
// s = 0, 4, 8, 12 for the four accumulators of a group the bits land in nibbles 0..3 of one dword at nibble bit 0 (FP4 0.5, weight 2.0 in
// the feedback slice): a packed add per two accumulators and one v_and_or per accumulator.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"

#ifndef T3_CRC_DEPTH
#define T3_CRC_DEPTH 4
#endif
#ifndef T3_CRC_NT
#define T3_CRC_NT 0
#endif

namespace t3 {

typedef int v8i_ __attribute__((ext_vector_type(8)));
typedef float v16f_ __attribute__((ext_vector_type(16)));

namespace {
typedef float v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld16(const uint8_t* q) {
#if T3_CRC_NT
    typedef uint32_t u4_ __attribute__((ext_vector_type(4)));
    const u4_ v = __builtin_nontemporal_load((const u4_*)q); return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *(const uint4*)q;
#endif
}
__device__ __forceinline__ uint32_t wave_apply4(const uint32_t* __restrict__ op, uint32_t x, uint32_t lane) {
    uint32_t v = (lane < 32u && ((x >> lane) & 1u)) ? op[lane] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ v16f_ mfma4(const uint32_t (&A)[4], const uint32_t b0, const uint32_t b1, const uint32_t b2, const uint32_t b3, const v16f_ acc) {
    const v8i_ a = {(int)A[0], (int)A[1], (int)A[2], (int)A[3], 0, 0, 0, 0}, b = {(int)b0, (int)b1, (int)b2, (int)b3, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);     // cbsz = blgp = 4: FP4; scales 2^0
}
// the 16 remainder bits a lane holds (parities of its accumulators) as FP4 0.5 in K slots 8 g + q of its half: accumulator 4 g + q ->
// dword g, nibble q
__device__ __forceinline__ void parity_nibbles(const v16f_& acc, uint32_t (&f)[4]) {
    const v2f_ m01 = {8388608.0f, 524288.0f}, m23 = {32768.0f, 2048.0f};           // 2^23, 2^19, 2^15, 2^11
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v2f_ lo = v2f_{acc[4 * g], acc[4 * g + 1]} + m01, hi = v2f_{acc[4 * g + 2], acc[4 * g + 3]} + m23;
        f[g] = (__float_as_uint(lo[0]) & 0x1u) | (__float_as_uint(lo[1]) & 0x10u) | (__float_as_uint(hi[0]) & 0x100u) | (__float_as_uint(hi[1]) & 0x1000u);
    }
}
}  // namespace

__global__ __launch_bounds__(256) void crc_fp4_kernel(const CrcMArgs a) {
    __shared__ uint32_t red[2 * 16];
    __shared__ uint32_t zp[kCrcPows * 32];                                          // the "append 2^j zero bytes" operators of the epilogue: from global memory,
    for (uint32_t e = threadIdx.x; e < (uint32_t)kCrcPows * 32u; e += blockDim.x) zp[e] = a.zpow[e];   // one dependent load per set bit of the distance, it cost microseconds per wave
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, kh = lane >> 5, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t wave_g = blockIdx.x * (blockDim.x >> 6) + wave;
    // Which rounds a wave owns.  Strided (stride_waves = W > 0): wave g takes rounds g, g + W, g + 2 W ... -- at any moment the chip reads one
    // moving window of W x kDepth x 2 KiB, the way a streaming copy does; a wave on rounds_per_wave consecutive rounds of its own makes
    // the chip read at 2048 places 90 KB apart, 2 KiB at a time (3.4 TB/s against the 4.5+ a read-only stream reaches).  A column's next
    // chunk is then 2048 W bytes further on: the feedback slice is the host-built "append 2048 W zero bytes" operator (a.afb).
    const uint32_t W = a.stride_waves;
    uint64_t r0, r1, step;                                                          // rounds r0, r0 + step, ... < r1
    if (W) { r0 = wave_g; r1 = a.n_rounds; step = W; }
    else { r0 = min((uint64_t)wave_g * a.rounds_per_wave, (uint64_t)a.n_rounds); r1 = min(r0 + a.rounds_per_wave, (uint64_t)a.n_rounds); step = 1; }   // a wave past the end runs zero rounds
    uint32_t A[9][4];
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const uint4 q = (s == 8 && W) ? *(const uint4*)(a.afb + (size_t)lane * 4u) : *(const uint4*)(a.afrag + ((size_t)s * 64u + lane) * 4u);
        A[s][0] = q.x; A[s][1] = q.y; A[s][2] = q.z; A[s][3] = q.w;
    }
    const uint64_t pstep = 2048u * step;
    const uint8_t* p = a.data + r0 * 2048u + 64u * n + 32u * kh;
    const v16f_ zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t f[4] = {0, 0, 0, 0}, sum = 0;
    // Rounds are latency-bound, not arithmetic-bound: with one round of loads in flight per wave a round took ~2,800 cycles (the memory
    // latency under load) against ~600 of arithmetic; kDepth rounds are kept in flight (8 registers each).
    constexpr uint32_t kDepth = T3_CRC_DEPTH;
    uint4 Q[kDepth][2];
    uint64_t r_end = r0;                                                           // one past the wave's last round (in rounds), for the distance to the stream's end
#pragma unroll
    for (uint32_t d = 0; d < kDepth; ++d) { Q[d][0] = make_uint4(0, 0, 0, 0); Q[d][1] = Q[d][0]; if (r0 + d * step < r1) { Q[d][0] = ld16(p + pstep * d); Q[d][1] = ld16(p + pstep * d + 16); } }
    for (uint64_t r = r0; r < r1; r += kDepth * step, p += pstep * kDepth) {
#pragma unroll
        for (uint32_t d = 0; d < kDepth; ++d) {
            if (r + d * step >= r1) break;
            r_end = r + d * step + 1u;
            const uint32_t w[8] = {Q[d][0].x, Q[d][0].y, Q[d][0].z, Q[d][0].w, Q[d][1].x, Q[d][1].y, Q[d][1].z, Q[d][1].w};
            if (r + (d + kDepth) * step < r1) { Q[d][0] = ld16(p + pstep * (d + kDepth)); Q[d][1] = ld16(p + pstep * (d + kDepth) + 16); }   // kDepth rounds ahead, in flight from here on
#pragma unroll
            for (int i = 0; i < 8; ++i) sum = __builtin_amdgcn_sad_u8(w[i], 0u, sum);
            v16f_ acc = mfma4(A[8], f[0], f[1], f[2], f[3], zero);                  // running remainder, one round step further on
#pragma unroll
            for (int s = 0; s < 8; ++s)                                             // this step: bytes 4 s .. 4 s + 3 of the lane's 32
                acc = mfma4(A[s], w[s] & 0x11111111u, w[s] & 0x22222222u, w[s] & 0x44444444u, (w[s] >> 1) & 0x44444444u, acc);
            parity_nibbles(acc, f);
        }
    }
    // Column n's remainder stands at the end of its last chunk, 64 (31 - n) bytes before the end of the wave's region: five masked steps
    // through the "append 64 * 2^b zero bytes" matrices (slices 9..13) bring every column to the region end
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const uint4 q = *(const uint4*)(a.afrag + ((size_t)(9 + b) * 64u + lane) * 4u);
        const uint32_t Ab[4] = {q.x, q.y, q.z, q.w};
        uint32_t m[4]; parity_nibbles(mfma4(Ab, f[0], f[1], f[2], f[3], zero), m);
        if (((31u - n) >> b) & 1u) { f[0] = m[0]; f[1] = m[1]; f[2] = m[2]; f[3] = m[3]; }
    }
    // XOR over the columns (lanes of the same half), then this half's 16 bits -> register bits (accumulator e = 4 g + q of half kh = row
    // (e & 3) + 8 (e >> 2) + 4 kh)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) f[g] ^= __shfl_xor(f[g], o);
    }
    uint32_t part = 0;
#pragma unroll
    for (uint32_t e = 0; e < 16; ++e) part |= ((f[e >> 2] >> (4u * (e & 3u))) & 1u) << ((e & 3u) + 8u * (e >> 2) + 4u * kh);
    part |= __shfl_xor(part, 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    uint64_t rest = W ? (r_end > r0 ? a.n_bytes - r_end * 2048u : 0u) : a.n_bytes - r1 * 2048u;   // (a wave without rounds carries part = 0)
    for (int j = 0; rest; ++j, rest >>= 1) if (rest & 1u) part = wave_apply4(zp + 32 * j, part, lane);
    if (lane == 0) { red[2 * wave] = part; red[2 * wave + 1] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x = 0, t = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { x ^= red[2 * w]; t += red[2 * w + 1]; }
        if (a.partials) { a.partials[2u * blockIdx.x] = x; a.partials[2u * blockIdx.x + 1u] = t; }
        else { if (x) atomicXor(a.chunk_crc, x); if (t) atomicAdd(a.sym_sum, t); }
    }
}

}  // namespace t3
