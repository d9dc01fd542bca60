// t3_crc_fp4.hip — the matrix-core CRC-32 (t3_crc_mfma.hip) on the block-scaled FP4 instruction of gfx950.
//
// Same algebra: the remainder of a 64-byte chunk is a 32 x 512 bit matrix times the chunk's bits, column n of the B operand = chunk n of
// the wave's current 2 KiB, the running remainder re-enters through the "append 2048 zero bytes" matrix.  v_mfma_scale_f32_32x32x64_f8f6f4
// with FP4 (e2m1) operands runs 64 K values per instruction in the cycles the i8 instruction needs for 32: a 2 KiB round is 8 data
// instructions + 1 feedback instead of 16 + 1.  Bits travel as FP4 1.0 (0b0010) / 0.0, block scales 2^0 (e8m0 127); products are 0 or 1,
// a dot product is at most 512 + 16, exact in f32; the remainder bit is the parity of its integer value.
// A byte becomes eight FP4 values (bit i -> nibble i) by ONE read of a 256-entry table kept in 32 per-bank copies in LDS (32 KiB; the
// kernel uses no other LDS to speak of): arithmetic spreading to nibbles costs more than twice the spreading to bytes of the i8 kernel
// (nibble gaps of 3 bits make the multiply trick collide), which would leave this kernel VALU-bound at the i8 kernel's time.
// Which K slot carries which bit is free (the hardware pairs position p of lane half kh in A with the same position in B): the host builds
// the matrix slices in the order the kernel feeds bits (t3_api_decode.cpp, crc_fp4_slices).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"

namespace t3 {

typedef int v8i_ __attribute__((ext_vector_type(8)));
typedef float v16f_ __attribute__((ext_vector_type(16)));

namespace {
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
// the 16 remainder bits a lane holds (parities of its accumulators) as FP4 values in K slots 0..15 of its half: two dwords
__device__ __forceinline__ void parity_nibbles(const v16f_& acc, uint32_t& f0, uint32_t& f1) {
    f0 = 0; f1 = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        f0 |= (((uint32_t)acc[j]) & 1u) << (4 * j + 1);
        f1 |= (((uint32_t)acc[8 + j]) & 1u) << (4 * j + 1);
    }
}
}  // namespace

__global__ __launch_bounds__(256) void crc_fp4_kernel(const CrcMArgs a) {
    __shared__ uint32_t T[256 * 32];                                               // byte -> eight FP4 values, copy c in bank c
    __shared__ uint32_t red[2 * 16];
    __shared__ uint32_t zp[kCrcPows * 32];                                          // the "append 2^j zero bytes" operators of the epilogue: from global memory,
    for (uint32_t e = threadIdx.x; e < (uint32_t)kCrcPows * 32u; e += blockDim.x) zp[e] = a.zpow[e];   // one dependent load per set bit of the distance, it cost microseconds per wave
    for (uint32_t e = threadIdx.x; e < 256u * 32u; e += blockDim.x) {
        const uint32_t x = e >> 5; uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) v |= ((x >> i) & 1u) << (4 * i + 1);
        T[e] = v;
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, kh = lane >> 5, wave = threadIdx.x >> 6;
    const uint32_t wave_g = blockIdx.x * (blockDim.x >> 6) + wave;
    const uint64_t r0 = min((uint64_t)wave_g * a.rounds_per_wave, (uint64_t)a.n_rounds);     // a wave past the end runs zero rounds
    const uint64_t r1 = min(r0 + a.rounds_per_wave, (uint64_t)a.n_rounds);
    uint32_t A[9][4];
#pragma unroll
    for (int s = 0; s < 9; ++s) { const uint4 q = *(const uint4*)(a.afrag + ((size_t)s * 64u + lane) * 4u); A[s][0] = q.x; A[s][1] = q.y; A[s][2] = q.z; A[s][3] = q.w; }
    const uint8_t* p = a.data + r0 * 2048u + 64u * n + 32u * kh;
    const v16f_ zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t tb = n;                                                         // this lane's table copy (dword index: byte * 32 + copy)
    uint32_t f0 = 0, f1 = 0, sum = 0;
    // Rounds are latency-bound, not arithmetic-bound: with one round of loads in flight per wave a round took ~2,800 cycles (the memory
    // latency under load) against ~600 of arithmetic; kDepth rounds are kept in flight (8 registers each).
    constexpr uint32_t kDepth = 4;
    uint4 Q[kDepth][2];
#pragma unroll
    for (uint32_t d = 0; d < kDepth; ++d) { Q[d][0] = make_uint4(0, 0, 0, 0); Q[d][1] = Q[d][0]; if (r0 + d < r1) { Q[d][0] = *(const uint4*)(p + 2048u * d); Q[d][1] = *(const uint4*)(p + 2048u * d + 16); } }
    for (uint64_t r = r0; r < r1; r += kDepth, p += 2048u * kDepth) {
#pragma unroll
        for (uint32_t d = 0; d < kDepth; ++d) {
            if (r + d >= r1) break;
            const uint32_t w[8] = {Q[d][0].x, Q[d][0].y, Q[d][0].z, Q[d][0].w, Q[d][1].x, Q[d][1].y, Q[d][1].z, Q[d][1].w};
            if (r + d + kDepth < r1) { Q[d][0] = *(const uint4*)(p + 2048u * (d + kDepth)); Q[d][1] = *(const uint4*)(p + 2048u * (d + kDepth) + 16); }   // round r + d + kDepth, in flight from here on
#pragma unroll
            for (int i = 0; i < 8; ++i) sum = __builtin_amdgcn_sad_u8(w[i], 0u, sum);
            v16f_ acc = mfma4(A[8], f0, f1, 0u, 0u, zero);                          // running remainder, 2048 bytes further on
#pragma unroll
            for (int s = 0; s < 8; ++s) {                                           // this step: bytes 4 s .. 4 s + 3 of the lane's 32
                const uint32_t b0 = T[((w[s] & 0xFFu) << 5) + tb], b1 = T[(((w[s] >> 8) & 0xFFu) << 5) + tb];
                const uint32_t b2 = T[(((w[s] >> 16) & 0xFFu) << 5) + tb], b3 = T[((w[s] >> 24) << 5) + tb];
                acc = mfma4(A[s], b0, b1, b2, b3, acc);
            }
            parity_nibbles(acc, f0, f1);
        }
    }
    // Column n's remainder stands at the end of its last chunk, 64 (31 - n) bytes before the end of the wave's region: five masked steps
    // through the "append 64 * 2^b zero bytes" matrices (slices 9..13) bring every column to the region end
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const uint4 q = *(const uint4*)(a.afrag + ((size_t)(9 + b) * 64u + lane) * 4u);
        const uint32_t Ab[4] = {q.x, q.y, q.z, q.w};
        uint32_t m0, m1; parity_nibbles(mfma4(Ab, f0, f1, 0u, 0u, zero), m0, m1);
        if (((31u - n) >> b) & 1u) { f0 = m0; f1 = m1; }
    }
    // XOR over the columns (lanes of the same half), then this half's 16 bits -> register bits (slot j of half kh = accumulator row
    // (j & 3) + 8 (j >> 2) + 4 kh)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { f0 ^= __shfl_xor(f0, o); f1 ^= __shfl_xor(f1, o); }
    uint32_t part = 0;
#pragma unroll
    for (uint32_t j = 0; j < 16; ++j) part |= (((j < 8 ? f0 : f1) >> (4u * (j & 7u) + 1u)) & 1u) << ((j & 3u) + 8u * (j >> 2) + 4u * kh);
    part |= __shfl_xor(part, 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    uint64_t rest = a.n_bytes - r1 * 2048u;
    for (int j = 0; rest; ++j, rest >>= 1) if (rest & 1u) part = wave_apply4(zp + 32 * j, part, lane);
    if (lane == 0) { red[2 * wave] = part; red[2 * wave + 1] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t x = 0, t = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) { x ^= red[2 * w]; t += red[2 * w + 1]; }
        if (x) atomicXor(a.chunk_crc, x);
        if (t) atomicAdd(a.sym_sum, t);
    }
}

}  // namespace t3
