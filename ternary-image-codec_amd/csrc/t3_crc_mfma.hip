// t3_crc_mfma.hip — CRC-32 of the coded payload on the matrix cores (crc32_acc, src/io_t3p_t3v.cpp:20-36; the T3V frame
// index and the payload CRC of the containers).
//
// The CRC register update is GF(2)-linear, so the remainder of a 64-byte chunk is a 32 x 512 bit-matrix times the chunk's
// bits.  As an integer product that is exact: 0/1 bytes in, dot products <= 544, parity = bit 0.  One
// v_mfma_i32_32x32x32_i8 multiplies a 32 x 32 slice of that matrix (A, host-built, 17 slices in 68 VGPRs) with 32 bits of
// 32 different chunks (B: column n = chunk n of the wave's current 2 KiB), so a wave reads 2 KiB per round perfectly
// coalesced (lane (n, h) takes bytes 64 n + 32 h .. + 32), expands them to bit-bytes in registers (three VALU operations
// per dword: nibble * 0x00204081 & 0x01010101) and issues 16 data MFMAs + 1 feedback MFMA: the column's running
// remainder re-enters as 32 more K inputs through the "append 2048 zero bytes" matrix (its chunks are 2 KiB apart).
// No LDS, no table lookups: the kernel runs beside the LDS-bound decoder without competing for its bottleneck; the
// table kernel (crc_chunks_kernel: one 2304-byte chunk per lane, 64 cache lines per load instruction) stays for the tail
// and for unaligned buffers.  After the last round a column's remainder moves to the end of the stream: 64 (31 - n)
// bytes per column (five masked MFMA steps), then, XOR-reduced over the columns, the common distance with one operator
// column per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t3hip.h"
#include "t3_decode.h"

namespace t3 {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

namespace {
// x through a 32-column GF(2) operator, one column per lane (x wave-uniform); every lane gets the result
__device__ __forceinline__ uint32_t wave_apply(const uint32_t* __restrict__ op, uint32_t x, uint32_t lane) {
    uint32_t v = (lane < 32u && ((x >> lane) & 1u)) ? op[lane] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
}  // namespace

// feedback operand from the accumulators: dword q, byte j <- acc[4q + j]; only bit 0 of a byte matters (see below)
__device__ __forceinline__ v4i parity_bytes(const v16i& acc) {
    v4i fb;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t lo = __builtin_amdgcn_perm((uint32_t)acc[4 * q + 1], (uint32_t)acc[4 * q], 0x0c0c0400u);       // bytes: a0.b0, a1.b0, 0, 0
        const uint32_t hi = __builtin_amdgcn_perm((uint32_t)acc[4 * q + 3], (uint32_t)acc[4 * q + 2], 0x04000c0cu);   // bytes: 0, 0, a2.b0, a3.b0
        fb[q] = (int)(lo | hi);
    }
    return fb;
}

#define T3_CRC_LD(p) (*(const uint4*)(p))      // (non-temporal loads measured slower here: 0.074 vs 0.070 ms per 187 MB frame)
__global__ __launch_bounds__(256) void crc_mfma_kernel(const CrcMArgs a) {
    __shared__ uint32_t red[2 * 16];
    const uint32_t lane = threadIdx.x & 63u, n = lane & 31u, kh = lane >> 5, wave = threadIdx.x >> 6;
    const uint32_t wave_g = blockIdx.x * (blockDim.x >> 6) + wave;
    const uint64_t r0 = min((uint64_t)wave_g * a.rounds_per_wave, (uint64_t)a.n_rounds);     // a wave past the end runs zero rounds
    const uint64_t r1 = min(r0 + a.rounds_per_wave, (uint64_t)a.n_rounds);
    v4i A[17];
#pragma unroll
    for (int s = 0; s < 17; ++s) A[s] = *(const v4i*)(a.afrag + ((size_t)s * 64u + lane) * 4u);
    const uint8_t* p = a.data + r0 * 2048u + 64u * n + 32u * kh;
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // Only the parity of a dot product is used, so only bit 0 of an operand byte matters: the other bits add an even number
    // (|sum| stays far below 2^31).  That saves the masks: nibble * 0x00204081 has bit t of the nibble as bit 0 of byte t.
    v4i fb = {0, 0, 0, 0};
    uint32_t sum = 0;
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (r0 < r1) { q0 = T3_CRC_LD(p); q1 = T3_CRC_LD(p + 16); }
    for (uint64_t r = r0; r < r1; ++r) {
        const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        p += 2048;
        if (r + 1 < r1) { q0 = T3_CRC_LD(p); q1 = T3_CRC_LD(p + 16); }          // next round's bytes, in flight under this round's arithmetic
#pragma unroll
        for (int i = 0; i < 8; ++i) sum = __builtin_amdgcn_sad_u8(w[i], 0u, sum);
        v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[16], fb, zero, 0, 0, 0);     // running remainder, 2048 bytes further on
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t h = w[s >> 1] >> (16 * (s & 1));                               // this step's 16 bits
            v4i B;
#pragma unroll
            for (int d = 0; d < 4; ++d) B[d] = (int)__umul24((h >> (4 * d)) & 15u, 0x00204081u);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s], B, acc, 0, 0, 0);
        }
        fb = parity_bytes(acc);
    }
    // Column n's remainder stands at the end of its last chunk, 64 (31 - n) bytes before the end of the wave's region: five
    // masked steps through the "append 64 * 2^b zero bytes" matrices (slices 17..21) bring every column to the region end
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        const v4i Ab = *(const v4i*)(a.afrag + ((size_t)(17 + b) * 64u + lane) * 4u);
        const v4i mv = parity_bytes(__builtin_amdgcn_mfma_i32_32x32x32_i8(Ab, fb, zero, 0, 0, 0));
        if (((31u - n) >> b) & 1u) fb = mv;
    }
    // XOR over the columns (lanes of the same half), then the 16 bits of this half -> register bits 8q + 4kh + j
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) fb[q] ^= __shfl_xor(fb[q], o);
    uint32_t part = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) part |= (((uint32_t)fb[q] >> (8 * j)) & 1u) << (8 * q + 4 * kh + j);
    part |= __shfl_xor(part, 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    // ... and from there to the end of the stream, the same distance for every lane
    uint64_t rest = a.n_bytes - r1 * 2048u;
    for (int j = 0; rest; ++j, rest >>= 1) if (rest & 1u) part = wave_apply(a.zpow + 32 * j, part, lane);
    // one pair of atomics per workgroup: the accumulators are single addresses (about 11 ns per atomic on one address)
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
