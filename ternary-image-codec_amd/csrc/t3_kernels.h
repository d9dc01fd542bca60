// t3_kernels.h — kernel declarations for the launcher (hipcc only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "t3_device.h"
#include "t3_rs_core.h"

namespace t3 {

__global__ void pack_pixels_kernel(const uint16_t* px, uint64_t n_px, uint8_t* words, uint64_t n_words);
__global__ void unpack_words_kernel(const uint8_t* words, uint64_t n_words, uint16_t* px);
template <int FE, int IL, int RSEL, bool BCN> __global__ void encode_kernel_k(const EncArgs a);   // every band shares k = 26-RSEL; BCN: beacon fused into the stores
template <int FE, int IL> __global__ void encode_kernel_mixed(const EncArgs a);         // mixed k (UEP), LUT path
template <int FE, int IL, bool BCN> __global__ void encode_kernel_uep(const EncArgs a);           // mixed k (UEP), matrix cores: bands grouped by k
__global__ void beacon_kernel(const BeaconArgs a);
__global__ void interleave_kernel(const uint8_t* in, uint8_t* out, uint32_t n, uint32_t w, uint32_t A, DevDiv div_A, DevDiv div_w);   // OLD:750-813, standalone
__global__ void rs_encode_blocks_kernel(const uint8_t* data, uint64_t n_blocks, int k, const uint8_t* P, const RsTables* tab, uint8_t* code);

}  // namespace t3
