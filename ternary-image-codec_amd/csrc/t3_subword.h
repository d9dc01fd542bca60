// t3_subword.h — kernels of SURVEY §8 row f3 (t3_subword.hip), launched from t3_api_subword.cpp
#pragma once
#include <stdint.h>

namespace t3 {
#if defined(__HIPCC__)
__global__ void subword_extract_kernel(const uint8_t* words, uint64_t n_words, int N, uint8_t* out);
__global__ void subword_build_kernel(const uint8_t* trits, uint64_t n_trits, int N, uint32_t fill, uint8_t* words, uint64_t n_words);
__global__ void base243_pack_kernel(const uint8_t* trits, uint64_t n_trits, uint8_t* out);
__global__ void base243_unpack_kernel(const uint8_t* in, uint64_t n_bytes, uint64_t total, uint8_t* trits);
__global__ void mod27_bytes_kernel(const uint8_t* in, uint64_t n, uint8_t* out);
__global__ void stream_copy_kernel(const uint4* src, uint64_t n_read16, uint4* dst, uint64_t n_write16, int nt);
__global__ void center_window_kernel(const uint8_t* src, uint64_t src_row_bytes, uint32_t sy0, uint64_t sx0, uint8_t* dst, uint64_t dst_row_bytes,
                                     uint64_t dst_bytes, uint32_t dy0, uint32_t n_rows, uint64_t dx0, uint64_t row_bytes);
#endif
}  // namespace t3
