// t3_rgb.h — SURVEY §8 row f1: RGB8 <-> quantised YCbCr bridge (t3_rgb.hip, t3_api_rgb.cpp)
#pragma once
#include <stdint.h>

namespace t3 {
// The reference's double-precision quantisers (old/include/io_image.hpp:69-84), tabulated on the host with its expressions
struct QuantTables {
    uint16_t yq[256];   // Y  -> clamp(lround(Y * (242.0/255.0)), 0, 242)
    int8_t   cq[256];   // C  -> clamp(lround((C - 128) * (40.0/128.0)), -40, 40)
    uint8_t  yd[244];   // Yq -> clamp(lround(Yq * (255.0/242.0)), 0, 255), Yq <= 242 (243 entries + pad)
    uint8_t  cd[84];    // Cq + 40 -> clamp(lround(128 + Cq * (128.0/40.0)), 0, 255), |Cq| <= 40 (81 entries + pad)
};
static_assert(sizeof(QuantTables) % 4 == 0, "copied by dwords");
#if defined(__HIPCC__)
__global__ void rgb_to_quant_kernel(const uint8_t* rgb, uint64_t n_px, uint16_t* px, const QuantTables* tab);
__global__ void quant_to_rgb_kernel(const uint16_t* px, uint64_t n_px, uint8_t* rgb, const QuantTables* tab);
#endif
}  // namespace t3
