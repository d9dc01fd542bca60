// t3_api_rgb.cpp — C-ABI of SURVEY §8 row f1 (RGB8 <-> quantised YCbCr bridge, old/include/io_image.hpp:47-90,156-195)
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "../../include/t3hip.h"
#include "t3_rgb.h"

namespace t3 {
int api_ready(); hipStream_t api_stream(); int api_scratch(int slot, size_t bytes, void** out, hipStream_t s = nullptr); std::recursive_mutex& api_host_mutex(); std::mutex& api_qt_mutex();
int api_fail_hip(hipError_t e, const char* what);
void*& api_slot(int id);
int api_encode_rgb_fused(const void* d_rgb, uint64_t n_px, const t3_cfg* cfg, void* d_out, uint64_t cap, uint64_t* n_out, hipStream_t s);
}  // namespace t3
using namespace t3;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return api_fail_hip(e_, #x); } while (0)

namespace {
#define d_qt (*(QuantTables**)&api_slot(32))     // per-context slot (t3_api.cpp)
#define g_qt_mu api_qt_mutex()                   // per context, like the table it guards
int tables(const QuantTables** out) {
    std::lock_guard<std::mutex> lk(g_qt_mu);
    if (!d_qt) {
        QuantTables t; memset(&t, 0, sizeof t);
        auto cl = [](long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); };
        for (int Y = 0; Y < 256; ++Y) t.yq[Y] = (uint16_t)cl(lround(Y * (242.0 / 255.0)), 0, 242);                  // io_image.hpp:72
        for (int c = 0; c < 256; ++c) t.cq[c] = (int8_t)cl(lround((c - 128) * (40.0 / 128.0)), -40, 40);           // :73-76
        for (int q = 0; q <= 242; ++q) t.yd[q] = (uint8_t)cl(lround(q * (255.0 / 242.0)), 0, 255);                  // :81
        for (int q = -40; q <= 40; ++q) t.cd[q + 40] = (uint8_t)cl(lround(128 + q * (128.0 / 40.0)), 0, 255);       // :82-83
        HIPCHK(hipMalloc((void**)&d_qt, sizeof t)); HIPCHK(hipMemcpy(d_qt, &t, sizeof t, hipMemcpyHostToDevice));
    }
    *out = d_qt; return T3_OK;
}
unsigned blocks_for(uint64_t items) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, (items + 255) / 256), 1u << 30); }
}  // namespace

namespace t3 { void rgb_shutdown() { std::lock_guard<std::mutex> lk(g_qt_mu); if (d_qt) { (void)hipFree(d_qt); d_qt = nullptr; } } }

extern "C" {

int t3hip_rgb_to_quant_dev(const uint8_t* d_rgb, uint64_t n_px, void* d_px6, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n_px) return T3_OK;
    if (!d_rgb || !d_px6) return T3_E_ARG;
    const QuantTables* t; int rc = tables(&t); if (rc) return rc;
    hipLaunchKernelGGL(rgb_to_quant_kernel, dim3(blocks_for((n_px + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_rgb, n_px, (uint16_t*)d_px6, t);
    HIPCHK(hipGetLastError()); return T3_OK;
}
int t3hip_quant_to_rgb_dev(const void* d_px6, uint64_t n_px, uint8_t* d_rgb, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n_px) return T3_OK;
    if (!d_rgb || !d_px6) return T3_E_ARG;
    const QuantTables* t; int rc = tables(&t); if (rc) return rc;
    hipLaunchKernelGGL(quant_to_rgb_kernel, dim3(blocks_for((n_px + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_px6, n_px, d_rgb, t);
    HIPCHK(hipGetLastError()); return T3_OK;
}
// slot 4 of the stream's scratch set holds the quantised pixels between the two launches (slots 2 and 3 belong to the codec)
int t3hip_encode_rgb_dev(const uint8_t* d_rgb, uint64_t n_px, const t3_cfg* cfg, void* d_out, uint64_t cap_words, uint64_t* n_out, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!cfg || !n_out || (n_px && !d_rgb)) return T3_E_ARG;
    // one launch: the bridge runs inside the encoder's phase 1 (3 bytes per pixel read, no intermediate); the framings that
    // kernel does not take (RAW mode, 2-D rows wider than 512) go through the bridge kernel and a per-stream scratch
    int rc = api_encode_rgb_fused(d_rgb, n_px, cfg, d_out, cap_words, n_out, (hipStream_t)stream);
    if (rc != 1) return rc;
    void* d_q; rc = api_scratch(4, 6 * n_px + 64, &d_q, (hipStream_t)stream); if (rc) return rc;
    rc = t3hip_rgb_to_quant_dev(d_rgb, n_px, d_q, stream); if (rc) return rc;
    return t3hip_encode_frame_dev(d_q, n_px, cfg, d_out, cap_words, n_out, stream);
}
int t3hip_decode_rgb_async(const void* d_in, uint64_t n_in, const t3_cfg* cfg, uint64_t n_px, uint8_t* d_rgb, uint32_t* d_verdict, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!cfg || !d_verdict || (n_px && !d_rgb)) return T3_E_ARG;
    const uint64_t n_raw = (n_px + 1) / 2;
    uint64_t n_units = 0;
    // one launch where the fused pixel decoder applies (FIXED, one k, 1-D): its output stage converts to RGB and stores 3 bytes per pixel
    int rc = t3hip_decode_frame_async(d_in, n_in, cfg, n_raw, d_rgb, n_px, &n_units, 2, d_verdict, stream);
    if (rc != 1) return rc;
    void* d_q; rc = api_scratch(4, 12 * n_raw + 64, &d_q, (hipStream_t)stream); if (rc) return rc;
    rc = t3hip_decode_frame_async(d_in, n_in, cfg, n_raw, d_q, 2 * n_raw, &n_units, 1, d_verdict, stream); if (rc) return rc;
    return t3hip_quant_to_rgb_dev(d_q, n_px, d_rgb, stream);
}
int t3hip_rgb_to_quant(const uint8_t* rgb, uint64_t n_px, void* px6) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n_px) return T3_OK;
    if (!rgb || !px6) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    void *di, *dout; int rc = api_scratch(0, 3 * n_px + 64, &di); if (rc) return rc;
    rc = api_scratch(1, 6 * n_px + 64, &dout); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(di, rgb, 3 * n_px, hipMemcpyHostToDevice, api_stream()));
    rc = t3hip_rgb_to_quant_dev((const uint8_t*)di, n_px, dout, api_stream()); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(px6, dout, 6 * n_px, hipMemcpyDeviceToHost, api_stream()));
    HIPCHK(hipStreamSynchronize(api_stream())); return T3_OK;
}
int t3hip_quant_to_rgb(const void* px6, uint64_t n_px, uint8_t* rgb) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n_px) return T3_OK;
    if (!rgb || !px6) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    void *di, *dout; int rc = api_scratch(0, 6 * n_px + 64, &di); if (rc) return rc;
    rc = api_scratch(1, 3 * n_px + 64, &dout); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(di, px6, 6 * n_px, hipMemcpyHostToDevice, api_stream()));
    rc = t3hip_quant_to_rgb_dev(di, n_px, (uint8_t*)dout, api_stream()); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(rgb, dout, 3 * n_px, hipMemcpyDeviceToHost, api_stream()));
    HIPCHK(hipStreamSynchronize(api_stream())); return T3_OK;
}

}  // extern "C"
