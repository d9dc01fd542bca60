// t3_api_subword.cpp — C-ABI of SURVEY §8 row f3: subword trit streams (OLD:834-859) and the wire packings of
// include/ternary_packing.hpp (TPACK:18-65).  Device entry points launch on the caller's stream; host entry points stage
// through the library's scratch buffers like the rest of the std::vector-facing API.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>

#include <algorithm>

#include "../../include/t3hip.h"
#include "t3_subword.h"

namespace t3 {
int api_ready(); hipStream_t api_stream(); int api_scratch(int slot, size_t bytes, void** out, hipStream_t s = nullptr); std::recursive_mutex& api_host_mutex();
int api_fail_hip(hipError_t e, const char* what);
}  // namespace t3
using namespace t3;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return api_fail_hip(e_, #x); } while (0)

namespace {
unsigned blocks_for(uint64_t items) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, (items + 255) / 256), 1u << 30); }
bool valid_n(int N) { return N >= 1 && N <= 27; }
}  // namespace

extern "C" {

uint64_t t3hip_subword_words(uint64_t n_trits, int N) { return N >= 1 ? (n_trits + (uint64_t)N - 1) / (uint64_t)N : 0; }
uint64_t t3hip_base243_bytes(uint64_t n_trits) { return 4 + (n_trits + 4) / 5; }

int t3hip_subword_extract_dev(const void* d_words, uint64_t n_words, int N, uint8_t* d_trits, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!valid_n(N)) return T3_E_ARG;
    if (!n_words) return T3_OK;
    if (!d_words || !d_trits) return T3_E_ARG;
    // < 2^32 trits per launch; chunks of a multiple of 4 words keep every chunk's output dword-aligned
    const uint64_t chunk = (0xF0000000ull / (uint64_t)N) & ~3ull;
    for (uint64_t w0 = 0; w0 < n_words; w0 += chunk) {
        const uint64_t nw = std::min(chunk, n_words - w0);
        hipLaunchKernelGGL(subword_extract_kernel, dim3((unsigned)std::min<uint64_t>((nw + 511) / 512, 256u * 8u)), dim3(256), 0, (hipStream_t)stream,
                           (const uint8_t*)d_words + 9 * w0, nw, N, d_trits + w0 * (uint64_t)N);
        HIPCHK(hipGetLastError());
    }
    return T3_OK;
}

int t3hip_subword_build_dev(const uint8_t* d_trits, uint64_t n_trits, int N, uint8_t fill, void* d_words, uint64_t cap_words, uint64_t* n_words, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!valid_n(N) || !n_words) return T3_E_ARG;
    *n_words = t3hip_subword_words(n_trits, N);                           // OLD:845-859: one word per started group of N trits
    if (*n_words > cap_words) return T3_E_CAPACITY;
    if (!*n_words) return T3_OK;
    if (!d_trits || !d_words) return T3_E_ARG;
    hipLaunchKernelGGL(subword_build_kernel, dim3((unsigned)std::min<uint64_t>((*n_words + 511) / 512, 256u * 8u)), dim3(256), 0, (hipStream_t)stream, d_trits, n_trits, N, (uint32_t)fill, (uint8_t*)d_words, *n_words);
    HIPCHK(hipGetLastError()); return T3_OK;
}

int t3hip_base243_pack_dev(const uint8_t* d_trits, uint64_t n_trits, uint8_t* d_out, uint64_t cap_bytes, uint64_t* n_bytes, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n_bytes || n_trits > 0xFFFFFFFFull) return T3_E_ARG;             // the header holds a uint32 count (TPACK:31)
    *n_bytes = t3hip_base243_bytes(n_trits);
    if (*n_bytes > cap_bytes) return T3_E_CAPACITY;
    if (!d_out || (n_trits && !d_trits)) return T3_E_ARG;
    hipLaunchKernelGGL(base243_pack_kernel, dim3(blocks_for(((n_trits + 4) / 5 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_trits, n_trits, d_out);
    HIPCHK(hipGetLastError()); return T3_OK;
}

// `total` = the count read from the 4-byte header (t3hip_base243_unpack reads it; device pipelines know it)
int t3hip_base243_unpack_dev(const uint8_t* d_in, uint64_t n_bytes, uint64_t total, uint8_t* d_trits, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (n_bytes < 4) return T3_E_HEADER;                                   // TPACK:41
    if (total > 5 * (n_bytes - 4)) return T3_E_HEADER;                     // TPACK:49: fewer trits than announced -> false
    if (!total) return T3_OK;
    if (!d_in || !d_trits) return T3_E_ARG;
    hipLaunchKernelGGL(base243_unpack_kernel, dim3(blocks_for((n_bytes - 4 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_in + 4, n_bytes - 4, total, d_trits);
    HIPCHK(hipGetLastError()); return T3_OK;
}

int t3hip_mod27_bytes_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n) return T3_OK;
    if (!d_in || !d_out) return T3_E_ARG;
    hipLaunchKernelGGL(mod27_bytes_kernel, dim3(blocks_for((n + 15) / 16)), dim3(256), 0, (hipStream_t)stream, d_in, n, d_out);
    HIPCHK(hipGetLastError()); return T3_OK;
}


// ---- centring blits (io_image.hpp:125-140, 215-235) ----
// blit_center_rgb: the source image lands in the middle of a zeroed canvas, x0 = max(0, (cw - sw) / 2), y0 likewise; source rows
// that fall below the canvas are dropped as the reference drops them.  A source wider than the canvas is refused: the reference
// copies sw * 3 bytes into a cw * 3 byte row there (it runs over the following rows and, on the last one, off the buffer).
int t3hip_blit_center_rgb_dev(const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int cw, int ch, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (sw < 0 || sh < 0 || cw < 0 || ch < 0) return T3_E_ARG;
    const uint64_t dst_bytes = (uint64_t)cw * (uint64_t)ch * 3u;
    if (!dst_bytes) return T3_OK;
    const uint32_t y0 = (uint32_t)(ch > sh ? (ch - sh) / 2 : 0), rows = (uint32_t)(sh < ch - (int)y0 ? sh : ch - (int)y0);
    if (rows && sw > cw) return T3_E_ARG;
    if (!d_dst || ((uintptr_t)d_dst & 3u) || ((uint64_t)sw * sh && !d_src)) return T3_E_ARG;
    const uint32_t x0 = (uint32_t)(cw > sw ? (cw - sw) / 2 : 0);
    hipLaunchKernelGGL(center_window_kernel, dim3(blocks_for((dst_bytes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_src, (uint64_t)sw * 3u, 0u, (uint64_t)0,
                       d_dst, (uint64_t)cw * 3u, dst_bytes, y0, rows, (uint64_t)x0 * 3u, (uint64_t)sw * 3u);
    HIPCHK(hipGetLastError()); return T3_OK;
}
// extract_center_q: the sw x sh window in the middle of a fw x fh frame of 6-byte pixels; window rows below the frame come out
// zero (the reference's resize()).  A window wider than the frame is refused (the reference reads on into the next frame row).
int t3hip_extract_center_q_dev(const void* d_full_px6, int fw, int fh, void* d_sub_px6, int sw, int sh, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (fw < 0 || fh < 0 || sw < 0 || sh < 0) return T3_E_ARG;
    const uint64_t dst_bytes = (uint64_t)sw * (uint64_t)sh * 6u;
    if (!dst_bytes) return T3_OK;
    const uint32_t y0 = (uint32_t)(fh > sh ? (fh - sh) / 2 : 0), rows = (uint32_t)(sh < fh - (int)y0 ? sh : fh - (int)y0);
    if (rows && sw > fw) return T3_E_ARG;
    if (!d_sub_px6 || ((uintptr_t)d_sub_px6 & 3u) || ((uint64_t)fw * fh && !d_full_px6)) return T3_E_ARG;
    const uint32_t x0 = (uint32_t)(fw > sw ? (fw - sw) / 2 : 0);
    hipLaunchKernelGGL(center_window_kernel, dim3(blocks_for((dst_bytes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_full_px6, (uint64_t)fw * 6u, y0,
                       (uint64_t)x0 * 6u, (uint8_t*)d_sub_px6, (uint64_t)sw * 6u, dst_bytes, 0u, rows, (uint64_t)0, (uint64_t)sw * 6u);
    HIPCHK(hipGetLastError()); return T3_OK;
}

// ---- host-buffer entry points (what the std::vector API of include/ternary_codec_v6.hpp binds) ----
static int roundtrip(const void* in, uint64_t in_bytes, void** di, uint64_t out_bytes, void** dout) {
    int rc = api_scratch(0, in_bytes + 64, di); if (rc) return rc;
    rc = api_scratch(1, out_bytes + 64, dout); if (rc) return rc;
    if (in_bytes) HIPCHK(hipMemcpyAsync(*di, in, in_bytes, hipMemcpyHostToDevice, api_stream()));
    return T3_OK;
}
static int fetch(void* out, const void* dout, uint64_t bytes) {
    if (bytes) HIPCHK(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, api_stream()));
    HIPCHK(hipStreamSynchronize(api_stream())); return T3_OK;
}

int t3hip_subword_extract(const void* words9, uint64_t n_words, int N, uint8_t* trits) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (!valid_n(N)) return T3_E_ARG;
    if (!n_words) return T3_OK;
    if (!words9 || !trits) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(words9, n_words * 9, &di, n_words * (uint64_t)N, &dout); if (rc) return rc;
    rc = t3hip_subword_extract_dev(di, n_words, N, (uint8_t*)dout, api_stream()); if (rc) return rc;
    return fetch(trits, dout, n_words * (uint64_t)N);
}
int t3hip_subword_build(const uint8_t* trits, uint64_t n_trits, int N, uint8_t fill, void* words9, uint64_t cap_words, uint64_t* n_words) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (!valid_n(N) || !n_words || (n_trits && !trits)) return T3_E_ARG;
    const uint64_t nw = t3hip_subword_words(n_trits, N); *n_words = nw;
    if (nw > cap_words) return T3_E_CAPACITY;
    if (!nw) return T3_OK;
    if (!words9) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(trits, n_trits, &di, nw * 9, &dout); if (rc) return rc;
    rc = t3hip_subword_build_dev((const uint8_t*)di, n_trits, N, fill, dout, nw, n_words, api_stream()); if (rc) return rc;
    return fetch(words9, dout, nw * 9);
}
int t3hip_base243_pack(const uint8_t* trits, uint64_t n_trits, uint8_t* out, uint64_t cap_bytes, uint64_t* n_bytes) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (!n_bytes || n_trits > 0xFFFFFFFFull || (n_trits && !trits)) return T3_E_ARG;
    *n_bytes = t3hip_base243_bytes(n_trits);
    if (*n_bytes > cap_bytes) return T3_E_CAPACITY;
    if (!out) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(trits, n_trits, &di, *n_bytes, &dout); if (rc) return rc;
    rc = t3hip_base243_pack_dev((const uint8_t*)di, n_trits, (uint8_t*)dout, *n_bytes, n_bytes, api_stream()); if (rc) return rc;
    return fetch(out, dout, *n_bytes);
}
int t3hip_base243_unpack(const uint8_t* in, uint64_t n_bytes, uint8_t* trits, uint64_t cap_trits, uint64_t* n_trits) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (!n_trits || (n_bytes && !in)) return T3_E_ARG;
    *n_trits = 0;
    if (n_bytes < 4) return T3_E_HEADER;
    uint32_t total; memcpy(&total, in, 4);
    if ((uint64_t)total > 5 * (n_bytes - 4)) return T3_E_HEADER;
    *n_trits = total;
    if (total > cap_trits) return T3_E_CAPACITY;
    if (!total) return T3_OK;
    if (!trits) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(in, n_bytes, &di, total, &dout); if (rc) return rc;
    rc = t3hip_base243_unpack_dev((const uint8_t*)di, n_bytes, total, (uint8_t*)dout, api_stream()); if (rc) return rc;
    return fetch(trits, dout, total);
}
int t3hip_mod27_bytes(const uint8_t* in, uint64_t n, uint8_t* out) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (!n) return T3_OK;
    if (!in || !out) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(in, n, &di, n, &dout); if (rc) return rc;
    rc = t3hip_mod27_bytes_dev((const uint8_t*)di, n, (uint8_t*)dout, api_stream()); if (rc) return rc;
    return fetch(out, dout, n);
}

int t3hip_blit_center_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int cw, int ch) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (sw < 0 || sh < 0 || cw < 0 || ch < 0) return T3_E_ARG;
    const uint64_t nb_in = (uint64_t)sw * sh * 3u, nb_out = (uint64_t)cw * ch * 3u;
    if (!nb_out) return T3_OK;
    if (!dst || (nb_in && !src)) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(src, nb_in, &di, nb_out, &dout); if (rc) return rc;
    rc = t3hip_blit_center_rgb_dev((const uint8_t*)di, sw, sh, (uint8_t*)dout, cw, ch, api_stream()); if (rc) return rc;
    return fetch(dst, dout, nb_out);
}
int t3hip_extract_center_q(const void* full_px6, int fw, int fh, void* sub_px6, int sw, int sh) {
    if (!api_ready()) return T3_E_NODEVICE;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    if (fw < 0 || fh < 0 || sw < 0 || sh < 0) return T3_E_ARG;
    const uint64_t nb_in = (uint64_t)fw * fh * 6u, nb_out = (uint64_t)sw * sh * 6u;
    if (!nb_out) return T3_OK;
    if (!sub_px6 || (nb_in && !full_px6)) return T3_E_ARG;
    void *di, *dout; int rc = roundtrip(full_px6, nb_in, &di, nb_out, &dout); if (rc) return rc;
    rc = t3hip_extract_center_q_dev(di, fw, fh, dout, sw, sh, api_stream()); if (rc) return rc;
    return fetch(sub_px6, dout, nb_out);
}

// measurement aid (profiles/copy_ceiling.py): a plain streaming kernel over the same byte volumes as a codec launch
int t3hip_diag_stream_copy_dev(const void* d_src, uint64_t n_read, void* d_dst, uint64_t n_write, int blocks_per_cu, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!d_src || !d_dst || (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u)) return T3_E_ARG;
    const int nt = blocks_per_cu < 0; if (nt) blocks_per_cu = -blocks_per_cu;                 // negative: non-temporal loads and stores
    const unsigned grid = 256u * (unsigned)(blocks_per_cu > 0 ? blocks_per_cu : 8);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)d_src, n_read / 16, (uint4*)d_dst, n_write / 16, nt);
    HIPCHK(hipGetLastError()); return T3_OK;
}

}  // extern "C"
