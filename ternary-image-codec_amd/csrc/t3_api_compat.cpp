// t3_api_compat.cpp — the block- and symbol-level entry points behind the reference's remaining public names
// (include/ternary_codec_v6.hpp wraps them back into GF27Context / RSCodec / CRC3 / scramble_symbol / ... so that the
// reference's own callers, e.g. old/src/main_bare.cpp, compile against the drop-in header unchanged):
//   host constants     gf27_add / gf27_sub / gf27_mul_poly (OLD:383-413), scramble_symbol / descramble_symbol (OLD:81-94),
//                      encode_beacon_symbol (OLD:107-113), CRC3::rem12 (OLD:176-205) -- single symbols and the 27-symbol
//                      header are control data, computed on the host from the same tables the kernels are built from
//   device, host bufs  RSCodec::encode_block / decode_block (OLD:517-535, 546-662) over the block-level kernels,
//                      interleave2D_boustrophedon / deinterleave2D_boustrophedon (OLD:750-813) over interleave_kernel
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

#include "../../include/t3hip.h"
#include "t3_host.hpp"
#include "t3_kernels.h"

namespace t3 {
int api_ready(); hipStream_t api_stream(); int api_scratch(int slot, size_t bytes, void** out, hipStream_t s = nullptr);
int api_fail_hip(hipError_t e, const char* what); std::recursive_mutex& api_host_mutex();
}  // namespace t3
using namespace t3;
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return api_fail_hip(e_, #x); } while (0)

extern "C" {

uint8_t t3hip_gf27_add(uint8_t a, uint8_t b) { return field().t.add[(a % 27) * 27 + b % 27]; }
uint8_t t3hip_gf27_sub(uint8_t a, uint8_t b) { const Field& F = field(); return F.t.add[(a % 27) * 27 + F.t.neg[b % 27]]; }
uint8_t t3hip_gf27_mul(uint8_t a, uint8_t b) { return field().t.mul[(a % 27) * 27 + b % 27]; }

uint8_t t3hip_scramble_symbol(uint8_t s, uint32_t a, uint32_t b, uint32_t* st, int inverse) {
    if (!st) return s;
    *st = (a * *st + b) % 3u;                                              // uint32 wrap-around as in OLD:83
    const Field& F = field();
    const uint8_t k = (uint8_t)(13u * *st);                               // (st, st, st)
    return F.t.add[(s % 27) * 27 + (inverse ? F.t.neg[k] : k)];
}
uint8_t t3hip_beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health) { return beacon_symbol(profile, frame_seq_mod, health); }
int t3hip_crc12(const uint8_t* trits, uint64_t n, uint8_t out12[12]) {
    if ((n && !trits) || !out12 || n > (1u << 30)) return T3_E_ARG;
    crc12(trits, (int)n, out12); return T3_OK;
}

// One block on the host: what a caller that loops RSCodec::encode_block / decode_block block by block needs (the reference's
// self-test OLD:1172-1207, its header RS OLD:1142-1158) -- 26 bytes are control data, like the header codec; the parity
// matrix and the decoder are the ones the kernels are built from (t3_host.cpp, t3_rs_core.h).
int t3hip_rs_encode_block_host(int k, int mode, const uint8_t* data_k, uint8_t* code26) {
    if (!valid_k(k) || mode < 0 || mode > 1 || !data_k || !code26) return T3_E_ARG;
    static uint8_t P[2][4][24 * 8]; static std::once_flag once;
    std::call_once(once, [] { for (int m = 0; m < 2; ++m) for (int q = 0; q < 4; ++q) rs_parity_matrix(24 - 2 * q, m, P[m][q]); });
    const int r = 26 - k; const uint8_t* Pk = P[mode][(24 - k) / 2]; const Field& F = field();
    uint8_t par[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < k; ++i) {
        const uint8_t di = data_k[i] % 27u; code26[i] = data_k[i];
        for (int j = 0; j < r; ++j) par[j] = F.t.add[par[j] * 27 + F.t.mul[di * 27 + Pk[i * r + j]]];
    }
    for (int j = 0; j < r; ++j) code26[k + j] = par[j];
    return T3_OK;
}
int t3hip_rs_decode_block_host(int k, int mode, uint8_t* code26, uint8_t* data_k) {
    if (!valid_k(k) || mode < 0 || mode > 1 || !code26 || !data_k) return T3_E_ARG;
    uint8_t c[26];
    for (int i = 0; i < 26; ++i) c[i] = code26[i] % 27u;
    const bool good = rs_decode_host(k, c, mode == T3_MODE_FIXED);
    memcpy(code26, c, 26);                                                 // corrected (or partially modified) in place, like inout_n
    if (good) memcpy(data_k, c, (size_t)k);                                // out_k only on success (OLD:564,660)
    return good ? 1 : 0;
}

int t3hip_rs_encode_blocks(int k, int mode, const uint8_t* data_k, uint64_t n_blocks, uint8_t* code26) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!valid_k(k) || mode < 0 || mode > 1) return T3_E_ARG;
    if (!n_blocks) return T3_OK;
    if (!data_k || !code26) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> lk(api_host_mutex());
    void *di, *dout; int rc = api_scratch(0, n_blocks * k + 64, &di); if (rc) return rc;
    rc = api_scratch(1, n_blocks * 26 + 64, &dout); if (rc) return rc;
    hipStream_t s = api_stream();
    HIPCHK(hipMemcpyAsync(di, data_k, n_blocks * k, hipMemcpyHostToDevice, s));
    rc = t3hip_rs_encode_blocks_dev(k, mode, (const uint8_t*)di, n_blocks, (uint8_t*)dout, s); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(code26, dout, n_blocks * 26, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s)); return T3_OK;
}
int t3hip_rs_decode_blocks(int k, int mode, uint8_t* code26, uint64_t n_blocks, uint8_t* data_k, uint8_t* ok) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!valid_k(k) || mode < 0 || mode > 1) return T3_E_ARG;
    if (!n_blocks) return T3_OK;
    if (!code26 || !data_k || !ok) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> lk(api_host_mutex());
    void *dc, *dd; int rc = api_scratch(0, n_blocks * 26 + 64, &dc); if (rc) return rc;
    rc = api_scratch(1, n_blocks * (k + 1) + 64, &dd); if (rc) return rc;
    uint8_t* dok = (uint8_t*)dd + n_blocks * k;
    hipStream_t s = api_stream();
    HIPCHK(hipMemcpyAsync(dc, code26, n_blocks * 26, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dd, data_k, n_blocks * k, hipMemcpyHostToDevice, s));   // out_k stays untouched where decode_block returns false
    rc = t3hip_rs_decode_blocks_dev(k, mode, (uint8_t*)dc, n_blocks, (uint8_t*)dd, dok, s); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(code26, dc, n_blocks * 26, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(data_k, dd, n_blocks * k, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(ok, dok, n_blocks, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s)); return T3_OK;
}

int t3hip_interleave2d_dev(const uint8_t* d_in, uint64_t n, uint16_t w, uint16_t h, uint8_t* d_out, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n) return T3_OK;
    if (!d_in || !d_out || d_in == d_out || n >= (1ull << 31)) return T3_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!w || !h) { HIPCHK(hipMemcpyAsync(d_out, d_in, n, hipMemcpyDeviceToDevice, s)); return T3_OK; }      // OLD:752: no-op
    const uint32_t A = (uint32_t)std::min<uint64_t>((uint64_t)w * h, n);
    const FastDiv fa = fastdiv(A), fw = fastdiv(w);
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_in, d_out, (uint32_t)n, (uint32_t)w, A,
                       DevDiv{fa.mul, fa.sh, fa.d}, DevDiv{fw.mul, fw.sh, fw.d});
    HIPCHK(hipGetLastError()); return T3_OK;
}
int t3hip_interleave2d(uint8_t* syms, uint64_t n, uint16_t w, uint16_t h, int inverse) {
    (void)inverse;                                                          // the map is an involution inside every row segment (OLD:750-813)
    if (!api_ready()) return T3_E_NODEVICE;
    if (!n || !w || !h) return T3_OK;
    if (!syms) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> lk(api_host_mutex());
    void *di, *dout; int rc = api_scratch(0, n + 64, &di); if (rc) return rc;
    rc = api_scratch(1, n + 64, &dout); if (rc) return rc;
    hipStream_t s = api_stream();
    HIPCHK(hipMemcpyAsync(di, syms, n, hipMemcpyHostToDevice, s));
    rc = t3hip_interleave2d_dev((const uint8_t*)di, n, w, h, (uint8_t*)dout, s); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(syms, dout, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s)); return T3_OK;
}

}  // extern "C"
