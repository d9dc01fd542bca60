// t3_api_decode.cpp — decode-side half of the C-ABI (include/t3hip.h): header parse on the host, body kernels
// on the device, block-level decode, error injector, frame index record.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/t3hip.h"
#include "t3_decode.h"
#include "t3_host.hpp"

namespace t3 {
int api_ready(); hipStream_t api_stream(); int api_scratch(int slot, size_t bytes, void** out, hipStream_t s = nullptr);
int api_fail_hip(hipError_t e, const char* what); uint32_t* api_flag(); RsTables* api_tables(); int api_n_cu(); int api_device(); std::recursive_mutex& api_host_mutex(); std::mutex& api_tab_mutex(); std::recursive_mutex& api_mail_mutex(); uint32_t* api_ticket_counters(hipStream_t s, int kind); int api_pipeline(uint32_t n_events, hipStream_t* s2, hipEvent_t** evs);
void*& api_slot(int id);          // per-context object slots (t3_api.cpp): this file owns 0..31
}  // namespace t3
using namespace t3;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return api_fail_hip(e_, #x); } while (0)

namespace {
// Device tables and pinned mailboxes of the decode half live in the calling thread's context (one per GPU, t3_api.cpp), in
// numbered slots; the names below resolve to the current context at every use.
#define T3_SLOT(T, id) (*(T*)&api_slot(id))
#define d_zpow        T3_SLOT(uint32_t*, 0)     // CRC "append 2^j zero bytes" operators
#define d_crc_acc     T3_SLOT(uint32_t*, 1)     // [0] xor accumulator, [1] symbol sum
#define d_crc_afrag   T3_SLOT(uint32_t*, 2)     // bit-matrix slices of the matrix-core CRC (t3_crc_mfma.hip)
#define d_fxtab       T3_SLOT(FxTables*, 3)     // field tables of the two-kernel FIXED decoder
#define d_fma         T3_SLOT(uint8_t*, 4)      // fma[x][y][a] = a + x y: one table read per multiply-accumulate of the corrector
#define d_synd_T      T3_SLOT(uint32_t*, 5)     // descramble + trit expansion table of the syndrome MFMA, 32 bank copies
#define d_synd_T16    T3_SLOT(uint32_t*, 6)     // ... 16 bank copies
#define d_fx2_small   T3_SLOT(uint8_t*, 7)      // log / exp / inverse byte tables + fold tables of the fused decoders
#define d_rgb_dq      T3_SLOT(uint8_t*, 8)      // dequantiser tables of the fused RGB output stage
#define h_mail        T3_SLOT(uint8_t*, 9)      // pinned mailbox of read_header
#define h_flag        T3_SLOT(uint32_t*, 10)    // mapped pinned failure counter of the synchronous decode entry
#define d_flag_map    T3_SLOT(uint32_t*, 11)
#define d_synd_lut    (&T3_SLOT(uint32_t*, 12)) // [4] per k index: syndrome LUT of the two-kernel decoder
#define d_roots       (&T3_SLOT(uint32_t*, 16)) // [4] per k index: the Chien search (OLD:611-623) of every locator, tabulated
#define d_crc_afrag4  T3_SLOT(uint32_t*, 24)    // ... of its FP4 form (t3_crc_fp4.hip)
#define d_crc_afb     T3_SLOT(uint32_t*, 25)    // FP4 CRC, strided rounds: the feedback slice "append 2048 W zero bytes" ...
#define crc_afb_w     T3_SLOT(uintptr_t, 26)    // ... and the W it was built for
#define d_synd_afrag  (&T3_SLOT(uint32_t*, 20)) // [4] per k index: A operand of the syndrome MFMA (t3_host.hpp build_mfma_syndrome)
uint32_t synd_lut_bytes[4] = {0, 0, 0, 0};      // (a size, the same for every context)

int k_index(int k) { return k == 24 ? 0 : k == 22 ? 1 : k == 20 ? 2 : k == 18 ? 3 : -1; }
DevDiv to_dev(FastDiv f) { return DevDiv{f.mul, f.sh, f.d}; }

// The streaming entry's header check (t3hip_decode_frame_async): pending until a decode path either takes it into its own launch (the pixel
// kernel with tile tickets: decode_fixed_fused) or launches hdr_compare_kernel in front of its kernels (hdr_flush) -- either way before
// anything counts failures into verdict[1].
struct HdrPending { HdrExpect ex; uint32_t hs; uint32_t* verdict; const uint8_t* in; bool pending; };
static thread_local HdrPending tl_hdr = {{}, 0, nullptr, nullptr, false};
static int hdr_flush(hipStream_t s) {
    if (!tl_hdr.pending) return T3_OK;
    tl_hdr.pending = false;
    hipLaunchKernelGGL(hdr_compare_kernel, dim3(1), dim3(128), 0, s, tl_hdr.in, tl_hdr.ex, tl_hdr.hs, tl_hdr.verdict);   // also zeroes the block counter
    HIPCHK(hipGetLastError());
    return T3_OK;
}

// Fused FIXED decode (t3_decode_fused.hip): uniform k, 1-D, no beacon.  Returns T3_OK after launching, or 1 if not applicable.
// The lazily built device tables below are shared by every caller thread of a context, and so are the pinned host mailboxes of the
// synchronous entry points: both locks live in the context (api_tab_mutex / api_mail_mutex), two contexts never share one.
#define g_tab_mu api_tab_mutex()
#define g_mail_mu api_mail_mutex()

// device tables of one code (syndrome LUT, root masks) and the multiply-accumulate table, built on first use
int ensure_fx_tables(int k) {
    const int ki = k_index(k);
    if (!d_synd_lut[ki]) {
        std::vector<uint32_t> img; build_syndrome_lut(k, img);
        synd_lut_bytes[ki] = (uint32_t)img.size() * 4u;
        HIPCHK(hipMalloc((void**)&d_synd_lut[ki], synd_lut_bytes[ki]));
        HIPCHK(hipMemcpy(d_synd_lut[ki], img.data(), synd_lut_bytes[ki], hipMemcpyHostToDevice));
    }
    if (!d_roots[ki]) {
        // sigma_0 = 1 always (Berlekamp-Massey), so sigma is its t = r/2 higher coefficients: 27^t locators (80 KB for
        // RS(26,20), 2 MB for RS(26,18)).  Entry = the 26-bit mask of positions i with sigma(alpha^-i) = 0.
        const Field& F = field(); const int t = (26 - k) / 2;
        size_t n = 1; for (int i = 0; i < t; ++i) n *= 27;
        std::vector<uint32_t> tbl(n);
        for (size_t idx = 0; idx < n; ++idx) {
            uint8_t sg[5] = {1, 0, 0, 0, 0}; { size_t v = idx; for (int q = 1; q <= t; ++q) { sg[q] = (uint8_t)(v % 27); v /= 27; } }
            uint32_t mask = 0;
            for (int i = 0; i < 26; ++i) {
                const uint8_t x = F.t.exp[i == 0 ? 0 : 26 - i];
                uint8_t acc = sg[t];
                for (int q = t - 1; q >= 0; --q) acc = F.t.add[F.t.mul[acc * 27 + x] * 27 + sg[q]];
                if (acc == 0) mask |= 1u << i;
            }
            tbl[idx] = mask;
        }
        HIPCHK(hipMalloc((void**)&d_roots[ki], n * 4));
        HIPCHK(hipMemcpy(d_roots[ki], tbl.data(), n * 4, hipMemcpyHostToDevice));
    }
    if (!d_synd_afrag[ki]) {
        std::vector<uint32_t> af; build_mfma_syndrome(k, af);
        HIPCHK(hipMalloc((void**)&d_synd_afrag[ki], af.size() * 4)); HIPCHK(hipMemcpy(d_synd_afrag[ki], af.data(), af.size() * 4, hipMemcpyHostToDevice));
    }
    if (!d_synd_T) {
        std::vector<uint32_t> img; build_syndrome_T(img);
        HIPCHK(hipMalloc((void**)&d_synd_T, img.size() * 4)); HIPCHK(hipMemcpy(d_synd_T, img.data(), img.size() * 4, hipMemcpyHostToDevice));
        build_syndrome_T(img, 16);
        HIPCHK(hipMalloc((void**)&d_synd_T16, img.size() * 4)); HIPCHK(hipMemcpy(d_synd_T16, img.data(), img.size() * 4, hipMemcpyHostToDevice));
        uint8_t sm[kFx2SmallBytes + kFx2ModBytes]; build_fx2_small(sm); build_fx2_mod(sm + kFx2SmallBytes);
        HIPCHK(hipMalloc((void**)&d_fx2_small, sizeof sm)); HIPCHK(hipMemcpy(d_fx2_small, sm, sizeof sm, hipMemcpyHostToDevice));
    }
    if (!d_fma) {
        const Field& F = field();
        std::vector<uint8_t> t(19696, 0);
        for (int x = 0; x < 27; ++x) for (int y = 0; y < 27; ++y) for (int c = 0; c < 27; ++c) t[(size_t)(x * 27 + y) * 27 + c] = F.t.add[c * 27 + F.t.mul[x * 27 + y]];
        HIPCHK(hipMalloc((void**)&d_fma, t.size())); HIPCHK(hipMemcpy(d_fma, t.data(), t.size(), hipMemcpyHostToDevice));
    }
    return T3_OK;
}

// dequantiser tables of the fused RGB output stage (old/include/io_image.hpp:79-84: the reference's double expressions, tabulated)
int rgb_dequant_tables(const uint8_t** out) {
    uint8_t*& d = d_rgb_dq;
    if (!d) {
        uint8_t t[328]; memset(t, 0, sizeof t);
        auto cl = [](long v) { return v < 0 ? 0 : (v > 255 ? 255 : v); };
        for (int q = 0; q <= 242; ++q) t[q] = (uint8_t)cl(lround(q * (255.0 / 242.0)));
        for (int q = -40; q <= 40; ++q) t[244 + q + 40] = (uint8_t)cl(lround(128 + q * (128.0 / 40.0)));
        HIPCHK(hipMalloc((void**)&d, sizeof t)); HIPCHK(hipMemcpy(d, t, sizeof t, hipMemcpyHostToDevice));
    }
    *out = d; return T3_OK;
}

// `body`: the coded stream with `hdr_syms` symbols of header in front of the band-serial body; bcn_period != 0: the body still carries
// its beacon symbols (slot bcn_slot of every bcn_period-th word, OLD:952-957) and the loads step over them
// tile_lo / tile_hi (pixels, no beacon): only that range of tiles -- the pipelined host entry decodes a frame chunk by chunk; the kernel
// sees a frame of its own (band offsets, block counts and the output pointer shifted: tiles are independent).  *tiles_out: the frame's tile count.
int decode_fixed_fused(const uint8_t* body, uint64_t body_bytes, uint32_t hdr_syms, const t3_layout& L, const ScrCycle& sc,
                       void* d_out, uint64_t units, int to_pixels, uint32_t* d_fail, hipStream_t s, uint32_t bcn_slot = 0, uint32_t bcn_period = 0,
                       uint32_t tile_lo = 0, uint32_t tile_hi = 0xFFFFFFFFu, uint32_t* tiles_out = nullptr, uint32_t* units_tile_out = nullptr) {
    if (L.interleave2d || L.n_raw_words == 0) return 1;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    for (int b = 1; b < 9; ++b) if (L.band_k[b] != L.band_k[0]) return 1;
    const int k = L.band_k[0], ki = k_index(k);
    { const int rc = ensure_fx_tables(k); if (rc) return rc; }
    DecFx2Args a; memset(&a, 0, sizeof a);
    a.in = body; a.in_bytes = body_bytes; a.out = d_out; a.n_units = units; a.fail = d_fail; a.roots = d_roots[ki];
    a.ttab = (to_pixels && T3_DEC_PX_TCOP == 16) ? d_synd_T16 : d_synd_T; a.small = d_fx2_small; a.afrag = d_synd_afrag[ki];
    const bool rgb = to_pixels == 2;
    if (rgb) { const int rc = rgb_dequant_tables(&a.dq); if (rc) return rc; }
    a.k = (uint32_t)k; a.nb = to_pixels ? (uint32_t)T3_DEC_PX_NB : 52u; a.div_nb = to_dev(fastdiv(a.nb)); a.TS = 9u * a.nb * (uint32_t)k; a.n_sym = (uint32_t)L.n_sym; a.hdr_syms = hdr_syms;
    uint64_t maxb = 0;
    for (int b = 0; b < 9; ++b) { a.band_blocks[b] = (uint32_t)L.band_blocks[b]; a.band_body_off[b] = L.band_body_off[b]; a.band_boff6[b] = (uint32_t)((L.band_body_off[b] + 4) % 6); maxb = std::max<uint64_t>(maxb, L.band_blocks[b]); }
    a.n_tiles = (uint32_t)((maxb + a.nb - 1) / a.nb);
    if (tiles_out) *tiles_out = a.n_tiles;
    if (units_tile_out) *units_tile_out = (a.TS / 13u) * 3u;
    if (tile_lo || tile_hi < a.n_tiles) {
        if (!to_pixels || bcn_period || tile_lo >= std::min(tile_hi, a.n_tiles)) return T3_E_ARG;
        const uint64_t u0 = (uint64_t)tile_lo * ((a.TS / 13u) * 3u);
        for (int b = 0; b < 9; ++b) {
            const uint64_t skip = (uint64_t)tile_lo * a.nb;
            a.band_blocks[b] = (uint32_t)(a.band_blocks[b] > skip ? a.band_blocks[b] - skip : 0); a.band_body_off[b] += 26 * skip; a.band_boff6[b] = (uint32_t)((a.band_body_off[b] + 4) % 6);
        }
        a.n_tiles = std::min(tile_hi, a.n_tiles) - tile_lo;
        a.out = (uint8_t*)d_out + u0 * (to_pixels == 2 ? 3u : 6u); a.n_units = units > u0 ? units - u0 : 0;
    }
    a.cyc24 = sc.cyc24; a.pre0 = sc.pre[0]; a.pre1 = sc.pre[1];
    {   // body symbol i >= 2 sees cyc[(i - 2) mod 6]; a block whose first symbol has phase c0 = (i0 + 4) mod 6 sees cyc[(c0 + p) mod 6] at position p
        uint8_t rows[12][16]; memset(rows, 0, sizeof rows);
        for (int r = 0; r < 11; ++r) for (int q = 0; q < 13; ++q) rows[r][q] = (uint8_t)(27u * sc.cyc[(r + q) % 6]);     // rows 0..10: the phase arrives un-reduced
        for (int q = 0; q < 13; ++q) rows[11][q] = (uint8_t)(27u * (q == 0 ? sc.pre[0] : q == 1 ? sc.pre[1] : sc.cyc[(4 + q) % 6]));   // the stream's first block: c0 = 4
        memcpy(a.pat, rows, sizeof rows);
    }
    const bool bcn = bcn_period != 0;
    if (bcn) { a.bcn_slot = bcn_slot; a.bcn_pb = 9u * bcn_period - 1u; a.bcn_div = to_dev(fastdiv(a.bcn_pb)); }
    a.fma = d_fma;
    const uint32_t ybytes = (a.TS + 16u + 15u) & ~15u;
    if (to_pixels) {   // [hdr][fold 512][T16 1024][FMA][A operand][Y0][Y1][Q0][Q1]
        a.fma_off = (uint32_t)kFx2TPx + 3u * 27u * 4u * (uint32_t)T3_DEC_PX_TCOP;
        a.af_off = a.fma_off + 19696u;
        a.pat_off = a.af_off + 3328u;
        a.y_off = a.pat_off + 192u; a.y_stride = ybytes;
        a.q_off = a.y_off + 2u * ybytes; a.q_stride = 10u * (uint32_t)kFx2QCap;    // 8 bytes of syndromes + 2 of item number per entry
        a.o_off = a.q_off + 2u * a.q_stride; a.dq_off = a.o_off;
        a.lds_bytes = a.o_off + (rgb ? 336u : 16u);                                 // <= 42 x 1280 B: LDS is handed out in 1280-byte units, 128 per CU (three workgroups)
    } else {           // [hdr][fold 3072][T32 3584][FMA][A operand][Y][Q][words]
        a.fma_off = (uint32_t)kFx2TSeq + 3u * 27u * 4u * 32u;
        a.af_off = a.fma_off + 19696u;
        a.pat_off = a.af_off + 3328u;
        a.y_off = a.pat_off + 192u; a.y_stride = 0;
        a.q_off = a.y_off + ybytes; a.q_stride = 0;
        a.o_off = a.q_off + 10u * 512u;
        a.lds_bytes = a.o_off + (a.TS / 26u) * 27u + 64u;
    }
    const void* fn = nullptr;
    switch (26 - k) {
#define T3_PICKB(R, B) (rgb ? (const void*)decode_fixed_px_kernel<R, true, B> : to_pixels ? (const void*)decode_fixed_px_kernel<R, false, B> : (const void*)decode_fixed_kernel<R, B>)
#define T3_PICK(R) (bcn ? T3_PICKB(R, true) : T3_PICKB(R, false))
        case 2: fn = T3_PICK(2); break;
        case 4: fn = T3_PICK(4); break;
        case 6: fn = T3_PICK(6); break;
        default: fn = T3_PICK(8); break;
#undef T3_PICK
#undef T3_PICKB
    }
    static std::map<std::pair<const void*, int>, int> occ;          // per device (the attribute is set on the device's copy of the function)
    const auto okey = std::make_pair(fn, api_device());
    auto it = occ.find(okey);
    if (it == occ.end()) {
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int o = 1; HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, to_pixels ? T3_DEC_PX_THREADS : 512, a.lds_bytes));
        // measured on MI355X (stamp build, workgroup start times): LDS is handed out in 1280-byte units, 128 per CU; the occupancy
        // query does not round, and a persistent grid sized one workgroup per CU too large runs its last third after the rest
        o = std::min<int>(o, (int)(128u / ((a.lds_bytes + 1279u) / 1280u)));
        it = occ.emplace(okey, std::max(1, o)).first;
    }
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(a.n_tiles, (uint32_t)(api_n_cu() * it->second)));
#ifdef T3_DEC_STAMPS
    static uint64_t* d_dbg = nullptr; static int calls = 0;
    if (!d_dbg) HIPCHK(hipMalloc((void**)&d_dbg, 16 * 8 * 4096));
    HIPCHK(hipMemsetAsync(d_dbg, 0, 16 * 8 * 4096, s));
    a.dbg = d_dbg;
#endif
    if (to_pixels) {   // dynamic tile tickets (decode_fixed_px_kernel); T3HIP_STATIC_TILES: measurement knob
        static const bool off = getenv("T3HIP_STATIC_TILES") != nullptr;
        a.tile_ctr = off ? nullptr : api_ticket_counters(s, 1); a.n_classes = std::min<uint32_t>(8u, grid);
    }
    if (tl_hdr.pending) {
        static const bool no_fold = getenv("T3HIP_HDR_KERNEL") != nullptr;                    // measurement / test knob: the separate header kernel
        if (to_pixels && a.tile_ctr && !no_fold && tile_lo == 0 && tile_hi == 0xFFFFFFFFu && tl_hdr.hs <= 96u && ((uintptr_t)tl_hdr.in & 3u) == 0) {
            tl_hdr.pending = false;
            a.verdict = tl_hdr.verdict; a.hdr_in = tl_hdr.in; a.hdr_n = tl_hdr.hs; memcpy(a.hx, tl_hdr.ex.b, 96);
            a.fail = a.tile_ctr + 64u * a.n_classes + 16u;                                    // library-owned, zero between launches; the last workgroup moves it to verdict[1]
        } else { const int rc = hdr_flush(s); if (rc) return rc; }
    }
    void* args[] = {(void*)&a};
    HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(to_pixels ? T3_DEC_PX_THREADS : 512), args, a.lds_bytes, s));
#ifdef T3_DEC_STAMPS
    if (++calls == 8 && to_pixels) {                        // one report, after warm-up: mean cycles of waves 0 / 4 per workgroup and phase
        std::vector<uint64_t> h(16 * grid);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(h.data(), d_dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double acc[16] = {0};
        for (uint32_t w = 0; w < grid; ++w) for (int i = 0; i < 16; ++i) acc[i] += (double)h[16 * w + i];
        fprintf(stderr, "[t3 dec stamps] grid=%u (%d per CU) tiles=%u lds=%u  mean cycles/WG: producer sets+e1=%.0f barrier=%.0f total=%.0f | consumer bm=%.0f rendezvous=%.0f d5=%.0f barrier=%.0f total=%.0f  clock=%.3f GHz  us/WG=%.1f\n",
                grid, it->second, a.n_tiles, a.lds_bytes, acc[0] / grid, acc[1] / grid, acc[6] / grid, acc[8 + 2] / grid, acc[8 + 3] / grid, acc[8 + 4] / grid, acc[8 + 5] / grid, acc[8 + 6] / grid,
                acc[6] / acc[7] * 0.1, acc[7] / grid * 0.01);
        uint64_t s0 = ~0ull, s1 = 0, e1 = 0; int late = 0;
        for (uint32_t w = 0; w < grid; ++w) { s0 = std::min(s0, h[16 * w + 2]); s1 = std::max(s1, h[16 * w + 2]); e1 = std::max(e1, h[16 * w + 3]); }
        for (uint32_t w = 0; w < grid; ++w) if (h[16 * w + 2] - s0 > 1000) ++late;
        fprintf(stderr, "[t3 dec stamps]   timeline (us from first loop start): last start=%.1f last end=%.1f  workgroups starting >10 us late=%d\n", (s1 - s0) * 0.01, (e1 - s0) * 0.01, late);
    }
#endif
    return T3_OK;
}

unsigned grid_for(uint64_t items, unsigned block) { return (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, (items + block - 1) / block), 1u << 20); }

int occupancy_of(const void* fn, int threads, uint32_t lds_bytes, int* out) {
    static std::map<std::pair<const void*, uint64_t>, int> occ;     // per device
    auto key = std::make_pair(fn, (uint64_t)(uint32_t)api_device() << 32 | lds_bytes);
    auto it = occ.find(key);
    if (it == occ.end()) {
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int o = 1; HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, fn, threads, lds_bytes));
        it = occ.emplace(key, std::max(1, o)).first;
    }
    *out = it->second; return T3_OK;
}

// One-launch FIXED decode for per-band k (two codes at most) and / or the 2-D interleave, pixels out (t3_decode_uep.hip).  Returns T3_OK
// after launching, 1 if not applicable (the two-kernel path then takes the frame).
int decode_fixed_uep(const uint8_t* body, uint64_t body_bytes, uint32_t hdr_syms, const t3_cfg& cfg, const t3_layout& L, const ScrCycle& sc,
                     void* d_out, uint64_t units, uint32_t* d_fail, hipStream_t s) {
    if (L.n_raw_words == 0 || L.n_sym + (1u << 20) >= (1ull << 32) || body_bytes + hdr_syms >= (1ull << 32)) return 1;
    if (getenv("T3HIP_GENERIC_DECODE") != nullptr || getenv("T3HIP_TWO_KERNEL_DECODE") != nullptr) return 1;
    if (body_bytes >= (1ull << 28)) return 1;                              // the kernel packs a block's byte offset into 28 bits
    const bool il = L.interleave2d != 0 && cfg.tile_w > 1;                 // (rows of one symbol: the map is the identity)
    DecUepArgs a; memset(&a, 0, sizeof a);
    int gk[kUepMaxGrp]; uint32_t gn[kUepMaxGrp] = {0, 0}; int ng = 0;
    for (int b = 0; b < 9; ++b) {
        int g = -1;
        for (int q = 0; q < ng; ++q) if (gk[q] == L.band_k[b]) g = q;
        if (g < 0) { if (ng == kUepMaxGrp) return 1; g = ng++; gk[g] = L.band_k[b]; }
        a.grp[g].bands[gn[g]++] = (uint8_t)b;
    }
    if (ng == 2 && gk[0] > gk[1]) {                                        // group 0 = the code with more parity (the kernel is instantiated for r0 >= r1)
        std::swap(gk[0], gk[1]); std::swap(gn[0], gn[1]);
        uint8_t t[12]; memcpy(t, a.grp[0].bands, 12); memcpy(a.grp[0].bands, a.grp[1].bands, 12); memcpy(a.grp[1].bands, t, 12);
    }
    if (il) {
        const uint64_t A = (uint64_t)cfg.tile_w * cfg.tile_h;
        if (cfg.tile_w % 4u || (A % 4u && A < L.n_sym)) return 1;                       // the in-place row reversal works on dword granules (the stream's last row: bytes)
        a.il_on = 1; a.il_w = cfg.tile_w; a.il_A = (uint32_t)std::min<uint64_t>(A, L.n_sym);
        if (a.il_A < 2 || a.il_w < 2) return 1;
        a.div_A = to_dev(fastdiv(a.il_A)); a.div_w = to_dev(fastdiv(a.il_w));
    }
    std::lock_guard<std::mutex> lk(g_tab_mu);
    uint32_t lcm = 1;
    for (int g = 0; g < ng; ++g) { const int rc = ensure_fx_tables(gk[g]); if (rc) return rc; uint32_t x = lcm, y = (uint32_t)gk[g]; while (y) { const uint32_t t = x % y; x = y; y = t; } lcm = lcm / x * (uint32_t)gk[g]; }
    // LDS: [hdr][fold 512][T16 1024][FMA][A operands][pattern rows 192][records 128][Y0][Y1][Q0][Q1] within 42 x 1280 B (three workgroups per CU)
    const uint32_t fixed_bytes = (uint32_t)kFx2TPx + 3u * 27u * 4u * 16u + 19696u + 3328u + 192u + 128u;   // (one A operand: group 0's evaluation matrix holds group 1's)
    const uint32_t budget = 42u * 1280u;
    uint32_t best_m = 0;
    for (uint32_t m = 1; 9u * lcm * m <= 12000u; ++m) {
        const uint32_t Lq = lcm * m, TS = 9u * Lq;
        if (TS % 4u) continue;
        uint32_t pairs = 0; bool small = false;
        for (int g = 0; g < ng; ++g) { const uint32_t nbg = Lq / (uint32_t)gk[g], items = gn[g] * nbg; if (nbg < 2u) small = true; if (nbg >= 2048u) pairs = 99; pairs += ((items + 31u) / 32u + 1u) / 2u; }
        if (small) continue;                                                              // (the kernel's divisions need two blocks per band and tile at least)
        if (pairs > 8u) break;
        const uint32_t ybytes = (TS + 16u + 15u) & ~15u;
        // (larger tiles beat larger queues: measured on BASELINE configs[2], 0.189 ms with 9 x 1100 symbols per tile and room for 0.47 of
        // its blocks in the queues against 0.211 with 9 x 880 and room for 0.6; a block that finds its queue full is corrected on the spot)
        if (fixed_bytes + 2u * ybytes + 2u * 10u * 96u + 64u > budget) break;            // (at least 96 queue entries per buffer)
        best_m = m;
    }
    if (!best_m) return 1;
    const uint32_t Lq = lcm * best_m;
    a.TS = 9u * Lq; a.n_sym = (uint32_t)L.n_sym; a.hdr_syms = hdr_syms; a.n_grp = (uint32_t)ng;
    a.in = body; a.in_bytes = body_bytes; a.out = d_out; a.n_units = units; a.fail = d_fail;
    a.ttab = d_synd_T16; a.small = d_fx2_small; a.fma = d_fma;
    a.fma_off = (uint32_t)kFx2TPx + 3u * 27u * 4u * 16u;
    uint32_t off = a.fma_off + 19696u;
    uint64_t tiles = 0; uint32_t items_all = 0;
    for (int g = 0; g < ng; ++g) {
        auto& G = a.grp[g]; const int ki = k_index(gk[g]);
        G.r = 26u - (uint32_t)gk[g]; G.nb = Lq / (uint32_t)gk[g]; G.n_items = gn[g] * G.nb; G.div_nb = to_dev(fastdiv(G.nb));
        G.afrag = d_synd_afrag[ki]; G.roots = d_roots[ki];
        if (g == 0) { G.af_off = off; off += 3328u; } else G.af_off = a.grp[0].af_off;
        items_all += G.n_items;
        for (uint32_t i = 0; i < gn[g]; ++i) tiles = std::max<uint64_t>(tiles, (L.band_blocks[G.bands[i]] + G.nb - 1) / G.nb);
    }
    a.n_tiles = (uint32_t)tiles;
    a.pat_off = off; off += 192u; a.rec_off = off; off += 128u;
    const uint32_t ybytes = (a.TS + 16u + 15u) & ~15u;
    a.y_off = off; a.y_stride = ybytes; off += 2u * ybytes;
    const uint32_t q_entries = std::min<uint32_t>((budget - 64u - off) / 20u, items_all);  // per buffer, split over the groups by their share of the blocks (64: roundings below)
    uint32_t qrel = 0;
    for (int g = 0; g < ng; ++g) { auto& G = a.grp[g]; G.q_cap = std::max<uint32_t>(8u, (uint32_t)((uint64_t)q_entries * G.n_items / items_all) & ~3u); G.q_rel = qrel; qrel += 10u * G.q_cap; }
    a.q_off = off; a.q_stride = (qrel + 15u) & ~15u; off += 2u * a.q_stride;
    a.lds_bytes = off + 16u;
    if (a.lds_bytes > 160u * 1024u) return 1;
    {   // (wave, pass) pairs of sets: pair p -> wave p % 4, pass p / 4
        for (int i = 0; i < 8; ++i) a.pair_tab[i] = 0xFFFFFFFFu;
        uint32_t p = 0;
        for (int g = 0; g < ng; ++g) for (uint32_t it0 = 0; it0 < a.grp[g].n_items; it0 += 64u, ++p) a.pair_tab[2u * (p % 4u) + p / 4u] = (uint32_t)g | it0 << 8;
    }
    for (int b = 0; b < 9; ++b) { a.band_blocks[b] = (uint32_t)L.band_blocks[b]; a.band_body_off[b] = L.band_body_off[b]; a.band_boff6[b] = (uint32_t)((L.band_body_off[b] + 4) % 6); }
    {
        uint8_t rows[12][16]; memset(rows, 0, sizeof rows);
        for (int r = 0; r < 11; ++r) for (int q = 0; q < 13; ++q) rows[r][q] = (uint8_t)(27u * sc.cyc[(r + q) % 6]);
        for (int q = 0; q < 13; ++q) rows[11][q] = (uint8_t)(27u * (q == 0 ? sc.pre[0] : q == 1 ? sc.pre[1] : sc.cyc[(4 + q) % 6]));
        memcpy(a.pat, rows, sizeof rows);
    }
    void* d_e; int rc = api_scratch(3, L.n_sym + 64, &d_e, s); if (rc) return rc;
    a.edge = (uint8_t*)d_e;
    const void* fn = nullptr;
    {
        const int ra = 26 - gk[0], rb = ng == 2 ? 26 - gk[1] : ra;
#define T3_UEP(A, B) if (ra == A && rb == B) fn = (const void*)decode_uep_px_kernel<A, B>;
        T3_UEP(2, 2) T3_UEP(4, 4) T3_UEP(6, 6) T3_UEP(8, 8) T3_UEP(4, 2) T3_UEP(6, 2) T3_UEP(6, 4) T3_UEP(8, 2) T3_UEP(8, 4) T3_UEP(8, 6)
#undef T3_UEP
        if (!fn) return 1;
    }
    int occ = 1; rc = occupancy_of(fn, 512, a.lds_bytes, &occ); if (rc) return rc;
    occ = std::max(1, std::min<int>(occ, (int)(128u / ((a.lds_bytes + 1279u) / 1280u))));
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(a.n_tiles, (uint32_t)(api_n_cu() * occ)));
    static const bool st = getenv("T3HIP_STATIC_TILES") != nullptr;
    a.tile_ctr = st ? nullptr : api_ticket_counters(s, 2); a.n_classes = std::min<uint32_t>(8u, grid);
    if (tl_hdr.pending) {                                                                     // streaming entry: header check + verdict in this launch (as decode_fixed_fused)
        static const bool no_fold = getenv("T3HIP_HDR_KERNEL") != nullptr;
        if (a.tile_ctr && !no_fold && tl_hdr.hs <= 96u && ((uintptr_t)tl_hdr.in & 3u) == 0) {
            tl_hdr.pending = false;
            a.verdict = tl_hdr.verdict; a.hdr_in = tl_hdr.in; a.hdr_n = tl_hdr.hs; memcpy(a.hx, tl_hdr.ex.b, 96);
            a.fail = a.tile_ctr + 64u * a.n_classes + 16u;
        } else { const int rc = hdr_flush(s); if (rc) return rc; }
    }
    void* args[] = {(void*)&a};
    HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(512), args, a.lds_bytes, s));
    HIPCHK(hipLaunchKernel((const void*)uep_edge_kernel, dim3((3u * a.n_tiles + 1u + 255u) / 256u), dim3(256), args, 0, s));
    return T3_OK;
}

// Two-kernel FIXED decode (t3_decode_stream.hip): any per-band k, 1-D or 2-D.  Returns T3_OK after launching, 1 if not applicable.
int decode_fixed_stream(const uint8_t* body, uint64_t body_bytes, uint32_t hdr_syms, const t3_cfg& cfg, const t3_layout& L, const ScrCycle& sc,
                        void* d_out, uint64_t units, int to_pixels, uint32_t* d_fail, hipStream_t s) {
    if (L.n_raw_words == 0 || L.n_sym + (1u << 20) >= (1ull << 32) || body_bytes + hdr_syms >= (1ull << 32)) return 1;
    if (getenv("T3HIP_GENERIC_DECODE") != nullptr) return 1;
    const bool il = L.interleave2d != 0;
    std::lock_guard<std::mutex> lk(g_tab_mu);
    DecStArgs a; memset(&a, 0, sizeof a);
    // bands grouped by k, in order of first appearance
    int gk[kStMaxGrp]; uint32_t gn[kStMaxGrp] = {0, 0, 0, 0}; int ng = 0;
    for (int b = 0; b < 9; ++b) {
        int g = -1;
        for (int q = 0; q < ng; ++q) if (gk[q] == L.band_k[b]) g = q;
        if (g < 0) { g = ng++; gk[g] = L.band_k[b]; }
        a.grp[g].bands[gn[g]++] = (uint8_t)b;
    }
    uint32_t lcm = 1;
    for (int g = 0; g < ng; ++g) { const int rc = ensure_fx_tables(gk[g]); if (rc) return rc; uint32_t x = lcm, y = (uint32_t)gk[g]; while (y) { const uint32_t t = x % y; x = y; y = t; } lcm = lcm / x * (uint32_t)gk[g]; }
    // tile = 9 Lq stream symbols, Lq = lcm m: the m that fills the eight waves best with the tile's symbols within 16 KiB of LDS
    uint32_t best_m = 0; double best = -1.0;
    for (uint32_t m = 1; 9u * lcm * m <= 16384u; ++m) {
        uint32_t slots = 0, items = 0;
        for (int g = 0; g < ng; ++g) { const uint32_t n = gn[g] * (lcm * m / (uint32_t)gk[g]); items += n; slots += (n + 63u) / 64u; }
        const double fill = (double)items / (512.0 * ((slots + 7u) / 8u));
        if (fill > best + 1e-9) { best = fill; best_m = m; }
    }
    if (!best_m) best_m = 1;                                                // four different k: one lcm is already a large tile (LDS checked below)
    const uint32_t Lq = lcm * best_m;
    a.TS = 9u * Lq; a.n_sym = (uint32_t)L.n_sym; a.hdr_syms = hdr_syms; a.n_grp = (uint32_t)ng;
    uint32_t off = kFxLut, slots = 0; uint64_t tiles = 0;
    for (int g = 0; g < ng; ++g) {
        const int ki = k_index(gk[g]);
        auto& G = a.grp[g];
        G.r = 26u - (uint32_t)gk[g]; G.nb = Lq / (uint32_t)gk[g]; G.n_items = gn[g] * G.nb; G.wave0 = slots; G.n_waves = (G.n_items + 63u) / 64u; slots += G.n_waves;
        G.lut = d_synd_lut[ki]; G.lut_bytes = synd_lut_bytes[ki]; G.lut_off = off; off = (off + G.lut_bytes + 15u) & ~15u; G.roots = d_roots[ki];
        for (uint32_t i = 0; i < gn[g]; ++i) tiles = std::max<uint64_t>(tiles, (L.band_blocks[G.bands[i]] + G.nb - 1) / G.nb);
    }
    a.n_slots = slots; a.n_tiles = (uint32_t)tiles;
    a.fma = d_fma; a.fma_off = off; a.y_off = (off + 19696u + 15u) & ~15u; a.lds_bytes = a.y_off + a.TS + 64u;
    if (a.lds_bytes > 150u * 1024u) return 1;
    a.in = body; a.in_bytes = body_bytes; a.fail = d_fail; a.tab = d_fxtab;
    for (int b = 0; b < 9; ++b) { a.band_blocks[b] = (uint32_t)L.band_blocks[b]; a.band_body_off[b] = L.band_body_off[b]; a.band_boff6[b] = (uint32_t)((L.band_body_off[b] + 4) % 6); }
    a.cyc24 = sc.cyc24; a.pre0 = sc.pre[0]; a.pre1 = sc.pre[1];
    void* d_y; int rc = api_scratch(3, L.n_sym + 64, &d_y, s); if (rc) return rc;
    a.ystream = (uint8_t*)d_y;
    int occ = 1; rc = occupancy_of((const void*)decode_stream_kernel, 512, a.lds_bytes, &occ); if (rc) return rc;
    {
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(a.n_tiles, (uint32_t)(api_n_cu() * occ)));
        void* args[] = {(void*)&a};
        HIPCHK(hipLaunchKernel((const void*)decode_stream_kernel, dim3(grid), dim3(512), args, a.lds_bytes, s));
    }
    EmitStArgs e; memset(&e, 0, sizeof e);
    e.ystream = (const uint8_t*)d_y; e.n_sym = (uint32_t)L.n_sym; e.out = d_out; e.n_units = units;
    e.span = 52u * 512u; e.n_steps = (uint32_t)((L.n_sym + e.span - 1) / e.span);
    e.il_on = il ? 1 : 0;
    if (il) {
        const uint64_t A = (uint64_t)cfg.tile_w * cfg.tile_h;
        e.il_w = cfg.tile_w; e.il_A = (uint32_t)std::min<uint64_t>(A, L.n_sym);
        e.div_A = to_dev(fastdiv(e.il_A)); e.div_w = to_dev(fastdiv(e.il_w));
        e.il_fast = (e.il_w % 16u == 0 && (e.il_A % 16u == 0 || e.il_A == L.n_sym)) ? 1 : 0;
    }
    e.sym_off = 0; e.o_off = (e.span + 64u + 15u) & ~15u;
    e.lds_bytes = e.o_off + (to_pixels ? 0u : (e.span / 26u) * 27u + 64u);
    const void* fn = to_pixels ? (const void*)emit_stream_kernel<true> : (const void*)emit_stream_kernel<false>;
    rc = occupancy_of(fn, 512, e.lds_bytes, &occ); if (rc) return rc;
    {
        const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(e.n_steps, (uint32_t)(api_n_cu() * occ)));
        void* args[] = {(void*)&e};
        HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(512), args, e.lds_bytes, s));
    }
    return T3_OK;
}

// body decode with a known config (header already parsed)
int decode_body(const void* d_in, uint64_t n_in, const t3_cfg& cfg, uint64_t n_raw, const uint8_t next[3],
                void* d_out, uint64_t cap_units, uint64_t* n_out, int to_pixels, uint32_t* d_fail, hipStream_t s) {
    const bool fixed = cfg.mode == T3_MODE_FIXED;
    // to_pixels == 2: RGB8 out (row f1 fused into the pixel decoder's output stage); only the fused FIXED kernel takes it, every
    // other framing answers 1 and the caller converts a pixel scratch with the bridge kernel
    const bool want_rgb = to_pixels == 2;
    if (want_rgb && !fixed) return 1;
    DecArgs a; memset(&a, 0, sizeof a);
    EmitArgs e; memset(&e, 0, sizeof e);
    a.in = (const uint8_t*)d_in; a.fail = d_fail; a.tab = api_tables(); a.fixed = fixed ? 1 : 0;
    const ScrCycle sc = scrambler_cycle_from_next(next, cfg.seed_s0);
    a.cyc24 = sc.cyc24; a.pre0 = sc.pre[0]; a.pre1 = sc.pre[1];
    a.beacon_on = (cfg.beacon_enabled && cfg.beacon_words_period > 0) ? 1 : 0; a.period = cfg.beacon_words_period; a.slot = cfg.beacon_band_slot;
    uint64_t use_syms, n_words, total = 0;
    if (!fixed) {
        DecLayoutCompat D; plan_decode_compat(n_in, cfg, D);
        a.hdr_syms = 54;
        for (int b = 0; b < 9; ++b) { a.band_k[b] = D.band_k[b]; a.band_blocks[b] = D.band_blocks[b]; a.band_first[b] = total; a.band_off[b] = D.use_off[b]; total += D.band_blocks[b]; }
        use_syms = D.use_syms; n_words = D.out_words;
    } else {
        t3_layout L; int rc = plan(n_raw, cfg, L); if (rc) return T3_E_HEADER;
        if (L.out_words > n_in) return T3_E_HEADER;                      // truncated stream
        a.hdr_syms = 90; a.n_sym = L.n_sym;
        for (int b = 0; b < 9; ++b) { a.band_k[b] = L.band_k[b]; a.band_blocks[b] = L.band_blocks[b]; a.band_first[b] = total; a.band_off[b] = L.band_body_off[b]; total += L.band_blocks[b]; }
        use_syms = L.n_sym; n_words = n_raw;
        const uint64_t funits = want_rgb ? cap_units : to_pixels ? 2 * n_words : n_words;     // RGB: exactly the caller's pixel count (no pad pixel)
        if (want_rgb && (cap_units > 2 * n_words || cap_units + 1 < 2 * n_words || L.interleave2d)) return 1;
        if (want_rgb) { for (int b = 1; b < 9; ++b) if (L.band_k[b] != L.band_k[0]) return 1; }
        if (funits <= cap_units && (!to_pixels || ((uintptr_t)d_out & 15u) == 0) && getenv("T3HIP_GENERIC_DECODE") == nullptr && 9 * n_in < (1ull << 32)) {
            // the fully fused kernel where it applies (a beacon is stepped over in its loads); else a beacon is stripped by its own pass,
            // and the two-kernel path takes the rest
            const uint8_t* body = (const uint8_t*)d_in; uint64_t body_bytes = 9 * n_in; uint32_t hs = L.header_syms;
            const bool bcn_ok = L.beacon_on && cfg.beacon_band_slot < 9 && cfg.beacon_words_period >= 2 && cfg.beacon_words_period < (1u << 27) && getenv("T3HIP_BEACON_PASS") == nullptr;
            int frc = 1;
            if (bcn_ok) frc = decode_fixed_fused(body, body_bytes, hs, L, sc, d_out, funits, to_pixels, d_fail, s, cfg.beacon_band_slot, cfg.beacon_words_period);
            if (frc == 1) {
                if (L.beacon_on) {
                    void* d_b; int brc = api_scratch(2, L.body_syms + 64, &d_b, s); if (brc) return brc;
                    DebeaconArgs d; d.framed = (const uint8_t*)d_in + L.header_syms; d.framed_bytes = 9 * n_in - L.header_syms; d.body = (uint8_t*)d_b; d.body_syms = L.body_syms; d.period = cfg.beacon_words_period; d.slot = cfg.beacon_band_slot;
                    if (L.body_syms) { hipLaunchKernelGGL(debeacon_kernel, dim3(grid_for((L.body_syms + 15) / 16, 256)), dim3(256), 0, s, d); HIPCHK(hipGetLastError()); }
                    body = (const uint8_t*)d_b; body_bytes = L.body_syms; hs = 0;
                }
                if (!bcn_ok) frc = decode_fixed_fused(body, body_bytes, hs, L, sc, d_out, funits, to_pixels, d_fail, s);
                if (frc == 1 && want_rgb) { const int hrc = hdr_flush(s); return hrc ? hrc : 1; }
                if (frc == 1 && to_pixels == 1) frc = decode_fixed_uep(body, body_bytes, hs, cfg, L, sc, d_out, funits, d_fail, s);
                if (frc == 1) { const int hrc = hdr_flush(s); if (hrc) return hrc; }   // the other paths: header kernel in front
                if (frc == 1) frc = decode_fixed_stream(body, body_bytes, hs, cfg, L, sc, d_out, funits, to_pixels, d_fail, s);
            }
            if (frc == T3_OK) { *n_out = funits; return T3_OK; }
            if (frc < 0) return frc;
        }
    }
    if (want_rgb) return 1;
    { const int hrc = hdr_flush(s); if (hrc) return hrc; }
    a.total_blocks = total;
    const uint64_t units = to_pixels ? 2 * n_words : n_words;
    *n_out = units;
    if (units > cap_units) return T3_E_CAPACITY;
    void* d_use; int rc = api_scratch(3, use_syms + 64, &d_use, s); if (rc) return rc;
    a.use = (uint8_t*)d_use;
    if (total) { hipLaunchKernelGGL(dec_gather_rs_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, a); HIPCHK(hipGetLastError()); }
    e.use = (const uint8_t*)d_use; e.use_syms = use_syms; e.out = d_out; e.n_words = n_words; e.to_pixels = to_pixels ? 1 : 0;
    const bool il = cfg.profile == T3_P5_RS26_22_2D && cfg.tile_w && cfg.tile_h && use_syms;   // OLD:1018
    e.il_on = il ? 1 : 0;
    if (il) {
        const uint64_t A = (uint64_t)cfg.tile_w * cfg.tile_h;
        e.il_w = cfg.tile_w; e.il_A = (uint32_t)std::min<uint64_t>(A, use_syms);
        e.div_A = to_dev(fastdiv(e.il_A)); e.div_w = to_dev(fastdiv(e.il_w));
    }
    if (n_words) { hipLaunchKernelGGL(dec_emit_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, s, e); HIPCHK(hipGetLastError()); }
    return T3_OK;
}

int read_header(const void* d_in, uint64_t n_in, int mode, t3_cfg* seen, uint64_t* n_raw, uint8_t next[3], hipStream_t s) {
    const uint64_t hw = mode == T3_MODE_FIXED ? 10 : 6;
    if (n_in < hw) return T3_E_HEADER;                                     // OLD:920
    // pinned mailbox: the 54/90 header bytes come back by a real asynchronous DMA (a pageable target costs a staging copy)
    std::lock_guard<std::recursive_mutex> lk(g_mail_mu);
    uint8_t*& h = h_mail;
    if (!h) HIPCHK(hipHostMalloc((void**)&h, 128, hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(h, d_in, hw * 9, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    uint8_t hb[96]; memcpy(hb, h, hw * 9);
    return header_parse(hb, n_in, mode, *seen, n_raw, next);
}
}  // namespace

namespace t3 {
void rgb_shutdown();
void decode_shutdown() {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto fr = [](auto*& p) { if (p) (void)hipFree(p); p = nullptr; };
    fr(d_zpow); fr(d_crc_acc); fr(d_crc_afrag); fr(d_crc_afrag4); fr(d_crc_afb); crc_afb_w = 0; fr(d_fxtab); fr(d_fma); fr(d_rgb_dq);
    if (h_mail) { (void)hipHostFree(h_mail); h_mail = nullptr; }
    if (h_flag) { (void)hipHostFree(h_flag); h_flag = nullptr; d_flag_map = nullptr; }
    rgb_shutdown();
    for (int i = 0; i < 4; ++i) { fr(d_synd_lut[i]); fr(d_roots[i]); fr(d_synd_afrag[i]); }
    fr(d_synd_T); fr(d_synd_T16); fr(d_fx2_small);
}
int decode_init(const RsTables*) {
    // Z[0]: one zero byte through the byte-wise register update; Z[j+1] = Z[j] o Z[j]
    std::vector<uint32_t> z((size_t)kCrcPows * 32);
    uint32_t tbl[256];
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int j = 0; j < 8; ++j) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); tbl[i] = c; }
    for (int i = 0; i < 32; ++i) { const uint32_t x = 1u << i; z[i] = tbl[x & 0xFF] ^ (x >> 8); }
    for (int j = 1; j < kCrcPows; ++j)
        for (int i = 0; i < 32; ++i) { uint32_t x = z[(size_t)(j - 1) * 32 + i], y = 0; for (int q = 0; q < 32; ++q) if (x >> q & 1u) y ^= z[(size_t)(j - 1) * 32 + q]; z[(size_t)j * 32 + i] = y; }
    HIPCHK(hipMalloc((void**)&d_zpow, z.size() * 4));
    HIPCHK(hipMemcpy(d_zpow, z.data(), z.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&d_crc_acc, 64));
    {   // matrix-core CRC: column (step s, lane half kh, dword d, byte t) of the bit matrix = the remainder, at the end of a 64-byte
        // chunk, of the single input bit that K slot carries; rows = register bits.  Lane l = m + 32 kh holds row m (t3_host.cpp build_mfma_encode)
        auto adv = [&](uint32_t r, uint32_t nzero) { while (nzero--) r = tbl[r & 0xFF] ^ (r >> 8); return r; };
        std::vector<uint32_t> af((size_t)22 * 64 * 4, 0u);
        for (int st = 0; st < 22; ++st) for (int kh = 0; kh < 2; ++kh) for (int d = 0; d < 4; ++d) for (int t = 0; t < 4; ++t) {
            uint32_t vec;
            if (st < 16) { const int bit = 4 * d + t, byte = 2 * st + (bit >> 3), o = 32 * kh + byte; vec = adv(tbl[1u << (bit & 7)], 63u - (uint32_t)o); }
            else vec = adv(1u << (8 * d + 4 * kh + t), st == 16 ? 2048u : 64u << (st - 17));   // 16: the running remainder 2048 bytes further on; 17..21: 64 * 2^b
            for (int m = 0; m < 32; ++m) if (vec >> m & 1u) af[((size_t)st * 64 + m + 32 * kh) * 4 + d] |= 1u << (8 * t);
        }
        HIPCHK(hipMalloc((void**)&d_crc_afrag, af.size() * 4));
        HIPCHK(hipMemcpy(d_crc_afrag, af.data(), af.size() * 4, hipMemcpyHostToDevice));
        // FP4 form (t3_crc_fp4.hip).  8 data slices: in step s the lane's input dword w (bytes 4 s .. 4 s + 3 of its 32) is fed as the four
        // dwords w & 0x11111111, w & 0x22222222, w & 0x44444444, (w >> 1) & 0x44444444 -- K slot pos = 8 j + p of lane half kh carries bit
        // 4 p + j of w, standing at nibble bit j (j < 3: FP4 0.5, 1.0, 2.0) or 2 (j = 3); the slice holds the reciprocal weight, the
        // product is 1.  The feedback slice and five "append 64 * 2^b bytes" slices: K slot 8 g + q (g, q < 4) of half kh carries the
        // remainder bit of accumulator e = 4 g + q, row (e & 3) + 8 (e >> 2) + 4 kh, as FP4 0.5 (weight 2.0); the other slots are unused.
        // FP4 e2m1: 0b0001 = 0.5, 0b0010 = 1.0, 0b0100 = 2.0.
        std::vector<uint32_t> a4((size_t)14 * 64 * 4, 0u);
        static const uint32_t recip[4] = {4u, 2u, 1u, 1u};                 // weight nibble for a bit at nibble bit 0, 1, 2, 2
        for (int st = 0; st < 14; ++st) for (int kh = 0; kh < 2; ++kh) for (int pos = 0; pos < 32; ++pos) {
            uint32_t vec, wt;
            if (st < 8) { const int j = pos >> 3, pn = pos & 7, bit = 4 * pn + j, o = 32 * kh + 4 * st + (bit >> 3); vec = adv(tbl[1u << (bit & 7)], 63u - (uint32_t)o); wt = recip[j]; }
            else if ((pos & 7) < 4) { const int e = 4 * (pos >> 3) + (pos & 7); vec = adv(1u << ((e & 3) + 8 * (e >> 2) + 4 * kh), st == 8 ? 2048u : 64u << (st - 9)); wt = 4u; }
            else continue;
            for (int m = 0; m < 32; ++m) if (vec >> m & 1u) a4[((size_t)st * 64 + m + 32 * kh) * 4 + (pos >> 3)] |= wt << (4 * (pos & 7));
        }
        HIPCHK(hipMalloc((void**)&d_crc_afrag4, a4.size() * 4));
        HIPCHK(hipMemcpy(d_crc_afrag4, a4.data(), a4.size() * 4, hipMemcpyHostToDevice));
    }
    {   // field tables of the fused decoder
        const Field& F = field();
        static const uint8_t want_exp[26] = {1, 3, 9, 5, 15, 23, 13, 17, 20, 4, 12, 14, 11, 2, 6, 18, 7, 21, 16, 26, 22, 10, 8, 24, 25, 19};
        if (memcmp(F.t.exp, want_exp, 26) != 0) return T3_E_ARG;              // kExp in t3_decode_fused.hip is this table
        FxTables T; memset(&T, 0, sizeof T);
        memcpy(T.mul, F.t.mul, 729); memcpy(T.add, F.t.add, 729); memcpy(T.inv, F.t.inv, 27); memcpy(T.neg, F.t.neg, 27); memcpy(T.exp, F.t.exp, 26);
        for (int x = 0; x < 27; ++x) for (int y = 0; y < 27; ++y) T.sub[x * 27 + y] = F.t.add[x * 27 + F.t.neg[y]];
        for (int st = 0; st < 3; ++st) for (int c = 0; c < 27; ++c) T.descr[st][c] = (uint8_t)(4 * F.t.add[c * 27 + F.t.neg[13 * st]]);
        HIPCHK(hipMalloc((void**)&d_fxtab, sizeof T));
        HIPCHK(hipMemcpy(d_fxtab, &T, sizeof T, hipMemcpyHostToDevice));
    }
    return T3_OK;
}
}  // namespace t3

extern "C" {

int t3hip_read_header_dev(const void* d_in, uint64_t n_in, int mode, t3_cfg* out_cfg, uint64_t* n_raw, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!out_cfg || (n_in && !d_in)) return T3_E_ARG;
    uint8_t next[3];
    return read_header(d_in, n_in, mode, out_cfg, n_raw, next, (hipStream_t)stream);
}

int t3hip_decode_body_dev(const void* d_in, uint64_t n_in, const t3_cfg* cfg, uint64_t n_raw, void* d_out, uint64_t cap, uint64_t* n_out,
                          int to_pixels, uint32_t* d_fail, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!cfg || !n_out || !d_fail) return T3_E_ARG;
    const ScrCycle sc = scrambler_cycle(cfg->seed_a, cfg->seed_b, cfg->seed_s0);
    return decode_body(d_in, n_in, *cfg, n_raw, sc.next, d_out, cap, n_out, to_pixels, d_fail, (hipStream_t)stream);
}

int t3hip_decode_frame_async(const void* d_in, uint64_t n_in, const t3_cfg* cfg, uint64_t n_raw, void* d_out, uint64_t cap, uint64_t* n_out,
                             int to_pixels, uint32_t* d_verdict, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!cfg || !n_out || !d_verdict || (n_in && !d_in) || cfg->profile == T3_RAW_MODE) return T3_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    uint8_t hdr[96]; memset(hdr, 0, sizeof hdr);
    t3_layout L; int rc = plan(n_raw, *cfg, L); if (rc) return rc;
    const uint32_t hs = (uint32_t)header_encode(*cfg, n_raw, hdr);
    if (9 * n_in < hs) return T3_E_HEADER;
    // the expected header symbols travel as a kernel argument (96 bytes): nothing to allocate, no limit on how many different
    // headers a long-lived process may see
    memcpy(tl_hdr.ex.b, hdr, 96); tl_hdr.hs = hs; tl_hdr.verdict = d_verdict; tl_hdr.in = (const uint8_t*)d_in; tl_hdr.pending = true;
    const ScrCycle sc = scrambler_cycle(cfg->seed_a, cfg->seed_b, cfg->seed_s0);
    rc = decode_body(d_in, n_in, *cfg, n_raw, sc.next, d_out, cap, n_out, to_pixels, d_verdict + 1, s);
    if (tl_hdr.pending) { tl_hdr.pending = false; if (rc == T3_OK) rc = T3_E_ARG; }       // (every decode path takes or flushes it; a path that forgot would leave the verdict unwritten)
    return rc;
}

int t3hip_decode_profile_dev(const void* d_in, uint64_t n_in, t3_cfg* seen, void* d_out, uint64_t cap, uint64_t* n_out, int to_pixels, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!seen || !n_out || (n_in && !d_in)) return T3_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    *n_out = 0;
    if (seen->profile == T3_RAW_MODE) {                                       // OLD:998-1002
        const uint64_t units = to_pixels ? 2 * n_in : n_in; *n_out = units;
        if (units > cap) return T3_E_CAPACITY;
        if (!n_in) return T3_OK;
        if (to_pixels) return t3hip_unpack_words_dev(d_in, n_in, d_out, stream);
        HIPCHK(hipMemcpyAsync(d_out, d_in, n_in * 9, hipMemcpyDeviceToDevice, s));
        return T3_OK;
    }
    uint64_t n_raw = 0; uint8_t next[3];
    int rc = read_header(d_in, n_in, seen->mode, seen, &n_raw, next, s);
    if (rc) return rc;
    // failure counter in mapped pinned host memory: written only by lanes that give up on a block, read after the sync
    // without a copy (the previous synchronous call has drained, so the host may clear it directly)
    std::lock_guard<std::recursive_mutex> lk(g_mail_mu);
    if (!h_flag) {
        HIPCHK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&d_flag_map, h_flag, 0));
    }
    *(volatile uint32_t*)h_flag = 0;
    rc = decode_body(d_in, n_in, *seen, n_raw, next, d_out, cap, n_out, to_pixels, d_flag_map, s);
    if (rc) { if (rc != T3_E_CAPACITY) *n_out = 0; return rc; }
    HIPCHK(hipStreamSynchronize(s));
    if (*(volatile uint32_t*)h_flag) { *n_out = 0; return T3_E_RS; }           // OLD:987,1017: false, out stays empty
    return T3_OK;
}

// Pipelined host decode (round 3; see encode_host_pipelined in t3_api.cpp): FIXED, one k on all bands, 1-D, no beacon, pixels out, a frame
// of many tiles.  The header is parsed on the host from the caller's buffer; the caller's thread uploads the nine band runs of chunk c
// and launches the fused decoder on those tiles, a helper thread downloads the pixels of chunk c - 1 meanwhile.  1: not applicable.
static int decode_host_pipelined(const void* in, uint64_t n_in, t3_cfg* seen, void* out, uint64_t cap, uint64_t* n_out, void* di, void* dout) {
    if (seen->mode != T3_MODE_FIXED || seen->profile == T3_RAW_MODE || n_in < 10 || getenv("T3HIP_SERIAL_HOST") != nullptr || getenv("T3HIP_GENERIC_DECODE") != nullptr) return 1;
    t3_cfg cfg = *seen; uint64_t n_raw = 0; uint8_t next[3];
    if (header_parse((const uint8_t*)in, n_in, T3_MODE_FIXED, cfg, &n_raw, next) != T3_OK) return 1;    // (the serial path reports it)
    t3_layout L; if (plan(n_raw, cfg, L) != T3_OK || L.out_words > n_in) return 1;
    if (L.interleave2d || L.beacon_on || 9 * n_in >= (1ull << 32)) return 1;
    for (int b = 1; b < 9; ++b) if (L.band_k[b] != L.band_k[0]) return 1;
    const uint64_t units = 2 * n_raw;
    if (units > cap) return 1;
    const ScrCycle sc = scrambler_cycle_from_next(next, cfg.seed_s0);
    const uint32_t hs = L.header_syms, nb = (uint32_t)T3_DEC_PX_NB;
    uint64_t maxb = 0; for (int b = 0; b < 9; ++b) maxb = std::max<uint64_t>(maxb, L.band_blocks[b]);
    const uint32_t n_tiles = (uint32_t)((maxb + nb - 1) / nb), units_tile = (9u * nb * (uint32_t)L.band_k[0] / 13u) * 3u;
    static const uint32_t want_env = getenv("T3HIP_HOST_CHUNKS") ? (uint32_t)atoi(getenv("T3HIP_HOST_CHUNKS")) : 0u;
    const uint32_t want = want_env ? want_env : 6u;                                  // (FIXED streams start 90 symbols in: the band runs are 2-byte aligned, a strided copy of them is slow -- nine plain copies per chunk, few chunks; t3_api.cpp)
    if (n_tiles < 64u) return 1;
    const uint32_t per = (n_tiles + want - 1u) / want, n_chunks = (n_tiles + per - 1u) / per;
    hipStream_t s = api_stream(), s2 = nullptr; hipEvent_t* evs = nullptr;
    { const int rc = api_pipeline(n_chunks, &s2, &evs); if (rc) return rc; }
    *seen = cfg;                                                                       // OLD:1006-1013: the header decoded
    std::lock_guard<std::recursive_mutex> lk(g_mail_mu);
    if (!h_flag) { HIPCHK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped)); HIPCHK(hipHostGetDevicePointer((void**)&d_flag_map, h_flag, 0)); }
    *(volatile uint32_t*)h_flag = 0;
    std::atomic<uint32_t> launched{0}; std::atomic<int> abort_dl{0};
    hipError_t dl_err = hipSuccess; const int dev = api_device();
    uint8_t* const ho = (uint8_t*)out; const uint8_t* const dob = (const uint8_t*)dout;
    std::thread dl([&] {
        if (hipSetDevice(dev) != hipSuccess) { dl_err = hipErrorInvalidDevice; return; }
        for (uint32_t c = 0; c < n_chunks; ++c) {
            while (launched.load(std::memory_order_acquire) <= c) { if (abort_dl.load()) return; std::this_thread::yield(); }
            hipError_t er = hipEventSynchronize(evs[c]);
            const uint64_t u0 = std::min<uint64_t>(units, (uint64_t)c * per * units_tile), u1 = std::min<uint64_t>(units, ((uint64_t)c * per + per) * units_tile);
            if (er == hipSuccess && u1 > u0) er = hipMemcpyAsync(ho + 6 * u0, dob + 6 * u0, 6 * (u1 - u0), hipMemcpyDeviceToHost, s2);
            if (er != hipSuccess) { dl_err = er; return; }
        }
        dl_err = hipStreamSynchronize(s2);
    });
    int rc = T3_OK;
    {   // header words: the device copy of the stream starts with them (the kernel itself never reads them)
        const hipError_t er = hipMemcpyAsync(di, in, hs, hipMemcpyHostToDevice, s); if (er != hipSuccess) rc = api_fail_hip(er, "hipMemcpyAsync(header)");
    }
    for (uint32_t c = 0; c < n_chunks && rc == T3_OK; ++c) {
        const uint32_t t0 = c * per, t1 = std::min<uint32_t>(n_tiles, t0 + per);
        bool even = (uint64_t)t1 * nb <= L.band_blocks[0];
        for (int b = 1; b < 9; ++b) even = even && L.band_blocks[b] == L.band_blocks[0];
        const uint64_t o2 = hs + L.band_body_off[0] + 26ull * t0 * nb, wbytes = 26ull * (t1 - t0) * nb, pitch = 26 * L.band_blocks[0];
        if (even && o2 % 4u == 0 && wbytes % 4u == 0 && pitch % 4u == 0 && !getenv("T3HIP_NO_2D_COPY")) {      // nine equally long, 4-byte aligned band runs: one strided copy
            const hipError_t er = hipMemcpy2DAsync((uint8_t*)di + o2, pitch, (const uint8_t*)in + o2, pitch, wbytes, 9, hipMemcpyHostToDevice, s);
            if (er != hipSuccess) rc = api_fail_hip(er, "hipMemcpy2DAsync(band runs)");
        } else for (int b = 0; b < 9 && rc == T3_OK; ++b) {
            const uint64_t lo = std::min<uint64_t>(L.band_blocks[b], (uint64_t)t0 * nb), hi = std::min<uint64_t>(L.band_blocks[b], (uint64_t)t1 * nb);
            // (a lane's 16-byte load of a block's second half reaches 0 bytes past the block: runs are exact)
            if (hi > lo) { const hipError_t er = hipMemcpyAsync((uint8_t*)di + hs + L.band_body_off[b] + 26 * lo, (const uint8_t*)in + hs + L.band_body_off[b] + 26 * lo, 26 * (hi - lo), hipMemcpyHostToDevice, s); if (er != hipSuccess) rc = api_fail_hip(er, "hipMemcpyAsync(band run)"); }
        }
        if (rc == T3_OK) {
            rc = decode_fixed_fused((const uint8_t*)di, 9 * n_in, hs, L, sc, dout, units, 1, d_flag_map, s, 0, 0, t0, t1);
            if (rc == 1) rc = T3_E_ARG;
        }
        if (rc == T3_OK) { const hipError_t er = hipEventRecord(evs[c], s); if (er != hipSuccess) rc = api_fail_hip(er, "hipEventRecord"); }
        if (rc == T3_OK) launched.store(c + 1, std::memory_order_release);
    }
    if (rc != T3_OK) abort_dl.store(1);
    dl.join();
    if (rc == T3_OK && dl_err != hipSuccess) rc = api_fail_hip(dl_err, "chunk download");
    if (rc == T3_OK) { const hipError_t er = hipStreamSynchronize(s); if (er != hipSuccess) rc = api_fail_hip(er, "hipStreamSynchronize"); }
    if (rc != T3_OK) return rc;
    *n_out = units;
    if (*(volatile uint32_t*)h_flag) { *n_out = 0; return T3_E_RS; }                 // OLD:987,1017
    return T3_OK;
}

static int decode_host(const void* in, uint64_t n_in, t3_cfg* seen, void* out, uint64_t cap, uint64_t* n_out, int to_pixels) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!seen || !n_out || (n_in && !in)) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());          // one caller at a time on the shared stream and scratch
    void *di, *dout; int rc = api_scratch(0, n_in * 9 + 64, &di); if (rc) return rc;
    const uint64_t unit = to_pixels ? 6 : 9, dcap = (to_pixels ? 2 : 1) * (n_in + 16);
    rc = api_scratch(1, dcap * unit + 64, &dout); if (rc) return rc;
    hipStream_t s = api_stream();
    if (to_pixels == 1) { rc = decode_host_pipelined(in, n_in, seen, out, cap, n_out, di, dout); if (rc != 1) return rc; }   // 1: not that framing -> one upload, the kernels, one download
    if (n_in) HIPCHK(hipMemcpyAsync(di, in, n_in * 9, hipMemcpyHostToDevice, s));
    rc = t3hip_decode_profile_dev(di, n_in, seen, dout, dcap, n_out, to_pixels, s);
    if (rc) return rc;
    if (*n_out > cap) return T3_E_CAPACITY;
    if (*n_out) HIPCHK(hipMemcpyAsync(out, dout, *n_out * unit, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return T3_OK;
}
int t3hip_decode_profile(const void* in, uint64_t n_in, t3_cfg* seen, void* out, uint64_t cap, uint64_t* n_out) { return decode_host(in, n_in, seen, out, cap, n_out, 0); }
int t3hip_decode_frame(const void* in, uint64_t n_in, t3_cfg* seen, void* px, uint64_t cap_px, uint64_t* n_px) { return decode_host(in, n_in, seen, px, cap_px, n_px, 1); }

int t3hip_rs_decode_blocks_dev(int k, int mode, uint8_t* d_code, uint64_t n_blocks, uint8_t* d_data, uint8_t* d_ok, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!valid_k(k) || mode < 0 || mode > 1) return T3_E_ARG;
    if (!n_blocks) return T3_OK;
    if (!d_code || !d_data || !d_ok) return T3_E_ARG;
    hipLaunchKernelGGL(rs_decode_blocks_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_code, n_blocks, k, mode, api_tables(), d_data, d_ok);
    HIPCHK(hipGetLastError()); return T3_OK;
}

int t3hip_inject_errors_dev(void* d_words, uint64_t first_sym, uint64_t n_blocks, uint32_t seed, int max_err, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (max_err < 0 || max_err > 26) return T3_E_ARG;
    if (!n_blocks) return T3_OK;
    if (!d_words) return T3_E_ARG;
    hipLaunchKernelGGL(inject_errors_kernel, dim3(grid_for(n_blocks, 256)), dim3(256), 0, (hipStream_t)stream, (uint8_t*)d_words + first_sym, n_blocks, seed, max_err);
    HIPCHK(hipGetLastError()); return T3_OK;
}

// the CRC's leading 0xFFFFFFFF carried through n zero bytes (bitwise, 8 n steps would be too slow: square-and-multiply on the operator)
static uint32_t crc_lead(uint64_t n_bytes) {
    static thread_local uint64_t last_n = ~0ull; static thread_local uint32_t last_x = 0;   // (a stream of frames of one size asks the same question every frame)
    if (n_bytes == last_n) return last_x;
    uint32_t x = 0xFFFFFFFFu;
    uint32_t op[32], sq[32];                                          // operator "append 2^j zero bytes" as 32 columns; start with one zero byte
    for (int b = 0; b < 32; ++b) { uint32_t v = 1u << b; for (int i = 0; i < 8; ++i) v = (v & 1u) ? (0xEDB88320u ^ (v >> 1)) : (v >> 1); op[b] = v; }
    auto apply = [](const uint32_t* m, uint32_t v) { uint32_t r = 0; for (int b = 0; b < 32; ++b) if (v >> b & 1u) r ^= m[b]; return r; };
    for (uint64_t n = n_bytes; n; n >>= 1) {
        if (n & 1u) x = apply(op, x);
        for (int b = 0; b < 32; ++b) sq[b] = apply(op, op[b]);
        memcpy(op, sq, sizeof op);
    }
    last_n = n_bytes; last_x = x;
    return x;
}

// FP4 CRC with strided rounds (t3_crc_fp4.hip): wave g owns rounds g, g + W, ...; a column's running remainder re-enters 2048 W bytes
// further on.  A slice has the layout of slice 8 of the FP4 slices (decode_init).  Built once per context for W = slots >> l, l = 0 .. kCrcStrideLevels - 1 (shorter streams
// use fewer waves); the operators by square-and-multiply, crc_lead's way.
constexpr int kCrcStrideLevels = 8;
static uint32_t crc_stride_w(uint32_t slots, int l) { return std::max(4u, (slots >> l) & ~3u); }
static int ensure_crc_feedback(uint32_t slots) {
    std::lock_guard<std::mutex> lk(g_tab_mu);
    if (d_crc_afb && crc_afb_w == (uintptr_t)slots) return T3_OK;
    auto apply = [](const uint32_t* m, uint32_t v) { uint32_t r = 0; for (int b = 0; b < 32; ++b) if (v >> b & 1u) r ^= m[b]; return r; };
    std::vector<uint32_t> a4((size_t)kCrcStrideLevels * 64 * 4, 0u);
    for (int l = 0; l < kCrcStrideLevels; ++l) {
        uint32_t op[32], sq[32], acc[32];                             // op = "append 2^j bytes", acc = the operator of the bits of 2048 W seen so far
        for (int b = 0; b < 32; ++b) { uint32_t v = 1u << b; for (int i = 0; i < 8; ++i) v = (v & 1u) ? (0xEDB88320u ^ (v >> 1)) : (v >> 1); op[b] = v; acc[b] = 1u << b; }
        for (uint64_t n = 2048ull * crc_stride_w(slots, l); n; n >>= 1) {
            if (n & 1u) for (int b = 0; b < 32; ++b) acc[b] = apply(op, acc[b]);
            for (int b = 0; b < 32; ++b) sq[b] = apply(op, op[b]);
            memcpy(op, sq, sizeof op);
        }
        for (int kh = 0; kh < 2; ++kh) for (int e = 0; e < 16; ++e) {     // accumulator e -> dword e >> 2, nibble e & 3, weight 2.0
            const uint32_t vec = acc[(e & 3) + 8 * (e >> 2) + 4 * kh];
            for (int m = 0; m < 32; ++m) if (vec >> m & 1u) a4[(((size_t)l * 64) + m + 32 * kh) * 4 + (e >> 2)] |= 4u << (4 * (e & 3));
        }
    }
    if (!d_crc_afb) HIPCHK(hipMalloc((void**)&d_crc_afb, a4.size() * 4));
    else HIPCHK(hipDeviceSynchronize());                                // (cannot happen: slots is a constant of the context)
    HIPCHK(hipMemcpy(d_crc_afb, a4.data(), a4.size() * 4, hipMemcpyHostToDevice));
    crc_afb_w = (uintptr_t)slots;
    return T3_OK;
}

// CRC + symbol-sum accumulation of a payload into acc[0] / acc[1] (zeroed here): whole 2 KiB rounds on the matrix cores
// when the buffer is 16-byte aligned, the rest (or everything) through the table kernel
// tail_off != null: a rest shorter than 2 KiB behind the matrix-core rounds is left to the caller's record kernel (*tail_off = where it starts)
// partials / cap_wg / n_partials (frame record): when the strided FP4 kernel takes the whole stream but its rest and its grid fits cap_wg, the
// workgroups store their contributions side by side in `partials` (*n_partials = how many) and the accumulators are left alone: no
// 8-byte fill kernel in front (4.5 us of stream time) and no atomics; the record kernel folds them.
constexpr uint32_t kRecordPartialWgs = 1024;
static int launch_crc(const uint8_t* d_data, uint64_t n_bytes, uint32_t* acc, hipStream_t s, uint64_t* tail_off = nullptr, uint32_t* partials = nullptr, uint32_t cap_wg = 0, uint32_t* n_partials = nullptr) {
    const bool fp4_whole = ((uintptr_t)d_data & 15u) == 0 && n_bytes >= 64 * 2048 && (n_bytes >> 11) < (1ull << 32) && n_bytes < (1ull << kCrcPows) && getenv("T3HIP_CRC_TABLES") == nullptr &&
                           getenv("T3HIP_CRC_I8") == nullptr && getenv("T3HIP_CRC_BLOCKED") == nullptr && tail_off != nullptr;
    const bool use_partials = partials && n_partials && fp4_whole && getenv("T3HIP_CRC_ATOMICS") == nullptr;   // (T3HIP_CRC_ATOMICS: measurement / test knob)
    if (n_partials) *n_partials = 0;
    if (!use_partials) HIPCHK(hipMemsetAsync(acc, 0, 8, s));
    uint64_t done = 0;
    static const int rpw_env = [] { const char* e = getenv("T3HIP_CRC_ROUNDS_PER_WAVE"); return e ? atoi(e) : 0; }();
    if (((uintptr_t)d_data & 15u) == 0 && n_bytes >= 64 * 2048 && (n_bytes >> 11) < (1ull << 32) && n_bytes < (1ull << kCrcPows) && getenv("T3HIP_CRC_TABLES") == nullptr) {   // (the epilogue walks the distance bit by bit over kCrcPows operators)
        CrcMArgs m; memset(&m, 0, sizeof m);
        m.data = d_data; m.n_bytes = n_bytes; m.n_rounds = (uint32_t)(n_bytes >> 11);
        // Four-wave workgroups, two waves per SIMD over the whole chip: a wave needs ~100 VGPRs (the bit matrix), which is what a
        // SIMD has left beside the decoder's six waves, so the kernel can start under the decode instead of behind it (16-wave
        // workgroups had to wait for the decoder's persistent workgroups to drain: +0.1 ms per step).  At least 8 rounds per wave.
        static const int wps = [] { const char* e = getenv("T3HIP_CRC_WAVES_PER_SIMD"); const int v = e ? atoi(e) : 2; return v > 0 ? v : 2; }();
        const uint64_t slots = (uint64_t)api_n_cu() * 4 * (uint64_t)wps;
        m.rounds_per_wave = rpw_env > 0 ? (uint32_t)rpw_env : (uint32_t)std::max<uint64_t>(8, (m.n_rounds + slots - 1) / slots);
        m.afrag = d_crc_afrag; m.zpow = d_zpow; m.chunk_crc = acc; m.sym_sum = acc + 1;
        const uint64_t waves = ((uint64_t)m.n_rounds + m.rounds_per_wave - 1) / m.rounds_per_wave;
        const bool use_i8 = getenv("T3HIP_CRC_I8") != nullptr;                   // measurement / test knob: the i8 form (t3_crc_mfma.hip)
        const bool blocked = getenv("T3HIP_CRC_BLOCKED") != nullptr;             // measurement knob: round-2 assignment (consecutive rounds per wave)
        if (use_i8) hipLaunchKernelGGL(crc_mfma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, m);
        else if (blocked) { m.afrag = d_crc_afrag4; hipLaunchKernelGGL(crc_fp4_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, m); }
        else {
            // strided rounds: W = every wave slot of the chip, halved for shorter streams until a wave has at least 8 rounds
            { const int rc = ensure_crc_feedback((uint32_t)slots); if (rc) return rc; }
            int l = 0;
            while (l + 1 < kCrcStrideLevels && (uint64_t)crc_stride_w((uint32_t)slots, l) * 8 > m.n_rounds) ++l;
            const uint32_t W = crc_stride_w((uint32_t)slots, l);
            m.afrag = d_crc_afrag4; m.afb = d_crc_afb + (size_t)l * 64 * 4; m.stride_waves = W;
            if (use_partials && W / 4 <= cap_wg) { m.partials = partials; *n_partials = W / 4; }
            else if (use_partials) HIPCHK(hipMemsetAsync(acc, 0, 8, s));      // (cannot happen with the scratch size t3hip_frame_record_scratch_bytes asks for)
            hipLaunchKernelGGL(crc_fp4_kernel, dim3(W / 4), dim3(256), 0, s, m);
        }
        HIPCHK(hipGetLastError());
        done = (uint64_t)m.n_rounds << 11;
    }
    if (tail_off) { *tail_off = n_bytes; if (done && n_bytes - done < 2048) { *tail_off = done; return T3_OK; } }
    if (done < n_bytes) {
        CrcArgs c; memset(&c, 0, sizeof c);
        c.data = d_data + done; c.n_bytes = n_bytes - done; c.chunk_bytes = 2304;      // 256 words per lane
        c.n_chunks = (uint32_t)((c.n_bytes + c.chunk_bytes - 1) / c.chunk_bytes);
        c.chunk_crc = acc; c.sym_sum = acc + 1; c.zpow = d_zpow;
        hipLaunchKernelGGL(crc_chunks_kernel, dim3((c.n_chunks + 255) / 256), dim3(256), 0, s, c); HIPCHK(hipGetLastError());
    }
    return T3_OK;
}

uint64_t t3hip_frame_record_scratch_bytes(uint64_t) { return 64 + 8ull * kRecordPartialWgs; }   // two accumulators | one (xor, sum) per CRC workgroup; 64 bytes still work (accumulators + atomics)
int t3hip_frame_record_dev(const void* d_words, uint64_t n_words, uint64_t frame_idx, const t3_cfg* cfg, t3_frame_record* d_rec,
                           void* d_scratch, uint64_t scratch_bytes, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!cfg || !d_rec || (n_words && !d_words) || !d_scratch || scratch_bytes < 8) return T3_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t n_bytes = 9 * n_words; uint64_t tail_off = n_bytes;
    uint32_t* parts = scratch_bytes >= 64 + 8 ? (uint32_t*)((uint8_t*)d_scratch + 64) : nullptr; uint32_t n_parts = 0;
    const uint32_t cap_wg = parts ? (uint32_t)std::min<uint64_t>((scratch_bytes - 64) / 8, kRecordPartialWgs) : 0u;
    { const int rc = launch_crc((const uint8_t*)d_words, n_bytes, (uint32_t*)d_scratch, s, &tail_off, parts, cap_wg, &n_parts); if (rc) return rc; }
    hipLaunchKernelGGL(frame_record_kernel, dim3(1), dim3(64), 0, s, (const uint32_t*)d_scratch, crc_lead(n_bytes),
                       tail_off < n_bytes ? (const uint8_t*)d_words + tail_off : (const uint8_t*)nullptr, (uint32_t)(n_bytes - tail_off), (const uint32_t*)d_zpow,
                       (const uint8_t*)d_words, n_words, frame_idx, (uint32_t)cfg->profile, (uint32_t)cfg->mode, (void*)d_rec, (const uint32_t*)parts, n_parts);
    HIPCHK(hipGetLastError()); return T3_OK;
}

int t3hip_crc32_dev(const void* d_data, uint64_t n_bytes, uint32_t* crc_out, void* stream) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!crc_out || (n_bytes && !d_data)) return T3_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::recursive_mutex> lk(g_mail_mu);                       // one scratch accumulator per process
    void* d_sc = d_crc_acc;
    { const int rc = launch_crc((const uint8_t*)d_data, n_bytes, (uint32_t*)d_sc, s); if (rc) return rc; }
    uint32_t acc = 0;
    HIPCHK(hipMemcpyAsync(&acc, d_sc, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // the kernel leaves the xor of the chunk remainders moved to the end of the stream; the leading 0xFFFFFFFF travels
    // through n_bytes zero bytes on the host (bitwise, 8 n steps would be too slow: square-and-multiply on the operator)
    const uint32_t x = crc_lead(n_bytes);
    *crc_out = (x ^ acc) ^ 0xFFFFFFFFu;
    return T3_OK;
}

int t3hip_crc32(const void* data, uint64_t n_bytes, uint32_t* crc_out) {
    if (!api_ready()) return T3_E_NODEVICE;
    if (!crc_out || (n_bytes && !data)) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(api_host_mutex());
    std::lock_guard<std::recursive_mutex> lk(g_mail_mu);
    void* di; int rc = api_scratch(0, n_bytes + 64, &di); if (rc) return rc;
    hipStream_t s = api_stream();
    if (n_bytes) HIPCHK(hipMemcpyAsync(di, data, n_bytes, hipMemcpyHostToDevice, s));
    return t3hip_crc32_dev(di, n_bytes, crc_out, s);
}

int t3hip_index_assemble(t3_frame_record* recs, uint64_t n, uint64_t first_payload_offset) {
    if (n && !recs) return T3_E_ARG;
    std::sort(recs, recs + n, [](const t3_frame_record& x, const t3_frame_record& y) { return x.frame_idx < y.frame_idx; });
    uint64_t off = first_payload_offset;
    for (uint64_t i = 0; i < n; ++i) { recs[i].byte_offset = off; off += 9 * recs[i].n_words; }
    return T3_OK;
}

}  // extern "C"
