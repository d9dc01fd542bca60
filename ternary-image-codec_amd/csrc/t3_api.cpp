// t3_api.cpp — the C-ABI of libt3hip.so (include/t3hip.h): context, tile planning, kernel launches.
// Host logic only; all arithmetic on the data path happens in t3_kernels.hip / t3_decode.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/t3hip.h"
#include "t3_host.hpp"
#include "t3_kernels.h"
#include "t3_decode.h"

using namespace t3;

namespace {

struct LutImage { uint32_t* d_img = nullptr; uint32_t bytes = 0; uint32_t k_off[4] = {0, 0, 0, 0}; uint32_t* d_afrag = nullptr; };

struct Ctx {
    int dev = -1; bool ready = false; int n_cu = 256;
    hipStream_t stream = nullptr;                       // used by the host-buffer entry points
    RsTables* d_tab = nullptr;
    uint8_t* d_P[4][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    std::map<uint32_t, LutImage> luts;                  // key = kmask | mode << 8
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr}; size_t cap[4] = {0, 0, 0, 0};   // grow-only device scratch (slots 0, 1: host-buffer entry points)
    std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> sbuf;              // slots 2, 3: intermediates of the *_dev entry points, one set per caller stream
    uint32_t* d_ctr = nullptr; std::map<std::pair<hipStream_t, int>, uint32_t> ctr_slot;   // tile-ticket counters, one set per (stream, kernel kind) in use
    uint32_t* d_flag = nullptr;                         // failure counter for the synchronous decode entry points
    hipStream_t stream2 = nullptr;                      // the download side of the pipelined host entry points (created on first use)
    std::vector<hipEvent_t> chunk_ev;                   // ... and their per-chunk events
    std::string hip_err;
    std::mutex mu;
    // The host-buffer entry points share one stream and two scratch slots: each of them holds this for its whole upload ->
    // launch -> download -> synchronise sequence (two caller threads otherwise interleave on the stream and overwrite, or free,
    // each other's scratch).  Recursive: some of them are built from others.
    std::recursive_mutex host_mu;
    std::mutex tab_mu, qt_mu;                             // lazily built device tables of the decode / RGB halves (per context: contexts share no lock)
    std::recursive_mutex mail_mu;                         // pinned host mailboxes + CRC accumulator of the synchronous entry points
    void* slot[48] = {};                                  // device / pinned objects of the other translation units (api_slot), freed by their owners
};
// One context per GPU.  t3hip_init creates the process default; t3hip_create more (one per device for a host that drives a whole
// node from one process); t3hip_use binds the calling thread to one of them (thread-local), every entry point works on the
// calling thread's context.  `g` below IS that context.
Ctx g_null;                                               // stands in while no context exists: ready == false
Ctx* g_def = nullptr;
thread_local Ctx* tl_cur = nullptr;
inline Ctx* cur_ctx() { return tl_cur ? tl_cur : (g_def ? g_def : &g_null); }
#define g (*cur_ctx())

int fail_hip(hipError_t e, const char* what) { g.hip_err = std::string(what) + ": " + hipGetErrorString(e); return T3_E_HIP; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail_hip(e_, #x); } while (0)

int k_index(int k) { return k == 24 ? 0 : k == 22 ? 1 : k == 20 ? 2 : k == 18 ? 3 : -1; }
const int kOfIndex[4] = {24, 22, 20, 18};

// Tile-ticket counters of a persistent kernel: eight class counters + a done counter, 256 B apart, zero between launches (the kernel's
// last workgroup re-zeroes them).  One set per (stream, kernel kind: 0 encoder, 1 decoder): launches on one stream are ordered, streams
// are not.  nullptr (the kernels then stride statically): hipStreamPerThread (one handle value, a different real stream per thread), no
// slot left, or the allocation failed.
uint32_t* ticket_counters(hipStream_t s, int kind) {                  // caller holds g.mu (the encoder's launch path does)
    constexpr uint32_t kSlots = 64, kSlotWords = 64 * 9;
    if (s == hipStreamPerThread) return nullptr;
    if (!g.d_ctr) { if (hipMalloc((void**)&g.d_ctr, kSlots * kSlotWords * 4) != hipSuccess) { g.d_ctr = nullptr; return nullptr; } if (hipMemset(g.d_ctr, 0, kSlots * kSlotWords * 4) != hipSuccess) return nullptr; }
    const auto key = std::make_pair(s, kind);
    auto sl = g.ctr_slot.find(key);
    if (sl == g.ctr_slot.end() && g.ctr_slot.size() < kSlots) sl = g.ctr_slot.emplace(key, (uint32_t)g.ctr_slot.size()).first;
    return sl == g.ctr_slot.end() ? nullptr : g.d_ctr + kSlotWords * sl->second;
}

int scratch(int slot, size_t bytes, void** out, hipStream_t s = nullptr) {
    if (slot >= 2) {                                     // per stream: two streams may have frames in flight at the same time
        static thread_local char per_thread_key;          // hipStreamPerThread is one handle value but a different stream in every thread
        if (s == hipStreamPerThread) s = (hipStream_t)(void*)&per_thread_key;
        auto& e = g.sbuf[std::make_pair(slot, s)];
        if (bytes > e.second) {
            if (e.first) HIPCHK(hipFree(e.first));       // synchronises with whatever still reads it
            e.first = nullptr; e.second = 0;
            const size_t want = bytes + bytes / 8 + 4096;
            HIPCHK(hipMalloc(&e.first, want));
            e.second = want;
        }
        *out = e.first;
        return T3_OK;
    }
    if (bytes > g.cap[slot]) {
        if (g.buf[slot]) HIPCHK(hipFree(g.buf[slot]));
        g.buf[slot] = nullptr; g.cap[slot] = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        HIPCHK(hipMalloc(&g.buf[slot], want));
        g.cap[slot] = want;
    }
    *out = g.buf[slot];
    return T3_OK;
}

int get_lut(uint32_t kmask, int mode, const LutImage** out) {
    const uint32_t key = kmask | (uint32_t)mode << 8;
    auto it = g.luts.find(key);
    if (it == g.luts.end()) {
        LutImage L; std::vector<uint32_t> all;
        for (int i = 0; i < 4; ++i) if (kmask >> i & 1) {
            std::vector<uint32_t> img; build_encode_lut(kOfIndex[i], mode, img);
            L.k_off[i] = (uint32_t)all.size() * 4u;
            all.insert(all.end(), img.begin(), img.end());
        }
        L.bytes = (uint32_t)all.size() * 4u;
        HIPCHK(hipMalloc((void**)&L.d_img, L.bytes ? L.bytes : 16));
        HIPCHK(hipMemcpy(L.d_img, all.data(), L.bytes, hipMemcpyHostToDevice));
        it = g.luts.emplace(key, L).first;
    }
    *out = &it->second;
    return T3_OK;
}

// tables of the matrix-core encoder for one k (single-k launches)
int get_mfma_lut(int k, int mode, const LutImage** out) {
    const uint32_t key = 1u << k_index(k) | (uint32_t)mode << 8 | 1u << 16;
    auto it = g.luts.find(key);
    if (it == g.luts.end()) {
        LutImage L; std::vector<uint32_t> afrag, img; build_mfma_encode(k, mode, afrag, img);
        L.bytes = (uint32_t)img.size() * 4u;
        HIPCHK(hipMalloc((void**)&L.d_img, L.bytes)); HIPCHK(hipMemcpy(L.d_img, img.data(), L.bytes, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&L.d_afrag, afrag.size() * 4)); HIPCHK(hipMemcpy(L.d_afrag, afrag.data(), afrag.size() * 4, hipMemcpyHostToDevice));
        it = g.luts.emplace(key, L).first;
    }
    *out = &it->second;
    return T3_OK;
}

// tables of the UEP matrix-core kernel: the k-independent T/M image, then the A operand of every k in use (k_off = its offset)
int get_mfma_group_lut(uint32_t kmask, int mode, const LutImage** out) {
    const uint32_t key = kmask | (uint32_t)mode << 8 | 2u << 16;
    auto it = g.luts.find(key);
    if (it == g.luts.end()) {
        LutImage L; std::vector<uint32_t> all;
        for (int i = 0; i < 4; ++i) if (kmask >> i & 1) {
            std::vector<uint32_t> afrag, img; build_mfma_encode(kOfIndex[i], mode, afrag, img);
            if (all.empty()) all = img;                                   // T and M tables do not depend on k
            L.k_off[i] = (uint32_t)all.size() * 4u;
            all.insert(all.end(), afrag.begin(), afrag.end());
        }
        L.bytes = (uint32_t)all.size() * 4u;
        HIPCHK(hipMalloc((void**)&L.d_img, L.bytes)); HIPCHK(hipMemcpy(L.d_img, all.data(), L.bytes, hipMemcpyHostToDevice));
        it = g.luts.emplace(key, L).first;
    }
    *out = &it->second;
    return T3_OK;
}

DevDiv to_dev(FastDiv f) { return DevDiv{f.mul, f.sh, f.d}; }
uint32_t round16(uint32_t x) { return (x + 15u) & ~15u; }
uint64_t gcd64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }

// ------------------------------------------------------------------------------------------------
// K2 launch planning: one launch covers a set of bands whose k's have a manageable lcm
// ------------------------------------------------------------------------------------------------
struct EncLaunch { EncArgs a; uint32_t block; int rsel; };   // rsel = 26-k when all bands of the launch share k, else 0

// Phase 1 (pixels) gives a lane four consecutive pixel triples: waves that cover the worst-placed tile of TS symbols
// (tile starts cycle through S0 mod 52)
uint32_t p1_waves_per_parity(uint32_t TS) {
    uint32_t worst = 0;
    for (uint32_t t = 0; t < 52; ++t) {
        const uint64_t S0 = (uint64_t)t * TS;
        const uint64_t t_base = (S0 / 13) & ~3ull, t_end = (S0 + TS + 12) / 13;
        worst = std::max<uint32_t>(worst, (uint32_t)((t_end - t_base + 3) / 4));
    }
    return (worst + 63) / 64;
}

// ... and for raw words (1-D): four word triples = 104 symbols per lane (convert_words_packed)
uint32_t p1_waves_words(uint32_t TS) {
    uint32_t worst = 0;
    for (uint32_t t = 0; t < 104; ++t) {
        const uint64_t S0 = (uint64_t)t * TS;
        const uint64_t t_base = (S0 / 26) & ~3ull, t_end = (S0 + TS + 25) / 26;
        worst = std::max<uint32_t>(worst, (uint32_t)((t_end - t_base + 3) / 4));
    }
    return (worst + 63) / 64;
}

// grp = true: UEP on the matrix cores (all nine bands, several k): bands are grouped by k, a group's blocks of a tile are
// dealt linearly into sets of 32; eight waves take two sets each
bool plan_enc_group(const t3_layout& L, const t3_cfg& cfg, uint32_t band_mask, int fe, const LutImage& lut, EncLaunch& out, bool grp = false) {
    EncArgs& a = out.a; memset(&a, 0, sizeof a);
    const bool il2d = L.interleave2d && cfg.tile_w > 1;                  // rows of one symbol: the boustrophedon map is the identity (and the kernels' row divisions assume >= 2)
    const uint32_t GS = fe_px(fe) ? kGroupSyms : kGroupSymsW, GB = fe == FE_PIXELS ? kGroupBytes : fe == FE_RGB ? kGroupBytesRgb : kGroupBytesW;
    uint64_t Lk = 2;
    for (int b = 0; b < 9; ++b) if (band_mask >> b & 1) Lk = Lk / gcd64(Lk, L.band_k[b]) * L.band_k[b];
    uint32_t lut_bytes = lut.bytes;
    // 2-D through the pipelined flow (pixel / RGB input): rows up to 512 symbols -- a tile's input covers the row segments it overlaps (up to
    // w - 1 extra symbols each side) and a permutation pass follows phase 1 (il_async 1); wider rows -- the tile's pre-interleave symbols
    // are up to three runs, staged one behind the other at 1-KiB pitches, and phase 1 stores each symbol at its post-interleave place
    // (il_async 2; t3_kernels.hip, il_runs)
    static const uint32_t force_il = getenv("T3HIP_FORCE_IL") ? (uint32_t)atoi(getenv("T3HIP_FORCE_IL")) : 0u;   // measurement knob: 1 / 2 = that flow for every row width it can take
    const uint32_t il_async = !(il2d && fe_px(fe)) ? 0u : force_il == 2u ? 2u : (cfg.tile_w <= 512 ? 1u : 2u);
    const uint32_t il_extra = il_async == 1u ? 2u * cfg.tile_w : 0u;
    const uint32_t il_stage = il_async == 2u ? 2u * (1024u + 4u * GB + 32u) : 0u;   // two more runs: their rounding and pitch
    const uint32_t hdr = grp ? (uint32_t)kLdsHdrUep : (uint32_t)kLdsHdr;
    const bool words_packed = fe == FE_WORDS && !il2d;                     // 1-D raw words: the packed converter (whole lanes of 104 symbols: wider slack)
    const uint32_t sym_front = words_packed ? (uint32_t)kSymSlackW : (uint32_t)kSymFront, sym_back = words_packed ? (uint32_t)kSymSlackW : (uint32_t)kSymBack;
    bool mixed = false;
    { int k0 = 0; for (int b = 0; b < 9; ++b) if (band_mask >> b & 1) { if (!k0) k0 = L.band_k[b]; else if (k0 != L.band_k[b]) mixed = true; } }
    // pick q: tile = 9*Lk*q stream symbols; band b then owns Lk*q/k_b blocks
    double best_score = -1; uint32_t best_q = 0;
    for (int pass = 0; pass < 2 && !best_q; ++pass) {
        const uint32_t budget = pass == 0 ? 42u * 1280u : 160u * 1024u;     // LDS comes in 1280-byte units, 128 per CU (measured, t3_api_decode.cpp): <= 42 units = three workgroups per CU
        static const uint32_t force_q = getenv("T3HIP_FORCE_Q") ? (uint32_t)atoi(getenv("T3HIP_FORCE_Q")) : 0u;   // measurement knob
        for (uint32_t q = 1; q <= 4096; ++q) {
            if (force_q && q != force_q && q < force_q) continue;
            if (force_q && q > force_q) break;
            if (mixed && !grp && (q & 1u)) continue;                     // mixed k, LUT kernel: even multipliers only (measured: odd ones halve its speed)
            const uint64_t Lq = Lk * q; if (9 * Lq > 60000) break;
            uint32_t waves = 0, blocks_total = 0;
            for (int b = 0; b < 9; ++b) if (band_mask >> b & 1) {
                const uint32_t nb = (uint32_t)(Lq / L.band_k[b]);
                blocks_total += nb;
            }
            waves = (blocks_total + 63) / 64;                          // lanes are dealt to blocks linearly across bands
            uint32_t sets = 0;
            if (grp) {
                for (int i = 0; i < 4; ++i) { uint32_t items = 0; for (int b = 0; b < 9; ++b) if ((band_mask >> b & 1) && k_index(L.band_k[b]) == i) items += (uint32_t)(Lq / L.band_k[b]); sets += (items + 31) / 32; }
                if (sets > (uint32_t)kMaxSets || pass > 0) break;
                waves = 8;
            }
            if (waves > (pass == 0 ? 8u : (uint32_t)kMaxWaves)) break;   // pass 0: 512-thread workgroups, three per CU
            const uint32_t groups = (uint32_t)((9 * Lq + il_extra) / GS) + 8;
            uint32_t stage = groups * GB + 1024 + 32 + il_stage;                                              // +1 KiB: LDS-DMA pieces are whole
            if (il_async == 1u) stage = std::max<uint32_t>(stage, (uint32_t)(9 * Lq) + 64u);                  // the permutation pass writes the tile's 9 Lq symbols into the consumed stage buffer (RGB input is smaller than that)
            const uint32_t total = hdr + round16(lut_bytes) + sym_front + round16((uint32_t)(9 * Lq) + il_extra) + sym_back + ((il2d && !il_async) ? 1u : 2u) * round16(stage) + (fe == FE_RGB ? 256u : 0u);
            if (total > budget) break;
            // wave-instructions per stream symbol: phase 2 costs ~180 per wave, phase 1 (pixels) ~120 per wave-iteration
            const uint32_t wpp = words_packed ? p1_waves_words((uint32_t)(9 * Lq)) : p1_waves_per_parity((uint32_t)(9 * Lq) + il_extra + (il_async == 2u ? 312u : 0u));   // (three runs: up to six lane units of rounding)
            if ((fe_px(fe) || words_packed) && (words_packed ? 2u : 1u) * wpp > std::max(waves, 4u)) continue;
            // UEP kernel: the phases are barrier-separated and a wave runs its sets one after the other, so a tile costs one
            // phase-1 pass plus ceil(sets / 8) set times, whatever the number of busy waves
            // + a fixed cost per tile (barriers, ticket, prefetch issue, the runs' rounding in 2-D): without it the model preferred tiles of
            // 5 full waves to larger ones of 7-8 partly filled waves, measured 2-12 % slower (profiles/r03/notes.md: tile sweeps)
            const double cost = grp ? (220.0 + 100.0 * ((sets + 7) / 8)) / (double)(9 * Lq)
                                    : (600.0 + 180.0 * waves + (fe_px(fe) ? 220.0 * wpp : words_packed ? 250.0 * wpp : 180.0 * waves)) / (double)(9 * Lq);
            const double score = 1.0 / cost + 1e-9 * (double)Lq;
            if (score > best_score) { best_score = score; best_q = q; }
        }
    }
    if (!best_q) return false;
    const uint32_t Lq = (uint32_t)(Lk * best_q);
    a.Lq = Lq; a.lut_bytes = round16(lut_bytes);
    uint32_t off = hdr + a.lut_bytes;
    off += sym_front; a.sym_off = off; off += round16(9 * Lq + il_extra) + sym_back;   // slack either side: phase 1 writes whole pixel triples / whole lanes of word triples
    a.stage_off = off;
    a.stage_groups = (9 * Lq + il_extra) / GS + 8;
    uint32_t nw = 0, n_tiles = 0;
    for (int b = 0; b < 9; ++b) {
        a.band_k[b] = L.band_k[b]; a.band_blocks[b] = (uint32_t)L.band_blocks[b]; a.band_body_off[b] = L.band_body_off[b];
        a.band_boff6[b] = (uint32_t)((L.band_body_off[b] + 4) % 6);
        a.band_lut_off[b] = hdr + lut.k_off[k_index(L.band_k[b])];
        if (!(band_mask >> b & 1)) { a.band_nb_tile[b] = 0; a.band_blocks[b] = 0; continue; }
        const uint32_t nb = Lq / L.band_k[b];
        a.band_nb_tile[b] = nb;
        nw += nb;
        n_tiles = std::max<uint32_t>(n_tiles, (uint32_t)((L.band_blocks[b] + nb - 1) / nb));
    }
    a.n_items = nw; a.n_tiles = n_tiles;
    { uint32_t acc = 0; for (int b = 0; b < 9; ++b) { a.band_first[b] = acc; acc += a.band_nb_tile[b]; } a.band_first[9] = acc; }
    a.stage_stride = round16(std::max<uint32_t>(a.stage_groups * GB + 1024 + 32 + il_stage, il_async == 1u ? 9u * Lq + 64u : 0u));
    a.lds_bytes = a.stage_off + ((il2d && !il_async) ? 1u : 2u) * a.stage_stride;   // pipelined flow: two stage buffers (the next tile streams in early)
    if (fe == FE_RGB) { a.qt_off = a.lds_bytes; a.lds_bytes += 256u; }                            // chroma quantiser table of the fused bridge
    a.il_async = il_async;
    a.n_sym = (uint32_t)L.n_sym;
    const ScrCycle sc = scrambler_cycle(cfg.seed_a, cfg.seed_b, cfg.seed_s0);
    a.cyc24 = sc.cyc24; a.pre0 = sc.pre[0]; a.pre1 = sc.pre[1];
    mfma_scrambler_table(L.band_k[0], sc, a.scr);
    a.il_on = il2d;
    if (a.il_on) {
        const uint64_t A = (uint64_t)cfg.tile_w * cfg.tile_h;
        a.il_w = cfg.tile_w; a.il_A = (uint32_t)std::min<uint64_t>(A, std::max<uint64_t>(L.n_sym, 1));
        a.div_A = to_dev(fastdiv(a.il_A)); a.div_w = to_dev(fastdiv(a.il_w));
    }
    out.block = 64u * std::max<uint32_t>((nw + 63) / 64, 4u);
    a.p1_wpp = words_packed ? p1_waves_words(9 * Lq) : p1_waves_per_parity(9 * Lq + il_extra + (il_async == 2u ? 312u : 0u));
    out.rsel = 0;
    { int k0 = 0; bool same = true; for (int b = 0; b < 9; ++b) if (band_mask >> b & 1) { if (!k0) k0 = L.band_k[b]; else if (k0 != L.band_k[b]) same = false; } if (same && k0 && band_mask == 0x1FF) out.rsel = 26 - k0; }
    a.nb_uniform = out.rsel ? a.band_nb_tile[0] : 0u; a.div_nb = to_dev(fastdiv(a.nb_uniform ? a.nb_uniform : 1u));
    if (grp) {
        out.rsel = 1; out.block = 512;
        uint32_t ng = 0, ns = 0;
        for (int i = 0; i < 4; ++i) {
            EncArgs::Grp& G = a.grp[ng]; memset(&G, 0, sizeof G);
            uint32_t nbands = 0;
            for (int b = 0; b < 9; ++b) if ((band_mask >> b & 1) && k_index(L.band_k[b]) == i) G.bands[nbands++] = (uint8_t)b;
            if (!nbands) continue;
            G.nb = Lq / (uint32_t)kOfIndex[i]; G.div_nb = to_dev(fastdiv(G.nb)); G.n_items = nbands * G.nb; G.r = 26u - (uint32_t)kOfIndex[i];
            G.afrag_off = hdr + lut.k_off[i];
            mfma_scrambler_table(kOfIndex[i], sc, G.scr);
            for (uint32_t it0 = 0; it0 < G.n_items; it0 += 32) a.set_tab[ns++] = ng | it0 << 8;
            ++ng;
        }
        a.n_grp = ng; a.n_sets = ns;
    }
    return true;
}

int launch_fn(const void* fn, const EncLaunch& e, hipStream_t s) {
    // persistent grid = what is actually resident: workgroups/CU from the occupancy query (VGPR, LDS and wave limits)
    static std::map<std::pair<const void*, uint64_t>, int> occ_cache; static std::mutex occ_mu;   // per device: the dynamic-LDS attribute is set on the device's copy of the function
    std::lock_guard<std::mutex> occ_lk(occ_mu);
    const auto key = std::make_pair(fn, (uint64_t)(uint32_t)g.dev << 48 | (uint64_t)e.block << 32 | e.a.lds_bytes);
    auto it = occ_cache.find(key);
    if (it == occ_cache.end()) {
        HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        int occ = 1;
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)e.block, e.a.lds_bytes));
        occ = std::min<int>(occ, (int)(128u / ((e.a.lds_bytes + 1279u) / 1280u)));   // LDS is handed out in 1280-byte units, 128 per CU (t3_api_decode.cpp)
        it = occ_cache.emplace(key, std::max(1, occ)).first;
    }
    static const int occ_cap = getenv("T3HIP_MAX_WG_PER_CU") ? atoi(getenv("T3HIP_MAX_WG_PER_CU")) : 0;   // measurement knob
    const int per_cu = occ_cap > 0 ? std::min(occ_cap, it->second) : it->second;
    const uint32_t grid = std::max<uint32_t>(1u, std::min<uint32_t>(e.a.n_tiles, (uint32_t)(g.n_cu * per_cu)));
#ifdef T3_STAMPS
    static uint64_t* d_dbg = nullptr; static int calls = 0;
    if (!d_dbg) HIPCHK(hipMalloc((void**)&d_dbg, 16 * 8 * 4096));
    HIPCHK(hipMemsetAsync(d_dbg, 0, 16 * 8 * 4096, s));
    const_cast<EncLaunch&>(e).a.dbg = d_dbg;
#endif
    {   // dynamic tile tickets: a zeroed counter set per stream (launches on one stream are ordered; the kernel re-zeroes it)
        static const bool off = getenv("T3HIP_STATIC_TILES") != nullptr;      // measurement knob
        const_cast<EncLaunch&>(e).a.tile_ctr = off ? nullptr : ticket_counters(s, 0);
        const_cast<EncLaunch&>(e).a.n_classes = std::min<uint32_t>(8u, grid);
    }
    void* args[] = {(void*)&e.a};
    HIPCHK(hipLaunchKernel(fn, dim3(grid), dim3(e.block), args, e.a.lds_bytes, s));
#ifdef T3_STAMPS
    if (++calls == 8) {                                     // one report, after warm-up
        std::vector<uint64_t> h16(16 * grid), h(8 * grid);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(h16.data(), d_dbg, h16.size() * 8, hipMemcpyDeviceToHost));
        for (uint32_t w = 0; w < grid; ++w) for (int i = 0; i < 8; ++i) h[8 * w + i] = h16[16 * w + i];
        { std::map<uint32_t, std::vector<uint32_t>> per_cu; double xl[8] = {0}; int xn[8] = {0};
          for (uint32_t w = 0; w < grid; ++w) { const uint32_t hw = (uint32_t)h16[16 * w + 8], xcc = (uint32_t)h16[16 * w + 9] & 15u;
              per_cu[xcc << 16 | (hw >> 8 & 0xFFu)].push_back((uint32_t)h[8 * w + 5]); xl[xcc & 7] += (double)h[8 * w + 5] * 0.01; ++xn[xcc & 7]; }
          int hist[8] = {0}; for (auto& kv : per_cu) ++hist[std::min<size_t>(kv.second.size(), 7)];
          fprintf(stderr, "[t3 stamps]   CUs seen=%zu  CUs holding n WGs: 1:%d 2:%d 3:%d 4:%d 5:%d 6+:%d\n", per_cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5], hist[6] + hist[7]);
          fprintf(stderr, "[t3 stamps]   mean WG lifetime (us) per XCC:"); for (int x = 0; x < 8; ++x) fprintf(stderr, " %d:%.1f(n=%d)", x, xn[x] ? xl[x] / xn[x] : 0.0, xn[x]); fprintf(stderr, "\n");
          double ln[8] = {0}; int cn[8] = {0}; for (auto& kv : per_cu) { const size_t n = std::min<size_t>(kv.second.size(), 7); for (uint32_t v : kv.second) { ln[n] += v * 0.01; ++cn[n]; } }
          fprintf(stderr, "[t3 stamps]   mean WG lifetime (us) by WGs on its CU:"); for (int n = 1; n < 8; ++n) if (cn[n]) fprintf(stderr, " %d:%.1f", n, ln[n] / cn[n]); fprintf(stderr, "\n");
          fprintf(stderr, "[t3 stamps]   hw_id samples: %08x %08x %08x %08x\n", (unsigned)h16[8], (unsigned)h16[16 + 8], (unsigned)h16[32 + 8], (unsigned)h16[16 * 100 + 8]); }
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t w = 0; w < grid; ++w) for (int i = 0; i < 8; ++i) acc[i] += (double)h[8 * w + i];
        fprintf(stderr, "[t3 stamps] grid=%u tiles=%u  mean cycles/WG: stage=%.0f p1=%.0f p2=%.0f p3=%.0f total=%.0f  clock=%.3f GHz\n", grid, e.a.n_tiles,
                acc[0] / grid, acc[1] / grid, acc[2] / grid, acc[3] / grid, acc[4] / grid, acc[4] / acc[5] * 0.1);
        { uint64_t s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0; for (uint32_t w = 0; w < grid; ++w) { const uint64_t st = h[8 * w + 3], en = st + h[8 * w + 5]; s0 = std::min(s0, st); s1 = std::max(s1, st); e0 = std::min(e0, en); e1 = std::max(e1, en); }
          int late = 0; for (uint32_t w = 0; w < grid; ++w) if (h[8 * w + 3] - s0 > 1000) ++late;
          fprintf(stderr, "[t3 stamps]   timeline (us from first start): last start=%.2f first end=%.2f last end=%.2f  WGs starting >10us late=%d\n", (s1 - s0) * 0.01, (e0 - s0) * 0.01, (e1 - s0) * 0.01, late);
          fprintf(stderr, "[t3 stamps]   lds_bytes=%u block=%u\n", e.a.lds_bytes, e.block); }
        fprintf(stderr, "[t3 stamps]   p1 split (wave 0): prefetch issue=%.0f convert=%.0f barrier wait=%.0f\n", acc[6] / grid, acc[7] / grid, acc[1] / grid);
        { double il = 0; for (uint32_t w = 0; w < grid; ++w) il += (double)h16[16 * w + 10]; fprintf(stderr, "[t3 stamps]   2-D permutation pass (in p2): %.0f\n", il / grid); }
    }
#endif
    return T3_OK;
}
template <int FE, int IL, bool BCN> int launch_enc2(const EncLaunch& e, hipStream_t s) {
    const void* fn = BCN ? nullptr : (const void*)encode_kernel_mixed<FE, IL>;       // the LUT kernel has no fused beacon (the caller adds the pass)
    if (e.rsel == 1) fn = (const void*)encode_kernel_uep<FE, IL, BCN>;   // UEP on the matrix cores
    else if (e.a.afrag) switch (e.rsel) {                      // single-k launches: matrix-core kernels (<= 640 threads)
        case 2: fn = (const void*)encode_kernel_k<FE, IL, 2, BCN>; break;
        case 4: fn = (const void*)encode_kernel_k<FE, IL, 4, BCN>; break;
        case 6: fn = (const void*)encode_kernel_k<FE, IL, 6, BCN>; break;
        case 8: fn = (const void*)encode_kernel_k<FE, IL, 8, BCN>; break;
        default: break;
    }
    if (!fn) return T3_E_ARG;
    return launch_fn(fn, e, s);
}
template <int FE, int IL> int launch_enc1(const EncLaunch& e, hipStream_t s) { return e.a.bcn_pb ? launch_enc2<FE, IL, true>(e, s) : launch_enc2<FE, IL, false>(e, s); }
template <int FE> int launch_enc(const EncLaunch& e, hipStream_t s) {     // kernel flavour of the 2-D flow: see encode_body (raw words: always 1)
    if (!e.a.il_on) return launch_enc1<FE, 0>(e, s);
    if constexpr (FE != FE_WORDS) { if (e.a.il_async == 2u) return launch_enc1<FE, 2>(e, s); }
    return launch_enc1<FE, 1>(e, s);
}
int launch_enc_fe(int fe, const EncLaunch& e, hipStream_t s) { return fe == FE_PIXELS ? launch_enc<FE_PIXELS>(e, s) : fe == FE_RGB ? launch_enc<FE_RGB>(e, s) : launch_enc<FE_WORDS>(e, s); }

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// chroma quantiser of the fused RGB front end: C -> clamp(lround((C - 128) * (40.0 / 128.0)), -40, 40) + 40 (io_image.hpp:73-76), the
// reference's own double expression tabulated on the host
int rgb_quant_table(const uint8_t** out) {
    uint8_t*& d_qt = (uint8_t*&)g.slot[47];
    if (!d_qt) {
        uint8_t t[256];
        for (int c = 0; c < 256; ++c) { long v = lround((c - 128) * (40.0 / 128.0)); v = v < -40 ? -40 : (v > 40 ? 40 : v); t[c] = (uint8_t)(v + 40); }
        HIPCHK(hipMalloc((void**)&d_qt, sizeof t)); HIPCHK(hipMemcpy(d_qt, t, sizeof t, hipMemcpyHostToDevice));
    }
    *out = d_qt; return T3_OK;
}

// pixels|raw words (device) -> coded stream (device)
int encode_dev(int fe, const void* d_in, uint64_t n_units, const t3_cfg* cfg, void* d_out, uint64_t cap_words, uint64_t* n_out, hipStream_t s) {
    if (!g.ready) return T3_E_NODEVICE;
    if (fe == FE_RGB && cfg && n_out && d_in && !aligned16(d_in)) return 1;      // the bridge kernel takes any alignment
    const bool raw_mode = cfg && cfg->profile == T3_RAW_MODE;                       // RAW: a copy or the pack kernel, any alignment
    if (!cfg || !n_out || (n_units && !d_in) || (!raw_mode && (!aligned16(d_in) || !aligned16(d_out)))) return T3_E_ARG;
    const uint64_t n_raw = fe_px(fe) ? (n_units + 1) / 2 : n_units;
    t3_layout L; int rc = plan(n_raw, *cfg, L); if (rc != T3_OK) return rc;
    *n_out = L.out_words;
    // the fused RGB front end rides the pipelined flow only: RAW mode and 2-D rows wider than 512 go through the bridge kernel (1 = not taken)
    if (fe == FE_RGB && cfg->profile == T3_RAW_MODE) return 1;
    if (L.out_words > cap_words) return T3_E_CAPACITY;
    if (L.out_words && !d_out) return T3_E_ARG;
    if (cfg->profile == T3_RAW_MODE) {                                   // OLD:1046-1050: out = in
        if (fe == FE_WORDS) { if (n_raw) HIPCHK(hipMemcpyAsync(d_out, d_in, n_raw * 9, hipMemcpyDeviceToDevice, s)); }
        else if (n_raw) { hipLaunchKernelGGL(pack_pixels_kernel, dim3((unsigned)(((n_raw + 3) / 4 + 255) / 256)), dim3(256), 0, s, (const uint16_t*)d_in, n_units, (uint8_t*)d_out, n_raw); HIPCHK(hipGetLastError()); }
        return T3_OK;
    }
    std::lock_guard<std::mutex> lk(g.mu);
    uint8_t hdr[96]; memset(hdr, 0, sizeof hdr);
    const uint32_t hs = (uint32_t)header_encode(*cfg, n_raw, hdr);
    const uint32_t pad = (uint32_t)(9 * L.out_words - L.out_syms);
    uint8_t* body_out = (uint8_t*)d_out + hs; uint8_t* frame_out = (uint8_t*)d_out;
    // A beacon (OLD:1118-1141) rides in the store addressing of the matrix-core kernels when one launch covers the frame and a 16-byte
    // run can hold one beacon at most (period >= 2); otherwise the body goes to scratch and beacon_kernel frames it.
    const bool bcn_cand = L.beacon_on && cfg->beacon_band_slot < 9 && cfg->beacon_words_period >= 2 && cfg->beacon_words_period < (1u << 27) && !getenv("T3HIP_BEACON_PASS");
    bool bcn_fused = false;
    // group bands into launches: all together when the lcm of their k's keeps the tile small, else one launch per k
    uint32_t kmask = 0; for (int b = 0; b < 9; ++b) kmask |= 1u << k_index(L.band_k[b]);
    std::vector<uint32_t> groups;
    const bool single_k = (kmask & (kmask - 1)) == 0;                    // one k for all nine bands: matrix-core kernels
    {
        const LutImage* lut; rc = get_lut(kmask, cfg->mode, &lut); if (rc) return rc;
        EncLaunch e;
        if (single_k || plan_enc_group(L, *cfg, 0x1FF, fe, *lut, e)) groups.push_back(0x1FF);
        else {
            for (int i = 0; i < 4; ++i) if (kmask >> i & 1) { uint32_t m = 0; for (int b = 0; b < 9; ++b) if (k_index(L.band_k[b]) == i) m |= 1u << b; groups.push_back(m); }
            // three or four different k: the lcm of all of them makes a tile no LDS holds, the lcm of two does -- the bands go in two
            // launches of the matrix-core UEP kernel, by pairs of k (each launch runs phase 1 over the whole frame and encodes its bands)
            if (groups.size() >= 3) { std::vector<uint32_t> pr; for (size_t i = 0; i < groups.size(); i += 2) pr.push_back(groups[i] | (i + 1 < groups.size() ? groups[i + 1] : 0u)); groups.swap(pr); }
        }
    }
    bool first = true;
    for (uint32_t m : groups) {
        uint32_t km = 0; for (int b = 0; b < 9; ++b) if (m >> b & 1) km |= 1u << k_index(L.band_k[b]);
        bool mfma = single_k && m == 0x1FF;
        const LutImage* lut; EncLaunch e;
        bool uep = false;
        if (!single_k) {                                                 // several k in the frame: try the matrix-core UEP kernel (bands grouped by k; any band subset)
            rc = get_mfma_group_lut(km, cfg->mode, &lut); if (rc) return rc;
            uep = plan_enc_group(L, *cfg, m, fe, *lut, e, true);
        }
        if (mfma) {                                                      // matrix-core kernel: 64 or 32 blocks per band and tile
            rc = get_mfma_lut(L.band_k[0], cfg->mode, &lut); if (rc) return rc;
            mfma = plan_enc_group(L, *cfg, m, fe, *lut, e) && e.rsel && e.block <= 512;   // the tile must fit eight waves
        }
        if (!mfma && !uep) {
            rc = get_lut(km, cfg->mode, &lut); if (rc) return rc;
            if (!plan_enc_group(L, *cfg, m, fe, *lut, e)) return T3_E_ARG;
        }
        if (L.beacon_on) {
            bcn_fused = bcn_cand && groups.size() == 1 && (mfma || uep);
            if (bcn_fused) {
                const uint64_t cyc = 9ull * cfg->beacon_words_period, pb = cyc - 1, B = L.body_syms, slot = cfg->beacon_band_slot;
                e.a.bcn_slot = (uint32_t)slot; e.a.bcn_pb = (uint32_t)pb; e.a.bcn_div = to_dev(fastdiv((uint32_t)pb));
                e.a.bcn_sym = beacon_symbol(cfg->profile, (uint16_t)(cfg->superframe_words % 5), 0);     // OLD:1130
                // framed bytes after the last body byte (the rest of the last word): zeros, or a beacon whose slot comes after it
                const uint64_t next = B ? B + (B - 1 < slot ? 0 : 1 + (B - 1 - slot) / pb) : 0;
                e.a.bcn_tail_off = hs + next; e.a.bcn_tail_len = (uint32_t)(L.body_syms_framed - next); e.a.bcn_tail_vals = 0;
                if (e.a.bcn_tail_len > 8) return T3_E_ARG;                                   // (cannot happen: less than one word)
                for (uint64_t q = next; q < L.body_syms_framed; ++q) if (q >= slot && (q - slot) % cyc == 0) e.a.bcn_tail_vals |= (uint64_t)e.a.bcn_sym << (8 * (q - next));
            } else { void* p; rc = scratch(2, L.body_syms + 64, &p, s); if (rc) return rc; body_out = (uint8_t*)p; frame_out = nullptr; }
        }
        e.a.afrag = mfma ? lut->d_afrag : nullptr;
        e.a.in = (const uint8_t*)d_in; e.a.n_units = n_units; e.a.n_units_pad = fe_px(fe) ? 2 * n_raw : n_units;
        if (fe == FE_RGB) { rc = rgb_quant_table(&e.a.qt); if (rc) return rc; }
        e.a.body_out = body_out; e.a.frame_out = first ? frame_out : nullptr; e.a.lut_img = lut->d_img;
        e.a.hdr_syms = hs; e.a.pad_bytes = pad; e.a.out_syms = L.out_syms; memcpy(e.a.hdr, hdr, sizeof hdr);
        rc = launch_enc_fe(fe, e, s);
        if (rc) return rc;
        first = false;
    }
    if (L.beacon_on && !bcn_fused) {
        BeaconArgs b; memset(&b, 0, sizeof b);
        b.body = body_out; b.frame_out = (uint8_t*)d_out; b.body_syms = L.body_syms; b.framed_syms = L.body_syms_framed;
        b.period = cfg->beacon_words_period; b.slot = cfg->beacon_band_slot;
        b.sym = beacon_symbol(cfg->profile, (uint16_t)(cfg->superframe_words % 5), 0);     // OLD:1130
        b.hdr_syms = hs; b.pad_bytes = pad; memcpy(b.hdr, hdr, sizeof hdr);
        const unsigned nb = (unsigned)std::min<uint64_t>(std::max<uint64_t>(1, ((hs + L.body_syms_framed + 15) / 16 + 255) / 256), 65536);   // one lane per 16-byte granule
        hipLaunchKernelGGL(beacon_kernel, dim3(nb), dim3(256), 0, s, b);
        HIPCHK(hipGetLastError());
    }
    return T3_OK;
}

// host vector -> device scratch -> kernel -> host vector
int host_roundtrip_in(int slot, const void* h, size_t bytes, void** d) {
    int rc = scratch(slot, bytes + 64, d); if (rc) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(*d, h, bytes, hipMemcpyHostToDevice, g.stream));
    return T3_OK;
}

}  // namespace

extern "C" {

int t3hip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

static int ctx_init_in(Ctx* c, int device);
static void ctx_teardown(Ctx* c);
// Builds `c` on `device`.  The calling thread's current context is `c` for the whole of it, so a HIP error is recorded in `c` (and
// copied to the null context for t3hip_last_hip_error after a failed create); what a failed build had already allocated is released.
static int ctx_init(Ctx* c, int device) {
    Ctx* const prev = tl_cur; tl_cur = c;
    const int rc = ctx_init_in(c, device);
    tl_cur = prev;
    if (rc) { const std::string err = c->hip_err; ctx_teardown(c); g_null.hip_err = err; c->hip_err = err; }
    return rc;
}
static int ctx_init_in(Ctx* c, int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return T3_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t p; HIPCHK(hipGetDeviceProperties(&p, device));
    if (std::string(p.gcnArchName).find("gfx950") == std::string::npos) { c->hip_err = std::string("not a gfx950 device: ") + p.gcnArchName; return T3_E_NODEVICE; }
    c->n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    const Field& F = field();
    HIPCHK(hipMalloc((void**)&c->d_tab, sizeof(RsTables)));
    HIPCHK(hipMemcpy(c->d_tab, &F.t, sizeof(RsTables), hipMemcpyHostToDevice));
    for (int i = 0; i < 4; ++i) for (int m = 0; m < 2; ++m) {
        uint8_t P[24 * 8]; rs_parity_matrix(kOfIndex[i], m, P);
        HIPCHK(hipMalloc((void**)&c->d_P[i][m], sizeof P));
        HIPCHK(hipMemcpy(c->d_P[i][m], P, sizeof P, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMalloc((void**)&c->d_flag, 64));
    c->dev = device;
    const int rc = t3::decode_init(c->d_tab);             // the decode half builds its tables into the context it finds current (= c)
    if (rc) return rc;
    c->ready = true;
    return T3_OK;
}
static void ctx_teardown(Ctx* c) {                       // also of a partly built context (ctx_init failed): every member is null-safe
    Ctx* const prev = tl_cur; tl_cur = c;
    if (c->dev >= 0) { (void)hipSetDevice(c->dev); (void)hipDeviceSynchronize(); }
    for (auto& kv : c->luts) { (void)hipFree(kv.second.d_img); if (kv.second.d_afrag) (void)hipFree(kv.second.d_afrag); }
    c->luts.clear();
    for (auto& kv : c->sbuf) if (kv.second.first) (void)hipFree(kv.second.first);
    c->sbuf.clear();
    for (int i = 0; i < 4; ++i) { if (c->buf[i]) (void)hipFree(c->buf[i]); c->buf[i] = nullptr; c->cap[i] = 0; for (int m = 0; m < 2; ++m) { if (c->d_P[i][m]) (void)hipFree(c->d_P[i][m]); c->d_P[i][m] = nullptr; } }
    t3::decode_shutdown();                                // frees what the other translation units keep in this context's slots
    if (c->slot[47]) (void)hipFree(c->slot[47]);          // chroma quantiser table of the fused RGB front end (rgb_quant_table)
    for (void*& p : c->slot) p = nullptr;
    if (c->d_ctr) { (void)hipFree(c->d_ctr); c->d_ctr = nullptr; } c->ctr_slot.clear();
    for (hipEvent_t e : c->chunk_ev) (void)hipEventDestroy(e);
    c->chunk_ev.clear();
    if (c->stream2) { (void)hipStreamDestroy(c->stream2); c->stream2 = nullptr; }
    if (c->d_tab) (void)hipFree(c->d_tab); if (c->d_flag) (void)hipFree(c->d_flag); if (c->stream) (void)hipStreamDestroy(c->stream);
    c->d_tab = nullptr; c->d_flag = nullptr; c->stream = nullptr;
    c->ready = false; c->dev = -1;
    tl_cur = prev == c ? nullptr : prev;
}
static std::mutex g_ctx_mu;

int t3hip_init(int device) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_def && g_def->ready) return g_def->dev == device ? T3_OK : T3_E_ARG;      // the default context stays on its device: t3hip_shutdown first, or t3hip_create
    if (!g_def) g_def = new Ctx();
    const int rc = ctx_init(g_def, device);
    if (rc) { delete g_def; g_def = nullptr; }
    return rc;
}

int t3hip_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!g_def) return T3_OK;
    if (g_def->ready) ctx_teardown(g_def);
    if (tl_cur == g_def) tl_cur = nullptr;
    delete g_def; g_def = nullptr;
    return T3_OK;
}

int t3hip_create(int device, t3hip_ctx** out) {
    if (!out) return T3_E_ARG;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    int prev_dev = -1; (void)hipGetDevice(&prev_dev);
    Ctx* c = new Ctx();
    const int rc = ctx_init(c, device);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    if (rc) { delete c; return rc; }
    *out = (t3hip_ctx*)c;
    return T3_OK;
}
int t3hip_destroy(t3hip_ctx* h) {
    if (!h) return T3_OK;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    Ctx* c = (Ctx*)h;
    if (c == g_def) return T3_E_ARG;                      // the default context goes with t3hip_shutdown
    int prev_dev = -1; (void)hipGetDevice(&prev_dev);
    if (c->ready) ctx_teardown(c);
    if (tl_cur == c) tl_cur = nullptr;
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    delete c;
    return T3_OK;
}
int t3hip_use(t3hip_ctx* h) {
    Ctx* c = (Ctx*)h;
    if (c && !c->ready) return T3_E_ARG;
    tl_cur = c;
    Ctx* eff = cur_ctx();
    if (eff->ready) HIPCHK(hipSetDevice(eff->dev));
    return T3_OK;
}
t3hip_ctx* t3hip_current(void) { Ctx* c = cur_ctx(); return c->ready ? (t3hip_ctx*)c : nullptr; }
int t3hip_ctx_device(const t3hip_ctx* h) { const Ctx* c = h ? (const Ctx*)h : cur_ctx(); return c->ready ? c->dev : -1; }
int t3hip_is_ready(void) { return g.ready ? 1 : 0; }
const char* t3hip_strerror(int c) {
    switch (c) {
        case T3_OK: return "ok"; case T3_E_NODEVICE: return "no usable gfx950 device (t3hip_init not done or failed)";
        case T3_E_HIP: return "HIP runtime error"; case T3_E_ARG: return "bad argument"; case T3_E_CAPACITY: return "output buffer too small";
        case T3_E_HEADER: return "superframe header did not decode (RS/CRC-12)"; case T3_E_RS: return "uncorrectable RS block";
        case T3_E_COMM: return "RCCL unavailable or collective failed";
    }
    return "unknown";
}
const char* t3hip_last_hip_error(void) { return g.hip_err.c_str(); }
const char* t3hip_version(void) { return "t3hip 0.1 (gfx950)"; }

// ---- host-only metadata ---------------------------------------------------------------------------------
void t3hip_cfg_default(t3_cfg* c) {
    memset(c, 0, sizeof *c);
    c->profile = T3_P2_RS26_22; for (int b = 0; b < 9; ++b) c->band_profile[b] = 1;     // OLD:864,898
    c->seed_a = c->seed_b = c->seed_s0 = 1; c->superframe_words = 8192; c->subword = 27; c->centered = 1;
}
int t3hip_plan(uint64_t n_raw, const t3_cfg* cfg, t3_layout* out) { if (!cfg || !out) return T3_E_ARG; return plan(n_raw, *cfg, *out); }
uint64_t t3hip_encoded_words(uint64_t n_raw, const t3_cfg* cfg) { t3_layout L; if (!cfg || plan(n_raw, *cfg, L) != T3_OK) return 0; return L.out_words; }
int t3hip_gf27_tables(uint8_t* e78, int16_t* l27, uint8_t* m729, uint8_t* i27) {
    const Field& F = field();
    if (e78) memcpy(e78, F.exp78, 78); if (l27) memcpy(l27, F.log, sizeof F.log);
    if (m729) memcpy(m729, F.t.mul, 729); if (i27) memcpy(i27, F.t.inv, 27);
    return T3_OK;
}
int t3hip_rs_generator(int k, uint8_t* g_out) { if (!valid_k(k) || !g_out) return T3_E_ARG; rs_generator(k, g_out); return T3_OK; }
int t3hip_mfma_encode_tables(int k, int mode, uint32_t* afrag, uint32_t* lds_img) {
    if (!valid_k(k) || !afrag || !lds_img || mode < 0 || mode > 1) return T3_E_ARG;
    std::vector<uint32_t> a, l; build_mfma_encode(k, mode, a, l);
    memcpy(afrag, a.data(), a.size() * 4); memcpy(lds_img, l.data(), l.size() * 4);
    return T3_OK;
}
int t3hip_rs_parity_matrix(int k, int mode, uint8_t* P) { if (!valid_k(k) || !P || mode < 0 || mode > 1) return T3_E_ARG; rs_parity_matrix(k, mode, P); return T3_OK; }
int t3hip_header_pack(const t3_cfg* c, uint32_t fs, uint32_t bh, uint8_t s[27]) { if (!c || !s) return T3_E_ARG; header_pack(*c, fs, bh, s); return T3_OK; }
int t3hip_header_check(const uint8_t s[27]) { return s && header_check(s) ? 1 : 0; }
int t3hip_header_unpack(const uint8_t s[27], t3_cfg* o, uint32_t* fs, uint32_t* bh) { if (!s || !o) return T3_E_ARG; header_unpack(s, *o, fs, bh); return T3_OK; }
int t3hip_header_encode(const t3_cfg* c, uint64_t n_raw, uint8_t* out, uint32_t* n) {
    if (!c || !out || !n) return T3_E_ARG;
    t3_layout L; int rc = plan(n_raw, *c, L); if (rc) return rc;
    if (c->profile == T3_RAW_MODE) { *n = 0; return T3_OK; }
    *n = (uint32_t)header_encode(*c, n_raw, out); return T3_OK;
}

// ---- device-resident entry points -----------------------------------------------------------------------
int t3hip_pack_pixels_dev(const void* d_px, uint64_t n_px, void* d_words, void* stream) {
    if (!g.ready) return T3_E_NODEVICE;
    const uint64_t nw = (n_px + 1) / 2; if (!nw) return T3_OK;
    if (!d_px || !d_words) return T3_E_ARG;
    hipLaunchKernelGGL(pack_pixels_kernel, dim3((unsigned)(((nw + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_px, n_px, (uint8_t*)d_words, nw);
    HIPCHK(hipGetLastError()); return T3_OK;
}
int t3hip_unpack_words_dev(const void* d_words, uint64_t n_words, void* d_px, void* stream) {
    if (!g.ready) return T3_E_NODEVICE;
    if (!n_words) return T3_OK;
    if (!d_px || !d_words) return T3_E_ARG;
    hipLaunchKernelGGL(unpack_words_kernel, dim3((unsigned)(((n_words + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)d_words, n_words, (uint16_t*)d_px);
    HIPCHK(hipGetLastError()); return T3_OK;
}
int t3hip_encode_profile_dev(const void* d_raw, uint64_t n_raw, const t3_cfg* cfg, void* d_out, uint64_t cap, uint64_t* n_out, void* stream) {
    return encode_dev(FE_WORDS, d_raw, n_raw, cfg, d_out, cap, n_out, (hipStream_t)stream);
}
int t3hip_encode_frame_dev(const void* d_px, uint64_t n_px, const t3_cfg* cfg, void* d_out, uint64_t cap, uint64_t* n_out, void* stream) {
    return encode_dev(FE_PIXELS, d_px, n_px, cfg, d_out, cap, n_out, (hipStream_t)stream);
}
int t3hip_rs_encode_blocks_dev(int k, int mode, const uint8_t* d_data, uint64_t n_blocks, uint8_t* d_code, void* stream) {
    if (!g.ready) return T3_E_NODEVICE;
    if (!valid_k(k) || mode < 0 || mode > 1) return T3_E_ARG;
    if (!n_blocks) return T3_OK;
    hipLaunchKernelGGL(rs_encode_blocks_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_data, n_blocks, k, g.d_P[k_index(k)][mode], g.d_tab, d_code);
    HIPCHK(hipGetLastError()); return T3_OK;
}

// ---- host-buffer entry points ---------------------------------------------------------------------------
int t3hip_pack_pixels(const void* px, uint64_t n_px, void* words) {
    if (!g.ready) return T3_E_NODEVICE;
    const uint64_t nw = (n_px + 1) / 2; if (!nw) return T3_OK;
    if (!px || !words) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(g.host_mu);
    void *di, *dout; int rc;
    { std::lock_guard<std::mutex> lk(g.mu); rc = host_roundtrip_in(0, px, n_px * 6, &di); if (rc) return rc; rc = scratch(1, nw * 9, &dout); if (rc) return rc; }
    rc = t3hip_pack_pixels_dev(di, n_px, dout, g.stream); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(words, dout, nw * 9, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream)); return T3_OK;
}
int t3hip_unpack_words(const void* words, uint64_t n_words, void* px) {
    if (!g.ready) return T3_E_NODEVICE;
    if (!n_words) return T3_OK;
    if (!px || !words) return T3_E_ARG;
    std::lock_guard<std::recursive_mutex> hl(g.host_mu);
    void *di, *dout; int rc;
    { std::lock_guard<std::mutex> lk(g.mu); rc = host_roundtrip_in(0, words, n_words * 9, &di); if (rc) return rc; rc = scratch(1, n_words * 12, &dout); if (rc) return rc; }
    rc = t3hip_unpack_words_dev(di, n_words, dout, g.stream); if (rc) return rc;
    HIPCHK(hipMemcpyAsync(px, dout, n_words * 12, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream)); return T3_OK;
}
// Pipelined host entry (round 3): the frame crosses PCIe once in each direction, and the two directions overlap.  Tiles are independent
// (SURVEY 5), and a range of whole tiles that starts on a pixel-triple / word-triple boundary at a 16-byte aligned input offset is a
// frame of its own to the kernel -- same kernel, shifted pointers and band offsets, no tile-range logic in the hot loop.  The caller's
// thread uploads chunk c and launches it; a helper thread downloads the nine band runs of chunk c - 1 meanwhile (pageable memory: a HIP
// copy occupies its calling thread, so the two directions need two threads; measured on the box, profiles/exp/pcie_probe.cpp: 3.5 ms up
// + 3.3 ms down one after the other, 4.0-4.2 ms both at once).  One k on all bands, 1-D, no beacon; anything else: the serial path (1).
static int encode_host_pipelined(int fe, const void* in, uint64_t n_units, const t3_cfg* cfg, void* out, const t3_layout& L, void* di, void* dout) {
    if (fe == FE_RGB || L.interleave2d || L.beacon_on || cfg->profile == T3_RAW_MODE || getenv("T3HIP_SERIAL_HOST") != nullptr) return 1;
    for (int b = 1; b < 9; ++b) if (L.band_k[b] != L.band_k[0]) return 1;
    const uint64_t n_raw = fe == FE_PIXELS ? (n_units + 1) / 2 : n_units;
    EncLaunch e0; const LutImage* lut;
    uint8_t hdr[96]; memset(hdr, 0, sizeof hdr);
    uint32_t hs;
    {
        std::lock_guard<std::mutex> lk(g.mu);
        int rc = get_mfma_lut(L.band_k[0], cfg->mode, &lut); if (rc) return rc;
        if (!plan_enc_group(L, *cfg, 0x1FF, fe, *lut, e0) || !e0.rsel || e0.block > 512) return 1;
        hs = (uint32_t)header_encode(*cfg, n_raw, hdr);
    }
    const uint32_t TS = 9u * e0.a.Lq, unit_syms = fe == FE_PIXELS ? 104u : 416u;     // chunk starts: whole triples at 16-byte aligned input offsets
    uint32_t G = unit_syms; { uint32_t x = TS, y = unit_syms; while (y) { const uint32_t t = x % y; x = y; y = t; } G = unit_syms / x; }   // tiles per alignment unit
    const uint32_t n_tiles = e0.a.n_tiles;
    static const uint32_t want_env = getenv("T3HIP_HOST_CHUNKS") ? (uint32_t)atoi(getenv("T3HIP_HOST_CHUNKS")) : 0u;   // measurement knob
    // chunks: fill / drain of the pipeline against per-copy overheads.  The nine band runs of a chunk go down as ONE strided copy when the
    // bands are equally long and everything is 4-byte aligned (COMPAT: 52 header symbols; measured 4.70 ms per 8K frame with 12 chunks);
    // a strided copy at 2-byte alignment (FIXED: 90 header symbols) falls off a cliff (13 ms), so there: nine plain copies, 6 chunks (5.1 ms)
    bool even_bands = true; for (int b = 1; b < 9; ++b) even_bands = even_bands && L.band_blocks[b] == L.band_blocks[0];
    const bool strided = even_bands && hs % 4u == 0 && (26ull * L.band_blocks[0]) % 4u == 0 && getenv("T3HIP_NO_2D_COPY") == nullptr;
    const uint32_t want = want_env ? want_env : strided ? 12u : 6u;
    uint32_t per = (n_tiles / want + G - 1u) / G * G;
    if (per == 0 || n_tiles < 4u * G) return 1;                                      // small frames: the serial path
    const uint32_t n_chunks = (n_tiles + per - 1u) / per;
    if (!g.stream2) HIPCHK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
    while (g.chunk_ev.size() < n_chunks) { hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); g.chunk_ev.push_back(ev); }
    const uint32_t UB = fe == FE_PIXELS ? 6u : 9u;
    const uint64_t in_bytes = n_units * UB;
    std::atomic<uint32_t> launched{0}; std::atomic<int> abort_dl{0};
    hipError_t dl_err = hipSuccess;
    const int dev = g.dev; hipStream_t s2 = g.stream2; const std::vector<hipEvent_t>& evs = g.chunk_ev;
    uint8_t* const ho = (uint8_t*)out; const uint8_t* const dob = (const uint8_t*)dout;
    const uint32_t nb = e0.a.nb_uniform;
    std::thread dl([&] {
        if (hipSetDevice(dev) != hipSuccess) { dl_err = hipErrorInvalidDevice; return; }
        for (uint32_t c = 0; c < n_chunks; ++c) {
            while (launched.load(std::memory_order_acquire) <= c) { if (abort_dl.load()) return; std::this_thread::yield(); }
            hipError_t er = hipEventSynchronize(evs[c]);
            const uint64_t t0 = (uint64_t)c * per, t1 = std::min<uint64_t>(n_tiles, t0 + per);
            // nine band runs; one strided copy when the bands are equally long and the runs whole (every copy costs ~20 us of this thread)
            const uint64_t wbytes = 26 * (t1 - t0) * nb, o2 = hs + L.band_body_off[0] + 26 * t0 * nb;
            if (strided && t1 * nb <= L.band_blocks[0] && wbytes % 4u == 0 && o2 % 4u == 0 && er == hipSuccess) {
                const uint64_t pitch = 26 * L.band_blocks[0];
                er = hipMemcpy2DAsync(ho + o2, pitch, dob + o2, pitch, wbytes, 9, hipMemcpyDeviceToHost, s2);
            } else for (int b = 0; b < 9 && er == hipSuccess; ++b) {
                const uint64_t lo = std::min<uint64_t>(L.band_blocks[b], t0 * nb), hi = std::min<uint64_t>(L.band_blocks[b], t1 * nb);
                if (hi > lo) er = hipMemcpyAsync(ho + hs + L.band_body_off[b] + 26 * lo, dob + hs + L.band_body_off[b] + 26 * lo, 26 * (hi - lo), hipMemcpyDeviceToHost, s2);
            }
            if (c == 0 && er == hipSuccess) {                                          // header and the zero tail of the last word: written by the first chunk's first workgroup
                er = hipMemcpyAsync(ho, dob, hs, hipMemcpyDeviceToHost, s2);
                const uint64_t tail = 9 * L.out_words - (hs + L.body_syms);
                if (er == hipSuccess && tail) er = hipMemcpyAsync(ho + hs + L.body_syms, dob + hs + L.body_syms, tail, hipMemcpyDeviceToHost, s2);
            }
            if (er != hipSuccess) { dl_err = er; return; }
        }
        dl_err = hipStreamSynchronize(s2);
    });
    int rc = T3_OK; uint64_t up_done = 0;
    for (uint32_t c = 0; c < n_chunks && rc == T3_OK; ++c) {
        const uint32_t t0 = c * per, t1 = std::min<uint32_t>(n_tiles, t0 + per);
        const uint64_t S_lo = (uint64_t)t0 * TS, S_hi = (uint64_t)t1 * TS;
        const uint64_t off_lo = fe == FE_PIXELS ? S_lo / 13 * 18 : S_lo / 26 * 27;   // input bytes in front of the chunk (exact: S_lo is a multiple of the unit)
        const uint64_t up_hi = t1 == n_tiles ? in_bytes : std::min<uint64_t>(in_bytes, fe == FE_PIXELS ? S_hi / 13 * 18 : S_hi / 26 * 27);
        if (up_hi > up_done) { const hipError_t er = hipMemcpyAsync((uint8_t*)di + up_done, (const uint8_t*)in + up_done, up_hi - up_done, hipMemcpyHostToDevice, g.stream); if (er != hipSuccess) { rc = fail_hip(er, "hipMemcpyAsync(chunk upload)"); break; } up_done = up_hi; }
        EncLaunch e = e0;
        const uint64_t u_lo = off_lo / UB;                                            // pixels / words in front of the chunk
        e.a.in = (const uint8_t*)di + off_lo;
        e.a.n_units = n_units > u_lo ? n_units - u_lo : 0; e.a.n_units_pad = (fe_px(fe) ? 2 * n_raw : n_units) - u_lo;
        e.a.n_sym = (uint32_t)(L.n_sym > S_lo ? L.n_sym - S_lo : 0);
        e.a.n_tiles = t1 - t0;
        for (int b = 0; b < 9; ++b) {
            const uint64_t skip = (uint64_t)t0 * e0.a.band_nb_tile[b];
            e.a.band_blocks[b] = (uint32_t)(L.band_blocks[b] > skip ? L.band_blocks[b] - skip : 0);
            e.a.band_body_off[b] = L.band_body_off[b] + 26 * skip;
            e.a.band_boff6[b] = (uint32_t)((e.a.band_body_off[b] + 4) % 6);
        }
        e.a.afrag = lut->d_afrag; e.a.lut_img = lut->d_img;
        e.a.body_out = (uint8_t*)dout + hs; e.a.frame_out = c == 0 ? (uint8_t*)dout : nullptr;
        e.a.hdr_syms = hs; e.a.pad_bytes = (uint32_t)(9 * L.out_words - L.out_syms); e.a.out_syms = L.out_syms; memcpy(e.a.hdr, hdr, sizeof hdr);
        { std::lock_guard<std::mutex> lk(g.mu); rc = launch_enc_fe(fe, e, g.stream); }
        if (rc == T3_OK) { const hipError_t er = hipEventRecord(evs[c], g.stream); if (er != hipSuccess) rc = fail_hip(er, "hipEventRecord"); }
        if (rc == T3_OK) launched.store(c + 1, std::memory_order_release);
    }
    if (rc != T3_OK) abort_dl.store(1);
    dl.join();
    if (rc == T3_OK && dl_err != hipSuccess) rc = fail_hip(dl_err, "chunk download");
    if (rc == T3_OK) HIPCHK(hipStreamSynchronize(g.stream));
    return rc;
}

static int encode_host(int fe, const void* in, uint64_t n_units, const t3_cfg* cfg, void* out, uint64_t cap, uint64_t* n_out) {
    if (!g.ready) return T3_E_NODEVICE;
    if (!cfg || !n_out || (n_units && !in)) return T3_E_ARG;
    const uint64_t n_raw = fe == FE_PIXELS ? (n_units + 1) / 2 : n_units;
    t3_layout L; int rc = plan(n_raw, *cfg, L); if (rc) return rc;
    *n_out = L.out_words; if (L.out_words > cap) return T3_E_CAPACITY;
    std::lock_guard<std::recursive_mutex> hl(g.host_mu);
    void *di, *dout;
    { std::lock_guard<std::mutex> lk(g.mu); rc = scratch(0, n_units * (fe == FE_PIXELS ? 6 : 9) + 64, &di); if (rc) return rc; rc = scratch(1, L.out_words * 9 + 64, &dout); if (rc) return rc; }
    rc = encode_host_pipelined(fe, in, n_units, cfg, out, L, di, dout);                 // 1: not this framing / too small -> one upload, one launch, one download
    if (rc != 1) return rc;
    if (n_units) HIPCHK(hipMemcpyAsync(di, in, n_units * (fe == FE_PIXELS ? 6 : 9), hipMemcpyHostToDevice, g.stream));
    rc = encode_dev(fe, di, n_units, cfg, dout, L.out_words, n_out, g.stream); if (rc) return rc;
    if (L.out_words) HIPCHK(hipMemcpyAsync(out, dout, L.out_words * 9, hipMemcpyDeviceToHost, g.stream));
    HIPCHK(hipStreamSynchronize(g.stream)); return T3_OK;
}
int t3hip_encode_profile(const void* raw, uint64_t n_raw, const t3_cfg* cfg, void* out, uint64_t cap, uint64_t* n_out) { return encode_host(FE_WORDS, raw, n_raw, cfg, out, cap, n_out); }
int t3hip_encode_frame(const void* px, uint64_t n_px, const t3_cfg* cfg, void* out, uint64_t cap, uint64_t* n_out) { return encode_host(FE_PIXELS, px, n_px, cfg, out, cap, n_out); }

// ---- timing helper ------------------------------------------------------------------------------------------
int t3hip_event_create(void** ev) { if (!g.ready) return T3_E_NODEVICE; hipEvent_t e; HIPCHK(hipEventCreate(&e)); *ev = e; return T3_OK; }
int t3hip_event_record(void* ev, void* stream) { HIPCHK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return T3_OK; }
int t3hip_event_elapsed_ms(void* a, void* b, float* ms) { HIPCHK(hipEventSynchronize((hipEvent_t)b)); HIPCHK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return T3_OK; }
int t3hip_event_destroy(void* ev) { HIPCHK(hipEventDestroy((hipEvent_t)ev)); return T3_OK; }

}  // extern "C"

// decode-side entry points live in t3_api_decode.cpp (same library)
namespace t3 {
int api_encode_rgb_fused(const void* d_rgb, uint64_t n_px, const t3_cfg* cfg, void* d_out, uint64_t cap, uint64_t* n_out, hipStream_t s) { return encode_dev(FE_RGB, d_rgb, n_px, cfg, d_out, cap, n_out, s); }
int api_ready() { return g.ready ? 1 : 0; }
hipStream_t api_stream() { return g.stream; }
int api_scratch(int slot, size_t bytes, void** out, hipStream_t s) { std::lock_guard<std::mutex> lk(g.mu); return scratch(slot, bytes, out, s); }
int api_fail_hip(hipError_t e, const char* what) { return fail_hip(e, what); }
uint32_t* api_flag() { return g.d_flag; }
RsTables* api_tables() { return g.d_tab; }
int api_n_cu() { return g.n_cu; }
int api_device() { return g.dev; }
std::recursive_mutex& api_host_mutex() { return g.host_mu; }
std::mutex& api_tab_mutex() { return g.tab_mu; }
std::mutex& api_qt_mutex() { return g.qt_mu; }
std::recursive_mutex& api_mail_mutex() { return g.mail_mu; }
void*& api_slot(int id) { return g.slot[id]; }
uint32_t* api_ticket_counters(hipStream_t s, int kind) { std::lock_guard<std::mutex> lk(g.mu); return ticket_counters(s, kind); }
// the download stream and per-chunk events of the pipelined host entry points (the caller holds the context's host mutex)
int api_pipeline(uint32_t n_events, hipStream_t* s2, hipEvent_t** evs) {
    if (!g.stream2) HIPCHK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
    while (g.chunk_ev.size() < n_events) { hipEvent_t ev; HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); g.chunk_ev.push_back(ev); }
    *s2 = g.stream2; *evs = g.chunk_ev.data(); return T3_OK;
}
}  // namespace t3
