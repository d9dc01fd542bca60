// t3_decode.h — decode-side argument blocks and kernel declarations.
#pragma once
#include <stdint.h>

#include "t3_device.h"
#include "t3_rs_core.h"

namespace t3 {

// Where the reference decoder (COMPAT, OLD:948-993) or the v6c framing (FIXED) keeps block (b, m), and where its
// k corrected data symbols go.
struct DecArgs {
    const uint8_t* in;           // coded stream (whole frame, header included)
    uint8_t* use;                // scratch: COMPAT band-serial `use` vector (OLD:988) / FIXED stream-ordered symbols
    uint32_t* fail;              // incremented per uncorrectable block (decode_block false, OLD:987)
    const RsTables* tab;
    uint32_t fixed;              // 0 COMPAT framing + Forney add; 1 v6c framing + Forney sub
    uint32_t hdr_syms;           // 54 (six words, OLD:920-924) / 90
    uint32_t band_k[9];
    uint64_t band_blocks[9];
    uint64_t band_first[9];      // prefix sum of band_blocks (work item -> band)
    uint64_t band_off[9];        // COMPAT: use_off[b]; FIXED: band_body_off[b]
    uint64_t total_blocks;
    uint64_t n_sym;              // FIXED: stream symbols (9j+b >= n_sym is padding)
    uint32_t beacon_on, period, slot;
    uint32_t cyc24, pre0, pre1;  // scrambler (subtracting)
};

struct EmitArgs {
    const uint8_t* use; uint64_t use_syms;    // symbols available (COMPAT: sum k*blocks; FIXED: n_sym)
    void* out; uint64_t n_words;              // words to emit (COMPAT: 3*use_syms/26, OLD:1030; FIXED: n_raw)
    uint32_t to_pixels;                       // 0: Word27 (9 B), 1: two pixels (12 B)
    uint32_t il_on, il_w, il_A; DevDiv div_A, div_w;
};

constexpr int kCrcPows = 40;                     // "append 2^j zero bytes" operators, j < kCrcPows
struct CrcArgs {
    const uint8_t* data; uint64_t n_bytes; uint32_t chunk_bytes; uint32_t n_chunks;
    uint32_t* chunk_crc; uint32_t* sym_sum;      // device accumulators, zeroed by the launcher
    const uint32_t* zpow;                        // [kCrcPows][32] operator columns (device)
};

// CRC-32 on the matrix cores (t3_crc_mfma.hip): whole 2 KiB rounds of a 16-byte aligned stream; the tail goes to crc_chunks_kernel
struct CrcMArgs {
    const uint8_t* data; uint64_t n_bytes;       // whole stream (distance to its end)
    uint32_t n_rounds, rounds_per_wave;          // 2 KiB rounds in total / per wave
    uint32_t stride_waves;                       // FP4 kernel: 0 = a wave owns rounds_per_wave consecutive rounds; W > 0 = wave g owns rounds g, g + W, g + 2 W ..
    const uint32_t* afb;                         // ... and its feedback slice [64][4] ("append 2048 W zero bytes") comes from here
    const uint32_t* afrag;                       // [22][64][4]: 16 data slices, the feedback slice, five "append 64 * 2^b bytes" slices, in MFMA lane order
    const uint32_t* zpow;
    uint32_t* chunk_crc; uint32_t* sym_sum;
    uint32_t* partials;                          // FP4 kernel, != null: workgroup g stores its (xor, sum) at [2 g], [2 g + 1] instead of adding to the two accumulators (no zeroing pass, no atomics)
};

// Fused FIXED-mode decoder (uniform k, 1-D, no beacon): one tile = 9 bands x nb blocks -> a word-aligned slice of the
// output (27 * Lq trits, Lq = nb * k a multiple of 26).  LDS: [band rows][FxTables][syndrome LUT][symbols Y][out staging].
struct FxTables {                 // field tables the in-kernel corrector indexes (LDS resident)
    uint8_t mul[729], add[729], sub[729];
    uint8_t inv[27], neg[27], exp[26];
    uint8_t descr[3][32];         // descr[s][c] = 4 * (c - (s,s,s) trit-wise): descrambled symbol, pre-scaled (LUT entry offset)
    uint8_t pad_[37];
};
static_assert(sizeof(FxTables) == 2400, "FxTables layout");
constexpr int kFxHdr = 512, kFxTab = kFxHdr, kFxLut = 3072;     // LDS byte offsets (kFxTab + 2400 <= kFxLut)
struct DecFxArgs {
    const uint8_t* in; uint64_t in_bytes;      // whole coded stream
    void* out; uint64_t n_units;               // pixels (to_pixels) or words to emit
    uint32_t* fail;
    const FxTables* tab; const uint32_t* lut; uint32_t lut_bytes;
    const uint8_t* fma; uint32_t fma_off;      // [27][27][27]: fma[x][y][a] = a + x y in GF(27) (19683 bytes), and its LDS offset
    const uint32_t* roots;                     // [27^t]: bit i set <=> 1 + s1 x + .. + st x^t vanishes at alpha^-i, index s1 + 27 s2 + ..
    uint32_t k, nb, n_tiles, TS;               // nb blocks per band per tile (multiple of 13), TS = 9*nb*k stream symbols
    uint32_t n_sym;                            // real stream symbols (the rest of the last blocks is zero padding)
    uint32_t hdr_syms;
    uint32_t band_blocks[9]; uint64_t band_body_off[9]; uint32_t band_boff6[9];
    uint32_t cyc24, pre0, pre1;
    uint32_t y_off, o_off, lds_bytes;
};

// Fused FIXED-mode decoder, second version (t3_decode_fused.hip): syndromes on the matrix cores (two lanes per block, T table =
// descramble + trit expansion in one read), single errors fixed in closed form by the lane that owns the block, the remaining
// flagged blocks compacted (wave ballot) into an LDS queue and corrected by full waves.  Two kernels share the block stages:
//   decode_fixed_px_kernel  (pixels out)  waves 0-3 produce (sets, single errors, queue) tile k while waves 4-7 consume
//                           (queue -> Berlekamp-Massey, symbols -> pixels) tile k-1: double-buffered symbols and queues, one
//                           workgroup barrier per tile.  LDS: [hdr 0][fold tables 512][T16 1024][FMA][A operand][Y0][Y1][Q0][Q1]
//   decode_fixed_kernel     (raw words out) the phases run one after the other (sets | queue | words staged in LDS).
//                           LDS: [hdr 0][fold tables 3072][T32 3584][FMA][A operand][Y][Q][words]
// hdr: band rows 0..143, queue counters 160/164, consumer rendezvous 168, give-up flag 172, ticket slots 176/180, small byte tables 192, dummy 384..511
constexpr int kFx2Cnt = 160, kFx2Sync = 168, kFx2Abort = 172, kFx2Next = 176, kFx2FailWg = 184, kFx2Small = 192;   // kFx2Next: two words, the ticket slot of each barrier parity
// Geometry of the pixel kernel (decode_fixed_px_kernel): threads per workgroup (half producer, half consumer waves), bank copies of
// its T table, blocks per band per tile, queue capacity.  Measured in round 3 (profiles/r03/notes.md): 768 threads x two workgroups per
// CU with 32 conflict-free T copies and 78 blocks per band (-DT3_DEC_PX_THREADS=768 -DT3_DEC_PX_TCOP=32 -DT3_DEC_PX_NB=78
// -DT3_DEC_QCAP=400) is no faster than three 512-thread workgroups: the kernel is bound by vector-instruction issue, not by LDS conflicts.
#ifndef T3_DEC_PX_THREADS
#define T3_DEC_PX_THREADS 512
#endif
#ifndef T3_DEC_PX_TCOP
#define T3_DEC_PX_TCOP 16
#endif
#ifndef T3_DEC_PX_NB
#define T3_DEC_PX_NB 52
#endif
#ifndef T3_DEC_QCAP
#define T3_DEC_QCAP 256
#endif
constexpr int kFx2ModSeq = 3072, kFx2TSeq = 3584, kFx2ModPx = 512, kFx2TPx = 1024, kFx2QCap = T3_DEC_QCAP;
static_assert(T3_DEC_PX_THREADS % 128 == 0 && T3_DEC_PX_THREADS <= 1024 && T3_DEC_PX_NB % 26 == 0 && 9 * T3_DEC_PX_NB <= T3_DEC_PX_THREADS && T3_DEC_PX_NB < 2048, "pixel-decoder geometry");
struct DecFx2Args {
    const uint8_t* in; uint64_t in_bytes;      // whole coded stream
    void* out; uint64_t n_units;               // pixels or words to emit
    uint32_t* fail;
    const uint32_t* ttab; const uint8_t* small; const uint32_t* afrag;   // device images (t3_host.hpp): T (32 or 16 copies), [byte tables][fold tables], A operand
    const uint8_t* fma; uint32_t fma_off;      // [27][27][27]: fma[x][y][a] = a + x y in GF(27) (19683 bytes), and its LDS offset
    const uint32_t* roots;                     // [27^t]: bit i set <=> 1 + s1 x + .. + st x^t vanishes at alpha^-i, index s1 + 27 s2 + ..
    uint32_t k, nb, n_tiles, TS;               // nb blocks per band per tile (multiple of 13), TS = 9*nb*k stream symbols
    DevDiv div_nb;
    uint32_t n_sym;                            // real stream symbols (the rest of the last blocks is zero padding)
    uint32_t hdr_syms;
    uint32_t band_blocks[9]; uint64_t band_body_off[9]; uint32_t band_boff6[9];
    uint32_t cyc24, pre0, pre1;
    uint32_t* tile_ctr; uint32_t n_classes;    // px kernel: dynamic tile tickets (n_classes counters + a done counter, 256 B apart; zero between launches); null = static stride
    // px kernel with tickets, streaming entry (t3hip_decode_frame_async): the header check and the verdict words in the same launch.  verdict != null:
    // workgroup 0 compares the stream's first hdr_n bytes (hdr_in, 16-byte aligned) with hx and writes verdict[0]; `fail` then points at a library-owned
    // counter beside the tickets (zero between launches) and the workgroup that finishes last moves it to verdict[1] and re-zeroes it
    uint32_t* verdict; const uint8_t* hdr_in; uint32_t hdr_n; uint32_t hx[24];
    uint32_t pat[48], pat_off;                 // twelve 16-byte rows: byte q of row r < 11 = 27 x scrambler state of a position of class (r + q) mod 6; row 11: the stream's first block (pre-period states in bytes 0, 1)
    uint32_t y_off, y_stride, q_off, q_stride, o_off, af_off, lds_bytes;   // px kernel: two symbol buffers / queues, y_stride / q_stride apart
    uint32_t bcn_slot, bcn_pb; DevDiv bcn_div;   // BCN kernels: a beacon symbol sits in front of body byte bcn_slot + j bcn_pb (bcn_pb = 9 period - 1 >= 17); `in` is the framed stream
    const uint8_t* dq; uint32_t dq_off;          // RGB out (row f1 fused): dequantiser tables yd[244] | cd[84] (t3_rgb.h) and their LDS offset
    uint64_t* dbg;                             // diagnostic stamp builds only (T3_DEC_STAMPS); null in the product
};

// One-launch FIXED decoder for per-band k (two codes) and / or the 2-D interleave, pixels out (t3_decode_uep.hip): the block stages of the
// uniform-k kernel with the bands grouped by k, odd row pieces reversed in LDS, whole pixel triples of the tile's pre-interleave runs
// emitted from LDS and the triples the runs' ends cut through completed by uep_edge_kernel from a sparse scratch.
// LDS: [hdr 0][fold tables 512][T16 1024][FMA][A operand per group][pattern rows][group records, pair table][Y0][Y1][Q0][Q1]
constexpr int kUepMaxGrp = 2;
struct DecUepArgs {
    const uint8_t* in; uint64_t in_bytes;      // coded body (hdr_syms symbols of header in front)
    void* out; uint64_t n_units;               // pixels
    uint32_t* fail;
    const uint32_t* ttab; const uint8_t* small; const uint8_t* fma; uint32_t fma_off;
    struct Grp { uint32_t r, nb, n_items, af_off, q_rel, q_cap; DevDiv div_nb; const uint32_t* afrag; const uint32_t* roots; uint8_t bands[12]; } grp[kUepMaxGrp];   // q_rel: the group's queue inside a buffer's queue area
    uint32_t n_grp;
    uint32_t pair_tab[8];                      // slot 2 wave + pass -> group | first item << 8 (two sets of 32 blocks of one group); 0xFFFFFFFF: none
    uint32_t TS, n_tiles, n_sym, hdr_syms;
    uint32_t band_blocks[9]; uint64_t band_body_off[9]; uint32_t band_boff6[9];
    uint32_t pat[48], pat_off, rec_off;
    uint32_t y_off, y_stride, q_off, q_stride, lds_bytes;
    uint32_t* tile_ctr; uint32_t n_classes;
    uint32_t* verdict; const uint8_t* hdr_in; uint32_t hdr_n; uint32_t hx[24];   // header check + verdict words in this launch (as DecFx2Args)
    uint32_t il_on, il_w, il_A; DevDiv div_A, div_w;
    uint8_t* edge;                             // global scratch, n_sym bytes, written sparsely: the symbols of the triples the runs' ends cut through
};

// Two-kernel FIXED decoder for the framings the fully fused kernel does not take (mixed k, 2-D interleave; a beacon is
// stripped by a pre-pass): D1-D4 as above with the bands grouped by k (whole waves per group), corrected data symbols to a
// stream-ordered scratch; then symbols [through the de-interleave map] -> units.  (t3_decode_stream.hip)
constexpr int kStMaxGrp = 4;
struct DecStArgs {
    const uint8_t* in; uint64_t in_bytes;      // coded body (hdr_syms symbols of header in front)
    uint8_t* ystream;                          // scratch: n_sym corrected data symbols in stream order
    uint32_t* fail;
    const FxTables* tab; const uint8_t* fma; uint32_t fma_off;
    struct Grp { uint32_t r, nb, n_items, wave0, n_waves, lut_off, lut_bytes; const uint32_t* lut; const uint32_t* roots; uint8_t bands[12]; } grp[kStMaxGrp];
    uint32_t n_grp, n_slots;                   // wave slots of a tile: group g owns [wave0, wave0 + n_waves)
    uint32_t n_tiles, TS;                      // TS = 9 Lq stream symbols per tile, Lq = nb_g k_g for every group
    uint32_t n_sym, hdr_syms;
    uint32_t band_blocks[9]; uint64_t band_body_off[9]; uint32_t band_boff6[9];
    uint32_t cyc24, pre0, pre1;
    uint32_t y_off, lds_bytes;
};
struct EmitStArgs {
    const uint8_t* ystream; uint32_t n_sym;    // 16-byte aligned
    void* out; uint64_t n_units;               // pixels or words
    uint32_t span;                             // symbols per workgroup step (multiple of 52 x lanes resp. 26 x lanes)
    uint32_t n_steps;
    uint32_t il_on, il_w, il_A, il_fast;       // il_fast: rows and chunks are multiples of 16 symbols -> granule-wise permutation
    DevDiv div_A, div_w;
    uint32_t sym_off, o_off, lds_bytes;
};
struct HdrExpect { uint8_t b[96]; };          // the header symbols a configuration encodes to, passed by value
struct DebeaconArgs { const uint8_t* framed; uint64_t framed_bytes; uint8_t* body; uint64_t body_syms; uint32_t period, slot; };   // framed_bytes: readable bytes from `framed`

int decode_init(const RsTables* d_tab);
void decode_shutdown();                       // frees what decode_init and the lazily built decoder tables allocated

#if defined(__HIPCC__)
__global__ void dec_gather_rs_kernel(const DecArgs a);
template <int R, bool BCN> __global__ void decode_fixed_kernel(const DecFx2Args a);        // BCN: beacon symbols stepped over in the loads
template <int R, bool RGB, bool BCN> __global__ void decode_fixed_px_kernel(const DecFx2Args a);
__global__ void decode_stream_kernel(const DecStArgs a);
template <int RA, int RB> __global__ void decode_uep_px_kernel(const DecUepArgs a);    // RA >= RB: r of group 0 / 1
__global__ void uep_edge_kernel(const DecUepArgs a);
template <bool TO_PIXELS> __global__ void emit_stream_kernel(const EmitStArgs a);
__global__ void debeacon_kernel(const DebeaconArgs a);
__global__ void dec_emit_kernel(const EmitArgs a);
__global__ void rs_decode_blocks_kernel(uint8_t* code, uint64_t n_blocks, int k, int fixed, const RsTables* tab, uint8_t* data, uint8_t* ok);
__global__ void inject_errors_kernel(uint8_t* syms, uint64_t n_blocks, uint32_t seed, int max_err);
__global__ void hdr_compare_kernel(const uint8_t* in, HdrExpect expect, uint32_t n, uint32_t* mismatch);
__global__ void crc_chunks_kernel(const CrcArgs a);
__global__ void crc_mfma_kernel(const CrcMArgs a);
__global__ void crc_fp4_kernel(const CrcMArgs a);          // the same on the FP4 matrix instruction (t3_crc_fp4.hip); afrag = [14][64][4] FP4 slices
__global__ void frame_record_kernel(const uint32_t* acc, uint32_t lead, const uint8_t* tail, uint32_t tail_len, const uint32_t* zpow,
                                    const uint8_t* words, uint64_t n_words, uint64_t frame_idx, uint32_t profile, uint32_t mode, void* rec, const uint32_t* partials, uint32_t n_partials);
#endif

}  // namespace t3
