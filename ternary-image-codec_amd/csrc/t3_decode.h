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

int decode_init(const RsTables* d_tab);

#if defined(__HIPCC__)
__global__ void dec_gather_rs_kernel(const DecArgs a);
__global__ void dec_emit_kernel(const EmitArgs a);
__global__ void rs_decode_blocks_kernel(uint8_t* code, uint64_t n_blocks, int k, int fixed, const RsTables* tab, uint8_t* data, uint8_t* ok);
__global__ void inject_errors_kernel(uint8_t* syms, uint64_t n_blocks, uint32_t seed, int max_err);
__global__ void crc_chunks_kernel(const CrcArgs a);
__global__ void frame_record_kernel(const CrcArgs a, const uint8_t* words, uint64_t n_words, uint64_t frame_idx, uint32_t profile, uint32_t mode, void* rec);
#endif

}  // namespace t3
