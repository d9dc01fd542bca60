// t3_device.h — argument blocks shared by the host launcher (t3_api.cpp) and the gfx950 kernels
// (t3_kernels.hip).  Plain structs passed by value as kernel arguments.
#pragma once
#include <stdint.h>

namespace t3 {

enum FrontEnd : int { FE_PIXELS = 0, FE_WORDS = 1, FE_RGB = 2 };   // FE_RGB: RGB8 in, the io_image.hpp bridge fused into phase 1 (pipelined flow only)
constexpr bool fe_px(int fe) { return fe == FE_PIXELS || fe == FE_RGB; }          // pixel-shaped front ends: 6 px = one lane group of 26 symbols
constexpr int kGroupBytesRgb = 18;

constexpr int kMaxWaves = 16;            // 1024-thread workgroup
constexpr int kLdsHdr = 512;            // LDS header: 9 band rows (32 B), item prefix, ticket slot, scrambler dwords
constexpr int kLdsHdrUep = 1024;        // ... of the UEP matrix-core kernel: + group records and the set table
constexpr int kHdrGrp = 384, kHdrGrpStride = 96, kHdrSets = 768;   // header offsets of the group records and the set table (UEP matrix-core kernel)
constexpr int kMaxGrp = 4, kMaxSets = 16;
constexpr int kSymFront = 64;           // slack in front of the LDS symbol buffer (phase 1 writes whole pixel triples) ...
constexpr int kSymBack = 64;            // ... and behind it
constexpr int kSymSlackW = 112;         // raw words (1-D, packed converter): a lane writes four word triples = 104 symbols whole, either side
constexpr int kGroupSyms = 26;           // symbols one phase-1 lane produces from pixels: 6 px = 36 B -> 26 symbols
constexpr int kGroupBytes = 36;
constexpr int kGroupSymsW = 52;          // from raw words: 6 words = 54 B -> 52 symbols
constexpr int kGroupBytesW = 54;

struct DevDiv { uint32_t mul, sh, d; };

// K2: fused [pack ->] regroup -> [2-D interleave] -> 9-band split -> RS parity -> scramble -> band-serial emit
struct EncArgs {
    const uint8_t* in;              // pixels (6 B each) or raw Word27 (9 B each), 16-B aligned
    uint8_t*  body_out;             // address of body symbol 0 (final stream + header, or scratch when a beacon pass follows)
    uint8_t*  frame_out;            // final stream base; header/pad written by tile 0 when non-null
    const uint32_t* lut_img;        // LDS table image: concatenated LUTs of the k's in use (mixed k) or the T/M tables (single k)
    const uint32_t* afrag;          // single-k launches: matrix operand of the parity MFMA, 3 steps x 64 lanes x 4 dwords
    uint32_t  lut_bytes;            // total bytes to stage into LDS (multiple of 16)
    uint64_t  n_units;              // real pixels / words in `in`
    uint64_t  n_units_pad;          // pixels: 2*n_words (odd count pads one zero pixel, OLD:730); words: = n_units
    uint32_t  n_sym;                // stream symbols (ceil(26 W / 3))
    uint32_t  n_tiles;
    uint32_t  Lq;                   // data symbols per band per tile; tile = 9*Lq stream symbols
    uint32_t  band_k[9];
    uint32_t  band_nb_tile[9];      // blocks per tile
    uint32_t  band_blocks[9];       // blocks in the frame
    uint32_t  band_lut_off[9];      // LDS byte offset of the band's LUT
    uint32_t  band_boff6[9];        // (band_body_off + 4) % 6 : scrambler cycle phase of the band's first symbol
    uint64_t  band_body_off[9];
    uint32_t  band_first[10];       // phase-2 work item -> band: band b owns items [band_first[b], band_first[b+1]); one lane = one block
    uint32_t  n_items;              // = band_first[9] = sum of blocks per tile
    uint32_t  nb_uniform;           // blocks per band per tile when all bands share k (item / nb_uniform = band), else 0
    DevDiv    div_nb;
    uint32_t  sym_off, stage_off, lds_bytes;   // LDS carve-up: [hdr][LUT][symbols][stage 0][stage 1]
    uint32_t  stage_stride;         // bytes between the two input stage buffers (out staging aliases the current one)
    uint32_t  stage_groups;         // capacity of one input stage buffer, in lane groups
    uint32_t  cyc24; uint32_t pre0, pre1;      // scrambler: 6-periodic tail as 2-bit fields (x2), two pre-period states
    uint32_t  scr[12];              // single-k launches: scrambler dwords of the parity symbols per phase (mfma_scrambler_table)
    // UEP on the matrix cores: bands grouped by k; a group's blocks of a tile are dealt linearly into sets of 32
    struct Grp { uint32_t nb; DevDiv div_nb; uint32_t n_items, r; uint8_t bands[12]; uint32_t afrag_off; uint32_t scr[12]; } grp[kMaxGrp];   // afrag_off: LDS offset of the group's A operand
    uint32_t  n_grp, n_sets;
    uint32_t  set_tab[kMaxSets];    // group | first item << 8
    uint32_t  il_on, il_w, il_A;    // 2-D boustrophedon: row width, chunk area (clamped to n_sym)
    uint32_t  il_async;             // 2-D through the pipelined flow (pixels, moderate row width): symbol and stage buffers sized for the rows a tile overlaps
    DevDiv    div_A, div_w;
    uint32_t  hdr_syms; uint32_t pad_bytes;    // header symbols; zero bytes after the last symbol (OLD:1164-1167)
    uint64_t  out_syms;
    uint8_t   hdr[96];
    uint32_t* tile_ctr;             // ticket counters [64 * class], then workgroups-finished at [64 * n_classes] (last one re-zeroes); null = static striding
    uint32_t  n_classes;            // ticket classes (<= grid)
    uint32_t  p1_wpp;               // phase 1 (pixels): waves that cover a tile (one lane = four pixel triples)
    uint32_t  qt_off;               // FE_RGB: LDS offset of the chroma quantiser table (256 bytes: C -> Cq + 40, io_image.hpp:73-76)
    const uint8_t* qt;              // ... and its device image
    // beacon insertion fused into the stores (BCN kernels, OLD:1118-1141): a beacon symbol sits in front of body byte bcn_slot + j bcn_pb
    // (bcn_pb = 9 period - 1 >= 17), the stores go to framed offsets; the few framed bytes after the last body byte come from the host
    uint32_t  bcn_slot, bcn_pb, bcn_sym; DevDiv bcn_div;
    uint32_t  bcn_tail_len; uint64_t bcn_tail_off, bcn_tail_vals;   // frame_out offset, up to 8 byte values (low byte first)
    uint64_t* dbg;                  // diagnostic stamp builds only (T3_STAMPS); null in the product
};

// beacon insertion pass (OLD:1118-1141): framed[q] = beacon | body[q - #beacons before q] | 0
struct BeaconArgs {
    const uint8_t* body; uint8_t* frame_out;
    uint64_t body_syms, framed_syms;
    uint32_t period, slot, sym, hdr_syms, pad_bytes;
    uint8_t  hdr[96];
};

}  // namespace t3
