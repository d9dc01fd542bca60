// t3_host.hpp — host-side constants and closed-form stream geometry for the Word27 path.
// Pure C++17 (no HIP): field tables, generator / parity matrices, kernel LUT images, scrambler cycle,
// 27-symbol header + ternary CRC-12, frame layout and encode-tile planning.
// OLD:n = reference old/include/ternary_image_codec_v6_min.hpp line n (cited for parity checks).
#pragma once
#include <stdint.h>
#include <vector>

#include "../../include/t3hip.h"
#include "t3_rs_core.h"

namespace t3 {

// ---- GF(27) ------------------------------------------------------------------------------------
struct Field {
    RsTables t;            // mul/add/neg/inv/exp
    int16_t  log[27];
    uint8_t  exp78[78];    // the reference's tripled exp table (OLD:457)
    uint8_t  prim;
};
const Field& field();
inline RsView rs_view(const RsTables& t) { return RsView{t.mul, t.add, t.neg, t.inv, t.exp}; }

// ---- RS(26,k) constants ---------------------------------------------------------------------------
bool valid_k(int k);
int  k_of_band_profile(uint8_t bp);                 // OLD:1089-1100: {24,22,20,18}[bp%4]
void rs_generator(int k, uint8_t* g /*r+1*/);      // OLD:501-516
// P[i*r+j]: parity_j = sum_i d_i P[i][j].  COMPAT = the reference's recurrence applied to unit vectors
// (OLD:517-535, a GF(27)-linear map); FIXED = systematic RS with roots alpha^1..alpha^r.
void rs_parity_matrix(int k, int mode, uint8_t* P);
bool rs_decode_host(int k, uint8_t* c26, bool fixed);   // decode_block OLD:546-662 on the host (header)

// ---- encode LUT image consumed by the HIP kernels -------------------------------
// For data position p and symbol value d, the contribution of d at p to every parity trit, as 6-bit
// SWAR fields (5 per dword), plus the two non-trivial scrambled images of d.
// dword layout ("main" = first min(r,5) parities, plane-major; "extra" = the remaining ones, 3 fields each):
//   dw0/dw1/dw2 : trit 0/1/2 of parity j at bits [6j, 6j+6), j < min(r,5)
//   dw3..       : extra parity e = j-5: trit c at field 3e+c (5 fields per dword)
//   img         : T_s[d] = d + 13 s trit-wise (scramble_symbol OLD:81-87), in the top byte of the variant table entry
struct LutGeom {
    int k, r;
    int n_dw;          // accumulator dwords: r=2,4 -> 3; r=6 -> 4; r=8 -> 5
    int var_dw;        // accumulator dword that lives in the three per-scrambler-state variant tables
    int var_off;       // byte offset of variant table 0 inside the slab (variants are 256 B apart)
    int slab_bytes;    // LDS bytes per data position (27 entries x 8 B per table, tables 256-B aligned: conflict-free b64)
    int total_bytes;   // k * slab_bytes
};
LutGeom lut_geom(int k);
// Image per position p (entries are 8 B so that the LDS index is the pre-scaled symbol 8*d):
//   [A: 27 x {dw0,dw1}] [B (r=8 only): 27 x {dw2,dw3}] [V_0][V_1][V_2], V_s: 27 x {dw2, dw3|img<<24} (r=6) or {dwv|img<<24, 0}.
void build_encode_lut(int k, int mode, std::vector<uint32_t>& image);

// Tables of the matrix-core encoder (single-k launches).  Parity is a GF(3)-linear map of the data trits:
// per block an 3r x 3k matrix-vector product mod 3, done by v_mfma_i32_32x32x32_i8 for 32 blocks at a time.
// Trits travel as signed bytes (2 == -1), so a trit sum lies in [-60, 60]; the accumulators start at 64.
//   afrag : A operand (the matrix), 3 K-steps x 64 lanes x 4 dwords, in the instruction's lane order: lane l = (row m = l&31,
//           K-half kh = l>>5), dword d of step s = data position p = 8s + 4kh + d, byte tt = coefficient of trit tt of that
//           symbol (byte 3 = 0).  Row m holds accumulator register i = (m&3) + 4(m>>3) of lane half hh = (m>>2)&1 (the C/D
//           layout row = (i&3) + 8(i>>2) + 4hh), which is trit i%3 of parity symbol hh*r/2 + i/3 (rows with i >= 3r/2 are zero).
//           r >= 4: the unused positions mfma_scr_pos(k), +1 carry the scrambler states of the parity symbols, one per byte
//           (coefficient 1 on the three trit rows of parity symbol 4 (p - scr_pos) + byte).
//   lds   : [T: 3 scrambler states x 27 symbols x 32 bank copies, dword = trit0 | trit1<<8 | trit2<<16 | scrambled symbol<<24,
//           at v * 4096 + (d * 32 + bank) * 4: every lane reads its own bank, no conflicts]
//           [M_t, t<3: 128 bytes, M_t[x] = 3^t ((x - 64) mod 3)]: the mod-3 fold of a biased trit sum, 32 dwords = 32 banks
constexpr int kMfmaTState = 4096, kMfmaTBytes = 3 * kMfmaTState, kMfmaModOff = kMfmaTBytes, kMfmaLdsBytes = kMfmaTBytes + 3 * 128;
inline int mfma_scr_pos(int k) { return k == 22 ? 22 : 20; }     // k = 24 has no free position (states added after the MFMA)
void build_mfma_encode(int k, int mode, std::vector<uint32_t>& afrag, std::vector<uint32_t>& lds_img);


// Syndrome LUT for the fused decoder: position i (0..25), symbol c -> c * alpha^{(j+1) i} for j < r as 6-bit SWAR trit
// fields, same dword layout as the encode LUT ("main" parities -> syndromes 0..4 plane-major, extras after), two 8-byte
// tables per position (A: {dw0,dw1} at +0, B: {dw2,dw3} at +256; r=8 adds C: {dw4,0} at +512). 26 * slab bytes.
int  syndrome_lut_slab(int k);
void build_syndrome_lut(int k, std::vector<uint32_t>& image);

// Tables of the matrix-core syndrome stage of the fused FIXED decoders (decoder v2).  Syndromes are GF(3)-linear in the
// received trits: S_j = sum_p c_p alpha^{(j+1) p}, per block a 3r x 78 matrix-vector product mod 3, done by four
// v_mfma_i32_32x32x32_i8 for 32 blocks (two lanes per block: lane half kh holds positions 13 kh .. 13 kh + 12).
//   afrag : A operand, 4 K-steps x 64 lanes x 4 dwords: lane l = (row m = l&31, K-half kh = l>>5), dword d of step s = position
//           p = 13 kh + 4 s + d (slots 4 s + d >= 13 are zero); bytes 0, 1, 3 = coefficient of trit 0, 1, 2 of that symbol
//           (signed: 2 == -1), byte 2 = 0 (it meets the descrambled symbol).  Row m = accumulator register
//           i = (m&3) + 4 (m>>3) of lane half hh = (m>>2)&1, which is trit i%3 of syndrome hh r/2 + i/3 (i < 3 r/2).
//   T     : 3 scrambler states x 27 coded symbols x 32 bank copies (every lane reads its own bank), dword = trit0 | trit1<<8 |
//           d<<16 | trit2<<24 of the DEscrambled symbol d = c - (s,s,s); at state * kSyndTState + (c * 32 + bank) * 4
//   small : byte tables LG[27] (log, 0xFF for 0) | EX[26] | INV[27] | NEG[27] | NINV[27] (= -1/x) at kFx2* offsets
constexpr int kSyndTState = 27 * 128, kSyndTBytes = 3 * kSyndTState;
constexpr int kFx2LG = 0, kFx2EX = 32, kFx2INV = 64, kFx2NEG = 96, kFx2NINV = 128, kFx2SmallBytes = 160;
void build_mfma_syndrome(int k, std::vector<uint32_t>& afrag);          // 4 * 64 * 4 dwords
void build_syndrome_T(std::vector<uint32_t>& img, int copies = 32);      // 3 * 27 * copies dwords
void build_fx2_small(uint8_t out[kFx2SmallBytes]);
constexpr int kFx2ModBytes = 3 * 160;                                   // M_t[i] = 3^t ((i - 78) mod 3), i = trit sum + 64 (MFMA bias) + 14 < 160: fold of a biased trit sum
void build_fx2_mod(uint8_t out[kFx2ModBytes]);

// ---- scrambler (OLD:77-94) ------------------------------------------------------------------------
struct ScrCycle {
    uint8_t pre[2];      // state applied to body symbols 0 and 1
    uint8_t cyc[6];      // state applied to body symbol i >= 2: cyc[(i-2)%6]
    uint32_t cyc24;      // cyc as 2-bit fields, repeated twice (rotation by shift)
    uint8_t next[3];     // next-state table in uint32 wrap-around arithmetic
};
ScrCycle scrambler_cycle(uint32_t a, uint32_t b, uint32_t s0);
ScrCycle scrambler_cycle_from_next(const uint8_t next[3], uint32_t s0);
// the scrambler dwords of those positions for each phase c = (phase of the block's first parity symbol): out[2c], out[2c+1]
void mfma_scrambler_table(int k, const ScrCycle& sc, uint32_t out[12]);

// ---- header (OLD:155-380) ---------------------------------------------------------------------------
void crc12(const uint8_t* trits, int n, uint8_t out[12]);
void header_pack(const t3_cfg& c, uint32_t frame_seq, uint32_t band_map_hash, uint8_t s[27]);
bool header_check(const uint8_t s[27]);
void header_unpack(const uint8_t s[27], t3_cfg& out, uint32_t* frame_seq, uint32_t* band_map_hash);
// coded header in front of the body: COMPAT 52 symbols (OLD:1142-1162), FIXED 90
int  header_encode(const t3_cfg& c, uint64_t n_raw_words, uint8_t* out /*>=90*/);
// parse the first 6 (COMPAT) / 10 (FIXED) words of a coded stream. Returns T3_OK or T3_E_HEADER; on success
// `seen` is overwritten as OLD:1006-1013 does and (FIXED) *n_raw and next[] are filled.
int  header_parse(const uint8_t* words, uint64_t n_words, int mode, t3_cfg& seen, uint64_t* n_raw, uint8_t next[3]);
uint8_t beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health);   // OLD:107-113

// ---- frame geometry ----------------------------------------------------------------------------------
int  plan(uint64_t n_raw_words, const t3_cfg& c, t3_layout& L);    // T3_OK / T3_E_ARG
bool want_interleave(const t3_cfg& c);                              // OLD:1083
// COMPAT decoder framing (OLD:948-993): per slot, symbols and whole 26-blocks seen by the reference decoder
struct DecLayoutCompat {
    uint64_t body_words; uint64_t band_syms[9]; uint64_t band_blocks[9]; uint64_t use_off[9]; uint64_t use_syms;
    uint8_t band_k[9]; uint64_t out_words;
};
void plan_decode_compat(uint64_t n_in_words, const t3_cfg& seen, DecLayoutCompat& D);

// Unsigned division by a runtime constant on the device: q = (n * mul) >> 32 >> sh (n < 2^32).
struct FastDiv { uint32_t mul, sh, d; };
FastDiv fastdiv(uint32_t d);

}  // namespace t3
