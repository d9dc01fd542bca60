/*
 * oracle/t3_oracle.h — TEST INFRASTRUCTURE.  Plain-C, scalar, single-thread restatement of the
 * reference's Word27 hot path (OLD = /root/reference/old/include/ternary_image_codec_v6_min.hpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product
 * (ternary-image-codec_amd/) never does.  Pinned against the unmodified reference through
 * oracle/_ref (tests/test_oracle_vs_ref.py) and the committed vectors in tests/golden/.
 * Exceptions: the FIXED-mode ("v6c") functions restate this repo's own spec (the reference has no such behaviour), and
 * t3o_rgb_to_quant / t3o_quant_to_rgb (row f1) are PARITY UNPINNED: old/include/io_image.hpp does not compile in this
 * image (ImageU8 has no member `swap`, :218), so they are checked only against an independent numpy restatement of the
 * same expressions on all 2^24 inputs (tests/test_oracle_rgb.py).
 */
#ifndef T3_ORACLE_H
#define T3_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#include "../include/t3hip.h" /* t3_cfg (POD) and status codes only */

#ifdef __cplusplus
extern "C" {
#endif

/* field + RS */
void    t3o_gf_tables(uint8_t* exp78, int16_t* log27, uint8_t* mul729, uint8_t* inv27, uint8_t* prim);
uint8_t t3o_gf_add(uint8_t a, uint8_t b);
uint8_t t3o_gf_sub(uint8_t a, uint8_t b);
uint8_t t3o_gf_mul(uint8_t a, uint8_t b);
int     t3o_rs_generator(int k, uint8_t* g_out);
int     t3o_rs_parity_matrix(int k, int mode, uint8_t* P_out);
int     t3o_rs_encode_blocks(int k, int mode, const uint8_t* data, uint64_t n_blocks, uint8_t* code26);
int     t3o_rs_decode_blocks(int k, int mode, uint8_t* code26, uint64_t n_blocks, uint8_t* data_k, uint8_t* ok);

/* packer */
int t3o_pack_pixels(const void* px6, uint64_t n_px, void* words9);
int t3o_unpack_words(const void* words9, uint64_t n_words, void* px6);

/* stream stages */
uint64_t t3o_regroup(const uint8_t* words9, uint64_t n_words, uint8_t* syms_out);
void     t3o_interleave2d(uint8_t* syms, uint64_t n, uint16_t w, uint16_t h, int inverse);
void     t3o_scramble(uint8_t* syms, uint64_t n, uint32_t a, uint32_t b, uint32_t s0, int inverse);
uint8_t  t3o_beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health);
void     t3o_crc12(const uint8_t* trits, uint64_t n, uint8_t* out12);
void     t3o_header_pack(const t3_cfg* cfg, uint32_t frame_seq, uint32_t band_map_hash, uint8_t* syms27);
int      t3o_header_check(const uint8_t* syms27);
void     t3o_header_unpack(const uint8_t* syms27, t3_cfg* out, uint32_t* frame_seq, uint32_t* band_map_hash,
                           uint16_t* magic, uint8_t* version);

/* frame level */
uint64_t t3o_encoded_words(uint64_t n_raw, const t3_cfg* cfg);
int t3o_encode_profile(const void* raw9, uint64_t n_raw, const t3_cfg* cfg, void* out9, uint64_t cap, uint64_t* n_out);
int t3o_decode_profile(const void* in9, uint64_t n_in, t3_cfg* seen, void* out9, uint64_t cap, uint64_t* n_out);
int t3o_encode_frame(const void* px6, uint64_t n_px, const t3_cfg* cfg, void* out9, uint64_t cap, uint64_t* n_out);
int t3o_decode_frame(const void* in9, uint64_t n_in, t3_cfg* seen, void* px6, uint64_t cap_px, uint64_t* n_px);

/* subword + wire helpers */
uint64_t t3o_extract_subword_stream(const void* words9, uint64_t n_words, int N, uint8_t* trits_out);
uint64_t t3o_build_words_from_subword_stream(const uint8_t* trits, uint64_t n, int N, uint8_t fill, void* words9);
uint64_t t3o_ut_to_base243(const uint8_t* trits, uint64_t n, uint8_t* out);
int64_t  t3o_base243_to_ut(const uint8_t* bytes, uint64_t n, uint8_t* trits_out);
/* centring blits, old/include/io_image.hpp:125-140, 215-235 (parity unpinned, see the header note on io_image.hpp) */
void t3o_blit_center_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int cw, int ch);
void t3o_extract_center_q(const void* full_px6, int fw, int fh, void* sub_px6, int sw, int sh);
void     t3o_words_to_bytes(const void* words9, uint64_t n_words, uint8_t* out);
uint64_t t3o_bytes_to_words(const uint8_t* bytes, uint64_t n, void* words9);

/* row f1: RGB8 <-> quantised YCbCr bridge, old/include/io_image.hpp:47-90,156-195 */
void     t3o_rgb_to_quant(const uint8_t* rgb, uint64_t n_px, void* px6);    /* rgb_to_quant_stream */
void     t3o_quant_to_rgb(const void* px6, uint64_t n_px, uint8_t* rgb);    /* quant_stream_to_rgb */

/* checkers' utilities */
uint64_t t3o_fnv1a64(const void* data, uint64_t n);
uint32_t t3o_crc32(const void* data, uint64_t n);
uint32_t t3o_sym_sum(const void* data, uint64_t n);
void     t3o_lcg_pixels(void* px6, uint64_t n_px, uint32_t seed);            /* SURVEY §8d generator */
void     t3o_lcg_rgb(uint8_t* rgb, uint64_t n_px, uint32_t seed);
/* the error pattern t3hip_inject_errors_dev applies, restated on the host */
void     t3o_inject_errors(void* words9, uint64_t first_sym, uint64_t n_blocks, uint32_t seed, int max_err);

#ifdef __cplusplus
}
#endif
#endif
