/*
 * t3hip.h — C-ABI of libt3hip.so, the MI355X (gfx950) implementation of the
 * balanced-ternary Word27 encode/decode hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer:
 * its path is header-only C++ over std::vector.  Each entry point below names
 * the reference function (file:line, OLD = old/include/ternary_image_codec_v6_min.hpp)
 * whose work it performs; include/ternary_codec_v6.hpp wraps these back into the
 * reference's own C++ names and signatures.
 *
 * Conventions
 *   - plain pointers + sizes, no C++/torch types; all structs are POD.
 *   - "words" are the reference's Word27 ABI: 9 bytes, one GF(27) symbol (0..26) each
 *     (OLD:666-669).  "pixels" are PixelYCbCrQuant: {u16 Yq; i16 Cbq; i16 Crq} = 6 bytes
 *     (OLD:670-674).
 *   - functions return T3_OK (0) or a negative T3_E_* code.  The reference's `bool`
 *     results map to: true = T3_OK, false = T3_E_HEADER / T3_E_RS / T3_E_ARG.
 *   - *_dev entry points take DEVICE pointers and a hipStream_t (as void*), never
 *     allocate caller-visible memory and never synchronise unless documented.  Device buffers of the profile
 *     encode / decode entry points must be 16-byte aligned (T3_E_ARG otherwise; RAW mode and the pack / unpack,
 *     subword, RGB-bridge and CRC entry points take any alignment): consecutive frames packed into one buffer
 *     (9 * n_words bytes each) start on a 16-byte boundary only if the caller pads them.
 *   - there is no CPU fallback: compute entry points fail with T3_E_NODEVICE when
 *     no gfx950 device is usable.  Pure-metadata calls (plan, tables, header) are host-only.
 */
#ifndef T3HIP_H
#define T3HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------- */
#define T3_OK            0
#define T3_E_NODEVICE   -1  /* no usable HIP device / t3hip_init not successful   */
#define T3_E_HIP        -2  /* a HIP runtime call failed (see t3hip_last_hip_error) */
#define T3_E_ARG        -3  /* bad argument (null pointer, invalid subword, k ...) */
#define T3_E_CAPACITY   -4  /* caller's output buffer is too small                */
#define T3_E_HEADER     -5  /* header RS / CRC-12 failure  (OLD:920,929-934 -> false) */
#define T3_E_RS         -6  /* uncorrectable RS block      (OLD:987 -> false)     */
#define T3_E_COMM       -7  /* RCCL missing or a collective call failed (see t3hip_comm_last_error) */

/* ---- profile ids (OLD:34) -------------------------------------------------- */
#define T3_P1_RS26_24     0
#define T3_P2_RS26_22     1
#define T3_P3_RS26_20     2
#define T3_P4_RS26_18     3
#define T3_P5_RS26_22_2D  4
#define T3_RAW_MODE       0xFF

/* Arithmetic/framing flavour.
 * COMPAT: byte-exact with the reference (its non-RS "parity" map OLD:517-535, Forney
 *         with `add` OLD:658, its encoder/decoder framings as they are).
 * FIXED : true systematic RS consistent with the decoder's root convention, Forney
 *         with `sub`, self-consistent v6c framing so that
 *         decode(encode(x)) == x and <= t symbol errors per block are corrected. */
#define T3_MODE_COMPAT 0
#define T3_MODE_FIXED  1

/* POD mirror of EncoderConfig (OLD:862-873) / DecoderConfigSeen (OLD:874-884). */
typedef struct t3_cfg {
    uint8_t  profile;            /* ProfileID value; T3_RAW_MODE = 0xFF          */
    uint8_t  band_profile[9];    /* UEPLayout::band_profile (OLD:60-63), used %4  */
    uint16_t tile_w, tile_h;     /* Tile2D (OLD:73-76)                            */
    uint32_t seed_a, seed_b, seed_s0;      /* ScramblerSeed (OLD:77-80)           */
    uint32_t beacon_words_period;          /* SparseBeaconCfg (OLD:95-100)        */
    uint8_t  beacon_band_slot;
    uint8_t  beacon_enabled;
    uint8_t  subword;            /* SubwordMode value 27/24/21/18/15 (OLD:117)    */
    uint8_t  centered;
    uint32_t superframe_words;   /* EncoderConfig::superframe_words (OLD:869)     */
    uint8_t  coset;              /* CosetID (OLD:114)                             */
    uint8_t  mode;               /* T3_MODE_COMPAT / T3_MODE_FIXED (build-side)   */
    uint8_t  reserved[2];
} t3_cfg;

/* Closed-form stream layout of one encode_profile_from_raw call (OLD:1043-1169). */
typedef struct t3_layout {
    uint64_t n_raw_words;        /* input Word27 count                            */
    uint64_t n_sym;              /* regrouped symbols = ceil(26*W/3)  (OLD:1051-1082) */
    uint64_t band_len[9];        /* symbols per band (i%9 split, OLD:1087-1088)   */
    uint64_t band_blocks[9];     /* RS blocks per band (tail dropped in COMPAT, OLD:1107) */
    uint64_t band_body_off[9];   /* first body symbol of each band (band-serial)  */
    uint64_t body_syms;          /* 26 * sum(band_blocks)                         */
    uint64_t body_syms_framed;   /* after beacon insertion (OLD:1118-1141)        */
    uint64_t out_syms;           /* header + framed body                          */
    uint64_t out_words;          /* ceil(out_syms/9) (OLD:1164)                   */
    uint32_t header_syms;        /* 52 in COMPAT (OLD:1159-1162), 90 in FIXED     */
    uint8_t  band_k[9];          /* RS k of each band (OLD:1089-1100)             */
    uint8_t  interleave2d;       /* 1 if P5 && tile.w && tile.h (OLD:1083)        */
    uint8_t  beacon_on;          /* 1 if beacon.enabled && period>0 (OLD:1118)    */
    uint8_t  pad_;
} t3_layout;

/* Fixed-size per-frame index record; the multi-GPU exchange step all-gathers these
 * (T3V frame index io_t3p_t3v.cpp:252-289). 96 bytes. */
typedef struct t3_frame_record {
    uint64_t frame_idx;
    uint64_t n_words;            /* coded Word27 count of the frame               */
    uint64_t byte_offset;        /* filled by t3hip_index_assemble (prefix sum)   */
    uint32_t crc32;              /* CRC-32 (poly 0xEDB88320) of the 9*n_words payload bytes */
    uint32_t sym_sum;            /* sum of all payload symbol bytes mod 2^32      */
    uint8_t  header_syms[54];    /* first 6 coded words (superframe header)       */
    uint8_t  profile;
    uint8_t  mode;
    uint8_t  pad_[8];
} t3_frame_record;

/* ---- lifecycle ---------------------------------------------------------------
 * A context = one GPU's stream, tables and scratch.  t3hip_init creates the process DEFAULT context on `device` (idempotent for
 * the same device; T3_E_ARG for another one while it exists: t3hip_shutdown first, or use t3hip_create).  A host that drives
 * several GPUs from one process -- the reference's language has no process-per-GPU launcher -- creates one context per device
 * (t3hip_create) and binds each of its worker threads to one with t3hip_use (thread-local; also makes that device current for
 * HIP).  Every entry point of this header works on the calling thread's context: the one given to t3hip_use, else the default.
 * Contexts are independent: two threads on two contexts never share a stream, a scratch buffer or a lock. */
typedef struct t3hip_ctx t3hip_ctx;
int         t3hip_device_count(void);
int         t3hip_init(int device);              /* the process default context           */
int         t3hip_shutdown(void);                /* destroys the default context          */
int         t3hip_is_ready(void);                /* 1 when the calling thread has a usable context */
int         t3hip_create(int device, t3hip_ctx** out);
int         t3hip_destroy(t3hip_ctx* ctx);       /* not the default context (that is t3hip_shutdown's) */
int         t3hip_use(t3hip_ctx* ctx);           /* bind the calling thread; NULL = back to the default */
t3hip_ctx*  t3hip_current(void);                 /* the calling thread's context (NULL if none)   */
int         t3hip_ctx_device(const t3hip_ctx* ctx);   /* its device index (-1 if none); NULL = current */
const char* t3hip_strerror(int code);
const char* t3hip_last_hip_error(void);
const char* t3hip_version(void);

/* ---- host-only metadata (no device needed) ------------------------------------ */
void t3hip_cfg_default(t3_cfg* cfg);                         /* EncoderContext() defaults OLD:862-873,898 */
int  t3hip_plan(uint64_t n_raw_words, const t3_cfg* cfg, t3_layout* out);
uint64_t t3hip_encoded_words(uint64_t n_raw_words, const t3_cfg* cfg);   /* 0 on bad cfg */
/* GF27Context::init tables (OLD:436-466): exp[78], log[27], mul[729], inv[27]. */
int  t3hip_gf27_tables(uint8_t* exp78, int16_t* log27, uint8_t* mul729, uint8_t* inv27);
/* RSCodec::build_gen (OLD:501-516): g[0..r], r = 26-k. */
int  t3hip_rs_generator(int k, uint8_t* g_out);
/* parity = data * P, P is k x (26-k) row-major (row i = parity of unit vector e_i). */
int  t3hip_rs_parity_matrix(int k, int mode, uint8_t* P_out);
/* Host only: the tables of the matrix-core encoder for one k, for inspection and CPU tests.
 * afrag[768]: A operand of v_mfma_i32_32x32x32_i8, 3 K-steps x 64 lanes x 4 dwords (trit coefficients as signed bytes);
 * lds_img[3168]: three 4-KiB scrambler-state tables (27 symbols x 32 bank copies: trits | scrambled symbol << 24), then
 * the three 128-byte mod-3 fold tables M_t[x] = 3^t ((x - 64) mod 3). */
int  t3hip_mfma_encode_tables(int k, int mode, uint32_t* afrag768, uint32_t* lds_img3168);
/* HeaderCodec::pack / check / unpack (OLD:206-380). */
int  t3hip_header_pack(const t3_cfg* cfg, uint32_t frame_seq, uint32_t band_map_hash, uint8_t syms27[27]);
int  t3hip_header_check(const uint8_t syms27[27]);           /* 1 ok, 0 bad */
int  t3hip_header_unpack(const uint8_t syms27[27], t3_cfg* out, uint32_t* frame_seq, uint32_t* band_map_hash);
/* The coded header symbols the encoder emits in front of the body (52 / 90). */
int  t3hip_header_encode(const t3_cfg* cfg, uint64_t n_raw_words, uint8_t* syms_out, uint32_t* n_syms);

/* ---- host-buffer entry points (what the std::vector API binds) ----------------- */
/* encode_raw_pixels_to_words OLD:723-734; words9 must hold (n_px+1)/2 words. */
int t3hip_pack_pixels(const void* px6, uint64_t n_px, void* words9);
/* decode_raw_words_to_pixels OLD:735-747; px6 must hold 2*n_words pixels. */
int t3hip_unpack_words(const void* words9, uint64_t n_words, void* px6);
/* encode_profile_from_raw OLD:1043-1169. */
int t3hip_encode_profile(const void* raw9, uint64_t n_raw, const t3_cfg* cfg,
                         void* out9, uint64_t cap_words, uint64_t* n_out);
/* decode_profile_to_raw OLD:995-1041.  `seen` is DecoderContext::cfg_last_seen: read for the
 * RAW shortcut, overwritten from the header once it decodes (OLD:1006-1013) — also when a
 * later RS block fails, as in the reference. */
int t3hip_decode_profile(const void* in9, uint64_t n_in, t3_cfg* seen,
                         void* out9, uint64_t cap_words, uint64_t* n_out);
/* Build-side conveniences named by the north star: pixels -> coded words in one fused
 * launch (= OLD:723 + OLD:1043), coded words -> pixels (= OLD:995 + OLD:735). */
int t3hip_encode_frame(const void* px6, uint64_t n_px, const t3_cfg* cfg,
                       void* out9, uint64_t cap_words, uint64_t* n_out);
int t3hip_decode_frame(const void* in9, uint64_t n_in, t3_cfg* seen,
                       void* px6, uint64_t cap_px, uint64_t* n_px);

/* ---- device-resident entry points (async on `stream` unless noted) ---------------- */
int t3hip_pack_pixels_dev(const void* d_px6, uint64_t n_px, void* d_words9, void* stream);
int t3hip_unpack_words_dev(const void* d_words9, uint64_t n_words, void* d_px6, void* stream);
int t3hip_encode_profile_dev(const void* d_raw9, uint64_t n_raw, const t3_cfg* cfg,
                             void* d_out9, uint64_t cap_words, uint64_t* n_out, void* stream);
int t3hip_encode_frame_dev(const void* d_px6, uint64_t n_px, const t3_cfg* cfg,
                           void* d_out9, uint64_t cap_words, uint64_t* n_out, void* stream);
/* Decode with the header parsed on the host: copies the 6 (COMPAT) / 10 (FIXED) header words
 * device->host and synchronises `stream` once, launches the body kernels, then synchronises
 * again to read the block-failure flag.  `to_pixels` selects Word27 or pixel output. */
int t3hip_decode_profile_dev(const void* d_in9, uint64_t n_in, t3_cfg* seen,
                             void* d_out, uint64_t cap_units, uint64_t* n_out,
                             int to_pixels, void* stream);
/* Fully asynchronous body decode for a caller that already knows the stream's config
 * (e.g. from t3hip_read_header_dev): no synchronisation; *d_fail (device u32) is
 * incremented for every uncorrectable block. */
int t3hip_read_header_dev(const void* d_in9, uint64_t n_in, int mode, t3_cfg* out_cfg,
                          uint64_t* n_raw_words, void* stream);            /* synchronises */
int t3hip_decode_body_dev(const void* d_in9, uint64_t n_in, const t3_cfg* cfg, uint64_t n_raw_words,
                          void* d_out, uint64_t cap_units, uint64_t* n_out, int to_pixels,
                          uint32_t* d_fail, void* stream);

/* Streaming decode, no synchronisation.  Frames of one stream repeat their configuration — the reference keeps it in
 * DecoderContext::cfg_last_seen (OLD:1006-1013) — so a caller that has parsed one frame's header (t3hip_decode_profile_dev
 * / t3hip_read_header_dev) decodes the following frames with that configuration: the body kernels are launched at once,
 * and a device-side check compares the frame's header symbols with the header `cfg` / `n_raw_words` encode to
 * (header RS + CRC-12, OLD:1142-1162).  Device words, zeroed and written by the call on `stream`:
 *   d_verdict[0] = 1 if the header differs (another configuration, or symbols that need the header's RS correction):
 *                  the output is then meaningless and the frame goes through t3hip_decode_profile_dev;
 *   d_verdict[1] = number of uncorrectable RS blocks (the reference's `false`, OLD:987).                          */
int t3hip_decode_frame_async(const void* d_in9, uint64_t n_in, const t3_cfg* cfg, uint64_t n_raw_words,
                             void* d_out, uint64_t cap_units, uint64_t* n_out, int to_pixels,
                             uint32_t* d_verdict, void* stream);

/* ---- block-level RS(26,k) (RSCodec::encode_block OLD:517-535, decode_block OLD:546-662) */
int t3hip_rs_encode_blocks_dev(int k, int mode, const uint8_t* d_data_k, uint64_t n_blocks,
                               uint8_t* d_code26, void* stream);
/* d_code26 is corrected in place (inout_n), d_data_k receives the first k symbols (out_k),
 * d_ok[b] = 1/0 is decode_block's return value.  On a 0 the reference leaves inout_n
 * partially modified and out_k untouched; so does this. */
int t3hip_rs_decode_blocks_dev(int k, int mode, uint8_t* d_code26, uint64_t n_blocks,
                               uint8_t* d_data_k, uint8_t* d_ok, void* stream);

/* Host-buffer forms (what RSCodec::encode_block / decode_block of include/ternary_codec_v6.hpp bind): upload, the kernels above,
 * download.  data_k of a block whose decode fails is returned as it came in (the reference leaves out_k untouched). */
int t3hip_rs_encode_blocks(int k, int mode, const uint8_t* data_k, uint64_t n_blocks, uint8_t* code26);
int t3hip_rs_decode_blocks(int k, int mode, uint8_t* code26_inout, uint64_t n_blocks, uint8_t* data_k, uint8_t* ok);
/* ONE block, on the host (no device, no launch): what RSCodec::encode_block / decode_block bind when a caller walks blocks one at a
 * time (the reference's self-test OLD:1172-1207, its header RS OLD:1142-1158).  26 bytes are control data, like the header codec;
 * same parity matrix / decoder as the kernels.  decode returns 1 (decode_block true), 0 (false; out_k untouched) or T3_E_ARG. */
int t3hip_rs_encode_block_host(int k, int mode, const uint8_t* data_k, uint8_t* code26);
int t3hip_rs_decode_block_host(int k, int mode, uint8_t* code26_inout, uint8_t* data_k);

/* ---- the reference's symbol-level public helpers --------------------------------------------------------
 * Single symbols and the header are control data: host arithmetic on the tables the kernels are built from.
 *   gf27_add / gf27_sub / gf27_mul_poly OLD:383-413 (operands reduced mod 27)
 *   scramble_symbol / descramble_symbol OLD:81-94: *st <- (a * *st + b) % 3 in uint32 wrap-around, then every trit of s +/- *st
 *   encode_beacon_symbol OLD:107-113;  CRC3::rem12 OLD:176-205 (ternary CRC-12 of n trits, 12 zero trits appended)
 * interleave2D_boustrophedon / deinterleave2D_boustrophedon OLD:750-813 on a symbol vector: device kernel (the map is an
 * involution inside every row segment, so `inverse` selects nothing); w == 0 or h == 0 leaves the symbols as they are. */
uint8_t t3hip_gf27_add(uint8_t a, uint8_t b);
uint8_t t3hip_gf27_sub(uint8_t a, uint8_t b);
uint8_t t3hip_gf27_mul(uint8_t a, uint8_t b);
uint8_t t3hip_scramble_symbol(uint8_t s, uint32_t a, uint32_t b, uint32_t* st, int inverse);
uint8_t t3hip_beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health_flags);
int t3hip_crc12(const uint8_t* trits, uint64_t n, uint8_t out12[12]);
int t3hip_interleave2d(uint8_t* syms, uint64_t n, uint16_t w, uint16_t h, int inverse);
int t3hip_interleave2d_dev(const uint8_t* d_in, uint64_t n, uint16_t w, uint16_t h, uint8_t* d_out, void* stream);   /* d_out != d_in */

/* ---- trit error injector for the recovery test ------------------ */
/* For every 26-symbol block of the body region [first_sym, first_sym+26*n_blocks): a counter hash of
 * (seed, block) picks e in 0..max_err distinct positions and alters one trit of each. */
int t3hip_inject_errors_dev(void* d_words9, uint64_t first_sym, uint64_t n_blocks,
                            uint32_t seed, int max_err, void* stream);

/* ---- frame index record (multi-GPU exchange payload) -------------------------------- */
/* d_scratch: device memory the call may use until the record is written (stream-ordered), 4-byte aligned, content irrelevant.
 * t3hip_frame_record_scratch_bytes() says how much it would like (one partial result per CRC workgroup: no zeroing pass, no atomics);
 * anything from 8 bytes up works (two accumulators, zeroed by a fill in front of the CRC kernel). */
int t3hip_frame_record_dev(const void* d_words9, uint64_t n_words, uint64_t frame_idx,
                           const t3_cfg* cfg, t3_frame_record* d_rec, void* d_scratch,
                           uint64_t scratch_bytes, void* stream);
uint64_t t3hip_frame_record_scratch_bytes(uint64_t n_words);
/* Payload CRC of the T3P6/T3V6 containers (crc32_acc, src/io_t3p_t3v.cpp:20-36: poly 0xEDB88320, init and final
 * inversion 0xFFFFFFFF) of a device buffer / a host buffer (uploaded, same kernel).  Synchronous: *crc_out is host
 * memory and valid on return.  n_bytes == 0 gives 0, which is also what the containers store for an empty payload. */
int t3hip_crc32_dev(const void* d_data, uint64_t n_bytes, uint32_t* crc_out, void* stream);
int t3hip_crc32(const void* data, uint64_t n_bytes, uint32_t* crc_out);
/* Host: sort gathered records by frame_idx and fill byte_offset (T3V index, io_t3p_t3v.cpp:252-289). */
int t3hip_index_assemble(t3_frame_record* recs, uint64_t n_recs, uint64_t first_payload_offset);

/* ---- multi-GPU exchange step ------------------------------------------------------
 * Frames shard round-robin over GPUs with no data-path collective; the only exchange is one all-gather of the fixed-size
 * frame records above, from which every rank assembles the T3V frame index (io_t3p_t3v.cpp:252-289: offset, words per
 * frame) with t3hip_index_assemble.  The reference has no collective at all; a C++ host does
 *   rank 0: t3hip_comm_unique_id(id) -> hands the 128 bytes to the other ranks by its own means (file, socket, MPI, ...)
 *   every rank, after t3hip_init(device): t3hip_comm_create(id, world, rank, &c)         (ncclCommInitRank)
 *   per batch: t3hip_index_allgather(c, d_local, n_local, d_all, stream)                  (ncclAllGather, asynchronous)
 * RCCL is bound at run time (dlopen librccl.so.1); without it these return T3_E_COMM. */
#define T3_COMM_ID_BYTES 128
typedef struct t3_comm t3_comm;
/* T3_OK when RCCL can be bound in this process, T3_E_COMM otherwise; no collective, no device call.  A multi-rank host probes this on
 * EVERY rank and agrees on the result (its own control channel) before anyone draws an id or calls t3hip_comm_create: a rank that
 * cannot bind must not leave the others waiting inside ncclCommInitRank. */
int t3hip_comm_available(void);
int t3hip_comm_unique_id(uint8_t id[T3_COMM_ID_BYTES]);
int t3hip_comm_create(const uint8_t id[T3_COMM_ID_BYTES], int world, int rank, t3_comm** out);
int t3hip_comm_destroy(t3_comm* c);
int t3hip_comm_world(const t3_comm* c);
int t3hip_comm_rank(const t3_comm* c);
const char* t3hip_comm_last_error(void);
/* d_local: n_local records of this rank (device); d_all: world * n_local records (device), rank-major.  Every rank passes the
 * same n_local (pad with records whose frame_idx == UINT64_MAX). */
int t3hip_index_allgather(t3_comm* c, const t3_frame_record* d_local, uint64_t n_local, t3_frame_record* d_all, void* stream);

/* ---- subword trit streams and wire packings ------------------------------
 * One byte per trit (0..2) on the trit side.  N = trits kept per word (SubwordMode: 27/24/21/18/15; any 1..27 accepted).
 *   extract : extract_subword_stream_from_words  OLD:834-844 -> n_words * N trits (first N trits of every word)
 *   build   : build_words_from_subword_stream    OLD:845-859 -> ceil(n_trits / N) words; slots N..26 = fill, a short last
 *             group is zero-padded up to N
 *   base243 : tpack::ut_to_base243 / base243_to_ut  TPACK:28-50: uint32 LE trit count, then 5 trits per byte (LSD first);
 *             unpack returns T3_E_HEADER where the reference returns false (short input, fewer trits than announced)
 *   mod27   : tpack::words_to_bytes / bytes_to_words  TPACK:53-65: every byte % 27 (callers check n % 9 for bytes_to_words) */
uint64_t t3hip_subword_words(uint64_t n_trits, int N);
uint64_t t3hip_base243_bytes(uint64_t n_trits);
int t3hip_subword_extract(const void* words9, uint64_t n_words, int N, uint8_t* trits);
int t3hip_subword_build(const uint8_t* trits, uint64_t n_trits, int N, uint8_t fill, void* words9, uint64_t cap_words, uint64_t* n_words);
int t3hip_base243_pack(const uint8_t* trits, uint64_t n_trits, uint8_t* out, uint64_t cap_bytes, uint64_t* n_bytes);
int t3hip_base243_unpack(const uint8_t* in, uint64_t n_bytes, uint8_t* trits, uint64_t cap_trits, uint64_t* n_trits);
int t3hip_mod27_bytes(const uint8_t* in, uint64_t n, uint8_t* out);
int t3hip_subword_extract_dev(const void* d_words9, uint64_t n_words, int N, uint8_t* d_trits, void* stream);
int t3hip_subword_build_dev(const uint8_t* d_trits, uint64_t n_trits, int N, uint8_t fill, void* d_words9, uint64_t cap_words, uint64_t* n_words, void* stream);
int t3hip_base243_pack_dev(const uint8_t* d_trits, uint64_t n_trits, uint8_t* d_out, uint64_t cap_bytes, uint64_t* n_bytes, void* stream);
/* total = the trit count of the 4-byte header (the host entry point reads it itself) */
int t3hip_base243_unpack_dev(const uint8_t* d_in, uint64_t n_bytes, uint64_t total, uint8_t* d_trits, void* stream);
int t3hip_mod27_bytes_dev(const uint8_t* d_in, uint64_t n, uint8_t* d_out, void* stream);
/* Centring blits of row f3, old/include/io_image.hpp:125-140 (blit_center_rgb) and :215-235 (extract_center_q); parity unpinned
 * like the rest of io_image.hpp (restated from the text, checked against the oracle's restatement).
 *   blit    : sw x sh RGB8 image into the middle of a zeroed cw x ch canvas, x0 = max(0,(cw-sw)/2), y0 = max(0,(ch-sh)/2); source
 *             rows that fall below the canvas are dropped.  sw > cw is T3_E_ARG (the reference overruns the canvas row there).
 *   extract : the sw x sh window in the middle of a fw x fh frame of 6-byte pixels; window rows below the frame come out zero.
 *             sw > fw is T3_E_ARG (the reference reads on into the next frame row).
 * Device pointers: destination 4-byte aligned. */
int t3hip_blit_center_rgb(const uint8_t* src, int sw, int sh, uint8_t* dst, int cw, int ch);
int t3hip_extract_center_q(const void* full_px6, int fw, int fh, void* sub_px6, int sw, int sh);
int t3hip_blit_center_rgb_dev(const uint8_t* d_src, int sw, int sh, uint8_t* d_dst, int cw, int ch, void* stream);
int t3hip_extract_center_q_dev(const void* d_full_px6, int fw, int fh, void* d_sub_px6, int sw, int sh, void* stream);

/* ---- SURVEY 8 row f1 (first version, own kernels): RGB8 <-> quantised YCbCr bridge ------------
 * rgb_to_quant_stream / quant_stream_to_rgb, old/include/io_image.hpp:47-90,156-195: float BT.601-style conversion with
 * std::lround and clamps, then quantisation in double.  rgb = 3 bytes per pixel (R,G,B), px6 = PixelYCbCrQuant[n_px].
 * Bit-exact against the oracle's restatement for all 2^24 RGB values; parity UNPINNED against a reference build (that
 * header does not compile: ImageU8::swap, old/include/io_image.hpp:218). */
int t3hip_rgb_to_quant(const uint8_t* rgb, uint64_t n_px, void* px6);
int t3hip_quant_to_rgb(const void* px6, uint64_t n_px, uint8_t* rgb);
int t3hip_rgb_to_quant_dev(const uint8_t* d_rgb, uint64_t n_px, void* d_px6, void* stream);
int t3hip_quant_to_rgb_dev(const void* d_px6, uint64_t n_px, uint8_t* d_rgb, void* stream);
/* RGB frame -> coded stream and back in one call each (what a caller of io_image.hpp's rgb_to_quant_stream +
 * encode_raw_pixels_to_words + encode_profile_from_raw does, old/src/main.cpp:12-19): the bridge kernel into a per-stream
 * scratch, then the fused encode / after the decode.  Both asynchronous on `stream`; the decode uses the streaming entry
 * (known configuration, d_verdict as for t3hip_decode_frame_async).  n_px pixels -> (n_px + 1) / 2 raw words. */
int t3hip_encode_rgb_dev(const uint8_t* d_rgb, uint64_t n_px, const t3_cfg* cfg, void* d_out9, uint64_t cap_words,
                         uint64_t* n_out, void* stream);
int t3hip_decode_rgb_async(const void* d_in9, uint64_t n_in, const t3_cfg* cfg, uint64_t n_px, uint8_t* d_rgb,
                           uint32_t* d_verdict, void* stream);

/* ---- measurement aid: a plain streaming kernel (16 bytes per lane, four loads in flight) that reads n_read and writes n_write
 * bytes: the part's ceiling for a codec launch's byte volumes (profiles/copy_ceiling.py).  16-byte aligned buffers. */
int t3hip_diag_stream_copy_dev(const void* d_src, uint64_t n_read, void* d_dst, uint64_t n_write, int blocks_per_cu /* < 0: non-temporal */, void* stream);

/* ---- timing helper: HIP events on the caller's stream -------------------------------- */
int t3hip_event_create(void** ev);
int t3hip_event_record(void* ev, void* stream);
int t3hip_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on stop */
int t3hip_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* T3HIP_H */
