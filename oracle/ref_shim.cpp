// oracle/ref_shim.cpp — TEST INFRASTRUCTURE, not product code.
//
// Thin extern "C" doorway onto the UNMODIFIED reference, compiled from the sources where
// they lie under /root/reference (never copied): `make -C oracle ref` builds
// oracle/_ref/libt3ref.so with -I/root/reference/old/include -I/root/reference/include.
// Used to (1) generate tests/golden/*.json (tests/golden/make_golden.py), (2) validate the C
// restatement oracle/t3_oracle.c, (3) optionally serve as bench.py's cpu_baseline
// ("kind": "reference").  Nothing under ternary-image-codec_amd/ links or loads this.
#include <cstring>
#include <cstdint>
#include <vector>

#include "ternary_image_codec_v6_min.hpp"   // resolves to /root/reference/old/include (OLD)
#include "ternary_packing.hpp"              // /root/reference/include/ternary_packing.hpp (tpack::)
#include "../include/t3hip.h"               // t3_cfg POD only

static_assert(sizeof(Word27) == 9, "Word27 ABI");
static_assert(sizeof(PixelYCbCrQuant) == 6, "PixelYCbCrQuant ABI");

namespace {
SubwordMode sub_of(uint8_t v) {
    switch (v) { case 24: return SubwordMode::S24; case 21: return SubwordMode::S21;
                 case 18: return SubwordMode::S18; case 15: return SubwordMode::S15; default: return SubwordMode::S27; }
}
void cfg_to_enc(const t3_cfg& c, EncoderConfig& e) {
    e.profile = (ProfileID)c.profile;
    for (int i = 0; i < 9; ++i) e.uep.band_profile[i] = c.band_profile[i];
    e.tile.w = c.tile_w; e.tile.h = c.tile_h;
    e.seed.a = c.seed_a; e.seed.b = c.seed_b; e.seed.s0 = c.seed_s0;
    e.beacon.words_period = c.beacon_words_period; e.beacon.band_slot = c.beacon_band_slot;
    e.beacon.enabled = c.beacon_enabled != 0;
    e.superframe_words = c.superframe_words;
    e.subword = sub_of(c.subword); e.centered = c.centered != 0; e.coset = (CosetID)c.coset;
}
void cfg_to_seen(const t3_cfg& c, DecoderConfigSeen& d) {
    d.profile = (ProfileID)c.profile;
    for (int i = 0; i < 9; ++i) d.uep.band_profile[i] = c.band_profile[i];
    d.tile.w = c.tile_w; d.tile.h = c.tile_h;
    d.seed.a = c.seed_a; d.seed.b = c.seed_b; d.seed.s0 = c.seed_s0;
    d.beacon.words_period = c.beacon_words_period; d.beacon.band_slot = c.beacon_band_slot;
    d.beacon.enabled = c.beacon_enabled != 0;
    d.subword = sub_of(c.subword); d.centered = c.centered != 0; d.coset = (CosetID)c.coset;
}
void seen_to_cfg(const DecoderConfigSeen& d, t3_cfg& c) {
    c.profile = (uint8_t)d.profile;
    for (int i = 0; i < 9; ++i) c.band_profile[i] = d.uep.band_profile[i];
    c.tile_w = d.tile.w; c.tile_h = d.tile.h;
    c.seed_a = d.seed.a; c.seed_b = d.seed.b; c.seed_s0 = d.seed.s0;
    c.beacon_words_period = d.beacon.words_period; c.beacon_band_slot = d.beacon.band_slot;
    c.beacon_enabled = d.beacon.enabled ? 1 : 0;
    c.subword = (uint8_t)d.subword; c.centered = d.centered ? 1 : 0; c.coset = (uint8_t)d.coset;
}
RSCodec* codec_for(EncoderContext& e, int k) {
    switch (k) { case 24: return &e.rs_p1; case 22: return &e.rs_p2; case 20: return &e.rs_p3; case 18: return &e.rs_p4; }
    return nullptr;
}
EncoderContext& shared_ctx() { static EncoderContext e; return e; }
}  // namespace

extern "C" {

int ref_sizeof_word27(void) { return (int)sizeof(Word27); }
int ref_sizeof_pixel(void) { return (int)sizeof(PixelYCbCrQuant); }

void ref_gf_tables(uint8_t* exp78, int16_t* log27, uint8_t* mul729, uint8_t* inv27, uint8_t* prim) {
    GF27Context g; g.init();
    std::memcpy(exp78, g.tab.exp.data(), 78); std::memcpy(log27, g.tab.log.data(), 27 * sizeof(int16_t));
    std::memcpy(mul729, g.tab.mul.data(), 729); std::memcpy(inv27, g.tab.inv.data(), 27);
    *prim = g.tab.primitive;
}
uint8_t ref_gf_add(uint8_t a, uint8_t b) { return gf27_add(a, b); }
uint8_t ref_gf_sub(uint8_t a, uint8_t b) { return gf27_sub(a, b); }
uint8_t ref_gf_mul(uint8_t a, uint8_t b) { return gf27_mul_poly(a, b); }

int ref_rs_generator(int k, uint8_t* g_out) {
    RSCodec* r = codec_for(shared_ctx(), k); if (!r) return -1;
    for (size_t i = 0; i < r->g.size(); ++i) g_out[i] = r->g[i];
    return (int)r->g.size();
}
int ref_rs_encode_blocks(int k, const uint8_t* data, uint64_t n_blocks, uint8_t* code26) {
    RSCodec* r = codec_for(shared_ctx(), k); if (!r) return -1;
    for (uint64_t b = 0; b < n_blocks; ++b) r->encode_block(data + b * k, code26 + b * 26);
    return 0;
}
// decode_block: inout corrected in place, out_k written only on success, ok[b] = return value.
int ref_rs_decode_blocks(int k, uint8_t* code26, uint64_t n_blocks, uint8_t* data_k, uint8_t* ok) {
    RSCodec* r = codec_for(shared_ctx(), k); if (!r) return -1;
    for (uint64_t b = 0; b < n_blocks; ++b) ok[b] = r->decode_block(code26 + b * 26, data_k + b * k) ? 1 : 0;
    return 0;
}

int ref_pack_pixels(const void* px6, uint64_t n_px, void* words9) {
    std::vector<PixelYCbCrQuant> px(n_px); if (n_px) std::memcpy(px.data(), px6, n_px * 6);
    std::vector<Word27> out; bool ok = encode_raw_pixels_to_words(px, out);
    if (!out.empty()) std::memcpy(words9, out.data(), out.size() * 9);
    return ok ? (int)0 : -1;
}
int ref_unpack_words(const void* words9, uint64_t n_words, void* px6) {
    std::vector<Word27> in(n_words); if (n_words) std::memcpy(in.data(), words9, n_words * 9);
    std::vector<PixelYCbCrQuant> out; bool ok = decode_raw_words_to_pixels(in, out);
    if (!out.empty()) std::memcpy(px6, out.data(), out.size() * 6);
    return ok ? 0 : -1;
}

// encode_profile_from_raw; returns 0 and *n_out words (copied if cap allows), -4 if cap too small.
int ref_encode_profile(const void* raw9, uint64_t n_raw, const t3_cfg* cfg, void* out9, uint64_t cap, uint64_t* n_out) {
    EncoderContext e; cfg_to_enc(*cfg, e.cfg);
    std::vector<Word27> in(n_raw); if (n_raw) std::memcpy(in.data(), raw9, n_raw * 9);
    std::vector<Word27> out; bool ok = encode_profile_from_raw(in, out, e);
    *n_out = out.size();
    if (!ok) return -1;
    if (out.size() > cap) return -4;
    if (!out.empty()) std::memcpy(out9, out.data(), out.size() * 9);
    return 0;
}
// decode_profile_to_raw; `seen` in/out = DecoderContext::cfg_last_seen. Returns 0 (true) / -1 (false).
int ref_decode_profile(const void* in9, uint64_t n_in, t3_cfg* seen, void* out9, uint64_t cap, uint64_t* n_out) {
    DecoderContext d; cfg_to_seen(*seen, d.cfg_last_seen);
    std::vector<Word27> in(n_in); if (n_in) std::memcpy(in.data(), in9, n_in * 9);
    std::vector<Word27> out; bool ok = decode_profile_to_raw(in, out, d);
    seen_to_cfg(d.cfg_last_seen, *seen);
    *n_out = out.size();
    if (out.size() > cap) return -4;
    if (!out.empty()) std::memcpy(out9, out.data(), out.size() * 9);
    return ok ? 0 : -1;
}

void ref_header_pack(const t3_cfg* cfg, uint32_t frame_seq, uint32_t band_map_hash, uint8_t* syms27) {
    SuperframeHeader h{}; EncoderConfig e; cfg_to_enc(*cfg, e);
    h.profile = e.profile; h.uep = e.uep; h.tile = e.tile; h.seed = e.seed; h.beacon = e.beacon;
    h.subword = e.subword; h.centered = e.centered; h.coset = e.coset;
    h.frame_seq = frame_seq; h.band_map_hash = band_map_hash;
    HeaderPack p = HeaderCodec::pack(h); std::memcpy(syms27, p.symbols.data(), 27);
}
int ref_header_check(const uint8_t* syms27) { HeaderPack p{}; std::memcpy(p.symbols.data(), syms27, 27); return HeaderCodec::check(p) ? 1 : 0; }
void ref_header_unpack(const uint8_t* syms27, t3_cfg* out, uint32_t* frame_seq, uint32_t* band_map_hash, uint16_t* magic, uint8_t* version) {
    HeaderPack p{}; std::memcpy(p.symbols.data(), syms27, 27);
    SuperframeHeader h = HeaderCodec::unpack(p);
    DecoderConfigSeen d; d.profile = h.profile; d.uep = h.uep; d.tile = h.tile; d.seed = h.seed; d.beacon = h.beacon;
    d.subword = h.subword; d.centered = h.centered; d.coset = h.coset;
    seen_to_cfg(d, *out); *frame_seq = h.frame_seq; *band_map_hash = h.band_map_hash; *magic = h.magic; *version = h.version;
}
void ref_crc12(const uint8_t* trits, uint64_t n, uint8_t* out12) {
    std::vector<UTrit> m(trits, trits + n); std::array<UTrit, 12> r{}; CRC3::rem12(m, r); std::memcpy(out12, r.data(), 12);
}

void ref_interleave2d(uint8_t* syms, uint64_t n, uint16_t w, uint16_t h, int inverse) {
    std::vector<GF27> v(syms, syms + n); Tile2D t; t.w = w; t.h = h;
    if (inverse) deinterleave2D_boustrophedon(v, t); else interleave2D_boustrophedon(v, t);
    std::memcpy(syms, v.data(), n);
}
void ref_scramble(uint8_t* syms, uint64_t n, uint32_t a, uint32_t b, uint32_t s0, int inverse) {
    ScramblerSeed sd{a, b, s0}; uint32_t st = s0 % 3;
    for (uint64_t i = 0; i < n; ++i) syms[i] = inverse ? descramble_symbol(syms[i], sd, st) : scramble_symbol(syms[i], sd, st);
}
uint8_t ref_beacon_symbol(uint8_t profile, uint16_t frame_seq_mod, uint8_t health) {
    BeaconPayload b{(ProfileID)profile, frame_seq_mod, health}; return encode_beacon_symbol(b);
}

// subword helpers OLD:834-859
uint64_t ref_extract_subword_stream(const void* words9, uint64_t n_words, int N, uint8_t* trits_out) {
    std::vector<Word27> in(n_words); if (n_words) std::memcpy(in.data(), words9, n_words * 9);
    std::vector<UTrit> out; extract_subword_stream_from_words(in, N, out);
    if (!out.empty()) std::memcpy(trits_out, out.data(), out.size());
    return out.size();
}
uint64_t ref_build_words_from_subword_stream(const uint8_t* trits, uint64_t n, int N, uint8_t fill, void* words9) {
    std::vector<UTrit> in(trits, trits + n); std::vector<Word27> out; build_words_from_subword_stream(in, N, out, fill);
    if (!out.empty()) std::memcpy(words9, out.data(), out.size() * 9);
    return out.size();
}

// tpack:: (include/ternary_packing.hpp:18-65)
uint64_t ref_ut_to_base243(const uint8_t* trits, uint64_t n, uint8_t* out) {
    std::vector<UTrit> in(trits, trits + n); std::vector<uint8_t> o; tpack::ut_to_base243(in, o);
    std::memcpy(out, o.data(), o.size()); return o.size();
}
int64_t ref_base243_to_ut(const uint8_t* bytes, uint64_t n, uint8_t* trits_out) {
    std::vector<uint8_t> in(bytes, bytes + n); std::vector<UTrit> o; bool ok = tpack::base243_to_ut(in, o);
    if (!o.empty()) std::memcpy(trits_out, o.data(), o.size());
    return ok ? (int64_t)o.size() : -1;
}
void ref_words_to_bytes(const void* words9, uint64_t n_words, uint8_t* out) {
    std::vector<Word27> in(n_words); if (n_words) std::memcpy(in.data(), words9, n_words * 9);
    std::vector<uint8_t> o; tpack::words_to_bytes(in, o); if (!o.empty()) std::memcpy(out, o.data(), o.size());
}
uint64_t ref_bytes_to_words(const uint8_t* bytes, uint64_t n, void* words9) {
    std::vector<uint8_t> in(bytes, bytes + n); std::vector<Word27> o; tpack::bytes_to_words(in, o);
    if (!o.empty()) std::memcpy(words9, o.data(), o.size() * 9); return o.size();
}

int ref_selftest_rs_unit(void) { return selftest_rs_unit() ? 1 : 0; }
int ref_selftest_api_roundtrip(void) { return selftest_api_roundtrip() ? 1 : 0; }

// The reference encoder timed as a CPU baseline: pack + encode_profile_from_raw (old/src/main.cpp:15-19).
int ref_encode_frame(const void* px6, uint64_t n_px, const t3_cfg* cfg, void* out9, uint64_t cap, uint64_t* n_out) {
    std::vector<PixelYCbCrQuant> px(n_px); if (n_px) std::memcpy(px.data(), px6, n_px * 6);
    std::vector<Word27> raw; encode_raw_pixels_to_words(px, raw);
    EncoderContext e; cfg_to_enc(*cfg, e.cfg);
    std::vector<Word27> out; bool ok = encode_profile_from_raw(raw, out, e);
    *n_out = out.size(); if (!ok) return -1; if (out.size() > cap) return -4;
    if (!out.empty() && out9) std::memcpy(out9, out.data(), out.size() * 9);
    return 0;
}

}  // extern "C"
