// ternary_codec_v6.hpp — C++17 drop-in for the Word27 hot path of the reference's
// old/include/ternary_image_codec_v6_min.hpp (OLD), implemented on MI355X through libt3hip.so (include/t3hip.h).
//
// Same names, layouts and call signatures as the reference for everything its callers touch on this path
// (old/src/main.cpp:15-27, old/include/io_image.hpp:240-246, include/io_image.hpp:286-310, t3v/t3p writers):
// the types are ABI-identical (Word27 = 9 bytes, PixelYCbCrQuant = 6 bytes), the functions take and fill the same
// std::vector arguments and return the same bool.  What is different, by design:
//   * the arithmetic runs in HIP kernels; there is no CPU implementation behind these functions.  If libt3hip.so has no
//     usable gfx950 device every function returns false and t3::last_status() tells why (T3_E_NODEVICE).
//   * EncoderConfig / DecoderConfigSeen carry one extra field, `mode` (T3_MODE_COMPAT = byte-exact with the reference,
//     bugs included; T3_MODE_FIXED = the self-consistent v6c variant that actually round-trips).
//   * encode_frame / decode_frame fuse pack+encode and decode+unpack into single launches (north-star conveniences).
// Link with -lt3hip (ternary-image-codec_amd/libt3hip.so).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "t3hip.h"

// ---- symbols and words (OLD:21-31, 666-674) -------------------------------------------------------------
using UTrit = uint8_t;   // 0..2
using GF27 = uint8_t;    // 0..26
static constexpr int TRITS_PER_WORD = 27, SYM_PER_WORD = 9, NUM_BANDS = 9;
inline GF27 pack3(UTrit a, UTrit b, UTrit c) { return (GF27)(a + 3 * b + 9 * c); }
inline std::array<UTrit, 3> unpack3(GF27 s) { return {(UTrit)(s % 3), (UTrit)(s / 3 % 3), (UTrit)(s / 9 % 3)}; }

struct Word27 { std::array<GF27, SYM_PER_WORD> sym{}; };
struct PixelYCbCrQuant { uint16_t Yq = 0; int16_t Cbq = 0, Crq = 0; };
static_assert(sizeof(Word27) == 9 && sizeof(PixelYCbCrQuant) == 6, "reference ABI");

// ---- profiles, UEP, tiles, scrambler, beacon (OLD:34-114) -------------------------------------------------
enum class ProfileID : uint8_t { RAW_MODE = 0xFF, P1_RS26_24 = 0, P2_RS26_22 = 1, P3_RS26_20 = 2, P4_RS26_18 = 3, P5_RS26_22_2D = 4 };
struct RSParams { uint8_t n = 26, k = 22; };
inline RSParams rs_params_for(ProfileID p) {
    static const uint8_t ks[5] = {24, 22, 20, 18, 22};
    const unsigned i = (unsigned)p;
    return RSParams{26, i < 5 ? ks[i] : (uint8_t)22};
}
struct UEPLayout { std::array<uint8_t, NUM_BANDS> band_profile{}; };
inline void uep_uniform(UEPLayout& u, uint8_t idx = 1) { u.band_profile.fill(idx % 4); }
inline void uep_luma_priority(UEPLayout& u) { u.band_profile.fill(1); u.band_profile[0] = u.band_profile[3] = u.band_profile[6] = 2; }
struct Tile2D { uint16_t w = 0, h = 0; };
struct ScramblerSeed { uint32_t a = 1, b = 1, s0 = 1; };
// one step of the stream scrambler on one symbol (OLD:81-94): the state is stepped first, in uint32 wrap-around arithmetic
inline GF27 scramble_symbol(GF27 s, const ScramblerSeed& seed, uint32_t& st) { return t3hip_scramble_symbol(s, seed.a, seed.b, &st, 0); }
inline GF27 descramble_symbol(GF27 s, const ScramblerSeed& seed, uint32_t& st) { return t3hip_scramble_symbol(s, seed.a, seed.b, &st, 1); }
struct SparseBeaconCfg { uint32_t words_period = 0; uint8_t band_slot = 0; bool enabled = false; };
struct BeaconPayload { ProfileID profile; uint16_t frame_seq_mod; uint8_t health_flags; };
inline GF27 encode_beacon_symbol(const BeaconPayload& b) { return t3hip_beacon_symbol((uint8_t)b.profile, b.frame_seq_mod, b.health_flags); }   // OLD:107-113
enum class CosetID : uint8_t { C0 = 0, C1 = 1, C2 = 2 };

// ---- subword modes and centring (OLD:117-152) ---------------------------------------------------------------
enum class SubwordMode : uint8_t { S27 = 27, S24 = 24, S21 = 21, S18 = 18, S15 = 15 };
inline int payload_len_for(SubwordMode m) { return (int)m; }
struct StdRes { uint16_t w, h; };
inline StdRes std_res_for(SubwordMode m) {
    switch (m) {
        case SubwordMode::S24: return {3840, 2160};
        case SubwordMode::S21: return {1920, 1080};
        case SubwordMode::S18: return {1280, 720};
        case SubwordMode::S15: return {854, 480};
        default: return {7680, 4320};
    }
}
struct ActiveWindow { uint32_t x0, y0, w, h; };
inline ActiveWindow centered_window(SubwordMode m) {
    const StdRes full = std_res_for(SubwordMode::S27), t = std_res_for(m);
    return {(uint32_t)((full.w - t.w) / 2), (uint32_t)((full.h - t.h) / 2), t.w, t.h};
}
inline bool is_valid_subword(SubwordMode m) { const int v = (int)m; return v == 27 || v == 24 || v == 21 || v == 18 || v == 15; }

// ---- contexts (OLD:862-916) -------------------------------------------------------------------------------------
struct EncoderConfig {
    ProfileID profile = ProfileID::P2_RS26_22;
    UEPLayout uep{};
    Tile2D tile{};
    ScramblerSeed seed{1, 1, 1};
    SparseBeaconCfg beacon{};
    uint32_t superframe_words = 8192;
    SubwordMode subword = SubwordMode::S27;
    bool centered = true;
    CosetID coset = CosetID::C0;
    uint8_t mode = T3_MODE_COMPAT;      // build-side: T3_MODE_COMPAT | T3_MODE_FIXED
};
struct DecoderConfigSeen {
    ProfileID profile = ProfileID::P2_RS26_22;
    UEPLayout uep{};
    Tile2D tile{};
    ScramblerSeed seed{1, 1, 1};
    SparseBeaconCfg beacon{};
    SubwordMode subword = SubwordMode::S27;
    bool centered = true;
    CosetID coset = CosetID::C0;
    uint8_t mode = T3_MODE_COMPAT;
};
// (EncoderContext / DecoderContext: below, after GF27Context and RSCodec, which they own as in OLD:885-916)

// ---- header (OLD:155-380) ----------------------------------------------------------------------------------------
struct SuperframeHeader {
    uint16_t magic = 0x0A2; uint8_t version = 1;
    ProfileID profile = ProfileID::P2_RS26_22; UEPLayout uep{}; Tile2D tile{}; ScramblerSeed seed{};
    uint32_t band_map_hash = 0, frame_seq = 0, reserved = 0, crc3m = 0;
    SparseBeaconCfg beacon{}; SubwordMode subword = SubwordMode::S27; bool centered = true; CosetID coset = CosetID::C0;
};
struct HeaderPack { std::array<GF27, 27> symbols{}; };
struct CRC3 {   // OLD:176-205: ternary CRC-12, g = x^12 + x^7 + x^4 + x^3 + 1, twelve zero trits appended to the message
    static constexpr int L = 12;
    static inline void rem12(const std::vector<UTrit>& msg, std::array<UTrit, L>& out) { t3hip_crc12(msg.data(), msg.size(), out.data()); }
};

namespace t3 {
inline int& status_slot() { static thread_local int s = T3_OK; return s; }
inline int last_status() { return status_slot(); }                     // T3_* code of the last call on this thread
inline bool ok(int rc) { status_slot() = rc; return rc == T3_OK; }
inline bool ensure_device(int dev = 0) { return t3hip_is_ready() || ok(t3hip_init(dev)); }

template <class Cfg> inline t3_cfg to_pod(const Cfg& c, uint32_t superframe_words) {
    t3_cfg p; std::memset(&p, 0, sizeof p);
    p.profile = (uint8_t)c.profile;
    for (int i = 0; i < 9; ++i) p.band_profile[i] = c.uep.band_profile[i];
    p.tile_w = c.tile.w; p.tile_h = c.tile.h;
    p.seed_a = c.seed.a; p.seed_b = c.seed.b; p.seed_s0 = c.seed.s0;
    p.beacon_words_period = c.beacon.words_period; p.beacon_band_slot = c.beacon.band_slot; p.beacon_enabled = c.beacon.enabled;
    p.subword = (uint8_t)c.subword; p.centered = c.centered; p.coset = (uint8_t)c.coset; p.mode = c.mode;
    p.superframe_words = superframe_words;
    return p;
}
inline void from_pod(const t3_cfg& p, DecoderConfigSeen& c) {
    c.profile = (ProfileID)p.profile;
    for (int i = 0; i < 9; ++i) c.uep.band_profile[i] = p.band_profile[i];
    c.tile.w = p.tile_w; c.tile.h = p.tile_h;
    c.seed.a = p.seed_a; c.seed.b = p.seed_b; c.seed.s0 = p.seed_s0;
    c.beacon.words_period = p.beacon_words_period; c.beacon.band_slot = p.beacon_band_slot; c.beacon.enabled = p.beacon_enabled != 0;
    c.subword = (SubwordMode)p.subword; c.centered = p.centered != 0; c.coset = (CosetID)p.coset; c.mode = p.mode;
}
}  // namespace t3

// ---- GF(27) and the RS(26,k) block codec (OLD:383-663) ------------------------------------------------------------------
inline GF27 gf27_add(GF27 a, GF27 b) { return t3hip_gf27_add(a, b); }
inline GF27 gf27_sub(GF27 a, GF27 b) { return t3hip_gf27_sub(a, b); }
inline GF27 gf27_mul_poly(GF27 a, GF27 b) { return t3hip_gf27_mul(a, b); }
struct GF27Tables {
    std::array<GF27, 26 * 3> exp{}; std::array<int16_t, 27> log{}; std::array<GF27, 27 * 27> mul{}; std::array<GF27, 27> inv{};
    GF27 primitive = 0;
};
struct GF27Context {
    GF27Tables tab{};
    int order_of(GF27 g) const { if (g == 0 || g == 1) return -1; GF27 x = 1; for (int i = 1; i <= 26; ++i) { x = gf27_mul_poly(x, g); if (x == 1) return i; } return -1; }
    void init() { t3hip_gf27_tables(tab.exp.data(), tab.log.data(), tab.mul.data(), tab.inv.data()); tab.primitive = tab.exp[1]; }   // OLD:436-466 (primitive = 3)
    GF27 add(GF27 a, GF27 b) const { return gf27_add(a, b); }
    GF27 sub(GF27 a, GF27 b) const { return gf27_sub(a, b); }
    GF27 mul(GF27 a, GF27 b) const { return tab.mul[a * 27 + b]; }
    GF27 inv(GF27 a) const { return tab.inv[a]; }
    GF27 pow_alpha(int e) const { return tab.exp[(e % 26 + 26) % 26]; }
    int log(GF27 a) const { return tab.log[a]; }
};
// One block per call, as in the reference: 26 symbols are control data and are coded on the host (t3hip_rs_encode_block_host /
// t3hip_rs_decode_block_host: the parity matrix and decoder the kernels are built from, no launch) -- a caller that walks blocks
// one at a time, like the reference's self-test OLD:1172-1207, pays integer arithmetic, not 10 us per block.  Bulk work goes
// through t3hip_rs_encode_blocks / t3hip_rs_decode_blocks (device) or, for whole frames, encode_profile_from_raw / decode_profile_to_raw.
// `mode` is build-side: T3_MODE_COMPAT = the reference's arithmetic (its parity map, Forney with add), T3_MODE_FIXED = v6c; a codec
// owned by a context follows that context's cfg.mode (mode_ref).
struct RSCodec {
    GF27Context* gf = nullptr; RSParams params{}; std::vector<GF27> g; uint8_t mode = T3_MODE_COMPAT; const uint8_t* mode_ref = nullptr;
    int arithmetic() const { return mode_ref ? *mode_ref : mode; }
    void init(GF27Context* c, RSParams p) { gf = c; params = p; build_gen(); }
    void build_gen() { g.assign((size_t)(params.n - params.k + 1), 0); if (t3hip_rs_generator(params.k, g.data()) != T3_OK) g.assign(1, 1); }   // OLD:501-516
    bool encode_block(const GF27* data_k, GF27* out_n) const { return t3hip_rs_encode_block_host(params.k, arithmetic(), data_k, out_n) == T3_OK; }   // OLD:517-535
    GF27 poly_eval(const std::vector<GF27>& p, GF27 x) const {   // OLD:536-545 (Horner)
        GF27 acc = 0;
        for (int i = (int)p.size() - 1; i >= 0; --i) acc = gf27_add(gf27_mul_poly(acc, x), p[(size_t)i]);
        return acc;
    }
    // OLD:546-662: inout_n corrected in place, out_k only on success
    bool decode_block(GF27* inout_n, GF27* out_k) const { return t3hip_rs_decode_block_host(params.k, arithmetic(), inout_n, out_k) == 1; }
};

// ---- contexts (OLD:885-916): the field tables and the five codecs as members, so reference code that reaches into them
// (ectx.rs_p3.encode_block(...), dctx.gf.mul(...)) compiles unchanged.  The codecs follow cfg.mode / cfg_last_seen.mode.  Unlike the
// reference's (RSCodec::gf is a raw pointer into the owning context, OLD:492) these can be copied: a copy re-binds its own members.
struct EncoderContext {
    GF27Context gf; RSCodec rs_p1, rs_p2, rs_p3, rs_p4, rs_hdr; EncoderConfig cfg;
    EncoderContext() { bind(); uep_uniform(cfg.uep, 1); }
    EncoderContext(const EncoderContext& o) : cfg(o.cfg) { bind(); }
    EncoderContext& operator=(const EncoderContext& o) { cfg = o.cfg; return *this; }
private:
    void bind() {
        gf.init();
        RSCodec* cs[5] = {&rs_p1, &rs_p2, &rs_p3, &rs_p4, &rs_hdr};
        const RSParams ps[5] = {rs_params_for(ProfileID::P1_RS26_24), rs_params_for(ProfileID::P2_RS26_22), rs_params_for(ProfileID::P3_RS26_20), rs_params_for(ProfileID::P4_RS26_18), RSParams{26, 18}};
        for (int i = 0; i < 5; ++i) { cs[i]->init(&gf, ps[i]); cs[i]->mode_ref = &cfg.mode; }
    }
};
struct DecoderContext {
    GF27Context gf; RSCodec rs_p1, rs_p2, rs_p3, rs_p4, rs_hdr; DecoderConfigSeen cfg_last_seen;
    DecoderContext() { bind(); uep_uniform(cfg_last_seen.uep, 1); }
    DecoderContext(const DecoderContext& o) : cfg_last_seen(o.cfg_last_seen) { bind(); }
    DecoderContext& operator=(const DecoderContext& o) { cfg_last_seen = o.cfg_last_seen; return *this; }
private:
    void bind() {
        gf.init();
        RSCodec* cs[5] = {&rs_p1, &rs_p2, &rs_p3, &rs_p4, &rs_hdr};
        const RSParams ps[5] = {rs_params_for(ProfileID::P1_RS26_24), rs_params_for(ProfileID::P2_RS26_22), rs_params_for(ProfileID::P3_RS26_20), rs_params_for(ProfileID::P4_RS26_18), RSParams{26, 18}};
        for (int i = 0; i < 5; ++i) { cs[i]->init(&gf, ps[i]); cs[i]->mode_ref = &cfg_last_seen.mode; }
    }
};

struct HeaderCodec {
    static HeaderPack pack(const SuperframeHeader& h) {
        EncoderConfig c; c.profile = h.profile; c.uep = h.uep; c.tile = h.tile; c.seed = h.seed; c.beacon = h.beacon;
        c.subword = h.subword; c.centered = h.centered; c.coset = h.coset;
        const t3_cfg p = t3::to_pod(c, 0);
        HeaderPack out; t3hip_header_pack(&p, h.frame_seq, h.band_map_hash, out.symbols.data());
        return out;
    }
    static bool check(const HeaderPack& p) { return t3hip_header_check(p.symbols.data()) == 1; }
    static SuperframeHeader unpack(const HeaderPack& p) {
        t3_cfg c; std::memset(&c, 0, sizeof c); uint32_t fs = 0, bh = 0;
        t3hip_header_unpack(p.symbols.data(), &c, &fs, &bh);
        DecoderConfigSeen d; t3::from_pod(c, d);
        SuperframeHeader h; h.magic = (uint16_t)(p.symbols[0] % 27 + 27 * (p.symbols[1] % 27)); h.version = p.symbols[2] % 27;
        h.profile = d.profile; h.uep = d.uep; h.tile = d.tile; h.seed = d.seed; h.beacon = d.beacon;
        h.subword = d.subword; h.centered = d.centered; h.coset = d.coset; h.frame_seq = fs; h.band_map_hash = bh;
        return h;
    }
};

// ---- RAW packer primitives (OLD:675-722) ---------------------------------------------------------------------------------
inline void i2tr(uint32_t v, int w, std::array<UTrit, 27>& d, int s) { for (int i = 0; i < w; ++i) { d[(size_t)(s + i)] = (UTrit)(v % 3); v /= 3; } }
inline uint32_t tr2i(const std::array<UTrit, 27>& d, int w, int s) { uint32_t val = 0, p = 1; for (int i = 0; i < w; ++i) { val += p * d[(size_t)(s + i)]; p *= 3; } return val; }
// One word = 26 trits = one integer below 3^26 < 2^42: the components are digit runs of it (Y: 5 trits, Cb + 40 and Cr + 40: 4 each,
// pixel b 13 trits further up, trit 26 = 0), the nine symbols are its base-27 digits.  Host integer arithmetic, no device: a
// caller that loops these as OLD:728-733 does pays nanoseconds per word.  No clamping, as in the reference: a component enters
// as (uint32 cast) mod 3^width (OLD:675-682, 697-702).
inline void pack_two_pixels(const PixelYCbCrQuant& a, const PixelYCbCrQuant& b, Word27& w) {          // OLD:693-705
    auto px13 = [](const PixelYCbCrQuant& p) -> uint64_t {
        return (uint64_t)((uint32_t)p.Yq % 243u) + 243ull * ((uint32_t)((int)p.Cbq + 40) % 81u) + 19683ull * ((uint32_t)((int)p.Crq + 40) % 81u);
    };
    uint64_t v = px13(a) + 1594323ull * px13(b);                      // 3^13
    for (int s = 0; s < SYM_PER_WORD; ++s) { w.sym[(size_t)s] = (GF27)(v % 27u); v /= 27u; }
}
inline void unpack_two_pixels(const Word27& w, PixelYCbCrQuant& a, PixelYCbCrQuant& b) {            // OLD:706-722 (unpack3 reduces every digit; trit 26 ignored)
    uint64_t v = 0;
    for (int s = SYM_PER_WORD - 1; s >= 0; --s) v = 27u * v + w.sym[(size_t)s] % 27u;
    auto px13 = [](uint64_t u, PixelYCbCrQuant& p) {
        p.Yq = (uint16_t)(u % 243u); p.Cbq = (int16_t)((int)(u / 243u % 81u) - 40); p.Crq = (int16_t)((int)(u / 19683u % 81u) - 40);
    };
    px13(v % 1594323ull, a); px13(v / 1594323ull % 1594323ull, b);
}
// ---- 2-D boustrophedon on a symbol vector (OLD:750-813) --------------------------------------------------------------------
inline void interleave2D_boustrophedon(std::vector<GF27>& syms, Tile2D tile) { if (!t3hip_is_ready()) t3hip_init(0); t3hip_interleave2d(syms.data(), syms.size(), tile.w, tile.h, 0); }
inline void deinterleave2D_boustrophedon(std::vector<GF27>& syms, Tile2D tile) { if (!t3hip_is_ready()) t3hip_init(0); t3hip_interleave2d(syms.data(), syms.size(), tile.w, tile.h, 1); }

// ---- RAW packer (OLD:723-747; `_subword` variants NEWH:113-125 / NEWC:139-155) ------------------------------------
inline bool encode_raw_pixels_to_words(const std::vector<PixelYCbCrQuant>& px, std::vector<Word27>& out) {
    out.clear();
    if (!t3::ensure_device()) return false;
    out.resize((px.size() + 1) / 2);
    return t3::ok(t3hip_pack_pixels(px.data(), px.size(), out.data()));
}
inline bool decode_raw_words_to_pixels(const std::vector<Word27>& in, std::vector<PixelYCbCrQuant>& out) {
    out.clear();
    if (!t3::ensure_device()) return false;
    out.resize(in.size() * 2);
    return t3::ok(t3hip_unpack_words(in.data(), in.size(), out.data()));
}
inline bool encode_raw_pixels_to_words_subword(const std::vector<PixelYCbCrQuant>& px, SubwordMode sub, std::vector<Word27>& out) {
    if (!is_valid_subword(sub)) { t3::status_slot() = T3_E_ARG; return false; }
    return encode_raw_pixels_to_words(px, out);
}
inline bool decode_raw_words_to_pixels_subword(const std::vector<Word27>& in, SubwordMode sub, std::vector<PixelYCbCrQuant>& out) {
    if (!is_valid_subword(sub)) { t3::status_slot() = T3_E_ARG; return false; }
    return decode_raw_words_to_pixels(in, out);
}

// ---- RGB8 <-> quantised YCbCr bridge (old/include/io_image.hpp:17-21,156-195) --------------------------------------------
struct ImageU8 { int w = 0, h = 0, c = 0; std::vector<uint8_t> data; };
inline void rgb_to_quant_stream(const ImageU8& rgb, std::vector<PixelYCbCrQuant>& out) {
    out.clear();
    const size_t n = (size_t)(rgb.w > 0 ? rgb.w : 0) * (size_t)(rgb.h > 0 ? rgb.h : 0);
    if (!n || rgb.data.size() < 3 * n || !t3::ensure_device()) return;
    out.resize(n);
    if (!t3::ok(t3hip_rgb_to_quant(rgb.data.data(), n, out.data()))) out.clear();
}
inline void quant_stream_to_rgb(const std::vector<PixelYCbCrQuant>& q, int w, int h, ImageU8& out) {
    out.w = w; out.h = h; out.c = 3;
    const size_t n = (size_t)(w > 0 ? w : 0) * (size_t)(h > 0 ? h : 0);
    out.data.assign(n * 3, 0);
    const size_t m = q.size() < n ? q.size() : n;                     // io_image.hpp:181: stops when the stream runs out
    if (!m || !t3::ensure_device()) return;
    t3::ok(t3hip_quant_to_rgb(q.data(), m, out.data.data()));
}

// centring blits, io_image.hpp:125-140 and :215-235 (a source wider than the canvas / a window wider than the frame, where the
// reference runs over the row, leaves the zeroed canvas / an all-zero window)
inline void blit_center_rgb(const ImageU8& src, int canvasW, int canvasH, ImageU8& dst) {
    dst.w = canvasW; dst.h = canvasH; dst.c = 3;
    dst.data.assign((size_t)(canvasW > 0 ? canvasW : 0) * (size_t)(canvasH > 0 ? canvasH : 0) * 3, 0);
    if (dst.data.empty() || src.w <= 0 || src.h <= 0 || !t3::ensure_device()) return;
    t3::ok(t3hip_blit_center_rgb(src.data.data(), src.w, src.h, dst.data.data(), canvasW, canvasH));
}
inline void extract_center_q(const std::vector<PixelYCbCrQuant>& q_full, int fullW, int fullH, int subW, int subH, std::vector<PixelYCbCrQuant>& q_sub) {
    q_sub.assign((size_t)(subW > 0 ? subW : 0) * (size_t)(subH > 0 ? subH : 0), PixelYCbCrQuant{});
    if (q_sub.empty() || fullW <= 0 || fullH <= 0 || q_full.size() < (size_t)fullW * (size_t)fullH || !t3::ensure_device()) return;
    t3::ok(t3hip_extract_center_q(q_full.data(), fullW, fullH, q_sub.data(), subW, subH));
}

// ---- subword trit streams (OLD:834-859) and wire packings (include/ternary_packing.hpp, namespace tpack) --------------
inline void extract_subword_stream_from_words(const std::vector<Word27>& words, int N, std::vector<UTrit>& out) {
    out.clear();
    if (N < 1 || N > 27 || !t3::ensure_device()) return;
    out.resize(words.size() * (size_t)N);
    if (!t3::ok(t3hip_subword_extract(words.data(), words.size(), N, out.data()))) out.clear();
}
inline void build_words_from_subword_stream(const std::vector<UTrit>& in, int N, std::vector<Word27>& out, UTrit fill = 0) {
    out.clear();
    if (N < 1 || N > 27 || !t3::ensure_device()) return;
    out.resize((size_t)t3hip_subword_words(in.size(), N));
    uint64_t nw = 0;
    if (!t3::ok(t3hip_subword_build(in.data(), in.size(), N, fill, out.data(), out.size(), &nw))) { out.clear(); return; }
    out.resize((size_t)nw);
}
namespace tpack {
inline void ut_to_base243(const std::vector<UTrit>& in, std::vector<uint8_t>& out) {   // TPACK:28-38
    out.clear();
    if (!t3::ensure_device()) return;
    out.resize((size_t)t3hip_base243_bytes(in.size()));
    uint64_t nb = 0;
    if (!t3::ok(t3hip_base243_pack(in.data(), in.size(), out.data(), out.size(), &nb))) { out.clear(); return; }
    out.resize((size_t)nb);
}
inline bool base243_to_ut(const std::vector<uint8_t>& in, std::vector<UTrit>& out) {     // TPACK:40-50
    out.clear();
    if (in.size() < 4 || !t3::ensure_device()) return false;
    out.resize(5 * (in.size() - 4));
    uint64_t nt = 0;
    if (!t3::ok(t3hip_base243_unpack(in.data(), in.size(), out.data(), out.size(), &nt))) { out.clear(); return false; }
    out.resize((size_t)nt);
    return true;
}
inline void words_to_bytes(const std::vector<Word27>& words, std::vector<uint8_t>& out) {  // TPACK:53-58
    out.clear();
    if (!t3::ensure_device()) return;
    out.resize(words.size() * 9);
    if (!t3::ok(t3hip_mod27_bytes((const uint8_t*)words.data(), out.size(), out.data()))) out.clear();
}
inline void bytes_to_words(const std::vector<uint8_t>& in, std::vector<Word27>& out) {     // TPACK:60-65
    out.clear();
    if (in.size() % 9 != 0 || !t3::ensure_device()) return;
    out.resize(in.size() / 9);
    if (!t3::ok(t3hip_mod27_bytes(in.data(), in.size(), (uint8_t*)out.data()))) out.clear();
}
}  // namespace tpack

// ---- profile encode / decode (OLD:995-1169) --------------------------------------------------------------------------
inline bool encode_profile_from_raw(const std::vector<Word27>& in, std::vector<Word27>& out, EncoderContext& ectx) {
    out.clear();
    if (!t3::ensure_device()) return false;
    const t3_cfg c = t3::to_pod(ectx.cfg, ectx.cfg.superframe_words);
    const uint64_t cap = t3hip_encoded_words(in.size(), &c);
    out.resize(cap); uint64_t n = 0;
    const bool good = t3::ok(t3hip_encode_profile(in.data(), in.size(), &c, out.data(), cap, &n));
    out.resize(good ? n : 0);
    return good;
}
inline bool decode_profile_to_raw(const std::vector<Word27>& in, std::vector<Word27>& out, DecoderContext& dctx) {
    out.clear();
    if (!t3::ensure_device()) return false;
    t3_cfg c = t3::to_pod(dctx.cfg_last_seen, 0);
    out.resize(in.size() + 16); uint64_t n = 0;
    const int rc = t3hip_decode_profile(in.data(), in.size(), &c, out.data(), out.size(), &n);
    t3::from_pod(c, dctx.cfg_last_seen);                 // the header, once decoded, is remembered even if a block fails (OLD:1006-1017)
    out.resize(rc == T3_OK ? n : 0);
    return t3::ok(rc);
}
// pixels -> coded words and back in one launch each (= OLD:723 + OLD:1043, OLD:995 + OLD:735)
inline bool encode_frame(const std::vector<PixelYCbCrQuant>& px, std::vector<Word27>& out, EncoderContext& ectx) {
    out.clear();
    if (!t3::ensure_device()) return false;
    const t3_cfg c = t3::to_pod(ectx.cfg, ectx.cfg.superframe_words);
    const uint64_t cap = t3hip_encoded_words((px.size() + 1) / 2, &c);
    out.resize(cap); uint64_t n = 0;
    const bool good = t3::ok(t3hip_encode_frame(px.data(), px.size(), &c, out.data(), cap, &n));
    out.resize(good ? n : 0);
    return good;
}
inline bool decode_frame(const std::vector<Word27>& in, std::vector<PixelYCbCrQuant>& px, DecoderContext& dctx) {
    px.clear();
    if (!t3::ensure_device()) return false;
    t3_cfg c = t3::to_pod(dctx.cfg_last_seen, 0);
    px.resize(2 * (in.size() + 16)); uint64_t n = 0;
    const int rc = t3hip_decode_frame(in.data(), in.size(), &c, px.data(), px.size(), &n);
    t3::from_pod(c, dctx.cfg_last_seen);
    px.resize(rc == T3_OK ? n : 0);
    return t3::ok(rc);
}

// ---- self-tests with the reference's inputs (OLD:1172-1230) ------------------------------------------------------------
// The reference's own versions return false / false (its encode_block is not an RS encoder and its framings disagree).
// Default: the same inputs through FIXED arithmetic, where both pass.  Compile with -DT3_SELFTEST_REFERENCE_ARITHMETIC to run them
// exactly as the reference does (COMPAT arithmetic, the reference's statements in the reference's order): both then return the
// reference's verdicts, false / false (tests/golden/ref_vectors.json "selftests").
#ifdef T3_SELFTEST_REFERENCE_ARITHMETIC
static constexpr uint8_t T3_SELFTEST_MODE = T3_MODE_COMPAT;
#else
static constexpr uint8_t T3_SELFTEST_MODE = T3_MODE_FIXED;
#endif
inline bool selftest_rs_unit() {   // OLD:1172-1207: same data pattern, error positions and values (std::mt19937 rng(1)); host arithmetic only
    GF27Context gf; gf.init();
    std::mt19937 rng(1);
    for (ProfileID pid : {ProfileID::P1_RS26_24, ProfileID::P2_RS26_22, ProfileID::P3_RS26_20, ProfileID::P4_RS26_18}) {
        RSCodec rs; rs.mode = T3_SELFTEST_MODE; rs.init(&gf, rs_params_for(pid));
        const int n = rs.params.n, k = rs.params.k, t = (n - k) / 2;
        std::vector<GF27> data((size_t)k);
        for (int i = 0; i < k; ++i) data[(size_t)i] = (GF27)((i * 5 + 7) % 27);
        std::vector<GF27> code((size_t)n);
        rs.encode_block(data.data(), code.data());
        std::uniform_int_distribution<int> pos(0, n - 1), val(1, 26);
        std::vector<int> used;
        for (int e = 0; e < t; ++e) {
            int p;
            do { p = pos(rng); } while (std::find(used.begin(), used.end(), p) != used.end());
            used.push_back(p);
            code[(size_t)p] = gf.add(code[(size_t)p], (GF27)val(rng));
        }
        std::vector<GF27> outk((size_t)k);
        if (!rs.decode_block(code.data(), outk.data())) return false;
        if (outk != data) return false;
    }
    return true;
}
inline bool selftest_api_roundtrip() {   // OLD:1208-1230
    std::vector<PixelYCbCrQuant> px(64);
    for (size_t i = 0; i < px.size(); ++i) { px[i].Yq = (uint16_t)(i * 7 % 243); px[i].Cbq = (int16_t)((int)(i * 3 % 81) - 40); px[i].Crq = (int16_t)((int)(i * 5 % 81) - 40); }
    std::vector<Word27> raw, coded, back;
    if (!encode_raw_pixels_to_words(px, raw)) return false;
    EncoderContext e; e.cfg.profile = ProfileID::P2_RS26_22; e.cfg.mode = T3_SELFTEST_MODE; uep_luma_priority(e.cfg.uep);
    if (!encode_profile_from_raw(raw, coded, e)) return false;
    DecoderContext d; d.cfg_last_seen.mode = T3_SELFTEST_MODE;
    if (!decode_profile_to_raw(coded, back, d)) return false;
    const size_t L = std::min(raw.size(), back.size());               // the reference compares the common prefix (OLD:1226-1228)
    return std::memcmp(back.data(), raw.data(), L * 9) == 0;
}
