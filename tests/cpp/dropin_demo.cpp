// tests/cpp/dropin_demo.cpp — the reference's canonical caller sequence (old/src/main.cpp:15-27: quantised pixels ->
// encode_raw_pixels_to_words -> EncoderContext{P2, tile 64x64, beacon {83,2,true}} -> encode_profile_from_raw ->
// decode_profile_to_raw) written against include/ternary_codec_v6.hpp, i.e. what a maintainer gets by swapping the include.
// Prints one JSON line; tests/test_gpu_dropin.py checks it against the oracle.  Host compiler only (g++), links -lt3hip.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ternary_codec_v6.hpp"
#include "io_t3p_t3v.hpp"

static uint64_t fnv(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }

int main(int argc, char** argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 256, h = argc > 2 ? atoi(argv[2]) : 256;
    std::vector<PixelYCbCrQuant> q((size_t)w * h);
    uint32_t s = 12345;                                      // SURVEY §8d generator
    auto draw = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& p : q) { p.Yq = (uint16_t)(draw() % 243); p.Cbq = (int16_t)((int)(draw() % 81) - 40); p.Crq = (int16_t)((int)(draw() % 81) - 40); }

    std::vector<Word27> raw; bool ok_raw = encode_raw_pixels_to_words(q, raw);
    EncoderContext e; e.cfg.profile = ProfileID::P2_RS26_22; e.cfg.tile = {64, 64}; e.cfg.beacon = {83, 2, true};   // old/src/main.cpp:19
    std::vector<Word27> prof; bool ok_enc = encode_profile_from_raw(raw, prof, e);
    DecoderContext d; std::vector<Word27> raw2; bool ok_dec = decode_profile_to_raw(prof, raw2, d);                    // reference: false

    EncoderContext ef = e; ef.cfg.mode = T3_MODE_FIXED;     // the variant that does round-trip
    std::vector<Word27> proff; bool ok_encf = encode_frame(q, proff, ef);
    DecoderContext df; df.cfg_last_seen.mode = T3_MODE_FIXED;
    std::vector<PixelYCbCrQuant> q2; bool ok_decf = decode_frame(proff, q2, df);
    bool same = ok_decf && q2.size() == q.size() && memcmp(q2.data(), q.data(), q.size() * 6) == 0;

    // row f3: S24 subword stream of the raw words, base-243 wire form and back (OLD:834-859, TPACK:28-50)
    std::vector<UTrit> tr; extract_subword_stream_from_words(raw, 24, tr);
    std::vector<uint8_t> b243; tpack::ut_to_base243(tr, b243);
    std::vector<UTrit> tr2; bool ok_b243 = tpack::base243_to_ut(b243, tr2) && tr2 == tr;
    std::vector<Word27> w24; build_words_from_subword_stream(tr2, 24, w24);

    // row f2: the coded frame into a T3V6 / T3P6 container and back (io_t3p_t3v.hpp:40-84); payload CRC on the device
    int cont_ok = 0; unsigned long long t3v_hash = 0; size_t t3v_bytes = 0;
    if (argc > 3) {
        const std::string dir = argv[3], v = dir + "/demo.t3v", pth = dir + "/demo.t3p";
        std::string err;
        bool okc = T3Container::t3v_write(v, SubwordMode::S27, w, h, {prof, {}, proff}, "{\"codec\":\"v6\"}", {"{\"f\":0}", "", "{\"f\":2}"}, &err);
        SubwordMode sm; int rw = 0, rh = 0; std::string mg; uint64_t fc = 0; std::vector<T3Container::T3VFrameIndex> idx;
        okc = okc && T3Container::t3v_read_header(v, sm, rw, rh, mg, fc, idx, &err) && fc == 3 && rw == w && rh == h && idx[0].words == prof.size() && idx[1].words == 0;
        std::vector<Word27> back;
        okc = okc && T3Container::t3v_read_frame(v, 2, nullptr, back, &err) && back.size() == proff.size() && memcmp(back.data(), proff.data(), 9 * back.size()) == 0;
        okc = okc && T3Container::t3v_read_frame(v, 1, nullptr, back, &err) && back.empty();
        okc = okc && !T3Container::t3v_read_frame(v, 0, [](const std::string& m) { return m != "{\"f\":0}"; }, back, &err) && back.empty();
        okc = okc && T3Container::t3p_write(pth, SubwordMode::S24, w, h, raw, "{}", &err);
        okc = okc && T3Container::t3p_read_payload(pth, nullptr, back, &err) && back.size() == raw.size() && memcmp(back.data(), raw.data(), 9 * raw.size()) == 0;
        { FILE* f = fopen(pth.c_str(), "r+b"); fseek(f, 40, SEEK_SET); int c = fgetc(f); fseek(f, 40, SEEK_SET); fputc(c ^ 1, f); fclose(f); }   // flip a payload bit
        okc = okc && !T3Container::t3p_read_payload(pth, nullptr, back, &err) && err == "t3p: payload crc mismatch";
        cont_ok = okc ? 1 : 0;
        if (FILE* f = fopen(v.c_str(), "rb")) { std::vector<uint8_t> all; uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) all.insert(all.end(), buf, buf + n); fclose(f); t3v_bytes = all.size(); t3v_hash = fnv(all.data(), all.size()); }
    }
    printf("{\"containers_ok\":%d,\"t3v_bytes\":%zu,\"t3v_hash\":\"%016llx\",", cont_ok, t3v_bytes, t3v_hash);
    printf("\"sub_trits\":%zu,\"sub_hash\":\"%016llx\",\"b243_bytes\":%zu,\"b243_hash\":\"%016llx\",\"ok_b243\":%d,\"w24_hash\":\"%016llx\",",
           tr.size(), (unsigned long long)fnv(tr.data(), tr.size()), b243.size(), (unsigned long long)fnv(b243.data(), b243.size()), ok_b243 ? 1 : 0,
           (unsigned long long)fnv(w24.data(), w24.size() * 9));
    printf("\"ok_raw\":%d,\"raw_words\":%zu,\"raw_hash\":\"%016llx\",\"ok_enc\":%d,\"enc_words\":%zu,\"enc_hash\":\"%016llx\",\"ok_dec_compat\":%d,"
           "\"seen_profile\":%d,\"seen_tile_w\":%d,\"ok_enc_fixed\":%d,\"ok_dec_fixed\":%d,\"roundtrip_equal\":%d,\"selftest_api_roundtrip\":%d,\"status\":%d}\n",
           ok_raw, raw.size(), (unsigned long long)fnv(raw.data(), raw.size() * 9), ok_enc, prof.size(), (unsigned long long)fnv(prof.data(), prof.size() * 9), ok_dec,
           (int)d.cfg_last_seen.profile, (int)d.cfg_last_seen.tile.w, ok_encf, ok_decf, same, selftest_api_roundtrip(), t3::last_status());
    return 0;
}
