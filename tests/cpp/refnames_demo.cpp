// tests/cpp/refnames_demo.cpp — the reference's remaining public names (GF27Context, RSCodec incl. poly_eval, gf27_*,
// scramble_symbol / descramble_symbol, encode_beacon_symbol, CRC3::rem12, interleave2D_boustrophedon, pack_two_pixels,
// selftest_rs_unit: OLD:81-113, 176-205, 383-663, 675-722, 750-813, 1172-1207) driven through include/ternary_codec_v6.hpp the
// way a caller of the reference header drives them.  One JSON line; tests/test_gpu_dropin.py checks it against the oracle
// and the golden vectors.  Host compiler only (g++), links -lt3hip.
#include <cstdio>
#include <vector>

#include "ternary_image_codec_v6_min.hpp"   // include/compat: the forwarding header a maintainer puts in front of the reference's

static uint64_t fnv(const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; uint64_t h = 1469598103934665603ull; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }

int main() {
    const bool rs_unit = selftest_rs_unit(), api = selftest_api_roundtrip();           // what old/src/main_bare.cpp:8 runs
    GF27Context gf; gf.init();
    // field: every sum, difference and product, and the context's tables
    std::vector<uint8_t> fld;
    for (int a = 0; a < 27; ++a) for (int b = 0; b < 27; ++b) { fld.push_back(gf27_add((GF27)a, (GF27)b)); fld.push_back(gf27_sub((GF27)a, (GF27)b)); fld.push_back(gf27_mul_poly((GF27)a, (GF27)b)); fld.push_back(gf.mul((GF27)a, (GF27)b)); }
    for (int a = 0; a < 27; ++a) { fld.push_back(gf.inv((GF27)a)); fld.push_back((uint8_t)(gf.log((GF27)a) & 0xFF)); }
    for (int e = -30; e < 60; ++e) fld.push_back(gf.pow_alpha(e));
    // RSCodec in the reference's own arithmetic (COMPAT): generator, encode_block on the self-test pattern, poly_eval, decode_block
    std::vector<uint8_t> rsv; int dec_ok = 0;
    for (ProfileID pid : {ProfileID::P1_RS26_24, ProfileID::P2_RS26_22, ProfileID::P3_RS26_20, ProfileID::P4_RS26_18}) {
        RSCodec rs; rs.init(&gf, rs_params_for(pid));
        const int k = rs.params.k;
        std::vector<GF27> data((size_t)k), code(26), outk((size_t)k, 77);
        for (int i = 0; i < k; ++i) data[(size_t)i] = (GF27)((i * 5 + 7) % 27);
        rs.encode_block(data.data(), code.data());
        rsv.insert(rsv.end(), rs.g.begin(), rs.g.end()); rsv.insert(rsv.end(), code.begin(), code.end());
        for (int x = 0; x < 27; ++x) rsv.push_back(rs.poly_eval(rs.g, (GF27)x));
        code[3] = gf.add(code[3], 5);
        const bool okd = rs.decode_block(code.data(), outk.data());
        dec_ok = dec_ok * 2 + (okd ? 1 : 0);
        rsv.insert(rsv.end(), code.begin(), code.end()); rsv.insert(rsv.end(), outk.begin(), outk.end());
    }
    // scrambler chain on a symbol ramp, there and back, with a wrap-around seed
    std::vector<uint8_t> scr; bool scr_back = true;
    for (ScramblerSeed seed : {ScramblerSeed{1, 1, 1}, ScramblerSeed{0xFFFFFFFFu, 0xFFFFFFFEu, 5}}) {
        uint32_t st = seed.s0 % 3, st2 = seed.s0 % 3;
        for (int i = 0; i < 60; ++i) { const GF27 s = (GF27)(i % 27), t = scramble_symbol(s, seed, st); scr.push_back(t); scr_back = scr_back && descramble_symbol(t, seed, st2) == s; }
    }
    std::vector<uint8_t> bea;
    for (int p = 0; p < 5; ++p) for (int f : {0, 1, 2, 4, 8192 % 5}) for (int h = 0; h < 3; ++h) bea.push_back(encode_beacon_symbol(BeaconPayload{(ProfileID)p, (uint16_t)f, (uint8_t)h}));
    std::vector<UTrit> msg; for (int i = 0; i < 69; ++i) msg.push_back((UTrit)((i * 7 + i / 5) % 3));
    std::array<UTrit, CRC3::L> crc{}; CRC3::rem12(msg, crc);
    std::vector<GF27> il(1000); for (size_t i = 0; i < il.size(); ++i) il[i] = (GF27)((i * 11 + i / 7) % 27);
    std::vector<GF27> il2 = il; interleave2D_boustrophedon(il2, Tile2D{7, 5}); std::vector<GF27> il3 = il2; deinterleave2D_boustrophedon(il3, Tile2D{7, 5});
    PixelYCbCrQuant a{156, -1, 21}, b{17, 31, 20}, a2, b2; Word27 w; pack_two_pixels(a, b, w); unpack_two_pixels(w, a2, b2);     // first packed word
    std::array<UTrit, 27> T{}; i2tr(200, 5, T, 3);
    printf("{\"selftest_rs_unit\":%d,\"selftest_api_roundtrip\":%d,\"field_hash\":\"%016llx\",\"primitive\":%d,\"order3\":%d,\"rs_hash\":\"%016llx\",\"dec_ok\":%d,"
           "\"scr_hash\":\"%016llx\",\"scr_back\":%d,\"beacon_hash\":\"%016llx\",\"crc12\":\"%d%d%d%d%d%d%d%d%d%d%d%d\",\"il_hash\":\"%016llx\",\"il_back\":%d,"
           "\"word\":[%d,%d,%d,%d,%d,%d,%d,%d,%d],\"unpack_equal\":%d,\"tr2i\":%u,\"status\":%d}\n",
           rs_unit, api, (unsigned long long)fnv(fld.data(), fld.size()), (int)gf.tab.primitive, gf.order_of(3), (unsigned long long)fnv(rsv.data(), rsv.size()), dec_ok,
           (unsigned long long)fnv(scr.data(), scr.size()), scr_back ? 1 : 0, (unsigned long long)fnv(bea.data(), bea.size()),
           crc[0], crc[1], crc[2], crc[3], crc[4], crc[5], crc[6], crc[7], crc[8], crc[9], crc[10], crc[11],
           (unsigned long long)fnv(il2.data(), il2.size()), il3 == il ? 1 : 0,
           w.sym[0], w.sym[1], w.sym[2], w.sym[3], w.sym[4], w.sym[5], w.sym[6], w.sym[7], w.sym[8],
           (a2.Yq == a.Yq && a2.Cbq == a.Cbq && a2.Crq == a.Crq && b2.Yq == b.Yq && b2.Cbq == b.Cbq && b2.Crq == b.Crq) ? 1 : 0, tr2i(T, 5, 3), t3::last_status());
    return 0;
}
