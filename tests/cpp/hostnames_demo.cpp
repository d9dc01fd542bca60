// tests/cpp/hostnames_demo.cpp — the names of the path that need NO device: pack_two_pixels / unpack_two_pixels (OLD:693-722, host
// integer arithmetic), RSCodec::encode_block / decode_block one block at a time (OLD:517-662), the context members the reference's
// callers may reach into (EncoderContext / DecoderContext: gf, rs_p1..rs_p4, rs_hdr, OLD:885-916) and selftest_rs_unit (OLD:1172-1207).
// stdin: n, then n pixels (Yq Cbq Crq) -> words of consecutive pixel pairs (odd n: zero pixel pad, OLD:730); then m, m words of 9
// symbols -> their pixels.  One JSON line on stdout.  Built twice by tests/test_host_logic.py: as is (self-tests in FIXED
// arithmetic) and with -DT3_SELFTEST_REFERENCE_ARITHMETIC (the reference's own arithmetic and verdicts).
#include <cstdio>
#include <vector>

#include "ternary_image_codec_v6_min.hpp"

int main() {
    int n = 0; if (scanf("%d", &n) != 1) return 2;
    std::vector<PixelYCbCrQuant> px((size_t)n);
    for (auto& p : px) { int y, cb, cr; if (scanf("%d %d %d", &y, &cb, &cr) != 3) return 2; p.Yq = (uint16_t)y; p.Cbq = (int16_t)cb; p.Crq = (int16_t)cr; }
    printf("{\"words\":[");
    for (size_t i = 0; i < px.size(); i += 2) {                                  // the reference's loop, OLD:728-733
        Word27 w; pack_two_pixels(px[i], i + 1 < px.size() ? px[i + 1] : PixelYCbCrQuant{}, w);
        for (int s = 0; s < 9; ++s) printf("%s%d", (i || s) ? "," : "", w.sym[(size_t)s]);
    }
    int m = 0; if (scanf("%d", &m) != 1) return 2;
    printf("],\"pixels\":[");
    for (int i = 0; i < m; ++i) {
        Word27 w; for (int s = 0; s < 9; ++s) { int v; if (scanf("%d", &v) != 1) return 2; w.sym[(size_t)s] = (GF27)v; }
        PixelYCbCrQuant a, b; unpack_two_pixels(w, a, b);
        printf("%s%d,%d,%d,%d,%d,%d", i ? "," : "", a.Yq, a.Cbq, a.Crq, b.Yq, b.Cbq, b.Crq);
    }
    // reaching into the contexts as reference code does; a context's codecs follow its cfg.mode; copies own their members
    EncoderContext e; DecoderContext d;
    std::vector<GF27> data(20), code(26), outk(20, 77), codef(26);
    for (int i = 0; i < 20; ++i) data[(size_t)i] = (GF27)((i * 5 + 7) % 27);
    e.rs_p3.encode_block(data.data(), code.data());                              // COMPAT: the reference's map
    e.cfg.mode = T3_MODE_FIXED; e.rs_p3.encode_block(data.data(), codef.data()); // FIXED: a real codeword
    d.cfg_last_seen.mode = T3_MODE_FIXED;
    DecoderContext d2 = d;                                                        // the copy decodes with its own tables
    codef[4] = d2.gf.add(codef[4], 7); codef[21] = d2.gf.add(codef[21], 19); codef[9] = d2.gf.add(codef[9], 1);
    const bool okf = d2.rs_p3.decode_block(codef.data(), outk.data());
    std::vector<GF27> hdr(18, 3), hc(26); e.rs_hdr.encode_block(hdr.data(), hc.data());
    printf("],\"compat_code\":[");
    for (int i = 0; i < 26; ++i) printf("%s%d", i ? "," : "", code[(size_t)i]);
    printf("],\"fixed_recovered\":%d,\"hdr_k\":%d,\"p1_k\":%d,\"p4_g\":%d,\"gf_mul\":%d,\"own_gf\":%d,\"selftest_rs_unit\":%d,\"reference_arithmetic\":%d}\n",
           (okf && outk == data) ? 1 : 0, e.rs_hdr.params.k, d.rs_p1.params.k, (int)e.rs_p4.g.size(), d.gf.mul(5, 7), (d2.rs_p3.gf == &d2.gf && d.rs_p3.gf == &d.gf) ? 1 : 0,
           selftest_rs_unit() ? 1 : 0, T3_SELFTEST_MODE == T3_MODE_COMPAT ? 1 : 0);
    return 0;
}
