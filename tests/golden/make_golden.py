#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.json from the UNMODIFIED reference (oracle/_ref/libt3ref.so, built by
`make -C oracle ref` from /root/reference in place).  Runs only in the build container; the JSON it writes
is DATA (inputs + expected outputs), committed so that the GPU box — where /root/reference does not
exist — can still check against the reference's behaviour.

    python tests/golden/make_golden.py [--with-8k]

--with-8k also regenerates the two 8K frame hashes (≈30 s of reference CPU time)."""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402

FRAME_CFGS = {
    # name: (cfg kwargs)                              SURVEY Appendix A rows
    "raw": dict(profile=0xFF),
    "p3_uniform20": dict(profile=2, uep=2),
    "p2_luma": dict(profile=1, uep="luma"),
    "p5_tile64_luma": dict(profile=4, uep="luma", tile=(64, 64)),
    "p2_beacon83": dict(profile=1, uep=1, beacon=(83, 2, 1)),
    "p5_tile64_uniform20": dict(profile=4, uep=2, tile=(64, 64)),
    "p4_mixed_tile7x5_seedwrap": dict(profile=4, uep=[0, 1, 2, 3, 0, 1, 2, 3, 1], tile=(7, 5), seed=(0xFFFFFFFF, 0xFFFFFFFE, 5)),
    "p1_beacon3_slot8": dict(profile=0, uep=0, beacon=(3, 8, 1), seed=(2, 1, 0)),
}


def L(a):
    return [int(x) for x in np.asarray(a).reshape(-1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--with-8k", action="store_true")
    args = ap.parse_args()
    ol.build_oracle()
    ref, orc = ol.Ref(), ol.oracle()  # orc is used ONLY for the LCG input generator and FNV hashing
    out_path = os.path.join(HERE, "ref_vectors.json")
    old = json.load(open(out_path)) if os.path.exists(out_path) else {}
    G = {"_about": "captured from oracle/_ref (unmodified reference) by tests/golden/make_golden.py"}

    t = ref.gf_tables()
    G["gf"] = {"exp": L(t["exp"]), "log": L(t["log"]), "mul": L(t["mul"]), "inv": L(t["inv"]), "prim": t["prim"]}
    G["rs"] = {}
    rng = np.random.default_rng(20251017)
    for k in (24, 22, 20, 18):
        tcap = (26 - k) // 2
        eye = np.eye(k, dtype=np.uint8)
        P = ref.rs_encode_blocks(k, eye)[:, k:]
        pat = ((np.arange(k) * 5 + 7) % 27).astype(np.uint8)  # selftest_rs_unit data OLD:1186
        data = rng.integers(0, 27, size=(24, k), dtype=np.uint8)
        code = ref.rs_encode_blocks(k, data)
        rx = []
        rx.append(code[:6])                                  # the encoder's own (invalid) codewords, clean
        for e in range(1, tcap + 2):                         # e = 1..t+1 symbol errors on them
            c = code[6:12].copy()
            for row in c:
                pos = rng.choice(26, size=e, replace=False)
                row[pos] = (row[pos] + rng.integers(1, 27, size=e)) % 27
            rx.append(c)
        rx.append(rng.integers(0, 27, size=(12, 26), dtype=np.uint8))  # arbitrary words
        # valid codewords under the decoder's convention, clean and with <= t errors (c+2e behaviour)
        valid = orc.rs_encode_blocks(k, data[:12], mode=1)
        assert ref.rs_decode_blocks(k, valid)[2].all()
        rx.append(valid[:4])
        v = valid[4:12].copy()
        for i, row in enumerate(v):
            e = 1 + i % tcap
            pos = rng.choice(26, size=e, replace=False)
            row[pos] = (row[pos] + rng.integers(1, 27, size=e)) % 27
        rx.append(v)
        rx = np.concatenate(rx)
        o, dk, ok = ref.rs_decode_blocks(k, rx)
        G["rs"][str(k)] = {"g": L(ref.rs_generator(k)), "P_compat": L(P), "pattern_parity": L(ref.rs_encode_blocks(k, pat)[0, k:]),
                           "enc_data": L(data), "enc_code": L(code),
                           "dec_in": L(rx), "dec_inout": L(o), "dec_outk": L(dk), "dec_ok": L(ok)}

    # header KATs
    G["header"] = []
    hk = [dict(profile=2, uep="luma", tile=(64, 64), beacon=(83, 2, 1), fs=1234, bh=0),
          dict(profile=1, uep=1, fs=0, bh=0),
          dict(profile=4, uep=[3, 2, 1, 0, 3, 2, 1, 0, 3], tile=(1920, 1080), seed=(5, 7, 11), beacon=(8192, 7, 1), subword=18, centered=0, coset=2, fs=987654, bh=31337)]
    for h in hk:
        h = dict(h); fs, bh = h.pop("fs"), h.pop("bh")
        c = ol.make_cfg(**h)
        s = ref.header_pack(c, fs, bh)
        u, ufs, ubh, mg, ver = ref.header_unpack(s)
        G["header"].append({"cfg": c.as_dict(), "frame_seq": fs, "band_map_hash": bh, "syms": L(s), "check": ref.header_check(s),
                            "unpacked": u.as_dict(), "u_frame_seq": ufs, "u_band_map_hash": ubh, "magic": mg, "version": ver})
    G["crc12"] = []
    for n in (0, 1, 12, 69):
        tr = rng.integers(0, 3, n, dtype=np.uint8)
        G["crc12"].append({"in": L(tr), "out": L(ref.crc12(tr))})
    G["scramble"] = []
    s = rng.integers(0, 27, 40, dtype=np.uint8)
    for seed in ((1, 1, 1), (0, 0, 0), (2, 1, 0), (0xFFFFFFFF, 0xFFFFFFFE, 5), (3, 1, 2)):
        G["scramble"].append({"seed": list(seed), "in": L(s), "out": L(ref.scramble(s, *seed, 0)), "inv": L(ref.scramble(s, *seed, 1))})
    G["interleave"] = []
    for n, w, h in ((30, 4, 3), (100, 7, 5), (64, 8, 8), (17, 100, 3)):
        s = rng.integers(0, 27, n, dtype=np.uint8)
        G["interleave"].append({"w": w, "h": h, "in": L(s), "out": L(ref.interleave2d(s, w, h, 0)), "inv": L(ref.interleave2d(s, w, h, 1))})
    G["beacon_symbol"] = [[p, f, hl, int(ref.beacon_symbol(p, f, hl))] for p in range(5) for f in (0, 1, 2, 4, 8192 % 5) for hl in (0, 1, 2)]

    # packer: first LCG word + quirk pixels (no clamping) + odd count
    px = orc.lcg_pixels(5)
    G["pack"] = {"lcg5_px": [[int(p["Yq"]), int(p["Cbq"]), int(p["Crq"])] for p in px], "lcg5_words": L(ref.pack_pixels(px))}
    q = np.zeros(4, ol.PIXEL_DT); q["Yq"] = [300, 65535, 242, 0]; q["Cbq"] = [-50, 32767, -32768, 41]; q["Crq"] = [40, -41, 100, -40]
    G["pack"]["quirk_px"] = [[int(p["Yq"]), int(p["Cbq"]), int(p["Crq"])] for p in q]
    G["pack"]["quirk_words"] = L(ref.pack_pixels(q))
    G["pack"]["quirk_unpacked"] = [[int(p["Yq"]), int(p["Cbq"]), int(p["Crq"])] for p in ref.unpack_words(ref.pack_pixels(q))]

    # small complete frame: selftest_api_roundtrip's 64-pixel pattern (OLD:1210-1223), P2 + luma UEP -> all 32 words
    i = np.arange(64)
    st = np.zeros(64, ol.PIXEL_DT); st["Yq"] = (i * 7) % 243; st["Cbq"] = (i * 3) % 81 - 40; st["Crq"] = (i * 5) % 81 - 40
    raw = ref.pack_pixels(st)
    _, enc = ref.encode_profile(raw, ol.make_cfg(profile=1, uep="luma"))
    G["selftest64"] = {"raw_words": L(raw), "p2_luma_words": L(enc)}

    # frame hashes (FNV-1a-64 over output symbol bytes), LCG input seed 12345
    G["frames"] = {}
    for (w, h) in ((8, 8), (256, 256), (1920, 1080)):
        px = orc.lcg_pixels(w * h)
        raw = ref.pack_pixels(px)
        ent = {"raw_words": len(raw), "raw_hash": ol.fnv_hex(raw), "cfgs": {}}
        for name, kw in FRAME_CFGS.items():
            if (w, h) == (1920, 1080) and name not in ("p3_uniform20", "p5_tile64_luma"):
                continue
            cfg = ol.make_cfg(**kw)
            rc, enc = ref.encode_profile(raw, cfg, cap=len(raw) * 2 + 64)
            assert rc == 0
            ent["cfgs"][name] = {"cfg": cfg.as_dict(), "out_words": len(enc), "hash": ol.fnv_hex(enc), "crc32": int(orc.crc32(enc)),
                                 "first_words": L(enc[:8]), "last_words": L(enc[-2:])}
        G["frames"]["%dx%d" % (w, h)] = ent
    if args.with_8k:
        px = orc.lcg_pixels(7680 * 4320)
        ent = {"cfgs": {}}
        for name in ("p3_uniform20", "p5_tile64_luma", "p2_beacon83"):
            cfg = ol.make_cfg(**FRAME_CFGS[name])
            rc, enc = ref.encode_frame(px, cfg, cap=len(px))
            assert rc == 0
            ent["cfgs"][name] = {"cfg": cfg.as_dict(), "out_words": len(enc), "hash": ol.fnv_hex(enc), "crc32": int(orc.crc32(enc)),
                                 "first_words": L(enc[:8]), "last_words": L(enc[-2:])}
            print("8k", name, len(enc), ent["cfgs"][name]["hash"], flush=True)
        G["frames"]["7680x4320"] = ent
    elif "frames" in old and "7680x4320" in old["frames"]:
        G["frames"]["7680x4320"] = old["frames"]["7680x4320"]

    # decode_profile_to_raw on decoder-consistent streams (see tests/test_oracle_vs_ref.py) — small ones, full bytes
    sys.path.insert(0, os.path.dirname(HERE))
    from test_oracle_vs_ref import decoder_consistent_stream
    G["decode_streams"] = []
    drng = np.random.default_rng(77)
    for name, kw, nbw, corrupt in (("p3_uniform20", dict(profile=2, uep=2), 60, 0), ("p2_luma", dict(profile=1, uep="luma"), 60, 1),
                                   ("p5_tile", dict(profile=4, uep="luma", tile=(7, 5)), 90, 0), ("p2_beacon", dict(profile=1, uep=1, beacon=(11, 2, 1)), 80, 2),
                                   ("p4_over_t", dict(profile=3, uep=0), 60, 4)):
        cfg = ol.make_cfg(**kw)
        s = decoder_consistent_stream(orc, drng, cfg, nbw, corrupt)
        seen = ol.make_cfg()
        rc, out = ref.decode_profile(s, seen)
        G["decode_streams"].append({"name": name, "in_words": L(s), "ok": rc == 0, "out_words": L(out), "seen": seen.as_dict()})
    # and the reference decoder on the reference encoder's output (false)
    G["decode_of_encode"] = []
    for name in ("p3_uniform20", "p2_luma", "p5_tile64_luma", "p2_beacon83"):
        cfg = ol.make_cfg(**FRAME_CFGS[name])
        raw = ref.pack_pixels(orc.lcg_pixels(2000, 99))
        _, enc = ref.encode_profile(raw, cfg)
        seen = ol.make_cfg()
        rc, out = ref.decode_profile(enc, seen)
        G["decode_of_encode"].append({"name": name, "cfg": cfg.as_dict(), "lcg_seed": 99, "n_px": 2000, "ok": rc == 0, "n_out": len(out), "seen": seen.as_dict()})
    G["selftests"] = list(ref.selftests())

    with open(out_path, "w") as f:
        json.dump(G, f, separators=(",", ":"))
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
