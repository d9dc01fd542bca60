"""GPU parity tests proper: the HIP path, called through the C-ABI, against the oracle on the same seeded inputs and
against the golden vectors captured from the reference.  Bit-exact everywhere (integer/byte work)."""
import json
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")))

CFGS = {
    "p3_uniform20": dict(profile=2, uep=2),
    "p2_luma": dict(profile=1, uep="luma"),
    "p5_tile64_luma": dict(profile=4, uep="luma", tile=(64, 64)),
    "p5_tile64_uniform20": dict(profile=4, uep=2, tile=(64, 64)),
    "p5_tile7x5_mixed4": dict(profile=4, uep=[0, 1, 2, 3, 0, 1, 2, 3, 1], tile=(7, 5)),
    "p2_beacon83": dict(profile=1, uep=1, beacon=(83, 2, 1)),
    "p1_beacon3_slot8": dict(profile=0, uep=0, beacon=(3, 8, 1), seed=(2, 1, 0)),
    "p4_seed_wrap": dict(profile=3, uep=3, seed=(0xFFFFFFFF, 0xFFFFFFFE, 5)),
    "p2_tile_ignored": dict(profile=1, uep=1, tile=(64, 64)),
    "p5_beacon_slot9": dict(profile=4, uep=1, tile=(3, 4), beacon=(5, 9, 1)),
    "p5_wide_tile": dict(profile=4, uep=2, tile=(5000, 3)),
    "p5_wide16_tile": dict(profile=4, uep=1, tile=(4112, 7)),
    "p5_one_symbol_rows": dict(profile=4, uep=2, tile=(1, 9)),
    "p3_seed_fixedpoint": dict(profile=2, uep=2, seed=(0, 2, 1)),
    "p3_seed_period2": dict(profile=2, uep=2, seed=(2, 0, 1)),
}


def both(t3, kw, mode=0):
    return t3.make_cfg(mode=mode, **kw), ol.make_cfg(mode=mode, **kw)


def rand_pixels(rng, n, in_range=True):
    px = np.zeros(n, ol.PIXEL_DT)
    if in_range:
        px["Yq"] = rng.integers(0, 243, n); px["Cbq"] = rng.integers(-40, 41, n); px["Crq"] = rng.integers(-40, 41, n)
    else:
        px["Yq"] = rng.integers(0, 65536, n); px["Cbq"] = rng.integers(-32768, 32768, n); px["Crq"] = rng.integers(-32768, 32768, n)
    return px


def test_pack_unpack(gpu, orc):
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 64, 1001, 65536):
        for in_range in (True, False):
            px = rand_pixels(rng, n, in_range)
            w = gpu.encode_raw_pixels_to_words(px)
            assert np.array_equal(w, orc.pack_pixels(px)), (n, in_range)
            assert np.array_equal(gpu.decode_raw_words_to_pixels(w), orc.unpack_words(w))
    w = rng.integers(0, 256, size=(4096, 9), dtype=np.uint8)       # non-canonical symbols
    assert np.array_equal(gpu.decode_raw_words_to_pixels(w), orc.unpack_words(w))
    assert len(gpu.encode_raw_pixels_to_words(np.zeros(0, ol.PIXEL_DT))) == 0
    g = GOLD["pack"]
    q = np.array([tuple(p) for p in g["quirk_px"]], ol.PIXEL_DT)
    assert list(gpu.encode_raw_pixels_to_words(q).reshape(-1)) == g["quirk_words"]
    assert gpu.encode_raw_pixels_to_words_subword(q, 13) is None and gpu.encode_raw_pixels_to_words_subword(q, 24) is not None


@pytest.mark.parametrize("k", [24, 22, 20, 18])
@pytest.mark.parametrize("mode", [0, 1])
def test_rs_block_level(gpu, orc, k, mode):
    import torch
    rng = np.random.default_rng(10 * k + mode)
    gk = GOLD["rs"][str(k)]
    data = np.concatenate([np.array(gk["enc_data"], np.uint8).reshape(-1, k), rng.integers(0, 27, size=(5000, k), dtype=np.uint8)])
    d = torch.from_numpy(data).cuda(); code = torch.zeros((len(data), 26), dtype=torch.uint8, device="cuda")
    gpu.rs_encode_blocks_dev(k, mode, d.data_ptr(), len(data), code.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(code.cpu().numpy(), orc.rs_encode_blocks(k, data, mode))
    if mode == 0:
        assert list(code.cpu().numpy()[:24].reshape(-1)) == gk["enc_code"]
    # decode: golden triples (reference behaviour), then random corruptions of valid and invalid codewords
    t = (26 - k) // 2
    rx = [np.array(gk["dec_in"], np.uint8).reshape(-1, 26)] if mode == 0 else []
    base = orc.rs_encode_blocks(k, data[:3000], mode=1)
    for e in range(0, t + 3):
        c = base[500 * (e % 6):500 * (e % 6) + 500].copy()
        for row in c:
            pos = rng.choice(26, size=e, replace=False)
            row[pos] = (row[pos] + rng.integers(1, 27, size=e)) % 27
        rx.append(c)
    rx.append(rng.integers(0, 27, size=(2000, 26), dtype=np.uint8))
    rx = np.concatenate(rx)
    c = torch.from_numpy(rx).cuda(); dk = torch.zeros((len(rx), k), dtype=torch.uint8, device="cuda"); ok = torch.zeros(len(rx), dtype=torch.uint8, device="cuda")
    gpu.rs_decode_blocks_dev(k, mode, c.data_ptr(), len(rx), dk.data_ptr(), ok.data_ptr())
    torch.cuda.synchronize()
    oc, odk, ook = orc.rs_decode_blocks(k, rx, mode)
    assert np.array_equal(ok.cpu().numpy(), ook)
    assert np.array_equal(c.cpu().numpy(), oc)
    assert np.array_equal(dk.cpu().numpy()[ook == 1], odk[ook == 1])
    if mode == 0:
        n = len(gk["dec_ok"])
        assert list(ok.cpu().numpy()[:n]) == gk["dec_ok"] and list(c.cpu().numpy()[:n].reshape(-1)) == gk["dec_inout"]


@pytest.mark.parametrize("name", sorted(CFGS))
def test_encode_profile_vs_oracle(gpu, orc, name):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cfg, ocfg = both(gpu, CFGS[name])
    for n in (0, 1, 2, 3, 9, 64, 270, 271, 1000, 4099, 70001):
        for canonical in (True, False):
            raw = rng.integers(0, 27 if canonical else 256, size=(n, 9), dtype=np.uint8)
            ok, enc = gpu.encode_profile_from_raw(raw, cfg)
            rc, want = orc.encode_profile(raw, ocfg)
            assert ok and rc == 0 and enc.shape == want.shape, (name, n)
            assert np.array_equal(enc, want), (name, n, canonical, np.flatnonzero(enc.reshape(-1) != want.reshape(-1))[:10])


@pytest.mark.parametrize("name", sorted(CFGS))
def test_encode_frame_vs_oracle(gpu, orc, name):
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 5)
    cfg, ocfg = both(gpu, CFGS[name])
    for n in (0, 1, 2, 5, 64, 539, 540, 541, 2161, 65536, 200003):
        px = rand_pixels(rng, n, in_range=(n % 2 == 0))
        ok, enc = gpu.encode_frame(px, cfg)
        rc, want = orc.encode_frame(px, ocfg, cap=n + 64)
        assert ok and rc == 0 and enc.shape == want.shape, (name, n)
        assert np.array_equal(enc, want), (name, n, np.flatnonzero(enc.reshape(-1) != want.reshape(-1))[:10])


def test_raw_mode_roundtrip_config1(gpu, orc):
    """BASELINE config 1: 256x256 synthetic frame, RAW mode, bit-exact Word27 round trip."""
    px = orc.lcg_pixels(256 * 256)
    ectx = gpu.EncoderContext(); ectx.cfg.profile = gpu.ProfileID.RAW_MODE
    raw = gpu.encode_raw_pixels_to_words(px)
    ok, enc = gpu.encode_profile_from_raw(raw, ectx)
    assert ok and ol.fnv_hex(enc) == GOLD["frames"]["256x256"]["raw_hash"] == "9d2b4253c595d3ff"
    dctx = gpu.DecoderContext(); dctx.cfg_last_seen.profile = gpu.ProfileID.RAW_MODE
    ok, back = gpu.decode_profile_to_raw(enc, dctx)
    assert ok and np.array_equal(back, raw)
    assert np.array_equal(gpu.decode_raw_words_to_pixels(back), px)
    ok, enc2 = gpu.encode_frame(px, ectx)
    assert ok and np.array_equal(enc2, raw)


@pytest.mark.parametrize("res", ["8x8", "256x256", "1920x1080"])
def test_golden_frame_hashes(gpu, orc, res):
    """Frame hashes captured from the unmodified reference (tests/golden/make_golden.py)."""
    w, h = (int(x) for x in res.split("x"))
    px = orc.lcg_pixels(w * h)
    for name, ent in GOLD["frames"][res]["cfgs"].items():
        cfg = ol.cfg_from_dict(ent["cfg"])
        tcfg = gpu.make_cfg(profile=cfg.profile, uep=list(cfg.band_profile), tile=(cfg.tile_w, cfg.tile_h), seed=(cfg.seed_a, cfg.seed_b, cfg.seed_s0),
                            beacon=(cfg.beacon_words_period, cfg.beacon_band_slot, cfg.beacon_enabled))
        ok, enc = gpu.encode_frame(px, tcfg)
        assert ok and len(enc) == ent["out_words"], (res, name)
        assert ol.fnv_hex(enc) == ent["hash"], (res, name)
        assert list(enc[:8].reshape(-1)) == ent["first_words"] and list(enc[-2:].reshape(-1)) == ent["last_words"]
    st = GOLD["selftest64"]
    i = np.arange(64); sp = np.zeros(64, ol.PIXEL_DT); sp["Yq"] = (i * 7) % 243; sp["Cbq"] = (i * 3) % 81 - 40; sp["Crq"] = (i * 5) % 81 - 40
    raw = gpu.encode_raw_pixels_to_words(sp)
    assert list(raw.reshape(-1)) == st["raw_words"]
    ok, enc = gpu.encode_profile_from_raw(raw, gpu.make_cfg(profile=1, uep="luma"))
    assert list(enc.reshape(-1)) == st["p2_luma_words"]


@pytest.mark.parametrize("name", ["p3_uniform20", "p5_tile64_luma", "p2_beacon83"])
def test_8k_frame_hash(gpu, orc, name):
    """BASELINE configs 2 and 3 (and a beacon configuration) at full size: device-resident fused encode, hash captured from
    the reference."""
    import torch
    ent = GOLD["frames"]["7680x4320"]["cfgs"][name]
    px = orc.lcg_pixels(7680 * 4320)
    c = ol.cfg_from_dict(ent["cfg"])
    cfg = gpu.make_cfg(profile=c.profile, uep=list(c.band_profile), tile=(c.tile_w, c.tile_h),
                       beacon=(c.beacon_words_period, c.beacon_band_slot, c.beacon_enabled))
    d_px = torch.from_numpy(px.view(np.uint8)).cuda()
    cap = gpu.encoded_words(len(px) // 2, cfg)
    assert cap == ent["out_words"]
    d_out = torch.zeros(cap * 9, dtype=torch.uint8, device="cuda")
    n = gpu.encode_frame_dev(d_px.data_ptr(), len(px), cfg, d_out.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    enc = d_out.cpu().numpy()
    assert n == ent["out_words"] and ol.fnv_hex(enc) == ent["hash"]
    assert orc.crc32(enc) == ent["crc32"]


@pytest.mark.parametrize("name,kw", [
    ("2-D rows of 7680 symbols (run flow)", dict(profile=4, uep=2, tile=(7680, 8))),
    ("2-D rows of 1024 symbols + luma UEP (run flow, UEP kernel)", dict(profile=4, uep="luma", tile=(1024, 16))),
    ("2-D 64x64, one k (pass flow)", dict(profile=4, uep=2, tile=(64, 64))),
    ("four different k (two UEP launches by pairs of k)", dict(profile=1, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0])),
    ("three different k", dict(profile=1, uep=[0, 1, 1, 3, 0, 1, 3, 3, 0])),
    ("luma UEP + beacon in the stores", dict(profile=1, uep="luma", beacon=(64, 4, 1))),
    ("RS(26,20) + beacon every 2 words, slot 8", dict(profile=2, uep=2, beacon=(2, 8, 1))),
])
def test_4k_frame_round2_paths(gpu, orc, name, kw):
    """The encode paths added in round 2, at a size that runs the persistent tile loop and its tickets for many rounds (3840 x 2160:
    ~3,600 tiles on 768 workgroups): byte-exact against the oracle's CPU restatement in both modes, and the FIXED stream decodes back
    to the pixels through the device entry points (fused decoder with the beacon stepped over, two-kernel decoder with the
    destination-driven de-interleave)."""
    import torch
    NPX = 3840 * 2160
    px = orc.lcg_pixels(NPX, 9001)
    d_px = torch.from_numpy(px.view(np.uint8)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for mode in (0, 1):
        cfg, ocfg = both(gpu, kw, mode)
        cap = gpu.encoded_words(NPX // 2, cfg)
        d_out = torch.zeros(cap * 9 + 64, dtype=torch.uint8, device="cuda")
        n = gpu.encode_frame_dev(d_px.data_ptr(), NPX, cfg, d_out.data_ptr(), cap, s)
        torch.cuda.synchronize()
        rc, want = orc.encode_frame(px, ocfg, cap=NPX // 2 + NPX // 4)
        assert rc == 0 and n == len(want), (name, mode, n, len(want))
        got = d_out[: n * 9].cpu().numpy()
        assert np.array_equal(got, np.asarray(want).reshape(-1)), (name, mode, np.flatnonzero(got != np.asarray(want).reshape(-1))[:8])
        if mode == 1:
            d_back = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
            seen = gpu.default_cfg(); seen.mode = 1
            rcd, nd = gpu.decode_profile_dev(d_out.data_ptr(), n, seen, d_back.data_ptr(), NPX, True, s)
            torch.cuda.synchronize()
            assert rcd == 0 and nd == NPX and torch.equal(d_back[: NPX * 6], d_px[: NPX * 6]), (name, rcd, nd)


@pytest.mark.parametrize("name,kw,max_err", [
    ("C2 RS(26,20) 1-D (fused decoder)", dict(profile=2, uep=2), 3),                                # BASELINE configs[4]
    ("C2 clean", dict(profile=2, uep=2), 0),
    ("C3 2-D 64x64 + luma UEP (two-kernel decoder)", dict(profile=4, uep="luma", tile=(64, 64)), 2),  # BASELINE configs[2]
    ("C3 clean", dict(profile=4, uep="luma", tile=(64, 64)), 0),
    ("RS(26,18) 1-D, t = 4", dict(profile=3, uep=3), 4),
])
def test_8k_fixed_decode_recovers_frame(gpu, orc, name, kw, max_err):
    """BASELINE configs[4] at full size: an 8K FIXED (v6c) stream with 0..t injected symbol errors in EVERY RS block decodes back
    to the exact pixels and to the exact 26-trit words, through both device entry points (synchronous with host-parsed header,
    streaming with the known configuration), verdict words clean.  Size-independent property (encode -> corrupt -> decode =
    identity); the small-size tests pin the same path against the oracle byte for byte."""
    import torch
    NPX = 7680 * 4320
    px = orc.lcg_pixels(NPX, 4711)
    d_px = torch.from_numpy(px.view(np.uint8)).cuda()
    s = torch.cuda.current_stream().cuda_stream
    cfg, _ = both(gpu, kw, mode=1)
    n_raw = NPX // 2; n_enc = gpu.encoded_words(n_raw, cfg)
    coded = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    assert gpu.encode_frame_dev(d_px.data_ptr(), NPX, cfg, coded.data_ptr(), n_enc, s) == n_enc
    L = gpu.plan(n_raw, cfg)
    if max_err:
        clean = coded.clone()
        gpu.inject_errors_dev(coded.data_ptr(), L.header_syms, L.body_syms // 26, 20261004, max_err, s)
        torch.cuda.synchronize()
        n_bad = int((coded != clean).sum().item())
        assert n_bad > (L.body_syms // 26) // 2                       # most blocks really carry errors
        del clean
    out = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
    seen = gpu.DecoderContext(mode=1).cfg_last_seen
    rc, n = gpu.decode_profile_dev(coded.data_ptr(), n_enc, seen, out.data_ptr(), NPX, True, s)
    assert rc == 0 and n == NPX and bool(torch.equal(out[: NPX * 6], d_px)), name
    out.zero_(); ver = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    assert gpu.decode_frame_async(coded.data_ptr(), n_enc, seen, n_raw, out.data_ptr(), NPX, ver.data_ptr(), True, s) == NPX
    torch.cuda.synchronize()
    assert ver.cpu().tolist() == [0, 0] and bool(torch.equal(out[: NPX * 6], d_px)), name
    raw = torch.zeros(n_raw * 9 + 64, dtype=torch.uint8, device="cuda")
    gpu.pack_pixels_dev(d_px.data_ptr(), NPX, raw.data_ptr(), s)
    out.zero_()
    rc, n = gpu.decode_profile_dev(coded.data_ptr(), n_enc, seen, out.data_ptr(), n_raw, False, s)
    assert rc == 0 and n == n_raw and bool(torch.equal(out[: n_raw * 9], raw[: n_raw * 9])), name
    if max_err:   # one block beyond its code's reach -> detected (reference-style false), nothing else disturbed
        flat = coded[: n_enc * 9]
        flat[L.header_syms: L.header_syms + 13] = (flat[L.header_syms: L.header_syms + 13] + 1) % 27
        ver.fill_(7)
        gpu.decode_frame_async(coded.data_ptr(), n_enc, seen, n_raw, out.data_ptr(), NPX, ver.data_ptr(), True, s)
        torch.cuda.synchronize()
        v = ver.cpu().tolist()
        host = flat[L.header_syms: L.header_syms + 26].cpu().numpy()
        k0 = int(L.band_k[0])
        _, _, okb = orc.rs_decode_blocks(k0, ol.oracle().scramble(host, seen.seed_a, seen.seed_b, seen.seed_s0, 1), mode=1)
        assert v[0] == 0 and (v[1] == 0) == bool(okb[0]), (name, v, okb)


def test_decode_compat_golden_streams(gpu, orc):
    for g in GOLD["decode_streams"]:
        s = np.array(g["in_words"], np.uint8).reshape(-1, 9)
        dctx = gpu.DecoderContext()
        ok, out = gpu.decode_profile_to_raw(s, dctx)
        assert ok == g["ok"], g["name"]
        assert list(out.reshape(-1)) == g["out_words"], g["name"]
        seen = dctx.cfg_last_seen.as_dict()
        for key, v in g["seen"].items():
            if key not in ("mode", "superframe_words"):
                assert seen[key] == v, (g["name"], key)
    for g in GOLD["decode_of_encode"]:          # the reference decoder on the reference encoder's output
        c = ol.cfg_from_dict(g["cfg"])
        cfg = gpu.make_cfg(profile=c.profile, uep=list(c.band_profile), tile=(c.tile_w, c.tile_h), beacon=(c.beacon_words_period, c.beacon_band_slot, c.beacon_enabled))
        ok, enc = gpu.encode_frame(orc.lcg_pixels(g["n_px"], g["lcg_seed"]), cfg)
        dctx = gpu.DecoderContext()
        ok, out = gpu.decode_profile_to_raw(enc, dctx)
        assert ok == g["ok"] and len(out) == g["n_out"], g["name"]
        seen = dctx.cfg_last_seen.as_dict()
        for key, v in g["seen"].items():
            if key not in ("mode", "superframe_words"):
                assert seen[key] == v, (g["name"], key)


@pytest.mark.parametrize("name", ["p3_uniform20", "p2_luma", "p5_tile64_luma", "p2_beacon83", "p1_beacon3_slot8", "p5_tile7x5_mixed4"])
def test_decode_compat_vs_oracle(gpu, orc, name):
    from test_oracle_vs_ref import decoder_consistent_stream
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 9)
    kw = dict(CFGS[name])
    if isinstance(kw.get("uep"), list):
        kw["uep"] = [x % 3 for x in kw["uep"]]
    ocfg = ol.make_cfg(**kw)
    for nbw in (0, 1, 25, 26, 27, 130, 600, 5000):
        for corrupt in (0, 1, 5):
            s = decoder_consistent_stream(orc, rng, ocfg, nbw, corrupt)
            dctx = gpu.DecoderContext(); oseen = ol.make_cfg()
            ok, out = gpu.decode_profile_to_raw(s, dctx)
            rc, want = orc.decode_profile(s, oseen)
            assert ok == (rc == 0), (name, nbw, corrupt)
            assert np.array_equal(out, want if rc == 0 else want[:0])
            a, b = dctx.cfg_last_seen.as_dict(), oseen.as_dict()
            assert {k: v for k, v in a.items() if k != "mode"} == {k: v for k, v in b.items() if k != "mode"}
            if rc == 0:
                okp, px = gpu.decode_frame(s, gpu.DecoderContext())
                assert okp and np.array_equal(px, orc.unpack_words(want))
    for n in (0, 3, 5, 6):
        s = rng.integers(0, 27, size=(n, 9), dtype=np.uint8)
        ok, out = gpu.decode_profile_to_raw(s, gpu.DecoderContext())
        assert ok == (orc.decode_profile(s, ol.make_cfg())[0] == 0)


@pytest.mark.parametrize("name", sorted(k for k in CFGS if k != "p5_beacon_slot9"))
def test_fixed_mode_roundtrip_with_errors(gpu, orc, name):
    """BASELINE config 5 semantics at test size: FIXED (v6c) encode -> inject <= t symbol errors per block -> decode ->
    exact recovery; and the HIP FIXED path equals the oracle's restatement of the same spec."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 13)
    cfg, ocfg = both(gpu, CFGS[name], mode=1)
    for n in (0, 1, 2, 64, 541, 4321, 100003):
        px = rand_pixels(rng, n)
        ok, enc = gpu.encode_frame(px, cfg)
        rc, want = orc.encode_frame(px, ocfg, cap=n + 64)
        assert ok and rc == 0 and np.array_equal(enc, want), (name, n)
        L = gpu.plan((n + 1) // 2, cfg)
        kmin = min(L.band_k)
        bad = enc
        if not L.beacon_on:     # body is contiguous 26-symbol blocks: every block gets 0..t errors
            bad = orc.inject_errors(enc, L.header_syms, L.body_syms // 26, 1234 + n, (26 - max(L.band_k)) // 2)
            assert (bad != enc).any() or n < 3
        dctx = gpu.DecoderContext(mode=1)
        ok, back = gpu.decode_frame(bad, dctx)
        assert ok, (name, n)
        padded = np.zeros(2 * ((n + 1) // 2), ol.PIXEL_DT); padded[:n] = px
        assert np.array_equal(back, padded), (name, n)
        oseen = ol.make_cfg(mode=1)
        rc, oback = orc.decode_frame(bad, oseen)
        assert rc == 0 and np.array_equal(oback, back)
        okr, rawback = gpu.decode_profile_to_raw(bad, gpu.DecoderContext(mode=1))      # the same stream to 26-trit words
        assert okr and np.array_equal(np.asarray(rawback).reshape(-1), np.asarray(orc.pack_pixels(padded)).reshape(-1)), (name, n)
        seen = dctx.cfg_last_seen.as_dict(); want_seen = oseen.as_dict()
        assert {k: v for k, v in seen.items()} == {k: v for k, v in want_seen.items()}
        # too many errors in one block -> detected, reference-style `false`
        if not L.beacon_on and L.body_syms >= 26 and kmin >= 20:
            worse = bad.copy().reshape(-1)
            worse[L.header_syms: L.header_syms + 13] = (worse[L.header_syms: L.header_syms + 13] + 1) % 27
            okw, _ = gpu.decode_frame(worse.reshape(-1, 9), gpu.DecoderContext(mode=1))
            rcw, _ = orc.decode_frame(worse.reshape(-1, 9), ol.make_cfg(mode=1))
            assert okw == (rcw == 0)


def test_error_injector_matches_host(gpu, orc):
    import torch
    rng = np.random.default_rng(3)
    w = rng.integers(0, 27, size=(26 * 400 // 9 + 10, 9), dtype=np.uint8)
    d = torch.from_numpy(w).cuda()
    gpu.inject_errors_dev(d.data_ptr(), 52, 390, 99, 3)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), orc.inject_errors(w, 52, 390, 99, 3))


@pytest.mark.parametrize("crc_kernel", ["fp4", "i8", "fp4_blocked"])
def test_frame_record(gpu, orc, crc_kernel, monkeypatch):
    import torch
    if crc_kernel == "i8":
        monkeypatch.setenv("T3HIP_CRC_I8", "1")          # the i8 form of the matrix-core CRC (the FP4 form is the default)
    if crc_kernel == "fp4_blocked":
        monkeypatch.setenv("T3HIP_CRC_BLOCKED", "1")     # consecutive rounds per wave (round 2) instead of strided ones
    rng = np.random.default_rng(4)
    # the last sizes go through the matrix-core rounds; their rests (9 n mod 2048 = 4 .. 2047) are taken by the record kernel itself
    for n in (0, 1, 6, 255, 256, 257, 100000) + tuple(range(14564, 14564 + 228, 19)) + (14564 + 227, 16384, 20766726 // 64, 1000003, 3000001):   # (the FP4 kernel strides its rounds over 16 .. 2048 waves by size)
        w = rng.integers(0, 27, size=(n, 9), dtype=np.uint8)
        d = torch.from_numpy(w).cuda() if n else torch.zeros(16, dtype=torch.uint8, device="cuda")
        # scratch of 64 bytes: two accumulators (fill kernel + atomics); of the size the library asks for: one partial per CRC workgroup
        for nscr in (64, gpu.frame_record_scratch_bytes(n)):
            rec = torch.zeros(gpu.FRAME_RECORD_BYTES, dtype=torch.uint8, device="cuda"); scr = torch.full((nscr,), 0xA5, dtype=torch.uint8, device="cuda")
            gpu.frame_record_dev(d.data_ptr(), n, 7, gpu.make_cfg(profile=2, uep=2), rec.data_ptr(), scr.data_ptr(), nscr)
            torch.cuda.synchronize()
            r = gpu.index_assemble(rec.cpu().numpy(), 100)[0]
            assert (r.frame_idx, r.n_words, r.byte_offset, r.profile) == (7, n, 100, 2)
            assert r.crc32 == orc.crc32(w) and r.sym_sum == orc.sym_sum(w)
            assert list(r.header_syms)[: min(54, 9 * n)] == list(w.reshape(-1)[:54])


# ---- subword trit streams and wire packings -----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N", [27, 24, 21, 18, 15, 7, 1])
def test_subword_streams_match_oracle(t3, orc, gpu, N):
    rng = np.random.default_rng(100 + N)
    for n_words in (0, 1, 3, 4, 5, 257, 4099):
        w = rng.integers(0, 256 if n_words == 257 else 27, (n_words, 9), dtype=np.uint8)     # one case with non-canonical bytes
        tr = t3.extract_subword_stream_from_words(w, N)
        assert np.array_equal(tr, orc.extract_subword_stream(w, N))
        for cut in (0, 1, N // 2):
            src = tr[: max(len(tr) - cut, 0)]
            for fill in (0, 2):
                assert np.array_equal(t3.build_words_from_subword_stream(src, N, fill).reshape(-1),
                                      np.asarray(orc.build_words_from_subword_stream(src, N, fill)).reshape(-1))


@pytest.mark.gpu
def test_base243_and_wire_bytes_match_oracle(t3, orc, gpu):
    rng = np.random.default_rng(7)
    for n in (0, 1, 4, 5, 6, 19, 20, 21, 1000, 65537):
        tr = rng.integers(0, 3, n, dtype=np.uint8)
        b = t3.ut_to_base243(tr)
        assert np.array_equal(b, orc.ut_to_base243(tr))
        back = t3.base243_to_ut(b)
        assert back is not None and np.array_equal(back, tr)
        if n >= 6:      # announced count larger than the payload holds: the reference returns false
            assert t3.base243_to_ut(b[:-1]) is None and orc.base243_to_ut(b[:-1]) is None
    assert t3.base243_to_ut(np.zeros(3, np.uint8)) is None
    raw = rng.integers(0, 256, 9 * 1001, dtype=np.uint8)
    assert np.array_equal(t3.words_to_bytes(raw.reshape(-1, 9)), orc.words_to_bytes(raw.reshape(-1, 9)))
    assert np.array_equal(t3.bytes_to_words(raw).reshape(-1), np.asarray(orc.bytes_to_words(raw)).reshape(-1))
    assert len(t3.bytes_to_words(raw[:-1])) == 0


@pytest.mark.gpu
def test_centring_blits_match_oracle(t3, orc, gpu):
    """blit_center_rgb / extract_center_q (io_image.hpp:125-140, 215-235) against the oracle's restatement (parity unpinned:
    io_image.hpp does not build here): smaller, equal, odd-sized, taller-than-canvas and empty cases, then an 8K canvas on
    the device entry points."""
    import torch
    rng = np.random.default_rng(21)
    for (sw, sh, cw, ch) in [(5, 3, 9, 8), (9, 8, 9, 8), (1, 1, 2, 2), (7, 11, 8, 5), (3, 20, 3, 7), (0, 0, 4, 4), (4, 4, 0, 0), (33, 17, 64, 64), (729, 486, 1024, 512)]:
        src = rng.integers(0, 256, sw * sh * 3, dtype=np.uint8)
        assert np.array_equal(t3.blit_center_rgb(src, sw, sh, cw, ch).reshape(-1), orc.blit_center_rgb(src, sw, sh, cw, ch)), (sw, sh, cw, ch)
    with pytest.raises(t3.T3Error):
        t3.blit_center_rgb(np.zeros(10 * 2 * 3, np.uint8), 10, 2, 8, 8)
    for (fw, fh, sw, sh) in [(9, 8, 5, 3), (9, 8, 9, 8), (8, 5, 7, 11), (3, 7, 3, 20), (64, 64, 33, 17), (4, 4, 0, 0), (1024, 512, 729, 486)]:
        full = rng.integers(0, 65536, fw * fh * 3, dtype=np.uint16)
        got = np.asarray(t3.extract_center_q(full.view(np.uint8), fw, fh, sw, sh)).view(np.uint8).reshape(-1)
        assert np.array_equal(got, orc.extract_center_q(full, fw, fh, sw, sh)), (fw, fh, sw, sh)
    with pytest.raises(t3.T3Error):
        t3.extract_center_q(np.zeros(4 * 4 * 6, np.uint8), 4, 4, 5, 2)
    # 8K canvas: a 6561 x 4320 window (the S24 centred raster is narrower than the frame) blitted and cut back out
    cw, ch, sw, sh = 7680, 4320, 6561, 4000
    src = torch.from_numpy(rng.integers(0, 256, sw * sh * 3, dtype=np.uint8)).cuda()
    canvas = torch.full((cw * ch * 3,), 7, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t3.blit_center_rgb_dev(src.data_ptr(), sw, sh, canvas.data_ptr(), cw, ch, s)
    c3 = canvas.view(ch, cw, 3); x0, y0 = (cw - sw) // 2, (ch - sh) // 2
    assert torch.equal(c3[y0:y0 + sh, x0:x0 + sw].reshape(-1), src)
    assert int(c3[:y0].max()) == 0 and int(c3[y0 + sh:].max()) == 0 and int(c3[:, :x0].max()) == 0 and int(c3[:, x0 + sw:].max()) == 0
    q = torch.from_numpy(rng.integers(0, 256, cw * ch * 6, dtype=np.uint8)).cuda()
    sub = torch.zeros(sw * sh * 6, dtype=torch.uint8, device="cuda")
    t3.extract_center_q_dev(q.data_ptr(), cw, ch, sub.data_ptr(), sw, sh, s)
    torch.cuda.synchronize()
    assert torch.equal(sub.view(sh, sw, 6), q.view(ch, cw, 6)[y0:y0 + sh, x0:x0 + sw])


@pytest.mark.gpu
def test_subword_dev_entry_points_full_frame(t3, orc, gpu):
    """8K-frame sizes through the device entry points: extract -> build is the identity on the kept trits (S24)."""
    import torch
    n_words, N = 16588800, 24
    rng = np.random.default_rng(3)
    w = torch.from_numpy(rng.integers(0, 27, (n_words, 9), dtype=np.uint8)).cuda()
    tr = torch.zeros(n_words * N, dtype=torch.uint8, device="cuda")
    back = torch.zeros((n_words, 9), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t3.subword_extract_dev(w.data_ptr(), n_words, N, tr.data_ptr(), s)
    assert t3.subword_build_dev(tr.data_ptr(), n_words * N, N, 0, back.data_ptr(), n_words, s) == n_words
    torch.cuda.synchronize()
    assert int(tr.max()) <= 2
    assert torch.equal(back[:, :8], w[:, :8]) and torch.equal(back[:, 8], torch.zeros_like(back[:, 8]))   # trits 24..26 dropped
    sample = slice(12345, 12345 + 4096)
    assert np.array_equal(tr[sample.start * N: sample.stop * N].cpu().numpy(), orc.extract_subword_stream(w[sample].cpu().numpy(), N))


@pytest.mark.gpu
@pytest.mark.parametrize("off", [2, 6, 8])
def test_pack_unpack_dev_unaligned(t3, orc, gpu, off):
    """K1 / K5 through the device entry points on buffers that are not 16-byte aligned, odd pixel counts included."""
    import torch
    rng = np.random.default_rng(40 + off)
    s = torch.cuda.current_stream().cuda_stream
    for n in (1, 5, 8, 1001, 40_003):
        px = rand_pixels(rng, n, in_range=(n != 1001))
        nw = (n + 1) // 2
        want = np.asarray(orc.pack_pixels(px)).reshape(-1)
        d_px = torch.zeros(6 * n + 64, dtype=torch.uint8, device="cuda"); d_px[off: off + 6 * n] = torch.from_numpy(px.view(np.uint8).reshape(-1)).cuda()
        d_w = torch.zeros(9 * nw + 64, dtype=torch.uint8, device="cuda")
        t3.pack_pixels_dev(d_px.data_ptr() + off, n, d_w.data_ptr() + off - 1, s)
        torch.cuda.synchronize()
        assert np.array_equal(d_w[off - 1: off - 1 + 9 * nw].cpu().numpy(), want), (n, off)
        assert int(d_w[off - 1 + 9 * nw:].max()) == 0 and (off == 1 or int(d_w[: off - 1].max()) == 0)
        d_b = torch.zeros(12 * nw + 64, dtype=torch.uint8, device="cuda")
        t3.unpack_words_dev(d_w.data_ptr() + off - 1, nw, d_b.data_ptr() + off, s)
        torch.cuda.synchronize()
        assert np.array_equal(d_b[off: off + 12 * nw].cpu().numpy(), np.asarray(orc.unpack_words(orc.pack_pixels(px))).view(np.uint8).reshape(-1)), (n, off)
        assert int(d_b[off + 12 * nw:].max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("off", [1, 2, 6])
def test_subword_dev_entry_points_unaligned(t3, orc, gpu, off):
    """Device buffers that are not 16-byte aligned take the byte-wise load/store branches of the staged kernels."""
    import torch
    rng = np.random.default_rng(off)
    s = torch.cuda.current_stream().cuda_stream
    for N, n_words in ((27, 1500), (24, 513), (15, 77)):
        w = rng.integers(0, 27, (n_words, 9), dtype=np.uint8)
        want = orc.extract_subword_stream(w, N)
        d_w = torch.zeros(9 * n_words + 64, dtype=torch.uint8, device="cuda"); d_w[off: off + 9 * n_words] = torch.from_numpy(w.reshape(-1)).cuda()
        d_t = torch.zeros(N * n_words + 64, dtype=torch.uint8, device="cuda")
        t3.subword_extract_dev(d_w.data_ptr() + off, n_words, N, d_t.data_ptr() + off, s)
        torch.cuda.synchronize()
        assert np.array_equal(d_t[off: off + N * n_words].cpu().numpy(), want)
        assert int(d_t[:off].max()) == 0 and int(d_t[off + N * n_words:].max()) == 0          # nothing written outside
        d_b = torch.zeros(9 * n_words + 64, dtype=torch.uint8, device="cuda")
        assert t3.subword_build_dev(d_t.data_ptr() + off, N * n_words - 5, N, 1, d_b.data_ptr() + off, n_words, s) == n_words
        torch.cuda.synchronize()
        assert np.array_equal(d_b[off: off + 9 * n_words].cpu().numpy(), np.asarray(orc.build_words_from_subword_stream(want[:-5], N, 1)).reshape(-1))
        assert int(d_b[:off].max()) == 0 and int(d_b[off + 9 * n_words:].max()) == 0
        n_tr = N * n_words - 3
        d_p = torch.zeros(n_tr // 5 + 80, dtype=torch.uint8, device="cuda")
        nb = t3.base243_pack_dev(d_t.data_ptr() + off, n_tr, d_p.data_ptr() + off, n_tr // 5 + 16, s)
        torch.cuda.synchronize()
        wantp = orc.ut_to_base243(want[:n_tr])
        assert nb == len(wantp) and np.array_equal(d_p[off: off + nb].cpu().numpy(), wantp)
        d_u = torch.zeros(n_tr + 64, dtype=torch.uint8, device="cuda")
        t3.base243_unpack_dev(d_p.data_ptr() + off, nb, n_tr, d_u.data_ptr() + off, s)
        torch.cuda.synchronize()
        assert np.array_equal(d_u[off: off + n_tr].cpu().numpy(), want[:n_tr]) and int(d_u[off + n_tr:].max()) == 0


# ---- RGB8 <-> quantised YCbCr bridge (parity against the oracle's restatement; the reference header
#      does not compile here, so this row is "parity unpinned" against a reference build) ---------------------------------
@pytest.mark.gpu
def test_rgb_bridge_exhaustive(t3, orc, gpu):
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=1).astype(np.uint8).reshape(-1)
    got = t3.rgb_to_quant_stream(rgb)
    want = orc.rgb_to_quant(rgb)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    names = want.dtype.names
    assert int(want[names[0]].max()) == 242 and int(want[names[1]].min()) >= -40 and int(want[names[2]].max()) <= 40
    for n in (0, 1, 2, 3, 5, 1023):                                   # ragged tails of the 4-pixel lanes
        assert np.array_equal(t3.rgb_to_quant_stream(rgb[: 3 * n]).view(np.uint8), want[:n].view(np.uint8))


@pytest.mark.gpu
def test_rgb_bridge_inverse(t3, orc, gpu):
    Y, Cb, Cr = np.meshgrid(np.arange(243), np.arange(-40, 41), np.arange(-40, 41), indexing="ij")
    px = np.zeros(Y.size + 6, ol.PIXEL_DT); names = px.dtype.names
    px[names[0]][: Y.size] = Y.reshape(-1); px[names[1]][: Y.size] = Cb.reshape(-1); px[names[2]][: Y.size] = Cr.reshape(-1)
    px[names[0]][Y.size:] = [243, 300, 65535, 0, 100, 242]            # out-of-range values: the clamps decide
    px[names[1]][Y.size:] = [-41, 41, -32768, 32767, -100, 40]; px[names[2]][Y.size:] = [41, -41, 32767, -32768, 100, -40]
    assert np.array_equal(t3.quant_stream_to_rgb(px), orc.quant_to_rgb(px))
    for n in (1, 2, 3, 5):
        assert np.array_equal(t3.quant_stream_to_rgb(px[1000: 1000 + n]), orc.quant_to_rgb(px[1000: 1000 + n]))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_beacon_in_the_store_addressing(gpu, orc, mode, monkeypatch):
    """Beacon insertion fused into the encode kernels' stores (OLD:1118-1141): every slot, short and long periods (period 1 keeps
    the separate pass), one k and luma UEP, 1-D and 2-D, pixel counts that end the body before / at / after the last word's beacon
    slot -- against the oracle, and against the separate beacon pass (T3HIP_BEACON_PASS) on the same input."""
    rng = np.random.default_rng(77 + mode)
    case = 0
    for period in (1, 2, 3, 5, 7, 64, 83, 300):
        for slot in range(9):
            case += 1
            uep = [2, 1, 1, 0, 2, 3, 1, 0, 2][case % 9] if case % 3 else [int(x) for x in rng.integers(0, 4, 9)]
            kw = dict(profile=4 if case % 4 == 0 else 1, uep=uep, tile=(int(rng.integers(1, 90)), int(rng.integers(1, 30))) if case % 4 == 0 else (0, 0),
                      beacon=(period, slot, 1), seed=(case, 7 * case + 1, 3))
            cfg, ocfg = both(gpu, kw, mode)
            n = int(rng.integers(1, 12000)) if case % 5 else int(rng.integers(0, 40))
            px = rand_pixels(rng, n)
            ok, enc = gpu.encode_frame(px, cfg)
            rc, want = orc.encode_frame(px, ocfg, cap=n + 64)
            assert ok and rc == 0 and enc.shape == want.shape and np.array_equal(enc, want), (kw, n)
            if case % 6 == 0:
                monkeypatch.setenv("T3HIP_BEACON_PASS", "1")
                ok2, enc2 = gpu.encode_frame(px, cfg)
                monkeypatch.delenv("T3HIP_BEACON_PASS")
                assert ok2 and np.array_equal(enc2, enc), (kw, n)
            if mode == 1:       # ... and stepped over in the fused decoder's loads: up to t errors in every block of the beaconed stream
                L = gpu.plan((n + 1) // 2, cfg)
                flat = enc.reshape(-1).copy(); hs = L.header_syms
                pos = np.arange(L.body_syms_framed)
                body_idx = np.flatnonzero(~((pos >= slot) & ((pos - slot) % (9 * period) == 0)))[: L.body_syms]
                body = np.zeros((L.body_syms + 8) // 9 * 9, np.uint8); body[: L.body_syms] = flat[hs + body_idx]
                hurt = np.asarray(orc.inject_errors(body.reshape(-1, 9), 0, L.body_syms // 26, 99 + case, (26 - max(L.band_k)) // 2)).reshape(-1)
                flat[hs + body_idx] = hurt[: L.body_syms]
                bad = flat.reshape(-1, 9)
                padded = np.zeros(2 * ((n + 1) // 2), ol.PIXEL_DT); padded[:n] = px
                for env in ((None,) if case % 4 else (None, "1")):
                    if env: monkeypatch.setenv("T3HIP_BEACON_PASS", env)
                    okd, back = gpu.decode_frame(bad, gpu.DecoderContext(mode=1))
                    okr, rawback = gpu.decode_profile_to_raw(bad, gpu.DecoderContext(mode=1))
                    if env: monkeypatch.delenv("T3HIP_BEACON_PASS")
                    assert okd and np.array_equal(back, padded), (kw, n, env)
                    assert okr and np.array_equal(np.asarray(rawback).reshape(-1), np.asarray(orc.pack_pixels(padded)).reshape(-1)), (kw, n, env)
                rc, oback = orc.decode_frame(bad, ol.make_cfg(mode=1))
                assert rc == 0 and np.array_equal(oback, padded), (kw, n)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_encode_frame_random_configurations(gpu, orc, mode):
    """Seeded fuzz over the configuration space (all four codes per band, 1-D / 2-D with odd tile shapes, beacons, seeds)
    and over pixel counts around the kernels' tile sizes (a single-k tile is 9 * 55 * k symbols ~ 2285 / 2057 / 2513 / 2742 px)."""
    rng = np.random.default_rng(2024 + mode)
    edge = [2284, 2285, 2286, 2 * 2285 - 1, 3 * 2285 + 1, 2056, 2057, 2513, 2742, 13 * 2285, 25 * 2285 + 7]
    for trial in range(28):
        uniform = trial % 2 == 0
        uep = int(rng.integers(0, 4)) if uniform else [int(x) for x in rng.integers(0, 4, 9)]
        two_d = trial % 3 == 0
        kw = dict(profile=4 if two_d else int(rng.integers(0, 4)), uep=uep,
                  tile=(int(rng.integers(1, 200)), int(rng.integers(1, 40))) if two_d else (0, 0),
                  seed=(int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 32)), int(rng.integers(0, 2 ** 32))))
        if trial % 5 == 4:
            kw["beacon"] = (int(rng.integers(2, 300)), int(rng.integers(0, 9)), 1)
        cfg, ocfg = both(gpu, kw, mode)
        n = int(edge[trial % len(edge)] + (rng.integers(-3, 4) if trial >= len(edge) else 0)) if trial % 4 else int(rng.integers(0, 90000))
        px = rand_pixels(rng, n, in_range=bool(trial % 7))
        ok, enc = gpu.encode_frame(px, cfg)
        rc, want = orc.encode_frame(px, ocfg, cap=n + 64)
        assert ok and rc == 0 and enc.shape == want.shape, (trial, kw, n)
        assert np.array_equal(enc, want), (trial, kw, n, np.flatnonzero(enc.reshape(-1) != want.reshape(-1))[:10])
        if mode == 1:                       # FIXED streams decode back to the pixels (odd counts come back padded by one zero pixel)
            okd, back = gpu.decode_frame(enc, gpu.DecoderContext(mode=1))
            red = orc.unpack_words(orc.pack_pixels(px))          # what survives the trit packing of out-of-range values
            assert okd and np.array_equal(back[: len(red)].view(np.uint8), red.view(np.uint8)), (trial, kw, n)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["p2_luma", "p5_tile64_luma", "p5_tile7x5_mixed4", "p2_beacon83", "p1_beacon3_slot8", "p3_uniform20"])
def test_fixed_decode_paths_agree(gpu, orc, name):
    """The fused / two-kernel FIXED decoders (t3_decode_fused.hip, t3_decode_stream.hip) against the generic gather kernels
    (T3HIP_GENERIC_DECODE=1) on the same corrupted stream, at a size of many tiles: pixels, raw words and the verdict."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 77)
    cfg, ocfg = both(gpu, CFGS[name], mode=1)
    n = 1_000_003
    px = rand_pixels(rng, n)
    ok, enc = gpu.encode_frame(px, cfg); assert ok
    L = gpu.plan((n + 1) // 2, cfg)
    bad = np.ascontiguousarray(enc).copy().reshape(-1)
    if not L.beacon_on:
        bad = np.asarray(orc.inject_errors(enc, L.header_syms, L.body_syms // 26, 99, (26 - max(L.band_k)) // 2)).reshape(-1).copy()
    else:                                      # framed positions: corrupt a sparse set of symbols (at most one per 26)
        idx = L.header_syms + 40 * rng.permutation((len(bad) - L.header_syms) // 40)[:20000]
        bad[idx] = (bad[idx] + 1 + rng.integers(0, 26, len(idx))) % 27
    bad = bad.reshape(-1, 9)
    outs = {}
    for generic in (False, True):
        if generic: os.environ["T3HIP_GENERIC_DECODE"] = "1"
        try:
            okp, back = gpu.decode_frame(bad, gpu.DecoderContext(mode=1))
            okw, raw = gpu.decode_profile_to_raw(bad, gpu.DecoderContext(mode=1))
        finally:
            os.environ.pop("T3HIP_GENERIC_DECODE", None)
        outs[generic] = (okp, np.asarray(back).view(np.uint8).reshape(-1).copy(), okw, np.asarray(raw).reshape(-1).copy())
    assert outs[False][0] == outs[True][0] and outs[False][2] == outs[True][2]
    assert outs[False][0] and np.array_equal(outs[False][1], outs[True][1]) and np.array_equal(outs[False][3], outs[True][3])
    padded = np.zeros(2 * ((n + 1) // 2), ol.PIXEL_DT); padded[:n] = px
    assert np.array_equal(outs[False][1], padded.view(np.uint8).reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["p3_uniform20", "p5_tile64_luma", "p2_beacon83"])
def test_decode_frame_async_streaming(gpu, orc, name):
    """t3hip_decode_frame_async: body decode launched with the stream's known configuration, header symbols checked on the
    device against the header that configuration encodes to.  Same pixels as the synchronous entry; a different header of the
    same length (another scrambler seed), a header that needs its RS correction, and an uncorrectable block are reported."""
    import torch
    rng = np.random.default_rng(31)
    n = 50_001
    px = rand_pixels(rng, n)
    padded = np.zeros(2 * ((n + 1) // 2), ol.PIXEL_DT); padded[:n] = px
    cfg, ocfg = both(gpu, CFGS[name], mode=1)
    kw2 = dict(CFGS[name]); kw2["seed"] = (2, 1, 0)
    cfg2, _ = both(gpu, kw2, mode=1)
    ok, enc = gpu.encode_frame(px, cfg); assert ok
    ok, enc2 = gpu.encode_frame(px, cfg2); assert ok and enc2.shape == enc.shape
    n_raw = (n + 1) // 2; L = gpu.plan(n_raw, cfg)
    flat = np.ascontiguousarray(enc).reshape(-1)
    hurt = flat.copy(); hurt[3] = (hurt[3] + 5) % 27
    dead = flat.copy(); dead[L.header_syms: L.header_syms + 13] = (dead[L.header_syms: L.header_syms + 13] + 1) % 27
    s = torch.cuda.current_stream().cuda_stream
    out = torch.zeros(len(padded) * 6 + 64, dtype=torch.uint8, device="cuda"); ver = torch.full((2,), 7, dtype=torch.int32, device="cuda")
    def run(stream_bytes, to_pixels=True):
        d = torch.from_numpy(np.ascontiguousarray(stream_bytes).reshape(-1)).cuda()
        nu = gpu.decode_frame_async(d.data_ptr(), d.numel() // 9, cfg, n_raw, out.data_ptr(), len(padded), ver.data_ptr(), to_pixels, s)
        torch.cuda.synchronize()
        return nu, ver.cpu().numpy().tolist()
    nu, v = run(flat); assert nu == len(padded) and v == [0, 0]
    assert np.array_equal(out[: len(padded) * 6].cpu().numpy(), padded.view(np.uint8).reshape(-1))
    nu, v = run(flat, to_pixels=False); assert nu == n_raw and v == [0, 0]
    assert np.array_equal(out[: n_raw * 9].cpu().numpy(), np.asarray(orc.pack_pixels(padded)).reshape(-1))
    assert run(np.ascontiguousarray(enc2).reshape(-1))[1][0] == 1
    assert run(hurt)[1][0] == 1
    if not L.beacon_on and min(L.band_k) >= 20:
        okd, _ = gpu.decode_frame(dead.reshape(-1, 9), gpu.DecoderContext(mode=1))
        v = run(dead)[1]; assert v[0] == 0 and (v[1] >= 1) == (not okd)



@pytest.mark.gpu
@pytest.mark.parametrize("npx", [1, 7, 4096, 100_003])
def test_rgb_frame_encode_decode(t3, orc, gpu, npx):
    """Row f1 end to end on the device: RGB8 -> bridge -> fused encode equals the oracle's bridge + encode; the streaming
    decode + inverse bridge gives back what the oracle's quantisation round trip gives (parity of the bridge itself: unpinned)."""
    import torch
    rng = np.random.default_rng(npx)
    rgb = rng.integers(0, 256, size=3 * npx, dtype=np.uint8)
    cfg, ocfg = both(gpu, dict(profile=2, uep=2), mode=1)
    q = orc.rgb_to_quant(rgb)
    rc, want = orc.encode_frame(q, ocfg, cap=npx + 64); assert rc == 0
    s = torch.cuda.current_stream().cuda_stream
    d_rgb = torch.from_numpy(rgb).cuda()
    n_cap = t3.encoded_words((npx + 1) // 2, cfg)
    d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
    n = t3.encode_rgb_dev(d_rgb.data_ptr(), npx, cfg, d_out.data_ptr(), n_cap, s)
    torch.cuda.synchronize()
    assert n == len(want) and np.array_equal(d_out[: 9 * n].cpu().numpy(), np.asarray(want).reshape(-1))
    d_back = torch.zeros(3 * npx + 64, dtype=torch.uint8, device="cuda"); ver = torch.full((2,), 9, dtype=torch.int32, device="cuda")
    t3.decode_rgb_async(d_out.data_ptr(), n, cfg, npx, d_back.data_ptr(), ver.data_ptr(), s)
    torch.cuda.synchronize()
    assert ver.cpu().numpy().tolist() == [0, 0]
    assert np.array_equal(d_back[: 3 * npx].cpu().numpy(), np.asarray(orc.quant_to_rgb(q)).reshape(-1))


@pytest.mark.gpu
def test_rgb_fused_all_colours(t3, orc, gpu):
    """Row f1 fused into the codec kernels, exhaustively: a 4096 x 4096 "frame" holding every one of the 2^24 RGB values goes
    through the encoder's fused front end (3 bytes per pixel in, phase 1 converts) and must equal the oracle's bridge followed by
    its encoder, byte for byte; decoded with RGB out (the decoder's fused output stage) it must equal the oracle's quantisation
    round trip.  (Parity of the bridge itself against the reference: unpinned -- io_image.hpp does not compile, DESIGN.md.)"""
    import torch
    n = 1 << 24
    v = np.arange(n, dtype=np.uint32)
    rgb = np.empty(3 * n, np.uint8); rgb[0::3] = v & 255; rgb[1::3] = (v >> 8) & 255; rgb[2::3] = v >> 16
    rng = np.random.default_rng(5); perm = rng.permutation(n)                  # every colour once, in random order (neighbours differ in all channels)
    rgb = rgb.reshape(n, 3)[perm].reshape(-1).copy()
    cfg, ocfg = both(gpu, dict(profile=2, uep=2), mode=1)
    q = orc.rgb_to_quant(rgb)
    rc, want = orc.encode_frame(q, ocfg, cap=n + 64); assert rc == 0
    s = torch.cuda.current_stream().cuda_stream
    d_rgb = torch.from_numpy(rgb).cuda()
    n_cap = t3.encoded_words(n // 2, cfg)
    d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
    assert t3.encode_rgb_dev(d_rgb.data_ptr(), n, cfg, d_out.data_ptr(), n_cap, s) == len(want)
    torch.cuda.synchronize()
    got = d_out[: 9 * len(want)].cpu().numpy()
    assert np.array_equal(got, np.asarray(want).reshape(-1)), np.flatnonzero(got != np.asarray(want).reshape(-1))[:8]
    d_back = torch.zeros(3 * n + 64, dtype=torch.uint8, device="cuda"); ver = torch.full((2,), 9, dtype=torch.int32, device="cuda")
    t3.decode_rgb_async(d_out.data_ptr(), len(want), cfg, n, d_back.data_ptr(), ver.data_ptr(), s)
    torch.cuda.synchronize()
    assert ver.cpu().numpy().tolist() == [0, 0]
    assert np.array_equal(d_back[: 3 * n].cpu().numpy(), np.asarray(orc.quant_to_rgb(q)).reshape(-1))
    assert int(d_back[3 * n:].sum().item()) == 0                               # nothing written past the frame


@pytest.mark.gpu
@pytest.mark.parametrize("tile", [(16, 3), (48, 5), (128, 2), (512, 4), (16, 1), (528, 3), (2048, 2), (4100, 3), (516, 7), (20000, 2)])
@pytest.mark.parametrize("uep", [2, "luma"])
def test_2d_in_place_flows(t3, orc, gpu, tile, uep):
    """Round 3: the 2-D encode flows that reverse odd rows IN PLACE -- rows of whole 16-byte granules up to 512 symbols (whole rows in the symbol
    buffer), and wider rows that are multiples of 4 (runs placed where they belong, pieces reversed as dword pairs); rows of 516 / 4100
    symbols take the dword flow with ragged chunks, odd chunk heights restart the row parity inside a tile, the stream's last row is
    short.  Pixel counts around tile and chunk edges, both arithmetic modes, against the oracle; then the decoder on the same stream."""
    import torch
    rng = np.random.default_rng(tile[0] * 31 + tile[1])
    s = torch.cuda.current_stream().cuda_stream
    for n in (400_003, 1_200_001):
        px = rand_pixels(rng, n)
        for mode in (0, 1):
            cfg, ocfg = both(gpu, dict(profile=4, uep=uep, tile=tile), mode)
            rc, want = orc.encode_frame(px, ocfg, cap=n); assert rc == 0
            n_cap = t3.encoded_words((n + 1) // 2, cfg)
            d_px = torch.from_numpy(px.view(np.uint8).reshape(-1).copy()).cuda()
            d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
            assert t3.encode_frame_dev(d_px.data_ptr(), n, cfg, d_out.data_ptr(), n_cap, s) == len(want)
            torch.cuda.synchronize()
            got = d_out[: 9 * len(want)].cpu().numpy()
            assert np.array_equal(got, np.asarray(want).reshape(-1)), (tile, uep, n, mode, np.flatnonzero(got != np.asarray(want).reshape(-1))[:8])
            if mode == 1:
                ok, back = gpu.decode_frame(np.asarray(want).reshape(-1, 9), gpu.DecoderContext(mode=1))
                assert ok and np.array_equal(np.asarray(back)[:n], px), (tile, uep, n)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(profile=4, uep=2, tile=(64, 64)), dict(profile=4, uep="luma", tile=(1024, 16)), dict(profile=4, uep=1, tile=(7, 5)),
                                dict(profile=1, uep="luma"), dict(profile=2, uep=2, beacon=(64, 4, 1)), dict(profile=1, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0])])
def test_rgb_fused_other_flows(t3, orc, gpu, kw):
    """The fused RGB front end through the other flows of the encode kernel (2-D with pass / run flow / odd geometry, UEP, beacon in the
    stores, pair launches): 1.5 M random colours, odd pixel count, against the oracle's bridge + encoder; bridge parity unpinned."""
    import torch
    n = 1500001
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, 3 * n, dtype=np.uint8)
    q = orc.rgb_to_quant(rgb)
    s = torch.cuda.current_stream().cuda_stream
    d_rgb = torch.from_numpy(rgb).cuda()
    for mode in (0, 1):
        cfg, ocfg = both(gpu, kw, mode)
        rc, want = orc.encode_frame(q, ocfg, cap=n); assert rc == 0
        n_cap = t3.encoded_words((n + 1) // 2, cfg)
        d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
        assert t3.encode_rgb_dev(d_rgb.data_ptr(), n, cfg, d_out.data_ptr(), n_cap, s) == len(want)
        torch.cuda.synchronize()
        got = d_out[: 9 * len(want)].cpu().numpy()
        assert np.array_equal(got, np.asarray(want).reshape(-1)), (kw, mode, np.flatnonzero(got != np.asarray(want).reshape(-1))[:8])


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(profile=4, uep=2, tile=(64, 64)), dict(profile=4, uep="luma", tile=(1024, 16)), dict(profile=4, uep=1, tile=(7, 5)),
                                dict(profile=1, uep="luma"), dict(profile=2, uep=2, beacon=(64, 4, 1)), dict(profile=1, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0]),
                                dict(profile=4, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0], tile=(640, 3), beacon=(5, 0, 1))])
def test_raw_word_input_other_flows(gpu, orc, kw):
    """encode_profile_from_raw (26-trit words in, OLD:1043) through the flows of the encode kernel that the pixel tests above reach
    with the packer fused: 700 k words (non-canonical trit 26 included), both modes, against the oracle; FIXED decodes back to the
    words."""
    import torch
    n = 700001
    rng = np.random.default_rng(12)
    raw = rng.integers(0, 27, (n, 9), dtype=np.uint8)
    d_raw = torch.from_numpy(raw).cuda()
    s = torch.cuda.current_stream().cuda_stream
    for mode in (0, 1):
        cfg, ocfg = both(gpu, kw, mode)
        rc, want = orc.encode_profile(raw, ocfg, cap=2 * n); assert rc == 0
        n_cap = gpu.encoded_words(n, cfg)
        d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
        assert gpu.encode_profile_dev(d_raw.data_ptr(), n, cfg, d_out.data_ptr(), n_cap, s) == len(want)
        torch.cuda.synchronize()
        got = d_out[: 9 * len(want)].cpu().numpy()
        assert np.array_equal(got, np.asarray(want).reshape(-1)), (kw, mode, np.flatnonzero(got != np.asarray(want).reshape(-1))[:8])
        if mode == 1:
            d_back = torch.zeros(n * 9 + 64, dtype=torch.uint8, device="cuda")
            seen = gpu.default_cfg(); seen.mode = 1
            rcd, nd = gpu.decode_profile_dev(d_out.data_ptr(), len(want), seen, d_back.data_ptr(), n, False, s)
            torch.cuda.synchronize()
            canon = raw.copy(); canon[:, 8] = raw[:, 8] % 9                         # trit 26 of a word is not carried (OLD:1068-1076)
            assert rcd == 0 and nd == n and np.array_equal(d_back[: n * 9].cpu().numpy().reshape(-1, 9), canon), (kw, rcd, nd)


@pytest.mark.gpu
@pytest.mark.parametrize("uep", [0, 1, 2, 3])
@pytest.mark.parametrize("beacon", [(0, 0, 0), (7, 3, 1)])
def test_rgb_out_every_code(t3, orc, gpu, uep, beacon):
    """RGB out of the fused decoder for each of the four codes, with and without a beacon stepped over in its loads, 0..t errors in
    every block (no beacon) / a clean stream (beacon): equals the oracle's dequantisation of the pixels; odd pixel count."""
    import torch
    n = 400001
    rng = np.random.default_rng(20 + uep)
    px = rand_pixels(rng, n)
    cfg = gpu.make_cfg(profile=uep, uep=uep, beacon=beacon, mode=1)
    s = torch.cuda.current_stream().cuda_stream
    d_px = torch.from_numpy(px.view(np.uint8).copy()).cuda()
    n_cap = t3.encoded_words((n + 1) // 2, cfg)
    d_out = torch.zeros(n_cap * 9 + 64, dtype=torch.uint8, device="cuda")
    nw = gpu.encode_frame_dev(d_px.data_ptr(), n, cfg, d_out.data_ptr(), n_cap, s)
    L = gpu.plan((n + 1) // 2, cfg)
    if not beacon[2]:
        gpu.inject_errors_dev(d_out.data_ptr(), L.header_syms, L.body_syms // 26, 31 + uep, (26 - max(L.band_k)) // 2, s)
    d_rgb = torch.full((3 * n + 64,), 0, dtype=torch.uint8, device="cuda"); ver = torch.full((2,), 9, dtype=torch.int32, device="cuda")
    t3.decode_rgb_async(d_out.data_ptr(), nw, cfg, n, d_rgb.data_ptr(), ver.data_ptr(), s)
    torch.cuda.synchronize()
    assert ver.cpu().numpy().tolist() == [0, 0]
    assert np.array_equal(d_rgb[: 3 * n].cpu().numpy(), np.asarray(orc.quant_to_rgb(px)).reshape(-1)), (uep, beacon)
    assert int(d_rgb[3 * n:].sum().item()) == 0


@pytest.mark.gpu
def test_host_api_two_threads(gpu, orc):
    """The std::vector-shaped entry points share one stream and two scratch slots inside the library; two caller threads encoding
    and decoding DIFFERENT frames of different sizes at the same time must each get their own frame's bytes (the library holds
    a host-path mutex over upload -> launch -> download; ctypes drops the GIL, so the calls really overlap)."""
    import threading
    res = {}
    def work(tag, n, seed, kw):
        try:
            cfg, ocfg = both(gpu, kw, mode=1)
            for rep in range(6):
                px = orc.lcg_pixels(n + 17 * rep, seed + rep)
                ok, enc = gpu.encode_frame(px, cfg)
                rc, want = orc.encode_frame(px, ocfg, cap=len(px) + 64)
                assert ok and rc == 0 and np.array_equal(enc, want), (tag, rep, "encode")
                okd, back = gpu.decode_frame(enc, gpu.DecoderContext(mode=1))
                assert okd and np.array_equal(back[: len(px)], px), (tag, rep, "decode")
                raw = gpu.encode_raw_pixels_to_words(px)
                assert np.array_equal(raw, orc.pack_pixels(px)), (tag, rep, "pack")
            res[tag] = "ok"
        except Exception as e:   # noqa: BLE001
            res[tag] = repr(e)
    th = [threading.Thread(target=work, args=("a", 200_003, 11, dict(profile=2, uep=2))),
          threading.Thread(target=work, args=("b", 90_001, 77, dict(profile=4, uep="luma", tile=(64, 64))))]
    for t in th: t.start()
    for t in th: t.join()
    assert res == {"a": "ok", "b": "ok"}, res


@pytest.mark.gpu
def test_device_entry_points_two_streams(t3, orc, gpu):
    """The *_dev entry points of ONE context on two HIP streams from two threads at the same time (and on hipStreamPerThread, one
    handle value that is a different stream per thread): per-stream intermediates and tile-ticket counters, so the frames cannot
    take each other's tiles or scratch.  Large frames (2K x 2K, beacon and two-kernel decode paths included) so that the kernels
    really overlap; every result against the oracle / the round trip."""
    import threading
    import torch
    HIP_STREAM_PER_THREAD = 2
    res = {}
    def work(tag, seed, kw, use_spt):
        try:
            n = 2048 * 2048 + 2 * seed
            st = None if use_spt else torch.cuda.Stream()
            sh = HIP_STREAM_PER_THREAD if use_spt else st.cuda_stream
            cfg, ocfg = both(gpu, kw, mode=1)
            px = orc.lcg_pixels(n, seed)
            rc, want = orc.encode_frame(px, ocfg, cap=n); assert rc == 0
            d_px = torch.from_numpy(px.view(np.uint8).copy()).cuda()
            cap = gpu.encoded_words(n // 2, cfg)
            d_out = torch.zeros(cap * 9 + 64, dtype=torch.uint8, device="cuda"); d_back = torch.zeros(n * 6 + 64, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for rep in range(12):
                d_out.zero_(); d_back.zero_(); torch.cuda.synchronize()
                nw = gpu.encode_frame_dev(d_px.data_ptr(), n, cfg, d_out.data_ptr(), cap, sh)
                seen = gpu.default_cfg(); seen.mode = 1
                rcd, nd = gpu.decode_profile_dev(d_out.data_ptr(), nw, seen, d_back.data_ptr(), n, True, sh)     # synchronises `sh`
                assert nw == len(want) and rcd == 0 and nd == n, (tag, rep, nw, rcd, nd)
                assert np.array_equal(d_out[: 9 * nw].cpu().numpy(), np.asarray(want).reshape(-1)), (tag, rep, "encode")
                assert torch.equal(d_back[: 6 * n], d_px[: 6 * n]), (tag, rep, "decode")
            res[tag] = "ok"
        except Exception as e:   # noqa: BLE001
            res[tag] = repr(e)
    for use_spt in (False, True):
        res.clear()
        th = [threading.Thread(target=work, args=("a", 5, dict(profile=2, uep=2, beacon=(64, 4, 1)), use_spt)),
              threading.Thread(target=work, args=("b", 9, dict(profile=4, uep="luma", tile=(64, 64)), use_spt))]
        for t in th: t.start()
        for t in th: t.join()
        assert res == {"a": "ok", "b": "ok"}, (use_spt, res)


@pytest.mark.gpu
def test_two_contexts_two_threads(t3, orc, gpu):
    """One handle per GPU (t3hip_create / t3hip_use): two contexts (both on device 0 here: the box has one card) driven by two
    threads at the same time, each encoding and decoding its own frames on its own stream, tables and scratch; the default
    context keeps working beside them and after they are destroyed.  t3hip_init for another device while the default exists
    is refused rather than silently rebinding."""
    import threading
    import torch
    res = {}
    def work(tag, n, seed, kw):
        ctx = None
        try:
            ctx = t3.Context(0); ctx.use()
            assert ctx.device() == 0 and t3.is_ready()
            cfg, ocfg = both(gpu, kw, mode=1)
            for rep in range(5):
                px = orc.lcg_pixels(n + 13 * rep, seed + rep)
                ok, enc = gpu.encode_frame(px, cfg)
                rc, want = orc.encode_frame(px, ocfg, cap=len(px) + 64)
                assert ok and rc == 0 and np.array_equal(enc, want), (tag, rep, "encode")
                bad = orc.inject_errors(enc, 90, (len(enc) * 9 - 90) // 26, seed, 2)
                okd, back = gpu.decode_frame(bad, gpu.DecoderContext(mode=1))
                assert okd and np.array_equal(back[: len(px)], px), (tag, rep, "decode")
            res[tag] = "ok"
        except Exception as e:   # noqa: BLE001
            res[tag] = repr(e)
        finally:
            if ctx is not None:
                t3.Context.use_default(); ctx.destroy()
    th = [threading.Thread(target=work, args=("a", 150_001, 5, dict(profile=2, uep=2))),
          threading.Thread(target=work, args=("b", 70_003, 9, dict(profile=1, uep=1)))]
    for t in th: t.start()
    px = orc.lcg_pixels(40_000, 3); cfg, ocfg = both(gpu, dict(profile=2, uep=2), mode=0)       # the default context, meanwhile
    ok, enc = gpu.encode_frame(px, cfg); rc, want = orc.encode_frame(px, ocfg, cap=len(px) + 64)
    for t in th: t.join()
    assert res == {"a": "ok", "b": "ok"}, res
    assert ok and rc == 0 and np.array_equal(enc, want)
    if torch.cuda.device_count() == 1:
        assert t3.lib().t3hip_init(1) != 0                      # no such device / the default context stays where it is
    ok, enc2 = gpu.encode_frame(px, cfg); assert ok and np.array_equal(enc2, want)


@pytest.mark.gpu
def test_shutdown_and_reinit(t3, orc, gpu):
    """t3hip_shutdown frees the device tables of both halves of the library; a new t3hip_init rebuilds them."""
    px = orc.lcg_pixels(5000, 1)
    cfg, _ = both(gpu, dict(profile=2, uep=2), mode=1)
    ok, enc = gpu.encode_frame(px, cfg); assert ok
    crc = t3.crc32(np.arange(300000, dtype=np.uint8))
    assert t3.lib().t3hip_shutdown() == 0
    t3.init(0)
    ok, enc2 = gpu.encode_frame(px, cfg); assert ok and np.array_equal(enc, enc2)
    ok, back = gpu.decode_frame(enc2, gpu.DecoderContext(mode=1)); assert ok and np.array_equal(back[:5000], px)
    assert t3.crc32(np.arange(300000, dtype=np.uint8)) == crc


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(profile=1, uep="luma"), dict(profile=4, uep="luma", tile=(64, 64)), dict(profile=4, uep=2, tile=(64, 64)),
                                dict(profile=4, uep=[3, 1, 1, 3, 1, 1, 3, 1, 1], tile=(128, 3)), dict(profile=4, uep=0, tile=(7680, 8)),
                                dict(profile=4, uep=[2, 0, 2, 0, 2, 0, 2, 0, 2], tile=(20, 20)), dict(profile=4, uep="luma", tile=(4112, 7)),
                                dict(profile=4, uep=1, tile=(12, 1))])
def test_uep_one_launch_decoder(gpu, orc, kw):
    """decode_uep_px_kernel + uep_edge_kernel (round 3: per-band k and / or 2-D in ONE launch, the tile's symbols never leave LDS) against
    the original pixels, the oracle's restatement of the FIXED decoder and the two-kernel path (T3HIP_TWO_KERNEL_DECODE=1) on corrupted
    streams: sizes of under one tile up to many, pixel counts whose last triple / last row / last block are ragged, two codes in either
    order, narrow, odd-height and very wide rows, 0..t errors in every block; one uncorrectable block is reported by both."""
    rng = np.random.default_rng(zlib.crc32(repr(sorted(kw.items())).encode()) & 0xFFFF)
    cfg, ocfg = both(gpu, kw, mode=1)
    for n in (2, 54, 1210, 6 * 4399, 200_004, 600_000 + 6 * 7):
        L = gpu.plan((n + 1) // 2, cfg)
        px = rand_pixels(rng, n)
        ok, enc = gpu.encode_frame(px, cfg); assert ok
        bad = np.asarray(orc.inject_errors(enc, L.header_syms, L.body_syms // 26, 1000 + n % 97, (26 - max(L.band_k)) // 2)).reshape(-1, 9)
        padded = np.zeros(2 * ((n + 1) // 2), ol.PIXEL_DT); padded[:n] = px
        outs = {}
        for two in (False, True):
            if two: os.environ["T3HIP_TWO_KERNEL_DECODE"] = "1"
            try:
                okp, back = gpu.decode_frame(bad, gpu.DecoderContext(mode=1))
            finally:
                os.environ.pop("T3HIP_TWO_KERNEL_DECODE", None)
            outs[two] = (okp, np.asarray(back).view(np.uint8).reshape(-1).copy())
        assert outs[False][0] and outs[True][0], (kw, n)
        assert np.array_equal(outs[False][1], padded.view(np.uint8).reshape(-1)), (kw, n)
        assert np.array_equal(outs[False][1], outs[True][1]), (kw, n)
        if n <= 30_000:
            rc, want = orc.decode_frame(bad, ol.make_cfg(mode=1))
            assert rc == 0 and np.array_equal(np.asarray(want).view(np.uint8).reshape(-1), outs[False][1]), (kw, n)
    # t + 1 errors in one block of the weakest band's code: uncorrectable, or (rarely; always for t = 1) miscorrected -- whichever the code
    # gives, both paths and the oracle's restatement of decode_block's FIXED flavour must agree on the verdict
    n = 60_006
    L = gpu.plan(n // 2, cfg)
    ok, enc = gpu.encode_frame(rand_pixels(rng, n), cfg); assert ok
    flat = np.ascontiguousarray(enc).reshape(-1).copy()
    b = int(np.argmax(L.band_k)); t = (26 - int(L.band_k[b])) // 2
    base = L.header_syms + int(L.band_body_off[b]) + 26 * 17
    for i in range(t + 1):
        flat[base + 2 * i] = (flat[base + 2 * i] + 1 + 3 * i) % 27
    rc, _ = orc.decode_frame(flat.reshape(-1, 9), ol.make_cfg(mode=1))
    verdicts = []
    for two in (False, True):
        if two: os.environ["T3HIP_TWO_KERNEL_DECODE"] = "1"
        try:
            okp, _ = gpu.decode_frame(flat.reshape(-1, 9), gpu.DecoderContext(mode=1))
        finally:
            os.environ.pop("T3HIP_TWO_KERNEL_DECODE", None)
        verdicts.append(bool(okp))
    assert verdicts[0] == verdicts[1] == (rc == 0), (kw, verdicts, rc)      # (short codes: the spheres of radius t cover 40 % of the space for r = 4)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_host_entry_points_pipelined(gpu, orc, mode):
    """The std::vector-shaped encode entry points on frames of many tiles take the pipelined path (round 3: chunks of whole tiles, upload
    of chunk c + 1, kernel on chunk c and download of chunk c - 1 overlap; t3_api.cpp encode_host_pipelined): byte-exact against the oracle
    for pixels and raw words, COMPAT and FIXED, sizes around the chunk and tile edges; the same frame through the serial path
    (T3HIP_SERIAL_HOST=1) gives the same bytes."""
    for k_uep, prof in ((2, 2), (0, 0), (3, 3)):
        cfg, ocfg = both(gpu, dict(profile=prof, uep=k_uep), mode=mode)
        for n in (1_500_001, 26 * 2284 * 5, 700_000):
            px = orc.lcg_pixels(n, 4000 + n % 1000)
            ok, enc = gpu.encode_frame(px, cfg); assert ok
            rc, want = orc.encode_frame(px, ocfg, cap=n + 64)
            assert rc == 0 and enc.shape == want.shape and np.array_equal(enc, want), (k_uep, n, "pixels")
            raw = orc.pack_pixels(px)
            ok, enc2 = gpu.encode_profile_from_raw(raw, cfg); assert ok
            assert np.array_equal(enc2, want), (k_uep, n, "words")
            if n == 700_000:
                os.environ["T3HIP_SERIAL_HOST"] = "1"
                try:
                    ok, enc3 = gpu.encode_frame(px, cfg)
                finally:
                    os.environ.pop("T3HIP_SERIAL_HOST", None)
                assert ok and np.array_equal(enc3, want)
