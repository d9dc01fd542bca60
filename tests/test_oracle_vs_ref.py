"""Pins the C restatement (oracle/t3_oracle.c) to the UNMODIFIED reference compiled in place
(oracle/_ref/libt3ref.so).  Runs only where the ref library exists (this container; the prebuilt .so
also travels to the GPU box).  Seeded; every comparison is bit-exact."""
import zlib

import numpy as np
import pytest

from oracle_lib import PIXEL_DT, make_cfg

pytestmark = pytest.mark.ref

CFGS = {
    "p3_uniform20": dict(profile=2, uep=2),
    "p2_luma": dict(profile=1, uep="luma"),
    "p5_tile64_luma": dict(profile=4, uep="luma", tile=(64, 64)),
    "p5_tile7x5_mixed": dict(profile=4, uep=[0, 1, 2, 3, 0, 1, 2, 3, 1], tile=(7, 5)),
    "p2_beacon": dict(profile=1, uep=1, beacon=(83, 2, 1)),
    "p1_beacon_small": dict(profile=0, uep=0, beacon=(3, 8, 1), seed=(2, 1, 0)),
    "p4_seed_wrap": dict(profile=3, uep=3, seed=(0xFFFFFFFF, 0xFFFFFFFE, 5)),
    "p2_tile_ignored": dict(profile=1, uep=1, tile=(64, 64)),  # tile set but profile != P5 -> no interleave
    "p5_beacon_slot9": dict(profile=4, uep=1, tile=(3, 4), beacon=(5, 9, 1)),  # slot never matches
}


def rand_words(rng, n, canonical=True):
    w = rng.integers(0, 27, size=(n, 9), dtype=np.uint8)
    if canonical:
        w[:, 8] %= 9  # trit 26 = 0, as pack_two_pixels leaves it
    return w


def test_gf_tables(orc, ref):
    a, b = orc.gf_tables(), ref.gf_tables()
    for key in ("exp", "log", "mul", "inv"):
        assert np.array_equal(a[key], b[key]), key
    assert a["prim"] == b["prim"] == 3


@pytest.mark.parametrize("k", [24, 22, 20, 18])
def test_rs_block_level(orc, ref, k):
    rng = np.random.default_rng(100 + k)
    assert np.array_equal(orc.rs_generator(k), ref.rs_generator(k))
    data = rng.integers(0, 27, size=(500, k), dtype=np.uint8)
    data[0] = 0
    data[1] = (np.arange(k) * 5 + 7) % 27  # selftest_rs_unit pattern OLD:1186
    ca, cb = orc.rs_encode_blocks(k, data), ref.rs_encode_blocks(k, data)
    assert np.array_equal(ca, cb)
    # decode_block as a pure function of 26 symbols: reference codewords (clean + corrupted) and random words
    t = (26 - k) // 2
    rx = [ca.copy()]
    for e in range(1, t + 2):
        c = ca.copy()
        for row in c:
            pos = rng.choice(26, size=e, replace=False)
            row[pos] = (row[pos] + rng.integers(1, 27, size=e)) % 27
        rx.append(c)
    rx.append(rng.integers(0, 27, size=(500, 26), dtype=np.uint8))
    rx = np.concatenate(rx)
    oa, ka, fa = orc.rs_decode_blocks(k, rx)
    ob, kb, fb = ref.rs_decode_blocks(k, rx)
    assert np.array_equal(fa, fb)
    assert np.array_equal(oa, ob)
    assert np.array_equal(ka[fa == 1], kb[fb == 1])


@pytest.mark.parametrize("k", [24, 22, 20, 18])
def test_rs_decode_valid_codewords_shows_forney_sign(orc, ref, k):
    """On VALID codewords (FIXED encoder) with e<=t errors the reference finds every position but leaves
    c+2e; the oracle's COMPAT decoder must reproduce exactly that, its FIXED one must fix it."""
    rng = np.random.default_rng(200 + k)
    t = (26 - k) // 2
    data = rng.integers(0, 27, size=(300, k), dtype=np.uint8)
    cw = orc.rs_encode_blocks(k, data, mode=1)
    _, _, ok = ref.rs_decode_blocks(k, cw)
    assert ok.all()
    rx = cw.copy()
    for i, row in enumerate(rx):
        e = 1 + i % t
        pos = rng.choice(26, size=e, replace=False)
        row[pos] = (row[pos] + rng.integers(1, 27, size=e)) % 27
    ob, kb, fb = ref.rs_decode_blocks(k, rx)
    oa, ka, fa = orc.rs_decode_blocks(k, rx, mode=0)
    assert np.array_equal(fa, fb) and np.array_equal(oa, ob) and np.array_equal(ka, kb)
    assert fb.all() and not np.array_equal(ob, cw)          # the reference does NOT recover
    of, kf, ff = orc.rs_decode_blocks(k, rx, mode=1)
    assert ff.all() and np.array_equal(of, cw) and np.array_equal(kf, data)


def test_packer(orc, ref):
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 7, 64, 1001):
        px = np.zeros(n, PIXEL_DT)
        px["Yq"] = rng.integers(0, 243, n); px["Cbq"] = rng.integers(-40, 41, n); px["Crq"] = rng.integers(-40, 41, n)
        wa, wb = orc.pack_pixels(px), ref.pack_pixels(px)
        assert np.array_equal(wa, wb)
        assert np.array_equal(orc.unpack_words(wa), ref.unpack_words(wa))
    # no clamping: any 16-bit value
    px = np.zeros(4096, PIXEL_DT)
    px["Yq"] = rng.integers(0, 65536, 4096); px["Cbq"] = rng.integers(-32768, 32768, 4096); px["Crq"] = rng.integers(-32768, 32768, 4096)
    assert np.array_equal(orc.pack_pixels(px), ref.pack_pixels(px))
    w = rng.integers(0, 256, size=(512, 9), dtype=np.uint8)  # non-canonical symbols
    assert np.array_equal(orc.unpack_words(w), ref.unpack_words(w))


def test_stages(orc, ref):
    rng = np.random.default_rng(4)
    for n, w, h in ((0, 4, 4), (1, 3, 3), (100, 7, 5), (4096, 64, 64), (5000, 64, 64), (777, 1, 9), (778, 9, 1), (50, 100, 3)):
        s = rng.integers(0, 27, n, dtype=np.uint8)
        for inv in (0, 1):
            assert np.array_equal(orc.interleave2d(s, w, h, inv), ref.interleave2d(s, w, h, inv)), (n, w, h, inv)
        assert np.array_equal(orc.interleave2d(orc.interleave2d(s, w, h, 0), w, h, 1), s)
    s = rng.integers(0, 27, 1000, dtype=np.uint8)
    for seed in ((1, 1, 1), (0, 0, 0), (2, 2, 2), (1, 0, 2), (0xFFFFFFFF, 7, 9), (123456789, 0xFFFFFFFF, 4), (3, 1, 1)):
        for inv in (0, 1):
            assert np.array_equal(orc.scramble(s, *seed, inv), ref.scramble(s, *seed, inv)), seed
    for p in (0, 1, 2, 3, 4):
        for f in range(7):
            for hl in range(4):
                assert orc.beacon_symbol(p, f, hl) == ref.beacon_symbol(p, f, hl)
    for n in (0, 1, 12, 69, 200):
        t = rng.integers(0, 3, n, dtype=np.uint8)
        assert np.array_equal(orc.crc12(t), ref.crc12(t))


def test_header(orc, ref):
    rng = np.random.default_rng(5)
    for i in range(200):
        c = make_cfg(profile=int(rng.integers(0, 5)), uep=[int(x) for x in rng.integers(0, 4, 9)],
                     tile=(int(rng.integers(0, 65536)), int(rng.integers(0, 65536))),
                     seed=tuple(int(x) for x in rng.integers(0, 2**32, 3)),
                     beacon=(int(rng.integers(0, 200)), int(rng.integers(0, 12)), int(rng.integers(0, 2))),
                     subword=int(rng.choice([27, 24, 21, 18, 15])), centered=int(rng.integers(0, 2)), coset=int(rng.integers(0, 3)))
        fs, bh = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        a, b = orc.header_pack(c, fs, bh), ref.header_pack(c, fs, bh)
        assert np.array_equal(a, b)
        assert orc.header_check(a) and ref.header_check(a)
        ua, ub = orc.header_unpack(a), ref.header_unpack(a)
        assert ua[0].as_dict() | {"mode": 0, "superframe_words": 0} == ub[0].as_dict() | {"mode": 0, "superframe_words": 0}
        assert ua[1:] == ub[1:]
        bad = a.copy(); bad[int(rng.integers(0, 27))] = (bad[int(rng.integers(0, 27))] + 1) % 27
        assert orc.header_check(bad) == ref.header_check(bad)
    # SURVEY Appendix A header KAT
    c = make_cfg(profile=2, uep="luma", tile=(64, 64), beacon=(83, 2, 1))
    assert list(ref.header_pack(c, 1234, 0)) == [0, 6, 1, 2, 22, 22, 22, 10, 10, 1, 1, 1, 9, 0, 0, 0, 0, 19, 18, 1, 22, 15, 1, 1, 2, 26, 2]


@pytest.mark.parametrize("name", sorted(CFGS))
def test_encode_profile(orc, ref, name):
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    cfg = make_cfg(**CFGS[name])
    for n in (0, 1, 2, 3, 9, 64, 270, 271, 1000, 4099):
        for canonical in (True, False):
            raw = rand_words(rng, n, canonical)
            ra, a = orc.encode_profile(raw, cfg)
            rb, b = ref.encode_profile(raw, cfg)
            assert ra == rb == 0
            assert a.shape == b.shape, (name, n)
            assert np.array_equal(a, b), (name, n)
            assert orc.encoded_words(n, cfg) == len(b)


def test_encode_raw_mode(orc, ref):
    rng = np.random.default_rng(7)
    raw = rand_words(rng, 100)
    cfg = make_cfg(profile=0xFF)
    assert np.array_equal(orc.encode_profile(raw, cfg)[1], ref.encode_profile(raw, cfg)[1])
    assert np.array_equal(orc.decode_profile(raw, make_cfg(profile=0xFF))[1], raw)


def decoder_consistent_stream(orc, rng, cfg, n_body_words, corrupt=0):
    """A stream laid out the way the reference DECODER reads it (OLD:918-993): 6 header words whose two
    26-symbol blocks are valid RS(26,18) codewords of a CRC-correct header, then body words whose slot b
    column is a sequence of valid RS(26,k_b) codewords, scrambled with the decoder's chain."""
    hp = orc.header_pack(cfg, 0, 0)
    A = orc.rs_encode_blocks(18, hp[:18], mode=1)[0]
    B = orc.rs_encode_blocks(18, np.concatenate([hp[18:], np.zeros(9, np.uint8)]), mode=1)[0]
    hdr = np.concatenate([A, B, np.zeros(2, np.uint8)])
    seen, *_ = orc.header_unpack(hp)
    ks = [24, 22, 20, 18]
    body = np.zeros((n_body_words, 9), np.uint8)
    skip = seen.beacon_enabled and seen.beacon_words_period > 0
    for b in range(9):
        k = ks[seen.band_profile[b] % 4]
        rows = [w for w in range(n_body_words) if not (skip and w % seen.beacon_words_period == 0 and b == seen.beacon_band_slot)]
        nblk = len(rows) // 26
        data = rng.integers(0, 27, size=(nblk, k), dtype=np.uint8)
        cw = orc.rs_encode_blocks(k, data, mode=1).reshape(-1)
        col = rng.integers(0, 27, len(rows), dtype=np.uint8)
        col[: len(cw)] = cw
        if corrupt:
            for m in range(nblk):
                if rng.random() < 0.3:
                    e = int(rng.integers(1, corrupt + 1))
                    pos = rng.choice(26, size=e, replace=False)
                    col[26 * m + pos] = (col[26 * m + pos] + rng.integers(1, 27, e)) % 27
        body[rows, b] = col
    flat = orc.scramble(body.reshape(-1), seen.seed_a, seen.seed_b, seen.seed_s0, 0)
    return np.concatenate([hdr, flat]).reshape(-1, 9)


@pytest.mark.parametrize("name", ["p3_uniform20", "p2_luma", "p5_tile64_luma", "p2_beacon", "p1_beacon_small", "p5_tile7x5_mixed"])
def test_decode_profile_decoder_consistent(orc, ref, name):
    """decode_profile_to_raw on streams its framing accepts: clean -> true; with errors -> whatever the
    reference does (c+2e 'corrections', early false on >t) — bit-exact either way, cfg_last_seen included."""
    rng = np.random.default_rng(zlib.crc32(name.encode()) + 1)
    kw = dict(CFGS[name])
    if kw.get("uep") not in ("luma",) and not isinstance(kw.get("uep"), int):
        kw["uep"] = [x % 3 for x in kw["uep"]]  # the header only carries bp%3 (OLD:222-224)
    cfg = make_cfg(**kw)
    for nbw in (0, 1, 25, 26, 27, 130, 600):
        for corrupt in (0, 1, 5):
            s = decoder_consistent_stream(orc, rng, cfg, nbw, corrupt)
            sa, sb = make_cfg(), make_cfg()
            ra, a = orc.decode_profile(s, sa)
            rb, b = ref.decode_profile(s, sb)
            assert (ra == 0) == (rb == 0), (name, nbw, corrupt, ra, rb)
            assert np.array_equal(a, b)
            assert sa.as_dict() == sb.as_dict()
            if corrupt == 0:
                assert rb == 0


def test_decode_profile_on_reference_encoder_output(orc, ref):
    """The reference's decoder on its own encoder's streams (returns false) — same verdict,
    same cfg_last_seen mutation, same (empty) output."""
    rng = np.random.default_rng(11)
    for name in sorted(CFGS):
        cfg = make_cfg(**CFGS[name])
        raw = rand_words(rng, 700)
        _, enc = ref.encode_profile(raw, cfg)
        sa, sb = make_cfg(), make_cfg()
        ra, a = orc.decode_profile(enc, sa)
        rb, b = ref.decode_profile(enc, sb)
        assert (ra == 0) == (rb == 0), name
        assert np.array_equal(a, b) and sa.as_dict() == sb.as_dict()
    for n in (0, 3, 5, 6):  # short inputs
        s = rng.integers(0, 27, size=(n, 9), dtype=np.uint8)
        sa, sb = make_cfg(), make_cfg()
        assert (orc.decode_profile(s, sa)[0] == 0) == (ref.decode_profile(s, sb)[0] == 0)


def test_subword_and_wire_helpers(orc, ref):
    rng = np.random.default_rng(12)
    w = rng.integers(0, 256, size=(100, 9), dtype=np.uint8)
    for N in (27, 24, 21, 18, 15):
        ta, tb = orc.extract_subword_stream(w, N), ref.extract_subword_stream(w, N)
        assert np.array_equal(ta, tb)
        for fill in (0, 1, 2):
            for cut in (0, 1, N, 5 * N + 3):
                assert np.array_equal(orc.build_words_from_subword_stream(ta[: len(ta) - cut], N, fill),
                                      ref.build_words_from_subword_stream(tb[: len(tb) - cut], N, fill))
    for n in (0, 1, 4, 5, 6, 1003):
        t = rng.integers(0, 3, n, dtype=np.uint8)
        ba, bb = orc.ut_to_base243(t), ref.ut_to_base243(t)
        assert np.array_equal(ba, bb)
        assert np.array_equal(orc.base243_to_ut(ba), t) and np.array_equal(ref.base243_to_ut(bb), t)
    assert orc.base243_to_ut(np.zeros(3, np.uint8)) is None and ref.base243_to_ut(np.zeros(3, np.uint8)) is None
    assert np.array_equal(orc.words_to_bytes(w), ref.words_to_bytes(w))
    b = rng.integers(0, 256, 900, dtype=np.uint8)
    assert np.array_equal(orc.bytes_to_words(b), ref.bytes_to_words(b))
    assert len(orc.bytes_to_words(b[:-1])) == len(ref.bytes_to_words(b[:-1])) == 0


def test_reference_selftests_fail_as_surveyed(ref):
    assert ref.selftests() == (False, False)  # RS:FAIL API:FAIL
