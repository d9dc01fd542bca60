"""CPU-side checks (no GPU): the product's host-only metadata (field tables, generators, parity matrices, header codec,
stream layout) against the oracle and the golden vectors captured from the reference; the C-ABI library loads and exports
every symbol include/t3hip.h declares; compute entry points refuse to run without a device (no CPU fallback)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


def to_t3(t3, c):
    return t3.make_cfg(profile=c.profile, uep=list(c.band_profile), tile=(c.tile_w, c.tile_h), seed=(c.seed_a, c.seed_b, c.seed_s0),
                       beacon=(c.beacon_words_period, c.beacon_band_slot, c.beacon_enabled), superframe_words=c.superframe_words,
                       subword=c.subword, centered=c.centered, coset=c.coset, mode=c.mode)


def test_abi_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, "include", "t3hip.h")).read()
    names = sorted(set(re.findall(r"\b(t3hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    lib = built.lib()
    for n in names:
        assert hasattr(lib, n), "libt3hip.so does not export " + n
    assert C.sizeof(built.Cfg) == 44 and C.sizeof(built.FrameRecord) == 96
    assert C.sizeof(ol.Cfg) == C.sizeof(built.Cfg)


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    px = np.zeros(4, built.PIXEL_DT)
    with pytest.raises(built.T3Error) as e:
        built.encode_raw_pixels_to_words(px)
    assert e.value.code == built.E_NODEVICE
    with pytest.raises(built.T3Error):
        built.encode_frame(px, built.default_cfg())
    with pytest.raises(built.T3Error):
        built.init(0)


def test_field_and_rs_constants(built, orc):
    t, g = built.gf27_tables(), GOLD["gf"]
    assert list(t["exp"]) == g["exp"] and list(t["log"]) == g["log"] and list(t["mul"]) == g["mul"] and list(t["inv"]) == g["inv"]
    for k in (24, 22, 20, 18):
        gk = GOLD["rs"][str(k)]
        assert list(built.rs_generator(k)) == gk["g"]
        assert list(built.rs_parity_matrix(k, 0).reshape(-1)) == gk["P_compat"]
        assert np.array_equal(built.rs_parity_matrix(k, 1), orc.rs_parity_matrix(k, 1))
        # FIXED codewords have zero syndromes under the decoder's convention: the reference decoder accepts them unchanged
        d = np.arange(k, dtype=np.uint8) % 27
        par = np.zeros(26 - k, np.uint8)
        mul, exp = t["mul"], t["exp"]
        P = built.rs_parity_matrix(k, 1)
        for j in range(26 - k):
            acc = 0
            for i in range(k):
                acc = int(orc.lib.t3o_gf_add(C.c_uint8(acc), C.c_uint8(int(mul[int(d[i]) * 27 + int(P[i, j])]))))
            par[j] = acc
        cw = np.concatenate([d, par])
        _, _, ok = orc.rs_decode_blocks(k, cw, mode=0)
        assert ok[0] == 1


def test_header_codec(built, orc):
    for h in GOLD["header"]:
        c = built.make_cfg(**{k: v for k, v in dict(profile=h["cfg"]["profile"], uep=h["cfg"]["band_profile"], tile=(h["cfg"]["tile_w"], h["cfg"]["tile_h"]),
                                                    seed=(h["cfg"]["seed_a"], h["cfg"]["seed_b"], h["cfg"]["seed_s0"]),
                                                    beacon=(h["cfg"]["beacon_words_period"], h["cfg"]["beacon_band_slot"], h["cfg"]["beacon_enabled"]),
                                                    subword=h["cfg"]["subword"], centered=h["cfg"]["centered"], coset=h["cfg"]["coset"]).items()})
        s = built.header_pack(c, h["frame_seq"], h["band_map_hash"])
        assert list(s) == h["syms"]
        assert built.header_check(s) == h["check"]
        u, fs, bh = built.header_unpack(s)
        want = h["unpacked"]
        got = u.as_dict()
        for key in ("profile", "band_profile", "tile_w", "tile_h", "seed_a", "seed_b", "seed_s0", "beacon_words_period", "beacon_band_slot", "beacon_enabled", "subword", "centered", "coset"):
            assert got[key] == want[key], key
        assert (fs, bh) == (h["u_frame_seq"], h["u_band_map_hash"])
    rng = np.random.default_rng(1)
    for _ in range(100):
        oc = ol.make_cfg(profile=int(rng.integers(0, 5)), uep=[int(x) for x in rng.integers(0, 4, 9)], tile=(int(rng.integers(0, 65536)), int(rng.integers(0, 65536))),
                         seed=tuple(int(x) for x in rng.integers(0, 2**32, 3)), beacon=(int(rng.integers(0, 200)), int(rng.integers(0, 12)), int(rng.integers(0, 2))),
                         subword=int(rng.choice([27, 24, 21, 18, 15])), centered=int(rng.integers(0, 2)), coset=int(rng.integers(0, 3)))
        fs, bh = int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32))
        s = built.header_pack(to_t3(built, oc), fs, bh)
        assert np.array_equal(s, orc.header_pack(oc, fs, bh))
        bad = s.copy(); bad[int(rng.integers(0, 27))] = (int(bad[int(rng.integers(0, 27))]) + 1) % 27
        assert built.header_check(bad) == orc.header_check(bad)


@pytest.mark.parametrize("mode", [0, 1])
def test_layout_and_coded_header(built, orc, mode):
    rng = np.random.default_rng(2 + mode)
    for _ in range(300):
        oc = ol.make_cfg(profile=int(rng.integers(0, 5)), uep=[int(x) for x in rng.integers(0, 4, 9)], tile=(int(rng.integers(0, 100)), int(rng.integers(0, 100))),
                         seed=tuple(int(x) for x in rng.integers(0, 2**32, 3)), beacon=(int(rng.integers(0, 40)), int(rng.integers(0, 9 if mode else 12)), int(rng.integers(0, 2))), mode=mode)
        n = int(rng.integers(0, 5000))
        assert built.encoded_words(n, to_t3(built, oc)) == orc.encoded_words(n, oc)
        # the coded header the encoder emits = the first header_syms bytes of the oracle's stream
        raw = rng.integers(0, 27, size=(min(n, 40), 9), dtype=np.uint8)
        rc, enc = orc.encode_profile(raw, oc)
        assert rc == 0
        hs = built.header_encode(to_t3(built, oc), len(raw))
        assert len(hs) == (90 if mode else 52)
        assert np.array_equal(hs, enc.reshape(-1)[: len(hs)])
    L = built.plan(16588800, built.make_cfg(profile=2, uep=2))      # BASELINE C2
    assert (L.n_sym, L.body_syms, L.out_words) == (143769600, 186900480, 20766726) and list(L.band_blocks) == [798720] * 9
    L = built.plan(16588800, built.make_cfg(profile=4, uep="luma", tile=(64, 64)))   # C3
    assert L.out_words == 19508136 and L.interleave2d == 1
    assert built.encoded_words(123, built.make_cfg(profile=0xFF)) == 123


def test_selftest_inputs_known_answers(orc):
    """The reference's own self-test inputs (OLD:1186, 1210-1223) with outputs captured from it."""
    for k in (24, 22, 20, 18):
        pat = ((np.arange(k) * 5 + 7) % 27).astype(np.uint8)
        assert list(orc.rs_encode_blocks(k, pat)[0, k:]) == GOLD["rs"][str(k)]["pattern_parity"]
    assert GOLD["selftests"] == [False, False]


def test_device_division_tricks_are_exact():
    """The kernels divide by powers of three with 24-bit multiplies and reduce 16-bit components with a float reciprocal
    (csrc/t3_kernels.hip: div3/div9/div27/div81, red_y, red_c).  Exhaustive check of those formulas over their domains."""
    x = np.arange(512, dtype=np.uint64)
    assert ((x * 171) >> 9 == x // 3).all() and ((x * 228) >> 11 == x // 9).all() and ((x * 152) >> 12 == x // 27).all()
    x = np.arange(885, dtype=np.uint64)
    assert ((x * 405) >> 15 == x // 81).all()
    y = np.arange(65536, dtype=np.uint32)
    for d in (243, 81):
        q = ((y.astype(np.float32) + np.float32(0.5)) * (np.float32(1.0) / np.float32(d))).astype(np.uint32)
        assert (q == y // d).all()
    c = np.arange(-32768, 32768, dtype=np.int64)
    v = c + 40
    want = (v % (1 << 32)) % 81                          # (uint32_t)(Cbq + 40) % 81, as i2tr sees it (OLD:698)
    xx = np.where(v < 0, v + 32854, v)
    assert (xx >= 0).all() and (xx < 65536).all() and (xx % 81 == want).all()


def _sb(x):
    """signed value of a byte"""
    return x - 256 if x >= 128 else x


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("k", [24, 22, 20, 18])
def test_mfma_encode_tables_reproduce_the_parity_matrix(t3, orc, k, mode):
    """CPU emulation of the matrix-core parity path from the host-built tables alone: the A operand times the
    trit bytes of a block, folded through the M tables, must equal RS parity as the oracle computes it, scrambler included."""
    r = 26 - k; H = r // 2
    af = np.zeros(768, np.uint32); img = np.zeros(3168, np.uint32)
    assert t3.lib().t3hip_mfma_encode_tables(k, mode, af.ctypes.data_as(C.c_void_p), img.ctypes.data_as(C.c_void_p)) == 0
    A = af.reshape(3, 64, 4)                                      # [step][lane][dword]
    T = img[: 3 * 1024].reshape(3, 32, 32)[:, :27, 0]             # [state][symbol] (bank copy 0; all copies equal)
    assert np.array_equal(img[: 3 * 1024].reshape(3, 32, 32)[:, :27, :], np.repeat(T[:, :, None], 32, axis=2))
    M = img[3 * 1024:].view(np.uint8).reshape(3, 128)
    rng = np.random.default_rng(k + mode)
    data = rng.integers(0, 27, (64, k), dtype=np.uint8)
    want = orc.rs_encode_blocks(k, data, mode)[:, k:]             # unscrambled parity symbols
    scr_pos = 22 if k == 22 else 20
    for blk in range(64):
        st = rng.integers(0, 3, 8)                                # arbitrary scrambler states of the parity symbols
        acc = np.full((2, 16), 64, np.int64)                      # [lane half hh][accumulator register i]
        for s in range(3):
            for kh in range(2):
                for d in range(4):
                    p = 8 * s + 4 * kh + d
                    if p < k:
                        bdw = int(T[0][data[blk, p]])             # state 0: the symbol's own trits
                    elif r >= 4 and p in (scr_pos, scr_pos + 1):
                        j0 = 4 * (p - scr_pos); bdw = sum(int(st[j0 + b]) << (8 * b) for b in range(4) if j0 + b < r)
                    else:
                        bdw = 0
                    bb = [_sb((bdw >> (8 * q)) & 0xFF) for q in range(4)]
                    for m in range(32):                            # A lane (row m, K-half kh)
                        adw = int(A[s][m + 32 * kh][d]); hh = (m >> 2) & 1; i = (m & 3) + 4 * (m >> 3)
                        acc[hh][i] += sum(_sb((adw >> (8 * q)) & 0xFF) * bb[q] for q in range(4))
        for j in range(r):
            hh, jj = divmod(j, H)
            x = [int(acc[hh][3 * jj + t]) + (int(st[j]) if r == 2 else 0) for t in range(3)]
            assert all(0 <= v < 128 for v in x)
            sym = int(M[0][x[0]]) + int(M[1][x[1]]) + int(M[2][x[2]])
            tr = [(int(want[blk, j]) // 3 ** t) % 3 for t in range(3)]
            assert sym == sum(((tr[t] + int(st[j])) % 3) * 3 ** t for t in range(3)), (k, mode, blk, j)
    for v in range(3):                                             # T: signed trits + scrambled image
        for d in range(27):
            e = int(T[v][d]); tr = [d % 3, (d // 3) % 3, d // 9]
            assert [_sb((e >> (8 * q)) & 0xFF) for q in range(3)] == [(-1 if t == 2 else t) for t in tr]
            assert e >> 24 == sum(((tr[t] + v) % 3) * 3 ** t for t in range(3))


def _build_hostnames(tmp, extra=()):
    import subprocess
    exe = os.path.join(tmp, "hostnames_demo" + ("_ref" if extra else ""))
    lib = os.path.join(ROOT, "ternary-image-codec_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", *extra, "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "hostnames_demo.cpp"), "-L" + lib, "-lt3hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_device_free_reference_names(built, tmp_path):
    """pack_two_pixels / unpack_two_pixels are host integer arithmetic in the drop-in header (OLD:693-722): golden lcg5 / quirk vectors
    captured from the reference, without a device.  RSCodec one block at a time and the context members the reference's callers can
    reach (OLD:885-916) work on the host too; selftest_rs_unit returns true in FIXED arithmetic and, compiled with
    -DT3_SELFTEST_REFERENCE_ARITHMETIC, the reference's own verdict (golden "selftests"[0] = false)."""
    import subprocess
    orc = ol.oracle()
    P = GOLD["pack"]
    px = P["lcg5_px"] + [[0, 0, 0]] + P["quirk_px"]                                  # (odd count: the reference pads a zero pixel, OLD:730)
    words = P["lcg5_words"] + P["quirk_words"]
    stdin = "%d %s %d %s\n" % (len(px), " ".join(str(v) for p in px for v in p), len(words) // 9, " ".join(str(v) for v in words))
    want_px = [v for p in (P["lcg5_px"] + [[0, 0, 0]] + P["quirk_unpacked"]) for v in p]
    data = np.array([(i * 5 + 7) % 27 for i in range(20)], np.uint8)
    for extra, verdict in (((), 1), (("-DT3_SELFTEST_REFERENCE_ARITHMETIC",), int(GOLD["selftests"][0]))):
        out = json.loads(subprocess.run([_build_hostnames(str(tmp_path), extra)], input=stdin, check=True, capture_output=True, text=True).stdout)
        assert out["words"] == words and out["pixels"] == want_px
        assert out["compat_code"] == [int(x) for x in orc.rs_encode_blocks(20, data, 0)[0]]
        assert out["fixed_recovered"] == 1 and out["hdr_k"] == 18 and out["p1_k"] == 24 and out["p4_g"] == 9 and out["own_gf"] == 1
        assert out["gf_mul"] == int(orc.gf_tables()["mul"][5 * 27 + 7])
        assert out["selftest_rs_unit"] == verdict and out["reference_arithmetic"] == (1 if extra else 0)


def test_host_block_codec_matches_oracle(built):
    """t3hip_rs_encode_block_host / t3hip_rs_decode_block_host (what RSCodec binds) against the oracle's block codec, which is pinned to
    the reference: every k, both arithmetics, 0..t+1 symbol errors, verdicts and bytes."""
    orc = ol.oracle(); lib = built.lib()
    lib.t3hip_rs_encode_block_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.t3hip_rs_decode_block_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    for k in (24, 22, 20, 18):
        t = (26 - k) // 2
        for mode in (0, 1):
            for trial in range(40):
                data = rng.integers(0, 27, k, dtype=np.uint8)
                code = np.zeros(26, np.uint8)
                assert lib.t3hip_rs_encode_block_host(k, mode, data.ctypes.data, code.ctypes.data) == 0
                assert np.array_equal(code, orc.rs_encode_blocks(k, data, mode)[0])
                bad = code.copy()
                for p in rng.choice(26, trial % (t + 2), replace=False):
                    bad[p] = (bad[p] + int(rng.integers(1, 27))) % 27
                cw, dk, okv = orc.rs_decode_blocks(k, bad, mode)
                mine = bad.copy(); outk = np.full(k, 77, np.uint8)
                rc = lib.t3hip_rs_decode_block_host(k, mode, mine.ctypes.data, outk.ctypes.data)
                assert rc == int(okv[0]) and np.array_equal(mine, cw[0])
                assert np.array_equal(outk, dk[0]) if rc else np.all(outk == 77)
    assert lib.t3hip_rs_encode_block_host(21, 0, data.ctypes.data, code.ctypes.data) == -3


def test_no_vgpr_spills_in_hot_kernels(built):
    """Register budget of the gfx950 code objects, read from the built objects' notes (hipcc cross-compiles without a GPU):
    the kernels of the bench step and every 1-D fused encoder (any input, any code) carry NO spilled VGPR, and none of them reloads a
    spilled register inside its persistent tile loop (a scratch reload is
    followed by s_waitcnt vmcnt(0), which drains the next tile's prefetch in the middle of a phase: profiles/r02/notes.md).
    Every instantiation's figures: `python3 profiles/kernel_resources.py`."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import kernel_resources as kr
    ks = kr.all_kernels()
    assert len(ks) > 100
    # every 1-D fused encoder (pixels / raw words / RGB in, the four codes, with and without the fused beacon), the 2-D ones of the headline
    # code without beacon, the UEP kernels, the bench step's decoder and CRC kernels
    must_be_clean = ["encode_kernel_k<%d, 0, %d, %s>" % (fe, r, b) for fe in (0, 1, 2) for r in (2, 4, 6, 8) for b in ("false", "true")]
    must_be_clean += ["encode_kernel_k<%d, %d, 6, false>" % (fe, il) for fe in (0, 2) for il in (1, 2)] + ["encode_kernel_k<1, 1, 6, false>"]
    must_be_clean += ["encode_kernel_uep<0, 0, false>", "encode_kernel_uep<2, 0, false>", "encode_kernel_uep<0, 1, false>", "encode_kernel_uep<2, 1, false>",
                      "decode_fixed_px_kernel<6, false, false>", "decode_fixed_px_kernel<6, true, false>", "crc_fp4_kernel", "crc_mfma_kernel"]
    for want in must_be_clean:
        hit = [n for n in ks if want in n]
        assert hit, want
        for n in hit:
            assert int(ks[n]["vgpr_spill_count"]) == 0, (n, ks[n])
            assert int(ks[n]["vgpr_count"]) <= 128
    loops = kr.loop_scratch()
    assert len(loops) > 60
    for n, (ld, st) in loops.items():
        if any(w in n for w in must_be_clean):
            assert (ld, st) == (0, 0), (n, ld, st)
