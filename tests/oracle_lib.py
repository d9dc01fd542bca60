"""ctypes doors onto the CHECKERS: oracle/libt3oracle.so (plain-C restatement) and, when present,
oracle/_ref/libt3ref.so (the unmodified reference compiled in place).  Test infrastructure only —
nothing under ternary-image-codec_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

PIXEL_DT = np.dtype([("Yq", "<u2"), ("Cbq", "<i2"), ("Crq", "<i2")])


class Cfg(C.Structure):
    """POD mirror of include/t3hip.h::t3_cfg (kept here so the checkers do not depend on the product)."""
    _fields_ = [("profile", C.c_uint8), ("band_profile", C.c_uint8 * 9), ("tile_w", C.c_uint16), ("tile_h", C.c_uint16),
                ("seed_a", C.c_uint32), ("seed_b", C.c_uint32), ("seed_s0", C.c_uint32),
                ("beacon_words_period", C.c_uint32), ("beacon_band_slot", C.c_uint8), ("beacon_enabled", C.c_uint8),
                ("subword", C.c_uint8), ("centered", C.c_uint8), ("superframe_words", C.c_uint32),
                ("coset", C.c_uint8), ("mode", C.c_uint8), ("reserved", C.c_uint8 * 2)]

    def as_dict(self):
        return dict(profile=self.profile, band_profile=list(self.band_profile), tile_w=self.tile_w, tile_h=self.tile_h,
                    seed_a=self.seed_a, seed_b=self.seed_b, seed_s0=self.seed_s0,
                    beacon_words_period=self.beacon_words_period, beacon_band_slot=self.beacon_band_slot,
                    beacon_enabled=self.beacon_enabled, subword=self.subword, centered=self.centered,
                    superframe_words=self.superframe_words, coset=self.coset, mode=self.mode)


def make_cfg(profile=1, uep=1, tile=(0, 0), seed=(1, 1, 1), beacon=(0, 0, 0), superframe_words=8192, subword=27,
             centered=1, coset=0, mode=0):
    """uep: int -> uep_uniform(idx) (OLD:64-67); 'luma' -> uep_luma_priority (OLD:68-72); list of 9 -> verbatim."""
    c = Cfg()
    c.profile = profile
    if uep == "luma":
        bp = [1] * 9
        bp[0] = bp[3] = bp[6] = 2
    elif isinstance(uep, int):
        bp = [uep % 4] * 9
    else:
        bp = list(uep)
    for i in range(9):
        c.band_profile[i] = bp[i]
    c.tile_w, c.tile_h = tile
    c.seed_a, c.seed_b, c.seed_s0 = seed
    c.beacon_words_period, c.beacon_band_slot, c.beacon_enabled = beacon
    c.superframe_words = superframe_words
    c.subword, c.centered, c.coset, c.mode = subword, centered, coset, mode
    return c


def cfg_from_dict(d):
    c = make_cfg(profile=d["profile"], uep=d["band_profile"], tile=(d["tile_w"], d["tile_h"]),
                 seed=(d["seed_a"], d["seed_b"], d["seed_s0"]),
                 beacon=(d["beacon_words_period"], d["beacon_band_slot"], d["beacon_enabled"]),
                 superframe_words=d.get("superframe_words", 8192), subword=d.get("subword", 27),
                 centered=d.get("centered", 1), coset=d.get("coset", 0), mode=d.get("mode", 0))
    return c


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def build_oracle():
    """Compile the checkers (port always; ref only where /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


class _Lib:
    """Uniform wrapper: the same method names drive either library (prefix 't3o_' or 'ref_')."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.path = path

    def _f(self, name, restype=C.c_int):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        return fn

    # ---- field / RS -------------------------------------------------------------------------
    def gf_tables(self):
        e = np.zeros(78, np.uint8); lg = np.zeros(27, np.int16); m = np.zeros(729, np.uint8); iv = np.zeros(27, np.uint8)
        pr = C.c_uint8()
        self._f("gf_tables", None)(_u8p(e), lg.ctypes.data_as(C.POINTER(C.c_int16)), _u8p(m), _u8p(iv), C.byref(pr))
        return dict(exp=e, log=lg, mul=m, inv=iv, prim=pr.value)

    def rs_generator(self, k):
        g = np.zeros(10, np.uint8)
        n = self._f("rs_generator")(C.c_int(k), _u8p(g))
        return g[:n].copy()

    def rs_encode_blocks(self, k, data, mode=0):
        data = np.ascontiguousarray(data, np.uint8).reshape(-1, k)
        out = np.zeros((data.shape[0], 26), np.uint8)
        if self.prefix == "ref_":
            assert mode == 0
            self._f("rs_encode_blocks")(C.c_int(k), _u8p(data), C.c_uint64(data.shape[0]), _u8p(out))
        else:
            self._f("rs_encode_blocks")(C.c_int(k), C.c_int(mode), _u8p(data), C.c_uint64(data.shape[0]), _u8p(out))
        return out

    def rs_decode_blocks(self, k, code, mode=0):
        code = np.ascontiguousarray(code, np.uint8).reshape(-1, 26).copy()
        dk = np.zeros((code.shape[0], k), np.uint8); ok = np.zeros(code.shape[0], np.uint8)
        if self.prefix == "ref_":
            assert mode == 0
            self._f("rs_decode_blocks")(C.c_int(k), _u8p(code), C.c_uint64(code.shape[0]), _u8p(dk), _u8p(ok))
        else:
            self._f("rs_decode_blocks")(C.c_int(k), C.c_int(mode), _u8p(code), C.c_uint64(code.shape[0]), _u8p(dk), _u8p(ok))
        return code, dk, ok

    def rs_parity_matrix(self, k, mode=0):
        P = np.zeros((k, 26 - k), np.uint8)
        self._f("rs_parity_matrix")(C.c_int(k), C.c_int(mode), _u8p(P))
        return P

    # ---- packer -----------------------------------------------------------------------------
    def pack_pixels(self, px):
        px = np.ascontiguousarray(px, PIXEL_DT)
        out = np.zeros(((len(px) + 1) // 2, 9), np.uint8)
        self._f("pack_pixels")(_vp(px), C.c_uint64(len(px)), _vp(out))
        return out

    def unpack_words(self, words):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
        px = np.zeros(2 * len(words), PIXEL_DT)
        self._f("unpack_words")(_vp(words), C.c_uint64(len(words)), _vp(px))
        return px

    # ---- stages -----------------------------------------------------------------------------
    def interleave2d(self, syms, w, h, inverse=0):
        s = np.ascontiguousarray(syms, np.uint8).copy()
        self._f("interleave2d", None)(_u8p(s), C.c_uint64(len(s)), C.c_uint16(w), C.c_uint16(h), C.c_int(inverse))
        return s

    def scramble(self, syms, a, b, s0, inverse=0):
        s = np.ascontiguousarray(syms, np.uint8).copy()
        self._f("scramble", None)(_u8p(s), C.c_uint64(len(s)), C.c_uint32(a), C.c_uint32(b), C.c_uint32(s0), C.c_int(inverse))
        return s

    def beacon_symbol(self, profile, fsm, health):
        return self._f("beacon_symbol", C.c_uint8)(C.c_uint8(profile), C.c_uint16(fsm), C.c_uint8(health))

    def crc12(self, trits):
        t = np.ascontiguousarray(trits, np.uint8); out = np.zeros(12, np.uint8)
        self._f("crc12", None)(_u8p(t), C.c_uint64(len(t)), _u8p(out))
        return out

    def header_pack(self, cfg, frame_seq=0, band_map_hash=0):
        s = np.zeros(27, np.uint8)
        self._f("header_pack", None)(C.byref(cfg), C.c_uint32(frame_seq), C.c_uint32(band_map_hash), _u8p(s))
        return s

    def header_check(self, syms):
        s = np.ascontiguousarray(syms, np.uint8)
        return bool(self._f("header_check")(_u8p(s)))

    def header_unpack(self, syms):
        s = np.ascontiguousarray(syms, np.uint8); c = Cfg(); fs = C.c_uint32(); bh = C.c_uint32(); mg = C.c_uint16(); ver = C.c_uint8()
        self._f("header_unpack", None)(_u8p(s), C.byref(c), C.byref(fs), C.byref(bh), C.byref(mg), C.byref(ver))
        return c, fs.value, bh.value, mg.value, ver.value

    # ---- frame level ------------------------------------------------------------------------
    def encode_profile(self, raw, cfg, cap=None):
        raw = np.ascontiguousarray(raw, np.uint8).reshape(-1, 9)
        if cap is None:
            cap = len(raw) * 2 + 64
        out = np.zeros((cap, 9), np.uint8); n = C.c_uint64()
        rc = self._f("encode_profile")(_vp(raw), C.c_uint64(len(raw)), C.byref(cfg), _vp(out), C.c_uint64(cap), C.byref(n))
        return rc, out[: n.value].copy() if rc == 0 else None

    def decode_profile(self, words, seen, cap=None):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
        if cap is None:
            cap = len(words) + 64
        out = np.zeros((cap, 9), np.uint8); n = C.c_uint64()
        rc = self._f("decode_profile")(_vp(words), C.c_uint64(len(words)), C.byref(seen), _vp(out), C.c_uint64(cap), C.byref(n))
        return rc, out[: n.value].copy()

    def encode_frame(self, px, cfg, cap=None, want_out=True):
        px = np.ascontiguousarray(px, PIXEL_DT)
        if cap is None:
            cap = len(px) + 64
        out = np.zeros((cap, 9), np.uint8); n = C.c_uint64()
        rc = self._f("encode_frame")(_vp(px), C.c_uint64(len(px)), C.byref(cfg), _vp(out), C.c_uint64(cap), C.byref(n))
        return rc, out[: n.value] if rc == 0 else None

    # ---- subword / wire ---------------------------------------------------------------------
    def extract_subword_stream(self, words, N):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
        out = np.zeros(len(words) * 27 + 1, np.uint8)
        n = self._f("extract_subword_stream", C.c_uint64)(_vp(words), C.c_uint64(len(words)), C.c_int(N), _u8p(out))
        return out[:n].copy()

    def build_words_from_subword_stream(self, trits, N, fill=0):
        t = np.ascontiguousarray(trits, np.uint8)
        out = np.zeros((len(t) // max(N, 1) + 2, 9), np.uint8)
        n = self._f("build_words_from_subword_stream", C.c_uint64)(_u8p(t), C.c_uint64(len(t)), C.c_int(N), C.c_uint8(fill), _vp(out))
        return out[:n].copy()

    def ut_to_base243(self, trits):
        t = np.ascontiguousarray(trits, np.uint8); out = np.zeros(len(t) // 5 + 8, np.uint8)
        n = self._f("ut_to_base243", C.c_uint64)(_u8p(t), C.c_uint64(len(t)), _u8p(out))
        return out[:n].copy()

    def blit_center_rgb(self, src, sw, sh, cw, ch):                       # io_image.hpp:125-140
        s_ = np.ascontiguousarray(src, np.uint8).reshape(-1); out = np.zeros(cw * ch * 3, np.uint8)
        self._f("blit_center_rgb", None)(_u8p(s_), C.c_int(sw), C.c_int(sh), _u8p(out), C.c_int(cw), C.c_int(ch))
        return out

    def extract_center_q(self, full, fw, fh, sw, sh):                     # io_image.hpp:215-235
        f_ = np.ascontiguousarray(full).view(np.uint8).reshape(-1); out = np.zeros(sw * sh * 6, np.uint8)
        self._f("extract_center_q", None)(_vp(f_), C.c_int(fw), C.c_int(fh), _vp(out), C.c_int(sw), C.c_int(sh))
        return out

    def base243_to_ut(self, b):
        b = np.ascontiguousarray(b, np.uint8); out = np.zeros(len(b) * 5 + 8, np.uint8)
        n = self._f("base243_to_ut", C.c_int64)(_u8p(b), C.c_uint64(len(b)), _u8p(out))
        return None if n < 0 else out[:n].copy()

    def words_to_bytes(self, words):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9); out = np.zeros(words.size, np.uint8)
        self._f("words_to_bytes", None)(_vp(words), C.c_uint64(len(words)), _u8p(out))
        return out

    def bytes_to_words(self, b):
        b = np.ascontiguousarray(b, np.uint8); out = np.zeros((len(b) // 9 + 1, 9), np.uint8)
        n = self._f("bytes_to_words", C.c_uint64)(_u8p(b), C.c_uint64(len(b)), _vp(out))
        return out[:n].copy()


class Oracle(_Lib):
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "libt3oracle.so")
        if not os.path.exists(path):
            build_oracle()
        super().__init__(path, "t3o_")

    def regroup(self, words):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9); out = np.zeros(words.size + 1, np.uint8)
        n = self._f("regroup", C.c_uint64)(_u8p(words), C.c_uint64(len(words)), _u8p(out))
        return out[:n].copy()

    def encoded_words(self, n_raw, cfg):
        return self._f("encoded_words", C.c_uint64)(C.c_uint64(n_raw), C.byref(cfg))

    def decode_frame(self, words, seen, cap_px=None):
        words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
        if cap_px is None:
            cap_px = 2 * len(words) + 64
        px = np.zeros(cap_px, PIXEL_DT); n = C.c_uint64()
        rc = self._f("decode_frame")(_vp(words), C.c_uint64(len(words)), C.byref(seen), _vp(px), C.c_uint64(cap_px), C.byref(n))
        return rc, px[: n.value].copy() if rc == 0 else None

    def fnv1a64(self, a):
        a = np.ascontiguousarray(a)
        return self._f("fnv1a64", C.c_uint64)(_vp(a), C.c_uint64(a.nbytes))

    def crc32(self, a):
        a = np.ascontiguousarray(a)
        return self._f("crc32", C.c_uint32)(_vp(a), C.c_uint64(a.nbytes))

    def sym_sum(self, a):
        a = np.ascontiguousarray(a)
        return self._f("sym_sum", C.c_uint32)(_vp(a), C.c_uint64(a.nbytes))

    def lcg_pixels(self, n_px, seed=12345):
        px = np.zeros(n_px, PIXEL_DT)
        self._f("lcg_pixels", None)(_vp(px), C.c_uint64(n_px), C.c_uint32(seed))
        return px

    # ---- row f1: RGB8 <-> quantised YCbCr bridge (old/include/io_image.hpp:156-195) ------------------------
    def rgb_to_quant(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1)
        px = np.zeros(len(rgb) // 3, PIXEL_DT)
        self._f("rgb_to_quant", None)(_u8p(rgb), C.c_uint64(len(px)), _vp(px))
        return px

    def quant_to_rgb(self, px):
        px = np.ascontiguousarray(px)
        rgb = np.zeros(3 * len(px), np.uint8)
        self._f("quant_to_rgb", None)(_vp(px), C.c_uint64(len(px)), _u8p(rgb))
        return rgb

    def lcg_rgb(self, n_px, seed=12345):
        rgb = np.zeros(3 * n_px, np.uint8)
        self._f("lcg_rgb", None)(_u8p(rgb), C.c_uint64(n_px), C.c_uint32(seed))
        return rgb

    def inject_errors(self, words, first_sym, n_blocks, seed, max_err):
        w = np.ascontiguousarray(words, np.uint8).copy()
        self._f("inject_errors", None)(_vp(w), C.c_uint64(first_sym), C.c_uint64(n_blocks), C.c_uint32(seed), C.c_int(max_err))
        return w


def ref_path():
    return os.path.join(ORACLE_DIR, "_ref", "libt3ref.so")


def have_ref():
    return os.path.exists(ref_path())


class Ref(_Lib):
    """The unmodified reference (COMPAT behaviour only)."""

    def __init__(self):
        super().__init__(ref_path(), "ref_")

    def selftests(self):
        return bool(self._f("selftest_rs_unit")()), bool(self._f("selftest_api_roundtrip")())


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def fnv_hex(a):
    return "%016x" % oracle().fnv1a64(a)
