"""ternary-image-codec_amd — host-side Python mirror of the reference's Word27 API over libt3hip.so (HIP, gfx950).

The names follow the reference (old/include/ternary_image_codec_v6_min.hpp = OLD): ProfileID, UEPLayout helpers,
EncoderContext / DecoderContext, encode_raw_pixels_to_words, decode_raw_words_to_pixels, encode_profile_from_raw,
decode_profile_to_raw (+ encode_frame / decode_frame conveniences).  Arrays are numpy: pixels as the structured
dtype PIXEL_DT (u16 Yq, i16 Cbq, i16 Crq = PixelYCbCrQuant OLD:670-674), words as uint8 [n, 9] (Word27 OLD:666-669).

This module is plumbing over the C-ABI in include/t3hip.h.  There is NO CPU fallback: if the HIP library is missing
or no gfx950 device is usable, compute calls raise T3Error.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T3HIP_LIB", os.path.join(_HERE, "libt3hip.so"))   # T3HIP_LIB: timing-only ablation builds

PIXEL_DT = np.dtype([("Yq", "<u2"), ("Cbq", "<i2"), ("Crq", "<i2")])

OK, E_NODEVICE, E_HIP, E_ARG, E_CAPACITY, E_HEADER, E_RS, E_COMM = 0, -1, -2, -3, -4, -5, -6, -7
MODE_COMPAT, MODE_FIXED = 0, 1


class ProfileID:  # OLD:34
    RAW_MODE = 0xFF
    P1_RS26_24 = 0
    P2_RS26_22 = 1
    P3_RS26_20 = 2
    P4_RS26_18 = 3
    P5_RS26_22_2D = 4


class Cfg(C.Structure):
    """t3_cfg: POD mirror of EncoderConfig (OLD:862-873) / DecoderConfigSeen (OLD:874-884) + mode."""
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

    def copy(self):
        c = Cfg()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Cfg))
        return c


class Layout(C.Structure):
    _fields_ = [("n_raw_words", C.c_uint64), ("n_sym", C.c_uint64), ("band_len", C.c_uint64 * 9), ("band_blocks", C.c_uint64 * 9),
                ("band_body_off", C.c_uint64 * 9), ("body_syms", C.c_uint64), ("body_syms_framed", C.c_uint64),
                ("out_syms", C.c_uint64), ("out_words", C.c_uint64), ("header_syms", C.c_uint32), ("band_k", C.c_uint8 * 9),
                ("interleave2d", C.c_uint8), ("beacon_on", C.c_uint8), ("pad_", C.c_uint8)]


class FrameRecord(C.Structure):
    _fields_ = [("frame_idx", C.c_uint64), ("n_words", C.c_uint64), ("byte_offset", C.c_uint64), ("crc32", C.c_uint32),
                ("sym_sum", C.c_uint32), ("header_syms", C.c_uint8 * 54), ("profile", C.c_uint8), ("mode", C.c_uint8), ("pad_", C.c_uint8 * 8)]


FRAME_RECORD_BYTES = C.sizeof(FrameRecord)


class T3Error(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        msg = "%s: %s" % (where, strerror(code))
        if code == E_HIP:
            msg += " [" + last_hip_error() + "]"
        if code == E_COMM:
            msg += " [" + lib().t3hip_comm_last_error().decode() + "]"
        super().__init__(msg)


_lib_handle = None


def lib():
    """The C-ABI library; raises loudly when it has not been built (no fallback path exists)."""
    global _lib_handle
    if _lib_handle is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP extension missing: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % LIB_PATH)
        try:
            # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; when torch is going to be
            # used next to this library (tests, bench.py) it must be the first to load it, and libt3hip.so then binds
            # to that already-loaded runtime by SONAME.  Without torch the system runtime under /opt/rocm is used.
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.t3hip_strerror.restype = C.c_char_p
        L.t3hip_last_hip_error.restype = C.c_char_p
        L.t3hip_version.restype = C.c_char_p
        L.t3hip_comm_last_error.restype = C.c_char_p
        L.t3hip_encoded_words.restype = C.c_uint64
        L.t3hip_frame_record_scratch_bytes.restype = C.c_uint64
        _lib_handle = L
    return _lib_handle


def strerror(code):
    return lib().t3hip_strerror(C.c_int(code)).decode()


def last_hip_error():
    return lib().t3hip_last_hip_error().decode()


def version():
    return lib().t3hip_version().decode()


def device_count():
    return lib().t3hip_device_count()


def init(device=0):
    rc = lib().t3hip_init(C.c_int(device))
    if rc != OK:
        raise T3Error(rc, "t3hip_init(%d)" % device)


def is_ready():
    return bool(lib().t3hip_is_ready())


class Context:
    """One GPU's context (t3hip_create): own stream, tables and scratch.  `use()` binds the CALLING THREAD to it -- every call of this
    module made by that thread then runs on it -- `use_default()` goes back to the process default (init())."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().t3hip_create(C.c_int(device), C.byref(self.h)), "t3hip_create(%d)" % device)

    def use(self):
        _chk(lib().t3hip_use(self.h), "t3hip_use")

    @staticmethod
    def use_default():
        _chk(lib().t3hip_use(C.c_void_p()), "t3hip_use(NULL)")

    def device(self):
        return lib().t3hip_ctx_device(self.h)

    def destroy(self):
        if self.h:
            _chk(lib().t3hip_destroy(self.h), "t3hip_destroy"); self.h = C.c_void_p()


def _chk(rc, where):
    if rc != OK:
        raise T3Error(rc, where)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- config helpers (reference names) ----------------------------------------------------------------
def default_cfg():
    c = Cfg()
    lib().t3hip_cfg_default(C.byref(c))
    return c


def uep_uniform(cfg, idx=1):  # OLD:64-67
    for i in range(9):
        cfg.band_profile[i] = idx % 4


def uep_luma_priority(cfg):  # OLD:68-72
    for i in range(9):
        cfg.band_profile[i] = 1
    cfg.band_profile[0] = cfg.band_profile[3] = cfg.band_profile[6] = 2


def make_cfg(profile=ProfileID.P2_RS26_22, uep=1, tile=(0, 0), seed=(1, 1, 1), beacon=(0, 0, 0), superframe_words=8192,
             subword=27, centered=1, coset=0, mode=MODE_COMPAT):
    c = default_cfg()
    c.profile = profile
    if uep == "luma":
        uep_luma_priority(c)
    elif isinstance(uep, int):
        uep_uniform(c, uep)
    else:
        for i in range(9):
            c.band_profile[i] = uep[i]
    c.tile_w, c.tile_h = tile
    c.seed_a, c.seed_b, c.seed_s0 = seed
    c.beacon_words_period, c.beacon_band_slot, c.beacon_enabled = beacon
    c.superframe_words, c.subword, c.centered, c.coset, c.mode = superframe_words, subword, centered, coset, mode
    return c


class EncoderContext:  # OLD:885-900
    def __init__(self, mode=MODE_COMPAT):
        self.cfg = default_cfg()
        self.cfg.mode = mode


class DecoderContext:  # OLD:901-916
    def __init__(self, mode=MODE_COMPAT):
        self.cfg_last_seen = default_cfg()
        self.cfg_last_seen.mode = mode


# ---- host-only metadata ----------------------------------------------------------------------------------
def plan(n_raw_words, cfg):
    L = Layout()
    _chk(lib().t3hip_plan(C.c_uint64(n_raw_words), C.byref(cfg), C.byref(L)), "t3hip_plan")
    return L


def encoded_words(n_raw_words, cfg):
    return lib().t3hip_encoded_words(C.c_uint64(n_raw_words), C.byref(cfg))


def gf27_tables():
    e = np.zeros(78, np.uint8); lg = np.zeros(27, np.int16); m = np.zeros(729, np.uint8); iv = np.zeros(27, np.uint8)
    _chk(lib().t3hip_gf27_tables(_vp(e), _vp(lg), _vp(m), _vp(iv)), "t3hip_gf27_tables")
    return dict(exp=e, log=lg, mul=m, inv=iv)


def rs_generator(k):
    g = np.zeros(27 - k, np.uint8)
    _chk(lib().t3hip_rs_generator(C.c_int(k), _vp(g)), "t3hip_rs_generator")
    return g


def rs_parity_matrix(k, mode=MODE_COMPAT):
    P = np.zeros((k, 26 - k), np.uint8)
    _chk(lib().t3hip_rs_parity_matrix(C.c_int(k), C.c_int(mode), _vp(P)), "t3hip_rs_parity_matrix")
    return P


def header_pack(cfg, frame_seq=0, band_map_hash=0):
    s = np.zeros(27, np.uint8)
    _chk(lib().t3hip_header_pack(C.byref(cfg), C.c_uint32(frame_seq), C.c_uint32(band_map_hash), _vp(s)), "t3hip_header_pack")
    return s


def header_check(syms):
    s = np.ascontiguousarray(syms, np.uint8)
    return bool(lib().t3hip_header_check(_vp(s)))


def header_unpack(syms):
    s = np.ascontiguousarray(syms, np.uint8); c = default_cfg(); fs = C.c_uint32(); bh = C.c_uint32()
    _chk(lib().t3hip_header_unpack(_vp(s), C.byref(c), C.byref(fs), C.byref(bh)), "t3hip_header_unpack")
    return c, fs.value, bh.value


def header_encode(cfg, n_raw_words):
    out = np.zeros(96, np.uint8); n = C.c_uint32()
    _chk(lib().t3hip_header_encode(C.byref(cfg), C.c_uint64(n_raw_words), _vp(out), C.byref(n)), "t3hip_header_encode")
    return out[: n.value].copy()


# ---- host-buffer API (the reference's std::vector functions) ------------------------------------------------
def encode_raw_pixels_to_words(px):  # OLD:723-734
    px = np.ascontiguousarray(px, PIXEL_DT)
    out = np.zeros(((len(px) + 1) // 2, 9), np.uint8)
    _chk(lib().t3hip_pack_pixels(_vp(px), C.c_uint64(len(px)), _vp(out)), "t3hip_pack_pixels")
    return out


def decode_raw_words_to_pixels(words):  # OLD:735-747
    words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
    px = np.zeros(2 * len(words), PIXEL_DT)
    _chk(lib().t3hip_unpack_words(_vp(words), C.c_uint64(len(words)), _vp(px)), "t3hip_unpack_words")
    return px


_VALID_SUB = (27, 24, 21, 18, 15)


def encode_raw_pixels_to_words_subword(px, sub):  # NEWH:119-121 / NEWC:139-146: validates `sub`, else identical
    if sub not in _VALID_SUB:
        return None
    return encode_raw_pixels_to_words(px)


def decode_raw_words_to_pixels_subword(words, sub):  # NEWH:123-125 / NEWC:148-155
    if sub not in _VALID_SUB:
        return None
    return decode_raw_words_to_pixels(words)


# ---- subword trit streams and wire packings (OLD:834-859, TPACK:18-65) ----------------------------
def _u8(a):
    return np.ascontiguousarray(a, np.uint8).reshape(-1)


def extract_subword_stream_from_words(words, N):  # OLD:834-844 -> uint8 trits, N per word
    w = np.ascontiguousarray(words, np.uint8).reshape(-1, 9); n = len(w)
    out = np.zeros(n * int(N), np.uint8)
    _chk(lib().t3hip_subword_extract(_vp(w), C.c_uint64(n), C.c_int(int(N)), _vp(out)), "t3hip_subword_extract")
    return out


def build_words_from_subword_stream(trits, N, fill=0):  # OLD:845-859 -> Word27 array
    t = _u8(trits)
    lib().t3hip_subword_words.restype = C.c_uint64
    cap = int(lib().t3hip_subword_words(C.c_uint64(len(t)), C.c_int(int(N))))
    out = np.zeros((cap, 9), np.uint8); nw = C.c_uint64()
    _chk(lib().t3hip_subword_build(_vp(t), C.c_uint64(len(t)), C.c_int(int(N)), C.c_uint8(fill), _vp(out), C.c_uint64(cap), C.byref(nw)), "t3hip_subword_build")
    return out[: nw.value]


def ut_to_base243(trits):  # TPACK:28-38
    t = _u8(trits)
    lib().t3hip_base243_bytes.restype = C.c_uint64
    cap = int(lib().t3hip_base243_bytes(C.c_uint64(len(t))))
    out = np.zeros(cap, np.uint8); nb = C.c_uint64()
    _chk(lib().t3hip_base243_pack(_vp(t), C.c_uint64(len(t)), _vp(out), C.c_uint64(cap), C.byref(nb)), "t3hip_base243_pack")
    return out[: nb.value]


def base243_to_ut(data):  # TPACK:40-50 -> trits, or None where the reference returns false
    b = _u8(data)
    cap = 5 * max(len(b), 4)
    out = np.zeros(cap, np.uint8); nt = C.c_uint64()
    rc = lib().t3hip_base243_unpack(_vp(b), C.c_uint64(len(b)), _vp(out), C.c_uint64(cap), C.byref(nt))
    if rc == E_HEADER:
        return None
    _chk(rc, "t3hip_base243_unpack")
    return out[: nt.value]


def blit_center_rgb(src, sw, sh, cw, ch):  # io_image.hpp:125-140 -> uint8 canvas (ch, cw, 3)
    a = _u8(src)
    if len(a) != sw * sh * 3:
        raise ValueError("blit_center_rgb: %d bytes for a %dx%d image" % (len(a), sw, sh))
    out = np.zeros(cw * ch * 3, np.uint8)
    _chk(lib().t3hip_blit_center_rgb(_vp(a), C.c_int(sw), C.c_int(sh), _vp(out), C.c_int(cw), C.c_int(ch)), "t3hip_blit_center_rgb")
    return out.reshape(ch, cw, 3)


def extract_center_q(full, fw, fh, sw, sh):  # io_image.hpp:215-235 -> pixel records (sh * sw)
    f = np.ascontiguousarray(full)
    raw = f.view(np.uint8).reshape(-1)
    if len(raw) != fw * fh * 6:
        raise ValueError("extract_center_q: %d bytes for a %dx%d frame" % (len(raw), fw, fh))
    out = np.zeros(sw * sh * 6, np.uint8)
    _chk(lib().t3hip_extract_center_q(_vp(raw), C.c_int(fw), C.c_int(fh), _vp(out), C.c_int(sw), C.c_int(sh)), "t3hip_extract_center_q")
    return out.view(f.dtype) if f.dtype.itemsize == 6 else out


def blit_center_rgb_dev(d_src, sw, sh, d_dst, cw, ch, stream=0):
    _chk(lib().t3hip_blit_center_rgb_dev(C.c_void_p(d_src), C.c_int(sw), C.c_int(sh), C.c_void_p(d_dst), C.c_int(cw), C.c_int(ch), C.c_void_p(stream)), "t3hip_blit_center_rgb_dev")


def extract_center_q_dev(d_full, fw, fh, d_sub, sw, sh, stream=0):
    _chk(lib().t3hip_extract_center_q_dev(C.c_void_p(d_full), C.c_int(fw), C.c_int(fh), C.c_void_p(d_sub), C.c_int(sw), C.c_int(sh), C.c_void_p(stream)), "t3hip_extract_center_q_dev")


def words_to_bytes(words):  # TPACK:53-58
    w = np.ascontiguousarray(words, np.uint8).reshape(-1)
    out = np.zeros(len(w), np.uint8)
    _chk(lib().t3hip_mod27_bytes(_vp(w), C.c_uint64(len(w)), _vp(out)), "t3hip_mod27_bytes")
    return out


def bytes_to_words(data):  # TPACK:60-65: nothing unless len % 9 == 0
    b = _u8(data)
    if len(b) % 9:
        return np.zeros((0, 9), np.uint8)
    out = np.zeros(len(b), np.uint8)
    _chk(lib().t3hip_mod27_bytes(_vp(b), C.c_uint64(len(b)), _vp(out)), "t3hip_mod27_bytes")
    return out.reshape(-1, 9)


# ---- RGB8 <-> quantised YCbCr bridge (old/include/io_image.hpp:156-195) ---------------------------------
def rgb_to_quant_stream(rgb):  # (n, 3) or flat uint8 RGB -> PixelYCbCrQuant array
    r = _u8(rgb); n = len(r) // 3
    px = np.zeros(n, PIXEL_DT)
    _chk(lib().t3hip_rgb_to_quant(_vp(r), C.c_uint64(n), _vp(px)), "t3hip_rgb_to_quant")
    return px


def quant_stream_to_rgb(px):  # PixelYCbCrQuant array -> flat uint8 RGB
    p = np.ascontiguousarray(px, PIXEL_DT)
    out = np.zeros(3 * len(p), np.uint8)
    _chk(lib().t3hip_quant_to_rgb(_vp(p), C.c_uint64(len(p)), _vp(out)), "t3hip_quant_to_rgb")
    return out


def rgb_to_quant_dev(d_rgb, n_px, d_px, stream=0):
    _chk(lib().t3hip_rgb_to_quant_dev(C.c_void_p(d_rgb), C.c_uint64(n_px), C.c_void_p(d_px), C.c_void_p(stream)), "t3hip_rgb_to_quant_dev")


def quant_to_rgb_dev(d_px, n_px, d_rgb, stream=0):
    _chk(lib().t3hip_quant_to_rgb_dev(C.c_void_p(d_px), C.c_uint64(n_px), C.c_void_p(d_rgb), C.c_void_p(stream)), "t3hip_quant_to_rgb_dev")


def encode_rgb_dev(d_rgb, n_px, cfg, d_out, cap_words, stream=0):
    """RGB8 frame (device) -> coded stream: bridge kernel + fused encode on `stream`; returns the coded word count."""
    n = C.c_uint64()
    _chk(lib().t3hip_encode_rgb_dev(C.c_void_p(d_rgb), C.c_uint64(n_px), C.byref(cfg), C.c_void_p(d_out), C.c_uint64(cap_words), C.byref(n), C.c_void_p(stream)), "t3hip_encode_rgb_dev")
    return n.value


def decode_rgb_async(d_in, n_in, cfg, n_px, d_rgb, d_verdict, stream=0):
    """Coded stream -> RGB8 frame with the stream's known configuration (streaming decode entry + bridge kernel)."""
    _chk(lib().t3hip_decode_rgb_async(C.c_void_p(d_in), C.c_uint64(n_in), C.byref(cfg), C.c_uint64(n_px), C.c_void_p(d_rgb), C.c_void_p(d_verdict), C.c_void_p(stream)), "t3hip_decode_rgb_async")


def subword_extract_dev(d_words, n_words, N, d_trits, stream=0):
    _chk(lib().t3hip_subword_extract_dev(C.c_void_p(d_words), C.c_uint64(n_words), C.c_int(int(N)), C.c_void_p(d_trits), C.c_void_p(stream)), "t3hip_subword_extract_dev")


def subword_build_dev(d_trits, n_trits, N, fill, d_words, cap_words, stream=0):
    nw = C.c_uint64()
    _chk(lib().t3hip_subword_build_dev(C.c_void_p(d_trits), C.c_uint64(n_trits), C.c_int(int(N)), C.c_uint8(fill), C.c_void_p(d_words), C.c_uint64(cap_words), C.byref(nw), C.c_void_p(stream)), "t3hip_subword_build_dev")
    return nw.value


def base243_pack_dev(d_trits, n_trits, d_out, cap_bytes, stream=0):
    nb = C.c_uint64()
    _chk(lib().t3hip_base243_pack_dev(C.c_void_p(d_trits), C.c_uint64(n_trits), C.c_void_p(d_out), C.c_uint64(cap_bytes), C.byref(nb), C.c_void_p(stream)), "t3hip_base243_pack_dev")
    return nb.value


def base243_unpack_dev(d_in, n_bytes, total, d_trits, stream=0):
    _chk(lib().t3hip_base243_unpack_dev(C.c_void_p(d_in), C.c_uint64(n_bytes), C.c_uint64(total), C.c_void_p(d_trits), C.c_void_p(stream)), "t3hip_base243_unpack_dev")


def encode_profile_from_raw(raw_words, ectx):  # OLD:1043-1169 -> (True, words)
    cfg = ectx.cfg if isinstance(ectx, EncoderContext) else ectx
    raw = np.ascontiguousarray(raw_words, np.uint8).reshape(-1, 9)
    cap = encoded_words(len(raw), cfg)
    out = np.zeros((cap, 9), np.uint8); n = C.c_uint64()
    _chk(lib().t3hip_encode_profile(_vp(raw), C.c_uint64(len(raw)), C.byref(cfg), _vp(out), C.c_uint64(cap), C.byref(n)), "t3hip_encode_profile")
    return True, out[: n.value]


def encode_frame(px, ectx):  # pack + profile encode in one fused launch
    cfg = ectx.cfg if isinstance(ectx, EncoderContext) else ectx
    px = np.ascontiguousarray(px, PIXEL_DT)
    cap = encoded_words((len(px) + 1) // 2, cfg)
    out = np.zeros((cap, 9), np.uint8); n = C.c_uint64()
    _chk(lib().t3hip_encode_frame(_vp(px), C.c_uint64(len(px)), C.byref(cfg), _vp(out), C.c_uint64(cap), C.byref(n)), "t3hip_encode_frame")
    return True, out[: n.value]


def _decode(fn, words, dctx, unit_dt, units_per_word, where):
    seen = dctx.cfg_last_seen if isinstance(dctx, DecoderContext) else dctx
    words = np.ascontiguousarray(words, np.uint8).reshape(-1, 9)
    cap = units_per_word * (len(words) + 16)
    out = np.zeros(cap if unit_dt is PIXEL_DT else (cap, 9), unit_dt); n = C.c_uint64()
    rc = fn(_vp(words), C.c_uint64(len(words)), C.byref(seen), _vp(out), C.c_uint64(cap), C.byref(n))
    if rc in (E_HEADER, E_RS):  # the reference's `false`: output left empty (OLD:997)
        return False, out[:0]
    _chk(rc, where)
    return True, out[: n.value]


def decode_profile_to_raw(words, dctx):  # OLD:995-1041 -> (ok, raw words); mutates dctx.cfg_last_seen like the reference
    return _decode(lib().t3hip_decode_profile, words, dctx, np.uint8, 1, "t3hip_decode_profile")


def decode_frame(words, dctx):
    return _decode(lib().t3hip_decode_frame, words, dctx, PIXEL_DT, 2, "t3hip_decode_frame")


# ---- device-resident API: raw device pointers (ints) + hipStream_t (int) -----------------------------------------
def pack_pixels_dev(d_px, n_px, d_words, stream=0):
    _chk(lib().t3hip_pack_pixels_dev(C.c_void_p(d_px), C.c_uint64(n_px), C.c_void_p(d_words), C.c_void_p(stream)), "t3hip_pack_pixels_dev")


def unpack_words_dev(d_words, n_words, d_px, stream=0):
    _chk(lib().t3hip_unpack_words_dev(C.c_void_p(d_words), C.c_uint64(n_words), C.c_void_p(d_px), C.c_void_p(stream)), "t3hip_unpack_words_dev")


def encode_profile_dev(d_raw, n_raw, cfg, d_out, cap_words, stream=0):
    n = C.c_uint64()
    _chk(lib().t3hip_encode_profile_dev(C.c_void_p(d_raw), C.c_uint64(n_raw), C.byref(cfg), C.c_void_p(d_out), C.c_uint64(cap_words), C.byref(n), C.c_void_p(stream)), "t3hip_encode_profile_dev")
    return n.value


def encode_frame_dev(d_px, n_px, cfg, d_out, cap_words, stream=0):
    n = C.c_uint64()
    _chk(lib().t3hip_encode_frame_dev(C.c_void_p(d_px), C.c_uint64(n_px), C.byref(cfg), C.c_void_p(d_out), C.c_uint64(cap_words), C.byref(n), C.c_void_p(stream)), "t3hip_encode_frame_dev")
    return n.value


def decode_profile_dev(d_in, n_in, seen, d_out, cap_units, to_pixels=False, stream=0):
    """Returns (rc, n_out): rc is OK, E_HEADER or E_RS (the reference's bool); other codes raise."""
    n = C.c_uint64()
    rc = lib().t3hip_decode_profile_dev(C.c_void_p(d_in), C.c_uint64(n_in), C.byref(seen), C.c_void_p(d_out), C.c_uint64(cap_units), C.byref(n), C.c_int(1 if to_pixels else 0), C.c_void_p(stream))
    if rc not in (OK, E_HEADER, E_RS):
        raise T3Error(rc, "t3hip_decode_profile_dev")
    return rc, n.value


def read_header_dev(d_in, n_in, mode, stream=0):
    c = default_cfg(); c.mode = mode; n = C.c_uint64()
    rc = lib().t3hip_read_header_dev(C.c_void_p(d_in), C.c_uint64(n_in), C.c_int(mode), C.byref(c), C.byref(n), C.c_void_p(stream))
    if rc not in (OK, E_HEADER):
        raise T3Error(rc, "t3hip_read_header_dev")
    return rc, c, n.value


def decode_body_dev(d_in, n_in, cfg, n_raw, d_out, cap_units, d_fail, to_pixels=False, stream=0):
    n = C.c_uint64()
    _chk(lib().t3hip_decode_body_dev(C.c_void_p(d_in), C.c_uint64(n_in), C.byref(cfg), C.c_uint64(n_raw), C.c_void_p(d_out), C.c_uint64(cap_units), C.byref(n),
                                     C.c_int(1 if to_pixels else 0), C.c_void_p(d_fail), C.c_void_p(stream)), "t3hip_decode_body_dev")
    return n.value


def decode_frame_async(d_in, n_in, cfg, n_raw, d_out, cap_units, d_verdict, to_pixels=True, stream=0):
    """Streaming decode with a known configuration, no synchronisation; d_verdict -> two device uint32: [0] header differs,
    [1] uncorrectable blocks.  Returns the unit count the launch will produce."""
    n = C.c_uint64()
    _chk(lib().t3hip_decode_frame_async(C.c_void_p(d_in), C.c_uint64(n_in), C.byref(cfg), C.c_uint64(n_raw), C.c_void_p(d_out), C.c_uint64(cap_units), C.byref(n),
                                        C.c_int(1 if to_pixels else 0), C.c_void_p(d_verdict), C.c_void_p(stream)), "t3hip_decode_frame_async")
    return n.value


def rs_encode_blocks_dev(k, mode, d_data, n_blocks, d_code, stream=0):
    _chk(lib().t3hip_rs_encode_blocks_dev(C.c_int(k), C.c_int(mode), C.c_void_p(d_data), C.c_uint64(n_blocks), C.c_void_p(d_code), C.c_void_p(stream)), "t3hip_rs_encode_blocks_dev")


def rs_decode_blocks_dev(k, mode, d_code, n_blocks, d_data, d_ok, stream=0):
    _chk(lib().t3hip_rs_decode_blocks_dev(C.c_int(k), C.c_int(mode), C.c_void_p(d_code), C.c_uint64(n_blocks), C.c_void_p(d_data), C.c_void_p(d_ok), C.c_void_p(stream)), "t3hip_rs_decode_blocks_dev")


def inject_errors_dev(d_words, first_sym, n_blocks, seed, max_err, stream=0):
    _chk(lib().t3hip_inject_errors_dev(C.c_void_p(d_words), C.c_uint64(first_sym), C.c_uint64(n_blocks), C.c_uint32(seed), C.c_int(max_err), C.c_void_p(stream)), "t3hip_inject_errors_dev")


def frame_record_scratch_bytes(n_words=0):
    """Scratch t3hip_frame_record_dev wants (64 bytes still work: two accumulators, a fill kernel and atomics instead of per-workgroup partials)."""
    return int(lib().t3hip_frame_record_scratch_bytes(C.c_uint64(n_words)))


def frame_record_dev(d_words, n_words, frame_idx, cfg, d_rec, d_scratch, scratch_bytes=64, stream=0):
    _chk(lib().t3hip_frame_record_dev(C.c_void_p(d_words), C.c_uint64(n_words), C.c_uint64(frame_idx), C.byref(cfg), C.c_void_p(d_rec), C.c_void_p(d_scratch), C.c_uint64(scratch_bytes), C.c_void_p(stream)), "t3hip_frame_record_dev")


def crc32_dev(d_data, n_bytes, stream=0):
    """CRC-32 of a device buffer (the containers' payload CRC, io_t3p_t3v.cpp:20-36), computed by crc_chunks_kernel."""
    out = C.c_uint32()
    _chk(lib().t3hip_crc32_dev(C.c_void_p(d_data), C.c_uint64(n_bytes), C.byref(out), C.c_void_p(stream)), "t3hip_crc32_dev")
    return out.value


def crc32(data):
    """CRC-32 of host bytes: uploaded, then the same kernel (no CPU path)."""
    buf = _u8(data); out = C.c_uint32()
    _chk(lib().t3hip_crc32(_vp(buf), C.c_uint64(buf.size), C.byref(out)), "t3hip_crc32")
    return out.value


def index_assemble(records_bytes, first_payload_offset=0):
    """records_bytes: uint8 array of concatenated t3_frame_record; returns a list of FrameRecord sorted by frame_idx with offsets."""
    buf = np.ascontiguousarray(records_bytes, np.uint8).copy()
    n = buf.size // FRAME_RECORD_BYTES
    _chk(lib().t3hip_index_assemble(_vp(buf), C.c_uint64(n), C.c_uint64(first_payload_offset)), "t3hip_index_assemble")
    return [FrameRecord.from_buffer_copy(buf[i * FRAME_RECORD_BYTES:(i + 1) * FRAME_RECORD_BYTES].tobytes()) for i in range(n)]


COMM_ID_BYTES = 128
PAD_FRAME_IDX = 2**64 - 1


def comm_available():
    """True when RCCL can be bound in this process (t3hip_comm_available: dlopen only -- no collective, no device call)."""
    return lib().t3hip_comm_available() == 0


def comm_unique_id():
    """Rank 0: the 128-byte rendezvous id of a new RCCL communicator (ncclGetUniqueId); hand it to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _chk(lib().t3hip_comm_unique_id(buf), "t3hip_comm_unique_id")
    return bytes(buf)


class Comm:
    """One RCCL communicator bound to this process's device (t3hip_comm_create after init()); the frame-index exchange
    step runs on it: index_allgather = ncclAllGather of 96-byte frame records, asynchronous on `stream`."""

    def __init__(self, unique_id, world, rank):
        self.h = C.c_void_p(); self.world, self.rank = world, rank
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _chk(lib().t3hip_comm_create(buf, C.c_int(world), C.c_int(rank), C.byref(self.h)), "t3hip_comm_create")

    def index_allgather(self, d_local, n_local, d_all, stream=0):
        _chk(lib().t3hip_index_allgather(self.h, C.c_void_p(d_local), C.c_uint64(n_local), C.c_void_p(d_all), C.c_void_p(stream)), "t3hip_index_allgather")

    def destroy(self):
        if self.h:
            lib().t3hip_comm_destroy(self.h); self.h = C.c_void_p()


class Event:
    """HIP event on the caller's stream (bench.py times kernels with these)."""

    def __init__(self):
        self.h = C.c_void_p()
        _chk(lib().t3hip_event_create(C.byref(self.h)), "t3hip_event_create")

    def record(self, stream=0):
        _chk(lib().t3hip_event_record(self.h, C.c_void_p(stream)), "t3hip_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float()
        _chk(lib().t3hip_event_elapsed_ms(self.h, stop.h, C.byref(ms)), "t3hip_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            lib().t3hip_event_destroy(self.h)
        except Exception:
            pass
