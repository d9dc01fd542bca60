"""SURVEY §8 row f2 — T3P6 / T3V6 containers.  CPU part: the byte layout of the Python mirror (which the C++ header
include/io_t3p_t3v.hpp restates field for field) against an independent restatement built here with struct + zlib, and
the reader's behaviour (meta approval before any payload byte, CRC and magic failures).  PARITY UNPINNED against the
reference: src/io_t3p_t3v.cpp does not compile in this image (jumps across initialisations), so there is no reference
output; the layout is checked against the format as documented in include/io_t3p_t3v.hpp:14-21 of the reference.
The payload CRC itself runs on the GPU in the product; here a zlib stand-in is patched in as the *checker* so that the
reader logic can be exercised without a device.  GPU part (marked): the device CRC against zlib, and full files."""
import importlib
import os
import struct
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cont():
    import __graft_entry__ as ge
    ge.build()
    pkg = ge.load_package()
    return importlib.import_module(pkg.__name__ + ".containers")


def words(n, seed):
    return np.random.default_rng(seed).integers(0, 27, size=9 * n, dtype=np.uint8)


def hdr24_t3p(sub, w, h, meta_len, n):     # the reference's HdrCrcBuf as the x86-64 ABI lays it out, padding = 0
    b = bytearray(24); b[0] = 6; b[1] = sub; b[2:4] = struct.pack("<H", w); b[4:6] = struct.pack("<H", h)
    b[8:12] = struct.pack("<I", meta_len); b[16:24] = struct.pack("<Q", n); return bytes(b)


def hdr24_t3v(sub, w, h, frames, meta_len):
    b = bytearray(24); b[0] = 6; b[1] = sub; b[2:4] = struct.pack("<H", w); b[4:6] = struct.pack("<H", h)
    b[8:16] = struct.pack("<Q", frames); b[16:20] = struct.pack("<I", meta_len); return bytes(b)


def expect_t3p(sub, w, h, wd, meta):
    out = b"T3P6" + bytes([6, sub]) + struct.pack("<HHIQ", w, h, len(meta), wd.size // 9)
    out += struct.pack("<I", zlib.crc32(hdr24_t3p(sub, w, h, len(meta), wd.size // 9))) + meta + wd.tobytes()
    return out + struct.pack("<I", zlib.crc32(wd.tobytes()) if wd.size else 0)


def expect_t3v(sub, w, h, frames, meta, metas):
    n = len(frames); per = len(metas) == n
    out = b"T3V6" + bytes([6, sub]) + struct.pack("<HHQI", w, h, n, len(meta))
    out += struct.pack("<I", zlib.crc32(hdr24_t3v(sub, w, h, n, len(meta)))) + meta
    pos = len(out) + 20 * n
    idx = b""; body = b""
    for i, f in enumerate(frames):
        m = metas[i] if per else b""
        idx += struct.pack("<QQI", pos, f.size // 9, len(m))
        blk = m + f.tobytes() + struct.pack("<I", zlib.crc32(f.tobytes()) if f.size else 0)
        body += blk; pos += len(blk)
    return out + idx + body


def test_t3p_layout(cont):
    for n, meta in ((0, b""), (1, b"{}"), (57, b'{"route_ttl":3,"route_phase":1}')):
        wd = words(n, n)
        got = cont.t3p_bytes(24, 3840, 2160, wd, meta, payload_crc=zlib.crc32(wd.tobytes()))
        assert got == expect_t3p(24, 3840, 2160, wd, meta)
    assert cont.t3p_hdr_crc(6, 27, 7680, 4320, 5, 12345) == zlib.crc32(hdr24_t3p(27, 7680, 4320, 5, 12345))


def test_t3v_layout_and_index(cont):
    frames = [words(40, 1), words(0, 2), words(13, 3)]
    crcs = [zlib.crc32(f.tobytes()) for f in frames]
    metas = [b'{"f":0}', b"", b'{"f":2,"k":"v"}']
    for ms in (metas, [], metas[:2]):                                     # a wrong-length list means "no per-frame meta"
        got = cont.t3v_bytes(27, 7680, 4320, frames, b'{"g":1}', ms, payload_crcs=crcs)
        assert got == expect_t3v(27, 7680, 4320, frames, b'{"g":1}', ms)
    assert cont.t3v_bytes(15, 854, 480, [], b"", [], payload_crcs=[]) == expect_t3v(15, 854, 480, [], b"", [])


def test_readers(cont, tmp_path, monkeypatch):
    monkeypatch.setattr(cont, "_device_crc32", lambda b: zlib.crc32(np.ascontiguousarray(b).tobytes()))   # checker stand-in, CPU run only
    wd = words(100, 7); p = str(tmp_path / "a.t3p")
    assert cont.t3p_write(p, 21, 1920, 1080, wd, b'{"route_phase":0}')
    ok, sub, w, h, meta, n, err = cont.t3p_read_header(p)
    assert ok and (sub, w, h, meta, n) == (21, 1920, 1080, b'{"route_phase":0}', 100)
    ok, got, err = cont.t3p_read_payload(p); assert ok and np.array_equal(got, wd)
    seen = []
    ok, got, err = cont.t3p_read_payload(p, approve_meta=lambda m: seen.append(m) or False)
    assert not ok and got.size == 0 and "not approved" in err and seen == [b'{"route_phase":0}']
    raw = bytearray(open(p, "rb").read())
    bad = bytearray(raw); bad[60] ^= 1; open(p, "wb").write(bad)               # payload byte
    ok, got, err = cont.t3p_read_payload(p); assert not ok and err == "t3p: payload crc mismatch"
    bad = bytearray(raw); bad[7] ^= 1; open(p, "wb").write(bad)                # width field
    assert cont.t3p_read_header(p)[-1] == "t3p: header crc mismatch"
    bad = bytearray(raw); bad[0] = ord("X"); open(p, "wb").write(bad)
    assert cont.t3p_read_header(p)[-1] == "t3p: bad magic"
    open(p, "wb").write(raw[:20]); assert cont.t3p_read_header(p)[-1] == "t3p_read_header: I/O error"
    empty = str(tmp_path / "e.t3p"); cont.t3p_write(empty, 27, 1, 1, np.zeros(0, np.uint8), b"")
    ok, got, err = cont.t3p_read_payload(empty); assert ok and got.size == 0

    frames = [words(40, 1), words(0, 2), words(13, 3)]; v = str(tmp_path / "a.t3v")
    assert cont.t3v_write(v, 27, 7680, 4320, frames, b'{"g":1}', [b"m0", b"m1", b"m2"])
    ok, sub, w, h, meta, n, index, err = cont.t3v_read_header(v)
    assert ok and (sub, w, h, meta, n) == (27, 7680, 4320, b'{"g":1}', 3) and [i[1] for i in index] == [40, 0, 13]
    for i, f in enumerate(frames):
        ok, got, err = cont.t3v_read_frame(v, i); assert ok and np.array_equal(got, f)
    assert cont.t3v_read_frame(v, 3)[2] == "t3v: frame idx OOB"
    ok, got, err = cont.t3v_read_frame(v, 2, approve_meta=lambda m: m != b"m2"); assert not ok and "not approved" in err
    raw = bytearray(open(v, "rb").read()); raw[index[0][0] + 5] ^= 2; open(v, "wb").write(raw)
    assert cont.t3v_read_frame(v, 0)[2] == "t3v: frame payload crc mismatch"
    ok, got, err = cont.t3v_read_frame(v, 2); assert ok and np.array_equal(got, frames[2])


@pytest.mark.gpu
def test_device_crc32_vs_zlib():
    import torch
    import __graft_entry__ as ge
    t3 = ge.load_package(); t3.init(0)
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, size=3_000_003, dtype=np.uint8)
    d = torch.from_numpy(big).cuda(); s = torch.cuda.current_stream().cuda_stream
    # >= 128 KiB on an aligned buffer: whole 2 KiB rounds on the matrix cores + a table-kernel tail; below, or misaligned: tables only
    for n in (0, 1, 8, 9, 15, 16, 17, 2303, 2304, 2305, 4608, 65536, 131071, 131072, 131073, 262144, 589_824 + 5, 2_999_296, 3_000_003):
        assert t3.crc32_dev(d.data_ptr(), n, s) == zlib.crc32(big[:n].tobytes()), n
        assert t3.crc32(big[:n]) == zlib.crc32(big[:n].tobytes()), n
    for off in (1, 2, 3, 4, 7, 9, 13, 16, 48):                             # misaligned starts take the byte path
        assert t3.crc32_dev(d.data_ptr() + off, 100_000, s) == zlib.crc32(big[off:off + 100_000].tobytes()), off
        assert t3.crc32_dev(d.data_ptr() + off, 2_000_001, s) == zlib.crc32(big[off:off + 2_000_001].tobytes()), off


@pytest.mark.gpu
def test_container_files_with_device_crc(tmp_path):
    import torch
    import __graft_entry__ as ge
    t3 = ge.load_package(); t3.init(0)
    cont = importlib.import_module(t3.__name__ + ".containers")
    import oracle_lib as ol
    # two encoded frames; the payload CRCs come from the frame records the encoder side produces on the device
    cfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)
    frames, crcs = [], []
    s = torch.cuda.current_stream().cuda_stream
    for f in range(2):
        px = ol.oracle().lcg_pixels(64 * 48, 12345 + f)
        ectx = t3.EncoderContext(); ectx.cfg = cfg
        ok, coded = t3.encode_frame(px, ectx); assert ok
        coded = np.ascontiguousarray(coded).view(np.uint8).reshape(-1)
        d = torch.from_numpy(coded).cuda(); rec = torch.zeros(96, dtype=torch.uint8, device="cuda"); scr = torch.zeros(64, dtype=torch.uint8, device="cuda")
        t3.frame_record_dev(d.data_ptr(), coded.size // 9, f, cfg, rec.data_ptr(), scr.data_ptr(), 64, s)
        torch.cuda.synchronize()
        r = t3.FrameRecord.from_buffer_copy(rec.cpu().numpy().tobytes())
        assert r.crc32 == zlib.crc32(coded.tobytes())
        frames.append(coded); crcs.append(r.crc32)
    v = str(tmp_path / "two.t3v")
    cont.t3v_write(v, 27, 64, 48, frames, b'{"codec":"v6"}', [b'{"f":0}', b'{"f":1}'], payload_crcs=crcs)
    assert open(v, "rb").read() == expect_t3v(27, 64, 48, frames, b'{"codec":"v6"}', [b'{"f":0}', b'{"f":1}'])
    for i in range(2):
        ok, got, err = cont.t3v_read_frame(v, i); assert ok and np.array_equal(got, frames[i])       # CRC checked on the device
    p = str(tmp_path / "one.t3p")
    cont.t3p_write(p, 27, 64, 48, frames[0])                                                         # CRC computed on the device
    assert open(p, "rb").read() == expect_t3p(27, 64, 48, frames[0], b"")
    raw = bytearray(open(p, "rb").read()); raw[40] ^= 1; open(p, "wb").write(raw)
    assert cont.t3p_read_payload(p)[2] == "t3p: payload crc mismatch"
