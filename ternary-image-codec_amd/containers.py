"""T3P6 / T3V6 containers — Python mirror of include/io_t3p_t3v.hpp, which mirrors the reference's
include/io_t3p_t3v.hpp:34-83 (src/io_t3p_t3v.cpp:56-389).  Words are uint8 arrays, 9 bytes per Word27.  The payload
CRC-32 runs on the GPU (`crc32`, crc_chunks_kernel) unless the caller passes the CRC the encoder side already produced
(`t3_frame_record.crc32`); there is no CPU path for it.  Header CRC: 24-byte image of the fields with the struct padding
defined as zero (see the C++ header for why).  PARITY UNPINNED: the reference's container source does not compile here.

Every reader returns (ok, ..., err) like the reference's bool + error string."""
import struct

import numpy as np

from . import crc32 as _device_crc32

S27 = 27


def _crc_fields(img):   # bitwise CRC-32 of the 24-byte header image (control data)
    c = 0xFFFFFFFF
    for ch in img:
        c ^= ch
        for _ in range(8):
            c = (0xEDB88320 ^ (c >> 1)) if (c & 1) else (c >> 1)
    return c ^ 0xFFFFFFFF


def t3p_hdr_crc(ver, sub, w, h, meta_len, words):
    return _crc_fields(struct.pack("<BBHHHIIQ", ver, sub, w, h, 0, meta_len, 0, words))


def t3v_hdr_crc(ver, sub, w, h, frames, meta_len):
    return _crc_fields(struct.pack("<BBHHHQII", ver, sub, w, h, 0, frames, meta_len, 0))


def _payload(words):
    b = np.ascontiguousarray(words, np.uint8).reshape(-1)
    assert b.size % 9 == 0, "Word27 arrays are 9 bytes per word"
    return b


def _payload_crc(b, given):
    if b.size == 0:
        return 0                                        # io_t3p_t3v.cpp:103-105
    return int(given) if given is not None else _device_crc32(b)


def t3p_bytes(sub, w, h, words, meta_json=b"", payload_crc=None):
    b = _payload(words); meta = bytes(meta_json); n = b.size // 9
    head = b"T3P6" + struct.pack("<BBHHIQ", 6, sub, w & 0xFFFF, h & 0xFFFF, len(meta), n)
    head += struct.pack("<I", t3p_hdr_crc(6, sub, w & 0xFFFF, h & 0xFFFF, len(meta), n))
    return head + meta + b.tobytes() + struct.pack("<I", _payload_crc(b, payload_crc))


def t3p_write(path, sub, w, h, words, meta_json=b"", payload_crc=None):   # io_t3p_t3v.hpp:40-44
    with open(path, "wb") as f:
        f.write(t3p_bytes(sub, w, h, words, meta_json, payload_crc))
    return True


def _t3p_head(f):
    fix = f.read(26)
    if len(fix) < 4:
        return None, "I/O error"
    if fix[:4] != b"T3P6":
        return None, "t3p: bad magic"
    if len(fix) < 26:
        return None, "I/O error"
    ver, sub, w, h, meta_len, n, crc = struct.unpack("<BBHHIQI", fix[4:])
    if t3p_hdr_crc(ver, sub, w, h, meta_len, n) != crc:
        return None, "t3p: header crc mismatch"
    meta = f.read(meta_len)
    if len(meta) != meta_len:
        return None, "I/O error"
    return (sub, w, h, meta, n), ""


def t3p_read_header(path):   # io_t3p_t3v.hpp:46-50 -> (ok, sub, w, h, meta, words_count, err)
    with open(path, "rb") as f:
        H, err = _t3p_head(f)
    if H is None:
        return False, S27, 0, 0, b"", 0, ("t3p_read_header: " + err) if err == "I/O error" else err
    return (True,) + H + ("",)


def t3p_read_payload(path, approve_meta=None):   # io_t3p_t3v.hpp:53-56 -> (ok, words, err)
    empty = np.zeros(0, np.uint8)
    with open(path, "rb") as f:
        H, err = _t3p_head(f)
        if H is None:
            return False, empty, ("t3p_read_payload: " + err) if err == "I/O error" else err
        sub, w, h, meta, n = H
        if approve_meta is not None and not approve_meta(meta):
            return False, empty, "t3p: meta not approved - payload not read"       # nothing of the payload has been read
        raw = f.read(9 * n); tail = f.read(4)
    if len(raw) != 9 * n or len(tail) != 4:
        return False, empty, "t3p_read_payload: I/O error"
    b = np.frombuffer(raw, np.uint8)
    if _payload_crc(b, None) != struct.unpack("<I", tail)[0]:
        return False, empty, "t3p: payload crc mismatch" if n else "t3p: payload crc mismatch (empty)"
    return True, b.copy(), ""


def t3v_bytes(sub, w, h, frames, meta_json_global=b"", metas_per_frame=(), payload_crcs=None):
    fr = [_payload(x) for x in frames]; meta = bytes(meta_json_global); n = len(fr)
    per_frame = len(metas_per_frame) == n                      # otherwise no frame carries meta (io_t3p_t3v.cpp:255)
    metas = [bytes(m) for m in metas_per_frame] if per_frame else [b""] * n
    head = b"T3V6" + struct.pack("<BBHHQI", 6, sub, w & 0xFFFF, h & 0xFFFF, n, len(meta))
    head += struct.pack("<I", t3v_hdr_crc(6, sub, w & 0xFFFF, h & 0xFFFF, n, len(meta))) + meta
    off = len(head) + 20 * n
    index = b""; blocks = []
    for i, b in enumerate(fr):
        index += struct.pack("<QQI", off, b.size // 9, len(metas[i]))
        crc = _payload_crc(b, None if payload_crcs is None else payload_crcs[i])
        blk = metas[i] + b.tobytes() + struct.pack("<I", crc)
        blocks.append(blk); off += len(blk)
    return head + index + b"".join(blocks)


def t3v_write(path, sub, w, h, frames, meta_json_global=b"", metas_per_frame=(), payload_crcs=None):   # io_t3p_t3v.hpp:65-70
    with open(path, "wb") as f:
        f.write(t3v_bytes(sub, w, h, frames, meta_json_global, metas_per_frame, payload_crcs))
    return True


def t3v_read_header(path):   # io_t3p_t3v.hpp:72-77 -> (ok, sub, w, h, meta_global, frame_count, index[(offset, words, meta_len)], err)
    bad = lambda e: (False, S27, 0, 0, b"", 0, [], e)
    with open(path, "rb") as f:
        fix = f.read(26)
        if len(fix) < 4:
            return bad("t3v_read_header: I/O error")
        if fix[:4] != b"T3V6":
            return bad("t3v: bad magic")
        if len(fix) < 26:
            return bad("t3v_read_header: I/O error")
        ver, sub, w, h, n, meta_len, crc = struct.unpack("<BBHHQII", fix[4:])
        if t3v_hdr_crc(ver, sub, w, h, n, meta_len) != crc:
            return bad("t3v: header crc mismatch")
        meta = f.read(meta_len); idx = f.read(20 * n)
    if len(meta) != meta_len or len(idx) != 20 * n:
        return bad("t3v_read_header: I/O error")
    return True, sub, w, h, meta, n, [struct.unpack_from("<QQI", idx, 20 * i) for i in range(n)], ""


def t3v_read_frame(path, frame_idx, approve_meta=None):   # io_t3p_t3v.hpp:80-84 -> (ok, words, err)
    empty = np.zeros(0, np.uint8)
    ok, sub, w, h, meta_g, n, index, err = t3v_read_header(path)
    if not ok:
        return False, empty, err
    if frame_idx >= n:
        return False, empty, "t3v: frame idx OOB"
    off, words, meta_len = index[frame_idx]
    with open(path, "rb") as f:
        f.seek(off)
        meta = f.read(meta_len)
        if len(meta) != meta_len:
            return False, empty, "t3v: read frame meta failed"
        if approve_meta is not None and not approve_meta(meta):
            return False, empty, "t3v: meta not approved - frame payload not read"
        raw = f.read(9 * words)
        if len(raw) != 9 * words:
            return False, empty, "t3v: read frame payload failed"
        tail = f.read(4)
        if len(tail) != 4:
            return False, empty, "t3v: read frame crc failed"
    b = np.frombuffer(raw, np.uint8)
    if _payload_crc(b, None) != struct.unpack("<I", tail)[0]:
        return False, empty, "t3v: frame payload crc mismatch" if words else "t3v: empty frame crc mismatch"
    return True, b.copy(), ""
