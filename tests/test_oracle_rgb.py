"""Row f1 (RGB8 <-> quantised YCbCr bridge, old/include/io_image.hpp:47-90): the reference header does not compile in
this image (ImageU8 has no member `swap`), so oracle/t3_oracle.c's restatement cannot be pinned to a reference build
("parity unpinned").  What can be checked on the CPU: an independent numpy float32/float64 restatement of the same
expressions agrees with the C restatement on every one of the 2^24 RGB inputs and every in-range quantised pixel."""
import numpy as np

import oracle_lib as ol


def _lround(x):
    return (np.sign(x) * np.floor(np.abs(x) + 0.5)).astype(np.int64)      # half away from zero; exact here: |x| < 2^23 and x + 0.5 is exact in float64


def _np_rgb_to_quant(rgb):
    f = np.float32
    r, g, b = (rgb[:, i].astype(f) for i in range(3))
    y = (f(0.299) * r + f(0.587) * g) + f(0.114) * b
    cb = ((f(-0.168736) * r - f(0.331264) * g) + f(0.5) * b) + f(128.0)
    cr = ((f(0.5) * r - f(0.418688) * g) - f(0.081312) * b) + f(128.0)
    Y, Cb, Cr = (np.clip(_lround(v.astype(np.float64)), 0, 255) for v in (y, cb, cr))
    out = np.zeros(len(rgb), ol.PIXEL_DT); n = out.dtype.names
    out[n[0]] = np.clip(_lround(Y * (242.0 / 255.0)), 0, 242)
    out[n[1]] = np.clip(_lround((Cb - 128) * (40.0 / 128.0)), -40, 40)
    out[n[2]] = np.clip(_lround((Cr - 128) * (40.0 / 128.0)), -40, 40)
    return out


def test_rgb_to_quant_restatements_agree_exhaustively(orc):
    v = np.arange(1 << 24, dtype=np.uint32)
    rgb = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], axis=1).astype(np.uint8)
    a = orc.rgb_to_quant(rgb.reshape(-1)); b = _np_rgb_to_quant(rgb)
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    names = a.dtype.names                                  # hand-checked points: black, white, pure red
    assert tuple(a[0]) == (0, 0, 0) and tuple(a[(1 << 24) - 1]) == (242, 0, 0) and tuple(a[255 << 16]) == (72, -13, 40)
    assert names[0] == "Yq"


def test_quant_to_rgb_restatements_agree(orc):
    Y, Cb, Cr = np.meshgrid(np.arange(243), np.arange(-40, 41), np.arange(-40, 41), indexing="ij")
    px = np.zeros(Y.size, ol.PIXEL_DT); n = px.dtype.names
    px[n[0]] = Y.reshape(-1); px[n[1]] = Cb.reshape(-1); px[n[2]] = Cr.reshape(-1)
    f = np.float32
    Yd = np.clip(_lround(px[n[0]].astype(np.float64) * (255.0 / 242.0)), 0, 255)
    Cbd = np.clip(_lround(128 + px[n[1]].astype(np.float64) * (128.0 / 40.0)), 0, 255)
    Crd = np.clip(_lround(128 + px[n[2]].astype(np.float64) * (128.0 / 40.0)), 0, 255)
    y, cb, cr = Yd.astype(f), Cbd.astype(f) - f(128.0), Crd.astype(f) - f(128.0)
    r = y + f(1.402) * cr; g = (y - f(0.344136) * cb) - f(0.714136) * cr; b = y + f(1.772) * cb
    want = np.stack([np.clip(_lround(v.astype(np.float64)), 0, 255) for v in (r, g, b)], axis=1).astype(np.uint8).reshape(-1)
    assert np.array_equal(orc.quant_to_rgb(px), want)


def test_centring_blit_restatements_agree(orc):
    """io_image.hpp:125-140 / 215-235 restated twice (C in oracle/, numpy slices here): parity unpinned, like the bridge."""
    rng = np.random.default_rng(5)
    for (sw, sh, cw, ch) in [(5, 3, 9, 8), (9, 8, 9, 8), (7, 11, 8, 5), (3, 20, 3, 7), (33, 17, 64, 64)]:
        src = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        want = np.zeros((ch, cw, 3), np.uint8); x0, y0 = max(0, (cw - sw) // 2), max(0, (ch - sh) // 2)
        rows = min(sh, ch - y0); want[y0:y0 + rows, x0:x0 + sw] = src[:rows]
        assert np.array_equal(orc.blit_center_rgb(src, sw, sh, cw, ch), want.reshape(-1))
        full = rng.integers(0, 256, (ch, cw, 6), dtype=np.uint8)             # window sw x sh out of a cw x ch frame
        want = np.zeros((sh, sw, 6), np.uint8); rows = min(sh, ch - y0); want[:rows] = full[y0:y0 + rows, x0:x0 + sw]
        assert np.array_equal(orc.extract_center_q(full, cw, ch, sw, sh), want.reshape(-1))
