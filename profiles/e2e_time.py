"""Host-buffer (std::vector-shaped) entry points on an 8K frame, pageable memory: t3hip_encode_frame / t3hip_decode_frame wall time, best of N."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
px = ol.oracle().lcg_pixels(NPX, 12345)
L = t3.lib()
for mode in (0, 1):
    cfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2, mode=mode)
    out = np.ones((t3.encoded_words(NPX // 2, cfg), 9), np.uint8); n = C.c_uint64()
    te = []
    for _ in range(5):
        t0 = time.perf_counter()
        rc = L.t3hip_encode_frame(px.ctypes.data_as(C.c_void_p), C.c_uint64(len(px)), C.byref(cfg), out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n))
        te.append(time.perf_counter() - t0); assert rc == 0
    line = "mode %d encode %.2f ms (best of 5: %s) hash %s" % (mode, min(te) * 1e3, " ".join("%.2f" % (x * 1e3) for x in te), ol.fnv_hex(out))
    if mode == 1:
        back = np.ones(NPX + 64, t3.PIXEL_DT); seen = t3.DecoderContext(mode=1).cfg_last_seen; td = []
        for _ in range(5):
            t0 = time.perf_counter()
            rc = L.t3hip_decode_frame(out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(seen), back.ctypes.data_as(C.c_void_p), C.c_uint64(len(back)), C.byref(n))
            td.append(time.perf_counter() - t0); assert rc == 0 and n.value == NPX
        line += " | decode %.2f ms (%s) exact=%s" % (min(td) * 1e3, " ".join("%.2f" % (x * 1e3) for x in td), bool(np.array_equal(back[:NPX], px)))
    print(line)
