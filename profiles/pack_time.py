"""K1 / K5 (pixels <-> raw Word27) on an 8K frame, device resident, HIP events, sustained clock."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
import numpy as np
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320; NW = NPX // 2
px = ol.oracle().lcg_pixels(NPX, 12345)
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
d_w = torch.zeros(NW * 9 + 64, dtype=torch.uint8, device="cuda"); d_back = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def tm(f, warm=400, n=50):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = tm(lambda: t3.pack_pixels_dev(d_px.data_ptr(), NPX, d_w.data_ptr(), s))
b = tm(lambda: t3.unpack_words_dev(d_w.data_ptr(), NW, d_back.data_ptr(), s))
ok = bool(torch.equal(d_back[:NPX * 6], d_px[:NPX * 6]))
print(json.dumps([{"kernel": "pack_pixels (K1)", "ms": round(a, 4), "GBps": round((6 * NPX + 9 * NW) / a / 1e6, 1)},
                  {"kernel": "unpack_words (K5)", "ms": round(b, 4), "GBps": round((6 * NPX + 9 * NW) / b / 1e6, 1), "roundtrip_exact": ok}]))
