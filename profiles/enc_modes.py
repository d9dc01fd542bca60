"""COMPAT vs FIXED encode of the same 8K frame, alternating (clock drift shows up as a trend, not as a mode difference)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
import numpy as np
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
px = ol.oracle().lcg_pixels(NPX, 12345)
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
out = torch.zeros(21_000_000 * 9, dtype=torch.uint8, device="cuda")
def run(cfg):
    n_enc = t3.encoded_words(NPX // 2, cfg)
    f = lambda: t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 10, 4)
P = t3.ProfileID
c = t3.make_cfg(profile=P.P3_RS26_20, uep=2); f = t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=t3.MODE_FIXED)
print(json.dumps({"compat_fixed_alternating_ms": [[run(c), run(f)] for _ in range(4)]}))
