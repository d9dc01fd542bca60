"""Launch time of BASELINE configs[2]-style encodes (8K): C3 (2-D + luma UEP, mixed k), its 1-D counterpart, and the
uniform-k 2-D case.  HIP events, 10 launches each."""
import json, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
d_px = torch.from_numpy(ol.oracle().lcg_pixels(NPX, 12345).view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
P = t3.ProfileID
def run(name, cfg):
    n_enc = t3.encoded_words(NPX // 2, cfg)
    out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    f = lambda: t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 10, 4))
run("C3 2-D + luma UEP", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64)))
run("1-D + luma UEP", t3.make_cfg(profile=P.P2_RS26_22, uep="luma"))
run("2-D uniform k=20", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(64, 64)))
