"""Row f1 on one 8K frame (33,177,600 px), HIP events at the sustained clock (300 launches of warm-up, 30 timed): the fused RGB front end
of the encoder (t3hip_encode_rgb_dev: 3 B/px in, coded words out, one launch), the decoder with RGB out (t3hip_decode_rgb_async), and the
stand-alone bridge kernels.  Algorithmic bytes: 3 B per pixel + 9 B per coded word."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import numpy as np
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
rgb_h = ol.oracle().lcg_rgb(NPX, 12345)
rgb = torch.from_numpy(rgb_h).cuda()
px = torch.zeros(6 * NPX, dtype=torch.uint8, device="cuda"); back = torch.zeros(3 * NPX + 64, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def timed(f, n=30, warm=300):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = []
for name, cfg in (("C2 COMPAT RS(26,20) 1-D", t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)),
                  ("C2 FIXED", t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2, mode=t3.MODE_FIXED)),
                  ("C3 2-D 64x64 + luma UEP", t3.make_cfg(profile=t3.ProfileID.P5_RS26_22_2D, uep="luma", tile=(64, 64)))):
    n_enc = t3.encoded_words(NPX // 2, cfg)
    out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    t = timed(lambda: t3.encode_rgb_dev(rgb.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s))
    res.append({"op": "encode_rgb (fused bridge) " + name, "ms": round(t, 4), "GBps": round((3 * NPX + 9 * n_enc) / t / 1e6, 1), "Mpix_s": round(NPX / t / 1e3, 1)})
    if cfg.mode == t3.MODE_FIXED:
        ver = torch.zeros(2, dtype=torch.int32, device="cuda")
        t = timed(lambda: t3.decode_rgb_async(out.data_ptr(), n_enc, cfg, NPX, back.data_ptr(), ver.data_ptr(), s))
        res.append({"op": "decode_rgb " + name, "ms": round(t, 4), "GBps": round((3 * NPX + 9 * n_enc) / t / 1e6, 1), "Mpix_s": round(NPX / t / 1e3, 1), "verdict": ver.cpu().tolist()})
a = timed(lambda: t3.rgb_to_quant_dev(rgb.data_ptr(), NPX, px.data_ptr(), s), 10, 20)
b = timed(lambda: t3.quant_to_rgb_dev(px.data_ptr(), NPX, back.data_ptr(), s), 10, 20)
res += [{"op": "rgb_to_quant kernel alone", "ms": round(a, 4), "GBps": round(9 * NPX / a / 1e6, 1)}, {"op": "quant_to_rgb kernel alone", "ms": round(b, 4), "GBps": round(9 * NPX / b / 1e6, 1)}]
print(json.dumps(res, indent=1))
