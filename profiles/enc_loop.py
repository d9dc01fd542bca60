"""Encode-only loop for profiling (rocprofv3 kernel-trace / PMC passes): N launches of the fused encode entry on one 8K frame.
argv: [c2|c3|c3u|w1024|w7680|luma1d|rgb] [launches] [warmup]   (c3 = BASELINE configs[2]: P5 2-D 64x64 + luma-priority UEP; c3u = 2-D with RS(26,20) on all bands)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import numpy as np
import oracle_lib as ol
conf = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 300
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320; P = t3.ProfileID
cfg = {"c2": t3.make_cfg(profile=P.P3_RS26_20, uep=2), "rgb": t3.make_cfg(profile=P.P3_RS26_20, uep=2),
       "c3": t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64)), "c3u": t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(64, 64)),
       "w1024": t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(1024, 16)), "w7680": t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(7680, 8)),
       "luma1d": t3.make_cfg(profile=P.P3_RS26_20, uep="luma")}[conf]
s = torch.cuda.current_stream().cuda_stream
n_enc = t3.encoded_words(NPX // 2, cfg)
out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
if conf == "rgb":
    d_in = torch.from_numpy(ol.oracle().lcg_rgb(NPX, 12345)).cuda()
    f = lambda: t3.encode_rgb_dev(d_in.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
else:
    d_in = torch.from_numpy(ol.oracle().lcg_pixels(NPX, 12345).view(np.uint8)).cuda()
    f = lambda: t3.encode_frame_dev(d_in.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
for _ in range(warm): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): f()
e1.record(); torch.cuda.synchronize()
print("enc_loop %s: %.4f ms per launch, hash %s" % (conf, e0.elapsed_time(e1) / n, ol.fnv_hex(out[: n_enc * 9].cpu().numpy())))
