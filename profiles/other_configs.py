"""Launch times of the encode entry point for the configurations that are parity-test cases rather than bench lines
(BASELINE configs[2] and friends), 8K frame, HIP events, 50 launches each after 500 of warm-up (sustained clock).  Output: one JSON object."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
W, H = 7680, 4320; NPX = W * H
px = ol.oracle().lcg_pixels(NPX, 12345)
import numpy as np
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
def run(name, cfg, words=False):
    n_raw = NPX // 2
    n_enc = t3.encoded_words(n_raw, cfg)
    out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    if words:
        raw = torch.zeros(n_raw * 9 + 64, dtype=torch.uint8, device="cuda")
        t3.pack_pixels_dev(d_px.data_ptr(), NPX, raw.data_ptr(), s)
        f = lambda: t3.encode_profile_dev(raw.data_ptr(), n_raw, cfg, out.data_ptr(), n_enc, s)
    else:
        f = lambda: t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
    for _ in range(500): f()                 # the card settles at its sustained clock after ~40 ms of continuous work (notes.md)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    return {"config": name, "ms": round(ms, 4), "coded_words": n_enc, "GBps": round((6 * NPX if not words else 9 * n_raw) / ms / 1e6 + 9 * n_enc / ms / 1e6, 1)}
res = []
P = t3.ProfileID
_only = [a.lower() for a in sys.argv[1:]]                      # optional substrings: run only the configurations whose name holds one of them
_run = run
def run(name, cfg, words=False):
    if _only and not any(o in name.lower() for o in _only): return None
    return _run(name, cfg, words)
res.append(run("C2 pixels, RS(26,20) all bands, 1-D, COMPAT", t3.make_cfg(profile=P.P3_RS26_20, uep=2)))
res.append(run("C2 same, FIXED", t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=t3.MODE_FIXED)))
res.append(run("C2 raw words in (encode_profile), COMPAT", t3.make_cfg(profile=P.P3_RS26_20, uep=2), words=True))
res.append(run("C3 P5 2-D 64x64 + luma-priority UEP (mixed k)", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64))))
res.append(run("P5 2-D 64x64, RS(26,20) all bands", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(64, 64))))
res.append(run("RS(26,24) all bands 1-D", t3.make_cfg(profile=P.P1_RS26_24, uep=0)))
res.append(run("RS(26,18) all bands 1-D", t3.make_cfg(profile=P.P4_RS26_18, uep=3)))
res.append(run("C2 + beacon every 64 words (fused into the stores)", t3.make_cfg(profile=P.P3_RS26_20, uep=2, beacon=(64, 4, 1))))
res.append(run("1-D luma-priority UEP (two k)", t3.make_cfg(profile=P.P3_RS26_20, uep="luma")))
res.append(run("1-D four different k per frame (bands 24,22,20,18,...)", t3.make_cfg(profile=P.P3_RS26_20, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0])))
res.append(run("2-D wide rows 1024x16, RS(26,20) all bands", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(1024, 16))))
res.append(run("2-D wide rows 7680x8, RS(26,20) all bands", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(7680, 8))))
res.append(run("2-D odd tile 7x5, RS(26,20) all bands", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(7, 5))))
res = [r for r in res if r is not None]
print(json.dumps(res, indent=1))
