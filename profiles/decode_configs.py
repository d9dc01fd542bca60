"""Launch times of the decode entry point (t3hip_decode_profile_dev, to pixels) for the FIXED-mode configurations, 8K frame,
HIP events, 30 launches each after 200 of warm-up (synchronous entry point: header read-back and verdict included); streams come from the
encoder; the "errors" rows carry 0..t injected symbol errors in every block (t of the weakest band's code)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
import numpy as np
t3 = g.load_package(); t3.init(0)
W, H = 7680, 4320; NPX = W * H
px = ol.oracle().lcg_pixels(NPX, 12345)
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
def run(name, cfg, errors=False, words=False):
    n_raw = NPX // 2
    n_enc = t3.encoded_words(n_raw, cfg)
    coded = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, coded.data_ptr(), n_enc, s)
    if errors:
        L = t3.plan(n_raw, cfg)
        t3.inject_errors_dev(coded.data_ptr(), L.header_syms, L.body_syms // 26, 4242, (26 - max(L.band_k)) // 2, s)
    out = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
    seen = t3.default_cfg(); seen.mode = cfg.mode
    f = lambda: t3.decode_profile_dev(coded.data_ptr(), n_enc, seen, out.data_ptr(), NPX if not words else n_raw, not words, s)
    for _ in range(200): f()                 # the card settles at its sustained clock after ~40 ms of continuous work (notes.md)
    torch.cuda.synchronize()
    if words:
        raw = torch.zeros(n_raw * 9 + 64, dtype=torch.uint8, device="cuda")
        t3.pack_pixels_dev(d_px.data_ptr(), NPX, raw.data_ptr(), s); torch.cuda.synchronize()
        ok = bool(torch.equal(out[:n_raw * 9], raw[:n_raw * 9]))
    else:
        ok = bool(torch.equal(out[:NPX * 6], d_px[:NPX * 6]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    return {"config": name, "ms": round(ms, 4), "pixels_exact": ok, "GBps": round(((9 * n_raw if words else 6 * NPX) + 9 * n_enc) / ms / 1e6, 1)}
P = t3.ProfileID; F = t3.MODE_FIXED
res = []
_only = [x.lower() for x in sys.argv[1:]]                      # optional substrings: run only the rows whose name holds one of them
_run = run
def run(name, cfg, errors=False, words=False):
    if _only and not any(o in name.lower() for o in _only): return None
    return _run(name, cfg, errors, words)
res.append(run("C2 FIXED RS(26,20) 1-D (fused decoder), clean stream", t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F)))
res.append(run("FIXED luma-priority UEP 1-D (one-launch UEP / 2-D decoder where it applies), clean", t3.make_cfg(profile=P.P3_RS26_20, uep="luma", mode=F)))
res.append(run("FIXED 2-D 64x64 RS(26,20) (one-launch UEP / 2-D decoder where it applies), clean", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(64, 64), mode=F)))
res.append(run("C3 FIXED 2-D 64x64 + luma UEP (one-launch UEP / 2-D decoder where it applies), clean", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64), mode=F)))
res.append(run("C2 FIXED + beacon every 64 words (stepped over in the fused decoder's loads), clean", t3.make_cfg(profile=P.P3_RS26_20, uep=2, beacon=(64, 4, 1), mode=F)))
res.append(run("C2 FIXED RS(26,20) 1-D (fused decoder), 0..3 errors per block", t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F), errors=True))
res.append(run("C3 FIXED 2-D 64x64 + luma UEP (one-launch UEP / 2-D decoder where it applies), 0..2 errors per block", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64), mode=F), errors=True))
res.append(run("C2 FIXED RS(26,20) 1-D (fused decoder) to raw words, clean stream", t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F), words=True))
res.append(run("C3 FIXED 2-D 64x64 + luma UEP (one-launch UEP / 2-D decoder where it applies) to raw words, clean", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64), mode=F), words=True))
res.append(run("FIXED 1-D four different k per frame (one-launch UEP / 2-D decoder where it applies), clean", t3.make_cfg(profile=P.P3_RS26_20, uep=[0, 1, 2, 3, 0, 1, 2, 3, 0], mode=F)))
res.append(run("FIXED 2-D wide rows 7680x8 RS(26,20) (one-launch UEP / 2-D decoder where it applies), clean", t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(7680, 8), mode=F)))
res = [r for r in res if r is not None]
print(json.dumps(res, indent=1))
