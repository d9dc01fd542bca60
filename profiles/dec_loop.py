"""Decode-only loop for profiling (rocprofv3 kernel-trace / PMC passes): N launches of the streaming decode entry on one 8K
FIXED stream (C2 settings).  argv: [clean|errors] [launches] [warmup] [config: c2|c3|beacon|words]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "errors"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 200
conf = sys.argv[4] if len(sys.argv) > 4 else "c2"
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
px = ol.oracle().lcg_pixels(NPX, 12345)
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
P = t3.ProfileID; F = t3.MODE_FIXED
cfg = {"c2": t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F), "words": t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F),
       "c3": t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64), mode=F),
       "il": t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(64, 64), mode=F), "ilwide": t3.make_cfg(profile=P.P5_RS26_22_2D, uep=2, tile=(7680, 8), mode=F),
       "uep1d": t3.make_cfg(profile=P.P3_RS26_20, uep="luma", mode=F), "k22": t3.make_cfg(profile=P.P2_RS26_22, uep=1, mode=F),
       "beacon": t3.make_cfg(profile=P.P3_RS26_20, uep=2, beacon=(64, 4, 1), mode=F)}[conf]
words = conf == "words"
n_raw = NPX // 2; n_enc = t3.encoded_words(n_raw, cfg)
coded = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, coded.data_ptr(), n_enc, s)
L = t3.plan(n_raw, cfg)
if mode == "errors" and conf != "beacon":
    t3.inject_errors_dev(coded.data_ptr(), L.header_syms, L.body_syms // 26, 4242, (26 - max(L.band_k)) // 2, s)
elif mode == "errors":   # the injector works on a contiguous body: hurt the beacon-free stream, then put its symbols between the beacons
    cfg0 = t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F)
    n0 = t3.encoded_words(n_raw, cfg0); L0 = t3.plan(n_raw, cfg0)
    plain = torch.zeros(n0 * 9 + 64, dtype=torch.uint8, device="cuda")
    t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg0, plain.data_ptr(), n0, s)
    t3.inject_errors_dev(plain.data_ptr(), L0.header_syms, L0.body_syms // 26, 4242, 3, s)
    torch.cuda.synchronize()
    pos = torch.arange(L.body_syms_framed, device="cuda")
    keep = ~((pos >= 4) & ((pos - 4) % (9 * 64) == 0))
    idx = torch.nonzero(keep).reshape(-1)[: L.body_syms] + L.header_syms
    coded[idx] = plain[L0.header_syms: L0.header_syms + L0.body_syms]
    del pos, keep, idx
out = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
ver = torch.zeros(2, dtype=torch.int32, device="cuda")
f = lambda: t3.decode_frame_async(coded.data_ptr(), n_enc, cfg, n_raw, out.data_ptr(), n_raw if words else NPX, ver.data_ptr(), not words, s)
for _ in range(warm): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): f()
e1.record(); torch.cuda.synchronize()
ok = bool(torch.equal(out[:NPX * 6], d_px[:NPX * 6])) if not words else None
print("dec_loop %s %s: %.4f ms per launch, exact=%s verdict=%s" % (conf, mode, e0.elapsed_time(e1) / n, ok, ver.cpu().tolist()))
