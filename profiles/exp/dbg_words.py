import os, sys, torch, numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
px = ol.oracle().lcg_pixels(NPX, 12345)
d_px = torch.from_numpy(px.view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
P = t3.ProfileID; F = t3.MODE_FIXED
for name, cfg in (("c2", t3.make_cfg(profile=P.P3_RS26_20, uep=2, mode=F)), ("c3", t3.make_cfg(profile=P.P5_RS26_22_2D, uep="luma", tile=(64, 64), mode=F))):
    n_raw = NPX // 2; n_enc = t3.encoded_words(n_raw, cfg)
    coded = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
    t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, coded.data_ptr(), n_enc, s)
    out = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device="cuda")
    seen = t3.default_cfg(); seen.mode = cfg.mode
    r = t3.decode_profile_dev(coded.data_ptr(), n_enc, seen, out.data_ptr(), n_raw, False, s)
    torch.cuda.synchronize()
    raw = torch.zeros(n_raw * 9 + 64, dtype=torch.uint8, device="cuda")
    t3.pack_pixels_dev(d_px.data_ptr(), NPX, raw.data_ptr(), s); torch.cuda.synchronize()
    neq = (out[:n_raw * 9] != raw[:n_raw * 9])
    nz = torch.nonzero(neq).reshape(-1)
    print(name, "rc", r, "mismatches", int(neq.sum()), "first", nz[:8].tolist(), "last", nz[-4:].tolist() if len(nz) else [])
    if len(nz):
        i = int(nz[0]); print(" out", out[i - 4:i + 12].tolist(), "raw", raw[i - 4:i + 12].tolist())
