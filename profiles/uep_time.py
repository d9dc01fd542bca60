"""1-D luma-priority UEP (mixed k) and uniform RS(26,20) encodes of one 8K frame, 12 launches each (a -DT3_STAMPS build of the
library prints its phase breakdown at the 8th launch)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
import oracle_lib as ol
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
d_px = torch.from_numpy(ol.oracle().lcg_pixels(NPX, 12345).view(np.uint8)).cuda()
s = torch.cuda.current_stream().cuda_stream
which = sys.argv[1] if len(sys.argv) > 1 else "luma"
cfg = t3.make_cfg(profile=t3.ProfileID.P2_RS26_22, uep="luma") if which == "luma" else t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)
n_enc = t3.encoded_words(NPX // 2, cfg)
out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(12):
    if i == 2: e0.record()
    t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, s)
e1.record(); torch.cuda.synchronize()
print(which, "ms", round(e0.elapsed_time(e1) / 10, 4))
