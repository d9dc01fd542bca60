"""Time t3hip_frame_record_dev (CRC-32 of a coded 8K frame, 187 MB) alone, against whatever library T3HIP_LIB names."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
t3 = g.load_package()
t3.init(0)
n_words = 20766726
d = torch.randint(0, 27, (n_words * 9 + 64,), dtype=torch.uint8, device="cuda")
rec = torch.zeros(t3.FRAME_RECORD_BYTES, dtype=torch.uint8, device="cuda"); NS = 64 if os.environ.get("T3_SCR64") else t3.frame_record_scratch_bytes(n_words); scr = torch.zeros(NS, dtype=torch.uint8, device="cuda")
cfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)
s = torch.cuda.current_stream().cuda_stream
for _ in range(300): t3.frame_record_dev(d.data_ptr(), n_words, 0, cfg, rec.data_ptr(), scr.data_ptr(), NS, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): t3.frame_record_dev(d.data_ptr(), n_words, 0, cfg, rec.data_ptr(), scr.data_ptr(), NS, s)
e1.record(); torch.cuda.synchronize()
print("frame_record_ms %.4f" % (e0.elapsed_time(e1) / 100))
