"""Practical HBM ceiling for the encode kernel's traffic mix: a plain device copy of the same byte volume
(199 MB read + 199 MB write, torch copy_), timed with events.  Reported beside the kernel's own time in DESIGN.md."""
import torch, json
n = 198_000_000
src = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty_like(src)
for _ in range(600): dst.copy_(src)          # ~45 ms: the card settles at its sustained clock (profiles/r01/notes.md)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    dst.copy_(src); ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
ms = ts[len(ts) // 2]
# read-only and write-only
s32 = src.view(torch.int32)
for _ in range(3): s32.sum()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): s32.sum()
e1.record(); torch.cuda.synchronize(); rd = e0.elapsed_time(e1) / 20
e0.record()
for _ in range(20): dst.zero_()
e1.record(); torch.cuda.synchronize(); wr = e0.elapsed_time(e1) / 20
print(json.dumps({"copy_ms": round(ms, 4), "copy_TBps": round(2 * n / ms / 1e9, 3), "read_ms": round(rd, 4), "read_TBps": round(n / rd / 1e9, 3),
                  "fill_ms": round(wr, 4), "fill_TBps": round(n / wr / 1e9, 3)}))
