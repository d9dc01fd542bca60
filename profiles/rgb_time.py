"""Row f1 kernels on one 8K frame (33,177,600 px), HIP events, 10 launches: time and algorithmic GB/s (3 B + 6 B per pixel)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
t3 = g.load_package(); t3.init(0)
NPX = 7680 * 4320
rgb = torch.randint(0, 256, (3 * NPX,), dtype=torch.uint8, device="cuda")
px = torch.zeros(6 * NPX, dtype=torch.uint8, device="cuda"); back = torch.zeros(3 * NPX, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def timed(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = timed(lambda: t3.rgb_to_quant_dev(rgb.data_ptr(), NPX, px.data_ptr(), s))
b = timed(lambda: t3.quant_to_rgb_dev(px.data_ptr(), NPX, back.data_ptr(), s))
print(json.dumps([{"kernel": "rgb_to_quant", "ms": round(a, 4), "GBps": round(9 * NPX / a / 1e6, 1), "Mpix_s": round(NPX / a / 1e3, 1)},
                  {"kernel": "quant_to_rgb", "ms": round(b, 4), "GBps": round(9 * NPX / b / 1e6, 1), "Mpix_s": round(NPX / b / 1e3, 1)}]))
