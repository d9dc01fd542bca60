"""The part's streaming ceiling for the codec launches' byte volumes: a hand-written copy kernel of the library (stream_copy_kernel: 16 bytes
per lane, four loads in flight, persistent grid-stride) reading / writing exactly what an 8K launch reads / writes, HIP events at the sustained
clock (500 launches of warm-up, median of 40).  The guide's figure for a float4 copy on MI355X is 6.29 TB/s (MI355X_MICROARCH.md)."""
import ctypes as C, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
t3 = g.load_package(); t3.init(0)
L = t3.lib()
NPX = 7680 * 4320; NW = 20766726
src = torch.empty(6 * NPX + 64, dtype=torch.uint8, device="cuda").random_(0, 255)
dst = torch.empty(6 * NPX + 64, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(name, n_read, n_write, bpc):
    f = lambda: L.t3hip_diag_stream_copy_dev(C.c_void_p(src.data_ptr()), C.c_uint64(n_read), C.c_void_p(dst.data_ptr()), C.c_uint64(n_write), C.c_int(bpc), C.c_void_p(s))
    for _ in range(500): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40)); ms = ts[len(ts) // 2]
    return {"case": name, "blocks_per_cu": bpc, "ms": round(ms, 4), "TBps": round((n_read + n_write) / ms / 1e9, 3), "frac_of_8TBps": round((n_read + n_write) / ms / 1e9 / 8.0, 3)}
res = []
for bpc in (2, 4, 8, -2, -4, -8):
    res.append(run("encode volumes: read 6 B/px (199.1 MB), write 9 B/word (186.9 MB)", 6 * NPX, (9 * NW) & ~15, bpc))
res.append(run("encode, RGB in: read 3 B/px, write 9 B/word", 3 * NPX, (9 * NW) & ~15, 8))
res.append(run("decode volumes: read 9 B/word, write 6 B/px", (9 * NW) & ~15, 6 * NPX, 8))
res.append(run("copy 199 MB -> 199 MB", 6 * NPX, 6 * NPX, 8))
print(json.dumps(res, indent=1))
