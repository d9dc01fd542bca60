"""Row f3 kernels on an 8K frame's worth of raw words (16,588,800 words), HIP events, 10 launches each: time and
algorithmic GB/s (bytes read + written per launch; one byte per trit on the trit side)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
t3 = g.load_package(); t3.init(0)
n_words = 16588800
w = torch.randint(0, 27, (n_words, 9), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def timed(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = []
for N in (27, 24, 15):
    tr = torch.zeros(n_words * N, dtype=torch.uint8, device="cuda"); back = torch.zeros((n_words, 9), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: t3.subword_extract_dev(w.data_ptr(), n_words, N, tr.data_ptr(), s))
    res.append({"kernel": "subword_extract", "N": N, "ms": round(ms, 4), "GBps": round(n_words * (9 + N) / ms / 1e6, 1)})
    ms = timed(lambda: t3.subword_build_dev(tr.data_ptr(), n_words * N, N, 0, back.data_ptr(), n_words, s))
    res.append({"kernel": "subword_build", "N": N, "ms": round(ms, 4), "GBps": round(n_words * (9 + N) / ms / 1e6, 1)})
    if N == 24:
        nt = n_words * N; nb = 4 + (nt + 4) // 5
        pk = torch.zeros(nb + 64, dtype=torch.uint8, device="cuda"); tr2 = torch.zeros(nt, dtype=torch.uint8, device="cuda")
        ms = timed(lambda: t3.base243_pack_dev(tr.data_ptr(), nt, pk.data_ptr(), nb + 64, s))
        res.append({"kernel": "base243_pack", "trits": nt, "ms": round(ms, 4), "GBps": round((nt + nb) / ms / 1e6, 1)})
        ms = timed(lambda: t3.base243_unpack_dev(pk.data_ptr(), nb, nt, tr2.data_ptr(), s))
        res.append({"kernel": "base243_unpack", "trits": nt, "ms": round(ms, 4), "GBps": round((nt + nb) / ms / 1e6, 1)})
        assert torch.equal(tr, tr2)
print(json.dumps(res, indent=1))
