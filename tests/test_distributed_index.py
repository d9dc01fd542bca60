"""N>1 path on CPU: two `gloo` ranks shard a batch of frames round-robin (no data-path collective), build each frame's
index record, all-gather the fixed-size records and assemble the T3V-style frame index on every rank — the one exchange
step bench.py performs with RCCL on the GPU node.  Record payloads come from the oracle here (no GPU in this container)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, W, H = 7, 32, 24


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _frame_record(t3, orc, ol, f):
    px = orc.lcg_pixels(W * H, 12345 + f)
    rc, enc = orc.encode_frame(px, ol.make_cfg(profile=2, uep=2))
    assert rc == 0
    r = t3.FrameRecord()
    r.frame_idx, r.n_words, r.crc32, r.sym_sum, r.profile, r.mode = f, len(enc), orc.crc32(enc), orc.sym_sum(enc), 2, 0
    flat = enc.reshape(-1)
    for i in range(54):
        r.header_syms[i] = int(flat[i]) if i < len(flat) else 0
    return np.frombuffer(bytes(r), np.uint8).copy(), enc


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib as ol
    t3 = ge.load_package()
    sf = __import__("ternary_image_codec_amd.superframe", fromlist=["x"])
    orc = ol.oracle()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = sf.frames_of_rank(N_FRAMES, rank, world)
    n_local = (N_FRAMES + world - 1) // world
    local = np.zeros((n_local, t3.FRAME_RECORD_BYTES), np.uint8)
    pad = t3.FrameRecord(); pad.frame_idx = sf.PAD_FRAME_IDX
    local[:] = np.frombuffer(bytes(pad), np.uint8)
    for i, f in enumerate(mine):
        local[i], _ = _frame_record(t3, orc, ol, f)
    gathered = sf.gather_records_torch(torch.from_numpy(local))
    index = sf.assemble_index(gathered, first_payload_offset=64)
    q.put((rank, [(r.frame_idx, r.n_words, r.byte_offset, r.crc32, r.sym_sum, bytes(r.header_syms)) for r in index]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_index_exchange(t3, orc):
    import torch.multiprocessing as mp
    import oracle_lib as ol
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process expectation
    want, off = [], 64
    for f in range(N_FRAMES):
        rec, enc = _frame_record(t3, orc, ol, f)
        want.append((f, len(enc), off, orc.crc32(enc), orc.sym_sum(enc), bytes(enc.reshape(-1)[:54].tobytes())))
        off += 9 * len(enc)
    assert got[0] == got[1] == want
    assert [r[0] for r in got[0]] == list(range(N_FRAMES))          # every frame exactly once, in order
    assert C.sizeof(t3.FrameRecord) == 96


def test_round_robin_shard():
    import __graft_entry__ as ge
    ge.load_package()
    sf = __import__("ternary_image_codec_amd.superframe", fromlist=["x"])
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            seen = sorted(f for r in range(world) for f in sf.frames_of_rank(n, r, world))
            assert seen == list(range(n))
    assert sf.frames_of_rank(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]           # BASELINE config 4: 8 frames per GPU


def _comm_worker(rank, world, port, q, disable_on):
    """make_comm with RCCL unbindable on the ranks in `disable_on`: every rank must come back with None, no rank may hang."""
    if rank in disable_on:
        os.environ["T3HIP_COMM_DISABLE"] = "1"
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    t3 = ge.load_package()
    sf = __import__("ternary_image_codec_amd.superframe", fromlist=["x"])
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    able = t3.comm_available()
    try:
        comm = sf.make_comm(); res = "none" if comm is None else "comm"
    except Exception as e:   # noqa: BLE001  (no device in this container: t3hip_comm_create refuses, on every rank alike)
        res = "raise:" + type(e).__name__
    t = torch.tensor([rank + 1]); dist.all_reduce(t)                 # the collectives still line up afterwards
    q.put((rank, able, res, int(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("disable_on", [(0, 1), (1,), (0,)])
def test_comm_rendezvous_cannot_strand_a_rank(t3, disable_on):
    """a rank that cannot bind RCCL (all ranks of a node without librccl, or one rank only) must not leave the others in a
    mismatched collective or inside ncclCommInitRank: make_comm probes locally, agrees with an all-reduce, always broadcasts, and
    returns None on EVERY rank -- bench.py then reports the torch.distributed fallback in config.exchange."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q, disable_on)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[0] for g in got] == [0, 1] and all(g[3] == 3 for g in got)
    for r, able, res, _ in got:
        assert able == (r not in disable_on)
        assert res == "none"
