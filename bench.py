#!/usr/bin/env python3
"""bench.py — headline benchmark of the Word27 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 from a bare shell: this process touches no GPU, starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`
on itself as a child process and exits with the child's code.  Under a launcher (RANK/WORLD_SIZE in the environment, the
driver's form) it is one rank.

One step = one synthetic 8K (7680x4320) frame through the hot path on each rank, inputs already resident in HBM:
  1. fused encode, COMPAT arithmetic, P3 / RS(26,20) on all 9 bands, 1-D (BASELINE configs[1]; the output of frame
     seed 12345 is pinned to the reference's hash b6c43f2f4aa44763),
  2. decode of the same frame's FIXED-mode (v6c) stream carrying 0..3 injected symbol errors in every RS block
     (BASELINE configs[4] semantics; exact recovery of the pixels is asserted after the timed region).  The stream's first
     frame is decoded before the timed region through the synchronous, reference-shaped entry (header read back and
     parsed on the host); the timed frames use the streaming entry t3hip_decode_frame_async: body decode launched with
     the configuration last seen, header symbols checked on the device, verdict words read inside the timed region
     after the last step (--sync-decode times the synchronous entry for every frame instead),
  3. the frame's index record (CRC-32 + header symbols) for the T3V-style super-frame index, on the same stream
     (--overlap-record: on a second HIP stream beside the decode, as in round 1; the round-2 decoder fills the register files,
     so the CRC kernel then runs behind it anyway).
Every rank holds --frames-per-rank (8) distinct frames, BASELINE configs[3]: global frame f = rank + N j has LCG seed
12345 + f and lives on rank f mod N; step i works on the rank's frame i mod 8.  Frames are independent, so ranks shard them
with no data-path collective (weak scaling); the only exchange is one all-gather of the K fixed-size index records per rank at
the end of the batch — t3hip_index_allgather, RCCL over xGMI — inside the timed region (one untimed exchange in the warm-up
takes RCCL's lazy set-up out of it).

Prints ONE JSON line on rank 0 (contract in the task statement; roofline / cpu_baseline / end_to_end objects included)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 7680, 4320
NPX = W * H
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
GOLD_HASH_C2 = "b6c43f2f4aa44763"
SEED0 = 12345


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(orc, ol, px):
    """The reference's own encoder (oracle/_ref, built from /root/reference in place; the C port where that build is absent)
    and the C port of the FIXED decoder (the reference cannot decode its own streams), timed on this box's host cores:
    1 thread on a bounded sample, then T threads frame-parallel (one sample per thread, the reference has no threads of
    its own), as SURVEY 8d asks.  About 25 s of wall time in all."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    model, nproc, usable = host_cpu()
    threads = max(1, min(usable, 16))
    cfg = ol.make_cfg(profile=2, uep=2)
    kind = "reference" if ol.have_ref() else "port"
    enc_lib = ol.Ref() if kind == "reference" else orc

    def enc_job(sample):
        t0 = time.perf_counter(); rc, _ = enc_lib.encode_frame(sample, cfg, cap=len(sample)); assert rc == 0
        return time.perf_counter() - t0

    enc_px = NPX // 2                                   # 1-thread sample: half a frame (about 3 s on the reference)
    t_enc1 = enc_job(px[:enc_px])
    enc1 = enc_px / t_enc1 / 1e6
    par_px = NPX // 4                                   # T threads, a quarter frame each (bounded memory: the reference copies its stream several times)
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter(); list(ex.map(enc_job, [px[(i % 4) * par_px:(i % 4 + 1) * par_px] for i in range(threads)])); t_encT = time.perf_counter() - t0
    encT = threads * par_px / t_encT / 1e6
    # decode leg: FIXED stream with 0..3 errors per block, C port
    dec_px = NPX // 4
    fcfg = ol.make_cfg(profile=2, uep=2, mode=1)
    sample = px[:dec_px]
    rc, fenc = orc.encode_frame(sample, fcfg, cap=len(sample)); assert rc == 0
    nblk = (len(fenc) * 9 - 90) // 26
    bad = orc.inject_errors(fenc, 90, nblk, 777, 3)

    def dec_job(_):
        t0 = time.perf_counter(); rc, back = orc.decode_frame(bad, ol.make_cfg(mode=1)); dt = time.perf_counter() - t0
        assert rc == 0 and np.array_equal(back, sample)
        return dt

    t_dec1 = dec_job(0)
    dec1 = dec_px / t_dec1 / 1e6
    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter(); list(ex.map(dec_job, range(threads))); t_decT = time.perf_counter() - t0
    decT = threads * dec_px / t_decT / 1e6
    both1 = 1.0 / (1.0 / enc1 + 1.0 / dec1); bothT = 1.0 / (1.0 / encT + 1.0 / decT)
    return {"value": round(both1, 4), "unit": "Mpix/s", "cores": 1, "kind": kind,
            "sample": "1 thread: %s encoder on the first 1/2 of the 8K frame (%.1f s), C port of the FIXED decoder on the first 1/4 with 0..3 errors per block (%.1f s); value = 1/(1/enc + 1/dec).  threads_value: the same two legs on %d threads, frame-parallel (one 1/4-frame sample per thread; %.1f s + %.1f s)"
                      % ("unmodified reference" if kind == "reference" else "C port of the reference", t_enc1, t_dec1, threads, t_encT, t_decT),
            "encode_mpix_s": round(enc1, 4), "decode_mpix_s": round(dec1, 4),
            "threads": threads, "threads_value": round(bothT, 4), "threads_encode_mpix_s": round(encT, 4), "threads_decode_mpix_s": round(decT, 4),
            "cpu_model": model, "nproc": nproc, "nproc_usable": usable}


def end_to_end(t3, px, cfg, fenc_bad, reps=3):
    """The std::vector-shaped entry points (host buffers in, host buffers out: H2D + kernels + D2H + sync), pageable memory as
    a std::vector is; buffers allocated and touched beforehand.  SURVEY 8d / BASELINE.md 3: reported beside the HBM-resident
    number, never `value`."""
    import ctypes as C
    import numpy as np
    L = t3.lib()
    out = np.ones((t3.encoded_words(len(px) // 2, cfg), 9), np.uint8); n = C.c_uint64()
    back = np.ones(len(px) + 64, t3.PIXEL_DT)
    seen = t3.DecoderContext(mode=t3.MODE_FIXED).cfg_last_seen
    enc_t, dec_t = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = L.t3hip_encode_frame(px.ctypes.data_as(C.c_void_p), C.c_uint64(len(px)), C.byref(cfg), out.ctypes.data_as(C.c_void_p), C.c_uint64(len(out)), C.byref(n))
        enc_t.append(time.perf_counter() - t0); assert rc == 0
        t0 = time.perf_counter()
        rc = L.t3hip_decode_frame(fenc_bad.ctypes.data_as(C.c_void_p), C.c_uint64(len(fenc_bad)), C.byref(seen), back.ctypes.data_as(C.c_void_p), C.c_uint64(len(back)), C.byref(n))
        dec_t.append(time.perf_counter() - t0); assert rc == 0 and n.value == len(px)
    e, d = min(enc_t) * 1e3, min(dec_t) * 1e3
    moved = (6 * len(px) + out.nbytes, fenc_bad.nbytes + 6 * len(px))
    return {"encode_ms": round(e, 2), "decode_ms": round(d, 2), "mpix_s": round(len(px) / (e + d) / 1e3, 1),
            "encode_gb_s_over_pcie": round(moved[0] / e / 1e6, 1), "decode_gb_s_over_pcie": round(moved[1] / d / 1e6, 1),
            "note": "t3hip_encode_frame / t3hip_decode_frame on pageable host buffers (what the std::vector API binds): upload, kernels, download, sync; best of %d" % reps}


def rgb_path(t3, orc, torch, dev, cfg, fcfg, stream, n=60, warm=150):
    """The same frame geometry entering and leaving as RGB8 (BASELINE: "synthetic 8K RGB frames"): the io_image.hpp bridge fused into
    the encoder's phase 1 and the decoder's output stage (row f1; parity of the bridge against a reference build is UNPINNED --
    that header does not compile -- it is exact against the oracle's restatement on all 2^24 colours).  HIP events, sustained clock."""
    rgb = torch.from_numpy(orc.lcg_rgb(NPX, SEED0)).to(dev)
    n_enc = t3.encoded_words(NPX // 2, cfg); n_fenc = t3.encoded_words(NPX // 2, fcfg)
    out = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device=dev); fout = torch.zeros(n_fenc * 9 + 64, dtype=torch.uint8, device=dev)
    back = torch.zeros(3 * NPX + 64, dtype=torch.uint8, device=dev); ver = torch.zeros(2, dtype=torch.int32, device=dev)
    t3.encode_rgb_dev(rgb.data_ptr(), NPX, fcfg, fout.data_ptr(), n_fenc, stream)
    L = t3.plan(NPX // 2, fcfg)
    t3.inject_errors_dev(fout.data_ptr(), L.header_syms, L.body_syms // 26, 999, 3, stream)

    def timed(f):
        for _ in range(warm): f()
        torch.cuda.synchronize()
        e0, e1 = t3.Event(), t3.Event()
        e0.record(stream)
        for _ in range(n): f()
        e1.record(stream)
        return e0.elapsed_ms(e1) / n
    te = timed(lambda: t3.encode_rgb_dev(rgb.data_ptr(), NPX, cfg, out.data_ptr(), n_enc, stream))
    td = timed(lambda: t3.decode_rgb_async(fout.data_ptr(), n_fenc, fcfg, NPX, back.data_ptr(), ver.data_ptr(), stream))
    assert ver.cpu().tolist() == [0, 0]
    eb, db = 3 * NPX + 9 * n_enc, 9 * n_fenc + 3 * NPX
    return {"encode_ms": round(te, 4), "decode_ms": round(td, 4), "encode_mpix_s": round(NPX / te / 1e3, 1), "decode_mpix_s": round(NPX / td / 1e3, 1),
            "encode_roofline_frac": round(eb / (te * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "decode_roofline_frac": round(db / (td * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": {"encode": eb, "decode": db},
            "note": "RGB8 in / out, bridge fused into the codec kernels (one launch per direction; decode stream carries 0..3 errors per block); bound: VALU (float conversion), not HBM; bridge parity unpinned against the reference, exact against the oracle on all 2^24 colours"}


def self_launch(args, argv):
    """--gpus N > 1 without a launcher: start the N ranks as a child process tree.  Nothing here has touched the GPU."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the card needs about a hundred steps (40 ms of continuous work) to settle at its sustained clock; with 20 steps
    # after 3 of warm-up every kernel measured 10-15 % slower (profiles/r01/notes.md).  300 steps are 0.1 s of GPU time.
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-ms", type=float, default=80.0, help="untimed clock-settling work before the warm-up steps (0 = none)")
    ap.add_argument("--frames-per-rank", type=int, default=8, help="distinct frames resident per rank (BASELINE configs[3]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-rgb", action="store_true", help="skip the RGB-in / RGB-out side measurement")
    ap.add_argument("--encode-only", action="store_true", help="profiling aid: skip decode + index in the loop")
    ap.add_argument("--overlap-record", action="store_true", help="index record on a second HIP stream beside the decode (round 1's default; the round-2 decoder leaves the CRC kernel no registers to run beside it, so the record now follows on the main stream)")
    ap.add_argument("--serial", action="store_true", help="(default now) index record on the main stream")
    ap.add_argument("--event-every", type=int, default=1, help="HIP events around the encode / decode launches on every n-th timed step only (measurement aid: recorded around every launch, the four events cost the step ~17 us of 0.29 ms -- value rises by 3-5 %% with n = 4 -- but the sampled launch intervals then come out 1-3 us LONGER, so the default keeps them on every step)")
    ap.add_argument("--record-first", action="store_true", help="index record right behind the encode whose output it reads instead of behind the decode (measured: the CRC kernel gains 4 us from the warm memory-side cache, the decoder and the encoder lose 3 -- the step is the same)")
    ap.add_argument("--sync-decode", action="store_true", help="every frame through the synchronous decode entry (host-parsed header, two synchronisations per frame)")
    ap.add_argument("--no-verify", action="store_true", help="profiling aid for timing-only ablation builds (results are wrong by construction)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib as ol

    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("T3_BENCH_RANK_PROBE") == "1":       # CPU test of the self-launch: rendezvous over gloo, report, leave (no GPU call)
        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1]); dist.all_reduce(t)
        os.write(1, ("probe rank %d of %d sum %d\n" % (rank, world, int(t.item()))).encode())     # one write: two ranks share the pipe
        dist.destroy_process_group()
        return
    # T3_BENCH_REHEARSE_ONE_GPU=1: dry run of the N>1 control flow on a single card (every rank on cuda:0, the exchange over
    # gloo with CPU tensors: RCCL refuses two ranks on one device); the numbers it prints mean nothing, the point is that the
    # multi-rank path is exercised end to end
    rehearse = world > 1 and os.environ.get("T3_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # T3_BENCH_FORCE_DIST=1 (tests): run the N > 1 code path -- process group, RCCL communicator, warm-up and timed exchange -- with
    # whatever world size the launcher gave, also 1 (a one-GPU box can then exercise every collective call for real)
    force_dist = os.environ.get("T3_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=dev)
    t3 = ge.load_package()
    sf = __import__("ternary_image_codec_amd.superframe", fromlist=["x"])
    t3.init(local)
    comm, exchange_via = None, None
    if multi and not rehearse:                                         # the library's own RCCL communicator (t3hip_comm_create)
        # make_comm returns a communicator on every rank or None on every rank (probe, agree, then rendezvous: superframe.py); without
        # one everybody uses torch.distributed's all-gather (RCCL as well), so the scaling run still completes -- reported in config.exchange
        comm = sf.make_comm()
        exchange_via = "t3hip_index_allgather (RCCL, library entry)" if comm is not None else "torch.distributed.all_gather_into_tensor (RCCL; the library communicator could not be created on every rank)"
    elif rehearse:
        exchange_via = "gloo on CPU copies (one-card rehearsal)"
    orc = ol.oracle()
    cur = torch.cuda.current_stream()
    stream = cur.cuda_stream
    # the index record (CRC-32 of the coded frame) only depends on the encode; --overlap-record puts it on a second HIP stream
    s2 = torch.cuda.Stream(device=dev) if (args.overlap_record and not args.serial) else None
    enc_done, rec_done = torch.cuda.Event(), torch.cuda.Event()

    # ---- synthetic input, resident in HBM before any timing (SURVEY §8d generator; global frame f has seed 12345 + f) ----
    FPR = max(1, args.frames_per_rank)
    cfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)
    fcfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2, mode=t3.MODE_FIXED)
    n_enc = t3.encoded_words(NPX // 2, cfg); n_fenc = t3.encoded_words(NPX // 2, fcfg)
    L = t3.plan(NPX // 2, fcfg)
    frames = [rank + world * j for j in range(FPR)]
    px_host, d_px, d_fenc = [], [], []
    for f in frames:
        px = orc.lcg_pixels(NPX, SEED0 + f)
        px_host.append(px)
        d_px.append(torch.from_numpy(px.view(np.uint8)).to(dev))
        fe = torch.zeros(n_fenc * 9 + 64, dtype=torch.uint8, device=dev)
        t3.encode_frame_dev(d_px[-1].data_ptr(), NPX, fcfg, fe.data_ptr(), n_fenc, stream)
        t3.inject_errors_dev(fe.data_ptr(), L.header_syms, L.body_syms // 26, 777 + f, 3, stream)
        d_fenc.append(fe)
    d_enc = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device=dev)
    d_back = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device=dev)
    d_recs = torch.zeros((max(args.steps, 1), t3.FRAME_RECORD_BYTES), dtype=torch.uint8, device=dev)
    scr_bytes = t3.frame_record_scratch_bytes(n_enc)
    d_scr = torch.zeros(scr_bytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # the stream's first frame: synchronous entry, header parsed on the host -> the configuration the following frames are decoded with
    dctx = t3.DecoderContext(mode=t3.MODE_FIXED)
    rc0, n0 = t3.decode_profile_dev(d_fenc[0].data_ptr(), n_fenc, dctx.cfg_last_seen, d_back.data_ptr(), NPX, True, stream)
    assert args.no_verify or (rc0 == 0 and n0 == NPX), (rc0, n0)
    d_verdict = torch.zeros((max(args.steps, args.warmup, 1), 2), dtype=torch.int32, device=dev)
    d_back.zero_()

    dec_start = 3 if (args.record_first or s2 is not None) else 1      # which event opens the decoder's interval

    def step(i, ev=None):
        j = i % FPR
        if s2 is not None:
            cur.wait_event(rec_done)                        # the previous frame's record has read d_enc
        if ev is not None:
            ev[0].record(stream)
        t3.encode_frame_dev(d_px[j].data_ptr(), NPX, cfg, d_enc.data_ptr(), n_enc, stream)
        if ev is not None:
            ev[1].record(stream)
        if args.encode_only:
            return
        rec_args = (d_enc.data_ptr(), n_enc, i * world + rank, cfg, d_recs[i % len(d_recs)].data_ptr(), d_scr.data_ptr(), scr_bytes)
        if s2 is not None:
            enc_done.record(cur); s2.wait_event(enc_done)
            t3.frame_record_dev(*rec_args, s2.cuda_stream); rec_done.record(s2)
        if s2 is None and args.record_first:
            t3.frame_record_dev(*rec_args, stream)          # the record reads what the encoder has just written (still in the memory-side cache)
        if ev is not None and dec_start == 3:
            ev[3].record(stream)                            # the decoder's interval starts here (record in between; else it starts at ev[1]: one event less per step)
        if args.sync_decode:
            seen = t3.DecoderContext(mode=t3.MODE_FIXED).cfg_last_seen
            rc, n = t3.decode_profile_dev(d_fenc[j].data_ptr(), n_fenc, seen, d_back.data_ptr(), NPX, True, stream)
            assert args.no_verify or (rc == 0 and n == NPX), (rc, n)
        else:
            n = t3.decode_frame_async(d_fenc[j].data_ptr(), n_fenc, dctx.cfg_last_seen, NPX // 2, d_back.data_ptr(), NPX, d_verdict[i % len(d_verdict)].data_ptr(), True, stream)
            assert n == NPX
        if ev is not None:
            ev[2].record(stream)
        if s2 is None and not args.record_first:
            t3.frame_record_dev(*rec_args, stream)

    def exchange():
        """The one exchange step: the batch's index records, all-gathered (RCCL; gloo on CPU copies in the one-card rehearsal)."""
        if rehearse:
            return sf.gather_records_torch(d_recs.cpu())
        if comm is None:
            return sf.gather_records_torch(d_recs)
        return sf.gather_records(comm, d_recs)

    # clock settling, untimed and in addition to the W warm-up steps: the card needs tens of milliseconds of continuous work to
    # reach its sustained clock (profiles/r01/notes.md); without it a short run (small K and W) measures the ramp, 10-15 % slow
    settle_steps = 0
    if args.settle_ms > 0:
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            for _ in range(8):
                step(settle_steps % max(args.steps, 1)); settle_steps += 1
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    if s2 is not None:
        cur.wait_stream(s2)
    if multi and not args.encode_only:                     # untimed: RCCL builds its channels on the first collective of a communicator
        exchange(); torch.cuda.synchronize(); dist.barrier()
    torch.cuda.synchronize()
    events = [[t3.Event(), t3.Event(), t3.Event(), t3.Event()] for _ in range(args.steps)]
    gathered = None

    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i] if i % max(args.event_every, 1) == 0 else None)
    if s2 is not None:
        cur.wait_stream(s2)
    ex_ev = (t3.Event(), t3.Event())
    if multi and not args.encode_only:
        ex_ev[0].record(stream); t_ex = time.perf_counter()
        gathered = exchange()
        ex_ev[1].record(stream); exchange_host_ms = (time.perf_counter() - t_ex) * 1e3
    torch.cuda.synchronize()
    if not args.encode_only and not args.sync_decode:      # the streaming entry's verdicts: header as expected, no uncorrectable block
        verdicts = d_verdict[: args.steps].cpu().numpy()
        assert args.no_verify or not verdicts.any(), verdicts
    dt = time.perf_counter() - t0                          # (no barrier here: the MAX over ranks below is the job's time)
    if multi:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness of what was just timed (outside the timed region) ----
    j_last = (args.steps - 1) % FPR
    enc = d_enc[: n_enc * 9].cpu().numpy()
    if not args.no_verify:
        if frames[j_last] == 0:
            assert ol.fnv_hex(enc) == GOLD_HASH_C2, "encoded stream does not match the reference hash"
        else:
            # other frames have no pinned hash: band 0 of the frame's first 1/16 against the CPU restatement (whole super-tiles of
            # 540 px, so band 0 of the short frame is a prefix of band 0 of the full one: same body offsets, same scrambler phase)
            part = NPX // 16 // 540 * 540
            rc_o, want = orc.encode_frame(px_host[j_last][:part], ol.make_cfg(profile=2, uep=2), cap=part)
            nb0 = 26 * t3.plan(part // 2, cfg).band_blocks[0]
            assert rc_o == 0 and np.array_equal(enc[: 52 + nb0], want.reshape(-1)[: 52 + nb0]), "encoded stream differs from the oracle"
        if rank == 0 and frames[j_last] != 0:              # the pinned hash, on frame 0 (untimed extra launch)
            t3.encode_frame_dev(d_px[0].data_ptr(), NPX, cfg, d_enc.data_ptr(), n_enc, stream); torch.cuda.synchronize()
            assert ol.fnv_hex(d_enc[: n_enc * 9].cpu().numpy()) == GOLD_HASH_C2, "encoded stream does not match the reference hash"
            t3.encode_frame_dev(d_px[j_last].data_ptr(), NPX, cfg, d_enc.data_ptr(), n_enc, stream); torch.cuda.synchronize()
    if not args.encode_only:
        back = d_back[: NPX * 6].cpu().numpy().view(ol.PIXEL_DT)
        assert args.no_verify or np.array_equal(back, px_host[j_last]), "FIXED decode did not recover the frame"
        index = sf.assemble_index(gathered if multi else d_recs, 0)
        assert len(index) == world * args.steps and [r.frame_idx for r in index] == list(range(world * args.steps))
        mine = index[(args.steps - 1) * world + rank]          # records are dealt round-robin: record i * world + rank comes from this rank's step i
        assert args.no_verify or (mine.n_words == n_enc and mine.crc32 == orc.crc32(enc)), "frame index record does not match the payload"

    if comm is not None:
        comm.destroy()
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return
    ev_steps = [i for i in range(args.steps) if i % max(args.event_every, 1) == 0]
    enc_ms = [events[i][0].elapsed_ms(events[i][1]) for i in ev_steps]
    dec_ms = [events[i][dec_start].elapsed_ms(events[i][2]) for i in ev_steps] if not args.encode_only else [float("nan")]
    enc_avg = sum(enc_ms) / len(enc_ms); dec_avg = sum(dec_ms) / len(dec_ms)
    alg_bytes = 6 * NPX + 9 * n_enc                       # read 6 B/px, write 9 B/word = 385,966,134 B
    achieved = alg_bytes / (enc_avg * 1e-3) / 1e9

    def pmc(name):
        p = os.path.join(ROOT, "profiles", name)
        return json.load(open(p)).get("hbm_bytes_per_launch") if os.path.exists(p) else None

    out = {
        "metric": "Mpix/s encode+decode 8K RS(26,20)", "value": round(world * args.steps * NPX / dt / 1e6, 3), "unit": "Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "one 7680x4320 frame per rank per step (BASELINE configs[1]; %d distinct frames resident per rank, global frame f = rank + N j, LCG seed 12345 + f: configs[3]): fused encode COMPAT P3 RS(26,20) 1-D, then FIXED-mode decode of the same frame with 0..3 injected symbol errors per block (" % FPR + ("synchronous entry, header parsed on the host per frame" if args.sync_decode else "streaming entry: configuration from the stream's first frame, header symbols checked on the device") + "), then index record" + (" [encode only]" if args.encode_only else ""),
                   "frame_px": NPX, "coded_words": n_enc, "frames_per_rank": FPR, "settle_ms": args.settle_ms, "settle_steps": settle_steps, "event_every": max(args.event_every, 1),
                   "sharding": "frames per rank, no data-path collective; one all-gather of index records per batch", "exchange": exchange_via},
        "exchange_ms": (round(ex_ev[0].elapsed_ms(ex_ev[1]), 4) if (multi and not args.encode_only) else None),
        "exchange_host_ms": (round(exchange_host_ms, 4) if (multi and not args.encode_only) else None),
        "encode_ms": round(enc_avg, 4), "decode_ms": round(dec_avg, 4),
        "encode_mpix_s": round(NPX / enc_avg / 1e3, 1), "decode_mpix_s": round(NPX / dec_avg / 1e3, 1),
        "roofline": {"kernel": "encode_kernel_k<FE_PIXELS, 1-D, r=6>", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc("pmc_encode_latest.json"),
                     "traffic_source": "profiles/pmc_encode_latest.json: rocprofv3 --pmc passes of this command in their own runs (FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024, gfx950 correction), not measured in this run",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "launch_ms": round(enc_avg, 4), "launch_samples": len(enc_ms),
                     "note": "north-star kernel (SURVEY 8d); launch_ms = HIP events around the launch inside the timed loop, on every %d-th timed step (an event costs the stream ~4 us; --event-every 1: every step)" % max(args.event_every, 1)},
    }
    if not args.encode_only:      # the decoder is the longer kernel of the step: same definition, SURVEY 8d decode bytes + 6 B/px
        dec_bytes = 9 * n_fenc + 6 * NPX
        out["roofline_decode"] = {"kernel": "decode_fixed_px_kernel<r=6, pixels>" + (" (+ header read-back and failure-flag sync of the synchronous entry point)" if args.sync_decode else " (header check inside the launch)"),
                                  "bound": "hbm", "achieved": round(dec_bytes / (dec_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(dec_bytes / (dec_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": pmc("pmc_decode_latest.json"), "traffic_source": "profiles/pmc_decode_latest.json (separate rocprofv3 --pmc runs of this command)",
                                  "algorithmic_bytes_per_launch": dec_bytes, "launch_ms": round(dec_avg, 4), "launch_samples": len(dec_ms)}
    if world == 1 and not args.no_verify:                  # reported beside the line, on rank 0 at N=1 only
        if not args.encode_only and not args.no_rgb:
            out["rgb_path"] = rgb_path(t3, orc, torch, dev, cfg, fcfg, stream)
        if not args.no_end_to_end and not args.encode_only:
            out["end_to_end"] = end_to_end(t3, px_host[0], cfg, d_fenc[0][: n_fenc * 9].cpu().numpy().reshape(-1, 9))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(orc, ol, px_host[0])
    print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
