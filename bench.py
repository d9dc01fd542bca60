#!/usr/bin/env python3
"""bench.py — headline benchmark of the Word27 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One step = one synthetic 8K (7680x4320) frame through the hot path on each rank, inputs already resident in HBM:
  1. fused encode, COMPAT arithmetic, P3 / RS(26,20) on all 9 bands, 1-D (BASELINE configs[1]; output hash pinned to the
     reference's b6c43f2f4aa44763),
  2. decode of the same frame's FIXED-mode (v6c) stream carrying 0..3 injected symbol errors in every RS block
     (BASELINE configs[4] semantics; exact recovery of the pixels is asserted after the timed region).  The stream's first
     frame is decoded before the timed region through the synchronous, reference-shaped entry (header read back and
     parsed on the host); the timed frames use the streaming entry t3hip_decode_frame_async: body decode launched with
     the configuration last seen, header symbols checked on the device, verdict words read inside the timed region
     after the last step (--sync-decode times the synchronous entry for every frame instead),
  3. the frame's index record (CRC-32 + header symbols) for the T3V-style super-frame index; it depends only on step 1 and
     runs on a second HIP stream under the decode (--serial puts it back on the main stream).
Frames are independent, so ranks shard them with no data-path collective (weak scaling); the only exchange is one
all-gather of the K fixed-size index records per rank at the end of the batch (RCCL), inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement; roofline/cpu_baseline objects included)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 7680, 4320
NPX = W * H
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
GOLD_HASH_C2 = "b6c43f2f4aa44763"


def cpu_baseline(orc, ol, px, budget_s=30.0):
    """The reference's own encoder (oracle/_ref, built from /root/reference in place) on one full 8K frame, 1 thread;
    decode leg: the C port of the FIXED decoder on a 1/2-frame sample (the reference cannot decode its own streams)."""
    import numpy as np
    cfg = ol.make_cfg(profile=2, uep=2)
    kind = "port"
    if ol.have_ref():
        ref = ol.Ref(); kind = "reference"
        t0 = time.perf_counter(); rc, enc = ref.encode_frame(px, cfg, cap=len(px)); t_enc = time.perf_counter() - t0
        enc_px = len(px)
    else:
        sample = px[: NPX // 8]
        t0 = time.perf_counter(); rc, enc = orc.encode_frame(sample, cfg, cap=len(sample)); t_enc = time.perf_counter() - t0
        enc_px = len(sample)
    assert rc == 0
    enc_mpix = enc_px / t_enc / 1e6
    sample = px[: NPX // 2]
    fcfg = ol.make_cfg(profile=2, uep=2, mode=1)
    rc, fenc = orc.encode_frame(sample, fcfg, cap=len(sample))
    nblk = (len(fenc) * 9 - 90) // 26
    bad = orc.inject_errors(fenc, 90, nblk, 777, 3)
    t0 = time.perf_counter(); rc, back = orc.decode_frame(bad, ol.make_cfg(mode=1)); t_dec = time.perf_counter() - t0
    assert rc == 0 and np.array_equal(back, sample)
    dec_mpix = len(sample) / t_dec / 1e6
    both = 1.0 / (1.0 / enc_mpix + 1.0 / dec_mpix)
    return {"value": round(both, 4), "unit": "Mpix/s", "cores": 1, "kind": kind,
            "sample": "encode: %s encoder on %d px of the 8K frame (%.1f s, %.3f Mpix/s); decode: C port of the FIXED decoder on the first 1/2 frame with injected errors (%.1f s, %.3f Mpix/s); value = 1/(1/enc+1/dec)"
                      % ("unmodified reference" if kind == "reference" else "C port of the reference", enc_px, t_enc, enc_mpix, t_dec, dec_mpix),
            "encode_mpix_s": round(enc_mpix, 4), "decode_mpix_s": round(dec_mpix, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the card needs about a hundred steps (40 ms of continuous work) to settle at its sustained clock; with 20 steps
    # after 3 of warm-up every kernel measured 10-15 % slower (profiles/r01/notes.md).  300 steps are 0.1 s of GPU time.
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--settle-ms", type=float, default=80.0, help="untimed clock-settling work before the warm-up steps (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encode-only", action="store_true", help="profiling aid: skip decode + index in the loop")
    ap.add_argument("--serial", action="store_true", help="index record on the main stream instead of overlapping it with the decode")
    ap.add_argument("--sync-decode", action="store_true", help="every frame through the synchronous decode entry (host-parsed header, two synchronisations per frame)")
    ap.add_argument("--no-verify", action="store_true", help="profiling aid for timing-only ablation builds (results are wrong by construction)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_lib as ol

    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    # T3_BENCH_REHEARSE_ONE_GPU=1: dry run of the N>1 control flow on a single card (every rank on cuda:0, collectives over
    # gloo with CPU tensors); the numbers it prints mean nothing, the point is that the multi-rank path is exercised end to end
    rehearse = world > 1 and os.environ.get("T3_BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    t3 = ge.load_package()
    sf = __import__("ternary_image_codec_amd.superframe", fromlist=["x"])
    t3.init(local)
    orc = ol.oracle()
    cur = torch.cuda.current_stream()
    stream = cur.cuda_stream
    # the index record (CRC-32 of the coded frame) only depends on the encode: it runs on a second HIP stream under the decode
    s2 = None if args.serial else torch.cuda.Stream(device=torch.device("cuda", local))
    enc_done, rec_done = torch.cuda.Event(), torch.cuda.Event()
    dev = torch.device("cuda", local)

    # ---- synthetic input, resident in HBM before any timing (SURVEY §8d generator; frame seed = 12345 + rank) ----
    px = orc.lcg_pixels(NPX, 12345 + rank)
    d_px = torch.from_numpy(px.view(np.uint8)).to(dev)
    cfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2)
    fcfg = t3.make_cfg(profile=t3.ProfileID.P3_RS26_20, uep=2, mode=t3.MODE_FIXED)
    n_enc = t3.encoded_words(NPX // 2, cfg); n_fenc = t3.encoded_words(NPX // 2, fcfg)
    d_enc = torch.zeros(n_enc * 9 + 64, dtype=torch.uint8, device=dev)
    d_fenc = torch.zeros(n_fenc * 9 + 64, dtype=torch.uint8, device=dev)
    d_back = torch.zeros(NPX * 6 + 64, dtype=torch.uint8, device=dev)
    d_recs = torch.zeros((max(args.steps, 1), t3.FRAME_RECORD_BYTES), dtype=torch.uint8, device=dev)
    d_scr = torch.zeros(64, dtype=torch.uint8, device=dev)
    L = t3.plan(NPX // 2, fcfg)
    t3.encode_frame_dev(d_px.data_ptr(), NPX, fcfg, d_fenc.data_ptr(), n_fenc, stream)
    t3.inject_errors_dev(d_fenc.data_ptr(), L.header_syms, L.body_syms // 26, 777 + rank, 3, stream)
    torch.cuda.synchronize()
    # the stream's first frame: synchronous entry, header parsed on the host -> the configuration the following frames are decoded with
    dctx = t3.DecoderContext(mode=t3.MODE_FIXED)
    rc0, n0 = t3.decode_profile_dev(d_fenc.data_ptr(), n_fenc, dctx.cfg_last_seen, d_back.data_ptr(), NPX, True, stream)
    assert args.no_verify or (rc0 == 0 and n0 == NPX), (rc0, n0)
    d_verdict = torch.zeros((max(args.steps, args.warmup, 1), 2), dtype=torch.int32, device=dev)
    d_back.zero_()

    def step(i, ev=None):
        if s2 is not None:
            cur.wait_event(rec_done)                        # the previous frame's record has read d_enc
        if ev is not None:
            ev[0].record(stream)
        t3.encode_frame_dev(d_px.data_ptr(), NPX, cfg, d_enc.data_ptr(), n_enc, stream)
        if ev is not None:
            ev[1].record(stream)
        if args.encode_only:
            return
        rec_args = (d_enc.data_ptr(), n_enc, i * world + rank, cfg, d_recs[i % len(d_recs)].data_ptr(), d_scr.data_ptr(), 64)
        if s2 is not None:
            enc_done.record(cur); s2.wait_event(enc_done)
            t3.frame_record_dev(*rec_args, s2.cuda_stream); rec_done.record(s2)
        if args.sync_decode:
            seen = t3.DecoderContext(mode=t3.MODE_FIXED).cfg_last_seen
            rc, n = t3.decode_profile_dev(d_fenc.data_ptr(), n_fenc, seen, d_back.data_ptr(), NPX, True, stream)
            assert args.no_verify or (rc == 0 and n == NPX), (rc, n)
        else:
            n = t3.decode_frame_async(d_fenc.data_ptr(), n_fenc, dctx.cfg_last_seen, NPX // 2, d_back.data_ptr(), NPX, d_verdict[i % len(d_verdict)].data_ptr(), True, stream)
            assert n == NPX
        if ev is not None:
            ev[2].record(stream)
        if s2 is None:
            t3.frame_record_dev(*rec_args, stream)

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
    torch.cuda.synchronize()
    events = [[t3.Event(), t3.Event(), t3.Event()] for _ in range(args.steps)]
    gathered = None

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, events[i])
    if s2 is not None:
        cur.wait_stream(s2)
    if world > 1 and not args.encode_only:
        gathered = sf.gather_records(d_recs.cpu() if rehearse else d_recs)   # the one exchange step: super-frame index records (RCCL all-gather)
    torch.cuda.synchronize()
    if not args.encode_only and not args.sync_decode:      # the streaming entry's verdicts: header as expected, no uncorrectable block
        verdicts = d_verdict[: args.steps].cpu().numpy()
        assert args.no_verify or not verdicts.any(), verdicts
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- correctness of what was just timed (outside the timed region) ----
    enc = d_enc[: n_enc * 9].cpu().numpy()
    if rank == 0 and not args.no_verify:
        assert ol.fnv_hex(enc) == GOLD_HASH_C2, "encoded stream does not match the reference hash"
    if not args.encode_only:
        back = d_back[: NPX * 6].cpu().numpy().view(ol.PIXEL_DT)
        assert args.no_verify or np.array_equal(back, px), "FIXED decode did not recover the frame"
        index = sf.assemble_index(gathered if world > 1 else d_recs, 0)
        assert len(index) == world * args.steps and [r.frame_idx for r in index] == list(range(world * args.steps))
        mine = index[(args.steps - 1) * world + rank]          # frames are dealt round-robin: frame f lives on rank f % world
        assert mine.n_words == n_enc and mine.crc32 == orc.crc32(enc), "frame index record does not match the payload"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    enc_ms = [events[i][0].elapsed_ms(events[i][1]) for i in range(args.steps)]
    dec_ms = [events[i][1].elapsed_ms(events[i][2]) for i in range(args.steps)] if not args.encode_only else [float("nan")]
    enc_avg = sum(enc_ms) / len(enc_ms); dec_avg = sum(dec_ms) / len(dec_ms)
    alg_bytes = 6 * NPX + 9 * n_enc                       # read 6 B/px, write 9 B/word = 385,966,134 B
    achieved = alg_bytes / (enc_avg * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_encode_latest.json")
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
    out = {
        "metric": "Mpix/s encode+decode 8K RS(26,20)", "value": round(world * args.steps * NPX / dt / 1e6, 3), "unit": "Mpix/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "one 7680x4320 frame per rank per step (BASELINE configs[1]): fused encode COMPAT P3 RS(26,20) 1-D, then FIXED-mode decode of the same frame with 0..3 injected symbol errors per block (" + ("synchronous entry, header parsed on the host per frame" if args.sync_decode else "streaming entry: configuration from the stream's first frame, header symbols checked on the device") + "), then index record" + (" [encode only]" if args.encode_only else ""),
                   "frame_px": NPX, "coded_words": n_enc, "settle_ms": args.settle_ms, "settle_steps": settle_steps, "sharding": "frames per rank, no data-path collective; one all-gather of index records per batch"},
        "encode_ms": round(enc_avg, 4), "decode_ms": round(dec_avg, 4),
        "encode_mpix_s": round(NPX / enc_avg / 1e3, 1), "decode_mpix_s": round(NPX / dec_avg / 1e3, 1),
        "roofline": {"kernel": "encode_kernel_k<FE_PIXELS, 1-D, r=6>", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                     "launch_ms": round(enc_avg, 4),
                     "note": "north-star kernel (SURVEY 8d); a plain device copy of the same volume (profiles/copy_ceiling.py) reaches 5.19 TB/s = 0.65 of peak on this part"},
    }
    if not args.encode_only:      # the decoder is the longer kernel of the step: same definition, SURVEY 8d decode bytes + 6 B/px
        dec_bytes = 9 * n_fenc + 6 * NPX
        out["roofline_decode"] = {"kernel": "decode_fixed_kernel<r=6, to_pixels>" + (" (+ header read-back and failure-flag sync of the synchronous entry point)" if args.sync_decode else " (+ header check kernel; the index record's CRC kernel runs beside it)"),
                                  "bound": "hbm", "achieved": round(dec_bytes / (dec_avg * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(dec_bytes / (dec_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                                  "algorithmic_bytes_per_launch": dec_bytes, "launch_ms": round(dec_avg, 4)}
    if not args.no_cpu_baseline and world == 1:        # reported baseline, timed on rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(orc, ol, px)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
