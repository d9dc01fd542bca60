"""The C++ drop-in header (include/ternary_codec_v6.hpp) driven the way the reference's own CLI drives the reference
(old/src/main.cpp:15-27), compiled with the host compiler only and linked against libt3hip.so."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_demo(tmp):
    exe = os.path.join(tmp, "dropin_demo")
    lib = os.path.join(ROOT, "ternary-image-codec_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "dropin_demo.cpp"),
                    "-L" + lib, "-lt3hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def build_refnames(tmp, src=None, extra=()):
    exe = os.path.join(tmp, ("refnames_demo" if src is None else "ref_main_bare") + ("_x" if extra else ""))
    lib = os.path.join(ROOT, "ternary-image-codec_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", *extra, "-I" + os.path.join(ROOT, "include", "compat"), "-I" + os.path.join(ROOT, "include"),
                    src or os.path.join(ROOT, "tests", "cpp", "refnames_demo.cpp"), "-L" + lib, "-lt3hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    return exe


def test_reference_caller_compiles_against_dropin_header(t3, tmp_path):
    """The reference's own self-test runner (old/src/main_bare.cpp: includes "ternary_image_codec_v6_min.hpp", calls selftest_rs_unit
    and selftest_api_roundtrip) compiles UNCHANGED against include/compat + include/ternary_codec_v6.hpp with g++ alone, and so
    does tests/cpp/refnames_demo.cpp, which touches every other public name of the path.  Without a GPU the frame-level self-test
    answers false / T3_E_NODEVICE (no fallback); selftest_rs_unit walks single blocks, which are host arithmetic, and passes.  The
    reference source is read where it lies, only in this container."""
    import torch
    exe2 = build_refnames(str(tmp_path))
    ref_main = "/root/reference/old/src/main_bare.cpp"
    exe = build_refnames(str(tmp_path), ref_main) if os.path.exists(ref_main) else None
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    if exe:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 1 and r.stdout.strip() == "RS:OK API:FAIL"            # no device: every frame-level call returns false
    out = json.loads(subprocess.run([exe2], check=True, capture_output=True, text=True).stdout)
    assert out["selftest_rs_unit"] == 1 and out["selftest_api_roundtrip"] == 0
    assert out["word"] == [21, 5, 13, 7, 26, 1, 17, 20, 6] and out["unpack_equal"] == 1      # pack_two_pixels: host arithmetic
    # the host-side control arithmetic needs no device: field, generator polynomials, scrambler, beacon symbol, CRC-12
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")))
    assert out["primitive"] == 3 and out["order3"] == 26 and out["scr_back"] == 1 and out["tr2i"] == 200


@pytest.mark.gpu
def test_reference_names_on_device(gpu, orc, tmp_path):
    """tests/cpp/refnames_demo.cpp on the card, every value against the oracle / the golden vectors captured from the reference."""
    out = json.loads(subprocess.run([build_refnames(str(tmp_path))], check=True, capture_output=True, text=True).stdout)
    assert out["selftest_rs_unit"] == 1 and out["selftest_api_roundtrip"] == 1 and out["status"] == 0
    gfm = orc.gf_tables()
    fld = []
    for a in range(27):
        for b in range(27):
            ta = [(a // 3 ** i) % 3 for i in range(3)]; tb = [(b // 3 ** i) % 3 for i in range(3)]
            s_ = sum(((ta[i] + tb[i]) % 3) * 3 ** i for i in range(3)); d_ = sum(((ta[i] - tb[i]) % 3) * 3 ** i for i in range(3))
            m = int(gfm["mul"][a * 27 + b])
            fld += [s_, d_, m, m]
    for a in range(27):
        fld += [int(gfm["inv"][a]), int(gfm["log"][a]) & 0xFF]
    for e in range(-30, 60):
        fld.append(int(gfm["exp"][e % 26]))
    assert out["field_hash"] == ol.fnv_hex(np.array(fld, np.uint8)) and out["primitive"] == gfm["prim"] == 3 and out["order3"] == 26
    # RSCodec in COMPAT arithmetic = the reference's encode_block / decode_block (oracle pinned to it)
    rsv = []; dec_ok = 0
    for k in (24, 22, 20, 18):
        g = orc.rs_generator(k)
        data = np.array([(i * 5 + 7) % 27 for i in range(k)], np.uint8)
        code = orc.rs_encode_blocks(k, data, 0)[0]
        rsv += list(g) + list(code)
        for x in range(27):          # Horner, as poly_eval
            acc = 0
            for c in reversed(list(g)):
                acc = int(gfm["mul"][acc * 27 + x]); ta = [(acc // 3 ** i) % 3 for i in range(3)]; tc = [(int(c) // 3 ** i) % 3 for i in range(3)]
                acc = sum(((ta[i] + tc[i]) % 3) * 3 ** i for i in range(3))
            rsv.append(acc)
        bad = code.copy(); t3_ = [(int(bad[3]) // 3 ** i) % 3 for i in range(3)]; f5 = [2, 1, 0]
        bad[3] = sum(((t3_[i] + f5[i]) % 3) * 3 ** i for i in range(3))
        cw, dk, okv = orc.rs_decode_blocks(k, bad, 0)
        dec_ok = dec_ok * 2 + int(okv[0])
        rsv += list(cw[0]) + (list(dk[0]) if okv[0] else [77] * k)
    assert out["rs_hash"] == ol.fnv_hex(np.array(rsv, np.uint8)) and out["dec_ok"] == dec_ok
    scr = []
    for seed in ((1, 1, 1), (0xFFFFFFFF, 0xFFFFFFFE, 5)):
        scr += list(orc.scramble(np.array([i % 27 for i in range(60)], np.uint8), *seed, 0))
    assert out["scr_hash"] == ol.fnv_hex(np.array(scr, np.uint8)) and out["scr_back"] == 1
    bea = [orc.beacon_symbol(p, f, h) for p in range(5) for f in (0, 1, 2, 4, 8192 % 5) for h in range(3)]
    assert out["beacon_hash"] == ol.fnv_hex(np.array(bea, np.uint8))
    msg = np.array([(i * 7 + i // 5) % 3 for i in range(69)], np.uint8)
    assert out["crc12"] == "".join(str(int(x)) for x in orc.crc12(msg))
    il = np.array([(i * 11 + i // 7) % 27 for i in range(1000)], np.uint8)
    assert out["il_hash"] == ol.fnv_hex(orc.interleave2d(il, 7, 5, 0)) and out["il_back"] == 1
    assert out["word"] == [21, 5, 13, 7, 26, 1, 17, 20, 6] and out["unpack_equal"] == 1 and out["tr2i"] == 200


@pytest.mark.gpu
def test_selftests_in_reference_arithmetic(gpu, tmp_path):
    """-DT3_SELFTEST_REFERENCE_ARITHMETIC: selftest_rs_unit / selftest_api_roundtrip run the reference's statements in COMPAT arithmetic
    and return the reference's own verdicts (golden "selftests", captured from the reference build: false, false) -- identical
    results on the same inputs for these two entry points as well; the default build keeps FIXED arithmetic, where they pass."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")))
    out = json.loads(subprocess.run([build_refnames(str(tmp_path), extra=("-DT3_SELFTEST_REFERENCE_ARITHMETIC",))], check=True, capture_output=True, text=True).stdout)
    assert [bool(out["selftest_rs_unit"]), bool(out["selftest_api_roundtrip"])] == g["selftests"] == [False, False]
    assert out["status"] in (-5, -6)                                              # T3_E_HEADER / T3_E_RS: the reference's decoder rejects its encoder's stream


def test_dropin_header_compiles_and_refuses_without_gpu(t3, tmp_path):
    """CPU: the header builds with g++ alone; with no device every call returns false with T3_E_NODEVICE (no fallback)."""
    import torch
    exe = build_demo(str(tmp_path))
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    out = json.loads(subprocess.run([exe, "16", "16"], check=True, capture_output=True, text=True).stdout)
    assert out["ok_raw"] == 0 and out["ok_enc"] == 0 and out["ok_enc_fixed"] == 0 and out["status"] == -1


@pytest.mark.gpu
def test_dropin_caller_sequence(gpu, orc, tmp_path):
    exe = build_demo(str(tmp_path))
    out = json.loads(subprocess.run([exe, "256", "256", str(tmp_path)], check=True, capture_output=True, text=True).stdout)
    px = orc.lcg_pixels(256 * 256)
    raw = orc.pack_pixels(px)
    cfg = ol.make_cfg(profile=1, uep=1, tile=(64, 64), beacon=(83, 2, 1))
    rc, enc = orc.encode_profile(raw, cfg)
    assert out["ok_raw"] == 1 and out["raw_words"] == len(raw) and out["raw_hash"] == ol.fnv_hex(raw)
    assert out["ok_enc"] == 1 and out["enc_words"] == len(enc) and out["enc_hash"] == ol.fnv_hex(enc)
    seen = ol.make_cfg()
    rcd, _ = orc.decode_profile(enc, seen)
    assert out["ok_dec_compat"] == (1 if rcd == 0 else 0)
    if rcd == 0 or seen.profile != ol.make_cfg().profile:
        assert out["seen_profile"] == seen.profile
    assert out["ok_enc_fixed"] == 1 and out["ok_dec_fixed"] == 1 and out["roundtrip_equal"] == 1 and out["selftest_api_roundtrip"] == 1
    tr = orc.extract_subword_stream(raw, 24)                       # row f3 through the header's reference-named functions
    b243 = orc.ut_to_base243(tr)
    assert out["sub_trits"] == len(tr) and out["sub_hash"] == ol.fnv_hex(tr)
    assert out["b243_bytes"] == len(b243) and out["b243_hash"] == ol.fnv_hex(b243) and out["ok_b243"] == 1
    assert out["w24_hash"] == ol.fnv_hex(orc.build_words_from_subword_stream(tr, 24, 0))
    # row f2: the containers written by the C++ header are byte-identical to the independent restatement in test_containers.py
    assert out["containers_ok"] == 1
    from test_containers import expect_t3v
    cfgf = ol.make_cfg(profile=1, uep=1, tile=(64, 64), beacon=(83, 2, 1), mode=1)
    rcf, encf = orc.encode_profile(raw, cfgf); assert rcf == 0
    f0 = np.ascontiguousarray(enc).view(np.uint8).reshape(-1); f2 = np.ascontiguousarray(encf).view(np.uint8).reshape(-1)
    want = expect_t3v(27, 256, 256, [f0, np.zeros(0, np.uint8), f2], b'{"codec":"v6"}', [b'{"f":0}', b"", b'{"f":2}'])
    assert out["t3v_bytes"] == len(want) and out["t3v_hash"] == ol.fnv_hex(np.frombuffer(want, np.uint8))


def test_bench_self_launch_cpu():
    """`python bench.py --gpus 2` from a bare shell (no launcher, no RANK in the environment) starts its own two ranks before
    any GPU call; the probe mode stops each rank right after the gloo rendezvous, so this runs without a GPU."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["T3_BENCH_RANK_PROBE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    lines = sorted(re.findall(r"probe rank \d of \d sum \d", r.stdout))                # (the two ranks write to one pipe)
    assert lines == ["probe rank 0 of 2 sum 3", "probe rank 1 of 2 sum 3"]


@pytest.mark.gpu
def test_bench_two_rank_rehearsal(gpu):
    """bench.py's N>1 control flow (self-launch from a bare command line, rank-sharded distinct frames, the untimed warm-up
    exchange, one all-gather of index records, max-over-ranks timing, the index assertions) on one card: two ranks on cuda:0,
    the exchange over gloo (T3_BENCH_REHEARSE_ONE_GPU=1: RCCL refuses two ranks on one device).  The driver runs the real
    thing over RCCL on a whole node; test_comm_single_rank covers the RCCL entry points themselves."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["T3_BENCH_REHEARSE_ONE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "1", "--settle-ms", "5", "--frames-per-rank", "2"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 8 and line["scaling"] == "weak" and line["value"] > 0 and "cpu_baseline" not in line


@pytest.mark.gpu
def test_bench_distributed_path_one_rank(gpu):
    """bench.py's N > 1 code path with the real collectives on this box's one card: launched under torch.distributed.run with one
    rank and T3_BENCH_FORCE_DIST=1 it initialises the process group (gloo + RCCL), creates the library's RCCL communicator from a
    broadcast unique id, runs the warm-up exchange and the timed t3hip_index_allgather, and checks the assembled index."""
    import socket
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(T3_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--settle-ms", "5", "--frames-per-rank", "2",
                        "--no-cpu-baseline", "--no-end-to-end", "--no-rgb"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 8 and line["value"] > 0


@pytest.mark.gpu
def test_comm_single_rank(gpu):
    """The library's RCCL entry points on real hardware (a communicator of one rank: unique id, ncclCommInitRank,
    ncclAllGather of frame records on a stream, destroy)."""
    import torch
    t3 = gpu
    comm = t3.Comm(t3.comm_unique_id(), 1, 0)
    rec = torch.arange(3 * t3.FRAME_RECORD_BYTES, dtype=torch.int32).to(torch.uint8).reshape(3, -1).cuda()
    out = torch.zeros_like(rec)
    comm.index_allgather(rec.data_ptr(), 3, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, rec)
    comm.destroy()
