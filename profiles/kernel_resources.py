"""Per-kernel register / spill / LDS figures of the built library's gfx950 code objects (hipcc cross-compiles without a GPU):
    python3 profiles/kernel_resources.py [substring ...]
Reads the objects under ternary-image-codec_amd/csrc/*.o: .hip_fatbin -> clang-offload-bundler -> llvm-readelf --notes.
Used by tests/test_host_logic.py::test_no_vgpr_spills_in_hot_kernels and by hand while budgeting registers."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"


def kernel_notes(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        r = subprocess.run([BIN + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True)
        if r.returncode or not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return {}
        r = subprocess.run([BIN + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], capture_output=True)
        if r.returncode:
            return {}
        txt = subprocess.run([BIN + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" or (k == "args"):
            cur = {} if k == "agpr_count" else cur
        if cur is None:
            cur = {}
        if k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size"):
            cur[k] = v
        if k == "wavefront_size":                      # last key of a kernel record
            if "name" in cur:
                out[cur["name"]] = cur
            cur = None
    return out


def demangle(names):
    for tool in (BIN + "/llvm-cxxfilt", "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            if r.returncode == 0 and len(r.stdout.splitlines()) == len(names):
                return dict(zip(names, r.stdout.splitlines()))
        except OSError:
            pass
    return {n: n for n in names}


def all_kernels():
    d = os.path.join(ROOT, "ternary-image-codec_amd", "csrc")
    res = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".o"):
            res.update(kernel_notes(os.path.join(d, f)))
    dm = demangle(list(res))
    return {dm[k]: v for k, v in res.items()}


if __name__ == "__main__":
    ks = all_kernels()
    for name in sorted(ks):
        if len(sys.argv) > 1 and not any(s in name for s in sys.argv[1:]):
            continue
        v = ks[name]
        print("%-90s vgpr %3s spill %2s | sgpr %3s spill %2s | scratch %s" % (name[:90], v.get("vgpr_count"), v.get("vgpr_spill_count"), v.get("sgpr_count"), v.get("sgpr_spill_count"), v.get("private_segment_fixed_size")))


def loop_scratch(obj_names=("t3_kernels.o", "t3_decode_fused.o")):
    """Scratch (spill) accesses INSIDE the persistent tile loop of every kernel that has one: {kernel: (loads, stores)}.
    The tile loop = the widest backward-branch range that holds a matrix instruction (llvm-objdump of the gfx950 code object: the
    reload of a spilled register is followed by s_waitcnt vmcnt(0), which also drains the next tile's prefetch -- profiles/r02/notes.md)."""
    d = os.path.join(ROOT, "ternary-image-codec_amd", "csrc")
    res = {}
    for f in obj_names:
        with tempfile.TemporaryDirectory() as td:
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
            subprocess.run([BIN + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, os.path.join(d, f)], check=True, capture_output=True)
            subprocess.run([BIN + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True, capture_output=True)
            txt = subprocess.run([BIN + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
        cur, ins = None, []
        funcs = {}
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1); funcs[cur] = []; continue
            m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
            if m and cur:
                funcs[cur].append((int(m.group(2), 16), m.group(1), m.group(3)))
        for name, body in funcs.items():
            if not body:
                continue
            loops = []
            for addr, text, tail in body:
                if text.startswith("s_cbranch") or text.startswith("s_branch"):
                    m = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", tail)
                    tgt = None
                    if m:
                        tgt = body[0][0] + int(m.group(1), 16)
                    if tgt is not None and tgt <= addr:
                        loops.append((tgt, addr))
            mf = [a for a, t, _ in body if "v_mfma" in t]
            best = None
            for lo, hi in loops:
                if any(lo <= a <= hi for a in mf) and (best is None or hi - lo > best[1] - best[0]):
                    best = (lo, hi)
            if best is None:
                continue
            ld = sum(1 for a, t, _ in body if best[0] <= a <= best[1] and t.startswith("scratch_load"))
            st = sum(1 for a, t, _ in body if best[0] <= a <= best[1] and t.startswith("scratch_store"))
            res[name] = (ld, st)
    dm = demangle(list(res))
    return {dm[k]: v for k, v in res.items()}
