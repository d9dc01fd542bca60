#!/bin/bash
# SQ / TCC counter passes on the decode-only loop (profiles/dec_loop.py); separate runs per counter group, never with tracing.
# Usage on the GPU box: bash profiles/pmc_dec2.sh <tag> [clean|errors] [c2|c3|beacon|words] -> gpurun_out/pmcd2_<tag>_summary.json
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}; mode=${2:-errors}; conf=${3:-c2}
run() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/pmcd2_${tag}_$name -o p --output-format csv -- python3 profiles/dec_loop.py $mode 3 2 $conf > gpurun_out/pmcd2_${tag}_$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run sq3 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
python3 - <<PY
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmcd2_${tag}_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for key in ('decode_fixed', 'decode_uep', 'uep_edge', 'decode_stream', 'emit_stream', 'debeacon'):
                if key in r['Kernel_Name']: agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in agg.items():
            for k, v in cs.items(): out.setdefault(key, {})[k] = sum(v) / len(v)
for key, c in out.items():
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:   # MI355X_MICROARCH.md, HBM: KiB units; gfx950 FETCH_SIZE counts 128-B requests at 64 B: double it
        c['hbm_bytes_per_launch'] = c['FETCH_SIZE'] * 1024 * 2 + c['WRITE_SIZE'] * 1024
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmcd2_${tag}_summary.json', 'w'), indent=1)
PY
