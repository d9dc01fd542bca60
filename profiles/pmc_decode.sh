#!/bin/bash
# PMC passes on the fused decoder (separate runs per counter group; --pmc is never combined with tracing domains).
# Usage on the GPU box: bash profiles/pmc_decode.sh <tag>   -> gpurun_out/pmcd_<tag>_summary.json
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}
run() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/pmcd_${tag}_$name -o p --output-format csv -- python3 bench.py --serial --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmcd_${tag}_$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run sq3 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
python3 - <<PY
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmcd_${tag}_*/')):
    for f in glob.glob(d + '*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            for key in ('decode_fixed', 'crc_mfma', 'crc_chunks', 'encode_kernel'):
                if key in r['Kernel_Name']: agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        for key, cs in agg.items():
            for k, v in cs.items(): out.setdefault(key, {})[k] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmcd_${tag}_summary.json', 'w'), indent=1)
PY
