#!/bin/bash
# PMC passes on the encode kernel (separate runs per counter group; --pmc is never combined with tracing domains).
# Usage on the GPU box: bash profiles/pmc_encode.sh <tag>   -> gpurun_out/pmc_<tag>_*.csv
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}
run() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/pmc_${tag}_$name -o p --output-format csv -- python3 bench.py --encode-only --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/pmc_${tag}_$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE GRBM_GUI_ACTIVE
python3 - <<PY
import csv, collections, glob, json
out = {}
for d in sorted(glob.glob('gpurun_out/pmc_${tag}_*/')):
    for f in glob.glob(d + '*counter_collection.csv'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'encode_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items(): out[k] = sum(v) / len(v)
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmc_${tag}_summary.json', 'w'), indent=1)
PY
