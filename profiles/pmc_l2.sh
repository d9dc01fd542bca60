#!/bin/bash
# L2 / vector-L1 request counters of the encode kernel (separate passes; --pmc never combined with tracing domains)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}
run() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/pmc_${tag}_$name -o p --output-format csv -- python3 bench.py --encode-only --no-cpu-baseline --no-verify --steps 3 --warmup 1 > gpurun_out/pmc_${tag}_$name.log 2>&1 || echo "pass $name failed"; }
run tcc1 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum
run tcc2 TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum
run tcc3 TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum
run tcp1 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
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
