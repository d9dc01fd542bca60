#!/bin/bash
# LDS counters of the encode kernel for the product build and every timing-only variant present
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for so in ternary-image-codec_amd/libt3hip.so ternary-image-codec_amd/libt3hip_*.so; do
  tag=$(basename $so .so)
  T3HIP_LIB=$PWD/$so rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES -d gpurun_out/pmcv_$tag -o p --output-format csv -- python3 bench.py --encode-only --no-cpu-baseline --no-verify --steps 3 --warmup 1 > gpurun_out/pmcv_$tag.log 2>&1
  python3 - <<PY
import csv, collections, glob
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmcv_$tag/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'encode_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$tag', {k: round(sum(v)/len(v)/12480) for k, v in sorted(agg.items())}, '(per tile)')
PY
done
