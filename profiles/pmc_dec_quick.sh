#!/bin/bash
# Quick counter passes on the decode-only loop (profiles/dec_loop.py) for one library: bash profiles/pmc_dec_quick.sh <tag> [lib.so] [conf]
# Separate --pmc runs, never with tracing.  -> gpurun_out/r03/pmcq_<tag>.json
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}; lib=${2:-ternary-image-codec_amd/libt3hip.so}; conf=${3:-c2}
export T3HIP_LIB=$PWD/$lib
mkdir -p gpurun_out/r03
run() { name=$1; shift; rocprofv3 --pmc "$@" -d gpurun_out/r03/pmcq_${tag}_$name -o p --output-format csv -- python3 profiles/dec_loop.py errors 6 10 $conf > gpurun_out/r03/pmcq_${tag}_$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_I8
python3 - <<PY
import csv, collections, glob, json
out = {}
for f in glob.glob('gpurun_out/r03/pmcq_${tag}_*/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in ('decode_fixed', 'decode_stream', 'emit_stream', 'decode_uep', 'encode_kernel'):
            if key in r['Kernel_Name']: agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for key, cs in agg.items():
        for k, v in cs.items(): out.setdefault(key, {})[k] = sum(v) / len(v)
json.dump(out, open('gpurun_out/r03/pmcq_${tag}.json', 'w'), indent=1)
for key, cs in out.items():
    cyc = cs.get('SQ_BUSY_CYCLES', 0) / 32
    print(key, {k: round(v / 1e6, 2) for k, v in sorted(cs.items())}, 'cycles/SE-avg', round(cyc))
PY
rm -rf gpurun_out/r03/pmcq_${tag}_a gpurun_out/r03/pmcq_${tag}_b
