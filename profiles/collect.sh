#!/bin/bash
# Evidence run on the GPU box (one gpurun call):  bash profiles/collect.sh <tag>
#   1. GPU parity tests                               -> gpurun_out/<tag>_pytest.log
#   2. bench.py, default flags (incl. cpu_baseline)   -> gpurun_out/<tag>_bench.json
#   3. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/<tag>_kernel_stats.csv (record overlapped with decode, as benched)
#      and of bench.py --serial                       -> gpurun_out/<tag>_kernel_stats_serial.csv (every kernel alone); no counters in these passes
#   4. PMC passes on the encode kernel (own passes)   -> gpurun_out/pmc_<tag>_summary.json + gpurun_out/<tag>_pmc_encode_latest.json
#      and on the decoder / CRC kernels               -> gpurun_out/pmcd_<tag>_summary.json
# Copy what is to be kept into profiles/ afterwards.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-x}
timeout -k 10 600 python3 -m pytest tests -m gpu -q > gpurun_out/${tag}_pytest.log 2>&1; tail -2 gpurun_out/${tag}_pytest.log
python3 bench.py 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench.json; cut -c1-400 gpurun_out/${tag}_bench.json
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_kt.log 2>&1
cp gpurun_out/${tag}_kt/kt_kernel_stats.csv gpurun_out/${tag}_kernel_stats.csv 2>/dev/null; head -7 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
# the same with the index record on the main stream: every kernel runs alone (its stand-alone duration)
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_kts -o kt --output-format csv -- python3 bench.py --serial --no-cpu-baseline > gpurun_out/${tag}_kts.log 2>&1
cp gpurun_out/${tag}_kts/kt_kernel_stats.csv gpurun_out/${tag}_kernel_stats_serial.csv 2>/dev/null; head -6 gpurun_out/${tag}_kernel_stats_serial.csv | cut -c1-160
bash profiles/pmc_encode.sh ${tag} > gpurun_out/${tag}_pmc.txt 2>&1
bash profiles/pmc_decode.sh ${tag} > gpurun_out/${tag}_pmcd.txt 2>&1      # SQ counters of the decoder / CRC kernels -> gpurun_out/pmcd_${tag}_summary.json
python3 - <<PY
import json
p = json.load(open('gpurun_out/pmc_${tag}_summary.json'))
# MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B: double it
fetch = p['FETCH_SIZE'] * 1024 * 2; write = p['WRITE_SIZE'] * 1024
out = {"kernel": "encode_kernel_k<FE_PIXELS, 1-D, r=6>", "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 correction", "counters": p}
json.dump(out, open('gpurun_out/${tag}_pmc_encode_latest.json', 'w'), indent=1); print(json.dumps({k: out[k] for k in ('fetch_bytes', 'write_bytes', 'hbm_bytes_per_launch')}))
PY
