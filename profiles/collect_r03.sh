#!/bin/bash
# Round-3 evidence run on the GPU box (one gpurun call):  bash profiles/collect_r03.sh <tag>   -> gpurun_out/<tag>/...
#   1. GPU parity tests                                  pytest.log
#   2. bench.py, default flags                           bench_line.json          and at the driver's flags (--steps 20 --warmup 5): bench_line_driver_flags.json
#   3. rocprofv3 --kernel-trace --stats of bench.py      kernel_stats_bench.csv   (no counters in this pass)
#      ... of the encode entry for BASELINE configs[2]   kernel_stats_c3_encode.csv, of its decode: kernel_stats_c3_decode.csv
#   4. PMC passes, each group in its own run: encode (C2: pmc_encode.json -> pmc_encode_latest.json; C3: pmc_encode_c3.json),
#      decoder on the bench workload (pmc_decode.json -> pmc_decode_latest.json), configs[2] one-launch UEP decoder (pmc_decode_c3.json)
#   5. side measurements: other_configs.json, decode_configs.json, rgb_time.json, copy_ceiling.json
# Copy what is to be kept into profiles/r03/ afterwards.
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
tag=${1:-r03x}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
python3 bench.py 2> $out/bench.err | tail -1 > $out/bench_line.json; cut -c1-300 $out/bench_line.json
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --no-rgb 2>> $out/bench.err | tail -1 > $out/bench_line_driver_flags.json
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --no-cpu-baseline --no-end-to-end --no-rgb > $out/kt.log 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats_bench.csv 2>/dev/null || cp $(find $out/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats_bench.csv; head -8 $out/kernel_stats_bench.csv | cut -c1-170
echo progress: bench traced
rocprofv3 --kernel-trace --stats -d $out/kt_c3e -o kt --output-format csv -- python3 profiles/enc_loop.py c3 50 300 > $out/kt_c3e.log 2>&1
cp $(find $out/kt_c3e -name "*kernel_stats.csv" | head -1) $out/kernel_stats_c3_encode.csv; head -3 $out/kernel_stats_c3_encode.csv | cut -c1-170
rocprofv3 --kernel-trace --stats -d $out/kt_c3d -o kt --output-format csv -- python3 profiles/dec_loop.py errors 50 300 c3 > $out/kt_c3d.log 2>&1
cp $(find $out/kt_c3d -name "*kernel_stats.csv" | head -1) $out/kernel_stats_c3_decode.csv; head -4 $out/kernel_stats_c3_decode.csv | cut -c1-170
echo progress: c3 traced
bash profiles/pmc_encode.sh ${tag} > $out/pmc_encode.txt 2>&1
python3 - <<PY
import json
p = json.load(open('gpurun_out/pmc_${tag}_summary.json'))
# MI355X_MICROARCH.md, HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B: double it
fetch = p['FETCH_SIZE'] * 1024 * 2; write = p['WRITE_SIZE'] * 1024
o = {"kernel": "encode_kernel_k<FE_PIXELS, 1-D, r=6>", "fetch_bytes": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 correction", "counters": p}
json.dump(o, open('$out/pmc_encode.json', 'w'), indent=1); print({k: o[k] for k in ('fetch_bytes', 'write_bytes', 'hbm_bytes_per_launch')})
PY
echo progress: encode counters
bash profiles/pmc_dec2.sh ${tag}_dec errors c2 > $out/pmc_decode.txt 2>&1
python3 - <<PY
import json
p = json.load(open('gpurun_out/pmcd2_${tag}_dec_summary.json'))['decode_fixed']
o = {"kernel": "decode_fixed_px_kernel<6, pixels>", "workload": "bench workload: 8K FIXED RS(26,20) stream, 0..3 symbol errors in every block",
     "fetch_bytes": p['FETCH_SIZE'] * 2048, "write_bytes": p['WRITE_SIZE'] * 1024, "hbm_bytes_per_launch": p['hbm_bytes_per_launch'],
     "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 correction (calibrated for 16-byte-per-lane streaming loads: the decoder's loads are 16 bytes per lane at 26-byte stride)", "counters": p}
json.dump(o, open('$out/pmc_decode.json', 'w'), indent=1); print({k: o[k] for k in ('fetch_bytes', 'write_bytes', 'hbm_bytes_per_launch')})
PY
bash profiles/pmc_dec2.sh ${tag}_c3 errors c3 > $out/pmc_decode_c3.txt 2>&1; cp gpurun_out/pmcd2_${tag}_c3_summary.json $out/pmc_decode_c3.json
python3 - <<PY
import json
p = json.load(open('$out/pmc_decode_c3.json'))
tot = sum(v.get('hbm_bytes_per_launch', 0) for v in p.values())
print('configs[2] decode traffic per frame (all kernels):', tot, '= %.3f x the algorithmic 374,638,824 B' % (tot / 374638824.0), ' MFMA_I8:', {k: v.get('SQ_INSTS_VALU_MFMA_I8') for k, v in p.items()})
PY
echo progress: decode counters
run() { name=$1; shift; rocprofv3 --pmc "$@" -d $out/pmc_c3e_$name -o p --output-format csv -- python3 profiles/enc_loop.py c3 3 2 > $out/pmc_c3e_$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<PY
import csv, collections, glob, json
o = {}
for f in glob.glob('$out/pmc_c3e_*/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'encode_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): o[k] = sum(v) / len(v)
if 'FETCH_SIZE' in o and 'WRITE_SIZE' in o: o['hbm_bytes_per_launch'] = o['FETCH_SIZE'] * 2048 + o['WRITE_SIZE'] * 1024
json.dump({"kernel": "encode_kernel_uep<FE_PIXELS, 2-D> (BASELINE configs[2])", "counters": o}, open('$out/pmc_encode_c3.json', 'w'), indent=1); print(o.get('hbm_bytes_per_launch'))
PY
echo progress: c3 encode counters
python3 profiles/other_configs.py > $out/other_configs.json 2>> $out/side.err
python3 profiles/decode_configs.py > $out/decode_configs.json 2>> $out/side.err
python3 profiles/rgb_time.py > $out/rgb_time.json 2>> $out/side.err
python3 profiles/copy_ceiling.py > $out/copy_ceiling.json 2>> $out/side.err
python3 profiles/e2e_time.py > $out/e2e_time.txt 2>> $out/side.err; cat $out/e2e_time.txt
for m in clean errors; do for c in c2 beacon c3 words il ilwide uep1d k22; do python3 profiles/dec_loop.py $m 50 300 $c 2>/dev/null | tail -1; done; done > $out/dec_loop.txt; cat $out/dec_loop.txt
echo done
