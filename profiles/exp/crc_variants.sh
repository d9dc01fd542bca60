#!/bin/bash
# CRC / frame-record variants (one gpurun call): bash profiles/exp/crc_variants.sh <tag>
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/crc_variants_${1:-x}.txt; : > $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "frame_record or container or t3p or t3v" 2>&1 | tail -3 >> $out || { cat $out; exit 1; }
for v in "X=1" "T3HIP_RECORD_KERNEL=1" "T3HIP_CRC_BLOCKED=1" "T3HIP_CRC_WAVES_PER_SIMD=1" "T3HIP_CRC_I8=1"; do
  echo "== $v" >> $out; env $v timeout -k 10 120 python3 profiles/crc_time.py 2>&1 | grep frame_record >> $out || exit 1
done
cat $out
