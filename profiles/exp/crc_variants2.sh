#!/bin/bash
# CRC kernel build variants (profiles/build_variant.sh crc_* ...) x waves per SIMD: bash profiles/exp/crc_variants2.sh <tag>
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; out=gpurun_out/crc_variants2_${1:-x}.txt; : > $out
for so in ternary-image-codec_amd/libt3hip.so ternary-image-codec_amd/libt3hip_crc_*.so; do
  for w in 1 2; do
    echo "== $so wps=$w" >> $out
    T3HIP_LIB=$PWD/$so T3HIP_CRC_WAVES_PER_SIMD=$w timeout -k 10 120 python3 profiles/crc_time.py 2>&1 | grep frame_record >> $out || exit 1
  done
done
cat $out
