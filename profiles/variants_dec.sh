#!/bin/bash
# run the full bench step (encode + decode + record) against every libt3hip_*.so variant present and print decode_ms
cd "$(dirname "$0")/.."
for so in ternary-image-codec_amd/libt3hip.so ternary-image-codec_amd/libt3hip_*.so; do
  echo "== $so"
  T3HIP_LIB=$PWD/$so python bench.py --serial --no-cpu-baseline --no-verify --steps 10 --warmup 3 2>&1 | grep -E "stamps|decode_ms" | sed 's/.*"encode_ms": \([0-9.]*\), "decode_ms": \([0-9.]*\).*/encode_ms \1 decode_ms \2/'
done
