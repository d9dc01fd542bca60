#!/bin/bash
# run bench.py --encode-only against every libt3hip_*.so variant present (timing/diagnostic builds)
cd "$(dirname "$0")/.."
for so in ternary-image-codec_amd/libt3hip_*.so; do
  echo "== $so"
  T3HIP_LIB=$PWD/$so python bench.py --encode-only --no-cpu-baseline --no-verify --steps 20 --warmup 8 2>&1 | grep -E "stamps|encode_ms" | sed 's/.*"encode_ms": \([0-9.]*\).*/encode_ms \1/'
done
