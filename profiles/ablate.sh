#!/bin/bash
# Timing-only ablation builds of the encode kernel (results are WRONG by construction; only the launch time matters).
# Usage (on the GPU box): bash profiles/ablate.sh > gpurun_out/ablate.log
set -e
cd "$(dirname "$0")/.."
for v in NO_P1 NO_P2 NO_P3 "NO_P1 -DT3_ABL_NO_P2" "NO_P2 -DT3_ABL_NO_P3" "NO_P1 -DT3_ABL_NO_P2 -DT3_ABL_NO_P3"; do
  tag=$(echo "$v" | tr -d ' ' | tr -c 'A-Za-z0-9_\n' '_')
  make -s -j8 -C ternary-image-codec_amd/csrc OUT=../libt3hip_$tag.so OBJDIR=abl_$tag EXTRA="-DT3_ABL_$v" >/dev/null 2>&1
  echo "== ablation $v"
  T3HIP_LIB=$PWD/ternary-image-codec_amd/libt3hip_$tag.so python bench.py --encode-only --no-cpu-baseline --no-verify --steps 20 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('encode_ms', d['encode_ms'])"
done
