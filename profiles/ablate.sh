#!/bin/bash
# Timing-only ablation builds of the encode kernel (results are WRONG by construction; only the launch time matters).
# Build here (no GPU needed):  bash profiles/ablate.sh build      -> ternary-image-codec_amd/libt3hip_abl_*.so
# Time on the GPU box:         bash profiles/variants.sh
set -e
cd "$(dirname "$0")/.."
for v in NO_P1 NO_P2 NO_STORE NO_MFMA NO_PREFETCH "NO_P1 -DT3_ABL_NO_P2" "NO_P1 -DT3_ABL_NO_P2 -DT3_ABL_NO_PREFETCH"; do
  tag=$(echo "$v" | sed 's/ -DT3_ABL_/_/g')
  make -s -j8 -C ternary-image-codec_amd/csrc OUT=../libt3hip_abl_$tag.so OBJDIR=abl_$tag EXTRA="-DT3_ABL_$v" 2>&1 | grep -E "error" || true
done
ls ternary-image-codec_amd/libt3hip_abl_*.so
