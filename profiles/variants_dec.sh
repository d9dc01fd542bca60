#!/bin/bash
# decode-only loop (profiles/dec_loop.py: streaming entry, exactness checked) against the product library and every libt3hip_*.so
# variant present (profiles/build_variant.sh): clean stream and the bench workload (0..3 errors in every block)
cd "$(dirname "$0")/.."
for so in ternary-image-codec_amd/libt3hip.so ternary-image-codec_amd/libt3hip_*.so; do
  echo "== $so"
  for mode in errors clean; do
    T3HIP_LIB=$PWD/$so timeout -k 10 120 python profiles/dec_loop.py $mode 50 300 ${1:-c2} 2>&1 | grep -E "dec_loop|stamps|Error|error" | tail -3
  done
done
