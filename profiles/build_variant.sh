#!/bin/bash
# Build a timing variant of the library: profiles/build_variant.sh <tag> "<-D flags>" [objects to rebuild ...]
#   -> ternary-image-codec_amd/libt3hip_<tag>.so (objects under csrc/abl_<tag>/; everything not listed is copied from the product build)
# Select it at run time with T3HIP_LIB=$PWD/ternary-image-codec_amd/libt3hip_<tag>.so (see profiles/variants_dec.sh).
set -e
cd "$(dirname "$0")/../ternary-image-codec_amd/csrc"
tag=$1; flags=$2; shift 2
objs=${@:-t3_decode_fused.o t3_api_decode.o t3_decode_stream.o t3_decode.o}
make -s -j6 >/dev/null
mkdir -p abl_$tag
cp -p *.o abl_$tag/
for o in $objs; do rm -f abl_$tag/$o; done
touch -d '+1 minute' $(ls abl_$tag/*.o) 2>/dev/null || true
make -s -j6 OUT=../libt3hip_$tag.so OBJDIR=abl_$tag EXTRA="$flags" 2>&1 | grep -E "error|Error" || true
ls -la ../libt3hip_$tag.so
