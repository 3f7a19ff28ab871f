#!/bin/bash
# Builds variants of the zstd kernels (device/zstd2.hpp switches, given as name:flags arguments) as whole libraries next to the shipped one:
# datafusion-comet_amd/variants/libcomet_<name>.so (git-ignored, they travel to the GPU box with gpurun).  Run here (hipcc cross-compiles),
# then tools/gpu_zstd_variants.sh on the box.  The shipped libcomet.so is rebuilt unchanged at the end.
set -e
cd "$(dirname "$0")/../datafusion-comet_amd/csrc"
mkdir -p ../variants
build() {   # name, flags
  touch zstd2_kernels.hip
  make -s -j8 OUT=../variants/libcomet_$1.so EXTRA_HIPFLAGS="$2"
  echo "built variants/libcomet_$1.so ($2)"
}
for v in "$@"; do build "${v%%:*}" "${v#*:}"; done     # name:flags pairs, e.g.  round3:-DZS_SEQ_SPLIT=0
touch zstd2_kernels.hip
make -s -j8
echo "shipped libcomet.so rebuilt"
