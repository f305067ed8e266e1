#!/bin/bash
# Snapshot the current kernels as avsiam_amd/csrc/ab_<name>.so so that two builds can be timed back to back on ONE box
# (box-to-box clocks differ by ~5 %):   tools/ab_lib.sh old; <edit>; tools/ab_lib.sh new;
#   gpurun -- 'for v in old new old new; do AVSIAM_HIP_LIB=$PWD/avsiam_amd/csrc/ab_$v.so python tools/bench_epilogue.py; done'
set -e
cd "$(dirname "$0")/.."
# (a build of its own: the product library is not touched; AVSIAM_HIPCC_EXTRA adds flags, e.g. -DNT8_ABLATE=2)
python -m avsiam_amd.build --out "avsiam_amd/csrc/ab_$1.so" > /dev/null
echo "avsiam_amd/csrc/ab_$1.so"
