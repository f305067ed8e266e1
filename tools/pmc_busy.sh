#!/bin/bash
# MFMA-busy / VALU / wait counters of the roofline kernel families at HEAD (north_star: "rocprof HBM GB/s and MFMA-busy counters
# reported against gfx950 peak"): two rocprofv3 PMC passes (counters only - no trace domains beside --pmc) of the bench command with
# everything on ONE stream, summarised per kernel family and stamped with the kernel sources' SHA-1.
# usage (GPU box, repo root): bash tools/pmc_busy.sh <tag>   ->  gpurun_out/<tag>/pmc_busy.json  (commit as profiles/rNN/pmc_busy.json)
set -e
TAG=${1:-busy}
REPO=$PWD
OUT=$PWD/gpurun_out/${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AVSIAM_WGRAD_STREAM=0
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 $REPO/bench.py --secondary-steps 0 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/p${i}_bench.json 2> $OUT/p$i.err
  echo "busy pass $i done"
done
cd $REPO
python3 tools/pmc_busy_summary.py $(find $OUT/p1 $OUT/p2 -name "*counter_collection.csv") > $OUT/pmc_busy.json 2> $OUT/pmc_busy_by_kernel.txt
rm -rf $OUT/p1 $OUT/p2
head -60 $OUT/pmc_busy.json
