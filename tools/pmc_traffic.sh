#!/bin/bash
# HBM traffic of the roofline kernels for bench.py's `traffic` fields: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - separate
# runs, counters only) of the bench command, summarised per launch and stamped with the kernel sources' SHA-1.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag>   ->  gpurun_out/<tag>/traffic.json  (commit as profiles/rNN/traffic.json)
set -e
TAG=${1:-pmc}
REPO=$PWD
OUT=$PWD/gpurun_out/${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $REPO/bench.py --secondary-steps 0 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/fetch_bench.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $REPO/bench.py --secondary-steps 0 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/write_bench.json 2> $OUT/write.err
echo "write pass done"
cd $REPO
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1)
W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py $F $W > $OUT/traffic.json 2> $OUT/traffic_by_kernel.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
head -40 $OUT/traffic.json; head -14 $OUT/traffic_by_kernel.txt
