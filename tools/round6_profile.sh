#!/bin/bash
# Round-6 artefacts for the current build (GPU box, repo root):  bash tools/round6_profile.sh [part ...]   parts: bench prof pmc t1 fp8s big
# (in the build container first: git rev-parse HEAD > HEAD_COMMIT - the PMC summaries stamp it)
# Everything goes to gpurun_out/r06/ (copy what is to be judged into profiles/r06/).
set -e
OUT=$PWD/gpurun_out/r06
mkdir -p $OUT
REPO=$PWD
PARTS=${@:-bench prof pmc t1}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python bench.py > $OUT/a_bench.json 2> $OUT/a_bench.err
  python bench.py --frames 1 --no-cpu-baseline --secondary-steps 0 > $OUT/a_bench_T1.json 2>> $OUT/a_bench.err
  echo "bench done"; head -c 300 $OUT/a_bench.json; echo
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o k -- python3 $REPO/bench.py --secondary-steps 0 --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $OUT/a_bench_rocprof.json 2> $OUT/a_rocprof.err
  AVSIAM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_s0 -o k -- python3 $REPO/bench.py --secondary-steps 0 --steps 4 --warmup 2 --no-cpu-baseline --roofline-steps 0 > $OUT/a_s0_bench_rocprof.json 2> $OUT/a_s0_rocprof.err
  cd $REPO
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/a_kernel_stats.csv \;
  find $OUT/prof_s0 -name "*kernel_stats.csv" -exec cp {} $OUT/a_s0_kernel_stats.csv \;
  rm -rf $OUT/prof $OUT/prof_s0
  echo "prof done"; head -6 $OUT/a_s0_kernel_stats.csv
fi
if has t1; then
  # the reference's one-frame shape at batch 64: kernel trace, one stream
  cd /tmp && export TMPDIR=/tmp
  AVSIAM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_t1 -o k -- python3 $REPO/bench.py --frames 1 --secondary-steps 0 --steps 6 --warmup 2 --no-cpu-baseline --roofline-steps 0 --no-kernel-events > $OUT/t1_s0_bench_rocprof.json 2> $OUT/t1_rocprof.err
  cd $REPO
  find $OUT/prof_t1 -name "*kernel_stats.csv" -exec cp {} $OUT/t1_kernel_stats.csv \;
  rm -rf $OUT/prof_t1
  echo "t1 done"; head -8 $OUT/t1_kernel_stats.csv
fi
if has pmc; then
  bash tools/pmc_traffic.sh r06/pmc_traffic > $OUT/pmc_traffic.log 2>&1 && cp gpurun_out/r06/pmc_traffic/traffic.json $OUT/traffic.json && cp gpurun_out/r06/pmc_traffic/traffic_by_kernel.txt $OUT/traffic_by_kernel.txt
  bash tools/pmc_busy.sh r06/pmc_busy > $OUT/pmc_busy.log 2>&1 && cp gpurun_out/r06/pmc_busy/pmc_busy.json $OUT/pmc_busy.json && cp gpurun_out/r06/pmc_busy/pmc_busy_by_kernel.txt $OUT/pmc_busy_by_kernel.txt
  echo "pmc done"; cat $OUT/pmc_busy_by_kernel.txt
fi
if has fp8s; then
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 > $OUT/b_vitb_bf16_same_box_bench.json 2> $OUT/fp8s.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 --fp8-wgrad > $OUT/b_vitb_fp8_wgrad_bench.json 2>> $OUT/fp8s.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers > $OUT/h_vit_huge14_b64_pooled_bench.json 2>> $OUT/fp8s.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_pooled_fp8_wgrad_bench.json 2>> $OUT/fp8s.err
  python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/[bh]_*bench.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 2), "bf16 frac", round(d["roofline"]["frac"], 3), "fp8 frac", d.get("roofline_fp8", {}).get("frac"), "GiB", d["config"].get("peak_memory_gib"), d["config"].get("activation_pool_gib"))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
if has big; then
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_large > $OUT/l_vit_large_bench.json 2> $OUT/big.err
  python -c "import json; d=json.load(open('$OUT/l_vit_large_bench.json')); print('vit_large', d['value'], d['ms_per_step'], d['roofline']['frac'])"
fi
