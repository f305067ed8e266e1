#!/bin/bash
# Round-5 artefacts for the current build (GPU box, repo root):  bash tools/round4_profile.sh [part ...]   parts: bench prof prof8 pmc dp fp8 fp8s big ln rehearsal attn shapes b4
# (in the build container first: git rev-parse HEAD > HEAD_COMMIT - the PMC summaries stamp it)
# Everything goes to gpurun_out/r05/ (copy what is to be judged into profiles/r05/).
set -e
OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
REPO=$PWD
PARTS=${@:-bench prof pmc dp fp8 big ln rehearsal}
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
if has pmc; then
  bash tools/pmc_traffic.sh r05/pmc_traffic > $OUT/pmc_traffic.log 2>&1 && cp gpurun_out/r05/pmc_traffic/traffic.json $OUT/traffic.json && cp gpurun_out/r05/pmc_traffic/traffic_by_kernel.txt $OUT/traffic_by_kernel.txt
  bash tools/pmc_busy.sh r05/pmc_busy > $OUT/pmc_busy.log 2>&1 && cp gpurun_out/r05/pmc_busy/pmc_busy.json $OUT/pmc_busy.json && cp gpurun_out/r05/pmc_busy/pmc_busy_by_kernel.txt $OUT/pmc_busy_by_kernel.txt
  echo "pmc done"; cat $OUT/pmc_busy_by_kernel.txt
fi
if has dp; then
  # the data-parallel branch on hardware at world size 1: every collective issued; torch.distributed and the C ABI's communicator,
  # fp32 and bf16 wire, overlapped chunks and one blocking message, deferred MAE-only update - against the plain line of the same box
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 > $OUT/dp_plain.json 2> $OUT/dp.err
  torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_torch.json 2>> $OUT/dp.err
  AVSIAM_COMM=rccl torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_rccl.json 2>> $OUT/dp.err
  AVSIAM_DP_WIRE=bf16 torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_torch_bf16wire.json 2>> $OUT/dp.err
  AVSIAM_DP_OVERLAP=0 torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_torch_blocking.json 2>> $OUT/dp.err
  AVSIAM_DP_DEFER=1 torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_torch_defer.json 2>> $OUT/dp.err
  # round 5: under data parallelism every persistent kernel leaves 8 CUs to the collectives (cu_reserve, set by set_distributed); its cost at one
  # rank: the same --force-dp line with the reservation switched off, and the plain line with it switched on
  AVSIAM_CU_RESERVE=0 torchrun --standalone --local-addr 127.0.0.1 --nproc-per-node 1 bench.py --secondary-steps 0 --steps 10 --no-cpu-baseline --force-dp > $OUT/dp_force_torch_no_cu_reserve.json 2>> $OUT/dp.err
  AVSIAM_CU_RESERVE=8 python bench.py --no-cpu-baseline --steps 10 --secondary-steps 0 > $OUT/dp_plain_cu_reserve8.json 2>> $OUT/dp.err
  AVSIAM_CU_RESERVE=16 python bench.py --no-cpu-baseline --steps 10 --secondary-steps 0 > $OUT/dp_plain_cu_reserve16.json 2>> $OUT/dp.err
  python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/dp_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 2), d["config"].get("force_dp"))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
if has fp8; then
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 > $OUT/b_vitb_fp8_bench.json 2> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 --fp8-dgrad > $OUT/b_vitb_fp8_dgrad_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 --fp8-wgrad > $OUT/b_vitb_fp8_wgrad_bench.json 2>> $OUT/fp8.err
  AVSIAM_FP8_LEAN=0 python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 --fp8-wgrad > $OUT/b_vitb_fp8_wgrad_bf16copies_bench.json 2>> $OUT/fp8.err      # A/B: every bf16 copy still written
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 > $OUT/b_vitb_bf16_same_box_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute > $OUT/h_vit_huge14_b64_recompute_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute --fp8 > $OUT/h_vit_huge14_b64_recompute_fp8_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute --fp8 --fp8-dgrad > $OUT/h_vit_huge14_b64_recompute_fp8_dgrad_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.375 > $OUT/h_vit_huge14_b64_recompute0375_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.375 --fp8 --fp8-dgrad > $OUT/h_vit_huge14_b64_recompute0375_fp8_dgrad_bench.json 2>> $OUT/fp8.err
  # fp8 weight gradients: with every bf16 copy kept (AVSIAM_FP8_LEAN=0) 3/8 recomputed leaves 2 GiB of the card, so that A/B point is taken at 1/2;
  # with 8-bit-only outputs (default) 1/4 recomputed fits with a tenth of the card free (the --recompute auto policy)
  AVSIAM_FP8_LEAN=0 python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.5 --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_recompute05_fp8_wgrad_bf16copies_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.5 --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_recompute05_fp8_wgrad_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.25 --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_recompute025_fp8_wgrad_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute 0.5 --fp8 --fp8-dgrad > $OUT/h_vit_huge14_b64_recompute05_fp8_dgrad_bench.json 2>> $OUT/fp8.err
  # both passes' activations from one pool (the card holds the larger pass, not the sum): nothing is recomputed (--recompute auto lands on 0)
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers > $OUT/h_vit_huge14_b64_pooled_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute auto --share-pass-buffers --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_pooled_fp8_wgrad_bench.json 2>> $OUT/fp8.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_huge14 --recompute --fp8 --fp8-wgrad > $OUT/h_vit_huge14_b64_recompute_fp8_wgrad_bench.json 2>> $OUT/fp8.err
  python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/[bh]_*bench.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], round(d["value"], 2), round(d["ms_per_step"], 2), "bf16 frac", round(d["roofline"]["frac"], 3), "fp8 frac", d.get("roofline_fp8", {}).get("frac"), "GiB", d["config"].get("peak_memory_gib"))
    except Exception as e:
        print(f, "ERR", e)
PY
fi
if has prof8; then
  # kernel trace of the fp8 mode 3 step (ViT-B): the evidence behind roofline_fp8 and the fp8 weight-gradient family
  cd /tmp && export TMPDIR=/tmp
  AVSIAM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof8 -o k -- python3 $REPO/bench.py --secondary-steps 0 --steps 4 --warmup 3 --no-cpu-baseline --roofline-steps 0 --fp8 --fp8-wgrad > $OUT/b_vitb_fp8_wgrad_s0_bench_rocprof.json 2> $OUT/b_rocprof.err
  cd $REPO
  find $OUT/prof8 -name "*kernel_stats.csv" -exec cp {} $OUT/b_vitb_fp8_wgrad_s0_kernel_stats.csv \;
  rm -rf $OUT/prof8
  echo "prof8 done"; head -8 $OUT/b_vitb_fp8_wgrad_s0_kernel_stats.csv
fi
if has big; then
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 5 --warmup 2 --model vit_large > $OUT/l_vit_large_bench.json 2> $OUT/big.err
  python -c "import json; d=json.load(open('$OUT/l_vit_large_bench.json')); print('vit_large', d['value'], d['ms_per_step'], d['roofline']['frac'])"
fi
if has ln; then
  for v in 0 1 0 1; do AVSIAM_LN_DMA=$v python tools/bench_ln.py --tag dma$v 2>/dev/null >> $OUT/layernorm_dma_ab.log; done
  cat $OUT/layernorm_dma_ab.log
fi
if has rehearsal; then
  # `python bench.py --gpus 2` started the way the driver starts it (no launcher), both ranks on the one GPU (gloo + host staging)
  AVSIAM_BENCH_SHARE_GPU=1 python bench.py --secondary-steps 0 --gpus 2 --steps 3 --warmup 1 --batch 16 --frames 2 --no-cpu-baseline --roofline-steps 0 > $OUT/rehearsal_gpus2.json 2> $OUT/rehearsal_gpus2.err
  tail -3 $OUT/rehearsal_gpus2.err; head -c 400 $OUT/rehearsal_gpus2.json; echo
fi
if has attn; then
  # attention kernels at the step's sequence mixes: register-staged vs LDS-DMA ring (interleaved rounds in one process, medians)
  python tools/bench_attn.py --step --ring-ab 2>/dev/null > $OUT/attn_ring_ab.log
  cat $OUT/attn_ring_ab.log
fi
if has shapes; then
  # L2 -> fabric read / write bytes of single forward-GEMM shapes against their algorithmic bytes (which operand is re-fetched)
  bash tools/pmc_gemm_shapes.sh > $OUT/pmc_shapes.log 2>&1 && cp gpurun_out/pmc_shapes/summary.txt $OUT/gemm_traffic_by_shape.txt
  cat $OUT/gemm_traffic_by_shape.txt
fi
if has b4; then
  # the reference's launch geometry (batch 4, one frame): where the 18 ms of a step go - kernel trace of 20 steps
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b4 -o k -- python3 $REPO/bench.py --batch 4 --frames 1 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-events --roofline-steps 0 --secondary-steps 0 > $OUT/b4_bench_rocprof.json 2> $OUT/b4_rocprof.err
  cd $REPO
  find $OUT/prof_b4 -name "*kernel_stats.csv" -exec cp {} $OUT/b4_kernel_stats.csv \;
  rm -rf $OUT/prof_b4
  head -12 $OUT/b4_kernel_stats.csv
fi
if has fp8s; then
  # the short fp8 set: ViT-B in the four precisions on one box, ViT-H/14 at batch 64 x 10 frames from one activation pool (bf16 and fp8 mode 3)
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 > $OUT/b_vitb_bf16_same_box_bench.json 2> $OUT/fp8s.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 > $OUT/b_vitb_fp8_bench.json 2>> $OUT/fp8s.err
  python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 --fp8 --fp8-dgrad > $OUT/b_vitb_fp8_dgrad_bench.json 2>> $OUT/fp8s.err
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
if has lnstep; then
  # LayerNorm backward variants IN THE STEP: the LDS-DMA kernel (default) against the register kernel, alternating on one box
  for v in 1 0 1 0; do
    AVSIAM_LN_DMA=$v python bench.py --secondary-steps 0 --no-cpu-baseline --steps 10 > $OUT/lnstep_dma$v.json 2>> $OUT/lnstep.err
    python - <<PY
import json
d = json.load(open("$OUT/lnstep_dma$v.json")); f = d["roofline_more"]["ms_per_step_by_family"]
print("AVSIAM_LN_DMA=$v", round(d["value"], 1), "samples/s", round(d["ms_per_step"], 2), "ms/step; layernorm_bwd", f["layernorm_bwd"], "ms single-stream")
PY
  done | tee $OUT/layernorm_dma_in_step_ab.log
fi
