# the round-5 behaviour of this tree (no dead-row pruning, bf16 gelu', weight-gradient stream mode 2) against the round-6 defaults, ONE box, alternating
run() { name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --secondary-steps 0 --roofline-steps 0 > gpurun_out/ab56_$name.json 2> gpurun_out/ab56_$name.err || { echo FAIL $name; tail -3 gpurun_out/ab56_$name.err; return; }
  python -c "
import json; d=json.load(open('gpurun_out/ab56_$name.json')); print('$name', round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['config']['peak_memory_gib'])"
}
for i in 1 2 3; do
run r05like_$i AVSIAM_PRUNE_DEAD=0 AVSIAM_GELU8=0 AVSIAM_WGRAD_STREAM=2
run r06_$i A=1
done
