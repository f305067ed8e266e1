#!/bin/bash
# L2-miss read traffic (FETCH_SIZE) and write traffic (WRITE_SIZE) of single forward-GEMM shapes, counters only, one pass per counter:
#   bash tools/pmc_gemm_shapes.sh        (GPU box, repo root)   -> gpurun_out/pmc_shapes/summary.txt
# FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950).
set -e
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_shapes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in "768 3072 1" "768 768 0" "768 2304 0" "3072 768 0"; do
  tag=$(echo $shape | tr ' ' '_')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/$tag.$c -o c -- python3 $REPO/tools/pmc_gemm.py $shape 1 0 > /dev/null 2>&1
  done
done
cd $REPO
python3 - <<'PY' | tee gpurun_out/pmc_shapes/summary.txt
import csv, glob, collections
M = 95630
for d in sorted(set(x.rsplit('.', 1)[0] for x in glob.glob('gpurun_out/pmc_shapes/*.FETCH_SIZE'))):
    K, N, act = (int(v) for v in d.split('/')[-1].split('_'))
    tot = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        v = n = 0
        for f in glob.glob(f'{d}.{c}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt8" in r["Kernel_Name"] and r['Counter_Name'] == c:
                    v += float(r['Counter_Value']); n += 1
        tot[c] = v * 1024 / max(n, 1) * (2 if c == 'FETCH_SIZE' else 1)
    alg_r = (M * K + N * K) * 2
    alg_w = M * N * 2 * (2 if act == 1 else 1)
    print(f"K={K} N={N} act={act}: read {tot['FETCH_SIZE']/1e6:7.1f} MB (algorithmic {alg_r/1e6:6.1f})   write {tot['WRITE_SIZE']/1e6:7.1f} MB (algorithmic {alg_w/1e6:6.1f})")
PY
