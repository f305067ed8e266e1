#!/bin/bash
# Register / LDS / occupancy table of one source's kernels with the product's flags:  bash tools/kernel_regs.sh attention.hip [filter-regex] [extra hipcc flags...]
SRC=${1:-attention.hip}; FILT=${2:-.}; shift 2 2>/dev/null
ROOT=$(cd "$(dirname "$0")/.." && pwd)
EXTRA=""; [ "$SRC" = attention.hip ] && EXTRA="-fno-slp-vectorize"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 $EXTRA "$@" -I$ROOT/include \
  -Rpass-analysis=kernel-resource-usage -c $ROOT/avsiam_amd/csrc/$SRC -o $T/a.o 2> $T/res.txt
python3 - "$T/res.txt" "$FILT" <<'PY'
import re, subprocess, sys
rows, cur = [], None
for line in open(sys.argv[1]):
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'occ':>4s} {'scratch':>8s} {'LDS':>7s}")
for r in rows:
    if re.search(sys.argv[2], r["name"]):
        print(f"{r['name'][:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('Occupancy','?'):>4s} {r.get('ScratchSize','?'):>8s} {r.get('LDS Size','?'):>7s}")
PY
rm -rf $T
