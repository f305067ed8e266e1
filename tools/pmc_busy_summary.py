#!/usr/bin/env python3
"""Per kernel family: matrix-pipe busy fraction, VALU-active fraction, waiting fraction, MFMA / VALU instruction counts, from the
rocprofv3 PMC passes of tools/pmc_busy.sh:

    python tools/pmc_busy_summary.py <counter_collection.csv> [...] > profiles/rNN/pmc_busy.json

Definitions (MI355X_MICROARCH.md, per-instruction constants / PMC slots; 256 CUs x 4 SIMDs, 8 XCDs):
  gpu_cycles   = GRBM_GUI_ACTIVE / 8                      (rocprofv3 sums the counter over the 8 XCDs)
  mfma_busy    = SQ_VALU_MFMA_BUSY_CYCLES / (gpu_cycles * 1024)      cycles in which a SIMD's matrix pipe is busy, of all SIMD cycles
  valu_active  = 4 * SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES-equivalent: reported as the share of WAVE time (both count quad-cycles)
  wait_any / wait_inst = SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES  (wave parked / issue stalled, share of wave time)
Totals over all dispatches of the family in the profiled run; the kernel sources' SHA-1 travels with the numbers (bench.py reports a
figure only while the source it was measured on is unchanged)."""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {
    "gemm_nt": ("gemm_nt",),
    "gemm_tn": ("gemm_tn",),
    "attn_hd32": ("attn_fwd_kernel<32", "attn_bwd_dq_kernel<32", "attn_bwd_dkv_kernel<32", "attn_bwd_fused_kernel<32"),
    "attn_hd64": ("attn_fwd_kernel<64", "attn_bwd_dq_kernel<64", "attn_bwd_dkv_kernel<64", "attn_bwd_fused_kernel<64"),
    "ln_bwd": ("ln_bwd_kernel", "ln_bwd_dma_kernel"),
    "ln_fwd": ("ln_fwd_kernel",),
}



def provenance():
    """box / date / commit of the measurement.  The GPU box has no .git: the commit is what `git rev-parse HEAD > HEAD_COMMIT` left in
    the repo root before the snapshot was sent (tools/round4_profile.sh says how); the kernel sources' SHA-1 tie the numbers to the code."""
    import socket
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "HEAD_COMMIT")
    commit = open(path).read().strip() if os.path.exists(path) else None
    return {"box": socket.gethostname(), "date": time.strftime("%Y-%m-%dT%H:%M:%S"), "commit": commit}


def main():
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                k, c = r["Kernel_Name"], r["Counter_Name"]
                tot[k][c] += float(r["Counter_Value"])
                cnt[k][c] += 1
    fam = {}
    for name, subs in FAMILIES.items():
        agg, n = defaultdict(float), defaultdict(int)
        for k in tot:
            if any(s in k for s in subs):
                for c, v in tot[k].items():
                    agg[c] += v
                    n[c] += cnt[k][c]
        if not agg:
            continue
        # GRBM_GUI_ACTIVE is collected in both passes: average it
        passes = max(1, round(n["GRBM_GUI_ACTIVE"] / max(1, n.get("SQ_WAVE_CYCLES", 1))))
        gpu_cycles = agg["GRBM_GUI_ACTIVE"] / passes / 8.0
        wave = agg.get("SQ_WAVE_CYCLES", 0.0)
        d = {"dispatches": n.get("SQ_WAVE_CYCLES", 0), "gpu_cycles": gpu_cycles,
             "mfma_busy": agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gpu_cycles * 1024.0) if gpu_cycles else None,
             "sq_busy_frac": agg.get("SQ_BUSY_CYCLES", 0.0) / (agg["GRBM_GUI_ACTIVE"] / passes) if gpu_cycles else None,
             "valu_active_of_wave_time": agg.get("SQ_ACTIVE_INST_VALU", 0.0) / wave if wave else None,
             "wait_any_of_wave_time": agg.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
             "wait_inst_of_wave_time": agg.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
             "insts_valu": agg.get("SQ_INSTS_VALU"), "insts_mfma": agg.get("SQ_INSTS_MFMA"),
             "raw": {c: v for c, v in sorted(agg.items())}}
        fam[name] = d
        print(f"{name:10s} x{d['dispatches']:5d}  mfma_busy {d['mfma_busy']:.3f}  valu {d['valu_active_of_wave_time']}  wait_any {d['wait_any_of_wave_time']}",
              file=sys.stderr)
    # per KERNEL (not family) for the attention kernels: VERDICT r2 asks for the decoder attention's counters kernel by kernel
    per_kernel = {}
    for k in sorted(tot):
        if "attn_" not in k:
            continue
        t, n = tot[k], cnt[k]
        disp = n.get("SQ_WAVE_CYCLES", 0)
        if not disp:
            continue
        passes = max(1, round(n["GRBM_GUI_ACTIVE"] / disp))
        gpu_cycles = t["GRBM_GUI_ACTIVE"] / passes / 8.0
        wave = t.get("SQ_WAVE_CYCLES", 0.0)
        name = k.split("(")[0].replace("void ", "")
        per_kernel[name] = {"dispatches": disp, "mfma_busy": t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gpu_cycles * 1024.0),
                            "valu_active_of_wave_time": t.get("SQ_ACTIVE_INST_VALU", 0.0) / wave if wave else None,
                            "wait_any_of_wave_time": t.get("SQ_WAIT_ANY", 0.0) / wave if wave else None,
                            "wait_inst_of_wave_time": t.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else None,
                            "valu_insts_per_mfma": (t.get("SQ_INSTS_VALU", 0.0) / t["SQ_INSTS_MFMA"]) if t.get("SQ_INSTS_MFMA") else None,
                            "gpu_cycles_per_dispatch": gpu_cycles / disp}
        print(f"  {name:44s} x{disp:4d} mfma_busy {per_kernel[name]['mfma_busy']:.3f} valu {per_kernel[name]['valu_active_of_wave_time']} wait {per_kernel[name]['wait_any_of_wave_time']}",
              file=sys.stderr)
    sha = {}
    csrc = os.path.join(ROOT, "avsiam_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(csrc, f), "rb") as fh:
                sha[f] = hashlib.sha1(fh.read()).hexdigest()
    print(json.dumps({"kernels": fam, "attention_kernels": per_kernel, "source_sha1": sha, **provenance(),
                      "method": "rocprofv3 --pmc (two counter-only passes) of `AVSIAM_WGRAD_STREAM=0 bench.py --steps 1 --warmup 1 --no-cpu-baseline "
                                "--no-kernel-events`; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)"}, indent=1))


if __name__ == "__main__":
    main()
