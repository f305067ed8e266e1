#!/usr/bin/env python3
"""HBM bytes per launch of the kernel families bench.py puts on the roofline, from two rocprofv3 PMC passes of the same bench
command (tools/pmc_traffic.sh):

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/rNN/traffic.json

FETCH_SIZE / WRITE_SIZE count KiB at the L2 -> fabric boundary (Infinity-Cache hits included); on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads and is doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  A "launch" is one call
of the C ABI: one dispatch for the GEMMs, dq + dkv kernels for the attention backward, ln_bwd + its slab reduction for the
LayerNorm backward.  The SHA-1 of every kernel source is stored beside the numbers: bench.py reports a figure only while
the source it was measured on is unchanged.  A per-kernel table goes to stderr.
"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def provenance():
    """box / date / commit of the measurement.  The GPU box has no .git: the commit is what `git rev-parse HEAD > HEAD_COMMIT` left in
    the repo root before the snapshot was sent (tools/round4_profile.sh says how); the kernel sources' SHA-1 tie the numbers to the code."""
    import socket
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "HEAD_COMMIT")
    commit = open(path).read().strip() if os.path.exists(path) else None
    return {"box": socket.gethostname(), "date": time.strftime("%Y-%m-%dT%H:%M:%S"), "commit": commit}


def load(path, counter):
    tot, n = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


# family -> (substrings of the kernel names that belong to it, substring of the kernel whose dispatches count the launches)
FAMILIES = {
    "gemm_nt": (("gemm_nt",), "gemm_nt"),
    "gemm_tn": (("gemm_tn",), "gemm_tn"),
    "attn_hd32": (("attn_fwd_kernel<32", "attn_bwd_dq_kernel<32", "attn_bwd_dkv_kernel<32", "attn_bwd_fused_kernel<32"), None),
    "attn_hd64": (("attn_fwd_kernel<64", "attn_bwd_dq_kernel<64", "attn_bwd_dkv_kernel<64", "attn_bwd_fused_kernel<64"), None),
    "ln_bwd": (("ln_bwd_kernel", "ln_bwd_dma_kernel", "ln_bwd_reduce_kernel"), "ln_bwd_"),      # (two main kernels since round 4: register loads / LDS-DMA)
    "ln_fwd": (("ln_fwd_kernel",), "ln_fwd_kernel"),
}


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        rows.append((k, nf.get(k, 0), 2 * fetch.get(k, 0) * 1024, write.get(k, 0) * 1024))
    for k, n, fb, wb in rows[:30]:
        print(f"{k[:60]:60s} x{n:5d}  read {fb / 1e9:8.2f} GB  write {wb / 1e9:8.2f} GB", file=sys.stderr)
    print(f"all kernels: read {sum(r[2] for r in rows) / 1e9:.1f} GB, write {sum(r[3] for r in rows) / 1e9:.1f} GB", file=sys.stderr)
    out = {}
    for fam, (subs, count_sub) in FAMILIES.items():
        mm = [r for r in rows if any(s in r[0] for s in subs)]
        if not mm:
            continue
        if count_sub is None:          # attention: forward launches + backward launches (dq and dkv are one call, a fused backward another)
            n = sum(r[1] for r in mm if "attn_fwd" in r[0]) + sum(r[1] for r in mm if "attn_bwd_dkv" in r[0] or "attn_bwd_fused" in r[0])
        else:
            n = sum(r[1] for r in mm if count_sub in r[0] and "reduce" not in r[0])
        fb, wb = sum(r[2] for r in mm), sum(r[3] for r in mm)
        out[fam] = {"launches": n, "dispatches": sum(r[1] for r in mm), "fetch_bytes_per_launch_corrected": fb / n, "write_bytes_per_launch": wb / n,
                    "hbm_bytes_per_launch": (fb + wb) / n}
    sha = {}
    csrc = os.path.join(ROOT, "avsiam_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(csrc, f), "rb") as fh:
                sha[f] = hashlib.sha1(fh.read()).hexdigest()
    print(json.dumps({"kernels": out, "source_sha1": sha, **provenance(),
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 1 --no-cpu-baseline "
                                "--no-kernel-events`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); "
                                "values in KiB; totals over the run divided by the launches of the run"}, indent=1))


if __name__ == "__main__":
    main()
