#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from two rocprofv3 PMC passes of the same bench command.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/pmc_write -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-events
    python tools/pmc_summary.py out/pmc_fetch/f_counter_collection.csv out/pmc_write/w_counter_collection.csv [CALLS] > profiles/rNN/traffic_gemm_nt.json

CALLS = number of avs_gemm_nt_bf16(_dual) calls in the profiled run (282 per step; a call is one dispatch, or two when the
two-buffer kernels hand leftover rows to a second launch), so that the figure is per launch as bench.py counts launches.

FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads and is doubled
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  Also prints a per-kernel table (sum over all dispatches) to stderr.
"""
import csv
import json
import sys
from collections import defaultdict


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


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, nw = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        rows.append((k, nf.get(k, 0), 2 * fetch.get(k, 0) * 1024, write.get(k, 0) * 1024))
    for k, n, fb, wb in rows[:25]:
        print(f"{k[:60]:60s} x{n:5d}  read {fb / 1e9:8.2f} GB  write {wb / 1e9:8.2f} GB", file=sys.stderr)
    print(f"all kernels: read {sum(r[2] for r in rows) / 1e9:.1f} GB, write {sum(r[3] for r in rows) / 1e9:.1f} GB", file=sys.stderr)
    mm = [r for r in rows if "gemm_nt" in r[0]]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else sum(r[1] for r in mm)
    fb, wb = sum(r[2] for r in mm), sum(r[3] for r in mm)
    print(json.dumps({"kernel": "gemm_nt8_kernel + gemm_nt_kernel (all instantiations)", "launches": n, "dispatches": sum(r[1] for r in mm), "fetch_bytes_per_launch_raw": fb / 2 / n,
                      "fetch_bytes_per_launch_corrected": fb / n, "write_bytes_per_launch": wb / n, "hbm_bytes_per_launch": (fb + wb) / n,
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 1 --warmup 1`; "
                                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); values in KiB"},
                     indent=1))


if __name__ == "__main__":
    main()
