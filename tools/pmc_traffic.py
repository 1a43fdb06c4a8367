#!/usr/bin/env python
"""HBM traffic per launch of bench.py's dominant kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: MI355X_MICROARCH.md "rocprofv3 PMC slots") over the bench command itself:

  cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/f -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary
  cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/w -o w -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-secondary
  python tools/pmc_traffic.py OUT/f/f_counter_collection.csv OUT/w/w_counter_collection.csv profiles/r02_x_pmc_bench_dominant_gemm.csv

Selection = every launch of gemm_p8_kernel whose grid is one of the M = 19,200 forward grids (the launches
bench.py brackets).  Corrections as the guide prescribes: FETCH_SIZE × 2 on gfx950 for 16-B/lane streaming reads (128-B requests
are tallied at 64 B), WRITE_SIZE as read; both in KB.  Writes profiles/dominant_gemm_traffic.json (read by bench.py), stamped with
the hash of the kernel's sources so that a later edit of the kernel nulls the figure until it is measured again."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GRIDS = {"115200": "N=768 (attention-out / FFN / video embedding)", "230400": "N=1536 (K/V of the [CLS]-only layer)", "345600": "N=2304 (Q/K/V)"}
# fourth argument (optional): the arithmetic mode the bench command ran in — "bf16" (gemm_p8_kernel, default) or "bf16x3" (gemm_p8x3_kernel)
MODE = sys.argv[4] if len(sys.argv) > 4 else "bf16"
SYMBOL = "gemm_p8x3_kernel" if MODE == "bf16x3" else "gemm_p8_kernel"
SYMBOL_DEMANGLED = SYMBOL


def read(path, counter):
    by_grid = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if SYMBOL not in name or (MODE != "bf16x3" and "gemm_p8x3" in name):
            continue
        if r.get("Counter_Name") != counter:
            continue
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        if grid in GRIDS:
            by_grid[grid].append(float(r["Counter_Value"]))
    return by_grid


def main():
    f_csv, w_csv, out_csv = sys.argv[1:4]
    fetch, write = read(f_csv, "FETCH_SIZE"), read(w_csv, "WRITE_SIZE")
    rows, tot_f, tot_w, n_f, n_w = [], 0.0, 0.0, 0, 0
    for g in sorted(GRIDS, key=int):
        for cname, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            v = d.get(g, [])
            if v:
                rows.append((cname, g, GRIDS[g], len(v), sum(v) / len(v)))
        tot_f += sum(fetch.get(g, [])); n_f += len(fetch.get(g, []))
        tot_w += sum(write.get(g, [])); n_w += len(write.get(g, []))
    assert n_f and n_w, "no launch of the dominant kernel found in the PMC output"
    mean_f, mean_w = tot_f / n_f, tot_w / n_w
    traffic = (2.0 * mean_f + mean_w) * 1024.0
    with open(out_csv, "w") as f:
        f.write("counter,kernel,grid_threads,shape,launches,mean_KB_per_launch\n")
        for cname, g, what, n, m in rows:
            f.write('%s,"%s",%s,"%s",%d,%.1f\n' % (cname, SYMBOL, g, what, n, m))
        f.write('FETCH_SIZE,"all M=19200 forward launches",,,%d,%.1f\n' % (n_f, mean_f))
        f.write('WRITE_SIZE,"all M=19200 forward launches",,,%d,%.1f\n' % (n_w, mean_w))
    rec = {"traffic_bytes_per_launch": traffic, "fetch_KB_mean": mean_f, "write_KB_mean": mean_w, "launches_fetch_pass": n_f,
           "launches_write_pass": n_w, "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: 128-B read requests are tallied at 64 B)",
           "kernel_sources_sha16": bench.dominant_kernel_sha(MODE), "source": "profiles/" + os.path.basename(out_csv), "kernel": SYMBOL}
    with open(bench.TRAFFIC_FILE_X3 if MODE == "bf16x3" else bench.TRAFFIC_FILE, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
