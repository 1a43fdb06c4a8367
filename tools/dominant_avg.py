#!/usr/bin/env python
"""Average duration of bench.py's dominant kernel over a rocprofv3 kernel trace of the bench command — the cross-check of the
`roofline.avg_launch_ms` bench.py brackets with HIP events.  Selection = the launches bench.py brackets: gemm_p8_kernel with one of
the M = 19,200 forward grids (the symbol also serves the decoder's 4,224-row launches, which are not part of the figure).
usage: python tools/dominant_avg.py <prefix>_kernel_trace.csv"""
import csv
import sys

GRIDS = {"115200": "N=768", "230400": "N=1536", "345600": "N=2304"}
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("gemm_p8_kernel" in r["Kernel_Name"] or "gemm_p8x3_kernel" in r["Kernel_Name"]) and r["Grid_Size_X"] in GRIDS]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("gemm_p8 / gemm_p8x3 kernel, M = 19,200 forward launches: %d calls, average %.2f us (min %.2f, max %.2f)"
      % (len(dur), sum(dur) / len(dur) / 1e3, min(dur) / 1e3, max(dur) / 1e3))
for g, name in GRIDS.items():
    d = [x for x, r in zip(dur, rows) if r["Grid_Size_X"] == g]
    if d:
        print("  grid %s (%s): %d calls, average %.2f us" % (g, name, len(d), sum(d) / len(d) / 1e3))
