"""Mean of every PMC counter per launch for kernels whose name contains a substring, from rocprofv3 counter_collection csv files.
usage: python tools/pmc_kernel.py SUBSTR file1.csv [file2.csv ...]"""
import collections, csv, sys
sub = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-60:] + " grid " + r["Grid_Size"]
            acc[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f   (%d launches, %.1f us each)" % (c, sum(x for x, _ in v) / len(v), len(v), sum(t for _, t in v) / len(v) / 1e3))
