#!/bin/bash
# PMC passes over any python command: tools/pmc_any.sh OUTNAME "CTR_A CTR_B|CTR_C" script.py [args…]  (one pass per |-separated group)
# → gpurun_out/pmc_OUTNAME.txt: mean counter value per dispatch, per kernel name (kernels matching $PMC_MATCH, default all)
name="$1"; groups="$2"; shift 2
root="$(pwd)"; out="$root/gpurun_out/pmcany_$name"
mkdir -p "$out"; export TMPDIR=/tmp
: > "$root/gpurun_out/pmc_$name.txt"
IFS='|' read -ra G <<< "$groups"
i=0
for grp in "${G[@]}"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$out/g$i" -o p -- python3 "$root/$1" "${@:2}" > "$out/g$i.log" 2>&1 ) || { tail -5 "$out/g$i.log"; exit 1; }
  f=$(find "$out/g$i" -name "p_counter_collection.csv" | head -1)
  python3 - "$f" "${PMC_MATCH:-}" >> "$root/gpurun_out/pmc_$name.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    if sys.argv[2] and sys.argv[2] not in k:
        continue
    acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
PY
  rm -rf "$out/g$i"
done
cat "$root/gpurun_out/pmc_$name.txt"
