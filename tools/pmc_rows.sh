#!/bin/bash
# HBM traffic of the O(n) rows from PMC counters (separate --pmc passes, counters only; (2*FETCH_SIZE + WRITE_SIZE) KiB per launch as the
# micro-architecture guide prescribes for gfx950 streams) beside their algorithmic bytes.   usage (gpurun): bash tools/pmc_rows.sh [log_n]
L=${1:-24}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_rows
mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/$C -o p -- python $R/tools/poly_rows_only.py $L 2 > $O/$C.out 2> $O/$C.err
done
python3 - $O $L <<'PY'
import csv, glob, sys, collections
d, L = sys.argv[1], int(sys.argv[2])
n = 1 << L
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for path in glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(s in k for s in ("poly_", "perm_", "scan_", "fr_inv", "fr_sum")):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
alg = {"poly_eval_kernel": 32, "poly_div_agg_kernel": 32, "poly_div_apply_kernel": 64, "perm_terms_kernel": 16 * 32 + 5 * 8 + 64}
for k, c in sorted(acc.items()):
    f = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    w = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
    tr = (2 * f + w) * 1024
    a = alg.get(k.split("<")[0])
    print(f"{k[:60]:60s} fetch {f * 1024 / 1e9:8.3f} GB (x2 corrected {2 * f * 1024 / 1e9:8.3f})  write {w * 1024 / 1e9:8.3f} GB  traffic {tr / 1e9:8.3f} GB" + (f"  algorithmic {a * n / 1e9:.3f} GB = x{tr / (a * n):.2f}" if a else ""))
PY
find $O -name "*.csv" -delete
