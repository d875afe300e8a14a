#!/bin/bash
# VALU / VMEM instruction counts of the class-route pass kernel with and without the precomputed-quotient butterflies (one rocprofv3
# counter pass each; counters only, no other trace domain).   usage (gpurun): bash tools/pmc_ntt_ab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ntt_ab
mkdir -p $O
for mode in shoup mont; do
  if [ $mode = mont ]; then export PLONK_NTT_NO_SHOUP=1; else unset PLONK_NTT_NO_SHOUP; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/$mode -o p -- python $R/tools/coset_eval_only.py 24 > $O/$mode.out 2> $O/$mode.err
  python3 - $O/$mode $mode <<'PY'
import csv, glob, sys, collections
d, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "ntt_pass_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(mode, k[-40:], {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
  find $O/$mode -name "*.csv" -delete
done
