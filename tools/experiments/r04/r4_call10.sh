#!/bin/bash
# Round 4, GPU call 10: what bounds the quotient kernel?  SQ stall counters + HBM bytes of the shipped compact kernel (quotient_fuse = 6) and of the
# round 2-3 kernel (0), 2^27 points, with the 25 inputs distinct and with all of them aliased to ONE buffer.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
P=$O/r4_quot_pmc
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
G3="FETCH_SIZE WRITE_SIZE"
for A in distinct aliased; do
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    ( [ $A = aliased ] && export QUOT_ALIAS=1
      timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $P/${A}_g$i -o p -- python $R/tools/quotient_bench.py 24 > $P/${A}_g$i.out 2> $P/${A}_g$i.err )
  done
done
python3 - $P <<'PY' | tee $O/r4_quot_pmc.txt
import csv, glob, sys, collections
for A in ("distinct", "aliased"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for path in glob.glob(sys.argv[1] + f"/{A}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "quotient_evals_kernel" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for path in glob.glob(sys.argv[1] + f"/{A}_g1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if "quotient_evals_kernel" in k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, c in sorted(acc.items()):
        print(A, k, "launches", max(len(v) for v in c.values()), "avg ms under the counters", round(sum(dur[k]) / max(len(dur[k]), 1), 2))
        for n, v in sorted(c.items()):
            print("   %-24s %16.0f" % (n, sum(v) / len(v)))
PY
find $P -name "*.csv" -delete
