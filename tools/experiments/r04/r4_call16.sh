#!/bin/bash
# Round 4, last GPU minutes: SQ stall counters + HBM bytes of msm_accumulate_kernel, one 2^24-point MSM alone (no source change; input for round 5).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
P=$O/r4_msm_pmc
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
G2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"
G3="FETCH_SIZE WRITE_SIZE"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $P/g$i -o p -- python $R/tools/msm_only.py 24 > $P/g$i.out 2> $P/g$i.err
done
python3 - $P <<'PY' | tee $O/r04_msm_accumulate_counters.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in ("msm_accumulate_kernel", "sort_scatter", "sort_partition", "msm_reduce_level_kernel")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for path in glob.glob(sys.argv[1] + "/g1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("one 2^24-point BN254 MSM alone (tools/msm_only.py 24), averages per launch, shipped library")
for k, c in sorted(acc.items()):
    row = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, "launches", max(len(v) for v in c.values()), "avg ms under the counters", round(sum(dur[k]) / max(len(dur[k]), 1), 3))
    for n, v in sorted(row.items()):
        print("   %-24s %16.0f" % (n, v))
    if "SQ_WAVE_CYCLES" in row:
        wc = row["SQ_WAVE_CYCLES"]; busy = row["SQ_BUSY_CYCLES"] / 32
        print("   => waves per SIMD %.2f | issuing %.1f %% | ready, not issued %.1f %% | parked %.1f %% | VALU instructions per SIMD x 4 cycles = %.0f %% of the busy cycles"
              % (wc * 4 / (1024 * busy), 100 * row["SQ_ACTIVE_INST_ANY"] / wc, 100 * row["SQ_WAIT_INST_ANY"] / wc, 100 * row["SQ_WAIT_ANY"] / wc, 100 * row["SQ_INSTS_VALU"] / 1024 * 4 / busy))
    if "FETCH_SIZE" in row:
        print("   => HBM bytes per launch (2 * FETCH_SIZE + WRITE_SIZE): %.2f GB" % ((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024 / 1e9))
PY
find $P -name "*.csv" -delete
