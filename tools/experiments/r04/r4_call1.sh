#!/bin/bash
# Round 4, GPU call 1 (VERDICT r3 "next" #1a, #2, #3): measure what round 3 built blind, then collect the evidence the next steps need.
#   gpurun --timeout 2100 -- 'bash tools/experiments/r04/r4_call1.sh'
#   A. tools/experiments/r04/round4_opening.sh (new tests, fuzzer on the real library, grid-reduction A/B, polynomial_parallel busiest rank)
#   B. stall attribution of ntt_pass_kernel<8,4,true,true>: two SQ counter passes of the zero-padded coset FFT at 2^24 (counters only)
#   C. BLS12-381 alone: MSM 2^22 and coset FFT 2^22 kernel stats (no second context beside them)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
sed -i 's/--seconds 180/--seconds 100/' tools/experiments/r04/round4_opening.sh
timeout 1300 bash tools/experiments/r04/round4_opening.sh > $O/r4open_all.txt 2>&1
tail -60 $O/r4open_all.txt

cd /tmp && export TMPDIR=/tmp
P=$O/r4_ntt_stall
mkdir -p $P
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU"
G2="SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
G3="SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"
i=0
for G in "$G1" "$G2" "$G3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $P/g$i -o p -- python $R/tools/coset_eval_only.py 24 > $P/g$i.out 2> $P/g$i.err
  tail -2 $P/g$i.err | cut -c1-300
done
python3 - $P <<'PY' | tee $O/r4_ntt_stall.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "ntt_pass_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k[-60:], "launches", max(len(v) for v in c.values()))
    for n, v in sorted(c.items()):
        print("   %-26s %16.0f" % (n, sum(v) / len(v)))
PY
find $P -name "*.csv" -size +2M -delete

for what in msm ntt; do
  D=/tmp/prof_bls_$what
  if [ $what = msm ]; then CMD="python $R/tools/msm_only.py 22"; else CMD="python $R/tools/coset_eval_only.py 22"; fi
  CURVE=bls12_381 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- $CMD > $O/r4_bls_${what}_alone.log 2>&1
  grep -v "^W2\|^E2\|amdgpu.ids" $O/r4_bls_${what}_alone.log | tail -8
  cp $(find $D -name "*kernel_stats.csv" | head -1) $O/r4_kernel_stats_bls12_381_2p22_${what}_alone.csv
done
head -30 $O/r4_kernel_stats_bls12_381_2p22_msm_alone.csv
