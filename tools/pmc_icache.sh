#!/bin/bash
# Instruction-cache behaviour of the hot kernels (one rocprofv3 counter pass per workload; counters only): the quotient kernel is
# ~68 KiB of straight-line code (131 KiB when built with the raised unroll budget), the NTT pass kernel ~150 KiB, against a 64 KiB
# instruction cache shared by two CUs.   usage (gpurun): bash tools/pmc_icache.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_icache
mkdir -p $O
rocm-smi --showvbios --showbus --showcomputepartition --showmemorypartition 2>/dev/null | grep "GPU\[0\]" > $O/box.txt
VAR=$R/distributed_plonk_amd/lib/variants/quotient_unroll/libplonk_hip.so
run() {   # name, lib-or-empty, command...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export PLONK_HIP_LIB=$lib; else unset PLONK_HIP_LIB; fi
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/$name -o p -- "$@" > $O/$name.out 2> $O/$name.err
  python3 - $O/$name $name <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in ("ntt_pass_kernel", "quotient_evals_kernel", "msm_accumulate_kernel")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    a = {n: sum(v) / len(v) for n, v in c.items()}
    req = a.get("SQC_ICACHE_REQ", 0) or 1
    print(f"{tag:14s} {k[:46]:46s} icache req {req:.3e} hit {a.get('SQC_ICACHE_HITS', 0) / req:.3f} miss {a.get('SQC_ICACHE_MISSES', 0) / req:.3f} dup-miss {a.get('SQC_ICACHE_MISSES_DUPLICATE', 0) / req:.3f}  busy_cycles {a.get('SQ_BUSY_CYCLES', 0):.3e} launches {max(len(v) for v in c.values())}")
PY
  find $O/$name -name "*.csv" -delete
  grep -hE "quotient_fuse=0|coset_eval n|commit" $O/$name.out | head -3
}
cat $O/box.txt
run quot_product "" python $R/tools/quotient_bench.py 24
run quot_unroll "$VAR" python $R/tools/quotient_bench.py 24
run ntt "" python $R/tools/coset_eval_only.py 24
run msm "" python $R/tools/msm_only.py 24
