#!/bin/bash
# kernel-trace (start / end timestamps per dispatch) of one step: how the two commitment contexts' kernels overlap on the GPU.
# usage (gpurun): bash tools/trace_commit_overlap.sh ["batch persist" ...]     then: python tools/trace_show.py gpurun_out/trace_b<batch>_p<persist>.tsv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
COMMON="--steps 1 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs --no-verify ${EXTRA:-}"
if [ $# -eq 0 ]; then set -- "1 4"; fi
for cfg in "$@"; do
  read b p <<< "$cfg"
  T=trace_b${b}_p${p}
  PLONK_BENCH_COMMIT_BATCH=$b PLONK_BENCH_ACC_PERSIST=$p rocprofv3 --kernel-trace --output-format csv -d $O/$T -o t -- python $R/bench.py $COMMON > $O/$T.json 2> $O/$T.err
  f=$(find $O/$T -name "*kernel_trace.csv" | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
out=open("$O/$T.tsv","w")
for r in rows:
    nm=r["Kernel_Name"].split("(")[0].replace("void ","")[:40]
    out.write("\t".join([nm, r.get("Queue_Id",""), r.get("Stream_Id",""), r["Start_Timestamp"], r["End_Timestamp"]])+"\n")
out.close()
PY
  rm -rf $O/$T
done
ls -la $O/*.tsv
