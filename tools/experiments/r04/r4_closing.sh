#!/bin/bash
# Round 4 closing collection on the SHIPPED sources: the whole -m gpu suite, the driver-style bench line, kernel stats + PMC
# (tools/collect_profiles.sh: pmc_current.json gets the shipped source hash), BLS12-381 alone under the SQ counters.
#   gpurun --timeout 3000 -- 'bash tools/experiments/r04/r4_closing.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== the whole -m gpu suite"
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/r4close_gpu_suite.txt
echo "== smoke()"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/r4close_smoke.txt
echo "== driver-style bench"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_2p24_final.json 2> $O/r4close_bench.err; echo "rc $?"
echo "== kernel stats + PMC, 2^24 BN254"
timeout 1200 bash tools/collect_profiles.sh r04 2>&1 | tail -12
echo "== BLS12-381 2^22 alone: SQ counters (MSM, then the zero-padded coset FFT)"
cd /tmp && export TMPDIR=/tmp
for what in msm ntt; do
  if [ $what = msm ]; then CMD="python $R/tools/msm_only.py 22"; else CMD="python $R/tools/coset_eval_only.py 22"; fi
  CURVE=bls12_381 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS \
      --output-format csv -d $O/pmc_bls_$what -o p -- $CMD > $O/r4close_bls_$what.out 2> $O/r4close_bls_$what.err
  CURVE=bls12_381 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $O/pmc_bls_${what}_mem -o p -- $CMD > /dev/null 2>> $O/r4close_bls_$what.err
done
cd $R
python3 - <<'PY' | tee $O/r04_pmc_bls12_381_2p22.txt
import csv, glob, collections
for what in ("msm", "ntt"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (f"gpurun_out/pmc_bls_{what}", f"gpurun_out/pmc_bls_{what}_mem"):
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if any(s in k for s in ("msm_accumulate_kernel", "msm_reduce_level_kernel", "msm_points_sum", "sort_scatter", "sort_partition", "ntt_pass_kernel")):
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== BLS12-381 2^22, {what} alone (tools/{'msm_only' if what == 'msm' else 'coset_eval_only'}.py 22), averages per launch")
    for k, c in sorted(acc.items()):
        row = {n: sum(v) / len(v) for n, v in c.items()}
        extra = ""
        if "FETCH_SIZE" in row:
            extra = "  HBM bytes (2*FETCH + WRITE, KiB -> GB): %.3f" % ((2 * row["FETCH_SIZE"] + row.get("WRITE_SIZE", 0)) * 1024 / 1e9)
        print("  %-60s launches %3d %s%s" % (k[:60], max(len(v) for v in c.values()), {n: round(x) for n, x in sorted(row.items()) if n.startswith("SQ_")}, extra))
PY
find $O/pmc_bls_msm $O/pmc_bls_ntt $O/pmc_bls_msm_mem $O/pmc_bls_ntt_mem -name "*.csv" -delete 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_2p24_final.json"))
print("step", d["ms_per_step"], d["phases_ms"], "verified", d["verified"], "proof", d.get("proof_ms"), d.get("prover_verified"), d.get("proof_variants_ms"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "traffic_note")})
for oc in d.get("other_configs") or []:
    print({k: oc.get(k) for k in ("config", "ms_per_step", "phases_ms", "verified", "proof_ms", "prover_verified", "error")})
print("next rows", {k: v.get("ms") for k, v in (d.get("next_rows") or {}).items() if isinstance(v, dict) and "ms" in v})
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind")} if d.get("cpu_baseline") else None)
PY
