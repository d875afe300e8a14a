#!/bin/bash
# Round 4, GPU call 4: the per-unit pin / SLP flags in the product build (MSM: no pins), the compact quotient kernel, the refactored bench.py.
#   gpurun --timeout 2400 -- 'bash tools/experiments/r04/r4_call4.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
echo "== parity of the product build (MSM without pins, quotient variants 6 / 7)"
timeout 1200 python -m pytest tests/test_gpu_field.py tests/test_gpu_msm.py tests/test_gpu_msm_table.py tests/test_gpu_commit_many.py tests/test_gpu_quotient.py tests/test_gpu_golden.py \
    tests/test_gpu_polyops.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee $O/r4c4_parity.txt
for V in "" q_use q_none; do
  echo "== quotient kernel, build ${V:-product}"
  ( [ -n "$V" ] && export PLONK_HIP_LIB=$R/distributed_plonk_amd/lib/variants/$V/libplonk_hip.so; timeout 300 python tools/quotient_bench.py 24 2>&1 | grep quotient_fuse )
done | tee $O/r4c4_quotient.txt
echo "== bench.py, default run (driver style)"
timeout 900 python bench.py > $O/r4c4_bench_default.json 2> $O/r4c4_bench_default.err; echo "rc $?"
for V in poly_none poly_use; do
  PLONK_HIP_LIB=$R/distributed_plonk_amd/lib/variants/$V/libplonk_hip.so timeout 600 python bench.py --no-cpu-baseline --no-other-configs --steps 2 > $O/r4c4_bench_$V.json 2>> $O/r4c4.err
done
python - <<'PY' | tee $O/r4c4_summary.txt
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4c4_bench_*.json")):
    try:
        d = json.load(open(f))
        nr = d.get("next_rows") or {}
        pr = nr.get("prover_rounds") or {}
        print(os.path.basename(f), "step", d["ms_per_step"], d["phases_ms"]["transforms"], d["phases_ms"]["commitments"], "verified", d["verified"], "proof", d.get("proof_ms"), d.get("prover_verified"),
              "rounds", pr.get("rounds_ms"))
        print("   rows:", {k: v.get("ms") for k, v in nr.items() if isinstance(v, dict) and "ms" in v and k != "prover_rounds"})
        print("   variants:", d.get("proof_variants_ms"))
        for oc in d.get("other_configs") or []:
            print("   ", {k: oc.get(k) for k in ("config", "ms_per_step", "phases_ms", "verified", "proof_ms", "prover_verified", "error")})
        print("   kernels:", {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items() if "<" not in k})
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -5 $O/r4c4_bench_default.err
