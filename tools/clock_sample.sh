#!/bin/bash
# Sustained shader clock while the headline step runs (the VALU-issue bound in bench.py assumes the 2.4 GHz peak clock)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 120 python bench.py --steps 12 --warmup 2 --no-next-rows --no-cpu-baseline --no-other-configs --no-verify > gpurun_out/clk_bench.json 2>/dev/null ) &
BP=$!
sleep 6
: > gpurun_out/clk_samples.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" >> gpurun_out/clk_samples.txt
  echo "--" >> gpurun_out/clk_samples.txt
  sleep 0.5
done
wait $BP
python - <<'PY'
import re, json
sclk = [int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", open("gpurun_out/clk_samples.txt").read())]
pw = [float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", open("gpurun_out/clk_samples.txt").read())]
print("samples", len(sclk), "sclk MHz min/avg/max", min(sclk or [0]), sum(sclk) / max(len(sclk), 1), max(sclk or [0]), "power W avg", sum(pw) / max(len(pw), 1) if pw else None)
d = json.loads(open("gpurun_out/clk_bench.json").read().strip().splitlines()[-1]); print("ms_per_step", d["ms_per_step"])
PY
head -12 gpurun_out/clk_samples.txt
