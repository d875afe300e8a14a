C="--steps 5 --warmup 1 --no-cpu-baseline --no-next-rows --no-other-configs"
timeout 14 python tools/bench_notorch.py --log-n 20 $C --overlap-phases on > gpurun_out/r04_bn20_on.json 2> gpurun_out/r04_bn20_on.err
timeout 8 python tools/bench_notorch.py --log-n 20 $C --overlap-phases off --no-verify > gpurun_out/r04_bn20_off.json 2> gpurun_out/r04_bn20_off.err
timeout 9 python tools/bench_notorch.py --log-n 22 --curve bls12_381 $C --overlap-phases on --no-verify > gpurun_out/r04_bls22_on.json 2> gpurun_out/r04_bls22_on.err
timeout 9 python tools/bench_notorch.py --log-n 22 --curve bls12_381 $C --overlap-phases off --no-verify > gpurun_out/r04_bls22_off.json 2> gpurun_out/r04_bls22_off.err
for f in bn20_on bn20_off bls22_on bls22_off; do python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/r04_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['phases_ms']['transforms'], d['phases_ms']['commitments'], d['verified'], d['config']['phase_overlap'])
except Exception as e: print('$f', 'ERR', e)
"; done; tail -n 2 gpurun_out/r04_bn20_on.err
