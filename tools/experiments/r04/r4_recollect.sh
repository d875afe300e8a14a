#!/bin/bash
# After a comment-only edit of csrc/quotient.hip (machine code unchanged, source hash changed): the new test of the collective order, the quotient and
# distributed suites, and the kernel stats + PMC passes again so that profiles/pmc_current.json carries the source hash of the library that ships.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_quotient.py -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/r4re_tests.txt
timeout 900 bash tools/collect_profiles.sh r04 2>&1 | tail -6
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-other-configs > $O/r4re_bench_check.json 2> /dev/null
python -c "
import json; d=json.load(open('$O/r4re_bench_check.json')); r=d['roofline']; print('step', d['ms_per_step'], 'verified', d['verified'], 'traffic', r['traffic'], 'note', r['traffic_note'], 'valu', r['valu_issue'])"
