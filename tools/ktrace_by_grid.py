#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel, grid size), for the kernels named on the command line.

Why: `rocprofv3 --stats` averages a kernel over EVERY launch of the profiled command.  bench.py's command also runs set-up work (18 key
interpolations of n points, the op-mix steps before the proofs), so the stats average of `ntt_pass_kernel<8>` mixes n-point and 8n-point passes
in a different ratio than one timed proof does (21 : 78).  Grouped by grid size the two populations separate, and the 21 : 78 mix of one proof
can be recomputed from the profile and compared with the line's HIP-event `avg_launch_ms`.

    python tools/ktrace_by_grid.py <dir with *kernel_trace.csv> ntt_pass msm_accumulate
"""
import collections
import csv
import glob
import os
import sys


def main():
    d, pats = sys.argv[1], sys.argv[2:]
    groups = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name", "")
                if pats and not any(p in name for p in pats):
                    continue
                try:
                    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                except (KeyError, ValueError):
                    continue
                grid = tuple(int(r.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                wg = int(r.get("Workgroup_Size_X", 0) or 0)
                groups[(name.split("(")[0].replace("void ", "").strip(), grid, wg)].append(dur)
    print(f"{'kernel':56s} {'grid (work-items)':>26s} {'wg':>5s} {'launches':>9s} {'avg ms':>9s} {'min':>8s} {'max':>8s} {'total ms':>10s}")
    for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: (kv[0][0], -sum(kv[1]))):
        print(f"{name:56s} {str(grid):>26s} {wg:5d} {len(v):9d} {sum(v) / len(v):9.4f} {min(v):8.4f} {max(v):8.4f} {sum(v):10.2f}")


if __name__ == "__main__":
    main()
