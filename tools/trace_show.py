"""Timeline of the commitment phase from a tools/trace_commit_overlap.sh .tsv: kernels above `min_us`, relative ms, queue and stream."""
import sys
f = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 150
t_lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0
t_hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
rows = [l.rstrip("\n").split("\t") for l in open(f)]
rows = [(r[0], r[1], r[2], int(r[3]), int(r[4])) for r in rows]
accs = [r for r in rows if r[0].startswith("msm_accumulate_kernel")]
half = accs[len(accs) // 2:]                      # the timed step (after the warm-up step)
tstart = max(r[3] for r in rows if r[0].startswith("msm_digits") and r[3] <= half[0][3])
tstart = min(r[3] for r in rows if r[0].startswith("msm_digits") and r[3] >= tstart - 20_000_000)
sel = [r for r in rows if r[3] >= tstart]
print("span ms", (max(r[4] for r in sel) - tstart) / 1e6)
for r in sel:
    a, b = (r[3] - tstart) / 1e6, (r[4] - tstart) / 1e6
    if (r[4] - r[3]) > min_us * 1000 and a >= t_lo and a <= t_hi:
        print("  %-36s q%s s%s  %8.2f -> %8.2f  (%.2f ms)" % (r[0][:36], r[1], r[2], a, b, b - a))
