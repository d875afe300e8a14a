#!/usr/bin/env python3
"""python tools/fetch_calib_report.py <rocprof dir> : FETCH_SIZE per launch vs the known byte counts of tools/fetch_calib.hip."""
import collections, csv, glob, os, sys
known = {"calib_stream_kernel": 4 << 30, "calib_gather72_kernel": (1 << 26) * 72}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, ctrs in sorted(acc.items()):
    if k not in known:
        continue
    for c, v in sorted(ctrs.items()):
        avg = sum(v) / len(v)
        if c == "FETCH_SIZE":
            print(f"{k}: FETCH_SIZE {avg:.0f} KiB/launch = {avg * 1024 / known[k]:.3f} x useful bytes ({known[k]} B); x2-corrected: {2 * avg * 1024 / known[k]:.3f}")
        else:
            print(f"{k}: {c} {avg:.0f} per launch ({avg / (known[k] / 72 if 'gather' in k else known[k] / 64):.3f} per record/64B)")
