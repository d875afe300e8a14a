#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_* counter_collection.csv -> per-kernel averages per launch (profiles/rNN_pmc_sq_*.json).

    python tools/pmc_sq_to_json.py gpurun_out/pmc_sq/<...>/p_counter_collection.csv profiles/r01_pmc_sq_2p24.json
"""
import collections
import csv
import json
import sys


def main():
    src, dst = sys.argv[1:3]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(src) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for name, ctrs in sorted(acc.items()):
        row = {c: int(round(sum(v) / len(v))) for c, v in sorted(ctrs.items())}
        row["launches"] = max(len(v) for v in ctrs.values())
        out[name] = row
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst, len(out), "kernels")


if __name__ == "__main__":
    main()
