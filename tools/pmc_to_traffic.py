#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/traffic.json (bench.py's `roofline.traffic`).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python tools/pmc_to_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 24 bn254 1

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, averaged over the launches of a kernel:
MI355X_MICROARCH.md §HBM — on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes (128-byte
requests tallied at 64 B), so it is doubled; WRITE_SIZE is taken as reported.  Both are separate passes
(FETCH_SIZE uses 3 of the 4 TCC slots, WRITE_SIZE 2).  The raw averages are kept next to the corrected sum.
"""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(os.path.join(path, "p_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def short(name):
    n = name.split("(")[0].replace("void ", "").strip()
    return n


def main():
    fetch_dir, write_dir, log_n, curve, world = sys.argv[1:6]
    fetch, cnt = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        db = json.load(open(out_path))
    except Exception:
        db = {}
    groups = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for k in fetch:
        s = short(k)
        base = s.split("<")[0]
        for name in {s, base}:
            g = groups[name]
            g[0] += fetch[k] * cnt[k]
            g[1] += write.get(k, 0.0) * cnt[k]
            g[2] += cnt[k]
    for name, (f, w, c) in groups.items():
        key = f"{name}@2^{log_n}@{curve}@{world}"
        db[key] = round((2 * f + w) / c * 1024)
        db[key + "#raw"] = {"FETCH_SIZE_KiB_avg": round(f / c, 1), "WRITE_SIZE_KiB_avg": round(w / c, 1), "launches": c}
    json.dump(db, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote", out_path, len(groups), "kernels")


if __name__ == "__main__":
    main()
