#!/usr/bin/env python3
"""rocprofv3 --pmc passes of bench.py  ->  one small JSON (profiles/pmc_current.json) that bench.py quotes in `roofline.traffic` /
`roofline.valu_issue`, stamped with the kernel-source hash so a stale file is never quoted for different kernels.

    python tools/pmc_collect.py <fetch_dir> <write_dir> <sq_dir> <log_n> <curve> <world> <out.json>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB averaged over a kernel's launches (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE
tallies a 128-byte request at 64 B, so streaming reads are doubled; WRITE_SIZE as reported).  The doubling is calibrated for wide
coalesced reads only — for the MSM's 72-byte gathers see profiles/r02_fetch_calibration.txt; raw counters are kept beside the sum."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rows(d):
    paths = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for path in paths:
        with open(path) as f:
            yield from csv.DictReader(f)


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    fetch_dir, write_dir, sq_dir, log_n, curve, world, out_path = sys.argv[1:8]
    from distributed_plonk_amd.build import source_hash
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (fetch_dir, write_dir, sq_dir):
        for r in rows(d):
            k = short(r["Kernel_Name"])
            for name in {k, k.split("<")[0]}:
                per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for name, ctrs in sorted(per.items()):
        row = {c: sum(v) / len(v) for c, v in ctrs.items()}
        ent = {"launches": max(len(v) for v in ctrs.values())}
        if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
            ent["traffic_bytes"] = round((2 * row["FETCH_SIZE"] + row["WRITE_SIZE"]) * 1024)
            ent["FETCH_SIZE_KiB"] = round(row["FETCH_SIZE"], 1)
            ent["WRITE_SIZE_KiB"] = round(row["WRITE_SIZE"], 1)
        for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES"):
            if c in row:
                ent[c] = round(row[c])
        kernels[name] = ent
    from distributed_plonk_amd.build import code_hashes
    out = {"source_hash": source_hash(), "code_hashes": code_hashes(), "config": f"2^{log_n}@{curve}@{world}", "kernels": kernels,
           "how": "rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench.py --headline op-mix --steps 1 --warmup 1 --no-cpu-baseline "
                  "--no-next-rows --no-other-configs --no-verify ; averages per launch (the op-mix step: the kernels and launch shapes of a proof)"}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print("wrote", out_path, len(kernels), "kernels, source hash", out["source_hash"])


if __name__ == "__main__":
    main()
