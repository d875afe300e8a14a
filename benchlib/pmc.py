"""PMC-derived numbers (profiles/pmc_current.json) that a bench line may quote, and when."""
import json
import os

from .common import ROOT


def load_pmc(config, dense_coset=False, path=None):
    """-> (per-kernel PMC numbers bench.py may quote for this run, note or None).  profiles/pmc_current.json (tools/pmc_collect.py) is quoted
    when it holds this workload and was collected from the kernel sources the loaded library was built from (build.source_hash) — or,
    kernel by kernel, when the sources differ but that kernel's gfx950 MACHINE CODE (every instantiation, hashed from the objects:
    distributed_plonk_amd/codehash.py) is byte-identical to what the counters ran: an edit elsewhere (an error path of the C ABI, a new
    kernel beside it) does not touch it.  Otherwise nothing is quoted and the note says why."""
    try:
        from distributed_plonk_amd.build import code_hashes, source_hash
        with open(path or os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            db = json.load(f)
        if db.get("config") != config or dense_coset:
            return {}, f"profiles/pmc_current.json holds {db.get('config')} (padded coset inputs): not this workload"
        if db.get("source_hash") == source_hash():
            return db["kernels"], None
        now, then = code_hashes(), db.get("code_hashes") or {}

        def host_same(k_):      # the launch shape: the host code of the kernel's unit and of plonk_api (option defaults) must be unchanged too
            u = now.get("unit_of:" + k_)
            return (u is not None and then.get("unit_of:" + k_) == u and now.get("host:" + u) == then.get("host:" + u)
                    and now.get("host:plonk_api") is not None and now.get("host:plonk_api") == then.get("host:plonk_api"))

        same = sorted(k_ for k_ in then if ":" not in k_ and now.get(k_) == then[k_] and host_same(k_))
        pmc = {k_: v_ for k_, v_ in db["kernels"].items() if k_.split("<")[0] in same}
        return pmc, (f"profiles/pmc_current.json was collected from other kernel sources ({db.get('source_hash')} != {source_hash()}); quoted only for "
                     f"kernels whose gfx950 machine code AND launching host code are byte-identical to the collection's: {', '.join(k_ for k_ in same if k_ in db['kernels']) or 'none'}")
    except Exception as ex:     # noqa: BLE001 - a missing or unreadable profile costs the PMC fields, never the run
        return {}, f"no PMC profile: {ex!r}"
