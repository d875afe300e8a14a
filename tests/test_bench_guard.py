"""bench.py's result line must survive an optional leg that never returns (an N > 1 collective that hangs): the watchdog prints
the line that exists by then and ends the process with exit code 0.  CPU only: the guard is exercised in a subprocess."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HUNG_LEG = r"""
import sys, time
sys.path.insert(0, {root!r})
import bench
g = bench.ResultLine(1, {rank}, {{"value": 42.0}} if {rank} == 0 else None)
g.start_watchdog(poll_s=0.05)
g.arm("class_prover", 0.3)
time.sleep(120)                 # the leg that never comes back
print("NOT REACHED")
"""

FINISHED_LEG = r"""
import sys, time
sys.path.insert(0, {root!r})
import bench
out = {{"value": 42.0}}
g = bench.ResultLine(1, 0, out)
g.start_watchdog(poll_s=0.05)
g.arm("other_scheme", 5.0)
out["other_scheme"] = {{"ms_per_step": 1.0}}
g.arm(None, 0)
time.sleep(0.3)                 # a disarmed watchdog stays quiet
g.emit()
g.emit()                        # only one line, ever
"""


def _run(src, **kw):
    return subprocess.run([sys.executable, "-c", src.format(root=ROOT, **kw)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)


def test_hung_leg_costs_the_leg_not_the_line():
    r = _run(HUNG_LEG, rank=0)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] == 42.0 and d["aborted_optional_leg"]["leg"] == "class_prover"


def test_other_ranks_leave_quietly():
    r = _run(HUNG_LEG, rank=3)
    assert r.returncode == 0 and r.stdout.decode().strip() == ""


def test_finished_leg_prints_one_complete_line():
    r = _run(FINISHED_LEG)
    assert r.returncode == 0, r.stderr.decode()
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert "aborted_optional_leg" not in d and d["other_scheme"]["ms_per_step"] == 1.0


def test_polynomial_parallel_assignment_covers_the_step_once_and_balances():
    """bench.py's polynomial-level-parallel leg (whole operations per rank, no data-path collective): every operation of the
    step — 13 commitments, 25 forward 8n coset FFTs, the 8n coset iFFT, 7 size-n iNTTs — lands on exactly one rank, and the modelled load
    of the busiest rank stays within one commitment of the mean (longest-processing-time-first)."""
    sys.path.insert(0, ROOT)
    import bench
    want = {("commit", i) for i in range(13)} | {("coset_fft_8n", i) for i in range(25)} | {("coset_ifft_8n", 0)} | {("intt_n", i) for i in range(7)}
    for ranks in (1, 2, 3, 4, 8, 16, 64):
        mine, load = bench.poly_parallel_assignment(ranks)
        flat = [op for ops in mine for op in ops]
        assert len(flat) == len(want) == 46 and set(flat) == want, ranks
        for ops, l in zip(mine, load):
            assert abs(sum(bench.POLY_OP_COST[k] for k, _ in ops) - l) < 1e-9
        assert max(load) <= sum(load) / ranks + max(bench.POLY_OP_COST.values()), (ranks, load)
    _, load8 = bench.poly_parallel_assignment(8)
    assert sum(load8) / max(load8) > 7.5          # 8 ranks: within 7 % of a perfect split before any measurement
    # the n-domain-only step (configs[4]) has no 8n transform
    mine, _ = bench.poly_parallel_assignment(4, nbig=0)
    assert sorted(op for ops in mine for op in ops) == sorted([("commit", i) for i in range(13)] + [("intt_n", i) for i in range(7)])


def test_pmc_numbers_are_quoted_only_for_the_code_they_were_collected_from(tmp_path):
    """bench.load_pmc: the committed PMC collection is quoted for its own workload only, for all kernels when the kernel sources are the
    collection's, and otherwise kernel by kernel when the machine-code hash recorded with the collection equals the built library's."""
    sys.path.insert(0, ROOT)
    import bench
    from distributed_plonk_amd import build
    pmc, note = bench.load_pmc("2^20@bn254@1")
    assert pmc == {} and "not this workload" in note
    assert bench.load_pmc("2^24@bn254@1", dense_coset=True)[0] == {}
    assert bench.load_pmc("2^24@bn254@1", path=str(tmp_path / "missing.json")) == ({}, bench.load_pmc("2^24@bn254@1", path=str(tmp_path / "missing.json"))[1])
    kern = {"ntt_pass_kernel": {"traffic_bytes": 1}, "ntt_pass_kernel<8, 4, true, true>": {"traffic_bytes": 2}, "msm_accumulate_kernel": {"traffic_bytes": 3}}
    # same sources: everything
    f1 = tmp_path / "same.json"
    f1.write_text(json.dumps({"config": "2^24@bn254@1", "source_hash": build.source_hash(), "kernels": kern}))
    assert bench.load_pmc("2^24@bn254@1", path=str(f1)) == (kern, None)
    # other sources: only kernels whose recorded machine-code hash is the library's
    now = build.code_hashes()
    if not now:
        return                                   # the library has not been built in this tree: nothing more to compare
    f2 = tmp_path / "other.json"
    host = {k_: v_ for k_, v_ in now.items() if k_.startswith(("host:", "unit_of:"))}
    f2.write_text(json.dumps({"config": "2^24@bn254@1", "source_hash": "0" * 16, "kernels": kern,
                              "code_hashes": dict(host, ntt_pass_kernel=now["ntt_pass_kernel"], msm_accumulate_kernel="f" * 16)}))
    pmc, note = bench.load_pmc("2^24@bn254@1", path=str(f2))
    assert sorted(pmc) == ["ntt_pass_kernel", "ntt_pass_kernel<8, 4, true, true>"] and "byte-identical" in note and "ntt_pass_kernel" in note
    # ... and whose LAUNCHING host code is the library's too (a kernel's counters depend on its launch shape): a profile taken with
    # another planner in ntt_engine, or other option defaults in plonk_api, is not quoted even for byte-identical kernel code
    for changed in ("host:ntt_engine", "host:plonk_api"):
        f4 = tmp_path / "planner.json"
        f4.write_text(json.dumps({"config": "2^24@bn254@1", "source_hash": "0" * 16, "kernels": kern,
                                  "code_hashes": dict(host, **{"ntt_pass_kernel": now["ntt_pass_kernel"], changed: "e" * 16})}))
        assert bench.load_pmc("2^24@bn254@1", path=str(f4))[0] == {}, changed
    f3 = tmp_path / "nohashes.json"
    f3.write_text(json.dumps({"config": "2^24@bn254@1", "source_hash": "0" * 16, "kernels": kern}))
    assert bench.load_pmc("2^24@bn254@1", path=str(f3))[0] == {}


def test_run_leg_turns_an_exception_into_an_error_field_and_disarms_the_watchdog():
    """Every optional leg of bench.py runs through benchlib.line.run_leg: an exception costs the leg's fields, never the line."""
    import bench

    class Guard:
        armed = []

        def arm(self, name, seconds):
            self.armed.append((name, seconds))

    g = Guard()
    assert bench.run_leg(g, "ok", 5.0, lambda: {"ms": 1.0}) == {"ms": 1.0}
    res = bench.run_leg(g, "boom", 5.0, lambda: 1 / 0)
    assert set(res) == {"error"} and "ZeroDivisionError" in res["error"]
    assert bench.run_leg(None, "unguarded", None, lambda: [][0], error=lambda ex: [{"error": repr(ex)}])[0]["error"].startswith("IndexError")
    assert g.armed == [("ok", 5.0), (None, 0), ("boom", 5.0), (None, 0)]


def test_the_bench_program_stays_reviewable():
    """bench.py is a thin entry point over benchlib/: no function of the program is longer than 150 lines (round 3's main() had 1200)."""
    import ast
    import glob
    files = [os.path.join(ROOT, "bench.py")] + sorted(glob.glob(os.path.join(ROOT, "benchlib", "*.py")))
    assert len(files) >= 8
    for f in files:
        for node in ast.walk(ast.parse(open(f).read())):
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
                assert node.end_lineno - node.lineno + 1 <= 150, (f, node.name)
    assert sum(1 for _ in open(os.path.join(ROOT, "bench.py"))) <= 200


def test_committed_pmc_profile_was_collected_from_the_shipped_sources():
    """`roofline.traffic` / `valu_issue` of the driver's bench line must come from counters collected on the sources that ship, not
    through the per-kernel machine-code exception: profiles/pmc_current.json carries the source hash of the kernel sources in this tree.  (After a
    kernel edit this fails until the closing collection — tools/closing_collection.sh on an MI355X — has been re-run and its pmc_current.json copied.)"""
    import bench
    from distributed_plonk_amd import build
    with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
        db = json.load(f)
    assert db["source_hash"] == build.source_hash(), "profiles/pmc_current.json is stale: re-run tools/closing_collection.sh on the GPU box and copy gpurun_out/pmc_current.json"
    pmc, note = bench.load_pmc("2^24@bn254@1")
    assert note is None and {"ntt_pass_kernel", "msm_accumulate_kernel"} <= set(pmc)
    assert pmc["ntt_pass_kernel"]["traffic_bytes"] > 2 ** 31 and pmc["msm_accumulate_kernel"]["SQ_INSTS_VALU"] > 10 ** 9


def test_bench_help_renders():
    """argparse formats help strings with `%`: a bare percent sign in one of them breaks `bench.py --help` (it did, once)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "--overlap-phases" in r.stdout, r.stderr[-1500:]


def test_other_configs_fall_back_to_the_phase_after_phase_run(monkeypatch):
    """benchlib/other_configs.py: the configs[1] / configs[3] sub-runs overlap their phases by default, under the headline watchdog; should such a
    run fail (exit code, time-out), the same configuration is run once more with --overlap-phases off and the proof's third context off,
    and the entry records what happened; both sub-runs and their fall-backs share ONE 240 s budget (the headline line is written
    after this leg, so its worst case must be minutes, not half an hour)."""
    import json
    import types
    import benchlib.other_configs as oc
    calls = []

    def fake_run(cmd, **kw):
        env = kw.get("env") or {}
        calls.append((list(cmd), env.get("PLONK_BENCH_WATCHDOG"), env.get("PLONK_BENCH_PROOF_HELPER"), kw.get("timeout")))
        if "--overlap-phases" not in cmd:
            raise subprocess.CalledProcessError(4, cmd)
        line = {"ms_per_step": 5.0, "value": 2.0, "steps": 3, "phases_ms": {"round1": 1, "round2": 2, "note": "x"}, "config": {"phase_overlap": False},
                "headline": "K verified five-round proofs", "op_mix": {"ms_per_step": 1.0, "constraints_per_s": 9.0, "phases_ms": {"transforms": 1, "commitments": 2, "note": "x"}},
                "roofline": {"kernel": "k", "frac": 0.1, "avg_launch_ms": 1}, "verified": True, "verification": {}, "proof_ms": 5.0, "prover_verified": True}
        return types.SimpleNamespace(stdout=json.dumps(line).encode())

    monkeypatch.setattr(oc.subprocess, "run", fake_run)
    res = oc.other_configs(types.SimpleNamespace(bases="distinct"))
    assert len(res) == 2 and all(r["phase_overlap"] is False and "CalledProcessError" in r["overlap_run_failed"] and r["verified"] for r in res), res
    # round 6: the sub-run's headline is its proof (ms_per_step = proof_ms), the op-mix step rides beside it; an un-overlapped run quotes its own frac
    assert all(r["ms_per_step"] == r["proof_ms"] == 5.0 and r["op_mix_ms_per_step"] == 1.0 and r["frac"] == 0.1 and "no context overlap" in r["frac_source"] for r in res), res
    assert [c[1:3] for c in calls] == [("1", None), (None, "0")] * 2
    assert all(20.0 <= c[3] <= 150.0 for c in calls) and oc.BUDGET_S <= 300
    assert all(c[0][-2:] == ["--overlap-phases", "off"] for c in calls[1::2])
    # a spent budget: the first attempt gets its 20-second floor, the fall-back is refused, and the entry says so
    monkeypatch.setattr(oc, "BUDGET_S", 0.0)
    calls.clear()
    res = oc.other_configs(types.SimpleNamespace(bases="distinct"))
    assert len(calls) == 2 and all(c[3] == 20.0 for c in calls) and all("no time left" in r["error"] for r in res), (calls, res)


def test_simulated_and_multipath_runs_never_quote_the_single_gpu_counters():
    """a --simulate-ranks / --multi-path line is a different workload from the N = 1 collection; its PMC key says so, so
    `traffic` and `valu_issue` stay null instead of pricing a 0.84 ms launch with the counters of a 4.2 ms one."""
    import types
    sys.path.insert(0, ROOT)
    from benchlib.headline import pmc_config_key
    import bench
    a = types.SimpleNamespace(log_n=24, curve="bn254")
    key = lambda **kw: pmc_config_key(types.SimpleNamespace(args=a, **kw))
    assert key(sim=0, multi=False, world=1) == "2^24@bn254@1"
    assert key(sim=8, multi=True, world=1) == "2^24@bn254@sim8"
    assert key(sim=0, multi=True, world=1) == "2^24@bn254@1-multipath"
    assert key(sim=0, multi=True, world=8) == "2^24@bn254@8"
    for k in ("2^24@bn254@sim8", "2^24@bn254@1-multipath"):
        pmc, note = bench.load_pmc(k)
        assert pmc == {} and "not this workload" in note


def test_no_committed_result_line_claims_more_valu_issue_than_time():
    """every bench line committed under profiles/ (and the driver's BENCH_r*.json): a kernel cannot issue for longer than it runs — a
    `valu_issue.frac_of_launch` above 1.2 (the 4.5-clk model's own slack) means the counters of another workload were quoted.  (The simulated /
    multi-path lines of rounds 3-4 that did — 4.3 and 6.0 — had those fields removed in round 5, with a note in the file.)"""
    import glob
    known = set()
    bad = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json")) + glob.glob(os.path.join(ROOT, "BENCH_r*.json"))):
        try:
            d = json.load(open(f))
        except Exception:       # noqa: BLE001 - multi-line logs
            continue
        d = d.get("parsed", d) if isinstance(d, dict) else {}
        if not isinstance(d, dict):
            continue
        for r in [d.get("roofline")] + list(d.get("roofline_other") or []):
            frac = ((r or {}).get("valu_issue") or {}).get("frac_of_launch")
            if frac is not None and frac > 1.2 and os.path.basename(f) not in known:
                bad.append((os.path.basename(f), r.get("kernel"), frac))
    assert not bad, bad


def test_other_configs_quote_the_unoverlapped_fraction_only():
    """a sub-run whose timed region overlaps contexts (2^20 / 2^22: --overlap-phases auto, Prover(fft_helper)) reports the roofline
    fraction of its UN-OVERLAPPED op-mix pass; without such a pass no fraction is quoted at all — never the one from stretched launches."""
    sys.path.insert(0, ROOT)
    from benchlib.other_configs import entry_of
    base = {"ms_per_step": 50.0, "value": 2.0e7, "steps": 3, "phases_ms": {"round1": 1.0, "note": "x"}, "config": {"phase_overlap": True}, "headline": "K proofs",
            "op_mix": {"ms_per_step": 40.0, "constraints_per_s": 2.6e7, "phases_ms": {"transforms_with_commitments_beside_them": 39.0,
                                                                                      "commitments_tail_after_the_last_transform": 1.0, "note": "x"}},
            "roofline": {"kernel": "ntt_pass_kernel", "frac": 0.03, "avg_launch_ms": 0.5, "overlap_note": "stretched"}, "verified": True, "proof_ms": 50.0}
    e = entry_of("c", dict(base, roofline_unoverlapped={"roofline": {"kernel": "ntt_pass_kernel", "frac": 0.055, "avg_launch_ms": 0.3}}))
    assert e["frac"] == 0.055 and e["avg_launch_ms"] == 0.3 and e["frac_in_the_overlapped_timed_region"] == {"kernel": "ntt_pass_kernel", "frac": 0.03}
    assert "roofline_unoverlapped" in e["frac_source"] and e["dominant_kernel"] == "ntt_pass_kernel"
    # the fraction quoted is ntt_pass_kernel's even where another kernel dominates the un-overlapped pass (BLS12-381: the accumulation); all of them sit beside it
    e2 = entry_of("c", dict(base, roofline_unoverlapped={"roofline": {"kernel": "msm_accumulate_kernel", "frac": 0.007, "avg_launch_ms": 18.0},
                                                         "roofline_other": [{"kernel": "ntt_pass_kernel", "frac": 0.05, "avg_launch_ms": 0.9}]}))
    assert e2["frac"] == 0.05 and e2["dominant_kernel"] == "ntt_pass_kernel" and e2["frac_by_kernel_unoverlapped"]["msm_accumulate_kernel"]["frac"] == 0.007
    assert "commitments" not in e["op_mix_phases_ms"] and "commitments_tail_after_the_last_transform" in e["op_mix_phases_ms"]
    e = entry_of("c", base)
    assert e["frac"] is None and e["frac_in_the_overlapped_timed_region"]["frac"] == 0.03 and "not quoted" in e["frac_source"]


class _FakeClock:
    """A deterministic clock for benchlib.cpu_baseline: every timed call takes exactly what the model says (the fitted exponent of
    2^8 / 2^10 WALL-CLOCK samples failed one run in four on a loaded box).  Consecutive clock() pairs bracket, in order: [ntt_par, ntt8_par,] ntt, ntt8,
    msm of each sample."""

    def __init__(self, durations):
        self.d, self.t, self.calls = list(durations), 0.0, 0

    def __call__(self):
        if self.calls % 2 == 1:
            self.t += self.d[self.calls // 2]
        self.calls += 1
        return self.t


def _cpu_args(**kw):
    import types
    return types.SimpleNamespace(**dict(dict(curve="bn254", log_n=12, cpu_sample_log_n=8, cpu_sample_log_n2=10, cpu_full_size="auto"), **kw))


def test_cpu_baseline_is_measured_at_the_bench_size_with_the_small_samples_beside_it():
    """`cpu_baseline.value` is a MEASUREMENT at the GPU line's size (`extrapolated: false`), the two smaller samples and the
    exponent they give stay beside it as a cross-check; with --cpu-full-size off (or no host memory) the figure is the fitted one, labelled.  The
    oracle really runs (tiny sizes); the clock is injected, so every number below is exact and the test does not depend on the box's load."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from benchlib.cpu_baseline import cpu_baseline
    from oracle import oracle as O
    bases = O.gen_bases(O.BN254, 3, 64, 1 << 12)
    # step time ~ n^1.1 exactly: sample s costs 2^(1.1 * log_n) * (7 * 1 + 26 * 8 + 13 * 4) time units of 1 us
    unit = lambda ln: 1e-6 * 2.0 ** (1.1 * ln)
    ops = lambda ln, par: ([0.5 * unit(ln), 4 * unit(ln)] if par else []) + [unit(ln), 8 * unit(ln), 4 * unit(ln)]
    clk = _FakeClock(ops(8, True) + ops(10, False) + ops(12, False))
    c = cpu_baseline(_cpu_args(), {}, bases, clock=clk)
    assert clk.calls == 2 * (5 + 3 + 3)
    assert c["kind"] == "port" and c["extrapolated"] is False and [s["log_n"] for s in c["samples"]] == [8, 10, 12]
    step12 = unit(12) * (7 + 26 * 8 + 13 * 4)
    assert c["value"] == c["value_at_bench_size"] == round(4096 / step12, 1) == c["samples"][2]["constraints_per_s"]
    assert "the GPU line's own size" in c["sample"] and "2^12" in c["sample"]
    assert c["fitted_exponent"] == 1.1
    x = c["extrapolated_to_bench_size"]
    assert abs(x["by_fitted_exponent"]["value"] / c["value"] - 1) < 1e-3 and "cross-check" in x["note"] and x["by_operation_counts"]["value"] > 0
    # rounds 4-5's form on request: two samples, the fitted figure, labelled as an extrapolation
    clk = _FakeClock(ops(8, True) + ops(10, False))
    c = cpu_baseline(_cpu_args(cpu_full_size="off"), {}, bases, clock=clk)
    assert c["extrapolated"] is True and [s["log_n"] for s in c["samples"]] == [8, 10] and c["value"] == c["samples"][1]["constraints_per_s"]
    assert c["value_at_bench_size"] == c["extrapolated_to_bench_size"]["by_fitted_exponent"]["value"] and "EXTRAPOLATED" in c["extrapolated_to_bench_size"]["note"]
    assert c["full_size_skipped"] == "--cpu-full-size off"
    # one sample only, at the bench size: nothing is extrapolated
    clk = _FakeClock(ops(8, True))
    c = cpu_baseline(_cpu_args(log_n=8, cpu_sample_log_n2=0), {}, bases[:256], clock=clk)
    assert c["extrapolated"] is False and c["value_at_bench_size"] == c["value"] and len(c["samples"]) == 1 and "extrapolated_to_bench_size" not in c
    assert isinstance(np.asarray(bases), np.ndarray)
