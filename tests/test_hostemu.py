"""The HIP kernels EXECUTED on the CPU (no GPU in this container): tests/hostemu compiles distributed_plonk_amd/csrc/*.hip unchanged
with g++ against an emulation of the HIP runtime (workgroups as fibers, block / wave barriers as scheduler yields, LDS as thread-local
storage, shared memory under the library's comm_* interface in place of RCCL) and this module runs a time-boxed selection of the
`-m gpu` parity tests against that library — the same test functions, the same oracle, bit-for-bit — plus `smoke()` and the
multi-rank programs of tests/test_gpu_multirank.py as 2 and 4 rank PROCESSES.

What this does and does not show: the kernels' logic (indexing, limb arithmetic, sort / accumulate / pyramid, pass planning, the host
orchestration and the N > 1 rank programs) is exercised and checked where no GPU exists; performance, LDS capacity, register pressure,
wave-lockstep or memory-model effects are not — those remain with `pytest -m gpu` on an MI355X.  The emulation is test infrastructure:
the package never loads it (`_ffi.lib()` refuses the library unless PLONK_ALLOW_HOSTEMU=1, which only this harness sets).

The whole `-m gpu` suite under the emulation (≈ 25 min on 8 cores; everything but the full-size and torch.cuda-transport tests passes):
    python -m tests.hostemu.build && PLONK_HIP_LIB=tests/hostemu/_build/plain/libplonk_hostemu.so PLONK_ALLOW_HOSTEMU=1 \\
        HIPEMU_DEVICES=4 python -m pytest tests -m gpu -q
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu_env():
    sys.path.insert(0, ROOT)
    from tests.hostemu import build as emu_build
    lib = emu_build.build(verbose=False)
    env = dict(os.environ)
    env.update(PLONK_HIP_LIB=lib, PLONK_ALLOW_HOSTEMU="1", HIPEMU_DEVICES="4", HIPEMU_THREADS=str(min(8, os.cpu_count() or 1)))
    return env


def _pytest(env, files, k=None, timeout=900, extra_env=None):
    e = dict(env)
    e.update(extra_env or {})
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", *files]
    if k:
        cmd += ["-k", k]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout, tail
    return r.stdout


def test_package_refuses_the_emulation_without_opt_in(emu_env):
    """distributed_plonk_amd has no CPU path: pointing PLONK_HIP_LIB at the emulation is an error unless the harness opted in."""
    env = dict(emu_env)
    del env["PLONK_ALLOW_HOSTEMU"]
    r = subprocess.run([sys.executable, "-c", "from distributed_plonk_amd import _ffi; _ffi.lib()"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "host EMULATION" in r.stderr


def test_smoke_entry_point(emu_env):
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=emu_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_field_ntt_golden_quotient_kernels(emu_env):
    _pytest(emu_env, ["tests/test_gpu_field.py", "tests/test_gpu_ntt.py", "tests/test_gpu_golden.py", "tests/test_gpu_quotient.py"])


def test_msm_kernels(emu_env):
    # every MSM parity test but the forced chunked level-2 sort (40 s of CPU); the persistent accumulation, redo / heavy paths,
    # forced windows, sharded MSMs and commit / round1 are in
    _pytest(emu_env, ["tests/test_gpu_msm.py"], k="not level2_sort_in_chunks")


def test_poly_kernels(emu_env):
    _pytest(emu_env, ["tests/test_gpu_polyops.py"], k="not valid_permutation_closes and not full_size")


def test_prover_and_verifier_on_emulated_device(emu_env):
    """The five rounds on the (emulated) device against the oracle prover, and device proofs accepted by the trapdoor verifier."""
    _pytest(emu_env, ["tests/test_gpu_prover.py"], k="rounds_match_oracle and (3-False or 6-True) or real_transcript and bn254 or rejects_unsatisfied "
                                                     "or six_coset and 4-False-bn254")
    _pytest(emu_env, ["tests/test_gpu_verifier.py"], k="synthetic_circuits and (4-coset8n-bn254 or 5-classes6-bn254) or oracle_circuit or trapdoor_key and bn254 "
                                                       "or unsatisfied")


def test_class_prover_ranks_as_threads(emu_env):
    _pytest(emu_env, ["tests/test_gpu_class_prover.py"], k="matches_oracle and 4-2-bn254 or sharded_commit_key and 5-2-bn254 or in_library_rccl or failure_in_one_gate_range")


def test_rank_programs_as_processes_world_2_and_4(emu_env):
    """tests/test_gpu_multirank.py's rank programs — RankProver.fft_dev in all four modes at n and 8n on two contexts, the zero-padded row
    pass, a round of sharded commitments through the point all-gather, and the whole ClassProver with a sharded key, the transcript
    on every rank and the verifier on rank 0 — one process per rank through plonk_comm_*."""
    out = _pytest(emu_env, ["tests/test_gpu_multirank.py"], k="bn254 and (2] or 9-4])", extra_env={"HIPEMU_THREADS": "2"})
    assert "3 passed" in out, out
    # collectives entered on different communicators / of different kinds on different ranks: refused on every rank, nothing hangs (world 2 and 4)
    out = _pytest(emu_env, ["tests/test_gpu_multirank.py"], k="different_orders and (2] or 4])", extra_env={"HIPEMU_THREADS": "2"}, timeout=600)
    assert "2 passed" in out, out


def test_distributed_transform_steps_and_coset_classes(emu_env):
    """fft_init / fft1 / fft2_prepare / fft2 per step against the oracle's helpers, S = 1, 2, 4 workloads in one process, the zero-padded row
    pass up to 2^19, call-order errors; the zero-padding-aware coset FFT for every class count (the 2^20 + 3 case and the full-size
    cross-check stay with the GPU)."""
    _pytest(emu_env, ["tests/test_gpu_distributed.py"], k="not (25-2 or 24-4 or 22-2 or rccl or two_contexts)")
    _pytest(emu_env, ["tests/test_gpu_coset_classes.py"], k="not full_size and not 1048579")


def test_batched_commitments_and_fixed_base_table(emu_env):
    _pytest(emu_env, ["tests/test_gpu_commit_many.py"], k="not full_size and not group_limit")
    _pytest(emu_env, ["tests/test_gpu_msm_table.py"], k="bn254")


def test_compiled_cpp_host_program(emu_env, tmp_path):
    """tests/host_cpp/host_check.cpp — the reference's worker / dispatcher orchestration as compiled C++ on the bare C ABI (host/plonk_host.hpp):
    distributed FFT in all four modes on S = 1, 2, 4 in-process workers, sharded MSM, commit_polynomial, a batched prover round — with no Python
    between the program and the (emulated) library; BN254 here, both curves on the GPU (tests/test_host_cpp.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpp import _build
    from oracle import oracle as O
    O.lib()
    exe = _build(tmp_path)
    shadow = tmp_path / "emu"
    shadow.mkdir()
    os.symlink(emu_env["PLONK_HIP_LIB"], shadow / "libplonk_hip.so")            # DT_RUNPATH yields to LD_LIBRARY_PATH
    env = dict(emu_env, LD_LIBRARY_PATH=str(shadow) + os.pathsep + emu_env.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([exe, os.path.join(ROOT, "oracle", "libplonk_oracle.so"), "0"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0 and "host_check ok" in res.stdout, (res.stdout + res.stderr)[-2000:]


def test_compiled_cpp_prover_program(emu_env, tmp_path):
    """tests/host_cpp/prover_check.cpp — the five rounds of `Prover::prove` with their merlin transcript as compiled C++ on the bare C ABI
    (host/plonk_prover.hpp), against the (emulated) library: verifying key, challenges, proof and serialization checked against the oracle's
    rounds and the Python transcript (tests/test_host_cpp.py: check_cpp_prover), both curves (larger sizes on the GPU)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_cpp import _build_prover, check_cpp_prover
    from oracle import oracle as O
    O.lib()
    exe = _build_prover(tmp_path)
    shadow = tmp_path / "emu"
    shadow.mkdir()
    os.symlink(emu_env["PLONK_HIP_LIB"], shadow / "libplonk_hip.so")            # DT_RUNPATH yields to LD_LIBRARY_PATH
    env = dict(emu_env, LD_LIBRARY_PATH=str(shadow) + os.pathsep + emu_env.get("LD_LIBRARY_PATH", ""))
    check_cpp_prover(tmp_path, exe, "bn254", 0, 5, seed=1300, env=env)
    check_cpp_prover(tmp_path, exe, "bls12_381", 1, 4, seed=1301, env=env)          # the six-limb Fq path of the point encodings


def _bench_dry_run(env, world, port, extra=()):
    """`bench.py --gpus N` exactly as the driver launches it (torch.distributed.run, one rank per device), against the emulation at a tiny size:
    a DRY RUN of the program's control flow.  The line it prints says `emulated`, has no value and no clock-derived field."""
    import json
    e = dict(env, HIPEMU_DEVICES="8", HIPEMU_THREADS="2", PLONK_BENCH_HEADLINE_BUDGET_S="800", PLONK_BENCH_LEG_BUDGET_S="600")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", str(world), "--steps", "1", "--warmup", "1", "--log-n", "9", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=1200)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world,extra", [(2, ()), (4, ("--scheme", "classes"))])
def test_bench_program_multi_rank_control_flow(emu_env, world, extra):
    """The N > 1 legs of bench.py have never met more than one real GPU; here every one of them executes: two lanes of distributed transforms
    with an exchange each, the sharded batched commitments and their point all-gather, the other scheme, the polynomial-level-parallel leg (whole operations per
    rank, no data-path collective), the verification leg (distributed iNTT and zero-padded coset FFT against single-rank recomputation on every rank,
    sharded commitment against all shards on one rank) and the class prover with a sharded key, whose proof rank 0 hands to the verifier."""
    from conftest import free_port
    d = _bench_dry_run(emu_env, world, free_port(), extra)
    assert d["emulated"] is True and d["value"] is None and d["n_gpus"] == world
    assert not [k for k in d if k.endswith("_ms") or k.startswith("ms_")], "an emulated run must not carry timings"
    assert d["verified"] is True and all(d["verification"].values()), d["verification"]
    assert d.get("aborted_optional_leg") is None
    assert "error" not in (d["other_scheme"] or {}), d["other_scheme"]
    pp = d["polynomial_parallel"]            # whole operations per rank: checked on every rank, one gathered commitment against the oracle
    assert pp["verified"] is True and pp["ranks"] == world and pp["data_path_collectives_per_step"] == 0, pp
    assert sum(sum(r.values()) for r in pp["operations_per_rank"]) == 7 + 26 + 13, pp
    cp = d["next_rows"]["class_prover"]
    assert cp["ranks"] == world and cp["accepted_by_verifier"] is True, cp
    # round 6: the class prover's K proofs ARE the headline of an N > 1 run (accepted by the verifier, or the line falls back to the op-mix step)
    assert d["headline"].startswith("K verified five-round proofs by the coset-class prover on all") and d["prover_verified"] is True, d["headline"]
    assert "proof_headline_error" not in d and d["op_mix"] == {"emulated": True}


@pytest.mark.parametrize("extra", [("--sim-exchange", "standin"), ("--sim-exchange", "none", "--scheme", "classes")])
def test_bench_program_simulated_ranks(emu_env, extra):
    """`bench.py --simulate-ranks 4`: rank 0's share of a 4-rank job on one (emulated) device — the reference2d / classes step with the library's
    exchange stand-in (device-to-device copies of the foreign blocks) or none, phases overlapped (auto), the polynomial-parallel leg's busiest
    rank and the class prover with rounds 1-2 distributed, against the stand-in communicator.  Results are garbage by construction; what is
    checked is that every leg runs and that the line says what it is (and quotes nobody else's counters)."""
    import json
    e = dict(emu_env, HIPEMU_DEVICES="4", HIPEMU_THREADS=str(min(8, os.cpu_count() or 1)))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--log-n", "9", "--simulate-ranks", "4", *extra], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["emulated"] is True and d["sim_exchange"]["mode"] == extra[1] and d["config"]["phase_overlap"] is True
    assert "SIMULATED rank 0 of 4" in d["config"]["parallelism"] and ("device-to-device" in d["config"]["parallelism"]) == (extra[1] == "standin")
    assert d["sim_exchange"]["bytes_out_per_rank_and_step"] == 3 * (7 * (512 // 16) * 32 + 26 * (4096 // 16) * 32)
    cp = d["next_rows"]["class_prover"]
    assert cp["simulated"] is True and cp["ranks"] == 4 and cp["rounds_1_2"].startswith("size-n iFFTs by residue class"), cp
    assert (cp["sim_exchange"]["device_bytes_out_per_proof"] > 0) and "error" not in (d.get("polynomial_parallel") or {}), d
    assert "SIMULATED ranks" in d["headline"] and "proof_headline_error" not in d, d["headline"]


def test_bench_program_multi_rank_with_phases_overlapped(emu_env):
    """`--overlap-phases on` at N > 1 ('auto' chooses it since round 5's measurement): the two transform lanes on contexts and
    communicators of their own, the distributed transforms issued from the main thread while the commitment threads run, the point all-gather
    after the last all-to-all — same order of collectives on every rank, or this run would hang in the shared-memory communicator."""
    from conftest import free_port
    d = _bench_dry_run(emu_env, 2, free_port(), ("--overlap-phases", "on", "--no-class-prover", "--no-poly-parallel"))
    assert d["emulated"] is True and d["config"]["phase_overlap"] is True and d["config"]["rccl"]["communicators_per_rank"] == 4
    assert d["config"]["rccl"]["ranks_seen"] == [0, 1] and d["config"]["rccl"]["world_seen_by_every_rank"] == [2], d["config"]["rccl"]
    assert d["verified"] is True and all(d["verification"].values()), d["verification"]
    assert d.get("aborted_optional_leg") is None and "error" not in (d["other_scheme"] or {}), d


@pytest.mark.parametrize("overlap", ["off", "auto"])
def test_bench_program_single_rank_line_has_every_field(emu_env, overlap):
    """`python bench.py` as the driver runs it at N = 1 (here: 2^7 gates on the emulation): the op-mix step, its verification against the oracle, the
    proof loop that takes the headline over (round 6), the next rows and the verifier all execute, and the line carries exactly these top-level
    fields (the round-3 program's, plus `headline`, `op_mix`, `roofline_unoverlapped`; an emulated line has no clock-derived field by construction).  "off": the step as the
    2^24 line runs it, one phase after the other; "auto": at this size the transforms are issued while the commitment threads run (the
    configs[1] / configs[3] sub-runs)."""
    import json
    e = dict(emu_env, HIPEMU_DEVICES="4", HIPEMU_THREADS=str(min(8, os.cpu_count() or 1)))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--log-n", "7", "--overlap-phases", overlap], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    assert set(d) == {"config", "cpu_baseline", "data", "dtype", "emulated", "headline", "higher_is_better", "kernels", "metric", "n_gpus", "next_rows", "op_mix",
                      "other_scheme", "prover_verified", "roofline", "roofline_other", "roofline_unoverlapped", "scaling", "steps", "unit", "value", "verification",
                      "verified", "vs_baseline", "warmup"}, sorted(d)
    assert d["verified"] is True and all(d["verification"].values()), d["verification"]
    assert d["prover_verified"] is True and d["headline"].startswith("K verified five-round proofs") and "proof_headline_error" not in d
    assert d["roofline_unoverlapped"] == {"ran": True}       # this size overlaps contexts in the timed region (phases and / or the proof's third context)
    assert set(d["next_rows"]) == {"quotient_evals_kernel", "perm_product", "poly_eval", "poly_lincomb_20_terms", "poly_div_linear", "prover_rounds"}
    assert all(v.get("same_proof_as_the_verified_one") for v in d["next_rows"]["prover_rounds"]["variants"].values()), d["next_rows"]["prover_rounds"]["variants"]
    assert set(d["config"]) >= {"workload", "log_n", "curve", "bases", "scheme", "parallelism", "coset_inputs", "commit_batching", "phase_overlap"}
    assert d["config"]["phase_overlap"] is (overlap == "auto")


def test_bench_program_proof_only_sub_run(emu_env):
    """`--next-rows proof` — what the configs[1] / configs[3] sub-runs of the default bench line use (benchlib/other_configs.py): the step, its
    verification and ONE verified proof, without the quotient row, the O(n) rows and the same-proof variants that do less than the
    reference's work.  At these sizes the proof runs with Prover(fft_helper=...) (round 5: measured, adopted); PLONK_BENCH_HELPER_AB=1 adds the
    same rounds without it as a variant, which must give the same proof."""
    import json
    e = dict(emu_env, HIPEMU_DEVICES="4", HIPEMU_THREADS=str(min(8, os.cpu_count() or 1)), PLONK_BENCH_HELPER_AB="1")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--log-n", "7", "--curve", "bls12_381", "--next-rows", "proof"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-3000:]
    d = json.loads(lines[0])
    assert d["verified"] is True and d["prover_verified"] is True and set(d["next_rows"]) == {"prover_rounds"}
    assert d["next_rows"]["prover_rounds"]["key_coset_ffts"].startswith("on a third context")
    var = d["next_rows"]["prover_rounds"]["variants"]          # only the A/B partner: the same rounds with the key coset FFTs inside round 3
    assert set(var) == {"key_coset_ffts_inside_round_3"} and var["key_coset_ffts_inside_round_3"]["same_proof_as_the_verified_one"] is True, var


def test_differential_fuzz_slice(emu_env):
    """A fixed-seed slice of tools/fuzz_abi.py (random operations, shapes, flags and options against the oracle); long runs are a manual tool —
    nineteen operations, from single kernels to whole proofs handed to the verifier; under AddressSanitizer it found the two defects recorded in DESIGN §5 and docs/HISTORY.md §0."""
    r = subprocess.run([sys.executable, "tools/fuzz_abi.py", "--seconds", "500", "--max-ops", "60", "--seed", "5", "--max-log", "10"], cwd=ROOT, env=emu_env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok: 60 operations" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_second_device_of_one_process_raises_its_own_lds_limits(emu_env):
    """The emulation enforces the runtime's dynamic-LDS rule per (device, kernel): a launch above 64 KiB aborts unless
    hipFuncSetAttribute(MaxDynamicSharedMemorySize) was called for that kernel ON THAT DEVICE.  A second worker on another device of the
    same process therefore only works because the library's guards are per device (plonk_internal.hpp: DeviceOnce) — a process-wide
    `static bool` would launch the NTT passes and the MSM sort of device 1 with device 0's raised limit only."""
    code = r"""
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
from distributed_plonk_amd._ffi import MsmWorkload
from oracle import oracle as O
for dev in (0, 1, 3):
    w = PlonkWorker(me=0, device=dev, curve="bn254")
    v = O.rand_fr(O.BN254, 5 + dev, 1 << 12)
    assert np.array_equal(w.ntt(v, False, True), O.ntt(O.BN254, v, False, True))
    bases = O.gen_bases(O.BN254, 7, 64, 1 << 11)
    sc = O.from_mont(O.BN254, O.rand_fr(O.BN254, 9, 1 << 11))
    w.init(bases, 1 << 11, 1 << 14)
    xy, inf = w.g1_to_affine(w.var_msm(MsmWorkload(0, 1 << 11), sc))
    oxy, oinf = O.jac_to_affine(O.BN254, O.msm(O.BN254, bases, sc, threads=4))
    assert inf == oinf and np.array_equal(xy, oxy)
    w.close()
print("devices ok")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=emu_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "devices ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_prover_key_coset_ffts_beside_rounds_1_and_2(emu_env):
    """Prover(fft_helper=...): the 18 proving-key coset FFTs of round 3 issued on a third context by a thread of their own while rounds 1 and 2
    run (built at the end of round 4 without a GPU: an option, off by default, kept off the `-m gpu` list until it has met the device).  Same
    proof as the plain prover, bit for bit; accepted by the verifier; an unsatisfied witness still
    raises and leaves no thread behind."""
    code = r"""
import threading
import numpy as np
from distributed_plonk_amd.worker import PlonkWorker
from distributed_plonk_amd.prover import Prover, WrongQuotientPolyDegree
from distributed_plonk_amd.synthetic import SyntheticInstance
from distributed_plonk_amd.transcript import PlonkTranscript
from oracle import bigint_ref as B, oracle as O, verifier_ref as V
for curve, cid, log_n in (("bn254", 0, 5),):
    TAU = 0x1234567 + cid
    w, c2, h = (PlonkWorker(me=0, device=0, curve=curve) for _ in range(3))
    inst = SyntheticInstance(w, log_n, seed=77 + cid, num_inputs=3, tau=TAU, helpers=(c2, h))
    bl = {"wires": O.rand_fr(cid, 3, 10).reshape(5, 2, 4), "perm": O.rand_fr(cid, 4, 3)}
    pub = inst.public_inputs()
    proofs = []
    for kw in ({}, {"fft_helper": h, "commit_helper": c2}):
        pv = Prover(w, log_n, **kw)
        pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
        for it in range(2):                                   # the second proof reuses the named work buffers
            pr = pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, pv.fiat_shamir(pub))
        vk = pv.verifying_key()
        proofs.append(pr)
        assert pv._key_ffts is None
        pv.close()
    for pr in proofs[1:]:
        for k_ in ("wires_poly_comms", "split_quot_poly_comms"):
            assert all(np.array_equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(pr[k_], proofs[0][k_])), k_
        for k_ in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
            assert np.array_equal(pr[k_][0], proofs[0][k_][0]) and pr[k_][1] == proofs[0][k_][1], k_
        for k_ in ("wires_evals", "wire_sigma_evals"):
            assert np.array_equal(np.stack(pr[k_]), np.stack(proofs[0][k_])), k_
        assert np.array_equal(pr["perm_next_eval"], proofs[0]["perm_next_eval"])
    V.verify(B.CURVES[curve], vk, pub, proofs[1], TAU, transcript=PlonkTranscript(curve))
    # an unsatisfied witness: the degree check raises after the helper's thread has been joined
    bad = w.alloc(inst.n * 32)
    bad.upload(O.rand_fr(cid, 99, inst.n))
    pv = Prover(w, log_n, fft_helper=h)
    pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
    before = threading.active_count()
    try:
        pv.prove_dev([bad.ptr] + list(inst.wev[1:]), inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, pv.fiat_shamir(pub))
        raise SystemExit("an unsatisfied witness was proved")
    except WrongQuotientPolyDegree:
        pass
    assert pv._key_ffts is None and threading.active_count() == before
    pv.close(); bad.free(); inst.close()
    for x in (w, c2, h):
        x.close()
print("helper ok")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=emu_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "helper ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
