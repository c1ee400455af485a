"""bench.py as the driver runs it: one JSON line on stdout with the contract's fields — on one GPU, and as
`python -m torch.distributed.run --nproc-per-node 2 … bench.py --gpus 2` (both ranks on the one GPU of the test box, gloo with host
staging instead of RCCL — MOLLYHIP_DIST_BACKEND / MOLLYHIP_FORCE_DEVICE — so everything but the RCCL transport is the production path)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _record(cmd, env=None, timeout=600):
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # stdout carries exactly one line
    return json.loads(lines[0])


def _check_contract(d, n_gpus, steps, warmup):
    assert d["metric"] == "ns_per_day" and d["unit"] == "ns/day" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "strong"
    assert d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "workload" in d["config"] and "timed_window" in d["config"]


def test_bench_single_gpu_record():
    d = _record([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--workload", "lj256k", "--equil", "300", "--no-cpu-baseline", "--traffic", "file"])
    _check_contract(d, 1, 20, 5)
    assert d["config"]["parallelism"] == "single domain"
    assert d["config"]["timed_window"].startswith("as scheduled: mean of 5 consecutive windows of 20 steps")   # never placed
    w = d["window_ms_per_step"]
    assert w["n"] == 5 and w["min"] <= d["ms_per_step"] <= w["max"] and abs(w["mean"] - d["ms_per_step"]) < 1e-12
    assert "secondary" not in d                                        # only the default workload appends the other configurations


def test_bench_default_workload_carries_the_other_configurations():
    """`python bench.py --gpus 1` as the driver runs it (shortened: no equilibration, no CPU legs): the 1M-atom record plus complete
    records of 6mrr with PME and of the 256k-atom fluid under `secondary`."""
    d = _record([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--equil", "100", "--no-cpu-baseline", "--profile-steps", "50"])
    _check_contract(d, 1, 20, 5)
    assert d["config"]["name"] == "lj1m" and [r["config"]["name"] for r in d["secondary"]] == ["6mrr_pme", "lj256k"]
    for r in d["secondary"]:
        assert r["metric"] == "ns_per_day" and r["value"] > 0 and r["roofline"]["achieved"] > 0 and r["roofline"]["step_frac"] > 0
        assert r["n_gpus"] == 1 and r["dtype"] == "f32" and r["config"]["timed_window"] == "as scheduled"
    assert d["secondary"][0]["roofline"]["stage_ms_per_step"]["pme_reciprocal"] > 0
    # the compact summary is the LAST key of the line and repeats every configuration's headline (a reader of the tail sees all three)
    assert list(d.keys())[-1] == "summary"
    for nm, r in (("lj1m", d), ("6mrr_pme", d["secondary"][0]), ("lj256k", d["secondary"][1])):
        assert abs(d["summary"][nm + "_ms_per_step"] - r["ms_per_step"]) < 1e-5 and abs(d["summary"][nm + "_ns_day"] - r["value"]) < 0.06
    assert len(json.dumps(d["summary"])) < 1200
    # roofline.traffic of the main workload is MEASURED in the run (two rocprofv3 PMC passes of the same command in child processes), with this library's ids;
    # the secondaries read the committed counter files and say whether those were taken with this build
    r = d["roofline"]
    assert r["traffic_source"].startswith("measured in this run") and r["traffic_is_of_this_build"] is True and 0.9 < r["traffic_over_algorithmic"] < 1.3
    assert r["list_upkeep"]["prune"]["traffic_source"].startswith("measured in this run") and r["list_upkeep"]["prune"]["traffic"] > 5e8
    assert d["secondary"][1]["roofline"]["traffic_source"].startswith("profiles/") and d["lib_build_id"] and d["kernel_src_id"]


def test_bench_two_ranks_record():
    port = _free_port()
    d = _record([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "lj256k", "--equil", "300"],
                env={"MOLLYHIP_DIST_BACKEND": "gloo", "MOLLYHIP_FORCE_DEVICE": "0"})
    _check_contract(d, 2, 20, 5)
    assert "spatial bricks 2x1x1" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d                                     # rank 0 at N = 1 only


def test_bench_two_ranks_falls_back_collectively_when_a_form_of_the_step_loop_fails():
    """The driver's scaling run is the first time two ranks meet over xGMI.  A form of the step loop that fails in the untimed part on ANY rank — injected here on the
    last rank, for the fused form and for the in-engine loop with separate launches — makes EVERY rank drop its context and start the next form from the initial state
    (domain.py bench_distributed); the record names the form that ran."""
    port = _free_port()
    d = _record([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "lj256k", "--equil", "100", "--fail-forms", "fused,separate launches"],
                env={"MOLLYHIP_DIST_BACKEND": "gloo", "MOLLYHIP_FORCE_DEVICE": "0"}, timeout=240)
    _check_contract(d, 2, 20, 5)
    assert "host loop" in d["config"]["parallelism"]
    port = _free_port()
    d = _record([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                 "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "lj256k", "--equil", "100", "--fail-forms", "fused"],
                env={"MOLLYHIP_DIST_BACKEND": "gloo", "MOLLYHIP_FORCE_DEVICE": "0"}, timeout=240)
    assert "separate launches per step" in d["config"]["parallelism"]


def test_bench_memlimit_cli():
    """`bench.py --workload memlimit` (the reference's GPU memory-limit recipe, docs/src/examples.md:969-1017) at a single size: the record's contract, the pair count
    on the closed form, Newton's third law over the box, the cube of 10^5 atoms against the oracle.  The full-size result is profiles/r06_memlimit_*.json."""
    d = _record([sys.executable, "bench.py", "--workload", "memlimit", "--memlimit-start", "400000", "--memlimit-max", "500000"], timeout=400)
    assert d["metric"] == "max_atoms_100_steps" and d["unit"] == "atoms" and d["value"] == 400000 and d["n_gpus"] == 1 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["name"] == "memlimit" and "docs/src/examples.md:969-1017" in d["config"]["workload"]
    t = d["trials"][0]
    assert t["ok"] and abs(t["pairs_deviation_sigma"]) < 5 and t["net_force_over_abs_force"] < 1e-6 and t["ms_per_step"] > 0 and t["hbm_in_use_gb"] > 0
    oc = t["oracle_check"]
    assert oc["passed"] and oc["inner_atoms"] > 90000 and oc["worst_err_over_tol"] <= 1.0
    assert d["smallest_that_failed"] is None and d["reference_published"]["NVIDIA RTX A6000 (48 GB)"] == 140000
