import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The multi-process tests (several ranks sharing the one GPU, spawned children, gloo rendezvous) run LAST: a spawn hiccup
    under `-x` must not hide the single-process parity tests behind it."""
    is_late = lambda it: "test_gpu_domain" in it.nodeid or "test_gpu_bench_cli" in it.nodeid
    late = [it for it in items if is_late(it)]
    if late:
        rest = [it for it in items if not is_late(it)]
        items[:] = rest + late


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory `molly.jl_amd/`, importable as `molly_jl_amd`)."""
    import molly_loader
    return molly_loader.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def slack(request):
    """slack(name, achieved, allowed): how much of a tolerance a test used.  Asserts achieved <= allowed, prints the ratio and appends it to
    gpurun_out/tolerance_slack.jsonl (on the GPU box the directory is merged back: DESIGN.md §2 quotes these numbers, and a bar with more than
    3× slack gets tightened)."""
    import json
    rows = []

    def record(name, achieved, allowed):
        achieved, allowed = float(achieved), float(allowed)
        rows.append({"test": request.node.nodeid, "what": name, "achieved": achieved, "allowed": allowed, "ratio": achieved / allowed if allowed else float("inf")})
        print(f"[slack] {request.node.name}: {name}: {achieved:.4g} of {allowed:.4g} allowed ({achieved / allowed:.3f})")
        assert achieved <= allowed, f"{name}: {achieved:.6g} > {allowed:.6g}"
    yield record
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "tolerance_slack.jsonl"), "a") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
    except OSError:
        pass
