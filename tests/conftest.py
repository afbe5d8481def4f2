import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(autouse=True)
def _fresh_splitk_slots(request):
    """GPU tests: every test starts with all 16 slots of the all-taps kernel's reduction-split workspace.  A slot is never
    handed back by the library, so late in a long session captures would run unsplit while eager launches still split --
    equal to fp32 rounding of the partial sums, but not bit for bit, which several graph-vs-eager tests assert."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            from madiff_amd import ops
            ops.reset_splitk_slots()
    yield


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


# ---- measured errors, recorded beside the bounds ("recording the measured worst would let drift
# show").  Every rel_err() / record_measured() call is filed under the running test (PYTEST_CURRENT_TEST); at the end of
# the session the worst value per test goes to gpurun_out/parity_measured_{gpu,cpu}.txt (merged back from the GPU box; the
# copies of the round's final runs are committed as profiles/rNN_parity_measured_{gpu,cpu}.txt).
_MEASURED = {}


def record_measured(value, kind="rel_err"):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
    n, worst = _MEASURED.get((test, kind), (0, 0.0))
    v = float(value)
    _MEASURED[(test, kind)] = (n + 1, max(worst, v) if v == v else float("nan"))
    return value


def pytest_sessionfinish(session, exitstatus):
    if not _MEASURED:
        return
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        which = "gpu" if any("test_gpu" in t or "no_alloc" in t for (t, _k) in _MEASURED) else "cpu"
        with open(os.path.join(out, f"parity_measured_{which}.txt"), "w") as f:
            f.write("# worst measured error per test (kind: rel_err = max(max-norm, L2) relative, l2 = relative L2); "
                    "bounds: see the tests\n")
            for (test, kind), (n, worst) in sorted(_MEASURED.items()):
                f.write(f"{worst:10.3e}  {kind:8s} n={n:<5d} {test}\n")
    except OSError:
        pass


def rel_err(a, b):
    """The parity metric of every tolerance in this suite: the LARGER of the max-norm relative error max|a-b| / max|b| and
    the L2 relative error ||a-b|| / ||b|| (round 1 used the max-norm form alone, the loosest reading of "relative error")."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mx = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    l2 = float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))
    return record_measured(max(mx, l2))
