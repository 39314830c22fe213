import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The suite's problems are tiny (a few trajectories, a few time steps).  Since round 6 grape_create gives such problems the
# kernels that are fastest for FEW batches -- a workgroup per derivative batch instead of the one-wave assembly kernels -- which
# would leave the assembly kernels, the subject of most GPU tests, unexercised.  GRAPE_DERIV3=1 pins the one-wave route as the
# default of the test session; tests/test_gpu_scan.py::test_few_batches_take_the_workgroup_per_batch_kernel removes it.
os.environ.setdefault("GRAPE_DERIV3", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ref():
    """The C oracle (oracle/grape_ref.c) -- test infrastructure only."""
    import grape_ref
    grape_ref.build()
    return grape_ref


def load_mpmath_pin(name):
    """tests/golden/mpmath_pin_<name>.json (tests/golden/make_mpmath_pin.py): inputs as exact doubles, and per built-in
    functional the 60-digit values of J, tau, G, Psi(T), tau_grads rounded to the nearest double."""
    import json

    import numpy as np
    z = json.load(open(os.path.join(ROOT, "tests", "golden", f"mpmath_pin_{name}.json")))
    c = lambda key: np.array(z[key + "_re"]) + 1j * np.array(z[key + "_im"])   # noqa: E731
    pr = dict(H0=c("H0"), Hc=c("Hc"), psi0=c("psi0"), target=c("target"), tlist=np.array(z["tlist"]),
              pulsevals=np.array(z["pulsevals"]), weights=np.array(z["weights"]))
    cf = lambda a: np.array([[float(x) for x in row] for row in np.reshape(a, (-1, 2))]).view(np.complex128).reshape(np.shape(a)[:-1])   # noqa: E731
    want = {}
    for f, d in z["functionals"].items():
        want[int(f)] = dict(J=float(d["J"]), tau=cf(d["tau"]), G=np.array([float(x) for x in d["G"]]),
                            psiT=cf(d["psiT"]), tau_grads=cf(d["tau_grads"]))
    return pr, want


def load_reference_outputs(npz_path):
    """tests/golden/ref_<name>.json written by julia/make_reference_fixtures.jl -- outputs of the REFERENCE itself for the
    inputs of tests/golden/<name>.npz -- or None while nobody with a Julia installation has produced them.  Returns
    {method: dict(J, G, tau, psiT, tau_grads[k][l][n])}."""
    import json

    import numpy as np
    path = os.path.join(os.path.dirname(npz_path), "ref_" + os.path.splitext(os.path.basename(npz_path))[0] + ".json")
    if not os.path.exists(path):
        return None
    z = json.load(open(path))
    out = {}
    for method, d in z.items():
        tg = np.array(d["tau_grads_re"]) + 1j * np.array(d["tau_grads_im"])          # [k][n][l] as Julia's matrices print row-wise
        out[method] = dict(J=float(d["J"]), G=np.array(d["G"]), tau=np.array(d["tau_re"]) + 1j * np.array(d["tau_im"]),
                           psiT=np.array(d["psiT_re"]) + 1j * np.array(d["psiT_im"]), tau_grads=np.transpose(tg, (0, 2, 1)),
                           pulsevals=np.array(d["pulsevals_as_discretized"]), source=d.get("source", "?"))
    return out
