"""The HIP row-block path with 2 and 3 ranks sharing cuda:0 (collectives staged
through gloo), against the single-rank HIP result and the CPU restatement."""
import numpy as np
import pytest

from krylov_amd import host, problems
from test_dist_gloo import launch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("nranks,name,cx,cy,m,tdp", [
    (2, "prob1", 38, 18, 10, 0),
    (3, "prob1", 38, 18, 10, 1),
    (2, "time_dependent", 12, 10, 8, 0),
    (3, "s_iso", 33, 20, 12, 0),
    (2, "prob1", 2, 5, 3, 0),
])
def test_slabs_match_single_rank(tmp_path, ctx, oracle, nranks, name, cx, cy, m, tdp):
    steps = 3
    out = tmp_path / "res.npz"
    launch(nranks, ["hip", out, name, cx, cy, m, tdp, steps], timeout=600)
    r = np.load(out)
    assert int(r["nranks"]) == nranks and int(r["allreduces"]) > 0 and int(r["sendrecvs"]) > 0
    prob = getattr(problems, name)(cx, cy)
    one = host.ImplicitEuler(ctx, **prob, m=m, tdp=bool(tdp))
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=bool(tdp))
    d1 = [one.timestep(0.25) for _ in range(steps)]
    dr = [ref.timestep(0.25) for _ in range(steps)]
    for a, b, c in zip(r["deltas"], d1, dr):
        assert abs(a - b) <= 1e-9 * abs(b) and abs(a - c) <= 1e-8 * abs(c)
    # ghost fill (incl. the wrap-around column) and the velocity fields: exact
    assert np.array_equal(r["U"], one.field(1), equal_nan=True)
    Tm = r["T"][1:-1, 1:-1]
    assert rel(Tm, one.field(0)[1:-1, 1:-1]) <= 1e-9
    assert rel(Tm, ref.field(0)[1:-1, 1:-1]) <= 1e-8
    # the last right-hand side is assembled from exchanged halo rows
    assert rel(one.compact(r["B"]), one.sol_vec()) <= 1e-8


@pytest.mark.parametrize("nranks,name,cx,cy,m,tdp", [(2, "prob1", 38, 18, 10, 0), (3, "prob1", 38, 18, 10, 1)])
def test_slabs_extended_solver(tmp_path, ctx, oracle, nranks, name, cx, cy, m, tdp):
    """Restarted solver on row blocks: every rank takes the same stop decision
    from the all-reduced Hessenberg entries; counts within +-1 of the CPU statement."""
    steps = 3
    out = tmp_path / "res.npz"
    launch(nranks, ["hip", out, name, cx, cy, m, tdp, steps, 1e-12], timeout=600)
    r = np.load(out)
    prob = getattr(problems, name)(cx, cy)
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=bool(tdp), extended=(1e-12, 60))
    for k in range(steps):
        d = ref.timestep(0.25)
        assert abs(int(r["iters"][k]) - ref.solver().last_stats()[0]) <= 1
        assert abs(float(r["deltas"][k]) - d) <= 1e-10 * abs(d)
    assert rel(r["T"][1:-1, 1:-1], ref.field(0)[1:-1, 1:-1]) <= 1e-10


def test_device_native_callbacks_one_rank(tmp_path):
    """backend nccl with a single rank: the callbacks accept the library's
    device pointers on its stream and leave the data intact."""
    out = tmp_path / "nccl.npz"
    launch(1, ["nccl_selftest", out], timeout=600)
    r = np.load(out)
    assert int(r["rc"]) == 0 and int(r["rc2"]) == 0 and int(r["rc3"]) == 0
    assert bool(r["same"])
    print("python-driven all-reduce of 10 doubles: %.1f us per call" % float(r["us_per_allreduce"]))
