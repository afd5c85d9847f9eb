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
    (3, "s_iso", 20, 300, 20, 0),  # rows wider than one strip of the one-sweep kernels
    (3, "s_iso", 1, 40, 5, 0),     # one grid row per rank
    (2, "s_iso", 400, 40, 12, 0),  # six bands of rows per rank: interior bands swept while the halo rows travel
    (3, "s_iso", 330, 270, 24, 0),  # three bands, two strips, the 512-thread kernel from step 18 on
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
    assert int(r["rc"]) == 0 and int(r["rc2"]) == 0 and int(r["rc3"]) == 0 and int(r["rc4"]) == 0
    assert bool(r["same"]) and bool(r["p2p_ok"])
    print("python-driven all-reduce of 10 doubles: %.1f us per call; grouped send+recv of 20 doubles: %.1f us"
          % (float(r["us_per_allreduce"]), float(r["us_per_sendrecv"])))


def test_library_side_rccl_one_rank(tmp_path):
    """kry_ctx_set_comm_rccl in a group of one: id hand-over, communicator,
    ncclAllReduce and grouped ncclSend/ncclRecv on the library's stream
    (kry_ctx_comm_selftest), and a slab run that must equal the plain run."""
    out = tmp_path / "rccl.npz"
    launch(1, ["rccl_selftest", out], timeout=600)
    r = np.load(out)
    assert int(r["native"]) == 1
    assert np.array_equal(r["plain"], r["slab"]) and np.array_equal(r["plain"], r["graph"])
    assert bool(r["same_T"])


def test_bench_contract_two_ranks_shared_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one
    process per rank), rehearsed with both ranks on this GPU and host-staged
    collectives: one JSON line from rank 0 with the contract's fields."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KRY_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--mesh", "256", "--gmres-m", "10"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout[-2000:]  # nothing but the JSON line on stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["collectives"] == "torch" and d["collectives_selfcheck_rel_err"] <= 1e-9
    # host-staged collectives cannot be captured: the step form is eager, nothing was turned down, first attempt
    assert d["graph"] is False and d["config"]["graph"] is False
    assert d["rejected_candidates"] == [] and d["attempt"] == 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    # the row blocks run the one-sweep Arnoldi kernels (8 of the 10 steps per cycle; the last is the closing sweep)
    assert d["kernels"]["arnoldi_fused"]["launches"] == 2 * 7
    assert abs(d["value"] - 10 * 2 / (d["ms_per_step"] * 2e-3)) <= 1e-6 * d["value"]


def _run_bench(extra_env, nproc, port, args=()):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--steps", "2",
           "--warmup", "1", "--mesh", "256", "--gmres-m", "10", *args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


def test_bench_graph_replayed_steps_with_library_side_rccl_one_rank():
    """The multi-rank path of bench.py in a group of ONE rank (all this build can run on real RCCL): the start-up
    check puts the graph-replayed step with the library's own RCCL calls first, validates it against the
    single-rank replica, and the timed region replays the captured step."""
    d, _ = _run_bench({"KRY_BENCH_FORCE_DIST": "1"}, 1, 29617, ["--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["collectives"] == "rccl"
    assert d["graph"] is True and d["config"]["graph"] is True
    assert d["rejected_candidates"] == [] and d["attempt"] == 0
    assert d["collectives_selfcheck_rel_err"] <= 1e-9


def test_bench_watchdog_and_fresh_processes_after_a_stalled_start(tmp_path):
    """A rank that makes no progress in the start-up check (simulated in the first attempt) is ended by its watchdog
    with exit code 3; every rank's supervisor then starts a fresh process with the next, more conservative
    configuration, and the job still delivers its line (attempt 1)."""
    d, err = _run_bench({"KRY_DIST_BACKEND": "gloo", "KRY_DIST_TEST_HANG_ATTEMPTS": "0", "KRY_DIST_STARTUP_LIMIT": "6",
                         "TMPDIR": str(tmp_path)}, 2, 29619, ["--no-cpu-baseline"])
    assert d["attempt"] == 1 and d["n_gpus"] == 2
    assert "made no progress" in err and "exit code 3" in err


@pytest.mark.parametrize("nranks,cx,cy,m", [(2, 400, 40, 12), (3, 330, 270, 24)])
def test_halo_overlap_changes_no_bit(tmp_path, monkeypatch, nranks, cx, cy, m):
    """On row blocks the one-sweep steps sweep the interior bands of rows while the
    neighbours' rows of the new vector travel on a second stream, then the first
    and the last band: two launches that fill one array of partial sums.  Every
    result bit must equal the run with one launch after the exchange
    (KRY_HALO_OVERLAP=0)."""
    runs = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("KRY_HALO_OVERLAP", overlap)
        out = tmp_path / ("res%s.npz" % overlap)
        launch(nranks, ["hip", out, "s_iso", cx, cy, m, 0, 3], timeout=600)
        runs.append(np.load(out))
    a, b = runs
    assert np.array_equal(a["deltas"], b["deltas"])
    assert np.array_equal(a["T"], b["T"], equal_nan=True) and np.array_equal(a["D"], b["D"], equal_nan=True)
