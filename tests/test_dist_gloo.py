"""Multi-rank path on CPU: partitioning, the host collectives, and a numpy
statement of the row-block algorithm (halo rows, all-reduced Gram-Schmidt)
under torch.distributed/gloo with 2 and 3 ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from krylov_amd.dist import partition_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(nranks, args, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")] + [str(a) for a in args]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def test_partition_rows():
    for total in (4, 7, 40, 8194, 16386):
        for n in (1, 2, 3, 4, 8):
            if total < n:
                continue
            parts = partition_rows(total, n)
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        partition_rows(3, 4)


@pytest.mark.parametrize("nranks,cx,cy", [(2, 38, 18), (3, 11, 7), (2, 2, 5)])
def test_row_block_algorithm_under_gloo(tmp_path, nranks, cx, cy):
    out = tmp_path / "res.npz"
    launch(nranks, ["numpy", out, "prob1", cx, cy, 8])
    r = np.load(out)
    assert int(r["nranks"]) == nranks
    assert bool(r["spmv_equal"])  # halo exchange feeds the exact neighbour rows
    err = np.linalg.norm(r["x"] - r["x_ref"]) / np.linalg.norm(r["x_ref"])
    assert err <= 1e-10, err
