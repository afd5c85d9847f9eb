"""Worker run by the multi-rank tests, one process per rank:

    python -m torch.distributed.run --nproc-per-node N ... tests/dist_worker.py MODE OUT [ARGS]

MODE numpy : CPU only.  A numpy statement of the row-block algorithm (ghost
             fill with the wrap-around column, operator apply with exchanged
             halo rows, all-reduced Gram-Schmidt) on top of the package's
             host collectives; checks the decomposition logic under gloo.
MODE hip   : the HIP path, every rank on cuda:0, collectives staged through
             gloo (several ranks may share one GPU this way).  A trailing RTOL
             argument > 0 selects the restarted, residual-monitored solver.
Rank 0 writes an .npz with the gathered results to OUT.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def numpy_mode(out, problem_name, cx, cy, m):
    import torch.distributed as dist
    from krylov_amd import capi, problems
    from krylov_amd.dist import HostCollectives, gather_rows, partition_rows
    from oracle import oracle as O

    host = HostCollectives()
    rank, nranks = host.rank, host.nranks
    prob = getattr(problems, problem_name)(cx, cy)
    # every rank builds the whole assembled system with the CPU restatement and
    # keeps only its rows: the test is about the exchange logic, not assembly
    ref = O.ImplicitEuler(**prob, m=m)
    ref.assemble_system(0.25)
    rowptr, col, val = ref.csr()
    P = cy + 2
    ranges = partition_rows(cx + 2, nranks)
    r0, r1 = ranges[rank]

    def expand(c):
        p = np.zeros((cx + 2, P))
        p[0, 1:-1] = c[:cy]
        p[1:-1, :] = c[cy:cy + cx * P].reshape(cx, P)
        p[-1, 1:-1] = c[cy + cx * P:]
        return p

    def compact(p):
        return np.concatenate([p[0, 1:-1], p[1:-1, :].reshape(-1), p[-1, 1:-1]])

    # five bands in padded layout from the CSR matrix
    bands = np.zeros((5, cx + 2, P))
    cell_of = -np.ones((cx + 2, P), dtype=np.int64)
    cell_of[0, 1:-1] = np.arange(cy)
    cell_of[1:-1, :] = cy + np.arange(cx * P).reshape(cx, P)
    cell_of[-1, 1:-1] = cy + cx * P + np.arange(cy)
    pos = {int(c): (i, j) for (i, j), c in np.ndenumerate(cell_of) if c >= 0}
    offs = {(-1, 0): 0, (0, -1): 1, (0, 0): 2, (0, 1): 3, (1, 0): 4}
    for r in range(ref.n):
        i, j = pos[r]
        for p_ in range(rowptr[r], rowptr[r + 1]):
            ci, cj = pos[int(col[p_])]
            bands[offs[(ci - i, cj - j)], i, j] = val[p_]
    bands[2, [0, 0, -1, -1], [0, -1, 0, -1]] = 1.0
    lb = bands[:, r0:r1, :]

    def apply_local(xl):
        lo, hi = host.halo_exchange(xl, P)
        xp = np.vstack([lo[None, :], xl, hi[None, :]])
        y = np.zeros_like(xl)
        c = xp[1:-1]
        west, east = xp[:-2], xp[2:]
        south = np.hstack([np.zeros((c.shape[0], 1)), c[:, :-1]])
        north = np.hstack([c[:, 1:], np.zeros((c.shape[0], 1))])
        # rows are contiguous in memory: the j-1 neighbour of column 0 is the
        # previous row's last entry; its coefficient is exactly 0 there
        y = ((((0.0 + lb[0] * west) + lb[1] * south) + lb[2] * c) + lb[3] * north) + lb[4] * east
        return y

    def gdot(a, b):
        s = np.array([float(np.sum(a * b))])
        host.allreduce(s, capi.REDUCE_SUM)
        return s[0]

    b_full = expand(ref.sol_vec())
    bl = b_full[r0:r1].copy()
    # distributed operator apply == the sequential row sum, bit for bit
    rng = np.random.default_rng(5)
    x_full = expand(rng.uniform(-1, 1, ref.n))
    y_loc = apply_local(x_full[r0:r1].copy())
    y_full = gather_rows(y_loc, ranges, host)
    y_ref = O.csr_matvec(rowptr, col, val, compact(x_full))
    spmv_equal = bool(np.array_equal(compact(y_full), y_ref))
    # fixed-m GMRES with modified Gram-Schmidt on slabs
    beta = np.sqrt(gdot(bl, bl))
    V = [bl / beta]
    H = np.zeros((m + 1, m))
    for j in range(m):
        w = apply_local(V[j])
        for i in range(j + 1):
            H[i, j] = gdot(w, V[i])
            w = w - H[i, j] * V[i]
        H[j + 1, j] = np.sqrt(gdot(w, w))
        V.append(w / H[j + 1, j])
    S = H.T @ H
    yv = np.linalg.solve(S, beta * H[0, :])
    xl = sum(yv[i] * V[i] for i in range(m))
    x_full = gather_rows(xl, ranges, host)
    x_ref = ref.solver().solve(ref.op(), ref.sol_vec())
    if rank == 0:
        np.savez(out, spmv_equal=spmv_equal, x=compact(x_full), x_ref=x_ref, nranks=nranks)
    dist.barrier()


def hip_mode(out, problem_name, cx, cy, m, tdp, steps, rtol=0.0):
    import torch
    import torch.distributed as dist
    from krylov_amd import capi, host, problems
    from krylov_amd.dist import HostCollectives, TorchComm, gather_rows, partition_rows

    torch.cuda.set_device(0)
    ctx = host.Context(0)
    comm = TorchComm(ctx)
    hostc = HostCollectives()
    prob = getattr(problems, problem_name)(cx, cy)
    ranges = partition_rows(cx + 2, comm.nranks)
    dev = host.ImplicitEuler(ctx, **prob, m=m, tdp=bool(tdp), row_range=ranges[comm.rank],
                             extended=(rtol, 60) if rtol > 0.0 else None)
    deltas, iters = [], []
    for _ in range(steps):
        deltas.append(dev.timestep(0.25))
        if rtol > 0.0:
            iters.append(dev.solver().last_stats()[0])
    T = gather_rows(dev.local(capi.LOCAL_T), ranges, hostc)
    U = gather_rows(dev.local(capi.LOCAL_U), ranges, hostc)
    D = gather_rows(dev.local(capi.LOCAL_DELTA), ranges, hostc)
    B = gather_rows(dev.local(capi.LOCAL_RHS), ranges, hostc)
    if comm.rank == 0:
        np.savez(out, T=T, U=U, D=D, B=B, deltas=np.array(deltas), iters=np.array(iters), nranks=comm.nranks,
                 allreduces=comm.calls["allreduce"], sendrecvs=comm.calls["sendrecv"])
    dist.barrier()


def nccl_selftest(out):
    """One rank, backend nccl (= RCCL): drives the device-native branch of the
    callbacks (device-pointer wrapping, external stream) directly."""
    import ctypes as C
    import time
    import torch
    import torch.distributed as dist
    from krylov_amd import capi, host, problems
    from krylov_amd.dist import TorchComm

    torch.cuda.set_device(0)
    ctx = host.Context(0)
    comm = TorchComm(ctx)
    assert comm.device_native
    dev = host.ImplicitEuler(ctx, **problems.prob1(), m=4)
    dev.assemble_system(0.25)
    before = dev.sol_vec()
    ptr = dev.rhs_device()
    rc = comm._allreduce(None, ptr, dev.nloc, capi.REDUCE_SUM, None)
    rc2 = comm._allreduce(None, ptr, 1, capi.REDUCE_MAX, None)
    rc3 = comm._sendrecv(None, ptr, -1, ptr, -1, 20, None)
    # a grouped send + receive through RCCL with this rank as its own peer, on raw
    # device pointers as the library hands them over
    scratch = torch.arange(64, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    sp = scratch.data_ptr()
    rc4 = comm._sendrecv(None, sp, 0, sp + 40 * 8, 0, 20, None)
    ctx.synchronize()
    got = scratch.cpu().numpy()
    p2p_ok = bool(np.array_equal(got[40:60], np.arange(20.0)) and np.array_equal(got[:40], np.arange(40.0)))
    # time the grouped exchange as the halo exchange issues it
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        comm._sendrecv(None, sp, 0, sp + 40 * 8, 0, 20, None)
    ctx.synchronize()
    us_p2p = (time.perf_counter() - t0) / 100 * 1e6
    t0 = time.perf_counter()
    for _ in range(200):
        comm._allreduce(None, ptr, 10, capi.REDUCE_SUM, None)
    ctx.synchronize()
    us = (time.perf_counter() - t0) / 200 * 1e6
    after = dev.sol_vec()
    np.savez(out, rc=rc, rc2=rc2, rc3=rc3, rc4=rc4, p2p_ok=p2p_ok, same=bool(np.array_equal(before, after)),
             us_per_allreduce=us, us_per_sendrecv=us_p2p)
    dist.barrier()


def rccl_selftest(out):
    """One rank, the library's own RCCL calls: communicator set-up from a
    broadcast id, the collective self-test (all-reduce and a send/receive with
    this rank as its own peer, through RCCL even in a group of one), and a
    slab run - eager and replayed from a hipGraph - against the plain one."""
    import torch
    import torch.distributed as dist
    from krylov_amd import capi, host, problems
    from krylov_amd.dist import RcclComm

    torch.cuda.set_device(0)
    ctx = host.Context(0)
    comm = RcclComm(ctx)
    comm.selftest()
    native = capi.lib().kry_ctx_comm_is_native(ctx.h)
    prob = problems.prob1()
    plain = host.ImplicitEuler(host.Context(0), **prob, m=10)
    slab = host.ImplicitEuler(ctx, **prob, m=10, row_range=(0, 40))
    graph = host.ImplicitEuler(ctx, **prob, m=10, row_range=(0, 40), use_graph=True)
    d = [[s.timestep(0.25) for _ in range(5)] for s in (plain, slab, graph)]
    np.savez(out, native=native, plain=np.array(d[0]), slab=np.array(d[1]), graph=np.array(d[2]),
             same_T=bool(np.array_equal(plain.field(0), slab.field(0), equal_nan=True)
                         and np.array_equal(plain.field(0), graph.field(0), equal_nan=True)))
    dist.barrier()


def main():
    import torch.distributed as dist
    mode, out = sys.argv[1], sys.argv[2]
    dist.init_process_group("nccl" if mode in ("nccl_selftest", "rccl_selftest") else "gloo")
    try:
        if mode == "nccl_selftest":
            nccl_selftest(out)
        elif mode == "rccl_selftest":
            rccl_selftest(out)
        elif mode == "numpy":
            numpy_mode(out, sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))
        else:
            hip_mode(out, sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]),
                     int(sys.argv[8]), float(sys.argv[9]) if len(sys.argv) > 9 else 0.0)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
