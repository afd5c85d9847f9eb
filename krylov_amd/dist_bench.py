"""bench.py for N > 1: the same mesh as the single-GPU run, split into row
blocks, one rank per GPU (strong scaling), halo rows and dot products over
RCCL (torch.distributed backend "nccl")."""
import json
import os
import sys
import time


def run(args, world, rank, local_rank, out_stream=None):
    import torch
    import torch.distributed as dist
    from krylov_amd import capi, host, problems
    from krylov_amd.dist import RcclComm, TorchComm, partition_rows

    backend = os.environ.get("KRY_DIST_BACKEND", "nccl")
    device = local_rank
    if backend != "nccl":
        # rehearsal on fewer GPUs than ranks: host-staged collectives let ranks share a device
        device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    dist.init_process_group(backend)

    # Collectives: the library's own RCCL calls when every rank has its own GPU, the
    # torch.distributed callbacks otherwise (KRY_DIST_COMM=torch|rccl overrides).  Either
    # choice is first validated on a small mesh against a single-rank replica run by every
    # rank; if the library-side RCCL path fails that check the callbacks take over.
    def make(kind):
        c = host.Context(device)
        return c, (RcclComm(c) if kind == "rccl" else TorchComm(c))

    def selfcheck(c):
        capi.check(capi.lib().kry_ctx_comm_selftest(c.h))
        # rows enough for three bands of rows per rank, so that the one-sweep steps take the
        # path on which the halo rows travel while the interior bands are swept
        rows = max(96 * world, 96)
        prob, mm = problems.s_iso(rows, 64), 8
        part = host.ImplicitEuler(c, **prob, m=mm, row_range=partition_rows(rows + 2, world)[rank])
        one = host.ImplicitEuler(host.Context(device), **prob, m=mm)
        err = 0.0
        for _ in range(2):
            a, b = part.timestep(0.25), one.timestep(0.25)
            err = max(err, abs(a - b) / abs(b))
        return err

    # candidates, tried in order until all ranks pass the check: the library's own RCCL calls
    # with and without the overlapped halo exchange, then the torch.distributed callbacks
    want = os.environ.get("KRY_DIST_COMM", "rccl" if backend == "nccl" else "torch")
    overlap_pref = os.environ.get("KRY_HALO_OVERLAP", "1")
    candidates = [(want, overlap_pref)]
    if overlap_pref != "0":
        candidates.append((want, "0"))
    if want != "torch":
        candidates.append(("torch", "0"))
    comm_kind, check_err, ctx, comm, overlap = None, None, None, None, None
    for kind, ov in candidates:
        os.environ["KRY_HALO_OVERLAP"] = ov  # read when a solver is created
        try:
            if ctx is None or kind != getattr(ctx, "_kind", None):
                ctx, comm = make(kind)
                ctx._kind = kind
            check_err = selfcheck(ctx)
            ok = check_err <= 1e-9
        except Exception as e:  # noqa: BLE001 - any failure means: try the next candidate
            if rank == 0:
                print("krylov_amd.dist_bench: %s collectives (halo overlap %s) unusable: %r" % (kind, ov, e),
                      file=sys.stderr, flush=True)
            ok = False
            ctx = None
        flag = torch.tensor([1 if ok else 0], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            comm_kind, overlap = kind, ov != "0"
            break
        ctx = None
    if comm_kind is None:
        raise RuntimeError("no working collectives (self-check error %r)" % (check_err,))
    n_side, m, dt = args.mesh, args.m, args.dt
    ranges = partition_rows(n_side + 2, world)
    # hipGraph replay of a step that contains collectives needs the library-side RCCL
    # transport; it is opt-in on several ranks (KRY_DIST_GRAPH=1): eager launches are the
    # configuration the start-up check above has validated
    graph = bool(args.graph) and comm_kind == "rccl" and os.environ.get("KRY_DIST_GRAPH") == "1"
    dev = host.ImplicitEuler(ctx, **problems.s_iso(n_side), m=m, bc_time_dependent=False,
                             row_range=ranges[rank], use_graph=graph)
    N = dev.n
    nloc = dev.nloc
    for _ in range(args.warmup):
        dev.timestep(dt)
    if comm_kind == "torch":
        comm.calls = {"allreduce": 0, "sendrecv": 0}  # count the timed steps only
    # no instrumentation inside the timed region; the per-kernel means come from a separate pass
    ctx.set_kernel_timing(False)
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    last = 0.0
    for _ in range(args.steps):
        last = dev.timestep(dt)
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    ms, cnt = [0.0] * capi.K_COUNT, [0] * capi.K_COUNT
    profile_steps = 0 if args.no_kernel_timing else args.profile_steps
    if profile_steps > 0:  # every rank takes part: the steps are collective
        dev.use_graph(False)
        ctx.set_kernel_timing(True)
        ctx.kernel_times()
        for _ in range(profile_steps):
            dev.timestep(dt)
        ms, cnt = ctx.kernel_times()
        ctx.set_kernel_timing(False)
        dist.barrier()
    # slowest rank defines the job time
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    if rank == 0:
        bytes_cycle, launches_cycle = dev.solver().cycle_bytes(dev.operator())
        kernels = {}
        for k, name in enumerate(capi.K_NAMES):
            if cnt[k] == 0:
                continue
            per_launch = bytes_cycle[k] / launches_cycle[k] if launches_cycle[k] else None
            avg_ms = ms[k] / cnt[k]
            kernels[name] = {"launches": cnt[k], "total_ms": round(ms[k], 3), "avg_ms": round(avg_ms, 4),
                             "alg_bytes_per_launch": per_launch,
                             "gbps": round(per_launch / (avg_ms * 1e-3) / 1e9, 1) if per_launch else None}
        roofline = None
        timed = {n: k for n, k in kernels.items() if n in ("spmv_dots", "mgs_block_full", "mgs_block_part", "mgs_last", "arnoldi_fused")
                 and k["gbps"]}
        if timed:
            # the kernel family with the largest share of rank 0's kernel time
            name = max(timed, key=lambda n: timed[n]["total_ms"])
            k = timed[name]
            achieved = k["alg_bytes_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "%s kernels on rank 0's row block" % name, "kernel_class": name,
                        "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(achieved / 8000.0, 4), "traffic": None,
                        "alg_bytes_per_launch": k["alg_bytes_per_launch"], "avg_launch_ms": k["avg_ms"],
                        "launches_timed": k["launches"]}
        kernel_ms = sum(ms) * (args.steps / profile_steps) if profile_steps > 0 else None
        out = {
            "metric": "gmres_arnoldi_iters_per_sec",
            "value": m * args.steps / elapsed,
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "S-iso %dx%d mesh, implicit-Euler step with fixed-m GMRES(%d), "
                                   "modified Gram-Schmidt, 5-band stencil operator" % (n_side, n_side, m),
                       "mesh": [n_side, n_side], "unknowns": N, "m": m, "dt": dt,
                       "parallelism": "row blocks over %d GPUs (halo rows + all-reduced dots, %s)" % (world, backend),
                       "rows_per_rank": [b - a for a, b in ranges]},
            # algorithmic bytes of rank 0's row block, scaled to the whole job (blocks differ by at most one row)
            "hbm_gbps_whole_job": round(world * sum(bytes_cycle) * args.steps / elapsed / 1e9, 1),
            "hbm_frac_whole_job": round(sum(bytes_cycle) * args.steps / elapsed / 1e9 / 8000.0, 4),
            "last_max_delta": last,
            # kernel time of the separate instrumented pass against the wall time of the timed region
            "rank0_kernel_ms_per_step": None if kernel_ms is None else round(kernel_ms / args.steps, 2),
            "rank0_comm_and_gap_ms_per_step": None if kernel_ms is None else
            round(elapsed / args.steps * 1e3 - kernel_ms / args.steps, 2),
            "graph": bool(graph),
            "collectives": comm_kind, "collectives_selfcheck_rel_err": check_err,
            "halo_exchange_overlapped_with_interior_bands": bool(overlap),
            "collectives_per_step": {k: (v // args.steps if v is not None else None) for k, v in comm.calls.items()},
            "kernels": kernels,
            "roofline": roofline,
        }
        print(json.dumps(out), file=out_stream or sys.stdout, flush=True)
    dist.barrier()
    dist.destroy_process_group()
