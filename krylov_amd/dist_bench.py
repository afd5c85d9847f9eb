"""bench.py for N > 1: the same mesh as the single-GPU run, split into row
blocks, one rank per GPU (strong scaling), halo rows and dot products over
RCCL (torch.distributed backend "nccl").

Never measured on more than one physical GPU by this build (none was available
to it): the transport and the step form are therefore validated at start-up on
the node the job runs on, candidate after candidate, and a watchdog turns a
rank that makes no progress into a diagnosed exit instead of a silent stall.
"""
import json
import os
import sys
import threading
import time


class Watchdog:
    """Ends the process (exit code 3) when no progress has been reported for
    `limit` seconds, after saying on stderr which phase was stuck.  A blocked
    collective cannot be interrupted from Python, so leaving is all that can be
    done; the caller's supervisor (bench.py) then starts a fresh process with a
    more conservative configuration."""

    def __init__(self, rank):
        self.rank = rank
        self.label, self.limit, self.t = "start", 0.0, time.monotonic()
        self.lock = threading.Lock()
        self.thread = threading.Thread(target=self._watch, daemon=True)
        self.thread.start()

    def phase(self, label, limit):
        with self.lock:
            self.label, self.limit, self.t = label, float(limit), time.monotonic()

    def progress(self):
        with self.lock:
            self.t = time.monotonic()

    def stop(self):
        self.phase("done", 0.0)

    def _watch(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                label, limit, idle = self.label, self.limit, time.monotonic() - self.t
            if limit > 0.0 and idle > limit:
                print("krylov_amd.dist_bench: rank %d made no progress for %.0f s in phase '%s': leaving with exit "
                      "code 3 (a blocked collective cannot be interrupted)" % (self.rank, idle, label),
                      file=sys.stderr, flush=True)
                os._exit(3)


def sweep_kernel_label():
    """Which form of the one-sweep kernel serves which steps, as configured in this process."""
    import ctypes as C
    from krylov_amd import capi
    first, last = C.c_int(), C.c_int()
    capi.check(capi.lib().kry_get_one_sweep_split(C.byref(first), C.byref(last)))
    return ("k_arnoldi_fused<i> / k_arnoldi_split<i> (one-sweep Arnoldi step: 256 threads per workgroup up to i = %d and "
            "beyond i = %d, 512 threads whose halves split the basis vectors in between)" % (first.value - 1, last.value))


def run(args, world, rank, local_rank, out_stream=None):
    import torch
    import torch.distributed as dist
    from krylov_amd import capi, host, problems
    from krylov_amd.dist import RcclComm, TorchComm, partition_rows

    wd = Watchdog(rank)
    startup_limit = float(os.environ.get("KRY_DIST_STARTUP_LIMIT", "120"))
    step_limit = float(os.environ.get("KRY_DIST_STEP_LIMIT", "180"))
    wd.phase("process group", startup_limit)
    backend = os.environ.get("KRY_DIST_BACKEND", "nccl")
    device = local_rank
    if backend != "nccl":
        # rehearsal on fewer GPUs than ranks: host-staged collectives let ranks share a device
        device = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    init_method = os.environ.get("KRY_DIST_INIT_METHOD")  # set by bench.py's supervisor for a retry
    if init_method:
        dist.init_process_group(backend, init_method=init_method, world_size=world, rank=rank)
    else:
        dist.init_process_group(backend)
    attempt = int(os.environ.get("KRY_DIST_ATTEMPT", "0"))
    test_hang = os.environ.get("KRY_DIST_TEST_HANG_ATTEMPTS", "")  # test hook: attempts that stall in the start-up check

    # Collectives: the library's own RCCL calls when every rank has its own GPU, the
    # torch.distributed callbacks otherwise (KRY_DIST_COMM=torch|rccl overrides).  Every
    # candidate is first validated on a small mesh against a single-rank replica run by every
    # rank; a candidate that fails the check (by raising or by value) is recorded and the next
    # one takes over.
    def make(kind):
        c = host.Context(device)
        return c, (RcclComm(c) if kind == "rccl" else TorchComm(c))

    def selfcheck(c, graph):
        capi.check(capi.lib().kry_ctx_comm_selftest(c.h))
        # rows enough for three bands of rows per rank, so that the one-sweep steps take the
        # path on which the halo rows travel while the interior bands are swept
        rows = max(96 * world, 96)
        prob, mm = problems.s_iso(rows, 64), 8
        part = host.ImplicitEuler(c, **prob, m=mm, row_range=partition_rows(rows + 2, world)[rank], use_graph=graph)
        one = host.ImplicitEuler(host.Context(device), **prob, m=mm)
        err = 0.0
        # with graph: the first step records and launches the graph, the next two replay it
        for _ in range(3 if graph else 2):
            a, b = part.timestep(0.25), one.timestep(0.25)
            err = max(err, abs(a - b) / abs(b))
            wd.progress()
        return err

    # Candidates, tried in order until all ranks pass the check: the library's own RCCL calls
    # with graph-replayed steps (the form BASELINE's 8-GPU configuration names), the same
    # launched eagerly, with and without the overlapped halo exchange, then the
    # torch.distributed callbacks.
    want = os.environ.get("KRY_DIST_COMM", "rccl" if backend == "nccl" else "torch")
    overlap_pref = os.environ.get("KRY_HALO_OVERLAP", "1")
    graph_pref = bool(args.graph) and os.environ.get("KRY_DIST_GRAPH", "1") != "0"
    candidates = []
    if want == "rccl" and graph_pref:
        candidates.append((want, overlap_pref, True))
    candidates.append((want, overlap_pref, False))
    if overlap_pref != "0":
        candidates.append((want, "0", False))
    if want != "torch":
        candidates.append(("torch", "0", False))
    comm_kind, check_err, ctx, comm, overlap, graph = None, None, None, None, None, False
    rejected = []
    for kind, ov, gr in candidates:
        wd.phase("start-up check: %s collectives, halo overlap %s, %s" % (kind, ov, "graph replay" if gr else "eager"),
                 startup_limit)
        os.environ["KRY_HALO_OVERLAP"] = ov  # read when a solver is created
        err, why = None, None
        try:
            if str(attempt) in test_hang.split(","):
                time.sleep(10.0 * startup_limit)
            if ctx is None or kind != getattr(ctx, "_kind", None):
                ctx, comm = make(kind)
                ctx._kind = kind
            err = selfcheck(ctx, gr)
            ok = err <= 1e-9
            if not ok:
                why = "self-check error %.3e against the single-rank replica" % err
        except Exception as e:  # noqa: BLE001 - any failure means: try the next candidate
            ok, why, ctx = False, repr(e), None
        flag = torch.tensor([1 if ok else 0], device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            comm_kind, overlap, graph, check_err = kind, ov != "0", gr, err
            break
        rejected.append({"collectives": kind, "halo_overlap": ov != "0", "graph": gr, "selfcheck_rel_err": err,
                         "reason_on_rank_0": why or "another rank rejected it"})
        if rank == 0:
            print("krylov_amd.dist_bench: candidate (%s collectives, halo overlap %s, %s) rejected: %s"
                  % (kind, ov, "graph replay" if gr else "eager", why or "another rank rejected it"),
                  file=sys.stderr, flush=True)
        ctx = None
    if comm_kind is None:
        raise RuntimeError("no working collectives (candidates rejected: %r)" % (rejected,))
    n_side, m, dt = args.mesh, args.m, args.dt
    ranges = partition_rows(n_side + 2, world)
    wd.phase("set-up and warm-up", step_limit)
    dev = host.ImplicitEuler(ctx, **problems.s_iso(n_side), m=m, bc_time_dependent=False,
                             row_range=ranges[rank], use_graph=graph)
    N = dev.n
    for _ in range(args.warmup):
        dev.timestep(dt)
        wd.progress()
    if comm_kind == "torch":
        comm.calls = {"allreduce": 0, "sendrecv": 0}  # count the timed steps only
    # no instrumentation inside the timed region; the per-kernel means come from a separate pass
    ctx.set_kernel_timing(False)
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    wd.phase("timed region", step_limit)
    t0 = time.perf_counter()
    last = 0.0
    for _ in range(args.steps):
        last = dev.timestep(dt)
        wd.progress()
    ctx.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    wd.phase("instrumented pass", step_limit)
    ms, cnt = [0.0] * capi.K_COUNT, [0] * capi.K_COUNT
    profile_steps = 0 if args.no_kernel_timing else args.profile_steps
    if profile_steps > 0:  # every rank takes part: the steps are collective
        dev.use_graph(False)
        ctx.set_kernel_timing(True)
        ctx.kernel_times()
        for _ in range(profile_steps):
            dev.timestep(dt)
            wd.progress()
        ms, cnt = ctx.kernel_times()
        ctx.set_kernel_timing(False)
        dist.barrier()
    # slowest rank defines the job time
    el = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    wd.stop()  # what follows is host work on rank 0 (the CPU sample takes a while) and the shutdown

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
            roofline = {"bound": "hbm",
                        "kernel": "%s on rank 0's row block" % (sweep_kernel_label() if name == "arnoldi_fused" else name + " kernels"),
                        "kernel_class": name,
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
                       "rows_per_rank": [b - a for a, b in ranges], "graph": bool(graph)},
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
            # what the start-up check turned down before it settled on the configuration above, and how often the
            # supervisor had to start the ranks again (0: the first processes ran through)
            "rejected_candidates": rejected,
            "attempt": attempt,
            "multi_gpu_status": "validated at start-up on this node; no run on more than one physical GPU was available "
                                "to the build of this library",
            "collectives_per_step": {k: (v // args.steps if v is not None else None) for k, v in comm.calls.items()},
            "kernels": kernels,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            # the same bounded CPU sample as in the single-GPU line, on rank 0 while the other ranks wait
            import bench
            out["cpu_baseline"] = bench.cpu_baseline(args.cpu_mesh or n_side, args.cpu_m, dt)
            out["cpu_baseline_all_cores"] = bench.cpu_baseline_all_cores(args.cpu_mesh or n_side, args.cpu_m, dt)
        print(json.dumps(out), file=out_stream or sys.stdout, flush=True)
    dist.barrier()
    dist.destroy_process_group()
