"""Row-block decomposition over several ranks (one process per GPU).

The C library owns the distributed algorithm (which rows travel where, which
scalars are summed when); this module supplies the two collectives it calls
back for, on top of ``torch.distributed``:

  * ``RcclComm``: the library calls RCCL itself on its stream; Python only
    hands the communicator id around (the fast path on real multi-GPU nodes);
  * ``TorchComm`` with backend ``nccl`` (RCCL over xGMI): device tensors
    wrapping the library's buffers, enqueued on the context's stream;
  * any other backend (``gloo``): the buffers are staged through host memory.
    Slower, but it lets several ranks share one GPU, or none at all (tests).

Mesh rows are split into contiguous ranges of padded rows (``partition_rows``);
rank r exchanges one boundary row with ranks r-1 and r+1 before every operator
apply, and small sums (dot products, norms, max|delta|) are all-reduced.
"""
import ctypes as C

import numpy as np

from . import capi

_dp = C.POINTER(C.c_double)


def partition_rows(total_rows, nranks):
    """Contiguous, balanced ranges [(begin, end)] of padded rows, one per rank.

    The first ``total_rows % nranks`` ranks get one extra row; every rank owns
    at least one row."""
    if nranks < 1 or total_rows < nranks:
        raise ValueError("cannot split %d rows over %d ranks" % (total_rows, nranks))
    base, extra = divmod(total_rows, nranks)
    out, start = [], 0
    for r in range(nranks):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


class _DevArray:
    """Minimal __cuda_array_interface__ carrier for a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8",
                                         "data": (int(ptr), False), "version": 2}


class HostCollectives:
    """The two collectives on numpy arrays over a torch.distributed group
    (any backend that accepts CPU tensors, e.g. gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)

    def allreduce(self, arr, op):
        import torch
        t = torch.from_numpy(arr)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == capi.REDUCE_MAX else self.dist.ReduceOp.SUM,
                             group=self.group)
        return arr

    def sendrecv(self, send, send_peer, recv, recv_peer):
        """Send `send` to send_peer and receive into `recv` from recv_peer
        (peer < 0: no such transfer).  Both transfers are posted before either
        is waited for, so matching calls on neighbouring ranks cannot deadlock."""
        import torch
        reqs = []
        if recv_peer >= 0:
            reqs.append(self.dist.irecv(torch.from_numpy(recv), src=recv_peer, group=self.group))
        if send_peer >= 0:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(send)), dst=send_peer, group=self.group))
        for r in reqs:
            r.wait()
        return recv

    def halo_exchange(self, local, row_len):
        """Numpy statement of the exchange the library performs on a slab
        vector: `local` is (rows, row_len); returns (lower_halo, upper_halo),
        zeros at the ends of the mesh.  The lower neighbour is rank-1."""
        down = self.rank - 1 if self.rank > 0 else -1
        up = self.rank + 1 if self.rank + 1 < self.nranks else -1
        lo = np.zeros(row_len)
        hi = np.zeros(row_len)
        self.sendrecv(np.ascontiguousarray(local[0]), down, hi, up)
        self.sendrecv(np.ascontiguousarray(local[-1]), up, lo, down)
        return lo, hi


class TorchComm:
    """Collectives for one context: installs the callbacks the library needs."""

    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx = ctx
        self.group = group
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        self.device_native = dist.get_backend(group) == "nccl"
        self.host = None if self.device_native else HostCollectives(group)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.stream = torch.cuda.ExternalStream(ctx.stream, device=self.device)
        self._streams = {int(ctx.stream or 0): self.stream}
        self._tensors = {}
        self.calls = {"allreduce": 0, "sendrecv": 0}
        self._ar = capi.ALLREDUCE_FN(self._allreduce)
        self._sr = capi.SENDRECV_FN(self._sendrecv)
        capi.check(capi.lib().kry_ctx_set_comm(ctx.h, self.rank, self.nranks,
                                               C.cast(self._ar, C.c_void_p), C.cast(self._sr, C.c_void_p), None))
        ctx._comm = self  # keep the callbacks alive as long as the context

    def _stream(self, handle):
        """torch view of the hipStream_t the library hands to a callback (its own stream, or
        the second one on which a halo exchange overlaps with compute)."""
        key = int(handle or 0)
        st = self._streams.get(key)
        if st is None:
            st = self.torch.cuda.ExternalStream(key, device=self.device)
            self._streams[key] = st
        return st

    def _tensor(self, ptr, count):
        key = (ptr, count)
        t = self._tensors.get(key)
        if t is None:
            t = self.torch.as_tensor(_DevArray(ptr, count), device=self.device)
            self._tensors[key] = t
        return t

    def _allreduce(self, user, ptr, count, op, stream):
        try:
            self.calls["allreduce"] += 1
            t = self._tensor(ptr, count)
            with self.torch.cuda.stream(self._stream(stream)):
                if self.device_native:
                    rop = self.dist.ReduceOp.MAX if op == capi.REDUCE_MAX else self.dist.ReduceOp.SUM
                    self.dist.all_reduce(t, op=rop, group=self.group)
                else:
                    h = t.cpu().numpy()  # synchronises the stream
                    self.host.allreduce(h, op)
                    t.copy_(self.torch.from_numpy(h))
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("krylov_amd.dist: allreduce failed: %r" % (e,), flush=True)
            return 1

    def _sendrecv(self, user, send_ptr, send_peer, recv_ptr, recv_peer, count, stream):
        try:
            self.calls["sendrecv"] += 1
            with self.torch.cuda.stream(self._stream(stream)):
                if self.device_native:
                    ops = []
                    if recv_peer >= 0:
                        ops.append(self.dist.P2POp(self.dist.irecv, self._tensor(recv_ptr, count), recv_peer, self.group))
                    if send_peer >= 0:
                        ops.append(self.dist.P2POp(self.dist.isend, self._tensor(send_ptr, count), send_peer, self.group))
                    if ops:
                        for w in self.dist.batch_isend_irecv(ops):
                            w.wait()
                else:
                    send = self._tensor(send_ptr, count).cpu().numpy() if send_peer >= 0 else None
                    recv = np.empty(count) if recv_peer >= 0 else None
                    self.host.sendrecv(send, send_peer, recv, recv_peer)
                    if recv_peer >= 0:
                        self._tensor(recv_ptr, count).copy_(self.torch.from_numpy(recv))
            return 0
        except Exception as e:
            print("krylov_amd.dist: sendrecv failed: %r" % (e,), flush=True)
            return 1


class RcclComm:
    """The library issues its collectives through RCCL itself (no Python in the
    data path, hipGraph-capturable); torch.distributed only carries the
    128-byte communicator id from rank 0 to the other ranks."""

    device_native = True

    def __init__(self, ctx, group=None):
        import torch.distributed as dist
        self.ctx = ctx
        self.rank = dist.get_rank(group)
        self.nranks = dist.get_world_size(group)
        L = capi.lib()
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            capi.check(L.kry_rccl_unique_id(buf))
        box = [bytes(buf.raw) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        capi.check(L.kry_ctx_set_comm_rccl(ctx.h, self.rank, self.nranks, box[0]))
        self.calls = {"allreduce": None, "sendrecv": None}  # not counted: nothing passes through Python
        ctx._comm = self

    def selftest(self):
        capi.check(capi.lib().kry_ctx_comm_selftest(self.ctx.h))


def gather_rows(local, ranges, host):
    """All ranks' owned rows -> the whole (total_rows, row_len) array on every
    rank (host collectives; for checks, not for the hot path)."""
    total = ranges[-1][1]
    full = np.zeros((total, local.shape[1]))
    r0, r1 = ranges[host.rank]
    full[r0:r1] = local
    # rows are disjoint, everything else is zero: a sum assembles the array;
    # NaN entries (ghost corners) are carried separately
    nan_mask = np.isnan(full)
    full[nan_mask] = 0.0
    host.allreduce(full.reshape(-1), capi.REDUCE_SUM)
    mask = nan_mask.astype(np.float64)
    host.allreduce(mask.reshape(-1), capi.REDUCE_SUM)
    full[mask > 0] = np.nan
    return full
