"""TSQR over row shards, one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

No reference counterpart: the reference is single-GPU (qr.cu:711,737).  Structure:

  step 1  every rank factors its own (m/P) x n row block in place           -> R_p  (n x n upper)
  step 2  ONE all-gather of the packed R_p (n*n doubles per rank: 2 MiB at n = 512)
  step 3  every rank factors the stacked (P*n) x n matrix redundantly        -> R (identical on all ranks)
  step 4  (only if Q is wanted) Q_p = Q_local_p * [Qtree_p ; 0]

RCCL has no user-defined reduction, so the "all-reduce of R factors" of the north-star is an
all-gather + a redundant small QR; the messages are latency-bound (KBs..MBs), far below the xGMI
per-link bandwidth, so there is exactly one collective per factorisation and nothing to overlap.

Two drivers of the same steps:

  DeviceTSQR   the product path: the C-ABI qr_tsqr_plan (include/mi355x_qr.h) -- schedule, stream ordering and the RCCL
               all-gather all live in C (qr_host.c, qr_comm.hip); Python only carries the 128-byte unique id from rank 0 to
               the other ranks through torch.distributed.  bench.py --gpus N uses this.
  TSQR         the same steps orchestrated here over a `backend` object with torch.distributed collectives.  Tests inject a
               CPU backend to exercise stacking order and the Q combine with gloo (world 2 / 3, runs without a GPU); HipBackend
               is the device backend of that orchestration.  Nothing in this file falls back to a CPU path by itself.
"""
import torch
import torch.distributed as dist


class _RawDeviceBuffer:
    """zero-copy torch view of a device buffer the library owns (its exchange buffers), through __cuda_array_interface__"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _device_view(ptr, count, device):
    return torch.as_tensor(_RawDeviceBuffer(ptr, count), device=device)


class DeviceTSQR:
    """One rank's device-resident TSQR step through the C-ABI plan (qr_tsqr_factor_dev / qr_tsqr_formq_dev).

    transport "rccl": the all-gather is issued by the library on its own stream (ncclAllGather via the dlopen()ed librccl).
    transport "host": bring-up only (more ranks than GPUs; RCCL refuses duplicate devices) -- the R factors travel through
    host memory and the torch.distributed group (gloo); local and stacked steps are the same C entry points."""

    def __init__(self, qr, m_local, n, world, rank, nb=0, transport="rccl", group=None):
        self.qr, self.m, self.n, self.world, self.rank, self.group = qr, m_local, n, world, rank, group
        self.transport = transport if world > 1 else "none"
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.fallback_reason = None
        if world == 1:
            self.tp = qr.TsqrPlan(m_local, n, 1, 0, nb)
        elif transport == "rccl":
            # the library's own communicator (ncclCommInitRank through the dlopen()ed librccl).  If ANY rank cannot build it, every
            # rank falls back to transport "torch" together: same C plan for the local and stacked steps, the R factors gathered by
            # torch.distributed's collective directly on the plan's device buffers (one collective, host-synchronised)
            ok, err = 1, None
            try:
                box = [qr.tsqr_unique_id() if rank == 0 else None]
            except Exception as e:          # rank 0 could not load RCCL: still take part in the broadcast, then in the vote
                box, ok, err = [None], 0, repr(e)
            dist.broadcast_object_list(box, src=0, group=group)
            self.tp = None
            if ok and box[0] is not None:
                try:
                    self.tp = qr.TsqrPlan(m_local, n, world, rank, nb, unique_id=box[0])
                except Exception as e:
                    ok, err = 0, repr(e)
            else:
                ok = 0
            vote = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=group)
            if int(vote.item()) == 0:
                if self.tp is not None:
                    self.tp.close()
                self.transport = "torch"
                self.fallback_reason = err or "another rank could not create the library's RCCL communicator"
                self.tp = qr.TsqrPlan(m_local, n, world, rank, nb, comm="external")
                send, recv = self.tp.exchange_buffers()
                self._send_t = _device_view(send, n * n, dev)
                self._recv_t = _device_view(recv, world * n * n, dev)
        else:
            self.tp = qr.TsqrPlan(m_local, n, world, rank, nb, comm="external")
            self._send, self._recv = self.tp.exchange_buffers()
            self._host_send = torch.empty(n * n, dtype=torch.float64).pin_memory()
            self._host_recv = torch.empty(world * n * n, dtype=torch.float64).pin_memory()
        self.plan = self.tp.local              # fills, norms, products, profiling on the local plan's stream
        self.R = torch.empty((n, n), dtype=torch.float64, device=dev)

    def new_matrix(self, rows, cols):
        return torch.empty((cols, rows), dtype=torch.float64, device=self.device)

    def fill(self, A, rows, cols, row_off, total_rows, seed):
        self.plan.fill_uniform(A, rows, rows, cols, row_off=row_off, total_rows=total_rows, seed=seed)
        self.plan.sync()

    def factor(self, A):
        """Steps 1-3 on this rank's shard A (overwritten with its local factors); returns the buffer that will hold the final
        R (n x n column-major).  Asynchronous with transport rccl / none: call sync() before reading R."""
        if self.transport == "torch":
            self.tp.local_factor(A, self.m)
            self.tp.sync()
            dist.all_gather_into_tensor(self._recv_t, self._send_t, group=self.group)
            torch.cuda.current_stream().synchronize()
            self.tp.stacked_factor(self.R)
            return self.R
        if self.transport != "host":
            self.tp.factor(A, self.m, self.R)
            return self.R
        nn = self.n * self.n
        self.tp.local_factor(A, self.m)
        self.tp.sync()
        self.qr.check(self.qr.lib.qr_copy_to_host(self._host_send.data_ptr(), self._send, 8 * nn), "copy R out")
        dist.all_gather_into_tensor(self._host_recv, self._host_send, group=self.group)
        self.qr.check(self.qr.lib.qr_copy_to_device(self._recv, self._host_recv.data_ptr(), 8 * nn * self.world), "copy R in")
        self.tp.stacked_factor(self.R)
        return self.R

    def local_only(self, A):
        """step 1 alone (timing split)."""
        self.tp.local_factor(A, self.m)

    def form_q(self, A):
        Q = self.new_matrix(self.m, self.n)
        self.tp.formq(A, self.m, Q, self.m)
        self.tp.sync()
        return Q

    def sync(self):
        self.tp.sync()

    def close(self):
        self.tp.close()


def stacked_step_ms(qr, P, n, nb=0, reps=8):
    """Milliseconds of the stacked step of a P-rank factorisation (qr_tsqr_stacked_dev: stack P gathered n x n upper-triangular
    R factors, factor the (P n) x n matrix, extract R), un-pipelined, on the current device -- what every rank does after the
    all-gather, measurable on one GPU (a plan without a communicator; the gathered factors are synthetic)."""
    import time
    import numpy as np
    tp = qr.TsqrPlan(n, n, P, 0, nb, comm="external")
    _, recv = tp.exchange_buffers()
    rng = np.random.default_rng(5)
    blocks = np.triu(rng.random((P, n, n)) + np.eye(n) * 4.0)             # R_p(r, c), r <= c
    host = np.ascontiguousarray(blocks.transpose(0, 2, 1))                # [p][c][r]: column-major n x n blocks in rank order
    qr.check(qr.lib.qr_copy_to_device(recv, host.ctypes.data, host.nbytes), "copy R in")
    R = torch.empty((n, n), dtype=torch.float64, device="cuda")
    tp.stacked_factor(R); tp.sync()
    best = 1e30
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            tp.stacked_factor(R)
            tp.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    ref = np.linalg.qr(blocks.reshape(P * n, n), mode="r")
    got = np.triu(R.cpu().numpy().T)
    sg, sr = np.sign(np.diag(got))[:, None], np.sign(np.diag(ref))[:, None]
    err = float(np.linalg.norm(sg * got - sr * ref) / np.linalg.norm(ref))
    tp.close()
    if not err < 1e-12:
        raise RuntimeError(f"stacked step: R differs from LAPACK by {err:.2e}")
    return best


def rank_step_latency(qr, m_local, n, P, nb=128, nmat=3, reps=3):
    """Latency (ms) of ONE rank's TSQR step of a P-GPU factorisation, measured on one GPU, every step drained before the next starts:
    the local QR alone, and the complete step with the collective replaced by device copies of the rank's own factor
    (qr_tsqr_factor_selfgather_dev: the launches, streams and events of a real rank -- which also factors the full stacked matrix
    redundantly -- minus the network)."""
    import time
    tp = qr.TsqrPlan(m_local, n, P, 0, nb, comm="external")
    A = [torch.empty((n, m_local), dtype=torch.float64, device="cuda") for _ in range(nmat)]
    R = torch.empty((n, n), dtype=torch.float64, device="cuda")
    out = {"m_local": m_local, "n": n, "P": P, "nb": nb, "panel_pipelined": tp.is_pipelined()}
    for name, fn in (("local_ms", lambda a: tp.local_factor(a, m_local)), ("step_ms", lambda a: tp.factor_selfgather(a, m_local, R))):
        best = 1e30
        for _ in range(reps):
            for i, a in enumerate(A):
                tp.local.fill_uniform(a, m_local, m_local, n, seed=12 + i)
            tp.sync()
            t0 = time.perf_counter()
            for a in A:
                fn(a)
                tp.sync()
            best = min(best, (time.perf_counter() - t0) / len(A) * 1e3)
        out[name] = best
    out["exposed_exchange_and_stacked_ms"] = out["step_ms"] - out["local_ms"]
    out["local_share_of_step_excl_network"] = out["local_ms"] / out["step_ms"]     # not a scaling efficiency: see bench.py tsqr_model_1gpu
    tp.close()
    return out


class HipBackend:
    """Local steps on the current HIP device through libmi355xqr.so."""

    def __init__(self, qr, m_local, n, world, nb=0, ib=0):
        self.qr, self.m, self.n, self.world = qr, m_local, n, world
        self.plan = qr.Plan(m_local, n, nb, ib)
        self.plan_stack = qr.Plan(world * n, n, nb, ib) if world > 1 else None
        dev = torch.device("cuda", torch.cuda.current_device())
        self.tau = torch.empty(n, dtype=torch.float64, device=dev)
        self.tau_stack = torch.empty(n, dtype=torch.float64, device=dev)
        self.device = dev

    def new_matrix(self, rows, cols):
        """column-major rows x cols buffer (torch tensor of shape (cols, rows))."""
        return torch.empty((cols, rows), dtype=torch.float64, device=self.device)

    def fill(self, A, rows, cols, row_off, total_rows, seed):
        self.plan.fill_uniform(A, rows, rows, cols, row_off=row_off, total_rows=total_rows, seed=seed)
        self.plan.sync()

    def local_factor(self, A, R_out):
        """in-place QR of the shard; R_out (n x n column-major buffer) = its R factor."""
        self.plan.geqrf(A, self.m, self.n, self.m, self.tau)
        self.plan.extract_r(A, self.m, self.n, self.m, R_out, self.n, self.n)
        self.plan.sync()            # hand R_out to the collective's stream

    def stack_factor(self, S, R_out, wait=True):
        """in-place QR of the stacked (P*n) x n matrix; R_out = final R.  wait=False leaves it queued on the stack
        plan's streams (stack_sync() before S, R_out or the factors are touched again)."""
        sm = self.world * self.n
        self.plan_stack.geqrf(S, sm, self.n, sm, self.tau_stack)
        self.plan_stack.extract_r(S, sm, self.n, sm, R_out, self.n, self.n)
        if wait:
            self.plan_stack.sync()

    def stack_sync(self):
        if self.plan_stack:
            self.plan_stack.sync()

    def tree_q(self, S, Qt):
        sm = self.world * self.n
        self.plan_stack.applyq(S, sm, self.n, sm, self.tau_stack, Qt, self.n, sm, True)
        self.plan_stack.sync()

    def thin_q(self, A):
        """thin Q (m_local x n) of the local factorisation alone (single-rank case)."""
        Q = self.new_matrix(self.m, self.n)
        self.plan.applyq(A, self.m, self.n, self.m, self.tau, Q, self.n, self.m, True)
        self.plan.sync()
        return Q

    def local_q(self, A, C):
        """C (m_local x n, holding [Qtree_p; 0]) <- Q_local * C."""
        self.plan.applyq(A, self.m, self.n, self.m, self.tau, C, self.n, self.m, False)
        self.plan.sync()

    def close(self):
        self.plan.close()
        if self.plan_stack:
            self.plan_stack.close()


class TSQR:
    """Reusable buffers + the 3(4)-step schedule.  `group` is a torch.distributed process group or
    None for a single process."""

    def __init__(self, backend, n, world, rank, group=None, stage_through_host=False):
        # stage_through_host: the R factors go through pinned host memory and a CPU collective (gloo).  Only for
        # bring-up of the multi-rank path on a box with fewer GPUs than ranks; RCCL (backend "nccl") is the product path.
        self.b, self.n, self.world, self.rank, self.group = backend, n, world, rank, group
        self.stage_through_host = stage_through_host
        self.R_local = backend.new_matrix(n, n)
        self.R = backend.new_matrix(n, n)
        # all ranks' R factors, rank-major: gathered[p] is rank p's n x n (column-major) block
        self.gathered = backend.new_matrix(n, n * world).view(world, n, n) if world > 1 else None
        self.stack = backend.new_matrix(world * n, n) if world > 1 else None

    def factor(self, A, pipelined=False):
        """Steps 1-3.  A (this rank's shard, column-major buffer) is overwritten with its local factors.
        Returns the buffer holding the final R (n x n, column-major, identical on every rank).
        pipelined=True: the stacked factorisation (small, latency-bound, redundant on every rank) is left queued on its
        own plan's streams, so that in a sequence of independent factorisations it runs under the next shard's local QR;
        call sync() before reading R or calling form_q()."""
        b, n, P = self.b, self.n, self.world
        if P == 1:
            b.local_factor(A, self.R)
            return self.R
        b.local_factor(A, self.R_local)
        if self.stage_through_host and self.R_local.is_cuda:
            loc = self.R_local.cpu()
            got = torch.empty((P * n, n), dtype=loc.dtype)
            dist.all_gather_into_tensor(got, loc, group=self.group)
            self.gathered.view(P * n, n).copy_(got)
        else:
            dist.all_gather_into_tensor(self.gathered.view(P * n, n), self.R_local, group=self.group)
        # gathered[p][c][r] = R_p(r, c); the stacked matrix is column-major (P*n) x n with R_p in rows
        # [p*n, (p+1)*n): stack[c][p*n + r] = gathered[p][c][r]
        if pipelined and hasattr(b, "stack_sync"):
            b.stack_sync()                      # the previous stacked factorisation still owns self.stack / self.R
        self.stack.view(n, P, n).copy_(self.gathered.permute(1, 0, 2))
        if self.stack.is_cuda:
            torch.cuda.current_stream().synchronize()
        if pipelined and hasattr(b, "stack_sync"):
            b.stack_factor(self.stack, self.R, wait=False)
        else:
            b.stack_factor(self.stack, self.R)
        return self.R

    def sync(self):
        if hasattr(self.b, "stack_sync"):
            self.b.stack_sync()

    def form_q(self, A):
        """Step 4 after factor(A): returns this rank's m_local x n block of the thin Q (new buffer)."""
        b, n, P = self.b, self.n, self.world
        if P == 1:
            return b.thin_q(A)
        self.sync()
        Qt = b.new_matrix(P * n, n)
        b.tree_q(self.stack, Qt)
        C = b.new_matrix(b.m, n)
        C.zero_()
        C.view(n, b.m)[:, :n].copy_(Qt.view(n, P * n)[:, self.rank * n:(self.rank + 1) * n])
        if C.is_cuda:
            torch.cuda.current_stream().synchronize()
        b.local_q(A, C)
        return C
