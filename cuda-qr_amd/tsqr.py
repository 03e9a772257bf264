"""TSQR over row shards, one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

No reference counterpart: the reference is single-GPU (qr.cu:711,737).  Structure:

  step 1  every rank factors its own (m/P) x n row block in place           -> R_p  (n x n upper)
  step 2  ONE all-gather of the packed R_p (n*n doubles per rank: 2 MiB at n = 512)
  step 3  every rank factors the stacked (P*n) x n matrix redundantly        -> R (identical on all ranks)
  step 4  (only if Q is wanted) Q_p = Q_local_p * [Qtree_p ; 0]

RCCL has no user-defined reduction, so the "all-reduce of R factors" of the north-star is an
all-gather + a redundant small QR; the messages are latency-bound (KBs..MBs), far below the xGMI
per-link bandwidth, so there is exactly one collective per factorisation and nothing to overlap.

The local factorisations go through a `backend` object.  The product backend is HipBackend (the
C-ABI library through cuda_qr_amd.Plan).  Tests inject a CPU backend to exercise the orchestration
with gloo; nothing in this file falls back to a CPU path by itself.
"""
import torch
import torch.distributed as dist


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
