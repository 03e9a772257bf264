"""N > 1 path on CPU: world_size-2 (and 3) gloo runs of the TSQR orchestration (cuda-qr_amd/tsqr.py).

The collective, the stacking order of the gathered R factors, the redundant stacked QR and the
Q_p = Q_local [Qtree_p; 0] combine are exercised exactly as on GPUs; only the local factorisation
backend is swapped for the numpy mirror from the test oracle (injected here, by the test -- the
product has no CPU backend of its own)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class NumpyBackend:
    """TEST-ONLY local-step backend (oracle.np_geqrf / np_orgqr) with the HipBackend interface."""

    def __init__(self, oracle, m_local, n, world):
        self.O, self.m, self.n, self.world = oracle, m_local, n, world
        self.f_local = self.f_stack = None

    def new_matrix(self, rows, cols):
        return torch.zeros((cols, rows), dtype=torch.float64)

    @staticmethod
    def _np(t):
        return t.numpy().T          # (rows x cols) view of a column-major buffer

    def local_factor(self, A, R_out):
        F, tau, _ = self.O.np_geqrf(self._np(A), 8)
        self._np(A)[:] = F
        self.f_local = (F, tau)
        self._np(R_out)[:] = np.triu(F[:self.n])

    def stack_factor(self, S, R_out):
        F, tau, _ = self.O.np_geqrf(self._np(S), 8)
        self._np(S)[:] = F
        self.f_stack = (F, tau)
        self._np(R_out)[:] = np.triu(F[:self.n])

    def tree_q(self, S, Qt):
        self._np(Qt)[:] = self.O.np_orgqr(self.f_stack[0], self.f_stack[1], self.n, 8)

    def thin_q(self, A):
        Q = self.new_matrix(self.m, self.n)
        self._np(Q)[:] = self.O.np_orgqr(self.f_local[0], self.f_local[1], self.n, 8)
        return Q

    def local_q(self, A, C):
        F, tau = self.f_local
        Qfull = self.O.np_orgqr(F, tau, self.m, 8)
        self._np(C)[:] = Qfull @ self._np(C)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, m_local, n, outdir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cuda_qr_amd as qr
    from cuda_qr_amd import tsqr as T
    from oracle import oracle as O
    m = m_local * world
    shard = qr.uniform_matrix_host(m_local, n, row_off=rank * m_local, total_rows=m, seed=12)
    be = NumpyBackend(O, m_local, n, world)
    ts = T.TSQR(be, n, world, rank)
    A = be.new_matrix(m_local, n)
    A.numpy().T[:] = shard
    R = ts.factor(A)
    Q = ts.form_q(A)
    np.save(os.path.join(outdir, f"R{rank}.npy"), R.numpy().T.copy())
    np.save(os.path.join(outdir, f"Q{rank}.npy"), Q.numpy().T.copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_tsqr_orchestration_gloo(tmp_path, oracle, world):
    import cuda_qr_amd as qr
    m_local, n = 96, 24
    mp.spawn(_worker, args=(world, _free_port(), m_local, n, str(tmp_path)), nprocs=world, join=True)
    m = m_local * world
    A = qr.uniform_matrix_host(m, n, seed=12)
    Rs = [np.load(tmp_path / f"R{r}.npy") for r in range(world)]
    Q = np.vstack([np.load(tmp_path / f"Q{r}.npy") for r in range(world)])
    for r in range(1, world):
        assert np.array_equal(Rs[0], Rs[r]), "every rank must hold the identical final R"
    R = Rs[0]
    ref = oracle.sign_normalise(np.linalg.qr(A, mode="r"))
    assert np.linalg.norm(oracle.sign_normalise(R) - ref) / np.linalg.norm(ref) < 1e-13
    assert np.linalg.norm(A - Q @ R) / np.linalg.norm(A) < 1e-13
    assert np.linalg.norm(Q.T @ Q - np.eye(n)) < 1e-12


def test_tsqr_single_rank_path(oracle):
    import cuda_qr_amd as qr
    from cuda_qr_amd import tsqr as T
    m, n = 120, 16
    A0 = qr.uniform_matrix_host(m, n, seed=3)
    be = NumpyBackend(oracle, m, n, 1)
    ts = T.TSQR(be, n, 1, 0)
    A = be.new_matrix(m, n)
    A.numpy().T[:] = A0
    R = ts.factor(A).numpy().T
    Q = ts.form_q(A).numpy().T
    assert np.linalg.norm(A0 - Q @ R) / np.linalg.norm(A0) < 1e-14


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["c4", "tsqr"])
def test_bench_under_torchrun_rccl_matches_single_gpu(workload, tmp_path):
    """The REAL N > 1 path: bench.py under torch.distributed.run with backend nccl (= RCCL over xGMI), one rank per GPU, on
    every GPU of the node (skipped where fewer than 2 are visible -- the development box has one; this is the test that
    exercises the ordering between the plans' own streams and torch's NCCL stream and the device all-gather)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs at least 2 GPUs")
    P = 2 if torch.cuda.device_count() < 4 else 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={P}",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                          "--gpus", str(P), "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == P and line["value"] > 0
    # residual and orthogonality are computed over ALL shards (all-reduced): the gathered R factors were the right ones
    assert line["accuracy"]["resid"] < 1e-12 and line["accuracy"]["orth"] < 1e-11


@pytest.mark.gpu
def test_bench_bringup_two_ranks_on_one_gpu_through_the_c_abi_plan(tmp_path):
    """Bring-up of the N > 1 bench path on a ONE-GPU box: 2 ranks under torch.distributed.run share the device, the R factors
    travel through gloo (RCCL refuses duplicate devices) and everything else -- local QR, stacking, stacked QR, thin Q -- is the
    same C-ABI qr_tsqr_plan the RCCL path uses.  Checks the fields the driver and the judge read."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", BENCH_WATCHDOG_S="120")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c4", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    assert line["accuracy"]["resid"] < 1e-12 and line["accuracy"]["orth"] < 1e-11
    sp = line["tsqr_step_split"]
    assert sp["unpipelined_latency_ms"] >= sp["local_qr_ms"] > 0 and sp["pipelined_ms_per_step"] > 0
    assert line["config"]["collective"].startswith("1 all_gather")
    # the exchange's own timestamps (here from the self-gather form: the transport is not RCCL) and the joint fall-back flag
    print("tsqr_step_split:", {k: sp[k] for k in ("gather_ms", "gather_max_ms", "gather_call_ms", "fell_back_to_one_collective", "gather_source")})
    assert sp["gather_ms"] > 0 and sp["gather_call_ms"] > sp["gather_max_ms"] > 0 and sp["fell_back_to_one_collective"] is False
    assert "self-gather" in sp["gather_source"]
    # both exchange schedules, every rank's own numbers (here in the self-gather form; over RCCL the same fields from the collectives)
    ab = line["tsqr_exchange_by_schedule"]
    print("tsqr_exchange_by_schedule:", json.dumps({k: ab[k] for k in ("pipelined", "one_collective", "chosen_in_timed_region", "source")})[:1500])
    assert "error" not in ab, ab
    for name in ("pipelined", "one_collective"):
        pr = ab[name]["per_rank"]
        assert sorted(r["rank"] for r in pr) == [0, 1] and all(r["step_ms"] > 0 for r in pr)
    assert all(r["gather_ms"] > 0 and r["call_ms"] > r["gather_max_ms"] > 0 for r in ab["pipelined"]["per_rank"])
    one = line["same_problem_1gpu"]           # the same 262144 x 256 matrix on rank 0's GPU alone: the strong-scaling denominator
    assert one["ms"] > 0 and one["speedup_latency"] > 0 and one["speedup_throughput"] > 0


@pytest.mark.gpu
def test_rccl_single_rank_round_trip_through_the_librarys_loader():
    """What ONE GPU can check of the RCCL transport: the library's own loader (dlopen librccl, ncclGetUniqueId, ncclCommInitRank,
    ncclAllGather on a plan's stream) in a process that has torch loaded, like bench.py -- one communicator of one rank, an
    all-gather on torch-allocated and on library-allocated buffers, and which HIP / RCCL libraries the process ended up with (ONE of
    each: the library binds to the runtime torch brought, so its streams and torch's NCCL backend share one HIP runtime)."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_single_rank_roundtrip.py")], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ranks_seen_by_rccl"] == 1 and line["allgather_torch_buffers_ok"] and line["allgather_library_buffers_ok"]
    libs = line["mapped_runtime_libraries"]
    assert sum("amdhip64" in l for l in libs) == 1 and sum("rccl" in l for l in libs) == 1, libs
