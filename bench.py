#!/usr/bin/env python3
"""bench.py -- the round benchmark contract.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one full fp64 Householder QR factorisation of a synthetic dense matrix that is already
resident in HBM when the timed region starts (uniform[0,1) from the library's counter-based
generator, seed-indexed per step; no regeneration, copy or PCIe traffic inside the timed region).

  N = 1      BASELINE config C3: 16384 x 16384 on one MI355X (--workload c2 for 4096 x 4096, nb 64;
             --workload tsqr for one 262144 x 512 shard)
  N = 2, 4   BASELINE config C4: tall-skinny 262144 x 256, row blocks of 262144/N rows per GPU (strong scaling)
  N = 8      BASELINE config C5: 2097152 x 512, one 262144 x 512 row block per GPU
  (other N, or --workload tsqr: the weak series, 262144 x 512 per GPU)
             TSQR = local QR + ONE RCCL all-gather of the R factors + redundant stacked QR.

Prints ONE JSON line on rank 0: metric fp64 GFLOP/s (F = 2mn^2 - 2n^3/3 per factorisation, whole job),
plus `roofline` for the dominant kernel (the trailing-update MFMA GEMM; for TSQR the panel kernels),
`cpu_baseline` (the REAL reference qr.c, or the oracle port of it, on one host core over a bounded
sample) and the accuracy figures of the north-star (||A-QR||_F/||A||_F, ||Q^T Q - I||_F).
"""
import argparse
import json
import os
import sys
import threading
import time

# The host driver of this pool only supports dmabuf IPC: RCCL's intra-node transport (and CUDA-tensor sharing across processes)
# fails with "hipIpcGetMemHandle: invalid argument" without this.  Must be in the environment BEFORE the HSA runtime starts, i.e.
# before `import torch` touches the GPU; a value set by the launcher wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X datasheet, fp64 matrix (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz);
                                 # /opt/skills/guides/MI355X_MICROARCH.md lists no fp64 MFMA row
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto", choices=["auto", "c2", "c3", "tsqr", "c4", "c5"])
    ap.add_argument("--nb", type=int, default=0)
    ap.add_argument("--ib", type=int, default=0)
    ap.add_argument("--cond", type=float, default=0.0,
                    help="ill-conditioned input: every 128-column panel gets condition ~COND (columns = the panel's first column + "
                         "uniform noise / COND), so the device-side guard refuses every full-width panel -- prices the refusal "
                         "(VERDICT r5 item 4); 0 = the plain uniform matrix")
    ap.add_argument("--no-check", action="store_true", help="skip the post-run residual/orthogonality check")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="792x384", help="m x n of the bounded multi-panel CPU-baseline sample")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("BENCH_WATCHDOG_S", "30")),
                    help="seconds a bring-up phase (rendezvous, communicator, first collective) may take before the process exits 124")
    return ap.parse_args()


class Watchdog:
    """A phase that does not finish in time ends THIS process with a non-zero exit code (os._exit: no re-exec of a process that
    has touched the GPU, no retry loop); the launcher (torch.distributed.run) then tears the other ranks down."""

    def __init__(self, rank):
        self.rank, self.timer = rank, None

    def arm(self, seconds, what):
        self.disarm()
        if seconds <= 0:
            return

        def fire():
            sys.stderr.write(json.dumps({"bench_watchdog": what, "rank": self.rank, "timeout_s": seconds,
                                         "hint": "RCCL on this pool needs HSA_ENABLE_IPC_MODE_LEGACY=0 and one visible GPU per rank"}) + "\n")
            sys.stderr.flush()
            os._exit(124)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer:
            self.timer.cancel()
            self.timer = None


def flops(m, n):
    return 2.0 * m * n * n - 2.0 * n ** 3 / 3.0


def cpu_baseline(sample):
    """The reference qr.c (oracle/_ref when it travelled, else the bitwise-pinned oracle restatement; Scalar=double, PR=64,
    PC=8) on ONE host core -- the reference has no threading anywhere -- over a bounded sample (~2-3 s in total):
      C1      BASELINE configs[0], 512 x 128 from srand(12): mmqr (the `value` of this object), qr.c:477
      multi   a multi-panel matrix of the same kind (default 792 x 384): mmqr
      expQ    explicitQR (dense m x m Q by one m^3 product per reflector, qr.c:494) once, on 120 x 32 -- at C1 itself it takes
              314 s on one core, which the line quotes as a committed measurement, not a live one."""
    import numpy as np
    from oracle import oracle as O
    PR, PC = 64, 8
    kind = "reference" if O.ref_path(np.float64, PR, PC) else "port"
    f_mmqr = (lambda A: O.ref_mmqr(A, PR, PC)) if kind == "reference" else (lambda A: O.mmqr(A, PR, PC))
    f_expq = (lambda F, t: O.ref_explicit_qr(F, t, PR, PC)) if kind == "reference" and hasattr(O, "ref_explicit_qr") \
        else (lambda F, t: O.explicit_qr(F, t, PR, PC))
    legs = {}

    def leg_mmqr(name, m, n):
        O.check_shape(m, n, PR, PC)
        A = O.fill_rand(m, n, 12, np.float64)
        t0 = time.perf_counter()
        out = f_mmqr(A)
        dt = time.perf_counter() - t0
        legs[name] = {"m": m, "n": n, "seconds": dt, "gflops": flops(m, n) / dt / 1e9}
        return out

    leg_mmqr("C1_mmqr_512x128", 512, 128)
    sm, sn = (int(x) for x in sample.split("x"))
    leg_mmqr(f"multi_mmqr_{sm}x{sn}", sm, sn)
    out = leg_mmqr("expQ_mmqr_120x32", 120, 32)
    try:
        t0 = time.perf_counter()
        f_expq(out[0], out[1])
        legs["expQ_explicitQR_120x32"] = {"m": 120, "n": 32, "seconds": time.perf_counter() - t0,
                                          "note": "dense m x m Q, one m^3 product per reflector (qr.c:415-429)"}
    except Exception as e:          # the checker's explicitQR wrapper differs between oracle builds: never fail the bench over it
        legs["expQ_explicitQR_120x32"] = {"error": repr(e)}
    c1 = legs["C1_mmqr_512x128"]
    return {"value": c1["gflops"], "unit": "GFLOP/s", "cores": 1, "kind": kind,
            "host_cores_available": os.cpu_count(), "seconds": sum(v.get("seconds", 0.0) for v in legs.values()),
            "legs": legs,
            "committed_not_live": {"C1_explicitQR_512x128_seconds": 313.7, "source": "BASELINE.md section 2 (real qr.c, 1 core)"},
            "sample": ("the real reference qr.c (compiled by oracle/Makefile into oracle/_ref) mmqr" if kind == "reference" else
                       "the oracle's C restatement of qr.c mmqr (oracle/mmqr_oracle.c: oracle/_ref did not travel / is not built)")
                      + " (Scalar=double, PR=64, PC=8) on C1 = 512x128 uniform[0,1) (srand(12) generator, "
                      "qr.c:468-474), stdout to /dev/null, 1 thread, useful flops 2mn^2-2n^3/3; legs: the same on a "
                      f"{sm}x{sn} multi-panel matrix, and mmqr + explicitQR once on 120x32"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    wd = Watchdog(rank)
    wd.arm(args.watchdog + 90.0, "first GPU touch (import torch pages the image in: up to 2 min on a fresh box)")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    ndev = torch.cuda.device_count()
    # BENCH_BACKEND=gloo: bring-up only -- lets N ranks share fewer GPUs (RCCL refuses duplicate devices); the R
    # factors are then all-gathered through host memory.  The default and every reported number use RCCL.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        wd.arm(args.watchdog + 60.0, "torch.distributed rendezvous / process group")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.ib:
        os.environ["MI355XQR_IB"] = str(args.ib)         # read once when the library first needs its defaults
    import cuda_qr_amd as qr
    from cuda_qr_amd import tsqr as T

    wl = args.workload
    if wl == "auto":        # the driver's --gpus N lands on BASELINE's configs: C3 / C4 / C4 / C5
        wl = {1: "c3", 2: "c4", 4: "c4", 8: "c5"}.get(world, "tsqr")
    if wl == "c2":
        m_local, n, nb, desc = 4096, 4096, args.nb or 64, "C2: 4096x4096 square fp64 QR, block size 64"
    elif wl == "c3":
        m_local, n, nb = 16384, 16384, args.nb or 256
        desc = f"C3: 16384x16384 square fp64 QR on 1 MI355X, nb={nb}"
    elif wl == "c4":
        m_local, n, nb = 262144 // world, 256, args.nb or 128     # round 5: shards of 131072 / 65536 rows take the full-width panel route at nb 128
                                                                  # (rank step 1.59 / 1.18 ms against 1.80 / 1.30 at nb 64, profiles/r06_tsqr_rank_step_latency.txt)
        desc = f"C4: tall-skinny 262144x256 fp64 TSQR, row-block sharded over {world} GPU(s)"
    elif wl == "c5":
        m_local, n, nb = 2097152 // world, 512, args.nb or 128
        desc = f"C5: tall-skinny 2097152x512 fp64 TSQR, row-block sharded over {world} GPU(s)"
    else:
        m_local, n, nb = 262144, 512, args.nb or 128
        desc = (f"TSQR weak scaling: {world} x (262144x512) row shards = {262144 * world}x512 fp64"
                + (" (= C5)" if world == 8 else ""))
    m_total = m_local * world
    if wl in ("c2", "c3") and world > 1:
        raise SystemExit("square configs do not shard (replicas only, DESIGN.md); use --workload c4 | c5 | tsqr")
    scaling = "strong" if wl in ("c4", "c5") else "weak"      # c4 / c5: the total matrix is fixed, shards shrink with N

    # the C-ABI device-resident step (qr_tsqr_plan: local QR -> ncclAllGather -> stacked QR, all issued from C); with one rank it
    # is the plain qr_plan of the square configs.  Creating it is collective for N > 1 (ncclCommInitRank).
    wd.arm(args.watchdog + (60.0 if world > 1 else 0.0), "qr_tsqr_plan_create (ncclCommInitRank over the ranks of this node)")
    be = T.DeviceTSQR(qr, m_local, n, world, rank, nb, transport="rccl" if backend == "nccl" else "host")
    ts = be
    rccl_ranks = be.tp.comm_ranks() if (world > 1 and backend == "nccl" and be.transport == "rccl") else None
    wd.disarm()
    K, W = args.steps, args.warmup
    bytes_per = 8 * m_local * n
    nbuf = min(K + W, max(1, int(160e9 // bytes_per)))
    bufs = [be.new_matrix(m_local, n) for _ in range(nbuf)]
    seeds = [12 + i for i in range(nbuf)]
    def condition(A):
        """--cond: columns c+1 .. c+127 of every 128-column panel <- the panel's first column + themselves / cond (A is the (n, m) torch
        image of the column-major matrix: a row of the tensor is a column of the matrix).  The panel then has one direction of size
        ~sqrt(m / 3) and 127 of size ~sqrt(m / 12) / cond: CholeskyQR2 at panel width cannot take it above cond ~ 1e7."""
        if args.cond > 0.0:
            torch.cuda.synchronize()
            for c in range(0, n, 128):
                A[c + 1:c + 128] = A[c:c + 1] + A[c + 1:c + 128] / args.cond
            torch.cuda.synchronize()

    for A, s in zip(bufs, seeds):
        be.fill(A, m_local, n, rank * m_local, m_total, s)
        condition(A)
    torch.cuda.synchronize()

    def barrier():
        ts.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        A = bufs[i % nbuf]
        if i >= nbuf:                       # only when K+W exceeds the buffer pool: regenerate (stated in config)
            be.fill(A, m_local, n, rank * m_local, m_total, seeds[i % nbuf])
            condition(A)
        return ts.factor(A)      # stream-ordered in C; independent factorisations: the small stacked QR of step i runs
                                 # under the local QR of step i+1 (everything is finished inside the timed bracket)

    wd.arm(args.watchdog + 30.0, "warm-up steps (first collective)")
    for i in range(W):
        step(i)
    barrier()
    wd.arm(max(300.0, args.watchdog), "timed region")
    # HIP events on the plan's own streams, inside the timed region -- around the launches of the DOMINANT kernel only (class 0:
    # the wide update's gemm_nt; tall-skinny: class 2, the panel): every record is two event packets on a stream (~4 us each),
    # and with all six classes recorded the 16384^2 step took 128.5 ms instead of 126
    dominant_cls = 0 if wl in ("c2", "c3") else 2
    be.plan.set_profile(2 * (1 << dominant_cls))
    # ... and on a SAMPLE of the timed steps (every `stride`-th, at least 4 of them when K allows): the records still bracket the
    # dominant kernel's launches inside the timed region, on the stream they are launched on
    stride = max(1, K // 4)
    t0 = time.perf_counter()
    for i in range(W, W + K):
        be.plan.pause_profile((i - W) % stride != 0)
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    be.plan.pause_profile(False)
    prof = be.plan.get_profile()
    # the other classes (W = (V T)^T A2, panel chain, ...): one more factorisation, fully profiled, OUTSIDE the timed region
    be.plan.set_profile(True)
    step(W + K)
    barrier()
    prof_full = be.plan.get_profile()
    be.plan.set_profile(False)
    wd.arm(max(300.0, args.watchdog), "post-run checks")
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    total_flops = K * flops(m_total, n)
    value = total_flops / dt / 1e9

    # ---- accuracy (outside the timed region): last factored matrix, thin Q, on-device norms
    acc = {}
    if not args.no_check:
        last = (W + K) % nbuf               # the matrix of the extra (fully profiled) step: the last one factored, ts.R is its R
        A = bufs[last]
        ts.sync()
        R = ts.R
        Q = ts.form_q(A)
        QR = be.new_matrix(m_local, n)
        G = be.new_matrix(n, n)
        be.plan.gemm("N", m_local, n, n, 1.0, Q, m_local, R, n, 0.0, QR, m_local)
        be.plan.gemm("T", n, n, m_local, 1.0, Q, m_local, Q, m_local, 0.0, G, n)
        be.plan.sync()
        if args.cond > 0.0:                 # the input is not the generator's any more: rebuild it (same seed, same transformation)
            cond_ref = be.new_matrix(m_local, n)
            be.fill(cond_ref, m_local, n, rank * m_local, m_total, seeds[last])
            condition(cond_ref)
            d, a = be.plan.diffnorm(QR, m_local, m_local, n, dY=cond_ref, ldy=m_local)
            del cond_ref
        else:
            d, a = be.plan.diffnorm(QR, m_local, m_local, n, row_off=rank * m_local, total_rows=m_total,
                                    seed=seeds[last])
        sums = torch.tensor([d, a], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(sums)
            if coll_dev == "cpu":
                Gh = G.cpu()
                dist.all_reduce(Gh)
                G.copy_(Gh)
            else:
                dist.all_reduce(G)                              # Q^T Q = sum over shards
            torch.cuda.synchronize()                            # the plan's streams are not ordered with torch's
        o, _ = be.plan.diffnorm(G, n, n, n, mode=1)
        acc = {"resid": float((sums[0] / sums[1]).sqrt().item()), "orth": float(o ** 0.5)}
        del Q, QR, G

    # ---- N > 1 only, outside the timed region: the UN-pipelined latency of one factorisation (every step drained before the
    # next starts, so the exchange and the stacked QR are fully exposed) and where it goes
    tsqr_split = None
    if world > 1:
        reps = max(1, min(K, 3))
        for i in range(reps):
            be.fill(bufs[i % nbuf], m_local, n, rank * m_local, m_total, seeds[i % nbuf])
        barrier()
        t1 = time.perf_counter()
        for i in range(reps):
            be.local_only(bufs[i % nbuf])
            ts.sync()
        loc_ms = (time.perf_counter() - t1) / reps * 1e3
        for i in range(reps):
            be.fill(bufs[i % nbuf], m_local, n, rank * m_local, m_total, seeds[i % nbuf])
        barrier()
        t1 = time.perf_counter()
        for i in range(reps):
            ts.factor(bufs[i % nbuf])
            ts.sync()
        lat_ms = (time.perf_counter() - t1) / reps * 1e3
        # the library's own timestamps of the last factorisation's gathers (stacked stream past its wait for the local panel ->
        # collective done, summed over the block columns): what the exchange costs on THIS node, max over ranks
        gs, gs_src = (be.tp.gather_stats(), "ncclAllGather per block column (RCCL)") if be.transport == "rccl" else (None, None)
        if gs is None and be.tp.is_pipelined():
            # bring-up transports (R factors through torch.distributed): the same pipelined schedule with this rank's own factor
            # copied into every slot -- the fields exist and the event plumbing has run before the first RCCL run reads them
            for i in range(3):
                be.tp.factor_selfgather(bufs[i % nbuf], m_local, be.R)
            gs, gs_src = be.tp.gather_stats(), "self-gather: device copies in place of the collective (transport %s)" % be.transport
        lat = torch.tensor([lat_ms, loc_ms] + ([gs["gather_ms"], gs["gather_max_ms"], gs["call_ms"]] if gs else [0.0, 0.0, 0.0]),
                           dtype=torch.float64, device=coll_dev)
        dist.all_reduce(lat, op=dist.ReduceOp.MAX)
        lat_ms, loc_ms = float(lat[0].item()), float(lat[1].item())
        tsqr_split = {"unpipelined_latency_ms": lat_ms, "local_qr_ms": loc_ms,
                      "exchange_and_stacked_qr_ms": lat_ms - loc_ms,
                      "gather_ms": float(lat[2].item()) if gs else None,
                      "gather_max_ms": float(lat[3].item()) if gs else None,
                      "gather_call_ms": float(lat[4].item()) if gs else None,
                      "fell_back_to_one_collective": bool(gs["fell_back"]) if gs else None,
                      "gather_source": gs_src,
                      "pipelined_ms_per_step": dt / K * 1e3,
                      "panel_pipelined_exchange": bool(be.transport == "rccl" and be.tp.is_pipelined()),
                      "unpipelined_gflops": flops(m_total, n) / (lat_ms * 1e-3) / 1e9,
                      "note": "max over ranks; un-pipelined = qr_tsqr_factor_dev + qr_tsqr_sync per step (the single-factorisation "
                              "latency); `value` is the throughput of K independent factorisations issued back to back; "
                              "panel_pipelined_exchange: inside ONE factorisation the R factors travel block column by block column "
                              "and the stacked (world*n) x n matrix (redundant on every rank, latency-bound) is factored "
                              "left-looking while the local QR continues"}

    # ---- N > 1: BOTH exchange schedules, three drained factorisations each, every rank's own numbers (qr_tsqr_set_schedule is collective:
    # all ranks switch between the same two calls).  The first real multi-GPU run must explain itself: what an all-gather of n*nb doubles per
    # rank costs beside a chip-filling local update, on THIS node, is not known before it.
    by_schedule = None
    if world > 1 and os.environ.get("BENCH_SCHEDULE_AB", "1") != "0":
        try:
            by_schedule = {}
            real = be.transport == "rccl"
            chosen = "pipelined" if be.tp.is_pipelined() else "one_collective"

            def one_step(buf):
                # bring-up transports (R factors through torch.distributed): the same two schedules with this rank's own factor copied
                # into every slot -- the launches, streams and events of a real rank, so the fields exist before the first RCCL run
                if real:
                    ts.factor(buf)
                else:
                    be.tp.factor_selfgather(buf, m_local, be.R)
                ts.sync()

            for name, mode in (("pipelined", 1), ("one_collective", 0)):
                try:
                    be.tp.set_schedule(mode)
                except Exception as e:                 # the shape cannot run the pipelined form
                    by_schedule[name] = {"error": repr(e)}
                    continue
                for i in range(4):
                    be.fill(bufs[i % nbuf], m_local, n, rank * m_local, m_total, seeds[i % nbuf])
                barrier()
                one_step(bufs[0])                       # the schedule's first call pays RCCL's set-up for its message sizes
                barrier()
                t1 = time.perf_counter()
                for i in range(1, 4):
                    one_step(bufs[i % nbuf])
                mine = {"rank": rank, "step_ms": (time.perf_counter() - t1) / 3 * 1e3}
                gs2 = be.tp.gather_stats() if mode == 1 else None
                if gs2:
                    mine.update({"gather_ms": gs2["gather_ms"], "gather_max_ms": gs2["gather_max_ms"], "call_ms": gs2["call_ms"]})
                allr = [None] * world
                dist.all_gather_object(allr, mine)
                by_schedule[name] = {"per_rank": allr, "step_ms_max": max(r["step_ms"] for r in allr)}
            be.tp.set_schedule(2)
            by_schedule["chosen_in_timed_region"] = chosen if real else "one_collective (transport %s: the factors travel through torch.distributed)" % be.transport
            by_schedule["source"] = "ncclAllGather (RCCL)" if real else "self-gather: device copies in place of the collective (transport %s)" % be.transport
            by_schedule["reserve_cus"] = os.environ.get("MI355XQR_TSQR_RESERVE_CUS", "0")
            by_schedule["note"] = ("drained latency of one factorisation per schedule and rank; gather_ms = sum over the block columns of [stacked stream "
                                   "past its wait for the local panel -> ncclAllGather done] from events on the stacked stream.  MI355XQR_TSQR_RESERVE_CUS=c "
                                   "masks the local stream to all but c compute units (RCCL's kernels then never queue behind the update); "
                                   "MI355XQR_TSQR_PIPE=0|1 pins the schedule")
        except Exception as e:
            by_schedule = {"error": repr(e)}

    # ---- N > 1, fixed-size configs (C4 / C5): the SAME matrix on ONE GPU, measured by rank 0 in this run (the other ranks wait at
    # the next barrier) -- the strong-scaling denominator next to the N-GPU numbers above
    same_1gpu = None
    if world > 1 and wl in ("c4", "c5") and rank == 0:
        try:
            one = T.DeviceTSQR(qr, m_total, n, 1, 0, 128)
            A1 = one.new_matrix(m_total, n)
            best = 1e30
            for i in range(3):
                one.fill(A1, m_total, n, 0, m_total, 12)
                t1 = time.perf_counter()
                one.factor(A1)
                one.sync()
                if i:
                    best = min(best, (time.perf_counter() - t1) * 1e3)
            one.close()
            del A1
            same_1gpu = {"ms": best, "gflops": flops(m_total, n) / (best * 1e-3) / 1e9, "nb": 128,
                         "speedup_latency": best / tsqr_split["unpipelined_latency_ms"],
                         "speedup_throughput": best / (dt / K * 1e3),
                         "note": "the whole %d x %d matrix factored on rank 0's GPU alone (plain qr_plan, drained per step); speedup = "
                                 "that time / the N-GPU time per factorisation (latency: one drained factorisation; throughput: "
                                 "ms_per_step of this line)" % (m_total, n)}
        except Exception as e:
            same_1gpu = {"error": repr(e)}

    # ---- roofline of the dominant kernel
    # dominant kernel: from the timed region; the rest: from the extra profiled step (per-step figures, K_full = 1)
    if wl in ("c2", "c3"):
        upd, tn, pan, K_pan = prof["update_nn"], prof_full["vta_tn"], prof_full["panel"], 1
    else:
        upd, tn, pan, K_pan = prof_full["update_nn"], prof_full["vta_tn"], prof["panel"], len(range(0, K, stride))
    measured = None
    if rank == 0:
        try:
            measured = qr.probe_mfma_f64_tflops()       # sustained v_mfma_f64_16x16x4_f64 rate of THIS device
        except Exception:
            measured = None
    # HBM traffic of the dominant kernel: NOT measured in this run (rocprofv3 counter passes cannot share a process with the
    # timed region, and crash on CU-masked streams on this pool).  The committed round-6 PMC pass (profiles/r06_pmc_traffic.json,
    # devtools/rounds/r5/scripts_r5_pmc.sh: the same kernel, shapes of every 8th C3 step, no CU masks) is quoted under its own key with
    # ITS algorithmic bytes, and `traffic` itself only when that file was made for this block size; otherwise null.
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")))
    except Exception:
        tj = None
    gen = 2
    kname = ("gemm_nt4_kernel<true, 0, 8, 3> (trailing update A2 -= V*Wt^T, W kept transposed; four 4-wave workgroups per CU on 128 x 64 tiles, "
             "direct-to-LDS k-tiles of 8 in three stages, v_mfma_f64_16x16x4_f64; MI355XQR_NT4=0: the 8-wave gemm_nt_kernel<true, 0, 1>)")
    if wl in ("c2", "c3") and upd["launches"]:
        ach = upd["flops"] / (upd["ms"] * 1e-3) / 1e12
        if tj and wl == "c3" and nb == tj.get("nb") and gen == 2 and "gemm_nt_kernel" in tj:
            e = tj["gemm_nt_kernel"]                    # (key kept from round 3: the entry is the update kernel of the pass, now gemm_nt4_kernel)
            traffic = e["hbm_bytes_per_launch"]
            traffic_src = {"file": "profiles/r06_pmc_traffic.json", "git_head_of_pass": tj.get("git_head"),
                           "launch_mix": tj.get("config"),
                           "algorithmic_bytes_per_launch_same_mix": e["algorithmic_bytes_per_launch"],
                           "ratio_traffic_to_algorithmic": e["ratio"],
                           "method": "2*FETCH_SIZE + WRITE_SIZE per dispatch (gfx950 correction, MI355X_MICROARCH.md HBM section), "
                                     "separate rocprofv3 --pmc passes; replayed from the committed file, not measured in this run",
                           "note": "x1.55: the write-allocated C tiles push the operand tiles of the 128 x 64 tiling (3.75 MB per XCD at a time) out of the 4 MB L2 "
                                   "and they are fetched again -- from the 256 MB Infinity Cache, which FETCH_SIZE (L2 misses) cannot tell from HBM; with "
                                   "non-temporal C accesses the refetches go away (-40 %) and nothing gets faster (profiles/r05_nt_ceiling.txt)"}
        # the committed rocprofv3 --kernel-trace --stats summary of this bench command: the kernel's average duration there, next to the
        # HIP-event figure of this run (an event bracket also holds two barrier packets and the dispatch latency: ~15 us per launch)
        rocprof_ref = None
        try:
            import csv
            for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06_bench_c3_kernel_stats.csv"))):
                if wl == "c3" and row["Name"].startswith("void gemm_nt4_kernel<true, 0, 8, 3"):      # (<.., RAG = false>: the aligned instantiation, the only one C3 launches)
                    ns = float(row["AverageNs"])
                    rocprof_ref = {"file": "profiles/r06_bench_c3_kernel_stats.csv", "calls": int(row["Calls"]), "avg_launch_ms": ns * 1e-6,
                                   "achieved_at_that_duration": (upd["flops"] / upd["launches"]) / (ns * 1e-9) / 1e12,
                                   "note": "same command under rocprofv3 (committed); the launch mix per step is the one timed here"}
        except Exception:
            rocprof_ref = None
        cus_u = None
        try:
            cus_u = int(be.plan.update_cus())
        except Exception:
            pass
        roof = {"bound": "mfma",
                "kernel": kname,
                "achieved": ach, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MATRIX_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "peak_source": "AMD MI355X datasheet fp64 matrix 78.6 TFLOP/s (MI355X_MICROARCH.md has no fp64 MFMA row)",
                # back-to-back independent v_mfma_f64_16x16x4_f64 (16 accumulators per wave, inline asm) on all CUs
                "measured_mfma_f64_sustained_tflops": measured["mfma_f64_tflops"] if measured else None,
                "frac_of_measured_sustained": ach / measured["mfma_f64_tflops"] if measured else None,
                "cu_partition_note": ("with look-ahead the wide update runs on the update stream's compute units "
                                      "(16384^2: 224 of 256, the panel chain owns 32 = 4 compute units of every XCD; smaller problems 192 / 64)"),
                "update_stream_cus": cus_u,
                # the same rate against the peak of the compute units the kernel actually runs on (`frac` above stays against the whole
                # chip, as the contract asks): what the kernel itself leaves on the table, apart from the schedule's CU partition
                "cus": cus_u,
                "frac_of_cus_used": (ach / (FP64_MATRIX_PEAK_TFLOPS * cus_u / float(qr.device_info()["compute_units"] or 256))) if cus_u else None,
                "rocprof_pmc": "profiles/r06_pmc_mfma_lds_util.txt (MfmaUtil, LdsUtil, LdsBankConflict of the same kernel, whole chip: rocprofv3 --pmc crashes on CU-masked streams on this pool)",
                "ceiling": "profiles/r05_nt_ceiling.txt (same launch mix with the C traffic / the operand loads compiled out: 58.9 / 61.4 TFLOP/s in situ bound any K = 256 kernel on 224 CUs)",
                "measured_probe": measured,
                "rocprofv3_same_kernel": rocprof_ref,
                "launches": upd["launches"], "avg_launch_ms": upd["ms"] / upd["launches"],
                "profiled_steps_in_timed_region": len(range(0, K, stride)),
                "algorithmic_flops_per_launch": upd["flops"] / upd["launches"],
                "algorithmic_bytes_per_launch": upd["bytes"] / upd["launches"],
                # the OTHER kernel of the update pair in the same shape: Wt = A2^T (V T), from the extra profiled step
                "companion_tn": (lambda a: {"kernel": "gemm_tn_kernel<4, 4, true, 1> + slab_reduce_kernel (Wt = A2^T (V T); split-K slabs summed in a fixed order)",
                                            "bound": "mfma", "achieved": a, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": (a / FP64_MATRIX_PEAK_TFLOPS) if a else None,
                                            "traffic": (tj or {}).get("gemm_tn_kernel<4,4,true,1>", {}).get("hbm_bytes_per_launch") if (tj and wl == "c3" and nb == tj.get("nb")) else None,
                                            "traffic_ratio_to_algorithmic": (tj or {}).get("gemm_tn_kernel<4,4,true,1>", {}).get("ratio") if (tj and wl == "c3" and nb == tj.get("nb")) else None,
                                            "traffic_note": "x2.28 of algorithmic: A2 is read once per 128-column tile of V T (N / 128 = 2 times at nb 256); the 128 x 256-tile form "
                                                            "(x1.25, lab knob MI355XQR_TN_WIDE) was re-measured this round at the final code and LOSES 3 ms at C3 "
                                                            "(profiles/r06_tn_wide_ab.txt): the product is matrix-core bound, the refetches come from the Infinity Cache",
                                            "launches": tn["launches"], "avg_launch_ms": tn["ms"] / tn["launches"] if tn["launches"] else None})(
                                     tn["flops"] / (tn["ms"] * 1e-3) / 1e12 if tn["ms"] else None),
                "panel_ms_per_step": pan["ms"] / K_pan,
                "other_classes_note": "wide_product_tn and panel_ms_per_step come from one extra, fully profiled factorisation after "
                                      "the timed region; inside it only the dominant kernel's launches carry HIP events"}
    else:
        # tall-skinny: the panel (TSQR leaf kernels + in-panel updates) is the dominant cost; its compulsory HBM
        # traffic is 16 * mk * w bytes per panel (read + write once)
        ach = pan["bytes"] / (pan["ms"] * 1e-3) / 1e9 if pan["ms"] else 0.0
        # measured HBM bytes of the leaf's three streaming kernels (PMC pass of the leaf entry point on a 262144 x 32 leaf,
        # profiles/r03_pmc_panel_hbm.json), replayed -- not measured in this run; a "launch" here is one outer panel = nb/32 leaves
        # plus their in-panel updates, whose bytes are not in the counter file
        ptraffic, psrc = None, None
        try:
            # round 4: a 262144-row, 128-column panel takes the full-width route (qr_panel_cqr.hip): PMC passes over one such panel
            pj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_panel_hbm.json")))
            if m_local == pj["mk"] and nb == pj["w"]:
                ptraffic = pj["hbm_bytes_per_panel"]
                psrc = {"file": "profiles/r06_pmc_panel_hbm.json", "covers": pj["covers"], "method": pj["method"] + "; replayed from the committed file"}
        except Exception:
            pass
        # whole-factorisation HBM bytes of this shape (every dispatch: leaf kernels, in-panel products and updates, outer updates),
        # PMC passes of devtools/rounds/r4/scripts_r4_pmc.sh -- replayed, not measured in this run
        whole = None
        try:
            wj = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_tsqr_total_traffic.json")))
            if m_local == wj["m"] and n == wj["n"]:
                whole = {"file": "profiles/r06_pmc_tsqr_total_traffic.json", "hbm_bytes_per_factorisation": wj["hbm_bytes_per_factorisation"],
                         "algorithmic_bytes_16mn": wj["algorithmic_bytes_16mn"], "ratio": wj["ratio"],
                         "achieved_GBps_whole_step": wj["hbm_bytes_per_factorisation"] / (dt / K) / 1e9}
        except Exception:
            pass
        roof = None if not pan["launches"] else {"bound": "hbm", "kernel": "panel factorisation: 128-column panels of > 8192 rows at full width (qr_panel_cqr.hip: cqr_gram / cqr_chol / cqr_stream<Q,G2> / cqr_lu / cqr_vpass kernels; round 4: from 196608 rows), up to 8192 rows in one launch (panel_fused_kernel), others by the leaf chain",
                "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "traffic": ptraffic, "traffic_source": psrc, "traffic_whole_factorisation": whole,
                "launches": pan["launches"], "avg_launch_ms": pan["ms"] / max(pan["launches"], 1),
                "profiled_steps_in_timed_region": len(range(0, K, stride)),
                "algorithmic_bytes_per_launch": pan["bytes"] / max(pan["launches"], 1),
                "note": "algorithmic bytes = 16*mk*nb per outer panel (read + write once); a full-width panel moves 2.6x that in three passes "
                        "(G1; Q + G2; V -- to one destination when the panel is parked; 3.15x with two, the stand-alone panel of the PMC file) -- its passes cost memory time plus "
                        "matrix-core time, and two one-workgroup factor kernels (0.12 ms in round 5, 0.27 in round 4) sit between them "
                        "(DESIGN.md section 3)",
                "update_nn_tflops": upd["flops"] / (upd["ms"] * 1e-3) / 1e12 if upd["ms"] else None,
                "measured_probe": measured}

    # which panel routes the timed plan took, and whether a device-side guard fired (include/mi355x_qr.h: qr_plan_route_stats)
    try:
        panel_routes = be.plan.route_stats()
    except Exception:
        panel_routes = None

    # ---- N = 1 only: the one-GPU leg of the multi-GPU (TSQR, weak scaling) series, so that the N > 1 lines of this
    # bench (262144 x 512 per GPU) have their own denominator next to the C3 headline
    weak_base, tsqr_model = None, None
    if world == 1 and wl == "c3":
        bufs.clear()
        be.close()
        be = T.DeviceTSQR(qr, 262144, 512, 1, 0, 128)
        reps = max(2, min(K, 5))
        tb = [be.new_matrix(262144, 512) for _ in range(reps)]
        for i, A in enumerate(tb):
            be.fill(A, 262144, 512, 0, 262144, 12 + i)
        be.factor(tb[0]); be.sync(); be.fill(tb[0], 262144, 512, 0, 262144, 12)      # warm-up, then restore the input
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            be.factor(tb[i])
        be.sync()
        dtb = (time.perf_counter() - t0) / reps
        weak_base = {"workload": "one 262144x512 TSQR shard on 1 GPU (the per-GPU work of the N>1 lines of this bench)",
                     "value": flops(262144, 512) / dtb / 1e9, "unit": "GFLOP/s", "ms_per_step": dtb * 1e3, "steps": reps}
        tb.clear()
        # the whole step of one rank of a multi-GPU factorisation, measured on this one GPU (every step drained before the next):
        # the local QR, and local QR + exchange + stacked QR with the collective replaced by device copies of the rank's own factor
        # -- the launches, streams and events of a real rank, which also factors the full stacked matrix redundantly, minus the network
        try:
            c5 = T.rank_step_latency(qr, 262144, 512, 8, 128)
            c4 = T.rank_step_latency(qr, 65536, 256, 4, 128)
            c4_2 = T.rank_step_latency(qr, 131072, 256, 2, 128)

            def whole_ms(m, ncol, nbw, reps_=2):
                # the strong-scaling denominator: the WHOLE matrix of the multi-GPU config factored on this one GPU
                b1 = T.DeviceTSQR(qr, m, ncol, 1, 0, nbw)
                Aw = b1.new_matrix(m, ncol)
                best = None
                for i in range(reps_ + 1):
                    b1.fill(Aw, m, ncol, 0, m, 12 + i)
                    b1.sync()
                    t0_ = time.perf_counter()
                    b1.factor(Aw)
                    b1.sync()
                    d_ = (time.perf_counter() - t0_) * 1e3
                    if i > 0:
                        best = d_ if best is None else min(best, d_)
                b1.close()
                del Aw
                torch.cuda.empty_cache()
                return best

            NET_MS = 0.1                  # ASSUMPTION, never measured: what the n/nb small all-gathers of a step cost on xGMI
            c5_whole, c4_whole = whole_ms(2097152, 512, 128), whole_ms(262144, 256, 128)
            tsqr_model = {"c5_rank_of_8": c5, "c4_rank_of_4": c4, "c4_rank_of_2": c4_2,
                          "stacked_step_alone_4096x512_ms": T.stacked_step_ms(qr, 8, 512, 128),
                          "assumed_network_ms_per_step": NET_MS,
                          "c5_whole_matrix_1gpu_ms": c5_whole, "c4_whole_matrix_1gpu_ms": c4_whole,
                          # STRONG scaling, the figure BASELINE's C4 / C5 configs ask for: the same whole matrix on 1 GPU over one rank's step
                          "predicted_c5_speedup_8gpu": c5_whole / (c5["step_ms"] + NET_MS),
                          "predicted_c5_strong_efficiency_8gpu": c5_whole / (c5["step_ms"] + NET_MS) / 8.0,
                          "predicted_c4_speedup_2gpu": c4_whole / (c4_2["step_ms"] + NET_MS),
                          "predicted_c4_speedup_4gpu": c4_whole / (c4["step_ms"] + NET_MS),
                          # weak-scaling reading of the same measurements (per-GPU work fixed): how much of a rank's step is its local QR
                          "predicted_c5_local_share_of_step": c5["local_ms"] / (c5["step_ms"] + NET_MS),
                          "note": "single-GPU measurements, one factorisation's latency (every step drained before the next).  A rank's step "
                                  "= local QR + panel-pipelined exchange (device copies in place of the all-gathers) + redundant stacked QR; "
                                  "the network term is the ASSUMED constant above.  predicted_*_speedup = whole matrix on ONE GPU / (rank "
                                  "step + network): the strong-scaling prediction; *_local_share_of_step is not a scaling efficiency"}
        except Exception as e:
            tsqr_model = {"error": repr(e)}

    line = None
    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample)
        info = qr.device_info()
        line = {
            "metric": "fp64 GFLOP/s (% of roofline) + ||A-QR||_F/||A||_F, m x n QR at 1/2/4/8 GPUs",
            "value": value, "unit": "GFLOP/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "m": m_total, "n": n, "m_per_gpu": m_local, "nb": nb,
                       "ib": args.ib or qr.get_block_size()[1],
                       "flops_per_step": flops(m_total, n), "input_buffers": nbuf,
                       "input": ("uniform[0,1) counter-hash generator, seed 12+i, resident in HBM" if args.cond <= 0.0 else
                                 "uniform[0,1) counter-hash generator, seed 12+i, then every 128-column panel made ill-conditioned (columns = the panel's "
                                 "first column + noise / %g): the guard refuses every full-width panel; resident in HBM" % args.cond),
                       "cond": args.cond or None,
                       "collective": "none" if world == 1 else (f"{n // nb} all_gathers of n*nb doubles per rank (panel-pipelined, RCCL)" if (backend == "nccl" and be.transport == "rccl" and be.tp.is_pipelined()) else f"1 all_gather of n*n doubles per rank ({'RCCL' if backend == 'nccl' else backend + ' via host, bring-up only'})")},
            "frac_of_fp64_matrix_peak": value / 1e3 / (FP64_MATRIX_PEAK_TFLOPS * world),
            "accuracy": acc,
            "roofline": roof,
            "panel_routes": panel_routes,
            "weak_scaling_base_1gpu": weak_base,
            "tsqr_model_1gpu": tsqr_model,
            "tsqr_step_split": tsqr_split,
            "tsqr_exchange_by_schedule": by_schedule,
            "same_problem_1gpu": same_1gpu,
            "rccl": ({"nranks_seen_by_rccl": rccl_ranks, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                      "transport": be.transport, "fallback_reason": be.fallback_reason,
                      "driver": ("C-ABI qr_tsqr_plan: ncclCommInitRank from a broadcast unique id, ncclAllGather on the plan's stream"
                                 if be.transport == "rccl" else
                                 "C-ABI qr_tsqr_plan for the local and stacked steps; R factors gathered by torch.distributed on the plan's "
                                 "device buffers (fallback: the library could not create its own communicator)")}
                     if world > 1 else None),
            "scaling_note": ("the N = 1 line of this bench is the square C3 headline, a different workload: the weak-scaling "
                             "denominator of this line is weak_scaling_base_1gpu of the N = 1 line (one 262144x512 shard on "
                             "one GPU)") if world > 1 else None,
            "cpu_baseline": cpu,
            "device": info,
        }
        print(json.dumps(line), flush=True)
    wd.arm(args.watchdog + 30.0, "teardown")
    be.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    wd.disarm()


if __name__ == "__main__":
    main()
