#!/usr/bin/env python3
"""bench.py -- pressure-Poisson V-cycles/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE config C2 -- single-level 512^3 Poisson (alpha=0, beta=1), homogeneous
Neumann on all faces, "mapped-Cartesian" = separable stretched DIAGONAL metric stored as full coefficient
arrays, LevelGSRB smoother, pre/post/bottom = 2/2/2 (utils/ProblemContext.cpp:1153-1163),
BiCGStab bottom solver.  A step = one MappedMultiGrid V-cycle (cycle(0)) from a zero correction on a fixed
compatible residual, with every input already resident in HBM.  N > 1: the SAME 512^3 problem, its box
layout sharded over N GPUs (one box per GPU, one process per GPU, halo exchange + scalar reductions over
RCCL/xGMI) => strong scaling.

One JSON line on rank 0.  Besides the contract keys it carries
  roofline     -- dominant kernel (k_gsrb_fused, depth 0; k_gsrb_ortho below the fused threshold): algorithmic bytes (32 B/cell per colour pass =
                  half of the 64 B/cell red+black sweep of SURVEY.md 8d) / HIP-event launch duration measured
                  in the timed region on the solver's stream, against the 8 TB/s HBM3E peak
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's Fortran+C++ path; the reference itself
                  cannot be built here) timed on a bounded 256^3 sample of the same workload, 1 core
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_GSRB_COLOR = 32.0            # algorithmic B/cell of ONE colour pass (64 B/cell per red+black sweep)
B_RESIDUAL = 56.0              # algorithmic B/cell of the fused residual
N_FINE = 512


def cpu_baseline(sample_n=256):
    """oracle V-cycle on a bounded sample, scaled to V-cycles/s at 512^3 (work is linear in cells)."""
    import ctypes as C
    import subprocess
    from oracle import somar_oracle as so
    from helpers import make_oracle_solver, make_problem
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"])
    so._LIB = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))  # -O3 -march=native build of the same source
    dom, grids, dx, Jgup, Jinv = make_problem(so, sample_n, sample_n, "stretched")
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    res = so.random_field(grids, 12345, (0, 0, 0), dom.box)
    so.remove_weighted_mean(res, Jinv)
    corr = so.LevelData(grids, 1, (1, 1, 1))
    amr.mg.init(corr, res)
    amr.mg.one_cycle(corr, res)  # warm-up
    reps = 12   # ~0.6 s each on one core: with set-up and warm-up about 10-15 s of CPU work
    t0 = time.perf_counter()
    for _ in range(reps):
        so.ld_set(corr, 0.0)
        amr.mg.one_cycle(corr, res)
    dt = (time.perf_counter() - t0) / reps
    scale = (N_FINE / sample_n) ** 3
    return {"value": 1.0 / (dt * scale), "unit": "V-cycles/s", "cores": 1, "kind": "port",
            "sample": "%d^3 single-box V-cycle (2/2/2, same metric/BCs), %d reps, %.2f s each, scaled by (512/%d)^3; "
                      "oracle = C restatement of the reference Fortran kernels (gcc -O3 -march=native) driven by "
                      "numpy orchestration" % (sample_n, reps, dt, sample_n)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--size", dest="n", type=int, default=N_FINE, help="fine grid size per direction (512 = BASELINE C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from somar_amd import api, synthetic   # the measured path never touches oracle/ (only cpu_baseline() below does)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)" % (args.gpus, world))
    # SOMAR_BENCH_COMM=shm: rehearsal of the N > 1 path on a box with ONE GPU (all ranks on device 0, host-staged
    # shared-memory transport, csrc/comm_shm.cpp).  Not a performance mode; the driver's runs use RCCL.
    use_shm = os.environ.get("SOMAR_BENCH_COMM") == "shm"
    torch.cuda.set_device(0 if use_shm else local_rank)
    comm = None
    dist = None
    if world > 1:
        import torch.distributed as dist
        # control plane only (id broadcast, barrier, max over ranks); the data path is RCCL inside the library
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if use_shm:
            names = ["/somar_bench_%d" % os.getpid() if rank == 0 else None]
            dist.broadcast_object_list(names, src=0)
            comm = api.comm_create_shm(names[0], rank, world, 256 << 20)
        else:
            ids = [api.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = api.comm_create(ids[0], rank, world, local_rank)

    n = args.n
    L = (1.0, 1.0, 1.0)
    dx = tuple(L[d] / n for d in range(3))
    boxes = synthetic.slab_partition(n, world)
    owner = list(range(world))
    t_def = -time.perf_counter()
    gpu = api.AMRPressureSolver()
    p = gpu._p
    gpu.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg,
                           p.hang, p.norm_thresh, 0)
    gpu.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, boxes, owner=owner, comm=comm)
    for q in range(gpu.num_local_patches):
        lo, hi, gi = gpu.patch_box(q)
        assert gi == rank
        # each rank evaluates the metric of ITS box only (host numpy; not part of define_seconds)
        t_def += time.perf_counter()
        jg, jinv = synthetic.stretched_diagonal_metric(lo, hi, dx, L)
        t_def -= time.perf_counter()
        gpu.setMetricOrtho(q, jg[0], jg[1], jg[2], jinv)
        del jg, jinv
    gpu.finalize()
    t_def += time.perf_counter()
    F = api
    depth = gpu.depth()
    cells_total = n ** 3
    cells_local = cells_total // world

    # the residual of a solve from phi = 0: uniform(-1,1) (seed 12345) minus its J-weighted mean
    gpu.fillHash(F.F_RES, 12345)
    gpu.removeMean(F.F_RES)

    def barrier():
        gpu.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def step():
        # one V-cycle from a zero correction, as every iteration of MappedAMRMultiGrid::solveNoInitResid runs it
        gpu.vcycleFromZero(F.F_CORR, F.F_RES)

    for _ in range(args.warmup):
        step()
    barrier()
    gpu.profileEnable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gpu.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    n_gsrb, ms_gsrb = gpu.profileGet(0)
    n_rr, ms_rr = gpu.profileGet(1)   # depth-0 residual launches of the V-cycle = fused residual + restriction
    # the plain residual (north-star unit "one residual + one red+black sweep"), timed outside the V-cycle timing
    for _ in range(5):
        gpu.residual(0, F.F_SCRATCH, F.F_CORR, F.F_RES)
    n_op, ms_op = gpu.profileGet(1)
    gpu.profileEnable(False)

    # contraction check of the timed operator (not timed): one more cycle must reduce the residual
    r0 = gpu.norm(F.F_RES, 0)
    gpu.setVal(F.F_PHI, 0.0)
    gpu.setVal(F.F_CORR, 0.0)
    gpu.vcycle(F.F_CORR, F.F_RES)
    gpu.residual(0, F.F_SCRATCH, F.F_CORR, F.F_RES)
    r1 = gpu.norm(F.F_SCRATCH, 0)

    ms_per_step = 1e3 * dt / args.steps
    value = args.steps / dt
    # depth 0 runs the fused red+black sweep (one launch = one sweep = 64 B/cell algorithmic) unless the
    # level is below SOMAR_FUSED_MIN_CELLS, in which case one launch = one colour pass = 32 B/cell
    fused = cells_total >= int(os.environ.get("SOMAR_FUSED_MIN_CELLS", "262144"))
    kname = "k_gsrb_fused (red+black sweep, depth 0)" if fused else "k_gsrb_ortho (one colour pass, depth 0)"
    b_launch = (2.0 if fused else 1.0) * B_GSRB_COLOR
    t_gsrb = ms_gsrb / max(n_gsrb, 1) * 1e-3      # s per GSRB launch (depth 0)
    t_op = ms_op / max(n_op, 1) * 1e-3
    achieved = b_launch * cells_local / t_gsrb / 1e9
    t_sweep = t_gsrb if fused else 2 * t_gsrb
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and n == N_FINE and world == 1:
        traffic = json.load(open(tf)).get("k_gsrb_fused_bytes_per_launch" if fused else "k_gsrb_ortho_bytes_per_launch")
    unit_t = t_sweep + t_op                       # one red+black sweep + one residual (north-star unit)
    out = {
        "metric": "pressure-Poisson V-cycles/sec", "value": value, "unit": "V-cycles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2 single-level %d^3 Poisson, diagonal stretched metric, Neumann, LevelGSRB V-cycle "
                               "2/2/2 + BiCGStab bottom; layout = %d box(es) %s, one per GPU"
                               % (n, world, "x".join(str(h - l + 1) for l, h in zip(*boxes[0]))),
                   "mg_depth": depth, "mg_ref_ratios": [list(r) for r in gpu.mgRefRatios()],
                   "cells": cells_total, "define_seconds": t_def},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": b_launch * cells_local, "launches": n_gsrb,
                     "avg_launch_ms": t_gsrb * 1e3},
        "gsrb_cell_updates_per_s": cells_local / t_sweep * world,
        "residual_kernel": {"avg_launch_ms": t_op * 1e3, "achieved_GBs": B_RESIDUAL * cells_local / t_op / 1e9,
                            "frac": B_RESIDUAL * cells_local / t_op / 1e9 / HBM_PEAK_GBS},
        "residual_restrict_kernel": {"avg_launch_ms": ms_rr / max(n_rr, 1), "launches": n_rr,
                                     "note": "k_resid_march<2>: residual + J-weighted restriction in one pass, "
                                             "49 B/cell algorithmic",
                                     "achieved_GBs": 49.0 * cells_local / (ms_rr / max(n_rr, 1) * 1e-3) / 1e9},
        "residual_plus_smooth_unit": {"ms": unit_t * 1e3, "algorithmic_GBs": 120.0 * cells_local / unit_t / 1e9,
                                      "frac_of_hbm_peak": 120.0 * cells_local / unit_t / 1e9 / HBM_PEAK_GBS},
        "vcycle_contraction": r1 / r0,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        out["cpu_baseline"] = None
    gpu.undefine()
    if comm is not None:
        api.comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
