#!/usr/bin/env python3
"""bench.py -- pressure-Poisson V-cycles/s on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline workload (config.workload): BASELINE config C2 -- single-level 512^3 Poisson (alpha=0, beta=1), homogeneous
Neumann on all faces, "mapped-Cartesian" = separable stretched DIAGONAL metric stored as full coefficient
arrays (SURVEY.md 8d variant ii), LevelGSRB smoother, pre/post/bottom = 2/2/2 (utils/ProblemContext.cpp:1153-1163),
BiCGStab bottom solver.  A step = one MappedMultiGrid V-cycle (cycle(0)) from a zero correction on a fixed
compatible residual, with every input already resident in HBM.  N > 1: the SAME 512^3 problem, its box
layout sharded over N GPUs (one box per GPU, one process per GPU, halo exchange + scalar reductions over
RCCL/xGMI) => strong scaling.  The timed loop runs the shipped code path (HIP-graph replay of the coarse depths,
no per-kernel events); the per-kernel HIP-event numbers of the `roofline` block come from a separate, untimed,
profiled pass of the same V-cycle.

One JSON line on rank 0.  Besides the contract keys it carries
  roofline      dominant kernel (k_gsrb_fused, depth 0): algorithmic bytes (64 B/cell per red+black sweep)
                / HIP-event launch duration on the solver's stream, against the 8 TB/s HBM3E peak; `measured_copy_GBs`
                = device-to-device copy bandwidth measured in this run (read + written bytes / time);
                `measured_stream_mix_GBs` = a bare kernel with the sweep's own stream mix (6 read streams + 1 write
                stream of 512^3 doubles, no stencil, no halo; somar_diag_stream_probe) measured in this run
  cpu_baseline  the reference's V-cycle call sequence orchestrated in C over the restated Fortran kernels
                (oracle/cpu_vcycle.c; the reference itself cannot be built here), OpenMP over k-slabs, timed on the
                REAL 512^3 problem on all host cores given to this process and on 1 core, with the measured STREAM triad
  c2_cartesian  SURVEY.md 8d variant (i): the same V-cycle with the all-ones (true Cartesian) metric -- the variant on
                which point-GSRB multigrid converges at 0.07-0.1 per cycle (on the stretched metric the cycle stalls at
                512^3, in the oracle too: DESIGN.md)
  c4_amr        BASELINE config C4, the workload the north star's >= 6x 1->8-GPU scaling is quoted on: 3-level
                1024x1024x128 LockExchange hierarchy ((2,2,1) refinements, 128^3 boxes, 940 M cells), every level's boxes
                sharded in y-slabs over the N GPUs, ms per AMR V-cycle (MappedAMRMultiGrid::AMRVCycle, 4/4/2), measured in
                the same run at every N
The run FAILS (non-zero exit) when a timed operator does not contract, when the one-GPU contraction departs from the
recorded value, or when an N > 1 run does not reproduce (to 1e-6) the contraction of the SAME N-box layout held by one
process, which rank 0 computes next to its shard: a broken exchange must not produce a plausible throughput line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_GSRB_COLOR = 32.0            # algorithmic B/cell of ONE colour pass (64 B/cell per red+black sweep)
B_RESIDUAL = 56.0              # algorithmic B/cell of the fused residual
N_FINE = 512
# single-GPU first-cycle contraction |r1|/|r0| of the timed problems (driver BENCH_r01 / profiles/): the N > 1 runs
# must reproduce them (same arithmetic; only the large-level mean sums associate differently)
# (the residual is std::mt19937_64(12345) since round 3; the device-side hash field of rounds 1-2 gave 0.460938)
SINGLE_RANK_CONTRACTION = {("stretched", 512): 0.49259591046055523}
# one AMR V-cycle of the full-size C4 / C5 hierarchies on their compatible composite residual (profiles/r03_*): the box layout
# -- hence the hierarchy and the arithmetic -- is the same at every N, only owners change
AMR_CONTRACTION = {"c4": 0.0244679, "c5": 0.0033342}


def host_cores():
    """the CPU threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands one
    GPU's share of the host -- 16 of 256 hardware threads -- through cpu.max, not through the affinity mask)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("SOMAR_CPU_THREADS")
    return max(1, int(env)) if env else n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def c2_random_field(n):
    """BASELINE C2's residual before its mean is removed: std::mt19937_64(12345), uniform(-1, 1), drawn in Fortran order over
    the n^3 domain (SURVEY.md 8d / BASELINE.md 4).  Generated ONCE on the host; the same array is uploaded to the GPU
    (before anything is timed) and handed to the CPU baseline."""
    from somar_amd import api
    return api.host_random_field((n, n, n), 12345)


def upload_c2_residual(gpu, F, field):
    """the local boxes' part of `field` -> F_RES, then the J-weighted mean is removed on the device"""
    import numpy as np
    for q in range(gpu.num_local_patches):
        lo, hi, _ = gpu.patch_box(q)
        gpu.upload(F.F_RES, q, np.asfortranarray(field[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]), (0, 0, 0))
    gpu.removeMean(F.F_RES)


def cpu_baseline(n=N_FINE, field=None):
    """The reference's V-cycle on the host: C orchestration + restated Fortran kernels + OpenMP (oracle/cpu_vcycle.c),
    the real n^3 problem on the SAME random residual as the GPU run.  Bounded: 1 timed cycle on one core, 3 on all cores
    (~20-40 s of CPU work at 512^3)."""
    import numpy as np
    from oracle import cpu_vcycle as cv
    from somar_amd import synthetic
    cores = host_cores()
    dx = (1.0 / n,) * 3
    jg, jinv = synthetic.stretched_diagonal_metric((0, 0, 0), (n - 1,) * 3, dx, (1.0, 1.0, 1.0))
    res = np.array(field if field is not None else c2_random_field(n), order="F")
    res -= float((res / jinv).sum() / (1.0 / jinv).sum())
    corr = np.zeros((n + 2,) * 3, order="F")
    h = cv.CpuVCycle((n,) * 3, dx, jg, jinv, pre=2, post=2, bottom=2, nthreads=cores)
    h.vcycle(corr, res)   # warm-up on all cores (first touch of every array)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        h.vcycle(corr, res)
    t_all = (time.perf_counter() - t0) / reps
    # per-kernel rates on all cores: one red+black sweep, one residual
    phi = np.zeros((n + 2,) * 3, order="F")
    out = np.zeros((n,) * 3, order="F")
    t0 = time.perf_counter()
    h.relax(phi, res, 1)
    t_sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    h.residual(out, phi, res)
    t_resid = time.perf_counter() - t0
    h.set_threads(1)
    t0 = time.perf_counter()
    h.vcycle(corr, res)
    t_one = time.perf_counter() - t0
    h.close()
    triad_all = cv.triad_gbs(cores)
    triad_one = cv.triad_gbs(1)
    return {"value": 1.0 / t_all, "unit": "V-cycles/s", "cores": cores, "kind": "port",
            "one_core_value": 1.0 / t_one, "cpu_model": cpu_model(),
            "stream_triad_GBs": {"all_cores": triad_all, "one_core": triad_one},
            "gsrb_cell_updates_per_s": n ** 3 / t_sweep, "residual_cells_per_s": n ** 3 / t_resid,
            "sample": "the full %d^3 C2 V-cycle (2/2/2, same stretched metric / Neumann BCs / hierarchy depth): %d timed "
                      "cycles on %d cores (%.2f s each) after one warm-up, 1 timed cycle on 1 core (%.2f s); reference call "
                      "sequence orchestrated in C (oracle/cpu_vcycle.c: flux temporaries, ghost fill per colour pass, 26 "
                      "boundary sub-box calls, BiCGStab bottom) over the restated Fortran kernels, gcc -O3 -march=native, "
                      "OpenMP over k-slabs standing in for one MPI rank per core" % (n, reps, cores, t_all, t_one)}


def measured_copy_gbs(torch, nbytes=1 << 30, reps=10):
    """device-to-device copy bandwidth (bytes read + bytes written per second) -- the streaming ceiling of this GPU"""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda")
    b = torch.ones(nbytes // 8, dtype=torch.float64, device="cuda")
    for _ in range(3):
        a.copy_(b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        a.copy_(b)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def build_c2(api, synthetic, n, world, rank, comm, variant, all_on_this_rank=False):
    """all_on_this_rank: the SAME box layout (one box per rank of a `world`-rank job) held by one process without a
    communicator -- the single-process twin a sharded run is checked against"""
    import numpy as np
    L = (1.0, 1.0, 1.0)
    dx = tuple(L[d] / n for d in range(3))
    boxes = synthetic.slab_partition(n, world)
    t_def = -time.perf_counter()
    gpu = api.AMRPressureSolver()
    p = gpu._p
    gpu.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg,
                           p.hang, p.norm_thresh, 0)
    if all_on_this_rank:
        gpu.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, boxes)
    else:
        gpu.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, boxes, owner=list(range(world)), comm=comm)
    for q in range(gpu.num_local_patches):
        lo, hi, gi = gpu.patch_box(q)
        assert all_on_this_rank or gi == rank
        t_def += time.perf_counter()    # host-side metric evaluation (numpy) is the caller's, not define's
        if variant == "stretched":
            jg, jinv = synthetic.stretched_diagonal_metric(lo, hi, dx, L)
        else:
            shp = [h - a + 1 for a, h in zip(lo, hi)]
            jg = [np.ones((shp[0] + (d == 0), shp[1] + (d == 1), shp[2] + (d == 2)), order="F") for d in range(3)]
            jinv = np.ones(shp, order="F")
        t_def -= time.perf_counter()
        gpu.setMetricOrtho(q, jg[0], jg[1], jg[2], jinv)
        del jg, jinv
    gpu.finalize()
    t_def += time.perf_counter()
    return gpu, boxes, t_def


def time_c2(gpu, F, torch, dist, steps, warmup):
    """-> (seconds for `steps` V-cycles, max over ranks)"""
    def barrier():
        gpu.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(warmup):
        gpu.vcycleFromZero(F.F_CORR, F.F_RES)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        # one V-cycle from a zero correction, as every iteration of MappedAMRMultiGrid::solveNoInitResid runs it
        gpu.vcycleFromZero(F.F_CORR, F.F_RES)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    return dt


def contraction_c2(gpu, F):
    """|rhs - L[corr]| / |rhs| after ONE V-cycle from zero (not timed)"""
    r0 = gpu.norm(F.F_RES, 0)
    gpu.setVal(F.F_PHI, 0.0)
    gpu.setVal(F.F_CORR, 0.0)
    gpu.vcycle(F.F_CORR, F.F_RES)
    gpu.residual(0, F.F_SCRATCH, F.F_CORR, F.F_RES)
    return gpu.norm(F.F_SCRATCH, 0) / r0


def bench_c4(api, torch, dist, comm, world, steps, warmup, scale, config="c4"):
    """BASELINE C4 (config "c4"): ms per AMR V-cycle on the 3-level 1024x1024x128 hierarchy, levels sharded in y-slabs;
    config "c5": BASELINE C5's shape, 4 levels of 512x512x64 cells each, terrain-following NON-diagonal metric produced on
    the device from the nodal depth (BathymetricBaseMap's own discretisation, somar_solver_set_metric_map)"""
    from bench_amr import build_hierarchy
    F = api
    gpu, levels, cells_local, t_def, dx0, ratios = build_hierarchy(config, scale, 128, comm=comm, nranks=world)
    nlev = len(levels)
    try:
        # a COMPATIBLE composite residual (the hierarchy is all-Neumann / periodic: L has the constants in its null space):
        # RES := 0 - L_composite[hash-random phi], covered coarse cells zeroed -- what a solve from phi = 0 would hand the cycle
        for l, v in enumerate(gpu.levels):
            v.fillHash(F.F_PHI, 12345 + l)
            v.setVal(F.F_RHS, 0.0)
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_RES)

        def step():
            for v in gpu.levels:
                v.setVal(F.F_CORR, 0.0)
            gpu.vcycleAMR(nlev - 1, 0)

        def barrier():
            gpu.levels[0].sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        # composite residual after the last cycle (CORR holds its correction): max over levels, covered cells zeroed
        r0 = max(v.norm(F.F_RES, 0) for v in gpu.levels)
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev, res_field=F.F_SCRATCH, phi_field=F.F_CORR, rhs_field=F.F_RES)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_SCRATCH)
        r1 = max(v.norm(F.F_SCRATCH, 0) for v in gpu.levels)
        s = scale
        cells = [sum((h[0] - a[0] + 1) * (h[1] - a[1] + 1) * (h[2] - a[2] + 1) for a, h in lev) for lev in levels]
        ms = 1e3 * dt / steps
        if config == "c5":
            what = ("C5-shaped 4-level hierarchy, %dx%dx%d cells on every level, (2,2,1) refinements nested around the "
                    "topographic bump, terrain-following NON-diagonal metric (19-point kernels) produced on the device from "
                    "the nodal depth (BathymetricBaseMap::fill_dxdXi + CONVERTFAB + GeoSourceInterface::fill_Jgup), no "
                    "periodic direction, 64x64x%d boxes, AMR V-cycle 4/4/2, multigrid solver; every level's boxes in y-slabs "
                    "over %d GPU(s)" % (512 // s, 512 // s, 64 // s, 64 // s, world))
        else:
            what = ("C4 LockExchange-shaped 3-level %dx%dx%d, (2,2,1) refinements of the central half / quarter in x, "
                    "Cartesian metric, y periodic, 128^3 boxes, AMR V-cycle 4/4/2; every level's boxes in y-slabs over "
                    "%d GPU(s)" % (1024 // s, 1024 // s, 128 // s, world))
        return {"workload": what,
                "ms_per_amr_vcycle": ms, "amr_vcycles_per_s": 1e3 / ms, "steps": steps, "warmup": warmup,
                "cells_per_level": cells, "boxes_per_level": [len(b) for b in levels], "define_seconds": t_def,
                "mg_depth_per_level": [v.depth() for v in gpu.levels],
                "gsrb_cell_updates_per_s": sum(cells) * 8 / (ms * 1e-3),   # 4 pre + 4 post sweeps per level per cycle
                "amr_vcycle_contraction": r1 / r0}
    finally:
        gpu.undefine()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--size", dest="n", type=int, default=N_FINE, help="fine grid size per direction (512 = BASELINE C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the c4_amr sub-record")
    ap.add_argument("--no-cartesian", action="store_true", help="skip the c2_cartesian sub-record")
    ap.add_argument("--no-c5", action="store_true", help="skip the c5_amr sub-record")
    ap.add_argument("--c4-steps", type=int, default=5)
    ap.add_argument("--c4-warmup", type=int, default=1)
    ap.add_argument("--c4-scale", type=int, default=1, help="divide every C4 extent (rehearsals on small boxes)")
    args = ap.parse_args()

    import torch
    from somar_amd import api, synthetic   # the measured path never touches oracle/ (only cpu_baseline() above does)

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
        api.comm_selftest(comm)   # all-reduce + ring exchange checked on the host before anything is timed

    F = api
    n = args.n
    failures = []
    copy_gbs = measured_copy_gbs(torch) if rank == 0 else None
    # what this device streams for the fused sweep's own mix of streams (6 reads + 1 write, no stencil): the practical ceiling
    mix_gbs = api.stream_probe(2, 512 ** 3, 10) if rank == 0 else None
    read_gbs = api.stream_probe(1, 512 ** 3, 10) if rank == 0 else None

    # ---------------- headline: C2, stretched diagonal metric ----------------
    gpu, boxes, t_def = build_c2(api, synthetic, n, world, rank, comm, "stretched")
    depth = gpu.depth()
    ratios = [list(r) for r in gpu.mgRefRatios()]
    cells_total = n ** 3
    cells_local = cells_total // world
    # the residual of a solve from phi = 0: std::mt19937_64(12345), uniform(-1,1), minus its J-weighted mean; generated on the host,
    # resident in HBM before the timed region; the CPU baseline below gets the same array
    res_host = c2_random_field(n)
    upload_c2_residual(gpu, F, res_host)
    dt = time_c2(gpu, F, torch, dist, args.steps, args.warmup)
    ms_per_step = 1e3 * dt / args.steps
    value = args.steps / dt

    # per-kernel HIP events: a separate profiled pass (launch by launch, events around the depth-0 kernels), NOT timed above
    gpu.profileEnable(True)
    for _ in range(5):
        gpu.vcycleFromZero(F.F_CORR, F.F_RES)
    gpu.sync()
    n_gsrb, ms_gsrb = gpu.profileGet(0)
    n_rr, ms_rr = gpu.profileGet(1)   # depth-0 residual launches of the V-cycle = fused residual + restriction
    n_x, ms_x = gpu.profileGet(2)     # ghost exchanges with other ranks, all sharded depths (pack + wire + unpack, NOT overlapped here)
    n_tail, ms_tail = gpu.profileGet(3)   # the replicated coarse tail: gather + its whole V-cycle incl. the bottom solver
    gpu.profileEnable(False)
    gpu.profileEnable(True)
    # the plain residual (north-star unit "one residual + one red+black sweep")
    for _ in range(5):
        gpu.residual(0, F.F_SCRATCH, F.F_CORR, F.F_RES)
    gpu.sync()
    n_op, ms_op = gpu.profileGet(1)
    gpu.profileEnable(False)

    contraction = contraction_c2(gpu, F)
    if not (0.0 < contraction < 1.0):
        failures.append("C2 V-cycle does not contract: |r1|/|r0| = %r" % contraction)
    want = SINGLE_RANK_CONTRACTION.get(("stretched", n))
    if world == 1 and want is not None and abs(contraction - want) > 1e-2 * want:
        failures.append("C2 contraction %.6f departs from the recorded single-GPU value %.6f" % (contraction, want))
    gpu.undefine()
    del gpu
    contraction_twin = None
    if world > 1:
        # The box layout decides the multigrid depth (the reference's coarsenable test works per box), so an N-box run does
        # NOT contract like the one-box run (256^3: 0.4892 on one box, 0.4999 on four).  What it must reproduce is the SAME
        # N-box layout held by ONE process: same hierarchy, same arithmetic, only the cross-rank association of the large-level
        # mean sums differs (round-off).  Rank 0 builds that twin next to its shard and compares; a broken exchange or
        # reduction on the real transport shows up here, before any throughput is believed.
        ok = 1
        if rank == 0:
            twin, _, _ = build_c2(api, synthetic, n, world, 0, None, "stretched", all_on_this_rank=True)
            upload_c2_residual(twin, F, res_host)
            contraction_twin = contraction_c2(twin, api)
            twin.undefine()
            del twin
            if abs(contraction - contraction_twin) > 1e-6 * abs(contraction_twin):
                ok = 0
                failures.append("sharded C2 contraction %.12f departs from the single-process run of the same %d-box layout "
                                "%.12f" % (contraction, world, contraction_twin))
        flag = [ok]
        dist.broadcast_object_list(flag, src=0)
        if not flag[0] and rank != 0:
            failures.append("rank 0 found the sharded contraction off its single-process twin")

    fused = cells_total >= int(os.environ.get("SOMAR_FUSED_MIN_CELLS", "262144"))
    kname = "k_gsrb_fused (red+black sweep, depth 0)" if fused else "k_gsrb_ortho (one colour pass, depth 0)"
    b_launch = (2.0 if fused else 1.0) * B_GSRB_COLOR
    t_gsrb = ms_gsrb / max(n_gsrb, 1) * 1e-3      # s per GSRB launch (depth 0)
    t_op = ms_op / max(n_op, 1) * 1e-3
    achieved = b_launch * cells_local / t_gsrb / 1e9
    t_sweep = t_gsrb if fused else 2 * t_gsrb
    traffic = None
    traffic_src = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf) and n == N_FINE and world == 1:
        tj = json.load(open(tf))
        traffic = tj.get("k_gsrb_fused_bytes_per_launch" if fused else "k_gsrb_ortho_bytes_per_launch")
        traffic_src = tj.get("source")
    unit_t = t_sweep + t_op                       # one red+black sweep + one residual (north-star unit)
    out = {
        "metric": "pressure-Poisson V-cycles/sec", "value": value, "unit": "V-cycles/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2 single-level %d^3 Poisson, diagonal stretched metric, Neumann, LevelGSRB V-cycle "
                               "2/2/2 + BiCGStab bottom; layout = %d box(es) %s, one per GPU"
                               % (n, world, "x".join(str(h - l + 1) for l, h in zip(*boxes[0]))),
                   "mg_depth": depth, "mg_ref_ratios": ratios, "cells": cells_total, "define_seconds": t_def},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/traffic.json: HBM bytes per launch of this kernel from separate rocprofv3 --pmc "
                                       "FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of this bench command (%s); cited, "
                                       "not measured in this run" % (traffic_src or "see the file's note"),
                     "algorithmic_bytes_per_launch": b_launch * cells_local, "launches": n_gsrb,
                     "avg_launch_ms": t_gsrb * 1e3, "measured_copy_GBs": copy_gbs,
                     "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                     "measured_stream_mix_GBs": mix_gbs, "measured_read_GBs": read_gbs,
                     "frac_of_stream_mix": (achieved / mix_gbs) if mix_gbs else None,
                     "note": "per-launch HIP events from a separate profiled pass of the same V-cycle (5 cycles); the timed "
                             "loop runs without events"},
        "gsrb_cell_updates_per_s": cells_local / t_sweep * world,
        "residual_kernel": {"avg_launch_ms": t_op * 1e3, "achieved_GBs": B_RESIDUAL * cells_local / t_op / 1e9,
                            "frac": B_RESIDUAL * cells_local / t_op / 1e9 / HBM_PEAK_GBS},
        "residual_restrict_kernel": {"avg_launch_ms": ms_rr / max(n_rr, 1), "launches": n_rr,
                                     "note": "k_resid_march<2>: residual + J-weighted restriction in one pass, "
                                             "49 B/cell algorithmic",
                                     "achieved_GBs": 49.0 * cells_local / (ms_rr / max(n_rr, 1) * 1e-3) / 1e9},
        "residual_plus_smooth_unit": {"ms": unit_t * 1e3, "algorithmic_GBs": 120.0 * cells_local / unit_t / 1e9,
                                      "frac_of_hbm_peak": 120.0 * cells_local / unit_t / 1e9 / HBM_PEAK_GBS},
        "vcycle_contraction": contraction,
        "vcycle_contraction_single_process_same_layout": contraction_twin,
    }
    if world > 1:
        # what a scaling run is made of (rank 0's view, from the profiled pass: every launch timed, nothing overlapped)
        out["multi_gpu"] = {
            "partition": os.environ.get("SOMAR_BENCH_PARTITION", "yz"),
            "exchange_ms_per_vcycle": ms_x / 5.0, "exchanges_per_vcycle": n_x / 5.0,
            "replicated_tail_ms_per_vcycle": ms_tail / 5.0,
            "note": "exchange = pack + grouped send/recv + unpack of every ghost exchange on the sharded depths, serial on the "
                    "solver's stream in this pass; in the timed loop the fused sweeps' exchanges travel on a second stream under "
                    "their interior tiles (SOMAR_NO_OVERLAP=1 switches that off).  The replicated tail (every rank runs the depths "
                    "below 128^3 cells redundantly, one allgather in, nothing out) does not shrink with N."}

    # ---------------- SURVEY 8d variant (i): true Cartesian metric, the converging case ----------------
    if not args.no_cartesian:
        g2, _, t_def2 = build_c2(api, synthetic, n, world, rank, comm, "cartesian")
        upload_c2_residual(g2, F, res_host)
        dt2 = time_c2(g2, F, torch, dist, args.steps, args.warmup)
        c2 = contraction_c2(g2, F)
        if not (0.0 < c2 < 1.0):
            failures.append("Cartesian C2 V-cycle does not contract: %r" % c2)
        out["c2_cartesian"] = {"workload": "C2 variant (i): %d^3, all-ones metric (stored as arrays), same V-cycle" % n,
                               "value": args.steps / dt2, "unit": "V-cycles/s", "ms_per_step": 1e3 * dt2 / args.steps,
                               "vcycle_contraction": c2, "define_seconds": t_def2}
        g2.undefine()
        del g2

    # ---------------- C4: the workload of the north star's scaling target ----------------
    if not args.no_c4:
        c4 = bench_c4(api, torch, dist, comm, world, args.c4_steps, args.c4_warmup, args.c4_scale)
        if not (0.0 < c4["amr_vcycle_contraction"] < 1.0):
            failures.append("C4 AMR V-cycle does not contract: %r" % c4["amr_vcycle_contraction"])
        elif args.c4_scale == 1 and abs(c4["amr_vcycle_contraction"] - AMR_CONTRACTION["c4"]) > 5e-3 * AMR_CONTRACTION["c4"]:
            failures.append("C4 AMR V-cycle contraction %.7f departs from the recorded %.7f"
                            % (c4["amr_vcycle_contraction"], AMR_CONTRACTION["c4"]))
        out["c4_amr"] = c4

    # ---------------- C5's shape: 4 levels, non-diagonal terrain-following metric (19-point kernels) ----------------
    if not args.no_c5:
        try:
            c5 = bench_c4(api, torch, dist, comm, world, args.c4_steps, args.c4_warmup, args.c4_scale, config="c5")
            if not (0.0 < c5["amr_vcycle_contraction"] < 1.0):
                failures.append("C5 AMR V-cycle does not contract: %r" % c5["amr_vcycle_contraction"])
            elif args.c4_scale == 1 and abs(c5["amr_vcycle_contraction"] - AMR_CONTRACTION["c5"]) > 5e-3 * AMR_CONTRACTION["c5"]:
                failures.append("C5 AMR V-cycle contraction %.7f departs from the recorded %.7f"
                                % (c5["amr_vcycle_contraction"], AMR_CONTRACTION["c5"]))
            out["c5_amr"] = c5
        except Exception as e:   # a sub-record must not take the headline down with it at N > 1
            if world == 1:
                raise
            out["c5_amr"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, res_host)
    elif rank == 0:
        out["cpu_baseline"] = None
    if comm is not None:
        api.comm_destroy(comm)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failures:
        out["failures"] = failures
    if rank == 0:
        print(json.dumps(out))
    if failures:
        sys.stderr.write("bench.py: FAILED: " + "; ".join(failures) + "\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
