// somar_amd/csrc/box_bicgstab.hip -- the BiCGStab bottom solver of a multi-box level as ONE persistent launch, one workgroup
// per box: 7-point (diagonal metric) and 19-point (non-diagonal metric) operators.
//
// Reference: Chombo 3.1 BiCGStabSolver<T>::solve (EXTERNAL; configured at projection/AMRPressureSolver.cpp:253-265) driving
// MappedAMRPoissonOp::preCond / applyOp / residual (calculus/AMRElliptic/MappedAMRPoissonOp.cpp:684-734, 740-765, 628-640) with
// LevelGSRB (RelaxationMethods/GSRB.cpp:58-98) -- restated launch by launch in PressureSolver::bottom_solve (solver.cpp), whose
// iterates this kernel reproduces bit for bit.  Stencil arithmetic: the expression order of kernels.hip (gsrb_ortho_body,
// op_ortho_body) and full19.hip (k_gsrb_full, k_op_full, k_ghost_ops), i.e. of GSRBITER3DORTHO / GSRBBOUNDARYITER3DORTHO,
// GSRBITER3D / GSRBBOUNDARYITER3D, MAPPEDGETFLUX[ORTHO], MAPPEDFLUXDIVERGENCE3D; -ffp-contract=off.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace somar {

namespace {
__device__ __forceinline__ long long cidx(const PatchDesc& p, int i, int j, int k)
{
    return p.off + i + (long long)p.pj * j + p.pk * k;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ void publish_scalars(const double* vals, int n, ScalarPublish pub)
{
    for (int i = 0; i < n; ++i) __hip_atomic_store(pub.host_dst + i, vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub.host_seq, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

// ---- the bottom solver of a multi-box level as ONE persistent launch, one workgroup per box --------------------------------
// BASELINE C3 / C4 end their base level's V-cycle on 8 192 / 65 536 cells in 16 / 64 boxes, where BiCGStab runs 50-70 iterations
// of ~27 launches each: 12 of C3's 20 ms, 27 of C4's 147 ms per AMR V-cycle, none of which shards.  Here the whole solve is one
// launch: workgroup b owns box b, a thread owns CPT cells of it for the whole solve and keeps their coefficients and the
// BiCGStab vectors (r, r~, e, p, v, t) in registers; the only data another workgroup reads are the preconditioned vectors p~ / s~
// (fields in the level's layout, written with write-through stores, read with cache-bypassing loads -- agent-scope relaxed
// atomics -- straight from the neighbouring box's valid cells through a host-built neighbour table, so there is no ghost
// exchange at all) and the per-box partial sums.  Kernel boundaries become device-wide barriers (1.2-1.9 us for 16-64
// workgroups with one arrival counter, tools/gridsync_probe.hip; per-workgroup flags here); every workgroup computes every scalar from the same partial sums in the
// same order, so the control flow is uniform across the grid.  On levels of at most ordered_max cells the sums follow the
// reference's SERIAL order (box after box, Fortran order inside a box): one thread walks its box's terms staged in LDS, then
// the box totals are added in layout order -- bit-identical to the launch-by-launch path and to the oracle.  Above that
// (where the launch path sums by tree as well) each box's terms go through a fixed tree: the 1024-add chain of a C4 box cost
// 20 us per iteration.  Every spin loop is bounded: a barrier that gives up raises the abort flag, every workgroup leaves,
// the host reports the failure (no fallback).
// Control flow: Chombo 3.1 BiCGStabSolver<T>::solve as PressureSolver::bottom_solve restates it (solver.cpp).
constexpr unsigned BOX_SPIN_MAX = 1u << 26;
// FULL: the 19-point operator of a non-diagonal metric (3-D, one cell per thread).  What the launch-by-launch path does before
// every colour pass and operator application -- exchange, psi := phi, the box's ghost PROGRAM (extrapolated ghosts of psi, the
// cross-term Neumann ghosts of phi; solver_full.cpp), 12-20 dependent stages -- happens in LDS: the workgroup stages its box
// grown by one cell (own cells and the neighbours' through the same host-built source table the 7-point variant uses for its
// six neighbours), runs the box's ops on that copy stage by stage (one wavefront per op, a workgroup barrier per stage) and
// takes all 19 stencil values from it.  On BASELINE C5 three quarters of the 7 700 dispatches of an AMR V-cycle were the bottom
// solver's ghost stages, colour passes and reductions on a 4096-cell level.

template <int CPT, int MAXT, bool FULL>
__global__ __launch_bounds__(MAXT) void k_box_bicgstab(BoxBicg A)
{
    static_assert(!FULL || CPT == 1, "the 19-point variant keeps one cell per thread");
    __shared__ double X[BOX_MAX_CELLS], Y[BOX_MAX_CELLS];
    __shared__ double Fp[FULL ? BOX_FAB_MAX : 1], Fe[FULL ? BOX_FAB_MAX : 1];   // the box grown by one cell: phi and its extrapolated copy
    // the box's ghost programs ([0] operator, [1] smoother) as per-cell entries, their stage starts, and the coefficient triples
    // (J g^{ab}, J g^{ac}, J g^{aa} on the boundary face) of their cross-term Neumann ghost cells
    __shared__ BoxProgEntry E[FULL ? 2 : 1][FULL ? BOX_MAX_ENT : 1];
    __shared__ int ST[FULL ? 2 : 1][FULL ? BOX_MAX_STAGES + 1 : 1];
    __shared__ double NJ[FULL ? 2 : 1][FULL ? BOX_MAX_NEUM : 1][3];
    __shared__ double S[2][BOX_MAX_WG];
    __shared__ double M[16];
    __shared__ int s_ok;
    const int tid = threadIdx.x, nth = blockDim.x;
    const int b = blockIdx.x, nwg = gridDim.x;
    const PatchDesc p = A.patches[b];
    const StencilParams& P = A.P;
    const int cells = p.n[0] * p.n[1] * p.n[2];
    const bool three = P.active[2] != 0;
    const double xxScale = 1.0 / (P.dx[0] * P.dx[0]), yyScale = 1.0 / (P.dx[1] * P.dx[1]), zzScale = 1.0 / (P.dx[2] * P.dx[2]);
    const double sx = 1.0 / P.dx[0], sy = 1.0 / P.dx[1], sz = 1.0 / P.dx[2];

    // ---- this thread's cells: position, coefficients, neighbour table, boundary classification, denominators ----
    bool act[CPT];
    int c[CPT], nbo[CPT][6], col[CPT];
    unsigned flg[CPT];   // bits 0-5: Neumann face at x-, x+, y-, y+, z-, z+; bit 6: the boundary form of the sweep applies
    double jl[CPT][3], jh[CPT][3], Ji[CPT], dd[CPT], dg[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = tid + q * nth;
        act[q] = idx < cells;
        const int ii = act[q] ? idx : 0;
        const int li = ii % p.n[0], r = ii / p.n[0];
        const int lj = r % p.n[1], lk = r / p.n[1];
        const long long cc = cidx(p, li, lj, lk);
        c[q] = (int)cc;
        const int gi = p.lo[0] + li, gj = p.lo[1] + lj, gk = p.lo[2] + lk;
        col[q] = (gi + gj + gk) & 1;
        const int* nbp = A.nb + 6ll * (A.cstart[b] + ii);
#pragma unroll
        for (int s = 0; s < 6; ++s) nbo[q][s] = nbp[s];
        jl[q][0] = A.jg[0][cc]; jh[q][0] = A.jg[0][cc + 1];
        jl[q][1] = A.jg[1][cc]; jh[q][1] = A.jg[1][cc + p.pj];
        jl[q][2] = three ? A.jg[2][cc] : 0.0; jh[q][2] = three ? A.jg[2][cc + p.pk] : 0.0;
        Ji[q] = A.jinv[cc];
        const double lap = A.lapd[cc];
        dd[q] = P.alpha + P.beta * lap;
        const bool xl = gi == P.dom_lo[0], xh = gi == P.dom_hi[0], yl = gj == P.dom_lo[1], yh = gj == P.dom_hi[1];
        const bool zl = three && gk == P.dom_lo[2], zh = three && gk == P.dom_hi[2];
        const bool onb = xl || xh || yl || yh || zl || zh;
        const bool nxl = xl && P.neum[0][0], nxh = xh && P.neum[0][1], nyl = yl && P.neum[1][0], nyh = yh && P.neum[1][1];
        const bool nzl = zl && P.neum[2][0], nzh = zh && P.neum[2][1];
        flg[q] = (nxl ? 1u : 0u) | (nxh ? 2u : 0u) | (nyl ? 4u : 0u) | (nyh ? 8u : 0u) | (nzl ? 16u : 0u) | (nzh ? 32u : 0u) |
                 (onb ? 64u : 0u);
        if (!onb) {
            dg[q] = dd[q];
        } else {   // GSRBBOUNDARYITER's diagonal: the faces that carry a flux, in its order (3-D: lo sides, then hi sides)
            double ld = 0.0;
            if (three) {
                if (!nxl) ld = ld - xxScale * jl[q][0];
                if (!nyl) ld = ld - yyScale * jl[q][1];
                if (!nzl) ld = ld - zzScale * jl[q][2];
                if (!nxh) ld = ld - xxScale * jh[q][0];
                if (!nyh) ld = ld - yyScale * jh[q][1];
                if (!nzh) ld = ld - zzScale * jh[q][2];
            } else {
                if (!nxl) ld = ld - xxScale * jl[q][0];
                if (!nxh) ld = ld - xxScale * jh[q][0];
                if (!nyl) ld = ld - yyScale * jl[q][1];
                if (!nyh) ld = ld - yyScale * jh[q][1];
            }
            ld = ld * Ji[q];
            dg[q] = P.alpha + P.beta * ld;
        }
    }

    // ---- 19-point variant: the box's copy in LDS, its programs, this thread's 18 face coefficients ----
    const int m0 = p.n[0] + 2, m01 = m0 * (p.n[1] + 2), mtot = m01 * (p.n[2] + 2);
    const int fs[3] = {1, m0, m01};
    int fc[CPT];                 // own cell in the LDS copy
    double cj[CPT][3][3][2];     // J g^{ab} on the low / high a-face of the cell
    double dgF[CPT];             // GSRBBOUNDARYITER3D's denominator (its diagonal takes the faces in the order x-, x+, y-, y+, z-, z+)
    int nstages[2] = {0, 0};
    const double xyScale = 0.25 / (P.dx[0] * P.dx[1]), yzScale = 0.25 / (P.dx[1] * P.dx[2]), zxScale = 0.25 / (P.dx[2] * P.dx[0]);
    if (FULL) {
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            const int ii = act[q] ? tid + q * nth : 0;
            const int li = ii % p.n[0], r = ii / p.n[0];
            const int lj = r % p.n[1], lk = r / p.n[1];
            fc[q] = (li + 1) + m0 * (lj + 1) + m01 * (lk + 1);
            const long long st[3] = {1, (long long)p.pj, p.pk};
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    cj[q][a][bb][0] = A.jgf[a][bb][c[q]];
                    cj[q][a][bb][1] = A.jgf[a][bb][c[q] + st[a]];
                }
            const unsigned f = flg[q];
            double ld = 0.0;
            if (!(f & 1u)) ld = ld - xxScale * cj[q][0][0][0];
            if (!(f & 2u)) ld = ld - xxScale * cj[q][0][0][1];
            if (!(f & 4u)) ld = ld - yyScale * cj[q][1][1][0];
            if (!(f & 8u)) ld = ld - yyScale * cj[q][1][1][1];
            if (!(f & 16u)) ld = ld - zzScale * cj[q][2][2][0];
            if (!(f & 32u)) ld = ld - zzScale * cj[q][2][2][1];
            ld = ld * Ji[q];
            dgF[q] = (f & 64u) ? P.alpha + P.beta * ld : dd[q];
        }
        for (int w = 0; w < 2; ++w) {
            if (!A.ent_first[w]) continue;   // a program without entries (no wall anywhere: every ghost comes from the exchange)
            const int e0 = A.ent_first[w][b], ne = A.ent_first[w][b + 1] - e0;
            const int s0 = A.stg_first[w][b], ns = A.stg_first[w][b + 1] - s0;   // stage starts: one more than stages
            const int g0 = A.nfg_first[w][b], ng = A.nfg_first[w][b + 1] - g0;
            nstages[w] = ns > 0 ? ns - 1 : 0;
            const int* src = reinterpret_cast<const int*>(A.ent[w] + e0);       // 12-byte entries, 4-byte aligned
            int* dst = reinterpret_cast<int*>(E[w]);
            for (int q = tid; q < 3 * ne; q += nth) dst[q] = src[q];
            for (int q = tid; q < ns; q += nth) ST[w][q] = A.stg[w][s0 + q];
            for (int q = tid; q < ng; q += nth) {
                // entry order = slot order within the box: (direction, side) travel with the entry; the face with the table
                const long long fg = A.nfg[w][g0 + q];
                const int a = (int)(fg & 3);
                const long long face = fg >> 2;
                const int bb = (a + 1) % 3, cc = (a + 2) % 3;
                NJ[w][q][0] = A.jgf[a][bb][face];
                NJ[w][q][1] = A.jgf[a][cc][face];
                NJ[w][q][2] = A.jgf[a][a][face];
            }
        }
        __syncthreads();
    }

    // debug timing (A.dbg != nullptr): thread 0 of workgroup 0 accumulates s_memtime ticks per phase
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    const bool timing = A.dbg != nullptr && b == 0 && tid == 0;
    auto tick = [&]() -> long long { return timing ? (long long)__builtin_amdgcn_s_memtime() : 0ll; };

    // ---- device-wide barrier: every workgroup stamps its own flag with the barrier's number (a plain write-through store, no
    // read-modify-write on a shared counter), the first wavefront polls all flags at once, one or two per lane ----
    unsigned epoch = 0;
    auto gsync = [&]() -> bool {
        const long long t_in = tick();
        struct Acc { long long& a; long long t0; bool on; __device__ ~Acc() { if (on) a += (long long)__builtin_amdgcn_s_memtime() - t0; } } acc{tk[3], t_in, timing};
        __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's write-through stores have been acknowledged
        __syncthreads();
        if (nwg == 1) return true;
        ++epoch;
        if (tid < 64) {
            if (tid == 0) __hip_atomic_store(A.sync + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int ok = 1;
            unsigned spins = 0;
            for (;;) {
                bool here = true;
                for (int q = tid; q < nwg; q += 64)
                    here = here && __hip_atomic_load(A.sync + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
                if (__all(here)) break;
                if (++spins > BOX_SPIN_MAX || __hip_atomic_load(A.sync + BOX_MAX_WG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    ok = 0;
                    break;
                }
            }
            if (tid == 0) {
                if (!ok) __hip_atomic_store(A.sync + BOX_MAX_WG, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ok = ok;
            }
        }
        __syncthreads();
        return s_ok != 0;
    };
    auto ld_shared = [](const double* f, int off) { return __hip_atomic_load(f + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st_shared = [](double* f, int off, double v) { __hip_atomic_store(f + off, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // ---- 19-point variant: stage the box grown by one cell, run one of its ghost programs on the copy ----
    // zf's valid cells are current everywhere (a device-wide barrier has passed since they were written)
    auto stage_fab = [&](const double* zf, int which) {
        const long long t0 = tick();
        const int* src = A.fab_src + A.fab_start[b];
        for (int q = tid; q < mtot; q += nth) {
            const int so = src[q];
            const double v = so >= 0 ? ld_shared(zf, so) : 0.0;   // a ghost cell no exchange fills: written by the program before it is read
            Fp[q] = v;
            Fe[q] = v;                                            // psi := phi
        }
        __syncthreads();
        const long long t1 = tick();
        tk[0] += t1 - t0;
        // stage by stage: every entry of a stage is independent of the others -- one thread each, a workgroup barrier between
        for (int sg = 0; sg < nstages[which]; ++sg) {
            const int e1 = ST[which][sg + 1];
            for (int q = ST[which][sg] + tid; q < e1; q += nth) {
                const BoxProgEntry en = E[which][q];
                double* dst = (en.flags & 1) ? Fe : Fp;
                const double* sr = (en.flags & 2) ? Fe : Fp;
                const int f = en.dst;
                if (en.kind == 0) {
                    dst[f] = sr[f];
                } else if (en.kind == 1) {
                    dst[f] = sr[en.s1];
                } else if (en.kind == 2) {
                    dst[f] = 2.0 * sr[en.s1] - sr[en.s2];
                } else if (en.kind == 3) {
                    dst[f] = 3.0 * (sr[en.s1] - sr[en.s2]) + sr[en.s3];
                } else {   // the cross-term Neumann ghost (k_ghost_ops, full19.hip): the phi ghost that makes the boundary flux zero
                    const int a = (en.flags >> 2) & 3, bb = (a + 1) % 3, cc = (a + 2) % 3;
                    const int sgn = (en.flags & 16) ? 1 : -1;
                    const int v = f - sgn * fs[a];
                    const double idxb = -0.25 / P.dx[bb], idxc = -0.25 / P.dx[cc];
                    const int bk = -sgn * fs[a];
                    const double* nj = NJ[which][en.nslot];
                    const double cross = (Fe[f + fs[bb]] - Fe[f - fs[bb]] + Fe[f + bk + fs[bb]] - Fe[f + bk - fs[bb]]) * nj[0] * idxb +
                                         (Fe[f + fs[cc]] - Fe[f - fs[cc]] + Fe[f + bk + fs[cc]] - Fe[f + bk - fs[cc]]) * nj[1] * idxc;
                    Fp[f] = Fp[v] + (0.0 - cross) * P.dx[a] / nj[2];
                }
            }
            __syncthreads();
        }
        tk[1] += tick() - t1;
        ++tk[5];
    };
    // one GSRB point update, 19-point: GSRBITER3D / GSRBBOUNDARYITER3D in k_gsrb_full's expression order, operands from the LDS copy
    auto relax_cell_full = [&](int q, double rhsv) {
        const int f0 = fc[q];
#define PN(di, dj, dk) Fp[f0 + (di) + m0 * (dj) + m01 * (dk)]
#define EN(di, dj, dk) Fe[f0 + (di) + m0 * (dj) + m01 * (dk)]
#define JC(a, bb, s) cj[q][a][bb][s]
        const unsigned f = flg[q];
        double lphi;
        if (!(f & 64u)) {
            const double pdx = EN(1, 0, 0) - EN(-1, 0, 0);
            const double pdy = EN(0, 1, 0) - EN(0, -1, 0);
            const double pdz = EN(0, 0, 1) - EN(0, 0, -1);
            const double JDxx = JC(0, 0, 1) * PN(1, 0, 0) + JC(0, 0, 0) * PN(-1, 0, 0);
            const double JDxy = JC(0, 1, 1) * (EN(1, 1, 0) - EN(1, -1, 0) + pdy) - JC(0, 1, 0) * (pdy + EN(-1, 1, 0) - EN(-1, -1, 0));
            const double JDxz = JC(0, 2, 1) * (EN(1, 0, 1) - EN(1, 0, -1) + pdz) - JC(0, 2, 0) * (pdz + EN(-1, 0, 1) - EN(-1, 0, -1));
            const double JDyx = JC(1, 0, 1) * (EN(1, 1, 0) - EN(-1, 1, 0) + pdx) - JC(1, 0, 0) * (pdx + EN(1, -1, 0) - EN(-1, -1, 0));
            const double JDyy = JC(1, 1, 1) * PN(0, 1, 0) + JC(1, 1, 0) * PN(0, -1, 0);
            const double JDyz = JC(1, 2, 1) * (EN(0, 1, 1) - EN(0, 1, -1) + pdz) - JC(1, 2, 0) * (pdz + EN(0, -1, 1) - EN(0, -1, -1));
            const double JDzx = JC(2, 0, 1) * (EN(1, 0, 1) - EN(-1, 0, 1) + pdx) - JC(2, 0, 0) * (pdx + EN(1, 0, -1) - EN(-1, 0, -1));
            const double JDzy = JC(2, 1, 1) * (EN(0, 1, 1) - EN(0, -1, 1) + pdy) - JC(2, 1, 0) * (pdy + EN(0, 1, -1) - EN(0, -1, -1));
            const double JDzz = JC(2, 2, 1) * PN(0, 0, 1) + JC(2, 2, 0) * PN(0, 0, -1);
            lphi = P.beta * Ji[q] *
                   (JDxx * xxScale + JDyy * yyScale + JDzz * zzScale + (JDxy + JDyx) * xyScale + (JDyz + JDzy) * yzScale +
                    (JDzx + JDxz) * zxScale);
        } else {
            double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0;
            if (!(f & 1u))
                JDloX = +xxScale * JC(0, 0, 0) * PN(-1, 0, 0) -
                        xyScale * JC(0, 1, 0) * (EN(0, 1, 0) - EN(0, -1, 0) + EN(-1, 1, 0) - EN(-1, -1, 0)) -
                        zxScale * JC(0, 2, 0) * (EN(0, 0, 1) - EN(0, 0, -1) + EN(-1, 0, 1) - EN(-1, 0, -1));
            if (!(f & 2u))
                JDhiX = +xxScale * JC(0, 0, 1) * PN(1, 0, 0) +
                        xyScale * JC(0, 1, 1) * (EN(1, 1, 0) - EN(1, -1, 0) + EN(0, 1, 0) - EN(0, -1, 0)) +
                        zxScale * JC(0, 2, 1) * (EN(1, 0, 1) - EN(1, 0, -1) + EN(0, 0, 1) - EN(0, 0, -1));
            if (!(f & 4u))
                JDloY = -xyScale * JC(1, 0, 0) * (EN(1, 0, 0) - EN(-1, 0, 0) + EN(1, -1, 0) - EN(-1, -1, 0)) +
                        yyScale * JC(1, 1, 0) * PN(0, -1, 0) -
                        yzScale * JC(1, 2, 0) * (EN(0, 0, 1) - EN(0, 0, -1) + EN(0, -1, 1) - EN(0, -1, -1));
            if (!(f & 8u))
                JDhiY = +xyScale * JC(1, 0, 1) * (EN(1, 1, 0) - EN(-1, 1, 0) + EN(1, 0, 0) - EN(-1, 0, 0)) +
                        yyScale * JC(1, 1, 1) * PN(0, 1, 0) +
                        yzScale * JC(1, 2, 1) * (EN(0, 1, 1) - EN(0, 1, -1) + EN(0, 0, 1) - EN(0, 0, -1));
            if (!(f & 16u))
                JDloZ = -zxScale * JC(2, 0, 0) * (EN(1, 0, 0) - EN(-1, 0, 0) + EN(1, 0, -1) - EN(-1, 0, -1)) -
                        yzScale * JC(2, 1, 0) * (EN(0, 1, 0) - EN(0, -1, 0) + EN(0, 1, -1) - EN(0, -1, -1)) +
                        zzScale * JC(2, 2, 0) * PN(0, 0, -1);
            if (!(f & 32u))
                JDhiZ = +zxScale * JC(2, 0, 1) * (EN(1, 0, 1) - EN(-1, 0, 1) + EN(1, 0, 0) - EN(-1, 0, 0)) +
                        yzScale * JC(2, 1, 1) * (EN(0, 1, 1) - EN(0, -1, 1) + EN(0, 1, 0) - EN(0, -1, 0)) +
                        zzScale * JC(2, 2, 1) * PN(0, 0, 1);
            lphi = P.beta * Ji[q] * (JDloX + JDhiX + JDloY + JDhiY + JDloZ + JDhiZ);
        }
        return (rhsv - lphi) / dgF[q];
    };
    // L[z] at one cell, 19-point: flux19 (MAPPEDGETFLUX) on its six faces + divergence, k_op_full's expression order
    auto op_cell_full = [&](int q) {
        const double dxi[3] = {1.0 / P.dx[0], 1.0 / P.dx[1], 1.0 / P.dx[2]};
        const unsigned f = flg[q];
        double fl[3], fh[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int bb = (a + 1) % 3, cc = (a + 2) % 3;
            const double aScale = 1.0 * dxi[a], bScale = 0.25 * 1.0 * dxi[bb], cScale = 0.25 * 1.0 * dxi[cc];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int g = fc[q] + side * fs[a];
                const double v = aScale * JC(a, a, side) * (Fp[g] - Fp[g - fs[a]]) +
                                 bScale * JC(a, bb, side) * (Fe[g + fs[bb]] - Fe[g - fs[bb]] + Fe[g + fs[bb] - fs[a]] - Fe[g - fs[bb] - fs[a]]) +
                                 cScale * JC(a, cc, side) * (Fe[g + fs[cc]] - Fe[g - fs[cc]] + Fe[g + fs[cc] - fs[a]] - Fe[g - fs[cc] - fs[a]]);
                if (side == 0) fl[a] = v; else fh[a] = v;
            }
            if (f & (1u << (2 * a))) fl[a] = 0.0;
            if (f & (2u << (2 * a))) fh[a] = 0.0;
            fl[a] *= P.beta;
            fh[a] *= P.beta;
        }
        double l = Ji[q] * ((fh[0] - fl[0]) * dxi[0] + (fh[1] - fl[1]) * dxi[1] + (fh[2] - fl[2]) * dxi[2]);
        if (P.alpha != 0.0) l = P.alpha * Fp[fc[q]] + 1.0 * l;
        return l;
    };
#undef PN
#undef EN
#undef JC

    // ---- stencil pieces on one of this thread's cells (operation order of gsrb_ortho_body / op_ortho_body) ----
    auto relax_cell = [&](int q, const double* zf, double rhsv) {
        if (FULL) return relax_cell_full(q, rhsv);
        double v[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) v[s] = (s < 4 || three) ? ld_shared(zf, nbo[q][s]) : 0.0;
        const unsigned f = flg[q];
        double lphi;
        if (!three) {
            if (!(f & 64u)) {
                const double JDxx = xxScale * (jh[q][0] * v[1] + jl[q][0] * v[0]);
                const double JDyy = yyScale * (jh[q][1] * v[3] + jl[q][1] * v[2]);
                lphi = P.beta * (JDxx + JDyy) * Ji[q];
            } else {
                double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0;
                if (!(f & 1u)) JDloX = jl[q][0] * v[0];
                if (!(f & 2u)) JDhiX = jh[q][0] * v[1];
                if (!(f & 4u)) JDloY = jl[q][1] * v[2];
                if (!(f & 8u)) JDhiY = jh[q][1] * v[3];
                lphi = P.beta * Ji[q] * ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale);
            }
        } else if (!(f & 64u)) {
            const double JDxx = xxScale * (jh[q][0] * v[1] + jl[q][0] * v[0]);
            const double JDyy = yyScale * (jh[q][1] * v[3] + jl[q][1] * v[2]);
            const double JDzz = zzScale * (jh[q][2] * v[5] + jl[q][2] * v[4]);
            lphi = P.beta * Ji[q] * (JDxx + JDyy + JDzz);
        } else {
            double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0;
            if (!(f & 1u)) JDloX = jl[q][0] * v[0];
            if (!(f & 4u)) JDloY = jl[q][1] * v[2];
            if (!(f & 16u)) JDloZ = jl[q][2] * v[4];
            if (!(f & 2u)) JDhiX = jh[q][0] * v[1];
            if (!(f & 8u)) JDhiY = jh[q][1] * v[3];
            if (!(f & 32u)) JDhiZ = jh[q][2] * v[5];
            lphi = P.beta * Ji[q] * ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale + (JDloZ + JDhiZ) * zzScale);
        }
        return (rhsv - lphi) / dg[q];
    };
    auto op_cell = [&](int q, const double* zf, double pc) {   // L[z] at the cell whose own value is pc
        if (FULL) return op_cell_full(q);
        double v[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) v[s] = (s < 4 || three) ? ld_shared(zf, nbo[q][s]) : 0.0;
        const unsigned f = flg[q];
        double fxl = jl[q][0] * sx * (pc - v[0]);
        double fxh = jh[q][0] * sx * (v[1] - pc);
        double fyl = jl[q][1] * sy * (pc - v[2]);
        double fyh = jh[q][1] * sy * (v[3] - pc);
        double fzl = 0.0, fzh = 0.0;
        if (three) {
            fzl = jl[q][2] * sz * (pc - v[4]);
            fzh = jh[q][2] * sz * (v[5] - pc);
        }
        if (f & 1u) fxl = 0.0;
        if (f & 2u) fxh = 0.0;
        if (f & 4u) fyl = 0.0;
        if (f & 8u) fyh = 0.0;
        if (f & 16u) fzl = 0.0;
        if (f & 32u) fzh = 0.0;
        fxl *= P.beta; fxh *= P.beta; fyl *= P.beta; fyh *= P.beta; fzl *= P.beta; fzh *= P.beta;
        double l = three ? Ji[q] * ((fxh - fxl) * sx + (fyh - fyl) * sy + (fzh - fzl) * sz)
                         : Ji[q] * ((fxh - fxl) * sx + (fyh - fyl) * sy);
        if (P.alpha != 0.0) l = P.alpha * pc + 1.0 * l;
        return l;
    };

    // ---- reductions: per-box serial chain(s), then the box totals in layout order ----
    int nred = 0;
    auto chain = [&](const double* x) {
        double s = x[0];
        int q = 1;
        for (; q + 16 <= cells; q += 16) {
            double xv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = x[q + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) s = s + xv[j];
        }
        for (; q < cells; ++q) s = s + x[q];
        return s;
    };
    // the terms of up to two sums are in X (and Y); -> the two totals.  false: the barrier gave up
    auto finish_sums = [&](bool two, double& ra, double& rb) -> bool {
        const long long ts0 = tick();
        struct AccS { long long& a; long long& g; long long t0, g0; bool on; __device__ ~AccS() { if (on) a += ((long long)__builtin_amdgcn_s_memtime() - t0) - (g - g0); } } accs{tk[4], tk[3], ts0, tk[3], timing};
        __syncthreads();
        double* mine = A.sums + (nred & 1) * 2 * BOX_MAX_WG;
        ++nred;
        if (A.serial) {
            if (tid == 0) st_shared(mine, b, chain(X));
            if (two && tid == (nth > 64 ? 64 : 0)) st_shared(mine, BOX_MAX_WG + b, chain(Y));   // on another wavefront, side by side
        } else {
            // levels above ordered_max cells (where the launch-by-launch path sums by tree too): a fixed tree per box -- a
            // thread's terms in order, the wavefront's 64 partial sums by shuffles, the wavefronts' results in order
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) if (act[q]) { sa = sa + X[tid + q * nth]; if (two) sb = sb + Y[tid + q * nth]; }
            sa = wave_sum(sa);
            if (two) sb = wave_sum(sb);
            if ((tid & 63) == 0) { M[tid >> 6] = sa; M[8 + (tid >> 6)] = sb; }
            __syncthreads();
            if (tid == 0) {
                double ta = M[0], tb = M[8];
                for (int w = 1; w < (nth + 63) / 64; ++w) { ta = ta + M[w]; tb = tb + M[8 + w]; }
                st_shared(mine, b, ta);
                if (two) st_shared(mine, BOX_MAX_WG + b, tb);
            }
        }
        if (!gsync()) return false;
        for (int q = tid; q < nwg; q += nth) {
            S[0][q] = ld_shared(mine, q);
            if (two) S[1][q] = ld_shared(mine, BOX_MAX_WG + q);
        }
        __syncthreads();
        double ta = 0.0, tb = 0.0;
        for (int q = 0; q < nwg; ++q) ta = ta + S[0][q];
        if (two) for (int q = 0; q < nwg; ++q) tb = tb + S[1][q];
        ra = ta;
        rb = tb;
        return true;
    };
    auto finish_max = [&](double m, double& out) -> bool {
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_down(m, o, 64);
            m = w > m ? w : m;
        }
        __syncthreads();
        if ((tid & 63) == 0) M[tid >> 6] = m;
        __syncthreads();
        double* mine = A.sums + (nred & 1) * 2 * BOX_MAX_WG;
        ++nred;
        if (tid == 0) {
            double r = M[0];
            for (int w = 1; w < (nth + 63) / 64; ++w) r = M[w] > r ? M[w] : r;
            st_shared(mine, b, r);
        }
        if (!gsync()) return false;
        for (int q = tid; q < nwg; q += nth) S[0][q] = ld_shared(mine, q);
        __syncthreads();
        double r = S[0][0];
        for (int q = 1; q < nwg; ++q) r = S[0][q] > r ? S[0][q] : r;
        out = r;
        return true;
    };

    double r[CPT], rt[CPT], e[CPT], pv[CPT], v[CPT], t[CPT], zo[CPT], pt[CPT];
    auto dot = [&](const double* a, const double* bb, double& out) -> bool {
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) X[tid + q * nth] = a[q] * bb[q];
        double dummy;
        return finish_sums(false, out, dummy);
    };
    auto norm = [&](const double* a, double& out) -> bool {
        if (A.normType == 0) {
            double m = 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) if (act[q]) { const double w = fabs(a[q]); m = w > m ? w : m; }
            return finish_max(m, out);
        }
        double dummy, sres;
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) X[tid + q * nth] = A.normType == 1 ? fabs(a[q]) : a[q] * a[q];
        if (!finish_sums(false, sres, dummy)) return false;
        out = A.normType == 1 ? sres : sqrt(sres);
        return true;
    };
    // DIAGPRECOND + point-GSRB sweeps on zf with right-hand side w (registers): the result is in zf and, for this thread's own
    // cells, in zo; the last barrier leaves zf readable by everybody
    auto pre_cond = [&](double* zf, const double* w) -> bool {
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) {
            zo[q] = A.precondIters <= 0 ? w[q] : w[q] / dd[q];
            st_shared(zf, c[q], zo[q]);
        }
        if (!gsync()) return false;
        for (int it = 0; it < A.precondIters; ++it)
            for (int pass = 0; pass < 2; ++pass) {
                if (FULL) stage_fab(zf, 1);   // exchange + psi snapshot + the smoother's ghost program, all in LDS
#pragma unroll
                for (int q = 0; q < CPT; ++q) if (act[q] && col[q] == pass) {
                    zo[q] = relax_cell(q, zf, w[q]);
                    st_shared(zf, c[q], zo[q]);
                }
                if (!gsync()) return false;
            }
        return true;
    };
    // r = rhs - L[phi]: phi goes through z[0] (shared reads of the neighbours)
    auto residual = [&](const double* phiv) -> bool {
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) st_shared(A.z[0], c[q], phiv[q]);
        if (!gsync()) return false;
        if (FULL) stage_fab(A.z[0], 0);
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) r[q] = A.rhs[c[q]] - op_cell(q, A.z[0], phiv[q]);
        if (!gsync()) return false;   // z[0] is rewritten by the next pre_cond: everybody has read it
        return true;
    };
    auto finish = [&](int iters, int exit_code) {
        if (timing) for (int q = 0; q < 6; ++q) A.dbg[q] = tk[q];
        if (b == 0 && tid == 0) {
            const double vals[2] = {(double)iters, (double)exit_code};
            A.info[0] = vals[0];
            A.info[1] = vals[1];
            publish_scalars(vals, 2, A.pub);
        }
    };

    double phv[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        phv[q] = act[q] ? A.phi[c[q]] : 0.0;
        r[q] = rt[q] = e[q] = pv[q] = v[q] = t[q] = zo[q] = pt[q] = 0.0;
    }
    int recount = 0;
    if (!residual(phv)) return;
#pragma unroll
    for (int q = 0; q < CPT; ++q) rt[q] = r[q];
    int i = 0;
    double rho[4] = {0, 0, 0, 0};
    double nrm[2];
    if (!norm(r, nrm[0])) return;
    double initial_norm = nrm[0];
    const double initial_rnorm = nrm[0];
    nrm[1] = nrm[0];
    double alpha[2] = {0, 0}, beta[2] = {0, 0}, omega[2] = {0, 0};
    bool init = true;
    int restarts = 0;
    if (A.metric > 0) initial_norm = A.metric;
    const double eps = A.eps;
    int bottom_exit = -1;
    auto add_e_to_phi = [&]() {
#pragma unroll
        for (int q = 0; q < CPT; ++q) phv[q] = phv[q] + 1.0 * e[q];
    };
    auto store_phi = [&]() {
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (act[q]) A.phi[c[q]] = phv[q];
    };
    while ((i < A.imax && nrm[0] > eps * nrm[1]) && (nrm[1] > 0)) {
        ++i;
        nrm[1] = nrm[0];
        alpha[1] = alpha[0]; beta[1] = beta[0]; omega[1] = omega[0];
        rho[3] = rho[2]; rho[2] = rho[1];
        if (!dot(rt, r, rho[1])) return;
        if (rho[1] == 0.0) {
            add_e_to_phi();
            store_phi();
            finish(i, 2);
            return;
        }
        if (init) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) pv[q] = r[q];
            init = false;
        } else {
            beta[1] = (rho[1] / rho[2]) * (alpha[1] / omega[1]);
            const double bt = beta[1], bw = -beta[1] * omega[1];
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                double u = pv[q] * bt;
                u = u + bw * v[q];
                u = u + 1.0 * r[q];
                pv[q] = u;
            }
        }
        if (!pre_cond(A.z[0], pv)) return;
        if (FULL) stage_fab(A.z[0], 0);
#pragma unroll
        for (int q = 0; q < CPT; ++q) { pt[q] = zo[q]; v[q] = act[q] ? op_cell(q, A.z[0], zo[q]) : 0.0; }
        double m;
        if (!dot(rt, v, m)) return;
        alpha[0] = rho[1] / m;
        if (fabs(m) > A.small * fabs(rho[1])) {
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                r[q] = r[q] + (-alpha[0]) * v[q];
                e[q] = e[q] + alpha[0] * pt[q];
            }
            if (!norm(r, nrm[0])) return;
        } else {
#pragma unroll
            for (int q = 0; q < CPT; ++q) r[q] = 0.0;
            nrm[0] = 0.0;
        }
        if (nrm[0] > eps * initial_norm && nrm[0] > A.reps * initial_rnorm) {
            if (!pre_cond(A.z[1], r)) return;
            if (FULL) stage_fab(A.z[1], 0);
#pragma unroll
            for (int q = 0; q < CPT; ++q) t[q] = act[q] ? op_cell(q, A.z[1], zo[q]) : 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) if (act[q]) { X[tid + q * nth] = t[q] * r[q]; Y[tid + q * nth] = t[q] * t[q]; }
            double tr, tt;
            if (!finish_sums(true, tr, tt)) return;
            omega[0] = tr / tt;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                e[q] = e[q] + omega[0] * zo[q];
                r[q] = r[q] + (-omega[0]) * t[q];
            }
            if (!norm(r, nrm[0])) return;
        }
        if (nrm[0] <= eps * initial_norm || nrm[0] <= A.reps * initial_rnorm) {
            bottom_exit = 1;
            break;
        }
        if (omega[0] == 0.0 || nrm[0] > (1 - A.hang) * nrm[1]) {
            if (recount == 0) {
                recount = 1;
            } else {
                recount = 0;
                add_e_to_phi();
                if (restarts == A.numRestarts) {
                    store_phi();
                    finish(i, 3);
                    return;
                }
                if (!residual(phv)) return;
                if (!norm(r, nrm[0])) return;
                rho[0] = rho[1] = rho[2] = rho[3] = 0.0;
                alpha[0] = beta[0] = omega[0] = 0.0;
#pragma unroll
                for (int q = 0; q < CPT; ++q) { rt[q] = r[q]; e[q] = 0.0; }
                ++restarts;
                init = true;
            }
        }
    }
    add_e_to_phi();
    store_phi();
    finish(i, bottom_exit);
}
void launch_box_bicgstab(hipStream_t st, const LevelDev& L, int max_box_cells, BoxBicg A)
{
    A.patches = L.patches; A.npatches = L.npatches;
    for (int d = 0; d < 3; ++d) A.jg[d] = L.jg[d];
    A.jinv = L.jinv; A.lapd = L.lapdiag; A.P = L.P;
    SOMAR_CHECK(L.npatches >= 1 && L.npatches <= BOX_MAX_WG && max_box_cells >= 1 && max_box_cells <= BOX_MAX_CELLS,
                "k_box_bicgstab: level outside the kernel's limits");
    SOMAR_HIP(hipMemsetAsync(A.sync, 0, (BOX_MAX_WG + 1) * sizeof(unsigned), st));
    // 512-thread workgroups (256 VGPRs each: a thread's cells, their coefficients and vectors stay in registers), 1 / 2 / 4 cells
    // per thread
    const int nth = std::min(512, (max_box_cells + 63) / 64 * 64);
    if (A.full) {
        SOMAR_CHECK(max_box_cells <= 256, "k_box_bicgstab: the 19-point variant takes boxes of at most 256 cells");
        for (int a = 0; a < 3; ++a)
            for (int bb = 0; bb < 3; ++bb) A.jgf[a][bb] = L.jgf[a][bb];
        // 256 threads whatever the box size (up to 512 VGPRs each, no spills): the ops of a ghost-program stage run one per
        // wavefront, side by side -- with the 64 threads a 4^3 box asks for a pass took 67 us, most of it ops in single file
        hipLaunchKernelGGL((k_box_bicgstab<1, 256, true>), dim3(L.npatches), dim3(256), 0, st, A);
    } else if (max_box_cells <= 512) hipLaunchKernelGGL((k_box_bicgstab<1, 512, false>), dim3(L.npatches), dim3(nth), 0, st, A);
    else if (max_box_cells <= 1024) hipLaunchKernelGGL((k_box_bicgstab<2, 512, false>), dim3(L.npatches), dim3(nth), 0, st, A);
    else hipLaunchKernelGGL((k_box_bicgstab<4, 512, false>), dim3(L.npatches), dim3(nth), 0, st, A);
}

}  // namespace somar
