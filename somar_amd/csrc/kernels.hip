// somar_amd/csrc/kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// The hot path of SOMAR's pressure projection is a 7-point variable-coefficient stencil:
// every kernel here is HBM-bandwidth bound (0.1-0.4 flop/byte), so there is no MFMA
// anywhere; what matters is 64-lane-wide unit-stride access along i, one shared index
// computation for all coefficient planes, XCD-contiguous tile order (done on the host when
// the tile table is built) and no temporaries in HBM (flux + divergence + subtract fused).
//
// Arithmetic mirrors the reference's Fortran operation order term by term and the file is
// compiled with -ffp-contract=off, so results are bit-identical to oracle/kernels.c
// (reductions excepted: those are tree sums, documented at each kernel).
//
// Reference kernels restated here (paths relative to /root/reference/src):
//   GSRBITER3DORTHO            calculus/AMRElliptic/RelaxationMethods/GSRBF.ChF:545-701
//   GSRBBOUNDARYITER3DORTHO    .../GSRBF.ChF:1362-1505          (fused into the same launch)
//   MAPPEDGETFLUXORTHO         calculus/AMRElliptic/MappedAMRPoissonOpOrthoF.ChF:33-82
//   MAPPEDFLUXDIVERGENCE3D     calculus/DivCurlGrad/DivCurlGradF.ChF:1122-1215
//   SUBTRACTOP/AXBYIP/DIAGPRECOND  calculus/AMRElliptic/MappedAMRPoissonOpF.ChF:36-85, 284-328
//   FILLMAPPEDLAPDIAG3D        .../MappedAMRPoissonOpF.ChF:215-274
//   MAPPEDAVERAGE2, UNMAPPEDAVERAGEHARMONIC, UNMAPPEDAVERAGEFACE
//                              MappedChombo/MappedCoarseAverageF.ChF:132-167, 48-82, 182-221
//   ConstInterpPS / ConstInterpWithAvgPS  calculus/AMRElliptic/MGStrategies/ProlongationStrategyF.ChF:36-160
//   JACOBIITER                 calculus/AMRElliptic/RelaxationMethods/JacobiF.ChF
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace somar {

// ------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------
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
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}

// 256-thread block reduction (4 waves). Result valid in thread 0.
template <bool MAX>
__device__ __forceinline__ double block_reduce(double v)
{
    __shared__ double s[8];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    v = MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();  // protect s[] against a previous use
    if (lane == 0) s[w] = v;
    __syncthreads();
    if (tid == 0) {
        const int nw = (blockDim.x * blockDim.y + 63) >> 6;
        double r = s[0];
        for (int q = 1; q < nw; ++q) r = MAX ? fmax(r, s[q]) : r + s[q];
        v = r;
    }
    return v;
}

__device__ __forceinline__ void publish_scalars(const double* vals, int n, ScalarPublish pub)
{
    for (int i = 0; i < n; ++i) __hip_atomic_store(pub.host_dst + i, vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(pub.host_seq, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------
// GSRB, diagonal metric, one colour.  Interior cells follow GSRBITER3DORTHO, cells that
// touch a DOMAIN face (periodic ones included: GSRB.cpp:76,87-92) follow
// GSRBBOUNDARYITER3DORTHO with a Neumann face contributing neither flux nor diagonal.
// Both in one launch: the reference's 26 one-cell-thick boundary sub-box calls per box
// become a per-lane branch that only diverges on the shell.
// Thread = one i-pair; exactly one cell of the pair has this colour.
// ------------------------------------------------------------------------------------
// bx / tx / ty: the workgroup's tile and the thread's place in it -- the block and thread indices in the kernel of the same
// name, virtual ones in the single-workgroup bottom solver of tiny levels (k_tiny_bicgstab)
__device__ __forceinline__ void gsrb_ortho_body(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                double* __restrict__ phi, const double* __restrict__ rhs,
                                                const double* __restrict__ jgx, const double* __restrict__ jgy,
                                                const double* __restrict__ jgz, const double* __restrict__ jinv,
                                                const double* __restrict__ lapd, const StencilParams& P, int color, int loose,
                                                int bx, int tx, int ty)
{
    // loose 0: LevelGSRB -- every cell of the colour; boundary form where the cell touches a DOMAIN face.
    // loose 1: LooseGSRB's interior phase -- cells at least one cell inside their BOX, interior form.
    // loose 2: LooseGSRB's boundaryGSRB(doAll) -- the one-cell shell of every box, boundary form (GSRB.cpp:104-141)
    const Tile t = tiles[bx];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + ty;
    const int li0 = t.i0 + 2 * tx;
    if (lj >= p.n[1] || li0 >= p.n[0]) return;
    const double xxScale = 1.0 / (P.dx[0] * P.dx[0]);
    const double yyScale = 1.0 / (P.dx[1] * P.dx[1]);
    const double zzScale = 1.0 / (P.dx[2] * P.dx[2]);
    const int gj = p.lo[1] + lj;
    const bool bj = (gj == P.dom_lo[1]) || (gj == P.dom_hi[1]);
    const long long sj = p.pj, sk = p.pk;
    for (int kk = 0; kk < t.nk; ++kk) {
        const int lk = t.k0 + kk;
        const int gk = p.lo[2] + lk;
        const int li = li0 + ((p.lo[0] + li0 + gj + gk + color) & 1);
        if (li >= p.n[0]) continue;
        const int gi = p.lo[0] + li;
        const long long c = cidx(p, li, lj, lk);
        bool onb = bj || (gk == P.dom_lo[2]) || (gk == P.dom_hi[2]) || (gi == P.dom_lo[0]) || (gi == P.dom_hi[0]);
        bool onb2d = bj || (gi == P.dom_lo[0]) || (gi == P.dom_hi[0]);
        if (loose) {
            const bool shell2 = (li == 0) || (li == p.n[0] - 1) || (lj == 0) || (lj == p.n[1] - 1);
            const bool shell = shell2 || (P.active[2] && ((lk == 0) || (lk == p.n[2] - 1)));
            if ((loose == 1) == shell) continue;  // phase 1 skips the shell, phase 2 skips the interior
            onb = shell;
            onb2d = shell2;
        }
        const double Ji = jinv[c];
        double out;
        if (!P.active[2]) {
            // SpaceDim == 2 build of the reference: GSRBITER2DORTHO / GSRBBOUNDARYITER2DORTHO
            // (GSRBF.ChF:440-540, 1254-1356) -- note the different association of beta, Jinv and the side order
            if (!onb2d) {
                const double JDxx = xxScale * (jgx[c + 1] * phi[c + 1] + jgx[c] * phi[c - 1]);
                const double JDyy = yyScale * (jgy[c + sj] * phi[c + sj] + jgy[c] * phi[c - sj]);
                const double lphi = P.beta * (JDxx + JDyy) * Ji;
                out = (rhs[c] - lphi) / (P.alpha + P.beta * lapd[c]);
            } else {
                const bool nxl = (gi == P.dom_lo[0]) && P.neum[0][0];
                const bool nxh = (gi == P.dom_hi[0]) && P.neum[0][1];
                const bool nyl = (gj == P.dom_lo[1]) && P.neum[1][0];
                const bool nyh = (gj == P.dom_hi[1]) && P.neum[1][1];
                double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, ld = 0.0;
                if (!nxl) { JDloX = jgx[c] * phi[c - 1];        ld = ld - xxScale * jgx[c]; }
                if (!nxh) { JDhiX = jgx[c + 1] * phi[c + 1];    ld = ld - xxScale * jgx[c + 1]; }
                if (!nyl) { JDloY = jgy[c] * phi[c - sj];       ld = ld - yyScale * jgy[c]; }
                if (!nyh) { JDhiY = jgy[c + sj] * phi[c + sj];  ld = ld - yyScale * jgy[c + sj]; }
                ld = ld * Ji;
                const double lphi = P.beta * Ji * ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale);
                out = (rhs[c] - lphi) / (P.alpha + P.beta * ld);
            }
        } else if (!onb) {
            const double JDxx = xxScale * (jgx[c + 1] * phi[c + 1] + jgx[c] * phi[c - 1]);
            const double JDyy = yyScale * (jgy[c + sj] * phi[c + sj] + jgy[c] * phi[c - sj]);
            const double JDzz = zzScale * (jgz[c + sk] * phi[c + sk] + jgz[c] * phi[c - sk]);
            const double lphi = P.beta * Ji * (JDxx + JDyy + JDzz);
            out = (rhs[c] - lphi) / (P.alpha + P.beta * lapd[c]);
        } else {
            const bool nxl = (gi == P.dom_lo[0]) && P.neum[0][0];
            const bool nxh = (gi == P.dom_hi[0]) && P.neum[0][1];
            const bool nyl = (gj == P.dom_lo[1]) && P.neum[1][0];
            const bool nyh = (gj == P.dom_hi[1]) && P.neum[1][1];
            const bool nzl = (gk == P.dom_lo[2]) && P.neum[2][0];
            const bool nzh = (gk == P.dom_hi[2]) && P.neum[2][1];
            double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0, ld = 0.0;
            if (!nxl) { JDloX = jgx[c] * phi[c - 1];        ld = ld - xxScale * jgx[c]; }
            if (!nyl) { JDloY = jgy[c] * phi[c - sj];       ld = ld - yyScale * jgy[c]; }
            if (!nzl) { JDloZ = jgz[c] * phi[c - sk];       ld = ld - zzScale * jgz[c]; }
            if (!nxh) { JDhiX = jgx[c + 1] * phi[c + 1];    ld = ld - xxScale * jgx[c + 1]; }
            if (!nyh) { JDhiY = jgy[c + sj] * phi[c + sj];  ld = ld - yyScale * jgy[c + sj]; }
            if (!nzh) { JDhiZ = jgz[c + sk] * phi[c + sk];  ld = ld - zzScale * jgz[c + sk]; }
            ld = ld * Ji;
            const double lphi = P.beta * Ji *
                                ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale +
                                 (JDloZ + JDhiZ) * zzScale);
            out = (rhs[c] - lphi) / (P.alpha + P.beta * ld);
        }
        phi[c] = out;
    }
}

// Pull exchange: before a stencil kernel's workgroup computes, it refreshes the ghost cells in the one-cell halo of ITS tile from
// their source cells (the clipped local copy items of Level::define) -- the values a preceding k_copy_items launch would have
// put there.  Workgroups refresh shared ghost cells redundantly with the same value.  Race-free for a colour pass: a cell of
// the pass's colour reads only neighbours of the other colour, which no workgroup updates in this launch; ghosts of the
// pass's own colour may catch an old or a new value, are not read here, and are pulled again before the next pass reads them.
__device__ __forceinline__ void pull_tile_ghosts(const CopyItem* __restrict__ items, const int* __restrict__ start,
                                                 const PatchDesc* __restrict__ patches, double* __restrict__ f)
{
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nt = blockDim.x * blockDim.y;
    for (int q = start[blockIdx.x]; q < start[blockIdx.x + 1]; ++q) {
        const CopyItem it = items[q];
        const PatchDesc sp = patches[it.src_patch];
        const PatchDesc dp = patches[it.dst_patch];
        const int n0 = it.n[0], n01 = it.n[0] * it.n[1];
        const int cells = n01 * it.n[2];
        for (int idx = tid; idx < cells; idx += nt) {
            const int k = idx / n01;
            const int r = idx - k * n01;
            const int j = r / n0, i = r - j * n0;
            f[cidx(dp, it.dst_lo[0] + i, it.dst_lo[1] + j, it.dst_lo[2] + k)] =
                f[cidx(sp, it.src_lo[0] + i, it.src_lo[1] + j, it.src_lo[2] + k)];
        }
    }
    __threadfence_block();
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_gsrb_ortho(const Tile* __restrict__ tiles,
                                                    const PatchDesc* __restrict__ patches,
                                                    double* __restrict__ phi,
                                                    const double* __restrict__ rhs,
                                                    const double* __restrict__ jgx,
                                                    const double* __restrict__ jgy,
                                                    const double* __restrict__ jgz,
                                                    const double* __restrict__ jinv,
                                                    const double* __restrict__ lapd,
                                                    StencilParams P, int color, int loose,
                                                    const CopyItem* __restrict__ titems, const int* __restrict__ tstart)
{
    if (tstart) pull_tile_ghosts(titems, tstart, patches, phi);
    gsrb_ortho_body(tiles, patches, phi, rhs, jgx, jgy, jgz, jinv, lapd, P, color, loose, blockIdx.x, threadIdx.x, threadIdx.y);
}

// ------------------------------------------------------------------------------------
// Operator / residual, diagonal metric: flux (MAPPEDGETFLUXORTHO) + zero Neumann boundary
// flux (EllipticConstNeumBCFluxClass) + "flux *= beta" + MAPPEDFLUXDIVERGENCE3D + AXBYIP
// + SUBTRACTOP fused; the reference's three flux temporaries never exist.
// MODE 0: out = rhs - L[phi]   MODE 1: out = L[phi]
// Thread = one i-pair (both cells), k-loop over the tile.
// ------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void op_ortho_body(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                              double* __restrict__ out, const double* __restrict__ phi,
                                              const double* __restrict__ rhs, const double* __restrict__ jgx,
                                              const double* __restrict__ jgy, const double* __restrict__ jgz,
                                              const double* __restrict__ jinv, const StencilParams& P, int bx, int tx, int ty)
{
    const Tile t = tiles[bx];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + ty;
    const int li0 = t.i0 + 2 * tx;
    if (lj >= p.n[1] || li0 >= p.n[0]) return;
    const double sx = 1.0 / P.dx[0], sy = 1.0 / P.dx[1], sz = 1.0 / P.dx[2];  // scale = ref/dx, dxinv
    const int gj = p.lo[1] + lj;
    const long long sj = p.pj, sk = p.pk;
    const bool zyl = (gj == P.dom_lo[1]) && P.neum[1][0];
    const bool zyh = (gj == P.dom_hi[1]) && P.neum[1][1];
    const int npair = (li0 + 1 < p.n[0]) ? 2 : 1;
    for (int kk = 0; kk < t.nk; ++kk) {
        const int lk = t.k0 + kk;
        const int gk = p.lo[2] + lk;
        const bool zzl = (gk == P.dom_lo[2]) && P.neum[2][0];
        const bool zzh = (gk == P.dom_hi[2]) && P.neum[2][1];
        for (int q = 0; q < npair; ++q) {
            const int li = li0 + q;
            const int gi = p.lo[0] + li;
            const long long c = cidx(p, li, lj, lk);
            const double pc = phi[c];
            double fxl = jgx[c] * sx * (pc - phi[c - 1]);
            double fxh = jgx[c + 1] * sx * (phi[c + 1] - pc);
            double fyl = jgy[c] * sy * (pc - phi[c - sj]);
            double fyh = jgy[c + sj] * sy * (phi[c + sj] - pc);
            double fzl = 0.0, fzh = 0.0;
            if (P.active[2]) {
                fzl = jgz[c] * sz * (pc - phi[c - sk]);
                fzh = jgz[c + sk] * sz * (phi[c + sk] - pc);
            }
            if ((gi == P.dom_lo[0]) && P.neum[0][0]) fxl = 0.0;
            if ((gi == P.dom_hi[0]) && P.neum[0][1]) fxh = 0.0;
            if (zyl) fyl = 0.0;
            if (zyh) fyh = 0.0;
            if (zzl) fzl = 0.0;
            if (zzh) fzh = 0.0;
            fxl *= P.beta; fxh *= P.beta; fyl *= P.beta; fyh *= P.beta; fzl *= P.beta; fzh *= P.beta;
            double l = P.active[2] ? jinv[c] * ((fxh - fxl) * sx + (fyh - fyl) * sy + (fzh - fzl) * sz)
                                   : jinv[c] * ((fxh - fxl) * sx + (fyh - fyl) * sy);  // MAPPEDFLUXDIVERGENCE2D
            if (P.alpha != 0.0) l = P.alpha * pc + 1.0 * l;
            out[c] = (MODE == 0) ? (rhs[c] - l) : l;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_op_ortho(const Tile* __restrict__ tiles,
                                                  const PatchDesc* __restrict__ patches,
                                                  double* __restrict__ out,
                                                  const double* __restrict__ phi,
                                                  const double* __restrict__ rhs,
                                                  const double* __restrict__ jgx,
                                                  const double* __restrict__ jgy,
                                                  const double* __restrict__ jgz,
                                                  const double* __restrict__ jinv, StencilParams P,
                                                  const CopyItem* __restrict__ titems, const int* __restrict__ tstart)
{
    // (the exchange writes phi's ghost cells, as Level::exchange does on the same pointer)
    if (tstart) pull_tile_ghosts(titems, tstart, patches, const_cast<double*>(phi));
    op_ortho_body<MODE>(tiles, patches, out, phi, rhs, jgx, jgy, jgz, jinv, P, blockIdx.x, threadIdx.x, threadIdx.y);
}

// ------------------------------------------------------------------------------------
// lapDiag fill (setup).  FILLMAPPEDLAPDIAG3D
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_lapdiag(const Tile* __restrict__ tiles,
                                                 const PatchDesc* __restrict__ patches,
                                                 double* __restrict__ lap,
                                                 const double* __restrict__ jgx,
                                                 const double* __restrict__ jgy,
                                                 const double* __restrict__ jgz,
                                                 const double* __restrict__ jinv, StencilParams P)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const double s0 = 1.0 / (P.dx[0] * P.dx[0]), s1 = 1.0 / (P.dx[1] * P.dx[1]),
                 s2 = 1.0 / (P.dx[2] * P.dx[2]);
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const long long c = cidx(p, li, lj, t.k0 + kk);
            if (P.active[2])
                lap[c] = -jinv[c] * ((jgx[c + 1] + jgx[c]) * s0 + (jgy[c + p.pj] + jgy[c]) * s1 +
                                     (jgz[c + p.pk] + jgz[c]) * s2);
            else  // FILLMAPPEDLAPDIAG2D
                lap[c] = -jinv[c] * ((jgx[c + 1] + jgx[c]) * s0 + (jgy[c + p.pj] + jgy[c]) * s1);
        }
}

// ------------------------------------------------------------------------------------
// pointwise: DIAGPRECOND  phi = rhs/(alpha+beta*lapDiag);  JACOBIITER phi += 0.5*res/(..)
// ------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void diag_body(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                          double* __restrict__ phi, const double* __restrict__ r,
                                          const double* __restrict__ lapd, double alpha, double beta, int bx, int tx, int ty)
{
    const Tile t = tiles[bx];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + ty;
    const int li0 = t.i0 + 2 * tx;
    if (lj >= p.n[1]) return;
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const long long c = cidx(p, li, lj, t.k0 + kk);
            if (MODE == 0) phi[c] = r[c] / (alpha + beta * lapd[c]);
            else           phi[c] = phi[c] + 0.5 * r[c] / (alpha + beta * lapd[c]);
        }
}

template <int MODE>
__global__ __launch_bounds__(512) void k_diag(const Tile* __restrict__ tiles,
                                              const PatchDesc* __restrict__ patches,
                                              double* __restrict__ phi, const double* __restrict__ r,
                                              const double* __restrict__ lapd, double alpha, double beta)
{
    diag_body<MODE>(tiles, patches, phi, r, lapd, alpha, beta, blockIdx.x, threadIdx.x, threadIdx.y);
}

// ------------------------------------------------------------------------------------
// Restriction (tiles run over the COARSE level; patch index is shared fine<->coarse
// because the coarse layout is coarsen(fine layout)).  MAPPEDAVERAGE2, loop ii2,ii1,ii0.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_restrict(const Tile* __restrict__ ctiles,
                                                  const PatchDesc* __restrict__ cpatches,
                                                  const PatchDesc* __restrict__ fpatches,
                                                  double* __restrict__ crse,
                                                  const double* __restrict__ fine,
                                                  const double* __restrict__ fjinv, int r0, int r1, int r2)
{
    const Tile t = ctiles[blockIdx.x];
    const PatchDesc cp = cpatches[t.patch];
    const PatchDesc fp = fpatches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= cp.n[1]) return;
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= cp.n[0]) continue;
            const int lk = t.k0 + kk;
            double s = 0.0, sj = 0.0;
            for (int ii2 = 0; ii2 < r2; ++ii2)
                for (int ii1 = 0; ii1 < r1; ++ii1)
                    for (int ii0 = 0; ii0 < r0; ++ii0) {
                        const long long f = cidx(fp, li * r0 + ii0, lj * r1 + ii1, lk * r2 + ii2);
                        const double ji = fjinv[f];
                        s = s + fine[f] / ji;
                        sj = sj + 1.0 / ji;
                    }
            crse[cidx(cp, li, lj, lk)] = s / sj;
        }
}

// W(ic) = sum over children of dxProduct / Jinv(child): setup-time companion of the folded prolongation
__global__ __launch_bounds__(512) void k_child_volume(const Tile* __restrict__ ctiles,
                                                      const PatchDesc* __restrict__ cpatches,
                                                      const PatchDesc* __restrict__ fpatches, double* __restrict__ crse,
                                                      const double* __restrict__ fjinv, int r0, int r1, int r2,
                                                      double dxProduct)
{
    const Tile t = ctiles[blockIdx.x];
    const PatchDesc cp = cpatches[t.patch];
    const PatchDesc fp = fpatches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= cp.n[1]) return;
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= cp.n[0]) continue;
            const int lk = t.k0 + kk;
            double s = 0.0;
            for (int ii2 = 0; ii2 < r2; ++ii2)
                for (int ii1 = 0; ii1 < r1; ++ii1)
                    for (int ii0 = 0; ii0 < r0; ++ii0)
                        s = s + dxProduct / fjinv[cidx(fp, li * r0 + ii0, lj * r1 + ii1, lk * r2 + ii2)];
            crse[cidx(cp, li, lj, lk)] = s;
        }
}

__global__ void k_combine_sums(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b,
                               const double* __restrict__ c)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = a[0] + b[0];
        out[1] = c[0];
    }
}

// ------------------------------------------------------------------------------------
// Prolongation: fine += coarse(i/m).  AVG also emits per-block partial sums of
// dvol*fine and dvol (dvol = dxProduct/Jinv) for the zero-average variant; the mean is
// removed by k_sub_mean without the value ever visiting the host.  (Tree sum: differs
// from the reference's sequential sum in the last bits -- the one non-bit-exact step.)
// ------------------------------------------------------------------------------------
template <bool AVG>
__global__ __launch_bounds__(512) void k_prolong(const Tile* __restrict__ ftiles,
                                                 const PatchDesc* __restrict__ fpatches,
                                                 const PatchDesc* __restrict__ cpatches,
                                                 double* __restrict__ fine,
                                                 const double* __restrict__ crse,
                                                 const double* __restrict__ jinv, int r0, int r1, int r2,
                                                 double dxProduct, double* __restrict__ partials)
{
    const Tile t = ftiles[blockIdx.x];
    const PatchDesc fp = fpatches[t.patch];
    const PatchDesc cp = cpatches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    double s = 0.0, v = 0.0;
    if (lj < fp.n[1]) {
        for (int kk = 0; kk < t.nk; ++kk)
            for (int q = 0; q < 2; ++q) {
                const int li = li0 + q;
                if (li >= fp.n[0]) continue;
                const int lk = t.k0 + kk;
                const long long f = cidx(fp, li, lj, lk);
                const double nv = fine[f] + crse[cidx(cp, li / r0, lj / r1, lk / r2)];
                fine[f] = nv;
                if (AVG) {
                    const double dvol = dxProduct / jinv[f];
                    s = s + dvol * nv;
                    v = v + dvol;
                }
            }
    }
    if (AVG) {
        s = block_reduce<false>(s);
        v = block_reduce<false>(v);
        if (threadIdx.x == 0 && threadIdx.y == 0) {
            partials[2 * blockIdx.x] = s;
            partials[2 * blockIdx.x + 1] = v;
        }
    }
}

// final stage of every reduction: ONE block, fixed order => deterministic.
// nvals interleaved values per partial; op 0 = sum, 1 = max (of non-negatives), 2 = signed max.
__global__ __launch_bounds__(512) void k_reduce_final(const double* __restrict__ partials, int nparts,
                                                      int nvals, int op, double* __restrict__ out, ScalarPublish pub)
{
    double first = 0.0;
    for (int v = 0; v < nvals; ++v) {
        double acc = (op == 2) ? -1.7976931348623157e308 : 0.0;
        for (int i = threadIdx.x; i < nparts; i += 256) {
            const double x = partials[(long long)i * nvals + v];
            acc = op ? fmax(acc, x) : acc + x;
        }
        acc = op ? block_reduce<true>(acc) : block_reduce<false>(acc);
        if (threadIdx.x == 0) out[v] = acc;
        if (v == 0) first = acc;
    }
    if (threadIdx.x == 0 && pub.host_seq && nvals == 1) publish_scalars(&first, 1, pub);
}

// field[all] -= sums[0]/sums[1]  (whole allocation incl. ghosts: a_phiThisLevel[dit] -= avgPhi,
// ProlongationStrategy.cpp:160-163)
__global__ void k_sub_mean(double* __restrict__ f, long long n, const double* __restrict__ sums)
{
    const double avg = sums[0] / sums[1];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        f[i] -= avg;
}

// ------------------------------------------------------------------------------------
// coarse metrics (setup): harmonic cell average and arithmetic face average
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_avg_harmonic(const Tile* __restrict__ ctiles,
                                                      const PatchDesc* __restrict__ cpatches,
                                                      const PatchDesc* __restrict__ fpatches,
                                                      double* __restrict__ crse,
                                                      const double* __restrict__ fine, int r0, int r1, int r2)
{
    const Tile t = ctiles[blockIdx.x];
    const PatchDesc cp = cpatches[t.patch];
    const PatchDesc fp = fpatches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= cp.n[1]) return;
    const double refScale = 1.0 / (double)(r0 * r1 * r2);
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= cp.n[0]) continue;
            const int lk = t.k0 + kk;
            double s = 0.0;
            for (int ii2 = 0; ii2 < r2; ++ii2)
                for (int ii1 = 0; ii1 < r1; ++ii1)
                    for (int ii0 = 0; ii0 < r0; ++ii0)
                        s = s + 1.0 / fine[cidx(fp, li * r0 + ii0, lj * r1 + ii1, lk * r2 + ii2)];
            crse[cidx(cp, li, lj, lk)] = 1.0 / (s * refScale);
        }
}

// Face average in direction dir over the coarse face box (n[dir]+1 faces): one thread per
// coarse face.  Plain 3-D launch; setup-time only.
__global__ void k_avg_face(const PatchDesc* __restrict__ cpatches, const PatchDesc* __restrict__ fpatches,
                           int patch, double* __restrict__ crse, const double* __restrict__ fine, int dir,
                           int r0, int r1, int r2)
{
    const PatchDesc cp = cpatches[patch];
    const PatchDesc fp = fpatches[patch];
    const int li = blockIdx.x * blockDim.x + threadIdx.x;
    const int lj = blockIdx.y * blockDim.y + threadIdx.y;
    const int lk = blockIdx.z;
    const int ni = cp.n[0] + (dir == 0), nj = cp.n[1] + (dir == 1), nk = cp.n[2] + (dir == 2);
    if (li >= ni || lj >= nj || lk >= nk) return;
    const int rr[3] = {r0, r1, r2};
    const double refScale = (double)rr[dir] / (double)(r0 * r1 * r2);
    const int b0 = dir == 0 ? 1 : r0, b1 = dir == 1 ? 1 : r1, b2 = dir == 2 ? 1 : r2;
    double s = 0.0;
    for (int ii2 = 0; ii2 < b2; ++ii2)
        for (int ii1 = 0; ii1 < b1; ++ii1)
            for (int ii0 = 0; ii0 < b0; ++ii0)
                s = s + fine[cidx(fp, li * r0 + ii0, lj * r1 + ii1, lk * r2 + ii2)];
    crse[cidx(cp, li, lj, lk)] = refScale * s;
}

// ------------------------------------------------------------------------------------
// ghost exchange inside one GPU: list of box-to-box copies (Chombo Copier motion items)
// grid.x = item, grid.y = chunk of (j,k) rows
// ------------------------------------------------------------------------------------
// one cell per thread, cells of an item numbered i-fastest: a 2-cell-wide x-face (n[0] = 2) keeps all 256 lanes
// busy instead of 2 of 64 (halo copies at 128-wide boxes took longer than a sweep of the box otherwise)
__global__ void k_copy_items(const CopyItem* __restrict__ items, const PatchDesc* __restrict__ patches,
                             double* __restrict__ f)
{
    const CopyItem it = items[blockIdx.x];
    const PatchDesc sp = patches[it.src_patch];
    const PatchDesc dp = patches[it.dst_patch];
    const int n0 = it.n[0], n01 = it.n[0] * it.n[1];
    const long long cells = (long long)n01 * it.n[2];
    for (long long idx = (long long)blockIdx.y * blockDim.x + threadIdx.x; idx < cells;
         idx += (long long)gridDim.y * blockDim.x) {
        const int k = (int)(idx / n01);
        const int r = (int)(idx - (long long)k * n01);
        const int j = r / n0, i = r - j * n0;
        f[cidx(dp, it.dst_lo[0] + i, it.dst_lo[1] + j, it.dst_lo[2] + k)] =
            f[cidx(sp, it.src_lo[0] + i, it.src_lo[1] + j, it.src_lo[2] + k)];
    }
}

// pack / unpack of halo regions into contiguous send/recv buffers (multi-GPU path).
template <bool PACK>
__global__ void k_pack_items(const CopyItem* __restrict__ items, const PatchDesc* __restrict__ patches,
                             double* __restrict__ f, double* __restrict__ buf, const long long* __restrict__ bufoff)
{
    const CopyItem it = items[blockIdx.x];
    const PatchDesc pp = patches[PACK ? it.src_patch : it.dst_patch];
    const int* lo = PACK ? it.src_lo : it.dst_lo;
    const long long b0 = bufoff[blockIdx.x];
    const int n0 = it.n[0], n01 = it.n[0] * it.n[1];
    const long long cells = (long long)n01 * it.n[2];
    for (long long idx = (long long)blockIdx.y * blockDim.x + threadIdx.x; idx < cells;
         idx += (long long)gridDim.y * blockDim.x) {
        const int k = (int)(idx / n01);
        const int r = (int)(idx - (long long)k * n01);
        const int j = r / n0, i = r - j * n0;
        const long long a = cidx(pp, lo[0] + i, lo[1] + j, lo[2] + k);
        if (PACK) buf[b0 + idx] = f[a];
        else      f[a] = buf[b0 + idx];
    }
}

// ------------------------------------------------------------------------------------
// flat BLAS-1 over whole allocations (LevelDataOps semantics: whole FAB incl. ghosts)
// ------------------------------------------------------------------------------------
__global__ void k_set(double* __restrict__ a, long long n, double v)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        a[i] = v;
}
__global__ void k_copy(double* __restrict__ d, const double* __restrict__ s, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = s[i];
}
// y += a*x
__global__ void k_incr(double* __restrict__ y, const double* __restrict__ x, double a, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = y[i] + a * x[i];
}
// y += a*x; x = y   (AMRVCycle's "m_correction += dCorr; uberCorrection = m_correction" in one pass)
__global__ void k_incr_copy(double* __restrict__ y, double* __restrict__ x, double a, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double t = y[i] + a * x[i];
        y[i] = t;
        x[i] = t;
    }
}
// two independent updates in one pass: y1 += a1*x1; y2 += a2*x2   (BiCGStab's "r -= alpha v; e += alpha p~" and its omega twin)
__global__ void k_incr2(double* __restrict__ y1, const double* __restrict__ x1, double a1, double* __restrict__ y2,
                        const double* __restrict__ x2, double a2, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        y1[i] = y1[i] + a1 * x1[i];
        y2[i] = y2[i] + a2 * x2[i];
    }
}
// BiCGStab's direction update, the three LevelDataOps calls of Chombo's solver as one pass with the same roundings:
// p *= beta; p += bw * v; p += 1.0 * r
__global__ void k_bicg_p(double* __restrict__ p, const double* __restrict__ v, const double* __restrict__ r, double beta,
                         double bw, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double t = p[i] * beta;
        t = t + bw * v[i];
        t = t + 1.0 * r[i];
        p[i] = t;
    }
}
__global__ void k_scale(double* __restrict__ y, double a, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = y[i] * a;
}
// y = y * x elementwise (FArrayBox::mult)
__global__ void k_mul(double* __restrict__ y, const double* __restrict__ x, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = y[i] * x[i];
}
// z = a*x + b*y
__global__ void k_axby(double* __restrict__ z, const double* __restrict__ x, const double* __restrict__ y,
                       double a, double b, long long n)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        z[i] = a * x[i] + b * y[i];
}

// valid-cell reductions.  MODE 0: sum a*b   1: max |a|   2: sum |a|   3: signed max a
template <int MODE>
__global__ __launch_bounds__(512) void k_reduce_valid(const Tile* __restrict__ tiles,
                                                      const PatchDesc* __restrict__ patches,
                                                      const double* __restrict__ a,
                                                      const double* __restrict__ b,
                                                      double* __restrict__ partials, unsigned int* __restrict__ counter,
                                                      double* __restrict__ out, ScalarPublish pub)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    double acc = (MODE == 3) ? -1.7976931348623157e308 : 0.0;
    if (lj < p.n[1]) {
        for (int kk = 0; kk < t.nk; ++kk)
            for (int q = 0; q < 2; ++q) {
                const int li = li0 + q;
                if (li >= p.n[0]) continue;
                const long long c = cidx(p, li, lj, t.k0 + kk);
                if (MODE == 0) acc = acc + a[c] * b[c];
                else if (MODE == 1) acc = fmax(acc, fabs(a[c]));
                else if (MODE == 2) acc = acc + fabs(a[c]);
                else if (MODE == 3) acc = fmax(acc, a[c]);
                else if (MODE == 4) acc = acc + a[c] / b[c];   // sum of a*J  (b = Jinv)
                else acc = acc + 1.0 / b[c];                   // sum of J
            }
    }
    acc = (MODE == 1 || MODE == 3) ? block_reduce<true>(acc) : block_reduce<false>(acc);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x] = acc;
    if (!counter) return;   // two-launch form: k_reduce_final follows
    // Single-launch form: the LAST workgroup to arrive adds the partials up, with k_reduce_final's own arithmetic -- 256
    // (virtual) threads striding the partials, a shuffle tree per (virtual) wavefront, the four wavefront sums added in order
    // -- so the result has the same bits as the two-launch form; it then clears the counter and publishes the scalar.
    constexpr int OP = (MODE == 1) ? 1 : (MODE == 3 ? 2 : 0);
    __shared__ int isLast;
    __shared__ double sv[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    __threadfence();
    if (tid == 0) isLast = (atomicAdd(counter, 1u) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!isLast) return;
    __threadfence();
    const int lane = tid & 63, w = tid >> 6, nw = (blockDim.x * blockDim.y) >> 6;
    const int nparts = (int)gridDim.x;
    for (int q = w; q < 4; q += nw) {
        double r = (OP == 2) ? -1.7976931348623157e308 : 0.0;
        for (int i = 64 * q + lane; i < nparts; i += 256) {
            const double x = __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r = OP ? fmax(r, x) : r + x;
        }
        r = OP ? wave_max(r) : wave_sum(r);
        if (lane == 0) sv[q] = r;
    }
    __syncthreads();
    if (tid == 0) {
        double r = sv[0];
        for (int q = 1; q < 4; ++q) r = OP ? fmax(r, sv[q]) : r + sv[q];
        out[0] = r;
        *counter = 0u;
        if (pub.host_seq) publish_scalars(&r, 1, pub);
    }
}

// Reference-ordered sums for SMALL levels (one wavefront): box after box, cells in Fortran order, one running
// sum -- the order of FArrayBox::dotProduct / sumPow (Chombo) and of CONSTINTERPWITHAVGPS, so the scalars that
// steer BiCGStab and the zero-average prolongation come out bit-identical to the reference's serial run.
// (BiCGStab amplifies a last-bit difference of its dot products by ~1e8 within 20-30 iterations; the bottom
// level is a few hundred cells, where 64 dependent adds per 64 cells cost microseconds.)
// MODE 0: out[0] = sum_boxes( seqsum(a*b) )   MODE 2: out[0] = sum_boxes( seqsum(|a|) )
// MODE 6: running s += (dxProduct / b) * a, v += dxProduct / b over all boxes: out[0] = s, out[1] = v   (b = Jinv)
template <int MODE>
__global__ __launch_bounds__(64) void k_reduce_ordered(const PatchDesc* __restrict__ patches, int npatches,
                                                       const double* __restrict__ a, const double* __restrict__ b,
                                                       double dxProduct, double* __restrict__ out, ScalarPublish pub)
{
    // one wavefront: 512 cells are fetched coalesced (8 per lane) into LDS, then every lane walks them in order
    // (uniform LDS reads are broadcasts), so the dependent chain is 512 adds per chunk, not 512 cross-lane hops
    constexpr int CH = 512;
    __shared__ double X[CH], Y[MODE == 6 ? CH : 1];
    const int lane = threadIdx.x;
    double tot = 0.0, run_s = 0.0, run_v = 0.0;
    for (int pi = 0; pi < npatches; ++pi) {
        const PatchDesc p = patches[pi];
        const long long n = (long long)p.n[0] * p.n[1] * p.n[2];
        double sbox = 0.0;
        for (long long base = 0; base < n; base += CH) {
            const int cnt = (int)((n - base) < CH ? (n - base) : CH);
            __syncthreads();  // the previous chunk has been consumed
            for (int q = lane; q < cnt; q += 64) {
                const long long idx = base + q;
                const int i = (int)(idx % p.n[0]);
                const long long r = idx / p.n[0];
                const int j = (int)(r % p.n[1]), k = (int)(r / p.n[1]);
                const long long c = cidx(p, i, j, k);
                if (MODE == 0) X[q] = a[c] * b[c];
                else if (MODE == 2) X[q] = fabs(a[c]);
                else { const double y = dxProduct / b[c]; Y[q] = y; X[q] = y * a[c]; }
            }
            __syncthreads();
            // the chain of dependent adds is the critical path: fetch 16 values (uniform LDS reads) ahead of it so that
            // their latency overlaps instead of sitting between every two adds (133 -> ~20 us per 4096 cells)
            if (MODE == 6) {
                int q = 0;
                for (; q + 16 <= cnt; q += 16) {
                    double xv[16], yv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { xv[j] = X[q + j]; yv[j] = Y[q + j]; }
#pragma unroll
                    for (int j = 0; j < 16; ++j) { run_s = run_s + xv[j]; run_v = run_v + yv[j]; }
                }
                for (; q < cnt; ++q) { run_s = run_s + X[q]; run_v = run_v + Y[q]; }
            } else {
                int q = 0;
                if (base == 0) { sbox = X[0]; q = 1; }
                for (; q + 16 <= cnt; q += 16) {
                    double xv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) xv[j] = X[q + j];
#pragma unroll
                    for (int j = 0; j < 16; ++j) sbox = sbox + xv[j];
                }
                for (; q < cnt; ++q) sbox = sbox + X[q];
            }
        }
        if (MODE != 6) tot = tot + sbox;
    }
    if (lane == 0) {
        if (MODE == 6) { out[0] = run_s; out[1] = run_v; }
        else out[0] = tot;
        if (pub.host_seq && MODE != 6) publish_scalars(&tot, 1, pub);
    }
}

// The same sums (MODE 0 / 2) with the boxes' chains running SIDE BY SIDE: the reference adds the cells of one box into a
// running sum of their own (FArrayBox::dotProduct starts at zero per box) and then the box totals in layout order, so the chains
// of different boxes are independent.  One 1024-thread workgroup stages the terms of up to ORD_CAP cells in LDS (all loads in
// flight at once), thread b walks box b, then the box totals are added in order: 64 boxes of 4^3 cost 64 + 64 dependent adds
// instead of 4096 (and one load latency instead of 64: 93 -> a few us on C5's 4096-cell bottom level).  A box larger than
// ORD_CAP cells is walked chunk by chunk by one thread.  Same values in the same order as k_reduce_ordered: same bits.
constexpr int ORD_CAP = 4096;
template <int MODE>
__global__ __launch_bounds__(1024) void k_reduce_ordered_par(const PatchDesc* __restrict__ patches, int npatches,
                                                             const double* __restrict__ a, const double* __restrict__ b,
                                                             double* __restrict__ out, ScalarPublish pub)
{
    __shared__ double X[ORD_CAP + 1024];   // box q of a group starts at (cells before it) + q: odd strides, no bank pile-up
    __shared__ double S[1024];
    const int tid = threadIdx.x;
    auto term = [&](const PatchDesc& p, long long idx) {
        const int i = (int)(idx % p.n[0]);
        const long long r = idx / p.n[0];
        const int j = (int)(r % p.n[1]), k = (int)(r / p.n[1]);
        const long long c = cidx(p, i, j, k);
        return MODE == 0 ? a[c] * b[c] : fabs(a[c]);
    };
    auto chain = [&](const double* x, int cnt, double s, bool first) {
        int q = 0;
        if (first) { s = x[0]; q = 1; }
        for (; q + 16 <= cnt; q += 16) {
            double xv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = x[q + j];
#pragma unroll
            for (int j = 0; j < 16; ++j) s = s + xv[j];
        }
        for (; q < cnt; ++q) s = s + x[q];
        return s;
    };
    __shared__ int CNT[1024];              // cells of the next (up to) 1024 boxes: the patch table is read once, all loads in flight
    double tot = 0.0;
    int pi = 0;
    while (pi < npatches) {
        const int nb = min(npatches - pi, 1024);
        __syncthreads();   // X, S and CNT of the previous group have been consumed
        if (tid < nb) {
            const PatchDesc q = patches[pi + tid];
            const long long nq = (long long)q.n[0] * q.n[1] * q.n[2];
            CNT[tid] = nq > ORD_CAP ? ORD_CAP + 1 : (int)nq;
        }
        __syncthreads();
        if (CNT[0] > ORD_CAP) {   // one large box: chunks of ORD_CAP cells, one chain
            const PatchDesc p0 = patches[pi];
            const long long n0 = (long long)p0.n[0] * p0.n[1] * p0.n[2];
            double sbox = 0.0;
            for (long long base = 0; base < n0; base += ORD_CAP) {
                const int cnt = (int)((n0 - base) < ORD_CAP ? (n0 - base) : ORD_CAP);
                __syncthreads();
                for (int q = tid; q < cnt; q += 1024) X[q] = term(p0, base + q);
                __syncthreads();
                sbox = chain(X, cnt, sbox, base == 0);   // every thread walks it (uniform LDS reads are broadcasts)
            }
            tot = tot + sbox;
            ++pi;
            continue;
        }
        // a group of consecutive boxes that fits: [pi, pi + G)
        int G = 0, cells = 0;
        while (G < nb && cells + CNT[G] <= ORD_CAP) { cells += CNT[G]; ++G; }
        const int tpb = 1024 / G;                 // threads staging one box
        const int mb = tid / tpb, sub = tid - mb * tpb;
        int myoff = 0, mycnt = 0, stoff = 0;
        {
            int off = 0;
            for (int q = 0; q < G; ++q) {         // uniform LDS reads
                if (q == mb) stoff = off + q;
                if (q == tid) { myoff = off + q; mycnt = CNT[q]; }
                off += CNT[q];
            }
        }
        if (mb < G) {
            const PatchDesc pq = patches[pi + mb];
            const int nq = CNT[mb];
            for (int idx = sub; idx < nq; idx += tpb) X[stoff + idx] = term(pq, idx);
        }
        __syncthreads();
        if (tid < G) S[tid] = chain(X + myoff, mycnt, 0.0, true);
        __syncthreads();
        for (int q = 0; q < G; ++q) tot = tot + S[q];
        pi += G;
    }
    if (tid == 0) {
        out[0] = tot;
        if (pub.host_seq) publish_scalars(&tot, 1, pub);
    }
}

// ---- the same serial sums on a SHARDED small level -------------------------------------------------------------
// Every rank writes the per-cell terms of ITS boxes at their position in the serial (box after box, Fortran order)
// sequence, zeros elsewhere; a sum-allreduce of that vector (x + 0 is exact) hands every rank the whole sequence,
// which one wavefront then walks exactly as k_reduce_ordered does.  <= 4096 cells: a 32-64 KB message.
template <int MODE>
__global__ __launch_bounds__(256) void k_ord_fill(const PatchDesc* __restrict__ patches, const long long* __restrict__ start,
                                                  const double* __restrict__ a, const double* __restrict__ b,
                                                  double dxProduct, double* __restrict__ X, double* __restrict__ Y)
{
    const PatchDesc p = patches[blockIdx.x];
    const long long n = (long long)p.n[0] * p.n[1] * p.n[2], s0 = start[blockIdx.x];
    for (long long idx = threadIdx.x; idx < n; idx += 256) {
        const int i = (int)(idx % p.n[0]);
        const long long r = idx / p.n[0];
        const int j = (int)(r % p.n[1]), k = (int)(r / p.n[1]);
        const long long c = cidx(p, i, j, k);
        if (MODE == 0) X[s0 + idx] = a[c] * b[c];
        else if (MODE == 2) X[s0 + idx] = fabs(a[c]);
        else { const double y = dxProduct / b[c]; Y[s0 + idx] = y; X[s0 + idx] = y * a[c]; }
    }
}
template <int MODE>
__global__ __launch_bounds__(64) void k_reduce_ordered_flat(int nboxes, const long long* __restrict__ box_start,
                                                            const double* __restrict__ Xg, const double* __restrict__ Yg,
                                                            double* __restrict__ out)
{
    constexpr int CH = 512;
    __shared__ double X[CH], Y[MODE == 6 ? CH : 1];
    const int lane = threadIdx.x;
    double tot = 0.0, run_s = 0.0, run_v = 0.0;
    for (int pi = 0; pi < nboxes; ++pi) {
        const long long s0 = box_start[pi], n = box_start[pi + 1] - s0;
        double sbox = 0.0;
        for (long long base = 0; base < n; base += CH) {
            const int cnt = (int)((n - base) < CH ? (n - base) : CH);
            __syncthreads();
            for (int q = lane; q < cnt; q += 64) {
                X[q] = Xg[s0 + base + q];
                if (MODE == 6) Y[q] = Yg[s0 + base + q];
            }
            __syncthreads();
            if (MODE == 6) {
                for (int q = 0; q < cnt; ++q) { run_s = run_s + X[q]; run_v = run_v + Y[q]; }
            } else {
                int q = 0;
                if (base == 0) { sbox = X[0]; q = 1; }
                for (; q < cnt; ++q) sbox = sbox + X[q];
            }
        }
        if (MODE != 6) tot = tot + sbox;
    }
    if (lane == 0) {
        if (MODE == 6) { out[0] = run_s; out[1] = run_v; }
        else out[0] = tot;
    }
}

// splitmix64(cell index, seed) -> uniform(-1,1): same integer recipe as
// oracle/somar_oracle.py::hash_uniform; used by bench/tests for device-side synthetic fills.
__global__ __launch_bounds__(512) void k_fill_hash(const Tile* __restrict__ tiles,
                                                   const PatchDesc* __restrict__ patches,
                                                   double* __restrict__ f, StencilParams P,
                                                   unsigned long long seed)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const unsigned long long n0 = P.dom_hi[0] - P.dom_lo[0] + 1, n1 = P.dom_hi[1] - P.dom_lo[1] + 1;
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const int lk = t.k0 + kk;
            const unsigned long long I = p.lo[0] + li - P.dom_lo[0], J = p.lo[1] + lj - P.dom_lo[1],
                                     K = p.lo[2] + lk - P.dom_lo[2];
            unsigned long long z = (I + n0 * (J + n1 * K)) + seed * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z = z ^ (z >> 31);
            f[cidx(p, li, lj, lk)] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        }
}

// ------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------
static inline dim3 tile_block(const LevelDev& L) { return dim3(64, L.tile_j, 1); }
static inline int flat_grid(long long n)
{
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

void launch_gsrb_ortho(hipStream_t st, const LevelDev& L, double* phi, const double* rhs, int color, int loose, bool pull)
{
    if (L.ntiles == 0) return;
    const bool pl = pull && L.tile_item_start && loose == 0;
    hipLaunchKernelGGL(k_gsrb_ortho, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, phi, rhs,
                       L.jg[0], L.jg[1], L.jg[2], L.jinv, L.lapdiag, L.P, color, loose, pl ? L.tile_items : nullptr,
                       pl ? L.tile_item_start : nullptr);
}
void launch_op_ortho(hipStream_t st, const LevelDev& L, double* out, const double* phi, const double* rhs, int mode, bool pull)
{
    if (L.ntiles == 0) return;
    const bool pl = pull && L.tile_item_start;
    const CopyItem* ti = pl ? L.tile_items : nullptr;
    const int* ts = pl ? L.tile_item_start : nullptr;
    if (mode == 0)
        hipLaunchKernelGGL(k_op_ortho<0>, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, out, phi,
                           rhs, L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, ti, ts);
    else
        hipLaunchKernelGGL(k_op_ortho<1>, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, out, phi,
                           rhs, L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, ti, ts);
}
void launch_lapdiag(hipStream_t st, const LevelDev& L)
{
    if (L.ntiles == 0) return;
    hipLaunchKernelGGL(k_lapdiag, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, L.lapdiag, L.jg[0],
                       L.jg[1], L.jg[2], L.jinv, L.P);
}
void launch_diag(hipStream_t st, const LevelDev& L, double* phi, const double* r, int mode)
{
    if (L.ntiles == 0) return;
    if (mode == 0)
        hipLaunchKernelGGL(k_diag<0>, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, phi, r, L.lapdiag,
                           L.P.alpha, L.P.beta);
    else
        hipLaunchKernelGGL(k_diag<1>, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, phi, r, L.lapdiag,
                           L.P.alpha, L.P.beta);
}
void launch_restrict(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine,
                     const int r[3])
{
    if (C.ntiles == 0) return;
    hipLaunchKernelGGL(k_restrict, dim3(C.ntiles), tile_block(C), 0, st, C.tiles, C.patches, F.patches, crse, fine,
                       F.jinv, r[0], r[1], r[2]);
}
void launch_prolong(hipStream_t st, const LevelDev& F, const LevelDev& C, double* fine, const double* crse,
                    const int r[3], bool zeroAvg, double dxProduct, double* partials, double* sums,
                    long long fieldElems, bool ordered)
{
    if (F.ntiles == 0) return;
    if (zeroAvg && ordered) {
        hipLaunchKernelGGL(k_prolong<false>, dim3(F.ntiles), tile_block(F), 0, st, F.tiles, F.patches, C.patches,
                           fine, crse, F.jinv, r[0], r[1], r[2], dxProduct, partials);
        hipLaunchKernelGGL(k_reduce_ordered<6>, dim3(1), dim3(64), 0, st, F.patches, F.npatches, fine, F.jinv, dxProduct,
                           sums, ScalarPublish{nullptr, nullptr, 0ull});
        return;
    }
    if (!zeroAvg) {
        hipLaunchKernelGGL(k_prolong<false>, dim3(F.ntiles), tile_block(F), 0, st, F.tiles, F.patches, C.patches,
                           fine, crse, F.jinv, r[0], r[1], r[2], dxProduct, partials);
    } else {
        hipLaunchKernelGGL(k_prolong<true>, dim3(F.ntiles), tile_block(F), 0, st, F.tiles, F.patches, C.patches,
                           fine, crse, F.jinv, r[0], r[1], r[2], dxProduct, partials);
        hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, F.ntiles, 2, 0, sums, ScalarPublish{nullptr, nullptr, 0ull});
    }
    (void)fieldElems;
}
void launch_child_volume(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const int r[3],
                         double dxProduct)
{
    if (C.ntiles == 0) return;
    hipLaunchKernelGGL(k_child_volume, dim3(C.ntiles), tile_block(C), 0, st, C.tiles, C.patches, F.patches, crse, F.jinv,
                       r[0], r[1], r[2], dxProduct);
}
void launch_sum_partials(hipStream_t st, const double* partials, int n, double* out)
{
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, n, 1, 0, out, ScalarPublish{nullptr, nullptr, 0ull});
}
void launch_combine_sums(hipStream_t st, double* out, const double* a, const double* b, const double* c)
{
    hipLaunchKernelGGL(k_combine_sums, dim3(1), dim3(64), 0, st, out, a, b, c);
}
void launch_sub_mean(hipStream_t st, double* f, long long n, const double* sums)
{
    hipLaunchKernelGGL(k_sub_mean, dim3(flat_grid(n)), dim3(256), 0, st, f, n, sums);
}
void launch_avg_harmonic(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine,
                         const int r[3])
{
    if (C.ntiles == 0) return;
    hipLaunchKernelGGL(k_avg_harmonic, dim3(C.ntiles), tile_block(C), 0, st, C.tiles, C.patches, F.patches, crse,
                       fine, r[0], r[1], r[2]);
}
void launch_avg_face(hipStream_t st, const LevelDev& C, const LevelDev& F, int patch, const int cn[3], double* crse,
                     const double* fine, int dir, const int r[3])
{
    const int ni = cn[0] + (dir == 0), nj = cn[1] + (dir == 1), nk = cn[2] + (dir == 2);
    dim3 b(64, 4, 1), g((ni + 63) / 64, (nj + 3) / 4, nk);
    hipLaunchKernelGGL(k_avg_face, g, b, 0, st, C.patches, F.patches, patch, crse, fine, dir, r[0], r[1], r[2]);
}
void launch_copy_items(hipStream_t st, const LevelDev& L, const CopyItem* items, int nitems, double* f)
{
    if (nitems == 0) return;
    hipLaunchKernelGGL(k_copy_items, dim3(nitems, 16), dim3(256), 0, st, items, L.patches, f);
}
void launch_pack(hipStream_t st, const LevelDev& L, const CopyItem* items, const long long* bufoff, int nitems,
                 double* f, double* buf, bool pack)
{
    if (nitems == 0) return;
    if (pack)
        hipLaunchKernelGGL(k_pack_items<true>, dim3(nitems, 16), dim3(256), 0, st, items, L.patches, f, buf, bufoff);
    else
        hipLaunchKernelGGL(k_pack_items<false>, dim3(nitems, 16), dim3(256), 0, st, items, L.patches, f, buf, bufoff);
}
void launch_set(hipStream_t st, double* a, long long n, double v)
{
    hipLaunchKernelGGL(k_set, dim3(flat_grid(n)), dim3(256), 0, st, a, n, v);
}
void launch_copy(hipStream_t st, double* d, const double* s, long long n)
{
    hipLaunchKernelGGL(k_copy, dim3(flat_grid(n)), dim3(256), 0, st, d, s, n);
}
__global__ void k_publish(const double* __restrict__ src, int n, double* host_dst, unsigned long long* host_seq,
                          unsigned long long seq)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < n; ++i) __hip_atomic_store(host_dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void launch_publish(hipStream_t st, const double* src, int n, double* host_dst, unsigned long long* host_seq,
                    unsigned long long seq)
{
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, src, n, host_dst, host_seq, seq);
}
void launch_incr(hipStream_t st, double* y, const double* x, double a, long long n)
{
    hipLaunchKernelGGL(k_incr, dim3(flat_grid(n)), dim3(256), 0, st, y, x, a, n);
}
void launch_incr_copy(hipStream_t st, double* y, double* x, double a, long long n)
{
    hipLaunchKernelGGL(k_incr_copy, dim3(flat_grid(n)), dim3(256), 0, st, y, x, a, n);
}
void launch_incr2(hipStream_t st, double* y1, const double* x1, double a1, double* y2, const double* x2, double a2, long long n)
{
    hipLaunchKernelGGL(k_incr2, dim3(flat_grid(n)), dim3(256), 0, st, y1, x1, a1, y2, x2, a2, n);
}
void launch_bicg_p(hipStream_t st, double* p, const double* v, const double* r, double beta, double bw, long long n)
{
    hipLaunchKernelGGL(k_bicg_p, dim3(flat_grid(n)), dim3(256), 0, st, p, v, r, beta, bw, n);
}
void launch_scale(hipStream_t st, double* y, double a, long long n)
{
    hipLaunchKernelGGL(k_scale, dim3(flat_grid(n)), dim3(256), 0, st, y, a, n);
}
void launch_mul(hipStream_t st, double* y, const double* x, long long n)
{
    hipLaunchKernelGGL(k_mul, dim3(flat_grid(n)), dim3(256), 0, st, y, x, n);
}
void launch_axby(hipStream_t st, double* z, const double* x, const double* y, double a, double b, long long n)
{
    hipLaunchKernelGGL(k_axby, dim3(flat_grid(n)), dim3(256), 0, st, z, x, y, a, b, n);
}
void launch_reduce(hipStream_t st, const LevelDev& L, const double* a, const double* b, int mode, double* partials,
                   double* out, bool ordered, const ScalarPublish* pub)
{
    const ScalarPublish P = pub ? *pub : ScalarPublish{nullptr, nullptr, 0ull};
    if (ordered && (mode == 0 || mode == 2) && L.ntiles > 0) {
        // SOMAR_ORDERED_SERIAL=1: the one-wavefront walk over all boxes (A/B switch; same bits)
        static const bool serial = getenv("SOMAR_ORDERED_SERIAL") != nullptr;
        if (serial) {
            if (mode == 0)
                hipLaunchKernelGGL(k_reduce_ordered<0>, dim3(1), dim3(64), 0, st, L.patches, L.npatches, a, b, 0.0, out, P);
            else
                hipLaunchKernelGGL(k_reduce_ordered<2>, dim3(1), dim3(64), 0, st, L.patches, L.npatches, a, b, 0.0, out, P);
        } else if (mode == 0) {
            hipLaunchKernelGGL(k_reduce_ordered_par<0>, dim3(1), dim3(1024), 0, st, L.patches, L.npatches, a, b, out, P);
        } else {
            hipLaunchKernelGGL(k_reduce_ordered_par<2>, dim3(1), dim3(1024), 0, st, L.patches, L.npatches, a, b, out, P);
        }
        return;
    }
    if (L.ntiles == 0) {
        hipLaunchKernelGGL(k_set, dim3(1), dim3(64), 0, st, out, 1, 0.0);
        if (pub) launch_publish(st, out, 1, pub->host_dst, pub->host_seq, pub->seq);
        return;
    }
    // EXPERIMENT, off by default (SOMAR_ONE_LAUNCH_REDUCE=1): the last workgroup to arrive finishes the reduction, one launch
    // instead of two.  Bit-identical, but measured slower on every workload -- each workgroup pays a device-scope fence and an
    // atomic on one counter, and the finishing workgroup's serial tail is longer than a launch: C3 19.4 -> 23.8 ms, C2 95.1 ->
    // 92.3 V-cycles/s on grids <= 512 tiles; on million-tile levels far worse (C4 147 -> 201 ms).
    static const bool one = getenv("SOMAR_ONE_LAUNCH_REDUCE") != nullptr;
    unsigned int* cnt = (!one || L.ntiles > 512) ? nullptr : L.red_counter;
#define SOMAR_RV(M)                                                                                                           \
    hipLaunchKernelGGL(k_reduce_valid<M>, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, a, b, partials, cnt, out, \
                       cnt ? P : ScalarPublish{nullptr, nullptr, 0ull})
    if (mode == 0) SOMAR_RV(0);
    else if (mode == 1) SOMAR_RV(1);
    else if (mode == 2) SOMAR_RV(2);
    else if (mode == 3) SOMAR_RV(3);
    else if (mode == 4) SOMAR_RV(4);
    else SOMAR_RV(5);
#undef SOMAR_RV
    if (cnt) return;
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, L.ntiles, 1,
                       mode == 1 ? 1 : (mode == 3 ? 2 : 0), out, P);
}
void launch_ord_fill(hipStream_t st, const LevelDev& L, const long long* start, const double* a, const double* b,
                     int mode, double dxProduct, double* X, double* Y)
{
    if (L.npatches == 0) return;
    if (mode == 0) hipLaunchKernelGGL(k_ord_fill<0>, dim3(L.npatches), dim3(256), 0, st, L.patches, start, a, b, dxProduct, X, Y);
    else if (mode == 2) hipLaunchKernelGGL(k_ord_fill<2>, dim3(L.npatches), dim3(256), 0, st, L.patches, start, a, b, dxProduct, X, Y);
    else hipLaunchKernelGGL(k_ord_fill<6>, dim3(L.npatches), dim3(256), 0, st, L.patches, start, a, b, dxProduct, X, Y);
}
void launch_reduce_ordered_flat(hipStream_t st, int nboxes, const long long* box_start, const double* X, const double* Y,
                                int mode, double* out)
{
    if (mode == 0) hipLaunchKernelGGL(k_reduce_ordered_flat<0>, dim3(1), dim3(64), 0, st, nboxes, box_start, X, Y, out);
    else if (mode == 2) hipLaunchKernelGGL(k_reduce_ordered_flat<2>, dim3(1), dim3(64), 0, st, nboxes, box_start, X, Y, out);
    else hipLaunchKernelGGL(k_reduce_ordered_flat<6>, dim3(1), dim3(64), 0, st, nboxes, box_start, X, Y, out);
}
void launch_fill_hash(hipStream_t st, const LevelDev& L, double* f, unsigned long long seed)
{
    if (L.ntiles == 0) return;
    hipLaunchKernelGGL(k_fill_hash, dim3(L.ntiles), tile_block(L), 0, st, L.tiles, L.patches, f, L.P, seed);
}

// min / max of one coefficient array over the valid cells (dir < 0) or the valid dir-faces of every patch, MM_CH chunks of
// k-planes per patch: out[2 * (patch * MM_CH + chunk)] = min, [... + 1] = max (+-inf for an empty chunk).  Runs once per depth
// at finalize (detect_uniform_metric).
__global__ __launch_bounds__(256) void k_minmax_valid(const PatchDesc* __restrict__ patches, const double* __restrict__ a, int dir,
                                                      double* __restrict__ out)
{
    const PatchDesc p = patches[blockIdx.x];
    const int n0 = p.n[0] + (dir == 0), n1 = p.n[1] + (dir == 1), n2 = p.n[2] + (dir == 2);
    const int per = (n2 + MM_CH - 1) / MM_CH;
    const int k0 = per * blockIdx.y, k1 = k0 + per < n2 ? k0 + per : n2;
    double lo = HUGE_VAL, hi = -HUGE_VAL;
    for (int k = k0; k < k1; ++k)
        for (int j = threadIdx.y; j < n1; j += 4) {
            const double* row = a + p.off + (long long)p.pj * j + p.pk * k;
            for (int i = threadIdx.x; i < n0; i += 64) {
                const double v = row[i];
                lo = v < lo ? v : lo;
                hi = v > hi ? v : hi;
            }
        }
    __shared__ double slo[256], shi[256];
    const int t = threadIdx.x + 64 * threadIdx.y;
    slo[t] = lo;
    shi[t] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            slo[t] = slo[t + s] < slo[t] ? slo[t + s] : slo[t];
            shi[t] = shi[t + s] > shi[t] ? shi[t + s] : shi[t];
        }
        __syncthreads();
    }
    if (t == 0) {
        const long long o = 2 * ((long long)blockIdx.x * MM_CH + blockIdx.y);
        out[o] = slo[0];
        out[o + 1] = shi[0];
    }
}
void launch_minmax_valid(hipStream_t st, const LevelDev& L, const double* a, int dir, double* out)
{
    if (L.npatches) hipLaunchKernelGGL(k_minmax_valid, dim3(L.npatches, MM_CH), dim3(64, 4), 0, st, L.patches, a, dir, out);
}

// ---- the bottom solver of a tiny level in ONE launch -----------------------------------------------------------------------------
// Chombo 3.1 BiCGStabSolver<T>::solve as PressureSolver::bottom_solve restates it (solver.cpp), run by one 1024-thread
// workgroup: the same operations in the same order -- residual, DIAGPRECOND + point-GSRB sweeps, operator, the vector updates
// (k_incr / k_incr2 / k_bicg_p's expressions), dot products and norms summed in the reference's serial order exactly as
// k_reduce_ordered sums them -- with __syncthreads() where the kernel boundaries were and every scalar (rho, alpha, omega,
// the stopping tests) computed redundantly by every thread from the same reduced values, so the control flow is uniform
// without a broadcast.  On BASELINE C2 the eight iterations on the 4^3 bottom box were about 100 launches and 25 host round
// trips per V-cycle; this is one launch and one round trip.  Bit-identical to the launch-by-launch path by construction
// (tests run both).  w: r, r~, e, p, p~, s~, t, v.
__global__ __launch_bounds__(1024) void k_tiny_bicgstab(TinyBicg A)
{
    __shared__ double X[512];
    __shared__ double M[16];
    const int tid = threadIdx.x;
    const int vper = 64 * A.tile_j;
    const int nvb = 1024 / vper;
    const int vb = tid / vper, vt = tid - vb * vper;
    const int tx = vt & 63, ty = vt >> 6;
    const long long n = A.field_elems;
    auto sync = [&]() { __threadfence_block(); __syncthreads(); };
    auto exchange = [&](double* f) {
        for (int it = vb; it < A.nitems; it += nvb) {
            const CopyItem ci = A.items[it];
            const PatchDesc sp = A.patches[ci.src_patch];
            const PatchDesc dp = A.patches[ci.dst_patch];
            const int n0 = ci.n[0], n01 = ci.n[0] * ci.n[1];
            const int cells = n01 * ci.n[2];
            for (int idx = vt; idx < cells; idx += vper) {
                const int k = idx / n01;
                const int r = idx - k * n01;
                const int j = r / n0, i = r - j * n0;
                f[cidx(dp, ci.dst_lo[0] + i, ci.dst_lo[1] + j, ci.dst_lo[2] + k)] =
                    f[cidx(sp, ci.src_lo[0] + i, ci.src_lo[1] + j, ci.src_lo[2] + k)];
            }
        }
        sync();
    };
    auto gsrb = [&](double* phi, const double* rhs, int color) {
        for (int b = vb; b < A.ntiles; b += nvb)
            gsrb_ortho_body(A.tiles, A.patches, phi, rhs, A.jg[0], A.jg[1], A.jg[2], A.jinv, A.lapd, A.P, color, 0, b, tx, ty);
        sync();
    };
    auto op0 = [&](double* out, const double* phi, const double* rhs) {   // rhs - L[phi]
        for (int b = vb; b < A.ntiles; b += nvb)
            op_ortho_body<0>(A.tiles, A.patches, out, phi, rhs, A.jg[0], A.jg[1], A.jg[2], A.jinv, A.P, b, tx, ty);
        sync();
    };
    auto op1 = [&](double* out, const double* phi) {                       // L[phi]
        for (int b = vb; b < A.ntiles; b += nvb)
            op_ortho_body<1>(A.tiles, A.patches, out, phi, phi, A.jg[0], A.jg[1], A.jg[2], A.jinv, A.P, b, tx, ty);
        sync();
    };
    auto setv = [&](double* f, double a) { for (long long i = tid; i < n; i += 1024) f[i] = a; sync(); };
    auto copy = [&](double* y, const double* x) { for (long long i = tid; i < n; i += 1024) y[i] = x[i]; sync(); };
    auto incr = [&](double* y, const double* x, double a) { for (long long i = tid; i < n; i += 1024) y[i] = y[i] + a * x[i]; sync(); };
    auto incr2 = [&](double* y1, const double* x1, double a1, double* y2, const double* x2, double a2) {
        for (long long i = tid; i < n; i += 1024) {
            y1[i] = y1[i] + a1 * x1[i];
            y2[i] = y2[i] + a2 * x2[i];
        }
        sync();
    };
    // serial-order sum over the valid cells, box after box, Fortran order (k_reduce_ordered): mode 0 a*b, mode 2 |a|
    auto ordsum = [&](const double* a, const double* b, int mode) {
        double tot = 0.0;
        for (int pi = 0; pi < A.npatches; ++pi) {
            const PatchDesc p = A.patches[pi];
            const long long cells = (long long)p.n[0] * p.n[1] * p.n[2];
            double sbox = 0.0;
            for (long long base = 0; base < cells; base += 512) {
                const int cnt = (int)((cells - base) < 512 ? (cells - base) : 512);
                __syncthreads();
                for (int q = tid; q < cnt; q += 1024) {
                    const long long idx = base + q;
                    const int i = (int)(idx % p.n[0]);
                    const long long r = idx / p.n[0];
                    const int j = (int)(r % p.n[1]), k = (int)(r / p.n[1]);
                    const long long c = cidx(p, i, j, k);
                    X[q] = mode == 0 ? a[c] * b[c] : fabs(a[c]);
                }
                __syncthreads();
                int q = 0;
                if (base == 0) { sbox = X[0]; q = 1; }
                for (; q + 16 <= cnt; q += 16) {
                    double xv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) xv[j] = X[q + j];
#pragma unroll
                    for (int j = 0; j < 16; ++j) sbox = sbox + xv[j];
                }
                for (; q < cnt; ++q) sbox = sbox + X[q];
            }
            tot = tot + sbox;
        }
        __syncthreads();
        return tot;
    };
    auto maxabs = [&](const double* a) {
        double m = 0.0;
        for (int pi = 0; pi < A.npatches; ++pi) {
            const PatchDesc p = A.patches[pi];
            const long long cells = (long long)p.n[0] * p.n[1] * p.n[2];
            for (long long idx = tid; idx < cells; idx += 1024) {
                const int i = (int)(idx % p.n[0]);
                const long long r = idx / p.n[0];
                const int j = (int)(r % p.n[1]), k = (int)(r / p.n[1]);
                const double v = fabs(a[cidx(p, i, j, k)]);
                m = v > m ? v : m;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double w = __shfl_down(m, o, 64);
            m = w > m ? w : m;
        }
        __syncthreads();
        if ((tid & 63) == 0) M[tid >> 6] = m;
        __syncthreads();
        double r = M[0];
        for (int w = 1; w < 16; ++w) r = M[w] > r ? M[w] : r;
        __syncthreads();
        return r;
    };
    auto norm = [&](const double* a) {
        if (A.normType == 0) return maxabs(a);
        if (A.normType == 1) return ordsum(a, a, 2);
        return sqrt(ordsum(a, a, 0));
    };
    auto residual = [&](double* out, double* phi, const double* rhs) { exchange(phi); op0(out, phi, rhs); };
    auto apply_op = [&](double* out, double* phi) { exchange(phi); op1(out, phi); };
    auto pre_cond = [&](double* phi, const double* rhs) {
        if (A.precondIters <= 0) { copy(phi, rhs); return; }
        for (int b = vb; b < A.ntiles; b += nvb) diag_body<0>(A.tiles, A.patches, phi, rhs, A.lapd, A.P.alpha, A.P.beta, b, tx, ty);
        sync();
        for (int it = 0; it < A.precondIters; ++it)
            for (int pass = 0; pass < 2; ++pass) { exchange(phi); gsrb(phi, rhs, pass); }
    };
    double *phi = A.phi, *r = A.w[0], *r_tilde = A.w[1], *e = A.w[2], *p = A.w[3], *p_tilde = A.w[4], *s_tilde = A.w[5],
           *t = A.w[6], *v = A.w[7];
    const double* rhs = A.rhs;
    auto finish = [&](int iters, int exit_code) {
        if (tid == 0) {
            const double vals[2] = {(double)iters, (double)exit_code};
            A.info[0] = vals[0];
            A.info[1] = vals[1];
            publish_scalars(vals, 2, A.pub);
        }
    };

    int recount = 0;
    residual(r, phi, rhs);
    copy(r_tilde, r);
    setv(e, 0.0);
    setv(p_tilde, 0.0);
    setv(s_tilde, 0.0);
    int i = 0;
    double rho[4] = {0, 0, 0, 0};
    double nrm[2];
    nrm[0] = norm(r);
    double initial_norm = nrm[0];
    const double initial_rnorm = nrm[0];
    nrm[1] = nrm[0];
    double alpha[2] = {0, 0}, beta[2] = {0, 0}, omega[2] = {0, 0};
    bool init = true;
    int restarts = 0;
    if (A.metric > 0) initial_norm = A.metric;
    const double eps = A.eps;
    int bottom_exit = -1;
    while ((i < A.imax && nrm[0] > eps * nrm[1]) && (nrm[1] > 0)) {
        ++i;
        nrm[1] = nrm[0];
        alpha[1] = alpha[0]; beta[1] = beta[0]; omega[1] = omega[0];
        rho[3] = rho[2]; rho[2] = rho[1];
        rho[1] = ordsum(r_tilde, r, 0);
        if (rho[1] == 0.0) {
            incr(phi, e, 1.0);
            finish(i, 2);
            return;
        }
        if (init) {
            copy(p, r);
            init = false;
        } else {
            beta[1] = (rho[1] / rho[2]) * (alpha[1] / omega[1]);
            const double bt = beta[1], bw = -beta[1] * omega[1];
            for (long long q = tid; q < n; q += 1024) {
                double u = p[q] * bt;
                u = u + bw * v[q];
                u = u + 1.0 * r[q];
                p[q] = u;
            }
            sync();
        }
        pre_cond(p_tilde, p);
        apply_op(v, p_tilde);
        const double m = ordsum(r_tilde, v, 0);
        alpha[0] = rho[1] / m;
        if (fabs(m) > A.small * fabs(rho[1])) {
            incr2(r, v, -alpha[0], e, p_tilde, alpha[0]);
            nrm[0] = norm(r);
        } else {
            setv(r, 0.0);
            nrm[0] = 0.0;
        }
        if (nrm[0] > eps * initial_norm && nrm[0] > A.reps * initial_rnorm) {
            pre_cond(s_tilde, r);
            apply_op(t, s_tilde);
            const double tr = ordsum(t, r, 0), tt = ordsum(t, t, 0);
            omega[0] = tr / tt;
            incr2(e, s_tilde, omega[0], r, t, -omega[0]);
            nrm[0] = norm(r);
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
                incr(phi, e, 1.0);
                if (restarts == A.numRestarts) {
                    finish(i, 3);
                    return;
                }
                residual(r, phi, rhs);
                nrm[0] = norm(r);
                rho[0] = rho[1] = rho[2] = rho[3] = 0.0;
                alpha[0] = beta[0] = omega[0] = 0.0;
                copy(r_tilde, r);
                setv(e, 0.0);
                ++restarts;
                init = true;
            }
        }
    }
    incr(phi, e, 1.0);
    finish(i, bottom_exit);
}
void launch_tiny_bicgstab(hipStream_t st, const LevelDev& L, const CopyItem* items, int nitems, long long field_elems,
                          TinyBicg A)
{
    A.tiles = L.tiles; A.ntiles = L.ntiles; A.tile_j = L.tile_j; A.patches = L.patches; A.npatches = L.npatches;
    A.items = items; A.nitems = nitems; A.field_elems = field_elems;
    for (int d = 0; d < 3; ++d) A.jg[d] = L.jg[d];
    A.jinv = L.jinv; A.lapd = L.lapdiag; A.P = L.P;
    hipLaunchKernelGGL(k_tiny_bicgstab, dim3(1), dim3(1024), 0, st, A);
}

// ---- what this device streams (somar_diag_stream_probe) -----------------------------------------------------------------------
// kind 0: copy (1 read + 1 write stream); 1: read only; 2: the fused sweep's mix, 6 read streams + 1 write stream, no stencil,
// no halo -- the ceiling the sweep's achieved bandwidth is to be read against (tools/bw_probe.hip measures more variants)
typedef double v2d_probe __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k_stream_probe(const v2d_probe* __restrict__ a, const v2d_probe* __restrict__ b,
                                                      const v2d_probe* __restrict__ c, const v2d_probe* __restrict__ d,
                                                      const v2d_probe* __restrict__ e, const v2d_probe* __restrict__ f,
                                                      v2d_probe* __restrict__ o, long long n)
{
    v2d_probe acc = {0.0, 0.0};
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
        if (KIND == 0) o[i] = a[i];
        else if (KIND == 1) acc += a[i];
        else o[i] = a[i] + b[i] + c[i] + d[i] + e[i] + f[i];
    }
    if (KIND == 1 && acc.x + acc.y == 1.2345e300) o[0] = acc;
}
void launch_stream_probe(hipStream_t st, int kind, int workgroups, double* const* in6, double* out, long long cells)
{
    const v2d_probe* p[6];
    for (int q = 0; q < 6; ++q) p[q] = reinterpret_cast<const v2d_probe*>(in6[q]);
    v2d_probe* o = reinterpret_cast<v2d_probe*>(out);
    const long long n = cells / 2;
    if (kind == 0) hipLaunchKernelGGL(k_stream_probe<0>, dim3(workgroups), dim3(256), 0, st, p[0], p[1], p[2], p[3], p[4], p[5], o, n);
    else if (kind == 1) hipLaunchKernelGGL(k_stream_probe<1>, dim3(workgroups), dim3(256), 0, st, p[0], p[1], p[2], p[3], p[4], p[5], o, n);
    else hipLaunchKernelGGL(k_stream_probe<2>, dim3(workgroups), dim3(256), 0, st, p[0], p[1], p[2], p[3], p[4], p[5], o, n);
}

}  // namespace somar
