// somar_amd/csrc/amr_kernels.hip -- table-driven kernels for everything that crosses AMR levels.
//
// These are surface operations (O(N^2) cells of an O(N^3) level): one thread per output cell, records
// precomputed on the host (amr.cpp).  Arithmetic follows the reference expression by expression so that the
// results equal the oracle's bit for bit:
//   k_cf_slopes / k_cf_quad  MappedQuadCFInterp::getPhiStar + interpOnIVS, MappedQuadCFInterp.cpp:222-562;
//                            packed variants MAPPEDPHISTAR / MAPPEDQUADINTERP, MappedQuadCFInterpF.ChF:9-127;
//                            derivative formulas MappedCFStencil.cpp:378-597
//   k_fine_register          MappedAMRPoissonOp::reflux (fine loop, MappedAMRPoissonOp.cpp:1676-1701) +
//                            MAPPEDGETFLUXORTHO + MAPPEDINCREMENTFINE (MappedLevelFluxRegisterF.ChF:9-44)
//   k_reflux                 incrementCoarse (MappedLevelFluxRegister.cpp:298-347) + reflux (:560-650)
#include "amr.h"

namespace somar {

__device__ __forceinline__ long long pidx(const PatchDesc& p, int i, int j, int k)
{
    return p.off + i + (long long)p.pj * j + p.pk * k;
}

// ---- box-to-box copies between two layouts ----------------------------------------------------------
__global__ void k_copy_items2(const CopyItem* __restrict__ items, const PatchDesc* __restrict__ spatches,
                              const PatchDesc* __restrict__ dpatches, const double* __restrict__ src,
                              double* __restrict__ dst)
{
    const CopyItem it = items[blockIdx.x];
    const PatchDesc sp = spatches[it.src_patch];
    const PatchDesc dp = dpatches[it.dst_patch];
    const int n0 = it.n[0], n01 = it.n[0] * it.n[1];
    const long long cells = (long long)n01 * it.n[2];
    for (long long idx = (long long)blockIdx.y * blockDim.x + threadIdx.x; idx < cells;
         idx += (long long)gridDim.y * blockDim.x) {
        const int k = (int)(idx / n01);
        const int r = (int)(idx - (long long)k * n01);
        const int j = r / n0, i = r - j * n0;
        dst[pidx(dp, it.dst_lo[0] + i, it.dst_lo[1] + j, it.dst_lo[2] + k)] =
            src[pidx(sp, it.src_lo[0] + i, it.src_lo[1] + j, it.src_lo[2] + k)];
    }
}

__global__ void k_fill_items(const FillItem* __restrict__ items, const PatchDesc* __restrict__ patches,
                             double* __restrict__ f, double v)
{
    const FillItem it = items[blockIdx.x];
    const PatchDesc p = patches[it.patch];
    const int rows = it.n[1] * it.n[2];
    for (int row = blockIdx.y * blockDim.y + threadIdx.y; row < rows; row += gridDim.y * blockDim.y) {
        const int j = row % it.n[1], k = row / it.n[1];
        const long long d = pidx(p, it.lo[0], it.lo[1] + j, it.lo[2] + k);
        for (int i = threadIdx.x; i < it.n[0]; i += blockDim.x) f[d + i] = v;
    }
}

// ---- quadratic coarse-fine interpolation ------------------------------------------------------------
struct D3 { double v[3]; };
struct I3 { int v[3]; };

// homogeneousCFInterp (calculus/interpolation/HomogeneousCFInterp.cpp:30-201): the ghost value of the
// quadratic through the two nearest valid cells and a ZERO coarse value half a coarse cell beyond the interface
__global__ void k_cf_homog(const CFCell* __restrict__ cells, int n, double* __restrict__ phi, D3 c1, D3 c2, D3 fac)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CFCell c = cells[i];
    const int d = c.dir & 3;
    const double pb = phi[c.off - (long long)c.stride];
    double v;
    if (c.dir & 4) {
        v = fac.v[d] * pb;
    } else {
        const double pa = phi[c.off - 2 * (long long)c.stride];
        v = c1.v[d] * pb + c2.v[d] * pa;
    }
    phi[c.off] = v;
}

__global__ void k_cf_slopes(const QCoarse* __restrict__ cc, int ncc, const QPoint* __restrict__ pts,
                            const double* __restrict__ buf, double* __restrict__ der, D3 dxc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncc) return;
    const QCoarse c = cc[i];
    const int t1 = c.dir == 0 ? 1 : 0;
    const int t2 = c.dir == 2 ? 1 : 2;
    const double h1 = dxc.v[t1], h2 = dxc.v[t2];
    const double* p = buf + c.boff;
    double out[5];
    if (c.flags & 1) {
        const double p0 = p[0];
        out[0] = (p[c.s1] - p[-c.s1]) / (2.0 * h1);
        out[1] = (p[c.s1] + p[-c.s1] - 2.0 * p0) / (h1 * h1);
        out[2] = (p[c.s2] - p[-c.s2]) / (2.0 * h2);
        out[3] = (p[c.s2] + p[-c.s2] - 2.0 * p0) / (h2 * h2);
        out[4] = (p[c.s1 + c.s2] + p[-c.s1 - c.s2] - p[c.s1 - c.s2] - p[-c.s1 + c.s2]) / (4.0 * h2 * h1);
    } else {
        int q = c.p0;
        for (int s = 0; s < 5; ++s) {
            double acc = 0.0;
            for (int n = 0; n < c.np[s]; ++n, ++q) acc = acc + pts[q].w * p[pts[q].off];
            double den;
            if (s == 0) den = h1;
            else if (s == 1) den = h1 * h1;
            else if (s == 2) den = h2;
            else if (s == 3) den = h2 * h2;
            else den = h1 * h2;
            out[s] = acc / den;
        }
    }
    for (int s = 0; s < 5; ++s) der[5 * (long long)i + s] = out[s];
}

__global__ void k_cf_quad(const QFine* __restrict__ fc, int nfc, const QCoarse* __restrict__ cc,
                          const double* __restrict__ der, const double* __restrict__ buf, double* __restrict__ fine,
                          D3 dxf, D3 dxc, I3 r)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nfc) return;
    const QFine f = fc[i];
    const int dir = f.dirflags & 3;
    const bool packed = (f.dirflags >> 2) & 1;
    const int t1 = dir == 0 ? 1 : 0;
    const int t2 = dir == 2 ? 1 : 2;
    const double x1 = ((double)f.ivf1 + 0.5) * dxf.v[t1] - ((double)f.ivc1 + 0.5) * dxc.v[t1];
    const double x2 = ((double)f.ivf2 + 0.5) * dxf.v[t2] - ((double)f.ivc2 + 0.5) * dxc.v[t2];
    const double* d = der + 5 * (long long)f.cc;
    const double pc = buf[cc[f.cc].boff];
    const double u1 = x1 * d[0] + 0.5 * x1 * x1 * d[1];
    const double u2 = x2 * d[2] + 0.5 * x2 * x2 * d[3];
    const double u3 = x1 * x2 * d[4];
    const double pstar = pc + u1 + u2 + u3;
    const double pa = fine[f.foff - 2 * (long long)f.stride];
    const double pb = fine[f.foff - (long long)f.stride];
    const double h = dxf.v[dir];
    const int nref = r.v[dir];
    double val;
    if (packed) {
        const double frac = 2.0 / (h * h);
        const double denom = (double)(nref * nref + 4 * nref + 3);
        const double mult = frac / denom;
        const double invh = 1.0 / h;
        const double x = 2.0 * h;
        const double xsq = 4.0 * h * h;
        const double a = mult * (2.0 * pstar + (double)(nref + 1) * pa - (double)(nref + 3) * pb);
        const double b = (pb - pa) * invh - a * h;
        val = xsq * a + b * x + pa;
    } else {
        const double a = (2.0 / h / h) * (2.0 * pstar + pa * ((double)nref + 1.0) - pb * ((double)nref + 3.0)) /
                         ((double)(nref * nref + 4 * nref) + 3.0);
        const double b = (pb - pa) / h - a * h;
        const double x = 2.0 * h;
        val = a * x * x + b * x + pa;
    }
    fine[f.foff] = val;
}

// ---- flux register -----------------------------------------------------------------------------------
struct JG3 { const double* v[3]; };
struct SC6 { double v[3][2]; };

// MAPPEDGETFLUX with beta = a_ref = 1 at the face that is the LOW face of cell c in direction a: flux19 of full19.hip, same
// expression (k_flux_full filled whole face fields with it; the register needs it on its own faces only)
__device__ __forceinline__ double reg_flux19(const double* __restrict__ phi, const FullFlux& F, long long c, int a,
                                             const long long st[3])
{
    const int b = (a + 1) % 3, cc = (a + 2) % 3;
    const long long sa = st[a], sb = st[b], sc = st[cc];
    const double* E = F.psi;
    const double aScale = 1.0 * F.dxi[a], bScale = 0.25 * 1.0 * F.dxi[b], cScale = 0.25 * 1.0 * F.dxi[cc];
    return aScale * F.J[a][a][c] * (phi[c] - phi[c - sa]) +
           bScale * F.J[a][b][c] * (E[c + sb] - E[c - sb] + E[c + sb - sa] - E[c - sb - sa]) +
           cScale * F.J[a][cc][c] * (E[c + sc] - E[c - sc] + E[c + sc - sa] - E[c - sc - sa]);
}

// fl: precomputed face fields (the composite divergence's face velocities) or all null; ff.psi: non-diagonal metric, fluxes
// evaluated here
__global__ void k_fine_register(const FRegCell* __restrict__ cells, int n, const PatchDesc* __restrict__ fpatches,
                                const double* __restrict__ phi, JG3 jg, JG3 fl, FullFlux ff, D3 dxf, SC6 sc, I3 r,
                                double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FRegCell c = cells[i];
    const PatchDesc p = fpatches[c.patch];
    const long long st[3] = {1, (long long)p.pj, p.pk};
    const int d = c.dir;
    const double scale = 1.0 / dxf.v[d];  // a_ref / m_dx[a_dir], a_ref = 1
    const double s = sc.v[d][c.side];
    const double* J = jg.v[d];
    const int n0 = d == 0 ? 1 : r.v[0], n1 = d == 1 ? 1 : r.v[1], n2 = d == 2 ? 1 : r.v[2];
    double acc = 0.0;
    for (int o2 = 0; o2 < n2; ++o2)
        for (int o1 = 0; o1 < n1; ++o1)
            for (int o0 = 0; o0 < n0; ++o0) {
                const long long f = c.cell0 + o0 + st[1] * o1 + st[2] * o2;
                const double flux = fl.v[d] ? fl.v[d][f] : (ff.psi ? reg_flux19(phi, ff, f, d, st) : J[f] * scale * (phi[f] - phi[f - st[d]]));
                acc = acc + s * flux;
            }
    out[i] = acc;
}

__global__ void k_gather(const int* __restrict__ idx, long long n, const double* __restrict__ src,
                         double* __restrict__ dst)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

__global__ void k_reflux(const RefluxCell* __restrict__ cells, int n, const RefluxA* __restrict__ A,
                         const int* __restrict__ B, const PatchDesc* __restrict__ cpatches,
                         const double* __restrict__ phi, JG3 jg, JG3 fl, FullFlux ff, const double* __restrict__ jinv, D3 dxc,
                         const double* __restrict__ freg, double* __restrict__ L)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RefluxCell c = cells[i];
    const PatchDesc p = cpatches[c.patch];
    const long long st[3] = {1, (long long)p.pj, p.pk};
    double coar = 0.0;
    for (int a = c.a0; a < c.a0 + c.na; ++a) {
        const RefluxA e = A[a];
        const double scale = 1.0 / dxc.v[e.dir];
        const double flux = fl.v[e.dir] ? fl.v[e.dir][e.face]
                                        : (ff.psi ? reg_flux19(phi, ff, e.face, e.dir, st)
                                                  : jg.v[e.dir][e.face] * scale * (phi[e.face] - phi[e.face - st[e.dir]]));
        coar = coar + e.sc * flux;
    }
    double inc = 0.0;
    inc = inc + -1.0 * coar;
    for (int b = c.b0; b < c.b0 + c.nb; ++b) inc = inc + -1.0 * freg[B[b]];
    inc = inc * jinv[c.coff];
    L[c.coff] = L[c.coff] + inc;
}

// ---- launchers ---------------------------------------------------------------------------------------
static inline int grid1(long long n) { return (int)((n + 255) / 256); }
static D3 d3(const double v[3]) { D3 o; for (int i = 0; i < 3; ++i) o.v[i] = v[i]; return o; }
static I3 i3(const int v[3]) { I3 o; for (int i = 0; i < 3; ++i) o.v[i] = v[i]; return o; }

void launch_cf_homog(hipStream_t st, const CFCell* cells, int n, double* phi, const double c1[3], const double c2[3],
                     const double fac[3])
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_cf_homog, dim3(grid1(n)), dim3(256), 0, st, cells, n, phi, d3(c1), d3(c2), d3(fac));
}
void launch_copy_items2(hipStream_t st, const PatchDesc* spatches, const PatchDesc* dpatches, const CopyItem* items,
                        int nitems, const double* src, double* dst)
{
    if (nitems == 0) return;
    hipLaunchKernelGGL(k_copy_items2, dim3(nitems, 16), dim3(256), 0, st, items, spatches, dpatches, src, dst);
}
void launch_fill_items(hipStream_t st, const PatchDesc* patches, const FillItem* items, int nitems, double* f, double v)
{
    if (nitems == 0) return;
    hipLaunchKernelGGL(k_fill_items, dim3(nitems, 8), dim3(64, 4), 0, st, items, patches, f, v);
}
void launch_cf_slopes(hipStream_t st, const QCoarse* cc, int ncc, const QPoint* pts, const double* buf, double* der,
                      const double dxc[3])
{
    if (ncc == 0) return;
    hipLaunchKernelGGL(k_cf_slopes, dim3(grid1(ncc)), dim3(256), 0, st, cc, ncc, pts, buf, der, d3(dxc));
}
void launch_cf_quad(hipStream_t st, const QFine* fc, int nfc, const QCoarse* cc, const double* der, const double* buf,
                    double* fine, const double dxf[3], const double dxc[3], const int r[3])
{
    if (nfc == 0) return;
    hipLaunchKernelGGL(k_cf_quad, dim3(grid1(nfc)), dim3(256), 0, st, fc, nfc, cc, der, buf, fine, d3(dxf), d3(dxc),
                       i3(r));
}
void launch_fine_register(hipStream_t st, const FRegCell* cells, int n, const PatchDesc* fpatches, const double* phi,
                          double* const jg[3], const double dxf[3], const double sc[3][2], const int r[3], double* out,
                          double* const* fluxes, const FullFlux* ff)
{
    if (n == 0) return;
    JG3 J, FL;
    SC6 S;
    for (int d = 0; d < 3; ++d) {
        J.v[d] = jg[d];
        FL.v[d] = fluxes ? fluxes[d] : nullptr;
        S.v[d][0] = sc[d][0];
        S.v[d][1] = sc[d][1];
    }
    hipLaunchKernelGGL(k_fine_register, dim3(grid1(n)), dim3(256), 0, st, cells, n, fpatches, phi, J, FL, ff ? *ff : FullFlux(),
                       d3(dxf), S, i3(r), out);
}
// CRSEONESIDEGRAD (calculus/DivCurlGrad/DivCurlGradF.ChF:626-697) over a precomputed face list
struct G3 { double* v[3]; };
__global__ void k_one_sided(const OneSided* __restrict__ e, int n, G3 g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const OneSided o = e[i];
    double* f = g.v[o.dirmode & 3];
    const long long s = o.stride;
    if ((o.dirmode >> 2) == 2) f[o.face] = 2.0 * f[o.face + s] - f[o.face + 2 * s];
    else f[o.face] = f[o.face + s];
}
void launch_one_sided(hipStream_t st, const OneSided* e, int n, double* const grad[3])
{
    if (n == 0) return;
    G3 g;
    for (int d = 0; d < 3; ++d) g.v[d] = grad[d];
    hipLaunchKernelGGL(k_one_sided, dim3(grid1(n)), dim3(256), 0, st, e, n, g);
}

// UNMAPPEDAVERAGE: one thread per coarse cell of the coarsened-fine layout, children summed in the Fortran's ii2, ii1, ii0 order
__global__ void k_avg_unweighted(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ cpatches,
                                 const PatchDesc* __restrict__ fpatches, double* __restrict__ crse,
                                 const double* __restrict__ fine, int r0, int r1, int r2)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc cp = cpatches[t.patch];
    const PatchDesc fp = fpatches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    if (lj >= cp.n[1]) return;
    const double refScale = 1.0 / (double)(r0 * r1 * r2);
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = t.i0 + 2 * threadIdx.x + q;
            if (li >= cp.n[0]) continue;
            const int lk = t.k0 + kk;
            double sum = 0.0;
            for (int o2 = 0; o2 < r2; ++o2)
                for (int o1 = 0; o1 < r1; ++o1)
                    for (int o0 = 0; o0 < r0; ++o0) sum = sum + fine[pidx(fp, li * r0 + o0, lj * r1 + o1, lk * r2 + o2)];
            crse[pidx(cp, li, lj, lk)] = sum * refScale;
        }
}
void launch_avg_unweighted(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine, const int r[3])
{
    if (C.ntiles == 0) return;
    hipLaunchKernelGGL(k_avg_unweighted, dim3(C.ntiles), dim3(64, C.tile_j, 1), 0, st, C.tiles, C.patches, F.patches, crse,
                       fine, r[0], r[1], r[2]);
}

__global__ void k_divide(double* __restrict__ a, double d, long long n)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] / d;
}
void launch_divide(hipStream_t st, double* a, double d, long long n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_divide, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, d, n);
}

void launch_gather(hipStream_t st, const int* idx, long long n, const double* src, double* dst)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_gather, dim3(grid1(n)), dim3(256), 0, st, idx, n, src, dst);
}
// the register scale m_beta / m_dx[idir] is taken when reflux runs (MappedAMRPoissonOp.cpp:1661, 1693): after
// setAlphaAndBeta the coarse-side table entries are rewritten with the same expression sgn * (beta / dx)
__global__ void k_reflux_rescale(RefluxA* __restrict__ A, long long n, D3 scale)
{
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RefluxA e = A[i];
    A[i].sc = -(e.sgn > 0 ? -1.0 : 1.0) * scale.v[e.dir];
}

void launch_reflux_rescale(hipStream_t st, RefluxA* A, long long n, const double scale[3])
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_reflux_rescale, dim3(grid1(n)), dim3(256), 0, st, A, n, d3(scale));
}

void launch_reflux(hipStream_t st, const RefluxCell* cells, int n, const RefluxA* A, const int* B,
                   const PatchDesc* cpatches, const double* phi, double* const jg[3], const double* jinv,
                   const double dxc[3], const double* freg, double* LofPhi, double* const* fluxes, const FullFlux* ff)
{
    if (n == 0) return;
    JG3 J, FL;
    for (int d = 0; d < 3; ++d) { J.v[d] = jg[d]; FL.v[d] = fluxes ? fluxes[d] : nullptr; }
    hipLaunchKernelGGL(k_reflux, dim3(grid1(n)), dim3(256), 0, st, cells, n, A, B, cpatches, phi, J, FL, ff ? *ff : FullFlux(),
                       jinv, d3(dxc), freg, LofPhi);
}

}  // namespace somar
