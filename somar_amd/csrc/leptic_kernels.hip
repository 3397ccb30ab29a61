// somar_amd/csrc/leptic_kernels.hip -- column kernels of the leptic level solver (leptic.cpp).
//
// Reference (per box, one Fortran call each, flat scratch FABs allocated per call):
//   TriDiagPoissonNN1DFAB      utils/TridiagUtilsF.ChF:85-166            vertical Neumann-Neumann line solves
//   UNMAPPEDVERTINTEGRAL       utils/SubspaceF.ChF:33-60                 vertical sums (excess, averaged gradient, metric)
//   ADDEXTRUSION               utils/SubspaceF.ChF:66-110
//   MAPPEDMACGRADORTHO         calculus/DivCurlGrad/DivCurlGradF.ChF:221-285 (normal derivative)
//   LEPTICACCUMDIV             calculus/LepticSolver/LevelLepticSolverF.ChF:285-331
//   call sites                 calculus/LepticSolver/LevelLepticSolver.cpp:981-1097, 1183-1240, 1248-1421, 1476-1517
//
// Layout: the level's boxes are vertically complete columns, so every serial recurrence (Thomas sweep, vertical sum)
// runs along k, the SLOW index, and one lane owns one (i,j) column: 64 adjacent columns per wavefront make every
// load of the k-march a unit-stride 512-byte row segment, no cross-lane traffic, no LDS.  Flat (one cell thick)
// companions -- excess, vertical boundary data, averaged gradients, the horizontal right-hand side -- live in
// fields of the 2-D horizontal level, whose patch p is the flattened patch p of the 3-D level.
// Arithmetic follows the reference term by term (no FMA contraction): results are bit-identical to the oracle.
#include "common.h"
#include "kernels.h"

namespace somar {

namespace {
struct Col {
    bool ok;
    int li, lj;
    long long c;   // cell (li, lj, 0) of the 3-D patch
    long long h;   // flat cell (li, lj) of the 2-D patch
};
__device__ __forceinline__ Col column(const Tile& t, const PatchDesc& p, const PatchDesc& hp, int q)
{
    Col o;
    o.li = t.i0 + threadIdx.x + 64 * q;
    o.lj = t.j0 + threadIdx.y;
    o.ok = o.li < p.n[0] && o.lj < p.n[1];
    o.c = p.off + o.li + (long long)p.pj * o.lj;
    o.h = hp.off + o.li + (long long)hp.pj * o.lj;
    return o;
}
}  // namespace

// vertical average of the horizontal J g^{aa}: the metric of the flat problem (createVertAvgFCJgupPtr)
__global__ __launch_bounds__(512) void k_lep_avg_metric(const Tile* __restrict__ tiles,
                                                        const PatchDesc* __restrict__ vp,
                                                        const PatchDesc* __restrict__ hpp,
                                                        const double* __restrict__ jgx, const double* __restrict__ jgy,
                                                        double* __restrict__ hjgx, double* __restrict__ hjgy)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const double scale = 1.0 / (double)p.n[2];
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const bool xh = o.li == p.n[0] - 1, yh = o.lj == p.n[1] - 1;
        double ax = 0.0, ay = 0.0, axh = 0.0, ayh = 0.0;
        long long c = o.c;
#pragma unroll 8
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) {
            ax = ax + jgx[c] * scale;
            ay = ay + jgy[c] * scale;
            if (xh) axh = axh + jgx[c + 1] * scale;
            if (yh) ayh = ayh + jgy[c + p.pj] * scale;
        }
        hjgx[o.h] = ax;
        hjgy[o.h] = ay;
        if (xh) hjgx[o.h + 1] = axh;
        if (yh) hjgy[o.h + hp.pj] = ayh;
    }
}

// excess = hiNeumBC - loNeumBC - Integral[rhs]      (computeVerticalExcess)
__global__ __launch_bounds__(512) void k_lep_excess(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                    const PatchDesc* __restrict__ hpp, const double* __restrict__ rhs,
                                                    const double* __restrict__ bcLo, const double* __restrict__ bcHi,
                                                    double* __restrict__ excess, double dzScale)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        double e = bcHi[o.h];
        e = e + (-1.0) * bcLo[o.h];
        long long c = o.c;
#pragma unroll 8
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) e = e + rhs[c] * dzScale;
        excess[o.h] = e;
    }
}

// verticalLineSolver, Neumann-Neumann columns: roll the boundary values into the end cells of rhs, Thomas sweep with
// the reference's special last row, remove the column mean, roll the boundary values out again (which, as in the
// reference, does not restore the end cells of rhs bit for bit).  gam: scratch field.
__global__ __launch_bounds__(512) void k_lep_vsolve(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                    const PatchDesc* __restrict__ hpp, double* __restrict__ phi,
                                                    double* __restrict__ rhs, const double* __restrict__ jgz,
                                                    double* __restrict__ gam, const double* __restrict__ bcLo,
                                                    const double* __restrict__ bcHi, double dz)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const int N = p.n[2];
    const long long sk = p.pk;
    const double dzsq = dz * dz;
    const double sLo = 1.0 / dz, sHi = -1.0 / dz;  // -isign / dz
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const long long c0 = o.c, cN = o.c + (long long)(N - 1) * sk;
        const double rollLo = 0.0 + bcLo[o.h] * sLo;
        const double rollHi = 0.0 + bcHi[o.h] * sHi;
        const double r0 = rhs[c0] + rollLo;
        const double rN = rhs[cN] + rollHi;
        // forward elimination; x goes to phi, gam to the scratch field
        double sig = jgz[c0 + sk];  // sigma(1)
        double cc = sig;
        double bet = -cc;
        double x = (r0 * dzsq) / bet;
        double g = cc / bet;
        phi[c0] = x;
        gam[c0] = g;
        double a_prev = 1.2345e10;  // a(0): the reference's sentinel, read by the last row when N == 2
        long long c = c0 + sk;
#pragma unroll 4
        for (int r = 1; r <= N - 2; ++r, c += sk) {
            const double a = sig;
            sig = jgz[c + sk];
            cc = sig;
            const double b = -(a + cc);
            bet = b - a * g;
            x = (rhs[c] * dzsq - a * x) / bet;
            g = cc / bet;
            phi[c] = x;
            gam[c] = g;
            a_prev = a;
        }
        // last row as written in the reference: a(r-1), plain b(r)
        {
            const double b = -sig;  // sigma(N-1)
            x = (rN * dzsq - a_prev * x) / b;
            phi[cN] = x;
        }
        double avg = x;
        c = cN - sk;
#pragma unroll 4
        for (int r = N - 2; r >= 0; --r, c -= sk) {
            x = phi[c] - gam[c] * x;
            avg = avg + x;
            phi[c] = x;
        }
        avg = avg / (double)N;
        c = c0;
#pragma unroll 8
        for (int r = 0; r < N; ++r, c += sk) phi[c] = phi[c] - avg;
        rhs[c0] = r0 + rollLo * (-1.0);
        rhs[cN] = rN + rollHi * (-1.0);
    }
}

// verticalLineSolver on columns that END at a Dirichlet wall or a coarse-fine interface: LepticLapackVerticalSolver
// (LevelLepticSolverF.ChF:161-283) -- the symmetric tridiagonal system D, DL (= dptsv's E), B = -rhs with the end rows of
// BCType Neum / Diri / CF (linear interpolation against a zero coarse value, alpha = 1 - 2 dz / (dzCrse + dz)) -- and
// LAPACK dptsv (EXTERNAL), restated as dpttrf's L D L^T loop fused with dptts2's forward substitution on the way up and
// dptts2's back substitution on the way down.  dfac / efac: scratch fields for the factors.  vbc[2 patch + side]: 0 Neum,
// 1 Diri, 2 CF.  A non-positive pivot (dptsv's INFO != 0) is counted in *bad.
__global__ __launch_bounds__(512) void k_lep_vsolve_lapack(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                           double* __restrict__ phi, const double* __restrict__ rhs,
                                                           const double* __restrict__ jgz, double* __restrict__ dfac,
                                                           double* __restrict__ efac, const int* __restrict__ vbc,
                                                           double dz, double dzCrse, int* __restrict__ bad)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch];
    const int N = p.n[2];
    const long long sk = p.pk;
    const double invdzsq = 1.0 / (dz * dz);
    const int lo = vbc[2 * t.patch], hi = vbc[2 * t.patch + 1];
    const double alpha = 1.0 - 2.0 * dz / (dzCrse + dz);
    for (int q = 0; q < 2; ++q) {
        const int li = t.i0 + threadIdx.x + 64 * q, lj = t.j0 + threadIdx.y;
        if (li >= p.n[0] || lj >= p.n[1]) continue;
        const long long c0 = p.off + li + (long long)p.pj * lj;
        int info = 0;
        // row k (1-based) sits in cell k-1; Jgzz(IDX(k)) is the low face of cell k
        auto diag = [&](int k, long long c) {   // D(k), c = cell k-1
            if (k == 1) {
                if (lo == 0) return jgz[c + sk] * invdzsq;
                if (lo == 1) return (2.0 * jgz[c] + jgz[c + sk]) * invdzsq;
                return ((1.0 - alpha) * jgz[c] + jgz[c + sk]) * invdzsq;
            }
            if (k == N) {
                if (hi == 0) return jgz[c] * invdzsq;
                if (hi == 1) return (jgz[c] + 2.0 * jgz[c + sk]) * invdzsq;
                return (jgz[c] + (1.0 - alpha) * jgz[c + sk]) * invdzsq;
            }
            return (jgz[c] + jgz[c + sk]) * invdzsq;
        };
        long long c = c0;
        double d = diag(1, c);
        double b = -rhs[c];
        for (int k = 1; k <= N - 1; ++k, c += sk) {
            if (!(d > 0.0)) info = 1;
            const double ei = -jgz[c + sk] * invdzsq;          // DL(k)
            const double e = ei / d;
            dfac[c] = d;
            efac[c] = e;
            phi[c] = b;
            d = diag(k + 1, c + sk) - e * ei;
            b = -rhs[c + sk] - b * e;
        }
        if (!(d > 0.0)) info = 1;
        double x = b / d;                                        // B(N) / D(N)
        phi[c] = x;
        c -= sk;
        for (int k = N - 1; k >= 1; --k, c -= sk) {
            x = phi[c] / dfac[c] - x * efac[c];
            phi[c] = x;
        }
        if (info) atomicAdd(bad, 1);
    }
}

// vertical average of the horizontal face gradients J g^{aa} d_a phi (MAPPEDMACGRADORTHO + UNMAPPEDVERTINTEGRAL);
// faces on the physical boundary carry the boundary data, which are zero.  Every column fills its low faces, the
// last column of a box also the high face (its phi ghost has been exchanged).
__global__ __launch_bounds__(512) void k_lep_hgrad(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                   const PatchDesc* __restrict__ hpp, const double* __restrict__ phi,
                                                   const double* __restrict__ jgx, const double* __restrict__ jgy,
                                                   double* __restrict__ gx, double* __restrict__ gy, StencilParams P)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const double dzScale = 1.0 / (double)p.n[2];
    const double dxinv = 1.0 / P.dx[0], dyinv = 1.0 / P.dx[1];
    const long long sj = p.pj;
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const int gi = p.lo[0] + o.li, gj = p.lo[1] + o.lj;
        const bool xh = o.li == p.n[0] - 1, yh = o.lj == p.n[1] - 1;
        const bool bxl = gi == P.dom_lo[0], bxh = gi == P.dom_hi[0];
        const bool byl = gj == P.dom_lo[1], byh = gj == P.dom_hi[1];
        double ax = 0.0, ay = 0.0, axh = 0.0, ayh = 0.0;
        long long c = o.c;
#pragma unroll 4
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) {
            const double pc = phi[c];
            if (!bxl) ax = ax + ((dxinv * jgx[c]) * (pc - phi[c - 1])) * dzScale;
            if (!byl) ay = ay + ((dyinv * jgy[c]) * (pc - phi[c - sj])) * dzScale;
            if (xh && !bxh) axh = axh + ((dxinv * jgx[c + 1]) * (phi[c + 1] - pc)) * dzScale;
            if (yh && !byh) ayh = ayh + ((dyinv * jgy[c + sj]) * (phi[c + sj] - pc)) * dzScale;
        }
        gx[o.h] = ax;
        gy[o.h] = ay;
        if (xh) gx[o.h + 1] = axh;
        if (yh) gy[o.h + hp.pj] = ayh;
    }
}

// ---- non-diagonal metric ----------------------------------------------------------------------------------------
struct LJ { const double* c[3][3]; };   // J g^{ab} on a-faces, c[faceDir][component]
struct HJ { double* c[2][2]; };         // the flat problem's J g^{ab}, a, b < 2

// vertical average of the horizontal block J g^{ab}, a, b < 2 (createVertAvgFCJgupPtr, LevelGeometryBasics.cpp:500-569)
__global__ __launch_bounds__(512) void k_lep_avg_metric_full(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                             const PatchDesc* __restrict__ hpp, LJ J, HJ H)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const double scale = 1.0 / (double)p.n[2];
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const bool xh = o.li == p.n[0] - 1, yh = o.lj == p.n[1] - 1;
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                const double* src = J.c[a][b];
                const long long hs = a == 0 ? 1 : (long long)p.pj;
                const bool hi = a == 0 ? xh : yh;
                double lo = 0.0, up = 0.0;
                long long c = o.c;
                for (int k = 0; k < p.n[2]; ++k, c += p.pk) {
                    lo = lo + src[c] * scale;
                    if (hi) up = up + src[c + hs] * scale;
                }
                H.c[a][b][o.h] = lo;
                if (hi) H.c[a][b][o.h + (a == 0 ? 1 : (long long)hp.pj)] = up;
            }
    }
}

// MAPPEDMACGRAD, normal branch (DivCurlGradF.ChF:87-155), at the face that is the LOW face of cell c in direction a, phi
// serving as its own extrap (LevelLepticSolver.cpp:1040-1049): the expression of flux19 (full19.hip) term for term
__device__ __forceinline__ double lep_macgrad(const double* __restrict__ phi, const LJ& J, long long c, int a,
                                              const long long st[3], const double dxi[3])
{
    const int b = (a + 1) % 3, cc = (a + 2) % 3;
    const long long sa = st[a], sb = st[b], sc = st[cc];
    const double aScale = 1.0 * dxi[a], bScale = 0.25 * 1.0 * dxi[b], cScale = 0.25 * 1.0 * dxi[cc];
    return aScale * J.c[a][a][c] * (phi[c] - phi[c - sa]) +
           bScale * J.c[a][b][c] * (phi[c + sb] - phi[c - sb] + phi[c + sb - sa] - phi[c - sb - sa]) +
           cScale * J.c[a][cc][c] * (phi[c + sc] - phi[c - sc] + phi[c + sc - sa] - phi[c - sc - sa]);
}

// the same vertical averages as k_lep_hgrad with the full gradient; phi carries every ghost (extrapAllGhosts + exchange)
__global__ __launch_bounds__(512) void k_lep_hgrad_full(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                        const PatchDesc* __restrict__ hpp, const double* __restrict__ phi,
                                                        LJ J, double* __restrict__ gx, double* __restrict__ gy,
                                                        StencilParams P)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const double dzScale = 1.0 / (double)p.n[2];
    const double dxi[3] = {1.0 / P.dx[0], 1.0 / P.dx[1], 1.0 / P.dx[2]};
    const long long st[3] = {1, (long long)p.pj, p.pk};
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const int gi = p.lo[0] + o.li, gj = p.lo[1] + o.lj;
        const bool xh = o.li == p.n[0] - 1, yh = o.lj == p.n[1] - 1;
        const bool bxl = gi == P.dom_lo[0], bxh = gi == P.dom_hi[0];
        const bool byl = gj == P.dom_lo[1], byh = gj == P.dom_hi[1];
        double ax = 0.0, ay = 0.0, axh = 0.0, ayh = 0.0;
        long long c = o.c;
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) {
            if (!bxl) ax = ax + lep_macgrad(phi, J, c, 0, st, dxi) * dzScale;
            if (!byl) ay = ay + lep_macgrad(phi, J, c, 1, st, dxi) * dzScale;
            if (xh && !bxh) axh = axh + lep_macgrad(phi, J, c + 1, 0, st, dxi) * dzScale;
            if (yh && !byh) ayh = ayh + lep_macgrad(phi, J, c + st[1], 1, st, dxi) * dzScale;
        }
        gx[o.h] = ax;
        gy[o.h] = ay;
        if (xh) gx[o.h + 1] = axh;
        if (yh) gy[o.h + hp.pj] = ayh;
    }
}

// LEPTICVERTHORIZGRAD (LevelLepticSolverF.ChF:59-99) on the bottom (side 0) and top (side 1) faces of every column:
// scale * J g^{z m} d(phi)/dx^m, m = x, y, over the ghost and the first valid layer (phi's vertical ghosts, and the x / y
// ghosts of those two layers, extrapolated by run_aux_program(1))
__global__ __launch_bounds__(512) void k_lep_vhgrad(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                    const PatchDesc* __restrict__ hpp, const double* __restrict__ phi,
                                                    const double* __restrict__ jgzx, const double* __restrict__ jgzy,
                                                    double* __restrict__ bcLo, double* __restrict__ bcHi, StencilParams P,
                                                    double scale)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    const double dxinv0 = scale * 0.25 / P.dx[0], dxinv1 = scale * 0.25 / P.dx[1];
    const long long sj = p.pj, sk = p.pk;
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        for (int side = 0; side < 2; ++side) {
            const long long f = o.c + (side ? (long long)p.n[2] * sk : 0);   // the cell whose LOW face the boundary face is
            const long long g = side ? f : f - sk, v = side ? f - sk : f;     // ghost / first valid cell
            const double val = jgzx[f] * dxinv0 * (phi[g + 1] - phi[g - 1] + phi[v + 1] - phi[v - 1]) +
                               jgzy[f] * dxinv1 * (phi[g + sj] - phi[g - sj] + phi[v + sj] - phi[v - sj]);
            (side ? bcHi : bcLo)[o.h] = val;
        }
    }
}

// LEPTICACCUMDIV over both horizontal directions, minus excess / H, then the addTo into the zeroed horizontal rhs
__global__ __launch_bounds__(512) void k_lep_hrhs(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                  const PatchDesc* __restrict__ hpp, const double* __restrict__ gx,
                                                  const double* __restrict__ gy, const double* __restrict__ excess,
                                                  double* __restrict__ hrhs, double sx, double sy, double negInvH,
                                                  int useExcess)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        double r = 0.0;
        r = r + (gx[o.h + 1] - gx[o.h]) * sx;
        r = r + (gy[o.h + hp.pj] - gy[o.h]) * sy;
        if (useExcess) r = r + excess[o.h] * negInvH;
        hrhs[o.h] = 0.0 + r;
    }
}

// ADDEXTRUSION: phi(i,j,k) += flat(i,j)
__global__ __launch_bounds__(512) void k_lep_extrude(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                     const PatchDesc* __restrict__ hpp, double* __restrict__ phi,
                                                     const double* __restrict__ flat)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch], hp = hpp[t.patch];
    for (int q = 0; q < 2; ++q) {
        const Col o = column(t, p, hp, q);
        if (!o.ok) continue;
        const double v = flat[o.h];
        long long c = o.c;
#pragma unroll 8
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) phi[c] = phi[c] + v;
    }
}

// valid cells only: MODE 0  y = x / b;  MODE 1  y = y + a * x
template <int MODE>
__global__ __launch_bounds__(512) void k_lep_valid(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ vp,
                                                   double* __restrict__ y, const double* __restrict__ x,
                                                   const double* __restrict__ b, double a)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = vp[t.patch];
    for (int q = 0; q < 2; ++q) {
        const int li = t.i0 + threadIdx.x + 64 * q, lj = t.j0 + threadIdx.y;
        if (li >= p.n[0] || lj >= p.n[1]) continue;
        long long c = p.off + li + (long long)p.pj * lj;
#pragma unroll 8
        for (int k = 0; k < p.n[2]; ++k, c += p.pk) {
            if (MODE == 0) y[c] = x[c] / b[c];
            else y[c] = y[c] + a * x[c];
        }
    }
}

// ---- launchers (V: the 3-D level, ctiles: its whole-column tiles; H: the flat level) ----------------------------
#define LEP_GRID(nct, tj) dim3(nct), dim3(64, tj, 1), 0, st

void launch_lep_avg_metric(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H)
{
    if (nct) hipLaunchKernelGGL(k_lep_avg_metric, LEP_GRID(nct, tj), ct, V.patches, H.patches, V.jg[0], V.jg[1], H.jg[0], H.jg[1]);
}
void launch_lep_excess(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       const double* rhs, const double* bcLo, const double* bcHi, double* excess, double dzScale)
{
    if (nct) hipLaunchKernelGGL(k_lep_excess, LEP_GRID(nct, tj), ct, V.patches, H.patches, rhs, bcLo, bcHi, excess, dzScale);
}
void launch_lep_vsolve(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       double* phi, double* rhs, double* gam, const double* bcLo, const double* bcHi, double dz)
{
    if (nct) hipLaunchKernelGGL(k_lep_vsolve, LEP_GRID(nct, tj), ct, V.patches, H.patches, phi, rhs, V.jg[2], gam, bcLo, bcHi, dz);
}
void launch_lep_vsolve_lapack(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* phi,
                              const double* rhs, double* dfac, double* efac, const int* vbc, double dz, double dzCrse, int* bad)
{
    if (nct) hipLaunchKernelGGL(k_lep_vsolve_lapack, LEP_GRID(nct, tj), ct, V.patches, phi, rhs, V.jg[2], dfac, efac, vbc, dz,
                                dzCrse, bad);
}
void launch_lep_hgrad(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                      const double* phi, double* gx, double* gy)
{
    if (nct) hipLaunchKernelGGL(k_lep_hgrad, LEP_GRID(nct, tj), ct, V.patches, H.patches, phi, V.jg[0], V.jg[1], gx, gy, V.P);
}
static LJ lj_of(const LevelDev& V)
{
    LJ J;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) J.c[a][b] = V.jgf[a][b];
    return J;
}
void launch_lep_avg_metric_full(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H)
{
    if (!nct) return;
    HJ h;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) h.c[a][b] = H.jgf[a][b];
    hipLaunchKernelGGL(k_lep_avg_metric_full, LEP_GRID(nct, tj), ct, V.patches, H.patches, lj_of(V), h);
}
void launch_lep_hgrad_full(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                           const double* phi, double* gx, double* gy)
{
    if (nct) hipLaunchKernelGGL(k_lep_hgrad_full, LEP_GRID(nct, tj), ct, V.patches, H.patches, phi, lj_of(V), gx, gy, V.P);
}
void launch_lep_vhgrad(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       const double* phi, double* bcLo, double* bcHi, double scale)
{
    if (nct) hipLaunchKernelGGL(k_lep_vhgrad, LEP_GRID(nct, tj), ct, V.patches, H.patches, phi, V.jgf[2][0], V.jgf[2][1], bcLo,
                                bcHi, V.P, scale);
}
void launch_lep_hrhs(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                     const double* gx, const double* gy, const double* excess, double* hrhs, double sx, double sy,
                     double negInvH, bool useExcess)
{
    if (nct) hipLaunchKernelGGL(k_lep_hrhs, LEP_GRID(nct, tj), ct, V.patches, H.patches, gx, gy, excess, hrhs, sx, sy, negInvH, useExcess ? 1 : 0);
}
void launch_lep_extrude(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                        double* phi, const double* flat)
{
    if (nct) hipLaunchKernelGGL(k_lep_extrude, LEP_GRID(nct, tj), ct, V.patches, H.patches, phi, flat);
}
void launch_lep_divide(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* y, const double* x,
                       const double* b)
{
    if (nct) hipLaunchKernelGGL(k_lep_valid<0>, LEP_GRID(nct, tj), ct, V.patches, y, x, b, 0.0);
}
void launch_lep_axpy(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* y, const double* x,
                     double a)
{
    if (nct) hipLaunchKernelGGL(k_lep_valid<1>, LEP_GRID(nct, tj), ct, V.patches, y, x, nullptr, a);
}

}  // namespace somar
