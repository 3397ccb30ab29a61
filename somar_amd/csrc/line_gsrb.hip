// somar_amd/csrc/line_gsrb.hip -- vertical-line Gauss-Seidel (relax_mode 3, LineGSRB), diagonal metric.
//
// Reference: LineGSRB::relax (RelaxationMethods/GSRB.cpp:148-330) -> LineGSRBIter3D
// (RelaxationMethods/GSRBF.ChF:1730-2042) -> LAPACK dgtsv per (i,j) column, called one column at a time
// with freshly allocated workspace FABs.
//
// Here: one launch per colour; a lane owns one (i-pair, j) column -- 64 adjacent columns per wavefront,
// so every load of the k-march is a unit-stride row segment -- and runs dgtsv's elimination in registers
// while it assembles the system (no D/DL/DU/B workspace: the modified diagonal goes to one scratch field,
// the modified right-hand side overwrites the column of phi it will replace anyway), then back-substitutes
// on the way down.  Columns of one colour are independent (their horizontal neighbours are the other
// colour), so there is no cross-lane dependence and no shuffle is needed; the serial recurrence lives in
// k, the slow index, exactly where the layout wants it.
//
// Arithmetic follows LineGSRBIter3D term by term (cross terms vanish identically for a diagonal metric)
// and dgtsv's no-interchange path (|d(i)| >= |dl(i)| always holds for these diagonally dominant systems):
//   fact = dl(i)/d(i); d(i+1) -= fact*du(i); b(i+1) -= fact*b(i);   x(n) = b(n)/d(n); x(i) = (b(i) - du(i)*x(i+1))/d(i)
// => bit-identical to the oracle (which is pinned against SciPy's LAPACK dgtsv).
// Deviation Q1: the reference's uninitialised jmin/jmax and parity-shifted imin are
// replaced by the intended test "cell on the box bound AND that side is Neumann", as in LineGSRBIter2D.
#include "common.h"
#include "kernels.h"

namespace somar {

struct LJ9 { const double* c[3][3]; };  // c[faceDir][component], non-diagonal metric

template <bool FULL>
__global__ __launch_bounds__(256) void k_line_gsrb_ortho(const Tile* __restrict__ tiles,
                                                         const PatchDesc* __restrict__ patches,
                                                         double* __restrict__ phi,
                                                         const double* __restrict__ rhs,
                                                         const double* __restrict__ jgx,
                                                         const double* __restrict__ jgy,
                                                         const double* __restrict__ jgz,
                                                         const double* __restrict__ jinv,
                                                         double* __restrict__ dmod, StencilParams P, int color,
                                                         LJ9 J, const double* __restrict__ E)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1] || li0 >= p.n[0]) return;
    const int gj = p.lo[1] + lj;
    const int li = li0 + ((p.lo[0] + li0 + gj + color) & 1);  // colour by i+j only: whole columns are solved
    if (li >= p.n[0]) return;
    const int gi = p.lo[0] + li;
    const int N = p.n[2];
    const long long sj = p.pj, sk = p.pk;
    const double xxScale = P.beta * 1.0 / (P.dx[0] * P.dx[0]);
    const double yyScale = P.beta * 1.0 / (P.dx[1] * P.dx[1]);
    const double zzScale = P.beta * 1.0 / (P.dx[2] * P.dx[2]);
    // BC codes of the box: Neumann iff it touches a non-periodic domain face; vertical ends: coeff1 = 0 for
    // Neumann and for "None" alike (GSRBF.ChF:1804-1817); Dirichlet ends: see c1lo / c1hi below
    const bool nxl = (gi == P.dom_lo[0]) && P.neum[0][0];
    const bool nxh = (gi == P.dom_hi[0]) && P.neum[0][1];
    const bool nyl = (gj == P.dom_lo[1]) && P.neum[1][0];
    const bool nyh = (gj == P.dom_hi[1]) && P.neum[1][1];
    // vertical ends of the column: Neumann and "None" (a box boundary inside the domain) 0, Dirichlet 2 -- the ghost
    // -phi folded into the diagonal (GSRBF.ChF:1804-1817); CF ends are not offered
    const double c1lo = (p.lo[2] == P.dom_lo[2] && P.diri[2][0]) ? 2.0 : 0.0;
    const double c1hi = (p.lo[2] + N - 1 == P.dom_hi[2] && P.diri[2][1]) ? 2.0 : 0.0;
    long long c = p.off + li + sj * lj;  // k = 0
    double d_prev = 0.0, b_prev = 0.0, dl_prev = 0.0;
    for (int k = 0; k < N; ++k, c += sk) {
        const double gzl = jgz[c], gzh = jgz[c + sk];
        double lapDiag;
        if (k == 0) lapDiag = -zzScale * (gzh + c1lo * gzl);
        else if (k == N - 1) lapDiag = -zzScale * (c1hi * gzh + gzl);
        else lapDiag = -zzScale * (gzl + gzh);
        if (N == 1) lapDiag = -zzScale * (c1hi * gzh + c1lo * gzl);
        double JDxx = 0.0, JDyy = 0.0;
        if (!nxl) { JDxx = JDxx + jgx[c] * phi[c - 1];        lapDiag = lapDiag - xxScale * jgx[c]; }
        if (!nxh) { JDxx = JDxx + jgx[c + 1] * phi[c + 1];    lapDiag = lapDiag - xxScale * jgx[c + 1]; }
        if (!nyl) { JDyy = JDyy + jgy[c] * phi[c - sj];       lapDiag = lapDiag - yyScale * jgy[c]; }
        if (!nyh) { JDyy = JDyy + jgy[c + sj] * phi[c + sj];  lapDiag = lapDiag - yyScale * jgy[c + sj]; }
        double lphi;
        if (FULL) {
            // cross terms from the extrapolated copy (explicit, as in LineGSRBIter3D): GSRBF.ChF:1730-2042
            const double xyScale = P.beta * 0.25 / (P.dx[0] * P.dx[1]);
            const double yzScale = P.beta * 0.25 / (P.dx[1] * P.dx[2]);
            const double zxScale = P.beta * 0.25 / (P.dx[2] * P.dx[0]);
#define EE(di, dj, dk) E[c + (di) + sj * (dj) + sk * (dk)]
            const double JDxy = J.c[0][1][c + 1] * (EE(1, 1, 0) - EE(1, -1, 0) + EE(0, 1, 0) - EE(0, -1, 0)) -
                                J.c[0][1][c] * (EE(0, 1, 0) - EE(0, -1, 0) + EE(-1, 1, 0) - EE(-1, -1, 0));
            const double JDxz = J.c[0][2][c + 1] * (EE(1, 0, 1) - EE(1, 0, -1) + EE(0, 0, 1) - EE(0, 0, -1)) -
                                J.c[0][2][c] * (EE(0, 0, 1) - EE(0, 0, -1) + EE(-1, 0, 1) - EE(-1, 0, -1));
            const double JDyx = J.c[1][0][c + sj] * (EE(1, 1, 0) - EE(-1, 1, 0) + EE(1, 0, 0) - EE(-1, 0, 0)) -
                                J.c[1][0][c] * (EE(1, 0, 0) - EE(-1, 0, 0) + EE(1, -1, 0) - EE(-1, -1, 0));
            const double JDyz = J.c[1][2][c + sj] * (EE(0, 1, 1) - EE(0, 1, -1) + EE(0, 0, 1) - EE(0, 0, -1)) -
                                J.c[1][2][c] * (EE(0, 0, 1) - EE(0, 0, -1) + EE(0, -1, 1) - EE(0, -1, -1));
            const double JDzx = J.c[2][0][c + sk] * (EE(1, 0, 1) - EE(-1, 0, 1) + EE(1, 0, 0) - EE(-1, 0, 0)) -
                                J.c[2][0][c] * (EE(1, 0, 0) - EE(-1, 0, 0) + EE(1, 0, -1) - EE(-1, 0, -1));
            const double JDzy = J.c[2][1][c + sk] * (EE(0, 1, 1) - EE(0, -1, 1) + EE(0, 1, 0) - EE(0, -1, 0)) -
                                J.c[2][1][c] * (EE(0, 1, 0) - EE(0, -1, 0) + EE(0, 1, -1) - EE(0, -1, -1));
#undef EE
            lphi = JDxx * xxScale + JDyy * yyScale + (JDyz + JDzy) * yzScale + (JDzx + JDxz) * zxScale +
                   (JDxy + JDyx) * xyScale;
        } else {
            lphi = JDxx * xxScale + JDyy * yyScale;
        }
        const double Ji = jinv[c];
        double B = -lphi + rhs[c] / Ji;
        double D = P.alpha / Ji + lapDiag;
        if (k > 0) {  // dgtsv elimination step i = k-1 (no interchange)
            const double fact = dl_prev / d_prev;
            D = D - fact * dl_prev;  // du(i) == dl(i): the system is symmetric
            B = B - fact * b_prev;
        }
        dmod[c] = D;
        phi[c] = B;
        d_prev = D;
        b_prev = B;
        dl_prev = gzh * zzScale;  // DL(k) = DU(k) = Jg2(k+1) * zzScale
    }
    // back substitution
    c -= sk;  // k = N-1
    double x = b_prev / d_prev;
    phi[c] = x;
    for (int k = N - 2; k >= 0; --k) {
        c -= sk;
        const double du = jgz[c + sk] * zzScale;
        x = (phi[c] - du * x) / dmod[c];
        phi[c] = x;
    }
}

void launch_line_gsrb_ortho(hipStream_t st, const Tile* ctiles, int nctiles, int tile_j, const LevelDev& L,
                            double* phi, const double* rhs, double* dmod, int color, const double* psi)
{
    if (nctiles == 0) return;
    LJ9 J;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) J.c[a][b] = L.jgf[a][b];
    if (psi)
        hipLaunchKernelGGL(k_line_gsrb_ortho<true>, dim3(nctiles), dim3(64, tile_j, 1), 0, st, ctiles, L.patches, phi, rhs,
                           L.jg[0], L.jg[1], L.jg[2], L.jinv, dmod, L.P, color, J, psi);
    else
        hipLaunchKernelGGL(k_line_gsrb_ortho<false>, dim3(nctiles), dim3(64, tile_j, 1), 0, st, ctiles, L.patches, phi, rhs,
                           L.jg[0], L.jg[1], L.jg[2], L.jinv, dmod, L.P, color, J, psi);
}

// ------------------------------------------------------------------------------------------------------------------------------
// CH_SPACEDIM = 2: LineGSRBIter2D (GSRBF.ChF:1529-1724).  The vertical is direction 1; one colour = the columns i with
// i + colour even (the 2-D routine colours by i alone); a lane owns one column, 64 adjacent columns per wavefront (half of
// them of the colour), and marches in j with the same in-register dgtsv elimination as the 3-D kernel.  The x neighbour of a
// column on the box bound is dropped iff that side is a Neumann domain face (the routine tests the REGION bound, :1607-1611).
// FULL: the cross terms J g^{01}, J g^{10} from the extrapolated copy E, explicit, as written.
// grid (chunks of 64 columns, patches), 64 threads.
// ------------------------------------------------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(64) void k_line_gsrb_2d(const PatchDesc* __restrict__ patches, double* __restrict__ phi,
                                                     const double* __restrict__ rhs, const double* __restrict__ jgx,
                                                     const double* __restrict__ jgy, const double* __restrict__ jinv,
                                                     double* __restrict__ dmod, StencilParams P, int color,
                                                     const double* __restrict__ jg01, const double* __restrict__ jg10,
                                                     const double* __restrict__ E)
{
    const PatchDesc p = patches[blockIdx.y];
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= p.n[0]) return;
    const int gi = p.lo[0] + li;
    if ((gi + color) & 1) return;   // imin = lbound + |mod(lbound + redBlack, 2)|, step 2: columns with i + redBlack even
    const int N = p.n[1];
    const long long sj = p.pj;
    const double xxScale = P.beta * 1.0 / (P.dx[0] * P.dx[0]);
    const double yyScale = P.beta * 1.0 / (P.dx[1] * P.dx[1]);
    const double xyScale = P.beta * 0.25 / (P.dx[0] * P.dx[1]);
    const bool nxl = (gi == P.dom_lo[0]) && P.neum[0][0];
    const bool nxh = (gi == P.dom_hi[0]) && P.neum[0][1];
    // vertical ends: Neumann and "None" 0, Dirichlet 2 (CF ends are refused at define)
    const double c1lo = (p.lo[1] == P.dom_lo[1] && P.diri[1][0]) ? 2.0 : 0.0;
    const double c1hi = (p.lo[1] + N - 1 == P.dom_hi[1] && P.diri[1][1]) ? 2.0 : 0.0;
    long long c = p.off + li;   // j = 0 (k = 0: the one plane of a 2-D level)
    double d_prev = 0.0, b_prev = 0.0, dl_prev = 0.0;
    for (int j = 0; j < N; ++j, c += sj) {
        const double gyl = jgy[c], gyh = jgy[c + sj];
        double lapDiag;
        if (j == 0) lapDiag = -yyScale * (gyh + c1lo * gyl);
        else if (j == N - 1) lapDiag = -yyScale * (c1hi * gyh + gyl);
        else lapDiag = -yyScale * (gyl + gyh);
        double JDxx = 0.0;
        if (!nxl) { JDxx = JDxx + jgx[c] * phi[c - 1];      lapDiag = lapDiag - xxScale * jgx[c]; }
        if (!nxh) { JDxx = JDxx + jgx[c + 1] * phi[c + 1];  lapDiag = lapDiag - xxScale * jgx[c + 1]; }
        double lphi;
        if (FULL) {
#define EE(di, dj) E[c + (di) + sj * (dj)]
            const double JDxy = jg01[c + 1] * (EE(1, 1) - EE(1, -1) + EE(0, 1) - EE(0, -1)) -
                                jg01[c] * (EE(0, 1) - EE(0, -1) + EE(-1, 1) - EE(-1, -1));
            const double JDyx = jg10[c + sj] * (EE(1, 1) - EE(-1, 1) + EE(1, 0) - EE(-1, 0)) -
                                jg10[c] * (EE(1, 0) - EE(-1, 0) + EE(1, -1) - EE(-1, -1));
#undef EE
            lphi = JDxx * xxScale + (JDxy + JDyx) * xyScale;
        } else {
            lphi = JDxx * xxScale;   // J g^{01} = J g^{10} = 0: (0 + 0) * xyScale adds nothing
        }
        const double Ji = jinv[c];
        double B = -lphi + rhs[c] / Ji;
        double D = P.alpha / Ji + lapDiag;
        if (j > 0) {   // dgtsv elimination step i = j-1 (no interchange), du(i) == dl(i)
            const double fact = dl_prev / d_prev;
            D = D - fact * dl_prev;
            B = B - fact * b_prev;
        }
        dmod[c] = D;
        phi[c] = B;
        d_prev = D;
        b_prev = B;
        dl_prev = gyh * yyScale;   // DL(j) = DU(j) = Jg1(j+1) * yyScale
    }
    c -= sj;
    double x = b_prev / d_prev;
    phi[c] = x;
    for (int j = N - 2; j >= 0; --j) {
        c -= sj;
        const double du = jgy[c + sj] * yyScale;
        x = (phi[c] - du * x) / dmod[c];
        phi[c] = x;
    }
}

void launch_line_gsrb_2d(hipStream_t st, const LevelDev& L, int maxN0, double* phi, const double* rhs, double* dmod, int color,
                         const double* psi)
{
    if (L.npatches == 0) return;
    const dim3 grid((maxN0 + 63) / 64, L.npatches);
    if (psi)
        hipLaunchKernelGGL(k_line_gsrb_2d<true>, grid, dim3(64), 0, st, L.patches, phi, rhs, L.jg[0], L.jg[1], L.jinv, dmod, L.P,
                           color, L.jgf[0][1], L.jgf[1][0], psi);
    else
        hipLaunchKernelGGL(k_line_gsrb_2d<false>, grid, dim3(64), 0, st, L.patches, phi, rhs, L.jg[0], L.jg[1], L.jinv, dmod, L.P,
                           color, nullptr, nullptr, psi);
}

}  // namespace somar
