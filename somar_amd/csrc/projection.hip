// somar_amd/csrc/projection.hip -- the two stencils either side of the pressure solve in a MAC level
// projection (diagonal metric), SURVEY.md rows a20-a22:
//
//   k_div_mac      rhs = Jinv * sum_a (U^a_{i+e_a} - U^a_i)/dx_a  [ / dt ]
//                  MAPPEDFLUXDIVERGENCE3D (calculus/DivCurlGrad/DivCurlGradF.ChF:1122-1215) as called by
//                  Divergence::levelDivergenceMAC (Divergence.cpp:44-127) + "rhs /= dt" (BaseProjectorI.H:250-257)
//   k_mac_correct  U^a -= dt * Jg^{aa} (phi_i - phi_{i-e_a})/dx_a on every a-face of every box
//                  MAPPEDMACGRADORTHO normal branch (DivCurlGradF.ChF:221-258) via Gradient::levelGradientMAC /
//                  singleBoxMacGrad (Gradient.cpp:85-206, 946-1101), physical-boundary ghosts by order-2
//                  extrapolation (ELLIPTICEXTRAPBCGHOST, EllipticBCUtilsF.ChF:165-173; BC holder
//                  PhysBCUtil.cpp:1432-1443) and LevelMACProjector::applyCorrection (LevelMACProjector.cpp:222-241)
//                  fused: the gradient temporary (3 face arrays written + read) never exists.
//   k_cell_to_edge U^a on a-faces = half*(u^a_i + u^a_{i-e_a}) from a cell-centred velocity with one ghost layer, and 0
//                  on physical boundary faces: Chombo CellToEdge (EXTERNAL) as called by Divergence::levelDivergenceCC
//                  (Divergence.cpp:361-396) + the solid-wall branch of BasicVelocityBCGhostClass, i.e. setSideDiriBC(0)
//                  on face-centred data (EllipticBCUtils.cpp:1284-1327, 96-100), applied by levelDivergenceMAC (:44-127)
//   k_cc_correct   u^a_i -= dt * half*(G^a_i + G^a_{i+e_a}),  G = the MAC gradient above: Gradient::levelGradientCC's
//                  levelGradientMAC + Chombo EdgeToCell (Gradient.cpp:469-495) + LevelCCProjector::applyCorrection
//                  (LevelCCProjector.cpp:232-255) in one pass; neither the face gradient nor its cell average is stored
//   k_edge_to_cell_axpy  the same average of a STORED face gradient (non-diagonal metric: singleBoxMacGrad's output)
// Same operation order as the reference => bit-identical to the oracle.
#include "common.h"
#include "kernels.h"

namespace somar {

__device__ __forceinline__ long long pidx(const PatchDesc& p, int i, int j, int k)
{
    return p.off + i + (long long)p.pj * j + p.pk * k;
}

__global__ __launch_bounds__(512) void k_div_mac(const Tile* __restrict__ tiles,
                                                 const PatchDesc* __restrict__ patches,
                                                 double* __restrict__ out, const double* __restrict__ u0,
                                                 const double* __restrict__ u1, const double* __restrict__ u2,
                                                 const double* __restrict__ jinv, StencilParams P, double dt)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const double dxinv0 = 1.0 / P.dx[0], dxinv1 = 1.0 / P.dx[1], dxinv2 = 1.0 / P.dx[2];
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const long long c = pidx(p, li, lj, t.k0 + kk);
            double d = jinv[c] * ((u0[c + 1] - u0[c]) * dxinv0 + (u1[c + p.pj] - u1[c]) * dxinv1 +
                                  (u2[c + p.pk] - u2[c]) * dxinv2);
            if (dt != 0.0) d = d / dt;
            out[c] = d;
        }
}

// One direction per launch.  Thread = cell (i-pair); it owns the LOW face of each of its cells and, for
// the last cell of the box in direction DIR, also the HIGH face.
template <int DIR>
__global__ __launch_bounds__(512) void k_mac_correct(const Tile* __restrict__ tiles,
                                                     const PatchDesc* __restrict__ patches,
                                                     double* __restrict__ vel, const double* __restrict__ phi,
                                                     const double* __restrict__ jg, StencilParams P, double dtScale)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const double dxinv = 1.0 / P.dx[DIR];
    const long long s = DIR == 0 ? 1 : (DIR == 1 ? (long long)p.pj : p.pk);
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const int lk = t.k0 + kk;
            const int l = DIR == 0 ? li : (DIR == 1 ? lj : lk);
            const int g = p.lo[DIR] + l;
            const long long c = pidx(p, li, lj, lk);
            const double pc = phi[c];
            // low face: the cell below is a physical ghost only on a non-periodic domain face
            double pm = phi[c - s];
            if (g == P.dom_lo[DIR] && !P.periodic[DIR]) pm = 3.0 * (pc - phi[c + s]) + phi[c + 2 * s];
            vel[c] = vel[c] + dtScale * (dxinv * jg[c] * (pc - pm));
            if (l == p.n[DIR] - 1) {
                double pp = phi[c + s];
                if (g == P.dom_hi[DIR] && !P.periodic[DIR]) pp = 3.0 * (pc - phi[c - s]) + phi[c - 2 * s];
                vel[c + s] = vel[c + s] + dtScale * (dxinv * jg[c + s] * (pp - pc));
            }
        }
}

// AlteredMetric::fill_Jgup (projection/AlteredMetric.cpp:82-198), the algebra after the map has been evaluated: one
// thread per face of the destination box, FArrayBox operation by FArrayBox operation (each line below is one of the
// reference's whole-FAB statements, so the roundings are the same).
__global__ void k_altered_jgup(long long n, double* __restrict__ dest, const double* __restrict__ nsq,
                               const double* __restrict__ dmu, const double* __restrict__ dnu,
                               const double* __restrict__ ix, const double* __restrict__ jy,
                               const double* __restrict__ iy, const double* __restrict__ jx,
                               const double* __restrict__ gup, const double* __restrict__ J, double theta,
                               double coriolisF, int offdiag)
{
    const double ftilde = coriolisF * theta;
    const double ftildesq = ftilde * ftilde;
    const double invfCoeff = 1.0 / (1.0 + ftildesq);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double d = nsq[i];
        d = d * (theta * theta);             // destAlias *= m_theta*m_theta
        const double t = d + 1.0;            // tmpFAB = destAlias + 1
        d = d / t;                           // destAlias /= tmpFAB
        d = d * -1.0;                        // = -omega^2/(1+omega^2)
        d = d + ftildesq * invfCoeff;        // + ftilde^2/(1+ftilde^2)
        d = d * dmu[i];                      // * dXi^mu/dz
        d = d * dnu[i];                      // * dXi^nu/dz
        if (offdiag) d = d + ftilde * invfCoeff * (ix[i] * jy[i] - iy[i] * jx[i]);
        d = d + gup[i] * invfCoeff;          // + g^{mu nu} * invfCoeff
        dest[i] = d * J[i];                  // * J * a_scale
    }
}

void launch_altered_jgup(hipStream_t st, long long n, double* dest, const double* nsq, const double* dmu,
                         const double* dnu, const double* ix, const double* jy, const double* iy, const double* jx,
                         const double* gup, const double* J, double theta, double coriolisF, bool offdiag)
{
    if (n <= 0) return;
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_altered_jgup, dim3(blocks), dim3(256), 0, st, n, dest, nsq, dmu, dnu, ix, jy, iy, jx, gup, J, theta,
                       coriolisF, offdiag ? 1 : 0);
}

// GeoSourceInterface's generic metric algebra, 3-D (geometry/GeoSourceInterface.cpp:200-450), after the map has handed over
// its Jacobian matrix dx^rho/dXi^sigma (x[3 * rho + sigma], each n values at the destination's centring) and det J:
//   fill_dXidx(mu, nu) = ((0 + A B) - C D) / J,  A = dx^{mu1}/dXi^{nu1}, B = dx^{mu2}/dXi^{nu2}, C = dx^{mu1}/dXi^{nu2},
//                        D = dx^{mu2}/dXi^{nu1}, mu1 = (nu+1)%3, mu2 = (nu+2)%3, nu1 = (mu+1)%3, nu2 = (mu+2)%3   (:236-291)
//   fill_gup(mu, nu)   = ((0 + dXidx(mu,0) dXidx(nu,0)) + dXidx(mu,1) dXidx(nu,1)) + dXidx(mu,2) dXidx(nu,2)        (:373-415)
//   fill_Jgup(mu, nu)  = gup * J [* scale when scale != 1]                                                         (:417-450)
// one whole-FAB statement of the reference per line, so the roundings are the same.  out: J g^{mu nu}, nu = 0..2 (3 n values)
struct DX9 { const double* x[9]; };
__global__ void k_jgup_from_dxdxi(long long n, int mu, DX9 X, const double* __restrict__ J, double scale, double* __restrict__ out)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double detJ = J[i];
        auto dXidx = [&](int a, int b) {
            const int mu1 = (b + 1) % 3, mu2 = (b + 2) % 3, nu1 = (a + 1) % 3, nu2 = (a + 2) % 3;
            double d = 0.0;
            d = d + X.x[3 * mu1 + nu1][i] * X.x[3 * mu2 + nu2][i];   // ADDPROD2
            d = d - X.x[3 * mu1 + nu2][i] * X.x[3 * mu2 + nu1][i];   // SUBPROD2
            return d / detJ;
        };
        double m[3];
        for (int rho = 0; rho < 3; ++rho) m[rho] = dXidx(mu, rho);
        for (int nu = 0; nu < 3; ++nu) {
            double g = 0.0;
            for (int rho = 0; rho < 3; ++rho) g = g + m[rho] * (nu == mu ? m[rho] : dXidx(nu, rho));
            g = g * detJ;
            if (scale != 1.0) g = g * scale;
            out[(long long)nu * n + i] = g;
        }
    }
}
void launch_jgup_from_dxdxi(hipStream_t st, long long n, int mu, const double* const x9[9], const double* J, double scale,
                            double* out)
{
    if (n <= 0) return;
    DX9 X;
    for (int q = 0; q < 9; ++q) X.x[q] = x9[q];
    const int blocks = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
    hipLaunchKernelGGL(k_jgup_from_dxdxi, dim3(blocks), dim3(256), 0, st, n, mu, X, J, scale, out);
}

// vel^a += s * g^a on every a-face of every box (low face at the cell, the last cell of a row also its high face)
struct FA3 { double* v[3]; const double* g[3]; };
__global__ __launch_bounds__(512) void k_face_axpy(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                   FA3 f, double s)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const long long st[3] = {1, (long long)p.pj, p.pk};
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int l[3] = {li0 + q, lj, t.k0 + kk};
            if (l[0] >= p.n[0]) continue;
            const long long c = p.off + l[0] + st[1] * l[1] + st[2] * l[2];
            for (int a = 0; a < 3; ++a) {
                if (!f.v[a]) continue;
                f.v[a][c] = f.v[a][c] + s * f.g[a][c];
                if (l[a] == p.n[a] - 1) f.v[a][c + st[a]] = f.v[a][c + st[a]] + s * f.g[a][c + st[a]];
            }
        }
}

void launch_face_axpy(hipStream_t st, const LevelDev& L, double* const vel[3], double* const grad[3], double s)
{
    if (L.ntiles == 0) return;
    FA3 f;
    for (int a = 0; a < 3; ++a) { f.v[a] = vel[a]; f.g[a] = grad[a]; }
    hipLaunchKernelGGL(k_face_axpy, dim3(L.ntiles), dim3(64, L.tile_j, 1), 0, st, L.tiles, L.patches, f, s);
}

void launch_div_mac(hipStream_t st, const LevelDev& L, double* out, const double* u0, const double* u1,
                    const double* u2, double dt)
{
    if (L.ntiles == 0) return;
    hipLaunchKernelGGL(k_div_mac, dim3(L.ntiles), dim3(64, L.tile_j, 1), 0, st, L.tiles, L.patches, out, u0, u1, u2,
                       L.jinv, L.P, dt);
}

void launch_mac_correct(hipStream_t st, const LevelDev& L, double* const vel[3], const double* phi, double dtScale)
{
    if (L.ntiles == 0) return;
    const dim3 g(L.ntiles), b(64, L.tile_j, 1);
    if (vel[0]) hipLaunchKernelGGL(k_mac_correct<0>, g, b, 0, st, L.tiles, L.patches, vel[0], phi, L.jg[0], L.P, dtScale);
    if (vel[1]) hipLaunchKernelGGL(k_mac_correct<1>, g, b, 0, st, L.tiles, L.patches, vel[1], phi, L.jg[1], L.P, dtScale);
    if (vel[2]) hipLaunchKernelGGL(k_mac_correct<2>, g, b, 0, st, L.tiles, L.patches, vel[2], phi, L.jg[2], L.P, dtScale);
}

// One direction per launch; thread = cell (i-pair), it owns the LOW face of each of its cells and, for the last cell of
// the box in direction DIR, also the HIGH face (the layout of k_mac_correct).  cc carries one filled ghost layer.
template <int DIR>
__global__ __launch_bounds__(512) void k_cell_to_edge(const Tile* __restrict__ tiles,
                                                      const PatchDesc* __restrict__ patches,
                                                      double* __restrict__ edge, const double* __restrict__ cc,
                                                      StencilParams P, int wall)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const long long s = DIR == 0 ? 1 : (DIR == 1 ? (long long)p.pj : p.pk);
    const bool walls = wall && !P.periodic[DIR];
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const int lk = t.k0 + kk;
            const int l = DIR == 0 ? li : (DIR == 1 ? lj : lk);
            const int g = p.lo[DIR] + l;
            const long long c = pidx(p, li, lj, lk);
            const double uc = cc[c];
            double v = 0.5 * (uc + cc[c - s]);
            if (walls && g == P.dom_lo[DIR]) v = 0.0;
            edge[c] = v;
            if (l == p.n[DIR] - 1) {
                double w = 0.5 * (cc[c + s] + uc);
                if (walls && g == P.dom_hi[DIR]) w = 0.0;
                edge[c + s] = w;
            }
        }
}

// BasicVelocityBCGhostClass on face-centred data, solid walls: setSideDiriBC(0) sets the wall-normal faces directly
// (EllipticBCUtils.cpp:1284-1327, 96-100) -- what levelDivergenceMAC does to the caller's velocity through a_fluxBC
template <int DIR>
__global__ __launch_bounds__(512) void k_face_wall(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                   double* __restrict__ edge, StencilParams P)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1] || P.periodic[DIR]) return;
    const long long s = DIR == 0 ? 1 : (DIR == 1 ? (long long)p.pj : p.pk);
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const int lk = t.k0 + kk;
            const int l = DIR == 0 ? li : (DIR == 1 ? lj : lk);
            const int g = p.lo[DIR] + l;
            const long long c = pidx(p, li, lj, lk);
            if (g == P.dom_lo[DIR]) edge[c] = 0.0;
            if (g == P.dom_hi[DIR]) edge[c + s] = 0.0;
        }
}

// BasicVelocityBCGhostClass with inflow / outflow sides (EllipticBCUtils.cpp:1244-1327) on face-centred data: per domain
// side kind 0 = solid wall, setSideDiriBC(0); 1 = prescribed normal velocity, setSideDiriBC(value) (on a face-centred FAB
// both set the boundary faces directly, :96-100); 2 = outflow, setSideExtrapBC(order 0) = ELLIPTICEXTRAPBCGHOST order 0
// on the boundary faces: the value of the next face inside (EllipticBCUtilsF.ChF:148-154)
struct FaceBC { int kind[6]; double value[6]; };
template <int DIR>
__global__ __launch_bounds__(512) void k_face_bc(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                 double* __restrict__ edge, StencilParams P, FaceBC B)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1] || P.periodic[DIR]) return;
    const long long s = DIR == 0 ? 1 : (DIR == 1 ? (long long)p.pj : p.pk);
    const int klo = B.kind[2 * DIR], khi = B.kind[2 * DIR + 1];
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int li = li0 + q;
            if (li >= p.n[0]) continue;
            const int lk = t.k0 + kk;
            const int l = DIR == 0 ? li : (DIR == 1 ? lj : lk);
            const int g = p.lo[DIR] + l;
            const long long c = pidx(p, li, lj, lk);
            if (g == P.dom_lo[DIR]) edge[c] = klo == 2 ? edge[c + s] : (klo == 1 ? B.value[2 * DIR] : 0.0);
            if (g == P.dom_hi[DIR]) edge[c + s] = khi == 2 ? edge[c] : (khi == 1 ? B.value[2 * DIR + 1] : 0.0);
        }
}

void launch_face_bc(hipStream_t st, const LevelDev& L, double* const edge[3], const int kind[6], const double value[6])
{
    if (L.ntiles == 0) return;
    FaceBC B;
    for (int i = 0; i < 6; ++i) { B.kind[i] = kind[i]; B.value[i] = value[i]; }
    const dim3 g(L.ntiles), b(64, L.tile_j, 1);
    if (edge[0]) hipLaunchKernelGGL(k_face_bc<0>, g, b, 0, st, L.tiles, L.patches, edge[0], L.P, B);
    if (edge[1]) hipLaunchKernelGGL(k_face_bc<1>, g, b, 0, st, L.tiles, L.patches, edge[1], L.P, B);
    if (edge[2]) hipLaunchKernelGGL(k_face_bc<2>, g, b, 0, st, L.tiles, L.patches, edge[2], L.P, B);
}

void launch_face_wall(hipStream_t st, const LevelDev& L, double* const edge[3])
{
    if (L.ntiles == 0) return;
    const dim3 g(L.ntiles), b(64, L.tile_j, 1);
    if (edge[0]) hipLaunchKernelGGL(k_face_wall<0>, g, b, 0, st, L.tiles, L.patches, edge[0], L.P);
    if (edge[1]) hipLaunchKernelGGL(k_face_wall<1>, g, b, 0, st, L.tiles, L.patches, edge[1], L.P);
    if (edge[2]) hipLaunchKernelGGL(k_face_wall<2>, g, b, 0, st, L.tiles, L.patches, edge[2], L.P);
}

struct CC3 { double* u[3]; const double* g[3]; };

// Diagonal metric: both face gradients of a cell from phi (exchanged; physical ghosts by order-2 extrapolation exactly as
// in k_mac_correct), their average, and the correction, per component.
__global__ __launch_bounds__(512) void k_cc_correct(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                    CC3 f, const double* __restrict__ phi, StencilParams P,
                                                    double dtScale)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const long long st[3] = {1, (long long)p.pj, p.pk};
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int l[3] = {li0 + q, lj, t.k0 + kk};
            if (l[0] >= p.n[0]) continue;
            const long long c = p.off + l[0] + st[1] * l[1] + st[2] * l[2];
            const double pc = phi[c];
            for (int a = 0; a < 3; ++a) {
                if (!f.u[a]) continue;
                const long long s = st[a];
                const int g = p.lo[a] + l[a];
                const double dxinv = 1.0 / P.dx[a];
                double pm = phi[c - s];
                if (g == P.dom_lo[a] && !P.periodic[a]) pm = 3.0 * (pc - phi[c + s]) + phi[c + 2 * s];
                double pp = phi[c + s];
                if (g == P.dom_hi[a] && !P.periodic[a]) pp = 3.0 * (pc - phi[c - s]) + phi[c - 2 * s];
                const double glo = dxinv * f.g[a][c] * (pc - pm);
                const double ghi = dxinv * f.g[a][c + s] * (pp - pc);
                f.u[a][c] = f.u[a][c] + dtScale * (0.5 * (glo + ghi));
            }
        }
}

__global__ __launch_bounds__(512) void k_edge_to_cell_axpy(const Tile* __restrict__ tiles,
                                                           const PatchDesc* __restrict__ patches, CC3 f, double dtScale)
{
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lj = t.j0 + threadIdx.y;
    const int li0 = t.i0 + 2 * threadIdx.x;
    if (lj >= p.n[1]) return;
    const long long st[3] = {1, (long long)p.pj, p.pk};
    for (int kk = 0; kk < t.nk; ++kk)
        for (int q = 0; q < 2; ++q) {
            const int l0 = li0 + q;
            if (l0 >= p.n[0]) continue;
            const long long c = p.off + l0 + st[1] * lj + st[2] * (t.k0 + kk);
            for (int a = 0; a < 3; ++a) {
                if (!f.u[a]) continue;
                f.u[a][c] = f.u[a][c] + dtScale * (0.5 * (f.g[a][c] + f.g[a][c + st[a]]));
            }
        }
}

void launch_cell_to_edge(hipStream_t st, const LevelDev& L, double* const edge[3], double* const cc[3], bool wall)
{
    if (L.ntiles == 0) return;
    const dim3 g(L.ntiles), b(64, L.tile_j, 1);
    const int w = wall ? 1 : 0;
    if (cc[0]) hipLaunchKernelGGL(k_cell_to_edge<0>, g, b, 0, st, L.tiles, L.patches, edge[0], cc[0], L.P, w);
    if (cc[1]) hipLaunchKernelGGL(k_cell_to_edge<1>, g, b, 0, st, L.tiles, L.patches, edge[1], cc[1], L.P, w);
    if (cc[2]) hipLaunchKernelGGL(k_cell_to_edge<2>, g, b, 0, st, L.tiles, L.patches, edge[2], cc[2], L.P, w);
}

void launch_cc_correct(hipStream_t st, const LevelDev& L, double* const cc[3], const double* phi, double dtScale)
{
    if (L.ntiles == 0) return;
    CC3 f;
    for (int a = 0; a < 3; ++a) { f.u[a] = cc[a]; f.g[a] = L.jg[a]; }
    hipLaunchKernelGGL(k_cc_correct, dim3(L.ntiles), dim3(64, L.tile_j, 1), 0, st, L.tiles, L.patches, f, phi, L.P, dtScale);
}

void launch_edge_to_cell_axpy(hipStream_t st, const LevelDev& L, double* const cc[3], double* const grad[3], double dtScale)
{
    if (L.ntiles == 0) return;
    CC3 f;
    for (int a = 0; a < 3; ++a) { f.u[a] = cc[a]; f.g[a] = grad[a]; }
    hipLaunchKernelGGL(k_edge_to_cell_axpy, dim3(L.ntiles), dim3(64, L.tile_j, 1), 0, st, L.tiles, L.patches, f, dtScale);
}

}  // namespace somar
