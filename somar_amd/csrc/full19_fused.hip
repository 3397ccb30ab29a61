// somar_amd/csrc/full19_fused.hip -- LevelGSRB with a NON-diagonal metric: red and black in ONE k-marching launch for the cells
// whose black update reads nothing the reference refreshes between the colours.
//
// Why: the two-pass form (full19_march.hip, MODE 2) streams the seven to nine J g^{ab} planes once per colour -- each face
// coefficient is needed by one red and one black cell -- 98 B/cell and pass measured, 196 per sweep for 120 algorithmic, and on
// large boxes the passes already run at 80 % of what HBM delivers to such a stream mix.  Only fewer bytes make the sweep faster.
//
// What can be fused: between the colours the reference exchanges ghosts, re-interpolates coarse-fine ghosts, re-extrapolates
// the copy `extrap` at the box edges and re-evaluates the cross-term Neumann ghosts (RelaxationMethod::fillGhostsAndExtrapolate,
// RelaxationMethod.cpp:376-435; ExtrapolateFaceAndCopy, ExtrapolationUtils.cpp:109-155).  All of that READS the post-red /
// pre-black state of the three cell layers next to a box face (quadratic extrapolation: three cells) and WRITES the one-cell
// frame.  So
//   (A) this kernel: red(k) from the old values, then black(k-1) from the new red of planes k-2, k-1, k for the cells at least
//       THREE layers inside their box; black cells of the three outer layers keep their old value.  Red is recomputed on a
//       one-cell ring around the tile (as gsrb_fused.hip does), which costs coefficient loads on 6 rows for 4 output rows;
//   (B) the caller runs the between-colour ghost work on the result exactly as before;
//   (C) the black cells of the three outer layers go through the two-pass kernel's MODE 3 (full19_march.hip): a black cell
//       reads its twelve same-colour neighbours at their PRE-pass value (the reference reads them from the snapshot `extrap`), and
//       since red never writes a black cell that value is the sweep's INPUT array -- MODE 3 merges red from (A)'s output with
//       black from the input on load and stores shell cells only.
// Traffic of (A) per cell: coefficients 1.5 x 56-72, phi 16, rhs 12, 1/J 12, out 8 = 132-156 B against 196-228 for two passes; the
// shell (C) adds 3.5 % of the cells of a 512^3 box, 13 % of a 128^3 box, 26 % of a 64^3 box.
//
// MEASURED (1 x MI355X, bathymetric metric, profiles/r03_fused19.txt): bit-identical to the two-pass form and to the oracle, and
// SLOWER at every box size: one 512^3 box 9.11 ms for (A) + 0.38 ms for (C) against 2 x 2.85 ms; 128^3 boxes 14.9 against 6.6 ms per
// sweep, 64^3 boxes 16.0 against 6.0.  Why: both colours' coefficient sets live in registers at once (219-235 VGPRs), so a CU
// holds ONE 8-wave workgroup whose waves march in lock step -- a plane costs the memory latency PLUS both colours' arithmetic
// (7.1 us against the two-pass kernel's 3.3 us per plane and workgroup) for 4 output rows instead of 6: 1.95 TB/s of its own
// 132 B/cell.  A second register set to prefetch the next plane (which would overlap the two) does not fit in 256 VGPRs.  The
// path therefore stays OFF by default (PressureSolver::fused19_min_box_ < 0; SOMAR_FUSED19_MIN_BOX=0 forces it for the parity tests).
//
// Arithmetic: GSRBITER3D / GSRBBOUNDARYITER3D's expression order as in k_full_march<2> (GSRBF.ChF:36-282, 1024-1253): same bits.
#include "common.h"
#include "kernels.h"

namespace somar {

constexpr int FF_J = 8;   // region rows: 1 halo + 1 red ring + 4 output rows + 1 red ring + 1 halo
constexpr int FF_S = 4;   // plane slots
constexpr int FF_I = 128;

struct JgFullF { const double* c[3][3]; };

__device__ __forceinline__ double2 ff_ld2(const double* __restrict__ a, long long idx, bool ok0, bool ok1, long long safe)
{
    const double2 v = *reinterpret_cast<const double2*>(a + ((ok0 || ok1) ? idx : safe));
    return make_double2(ok0 ? v.x : 0.0, ok1 ? v.y : 0.0);
}
__device__ __forceinline__ double ff_pick(const double2& v, int s) { return s ? v.y : v.x; }

template <bool ZXY>
__global__ __launch_bounds__(64 * FF_J) void k_full_fused(const Tile* __restrict__ tiles, const PatchDesc* __restrict__ patches,
                                                          double* __restrict__ out, const double* __restrict__ phi,
                                                          const double* __restrict__ psi, const double* __restrict__ rhs,
                                                          JgFullF J, const double* __restrict__ jinv, StencilParams P)
{
    __shared__ __attribute__((aligned(16))) double SP[FF_S][FF_J][FF_I];  // phi, old
    __shared__ __attribute__((aligned(16))) double SE[FF_S][FF_J][FF_I];  // E, old: phi inside the box, psi in its frame
    __shared__ __attribute__((aligned(16))) double SU[FF_S][FF_J][FF_I];  // phi after the red pass
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int lane = threadIdx.x, row = threadIdx.y;
    const int ri = 2 * lane;
    const int li = t.i0 - 2 + ri;  // even: rows are 16-byte aligned
    const int wi = t.pad_[0] > 0 ? t.pad_[0] : FF_I - 4;
    const int lj = t.j0 - 2 + row;
    const int gj = p.lo[1] + lj;

    // phi / psi may be touched inside the 1-cell ghost layer; coefficients only at cells of the box
    const bool fj = (lj >= -1) && (lj <= p.n[1]);
    const bool f0 = fj && (li >= -1) && (li <= p.n[0]) && (ri < wi + 4);
    const bool f1 = fj && (li + 1 >= -1) && (li + 1 <= p.n[0]) && (ri + 1 < wi + 4);
    const bool inj = (lj >= 0) && (lj < p.n[1]);
    const bool in0 = inj && (li >= 0) && (li < p.n[0]);
    const bool in1 = inj && (li + 1 >= 0) && (li + 1 < p.n[0]);
    // red is computed on the tile grown by one cell, inside the box; c[s] = this cell takes part (in i, j)
    const bool ring_j = inj && (row >= 1) && (row <= FF_J - 2) && (lj <= t.j0 + (FF_J - 4));
    const bool own_j = inj && (row >= 2) && (row <= FF_J - 3) && (lj < t.j0 + (FF_J - 4));
    bool c[2], o[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int l = li + s, r = ri + s;
        c[s] = ring_j && (l >= 0) && (l < p.n[0]) && (r >= 1) && (r <= wi + 2);
        o[s] = own_j && (l >= 0) && (l < p.n[0]) && (r >= 2) && (r < wi + 2);
    }
    const int left_c1 = __shfl_up((int)c[1], 1, 64);
    const bool gxc0 = c[0] || (left_c1 != 0);
    const bool anyc = c[0] || c[1];
    const bool anyo = o[0] || o[1];
    const long long sj = p.pj, sk = p.pk;
    const long long base = p.off + li + sj * lj;
    // layers from the box faces in i and j (the k part joins per plane)
    int lay_ij[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int l = li + s;
        lay_ij[s] = min(min(l, p.n[0] - 1 - l), min(lj, p.n[1] - 1 - lj));
    }

    const double xxScale = 1.0 / (P.dx[0] * P.dx[0]);
    const double yyScale = 1.0 / (P.dx[1] * P.dx[1]);
    const double zzScale = 1.0 / (P.dx[2] * P.dx[2]);
    const double xyScale = 0.25 / (P.dx[0] * P.dx[1]);
    const double yzScale = 0.25 / (P.dx[1] * P.dx[2]);
    const double zxScale = 0.25 / (P.dx[2] * P.dx[0]);

    auto load_plane = [&](int kp, double2& vp, double2& ve) {
        const bool fk = (kp >= -1) && (kp <= p.n[2]);
        const bool ink = (kp >= 0) && (kp < p.n[2]);
        const long long idx = base + sk * kp;
        vp = ff_ld2(phi, idx, f0 && fk, f1 && fk, p.off);
        const bool e0 = f0 && fk && !(in0 && ink), e1 = f1 && fk && !(in1 && ink);   // frame cells: E = psi
        const double2 vs = ff_ld2(psi, idx, e0, e1, p.off);
        ve = make_double2(e0 ? vs.x : vp.x, e1 ? vs.y : vp.y);
    };
    auto store_plane = [&](int kp, const double2& vp, const double2& ve) {
        const int slot = kp & (FF_S - 1);
        *reinterpret_cast<double2*>(&SP[slot][row][ri]) = vp;
        *reinterpret_cast<double2*>(&SE[slot][row][ri]) = ve;
    };

    int k = t.k0 - 1;                    // first red plane: the ring below the tile
    const int kend = t.k0 + t.nk;        // last red plane: the ring above it
    double2 Pc;                          // old phi of plane k (this lane's pair)
    {
        double2 vp, ve;
        load_plane(k - 1, vp, ve);
        store_plane(k - 1, vp, ve);
        load_plane(k, vp, ve);
        store_plane(k, vp, ve);
        Pc = vp;
    }
    // k-face components on the LOW face of plane k
    const bool ck0 = (k >= 0) && (k <= p.n[2]);
    double2 Jz0c = ff_ld2(J.c[2][0], base + sk * k, c[0] && ck0, c[1] && ck0, p.off);
    double2 Jz1c = ff_ld2(J.c[2][1], base + sk * k, c[0] && ck0, c[1] && ck0, p.off);
    double2 Jz2c = ff_ld2(J.c[2][2], base + sk * k, c[0] && ck0, c[1] && ck0, p.off);
    // the black cell of plane k-1: its coefficients (captured one step earlier) and the pair's values after the red pass
    double b_jx0l = 0, b_jx0h = 0, b_jx1l = 0, b_jx1h = 0, b_jx2l = 0, b_jx2h = 0;
    double b_jy0l = 0, b_jy0h = 0, b_jy1l = 0, b_jy1h = 0, b_jy2l = 0, b_jy2h = 0;
    double b_jz0l = 0, b_jz0h = 0, b_jz1l = 0, b_jz1h = 0, b_jz2l = 0, b_jz2h = 0;
    double b_ji = 1.0, b_rh = 0.0;
    double2 Uprev = make_double2(0.0, 0.0);

    for (; k <= kend; ++k) {
        const int gk = p.lo[2] + k;
        const bool ink = (k >= 0) && (k < p.n[2]);       // coefficients of plane k exist (cells of the box)
        const bool inkp = (k + 1 >= 0) && (k + 1 <= p.n[2]);  // k-face k+1 exists
        // ---- this step's loads: phi / psi of plane k+1, coefficients of plane k ----
        double2 vp, ve;
        load_plane(k + 1, vp, ve);
        const long long ck = base + sk * k;
        const bool a0 = c[0] && ink, a1 = c[1] && ink;
        const double2 Jz0p = ff_ld2(J.c[2][0], ck + sk, c[0] && inkp, c[1] && inkp, p.off);
        const double2 Jz1p = ff_ld2(J.c[2][1], ck + sk, c[0] && inkp, c[1] && inkp, p.off);
        const double2 Jz2p = ff_ld2(J.c[2][2], ck + sk, c[0] && inkp, c[1] && inkp, p.off);
        const double2 Rh = ff_ld2(rhs, ck, a0, a1, p.off);
        const double2 Ji = ff_ld2(jinv, ck, a0, a1, p.off);
        const double2 Jx0 = ff_ld2(J.c[0][0], ck, gxc0 && ink, anyc && ink, p.off);
        const double2 Jx1 = ZXY ? make_double2(0.0, 0.0) : ff_ld2(J.c[0][1], ck, gxc0 && ink, anyc && ink, p.off);
        const double2 Jx2 = ff_ld2(J.c[0][2], ck, gxc0 && ink, anyc && ink, p.off);
        const double2 Jy0 = ZXY ? make_double2(0.0, 0.0) : ff_ld2(J.c[1][0], ck, a0, a1, p.off);
        const double2 Jy1 = ff_ld2(J.c[1][1], ck, a0, a1, p.off);
        const double2 Jy2 = ff_ld2(J.c[1][2], ck, a0, a1, p.off);
        const double2 Jy0h = ZXY ? make_double2(0.0, 0.0) : ff_ld2(J.c[1][0], ck + sj, a0, a1, p.off);
        const double2 Jy1h = ff_ld2(J.c[1][1], ck + sj, a0, a1, p.off);
        const double2 Jy2h = ff_ld2(J.c[1][2], ck + sj, a0, a1, p.off);
        const double jx0n = __shfl_down(Jx0.x, 1, 64), jx1n = __shfl_down(Jx1.x, 1, 64), jx2n = __shfl_down(Jx2.x, 1, 64);

        store_plane(k + 1, vp, ve);
        __syncthreads();

        const int sm = (k - 1) & (FF_S - 1), sc = k & (FF_S - 1), sp = (k + 1) & (FF_S - 1);
        const int csel = (p.lo[0] + li + gj + gk) & 1;   // the pair's RED cell in this plane (red = pass 0 = even i+j+k)
        // coefficients on the low / high faces of the pair's cell s
#define FF_COEF(s, jx0l, jx0h, jx1l, jx1h, jx2l, jx2h, jy0l, jy0h, jy1l, jy1h, jy2l, jy2h, jz0l, jz0h, jz1l, jz1h, jz2l, jz2h) \
        const double jx0l = (s) ? Jx0.y : Jx0.x, jx0h = (s) ? jx0n : Jx0.y;                                                  \
        const double jx1l = (s) ? Jx1.y : Jx1.x, jx1h = (s) ? jx1n : Jx1.y;                                                  \
        const double jx2l = (s) ? Jx2.y : Jx2.x, jx2h = (s) ? jx2n : Jx2.y;                                                  \
        const double jy0l = ff_pick(Jy0, s), jy0h = ff_pick(Jy0h, s);                                                        \
        const double jy1l = ff_pick(Jy1, s), jy1h = ff_pick(Jy1h, s);                                                        \
        const double jy2l = ff_pick(Jy2, s), jy2h = ff_pick(Jy2h, s);                                                        \
        const double jz0l = ff_pick(Jz0c, s), jz0h = ff_pick(Jz0p, s);                                                       \
        const double jz1l = ff_pick(Jz1c, s), jz1h = ff_pick(Jz1p, s);                                                       \
        const double jz2l = ff_pick(Jz2c, s), jz2h = ff_pick(Jz2p, s);
        // ---- red(k): the two-pass kernel's colour-0 update, from the OLD values ----
        double2 U = Pc;
        if (anyc && ink && c[csel]) {
            const int s = csel;
            const int rc = ri + s;
            const double pc = ff_pick(Pc, s);
            const int gi = p.lo[0] + li + s;
#define Pn(di, dj, dk) SP[(dk) < 0 ? sm : ((dk) > 0 ? sp : sc)][row + (dj)][rc + (di)]
#define En(di, dj, dk) SE[(dk) < 0 ? sm : ((dk) > 0 ? sp : sc)][row + (dj)][rc + (di)]
            FF_COEF(s, jx0l, jx0h, jx1l, jx1h, jx2l, jx2h, jy0l, jy0h, jy1l, jy1h, jy2l, jy2h, jz0l, jz0h, jz1l, jz1h, jz2l, jz2h)
            const double ji = ff_pick(Ji, s);
            const double rh = ff_pick(Rh, s);
            (void)pc;
            const bool onb = (gi == P.dom_lo[0]) || (gi == P.dom_hi[0]) || (gj == P.dom_lo[1]) || (gj == P.dom_hi[1]) ||
                             (gk == P.dom_lo[2]) || (gk == P.dom_hi[2]);
            double r;
            if (!onb) {
                // GSRBITER3D (GSRBF.ChF:36-282)
                const double pdx = En(1, 0, 0) - En(-1, 0, 0);
                const double pdy = En(0, 1, 0) - En(0, -1, 0);
                const double pdz = En(0, 0, 1) - En(0, 0, -1);
                const double JDxx = jx0h * Pn(1, 0, 0) + jx0l * Pn(-1, 0, 0);
                const double JDxy = jx1h * (En(1, 1, 0) - En(1, -1, 0) + pdy) - jx1l * (pdy + En(-1, 1, 0) - En(-1, -1, 0));
                const double JDxz = jx2h * (En(1, 0, 1) - En(1, 0, -1) + pdz) - jx2l * (pdz + En(-1, 0, 1) - En(-1, 0, -1));
                const double JDyx = jy0h * (En(1, 1, 0) - En(-1, 1, 0) + pdx) - jy0l * (pdx + En(1, -1, 0) - En(-1, -1, 0));
                const double JDyy = jy1h * Pn(0, 1, 0) + jy1l * Pn(0, -1, 0);
                const double JDyz = jy2h * (En(0, 1, 1) - En(0, 1, -1) + pdz) - jy2l * (pdz + En(0, -1, 1) - En(0, -1, -1));
                const double JDzx = jz0h * (En(1, 0, 1) - En(-1, 0, 1) + pdx) - jz0l * (pdx + En(1, 0, -1) - En(-1, 0, -1));
                const double JDzy = jz1h * (En(0, 1, 1) - En(0, -1, 1) + pdy) - jz1l * (pdy + En(0, 1, -1) - En(0, -1, -1));
                const double JDzz = jz2h * Pn(0, 0, 1) + jz2l * Pn(0, 0, -1);
                const double lphi = P.beta * ji *
                                    (JDxx * xxScale + JDyy * yyScale + JDzz * zzScale + (JDxy + JDyx) * xyScale +
                                     (JDyz + JDzy) * yzScale + (JDzx + JDxz) * zxScale);
                const double lapd = -ji * ((jx0h + jx0l) * xxScale + (jy1h + jy1l) * yyScale + (jz2h + jz2l) * zzScale);
                r = (rh - lphi) / (P.alpha + P.beta * lapd);
            } else {
                // GSRBBOUNDARYITER3D (GSRBF.ChF:1024-1253)
                const bool nxl = (gi == P.dom_lo[0]) && P.neum[0][0];
                const bool nxh = (gi == P.dom_hi[0]) && P.neum[0][1];
                const bool nyl = (gj == P.dom_lo[1]) && P.neum[1][0];
                const bool nyh = (gj == P.dom_hi[1]) && P.neum[1][1];
                const bool nzl = (gk == P.dom_lo[2]) && P.neum[2][0];
                const bool nzh = (gk == P.dom_hi[2]) && P.neum[2][1];
                double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0, ld = 0.0;
                if (!nxl) {
                    JDloX = +xxScale * jx0l * Pn(-1, 0, 0) -
                            xyScale * jx1l * (En(0, 1, 0) - En(0, -1, 0) + En(-1, 1, 0) - En(-1, -1, 0)) -
                            zxScale * jx2l * (En(0, 0, 1) - En(0, 0, -1) + En(-1, 0, 1) - En(-1, 0, -1));
                    ld = ld - xxScale * jx0l;
                }
                if (!nxh) {
                    JDhiX = +xxScale * jx0h * Pn(1, 0, 0) +
                            xyScale * jx1h * (En(1, 1, 0) - En(1, -1, 0) + En(0, 1, 0) - En(0, -1, 0)) +
                            zxScale * jx2h * (En(1, 0, 1) - En(1, 0, -1) + En(0, 0, 1) - En(0, 0, -1));
                    ld = ld - xxScale * jx0h;
                }
                if (!nyl) {
                    JDloY = -xyScale * jy0l * (En(1, 0, 0) - En(-1, 0, 0) + En(1, -1, 0) - En(-1, -1, 0)) +
                            yyScale * jy1l * Pn(0, -1, 0) -
                            yzScale * jy2l * (En(0, 0, 1) - En(0, 0, -1) + En(0, -1, 1) - En(0, -1, -1));
                    ld = ld - yyScale * jy1l;
                }
                if (!nyh) {
                    JDhiY = +xyScale * jy0h * (En(1, 1, 0) - En(-1, 1, 0) + En(1, 0, 0) - En(-1, 0, 0)) +
                            yyScale * jy1h * Pn(0, 1, 0) +
                            yzScale * jy2h * (En(0, 1, 1) - En(0, 1, -1) + En(0, 0, 1) - En(0, 0, -1));
                    ld = ld - yyScale * jy1h;
                }
                if (!nzl) {
                    JDloZ = -zxScale * jz0l * (En(1, 0, 0) - En(-1, 0, 0) + En(1, 0, -1) - En(-1, 0, -1)) -
                            yzScale * jz1l * (En(0, 1, 0) - En(0, -1, 0) + En(0, 1, -1) - En(0, -1, -1)) +
                            zzScale * jz2l * Pn(0, 0, -1);
                    ld = ld - zzScale * jz2l;
                }
                if (!nzh) {
                    JDhiZ = +zxScale * jz0h * (En(1, 0, 1) - En(-1, 0, 1) + En(1, 0, 0) - En(-1, 0, 0)) +
                            yzScale * jz1h * (En(0, 1, 1) - En(0, -1, 1) + En(0, 1, 0) - En(0, -1, 0)) +
                            zzScale * jz2h * Pn(0, 0, 1);
                    ld = ld - zzScale * jz2h;
                }
                ld = ld * ji;
                const double lphi = P.beta * ji * (JDloX + JDhiX + JDloY + JDhiY + JDloZ + JDhiZ);
                r = (rh - lphi) / (P.alpha + P.beta * ld);
            }
#undef Pn
#undef En
            if (s) U.y = r; else U.x = r;
        }
        // phi after the red pass, for the black cells of planes k-1, k, k+1
        *reinterpret_cast<double2*>(&SU[sc][row][ri]) = U;
        __syncthreads();

        // ---- black(k-1): the pair's other cell, from the values after the red pass; only cells three layers inside the box ----
        {
            const int kb = k - 1;
            if (anyo && (kb >= t.k0) && (kb < t.k0 + t.nk)) {
                const int s = csel;   // red in plane k = black in plane k-1
                double2 O = Uprev;
                const int lay = min(lay_ij[s], min(kb, p.n[2] - 1 - kb));
                if (o[s] && lay >= 3) {
                    const int rc = ri + s;
                    const int um = (kb - 1) & (FF_S - 1), uc = kb & (FF_S - 1), up = (kb + 1) & (FF_S - 1);
#define Un(di, dj, dk) SU[(dk) < 0 ? um : ((dk) > 0 ? up : uc)][row + (dj)][rc + (di)]
                    // GSRBITER3D: three layers inside a box no cell touches a domain face
                    const double pdx = Un(1, 0, 0) - Un(-1, 0, 0);
                    const double pdy = Un(0, 1, 0) - Un(0, -1, 0);
                    const double pdz = Un(0, 0, 1) - Un(0, 0, -1);
                    const double JDxx = b_jx0h * Un(1, 0, 0) + b_jx0l * Un(-1, 0, 0);
                    const double JDxy = b_jx1h * (Un(1, 1, 0) - Un(1, -1, 0) + pdy) - b_jx1l * (pdy + Un(-1, 1, 0) - Un(-1, -1, 0));
                    const double JDxz = b_jx2h * (Un(1, 0, 1) - Un(1, 0, -1) + pdz) - b_jx2l * (pdz + Un(-1, 0, 1) - Un(-1, 0, -1));
                    const double JDyx = b_jy0h * (Un(1, 1, 0) - Un(-1, 1, 0) + pdx) - b_jy0l * (pdx + Un(1, -1, 0) - Un(-1, -1, 0));
                    const double JDyy = b_jy1h * Un(0, 1, 0) + b_jy1l * Un(0, -1, 0);
                    const double JDyz = b_jy2h * (Un(0, 1, 1) - Un(0, 1, -1) + pdz) - b_jy2l * (pdz + Un(0, -1, 1) - Un(0, -1, -1));
                    const double JDzx = b_jz0h * (Un(1, 0, 1) - Un(-1, 0, 1) + pdx) - b_jz0l * (pdx + Un(1, 0, -1) - Un(-1, 0, -1));
                    const double JDzy = b_jz1h * (Un(0, 1, 1) - Un(0, -1, 1) + pdy) - b_jz1l * (pdy + Un(0, 1, -1) - Un(0, -1, -1));
                    const double JDzz = b_jz2h * Un(0, 0, 1) + b_jz2l * Un(0, 0, -1);
                    const double lphi = P.beta * b_ji *
                                        (JDxx * xxScale + JDyy * yyScale + JDzz * zzScale + (JDxy + JDyx) * xyScale +
                                         (JDyz + JDzy) * yzScale + (JDzx + JDxz) * zxScale);
                    const double lapd = -b_ji * ((b_jx0h + b_jx0l) * xxScale + (b_jy1h + b_jy1l) * yyScale + (b_jz2h + b_jz2l) * zzScale);
                    const double r = (b_rh - lphi) / (P.alpha + P.beta * lapd);
#undef Un
                    if (s) O.y = r; else O.x = r;
                }
                double* dst = out + base + sk * kb;
                if (o[0] && o[1]) *reinterpret_cast<double2*>(dst) = O;
                else if (o[0]) dst[0] = O.x;
                else dst[1] = O.y;
            }
        }

        // ---- rotate: the black cell of plane k is the pair's other cell ----
        {
            const int s = csel ^ 1;
            FF_COEF(s, jx0l, jx0h, jx1l, jx1h, jx2l, jx2h, jy0l, jy0h, jy1l, jy1h, jy2l, jy2h, jz0l, jz0h, jz1l, jz1h, jz2l, jz2h)
            b_jx0l = jx0l; b_jx0h = jx0h; b_jx1l = jx1l; b_jx1h = jx1h; b_jx2l = jx2l; b_jx2h = jx2h;
            b_jy0l = jy0l; b_jy0h = jy0h; b_jy1l = jy1l; b_jy1h = jy1h; b_jy2l = jy2l; b_jy2h = jy2h;
            b_jz0l = jz0l; b_jz0h = jz0h; b_jz1l = jz1l; b_jz1h = jz1h; b_jz2l = jz2l; b_jz2h = jz2h;
            b_ji = ff_pick(Ji, s);
            b_rh = ff_pick(Rh, s);
        }
#undef FF_COEF
        Uprev = U;
        Pc = vp;
        Jz0c = Jz0p;
        Jz1c = Jz1p;
        Jz2c = Jz2p;
    }
}

void launch_full_fused(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out, const double* phi,
                       const double* psi, const double* rhs)
{
    if (ntiles == 0) return;
    JgFullF J;
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) J.c[d][c] = L.jgf[d][c];
    if (L.P.zero_xy)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_full_fused<true>), dim3(ntiles), dim3(64, FF_J, 1), 0, st, tiles, L.patches, out, phi,
                           psi, rhs, J, L.jinv, L.P);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_full_fused<false>), dim3(ntiles), dim3(64, FF_J, 1), 0, st, tiles, L.patches, out, phi,
                           psi, rhs, J, L.jinv, L.P);
}

}  // namespace somar
