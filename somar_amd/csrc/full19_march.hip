// somar_amd/csrc/full19_march.hip -- the NON-diagonal (19-point) operator / residual and GSRB colour pass of a large
// level as k-marching, LDS-staged kernels (the non-diagonal counterpart of resid_march.hip).
//
// Why: k_op_full / k_gsrb_full (full19.hip) read phi and its extrapolated copy psi through L2 with 19-point reach and
// nine coefficient planes per cell; the k+-1 reuse distance does not fit the 4 MiB per-XCD L2, every application is
// preceded by a whole-field copy psi := phi (16 B/cell), and lapDiag is a tenth coefficient stream.  Here
//   * a workgroup owns a (124 x 6) column of cells and marches in k; per plane it stages phi AND the cross-term
//     operand E in LDS (4 rotating plane slots each, 128 x 8 doubles per plane, one wavefront per row, one double2 per
//     lane, ONE barrier per plane; 512 threads: the nine coefficient streams need > 128 VGPRs per lane); all 18 neighbours come from LDS, the nine J g^{ab} planes stream through registers
//     exactly once per launch (k-face components are carried from plane to plane);
//   * psi is never copied: E = phi inside the box, psi only in the one-cell FRAME around it -- the only place where
//     the reference's `extrap` FAB differs from phi (extrapolated out-of-domain ghosts, and the box-edge cells
//     ExtrapolateFaceAndCopy rewrites; see solver_full.cpp: the frame-only ghost programs);
//   * lapDiag is recomputed from the diagonal components with FILLMAPPEDLAPDIAG3D's own expression (same bits);
//   * the colour pass writes a second array (updated cells of the colour + untouched cells of the other): the
//     same-colour diagonal neighbours a cross term reads must be PRE-pass values (the reference reads them from the
//     snapshot `extrap`), which an in-place update cannot give a neighbouring workgroup.
// Traffic per cell: residual phi 8 (x halo) + rhs 8 + Jg 72 + Jinv 8 + out 8 = ~106 B (algorithmic 112); colour pass the
// same without rhs -> with it: ~106 B per pass, 212 per red+black sweep (algorithmic 120; the two-pass form reads every
// coefficient line once per colour).
//
// Arithmetic: the expression order of k_op_full / k_gsrb_full (= MAPPEDGETFLUX + zero Neumann flux + flux *= beta +
// MAPPEDFLUXDIVERGENCE3D + AXBYIP + SUBTRACTOP; GSRBITER3D / GSRBBOUNDARYITER3D) => bit-identical to them and to the
// CPU oracle.  Reference: calculus/AMRElliptic/MappedAMRPoissonOpF.ChF:335-427, RelaxationMethods/GSRBF.ChF:36-282,
// 1024-1253.
#include "common.h"
#include "kernels.h"

namespace somar {

// region rows FM_J = tile rows + 1 low + 1 high, a template parameter: 8 (6-row tile, 512 threads, 64 KB LDS; at 160 VGPRs
// a SIMD holds 3 waves, so ONE such workgroup fits a CU: 8 waves per CU) or 6 (4-row tile, 384 threads, 48 KB LDS: TWO
// workgroups per CU = 12 waves, the occupancy limit, for 1.5 instead of 1.33 times the phi / psi halo traffic -- 1.4 B/cell
// of about 106).  SOMAR_FULL_ROWS selects; the tile tables follow (Level::define).
constexpr int FM_S = 4;    // plane slots: k-1, k, k+1 are read while k+2's slot is being written

struct JgFullM { const double* c[3][3]; };  // c[faceDir][component]

__device__ __forceinline__ double2 fm_ld2(const double* __restrict__ a, long long idx, bool ok0, bool ok1, long long safe)
{
    // branch-free predicated pair load (see resid_march.hip)
#ifdef SOMAR_NT_LOADS
    // streamed once per launch: keep the coefficient / right-hand-side lines out of the way of the phi halo reuse in L2
    typedef double v2d_ __attribute__((ext_vector_type(2)));
    const v2d_ w = __builtin_nontemporal_load(reinterpret_cast<const v2d_*>(a + ((ok0 || ok1) ? idx : safe)));
    const double2 v = make_double2(w.x, w.y);
#else
    const double2 v = *reinterpret_cast<const double2*>(a + ((ok0 || ok1) ? idx : safe));
#endif
    return make_double2(ok0 ? v.x : 0.0, ok1 ? v.y : 0.0);
}
__device__ __forceinline__ double fm_pick(const double2& v, int s) { return s ? v.y : v.x; }

// MODE 0: out = rhs - L[phi]   MODE 1: out = L[phi]   MODE 2: out = phi with the cells of `color` relaxed (one GSRB pass)
// MODE 3: the colour pass on the cells of the three outer layers of each box only, in place, after k_full_fused (see there)
// ZXY: J g^{xy} on x-faces and J g^{yx} on y-faces are identically zero (StencilParams::zero_xy): those two planes are not
// streamed -- three of the pass's thirteen coefficient loads per plane -- and zeros stand in for them; the products are the
// zeros the stored planes would give (up to the sign of zero).
// CLS: the tile's lane class (Tile::pad_[1], Level::define): a wavefront covers 2^CLS region rows of 128 >> CLS columns each.
// CLS 0 = one 128-column row per wavefront (124 output columns); CLS 1 = two 64-column rows (60 output columns: a 64-wide
// box is one such tile + a 4-wide one instead of a tile that leaves 30 of 64 lanes idle); CLS 4 = sixteen 8-column rows (the
// 4-column remainder of a box, 128 -> 124 + 4).  Region rows of a wavefront have the same parity (row = 2^(CLS+1) (y / 2) +
// (y & 1) + 2 sub), which the 7-point fused sweep needs and the others share.  Same arithmetic on the same values: same bits.
template <int MODE, int FM_J, bool ZXY, int CLS>
__device__ __forceinline__ void full_march_body(double* __restrict__ SP, double* __restrict__ SE, const Tile& t,
                                                const PatchDesc& p, double* __restrict__ out,
                                                const double* __restrict__ phi, const double* __restrict__ psi,
                                                const double* __restrict__ rhs, const JgFullM& J,
                                                const double* __restrict__ jinv, const StencilParams& P, int color,
                                                const double* __restrict__ phi2)
{
    constexpr int LPR = 64 >> CLS;                 // lanes per region row
    constexpr int NR = FM_J << CLS;                // region rows of the workgroup
    constexpr int PITCH = 2 * LPR + (CLS >= 2 ? 2 : 0);   // LDS row pitch; narrow rows padded against bank conflicts
#define SPx(slot, r, c) SP[((slot) * NR + (r)) * PITCH + (c)]
#define SEx(slot, r, c) SE[((slot) * NR + (r)) * PITCH + (c)]
    // class 0: the region row is the wavefront's index, a scalar -- everything derived from it stays in scalar registers
    const int lane = CLS == 0 ? (int)threadIdx.x : (int)(threadIdx.x & (LPR - 1));
    const int row = CLS == 0 ? (int)threadIdx.y
                             : (int)(((threadIdx.y >> 1) << (CLS + 1)) + (threadIdx.y & 1) + 2 * (threadIdx.x >> (6 - CLS)));
    const int ri = 2 * lane;
    const int li = t.i0 - 2 + ri;  // even: rows are 16-byte aligned
    const int wi = t.pad_[0] > 0 ? t.pad_[0] : 2 * LPR - 4;
    const int lj = t.j0 - 1 + row;
    const int gj = p.lo[1] + lj;

    // phi / psi may be touched inside the 1-cell ghost layer; coefficients only at the tile's own cells / faces
    const bool fj = (lj >= -1) && (lj <= p.n[1]);
    const bool f0 = fj && (li >= -1) && (li <= p.n[0]) && (ri < wi + 4);
    const bool f1 = fj && (li + 1 >= -1) && (li + 1 <= p.n[0]) && (ri + 1 < wi + 4);
    const bool inj = (lj >= 0) && (lj < p.n[1]);
    const bool in0 = inj && (li >= 0) && (li < p.n[0]);          // inside the box (in i, j): E = phi there
    const bool in1 = inj && (li + 1 >= 0) && (li + 1 < p.n[0]);
    const bool own_j = inj && (row >= 1) && (row <= NR - 2) && (lj < t.j0 + (NR - 2));
    bool o[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int l = li + s, r = ri + s;
        o[s] = own_j && (l >= 0) && (l < p.n[0]) && (r >= 2) && (r < wi + 2);
    }
    const int left_o1 = __shfl_up((int)o[1], 1, 64);   // all lanes active here
    const bool gxo0 = o[0] || (left_o1 != 0);           // my first x-face is the high face of the left lane's second cell
    const bool any = o[0] || o[1];
    const long long sj = p.pj, sk = p.pk;
    const long long base = p.off + li + sj * lj;

    const double dxi0 = 1.0 / P.dx[0], dxi1 = 1.0 / P.dx[1], dxi2 = 1.0 / P.dx[2];
    // MAPPEDGETFLUX's scales with beta = a_ref = 1: aScale = 1.0 * dxi[a], b/cScale = 0.25 * 1.0 * dxi[b/c]
    const double a0 = 1.0 * dxi0, a1 = 1.0 * dxi1, a2 = 1.0 * dxi2;
    const double q0 = 0.25 * 1.0 * dxi0, q1 = 0.25 * 1.0 * dxi1, q2 = 0.25 * 1.0 * dxi2;
    const double xxScale = 1.0 / (P.dx[0] * P.dx[0]);
    const double yyScale = 1.0 / (P.dx[1] * P.dx[1]);
    const double zzScale = 1.0 / (P.dx[2] * P.dx[2]);
    const double xyScale = 0.25 / (P.dx[0] * P.dx[1]);
    const double yzScale = 0.25 / (P.dx[1] * P.dx[2]);
    const double zxScale = 0.25 / (P.dx[2] * P.dx[0]);

    // stage plane kp of phi and E into its slot (global loads first, LDS writes by the caller's schedule)
    auto load_plane = [&](int kp, double2& vp, double2& ve) {
        const bool fk = (kp >= -1) && (kp <= p.n[2]);
        const bool ink = (kp >= 0) && (kp < p.n[2]);
        const long long idx = base + sk * kp;
        vp = fm_ld2(phi, idx, f0 && fk, f1 && fk, p.off);
        if (MODE == 3) {
            // the pass's colour INSIDE the box comes from phi2 (the sweep's input: the pre-pass value of a cell the fused kernel
            // may already have relaxed), everything else -- the other colour, every ghost cell -- from phi
            const int cs = (p.lo[0] + li + gj + p.lo[2] + kp + color) & 1;
            const bool m0 = in0 && ink && cs == 0, m1 = in1 && ink && cs == 1;
            const double2 vq = fm_ld2(phi2, idx, m0, m1, p.off);
            vp = make_double2(m0 ? vq.x : vp.x, m1 ? vq.y : vp.y);
        }
        const bool e0 = f0 && fk && !(in0 && ink), e1 = f1 && fk && !(in1 && ink);   // frame cells: E = psi
        const double2 vs = fm_ld2(psi, idx, e0, e1, p.off);
        ve = make_double2(e0 ? vs.x : vp.x, e1 ? vs.y : vp.y);
    };
    auto store_plane = [&](int kp, const double2& vp, const double2& ve) {
        const int slot = kp & (FM_S - 1);
        *reinterpret_cast<double2*>(&SPx(slot, row, ri)) = vp;
        *reinterpret_cast<double2*>(&SEx(slot, row, ri)) = ve;
    };

    int k = t.k0;
    const int kend = t.k0 + t.nk;
    {
        double2 vp, ve;
        load_plane(k - 1, vp, ve);
        store_plane(k - 1, vp, ve);
        load_plane(k, vp, ve);
        store_plane(k, vp, ve);
    }
    // k-face components on the LOW face of plane k
    double2 Jz0c = fm_ld2(J.c[2][0], base + sk * k, o[0], o[1], p.off);
    double2 Jz1c = fm_ld2(J.c[2][1], base + sk * k, o[0], o[1], p.off);
    double2 Jz2c = fm_ld2(J.c[2][2], base + sk * k, o[0], o[1], p.off);

    for (; k < kend; ++k) {
        const int gk = p.lo[2] + k;
        // ---- this step's loads: phi / psi of plane k+1, coefficients of plane k ----
        double2 vp, ve;
        load_plane(k + 1, vp, ve);
        const long long ck = base + sk * k;
        const double2 Jz0p = fm_ld2(J.c[2][0], ck + sk, o[0], o[1], p.off);
        const double2 Jz1p = fm_ld2(J.c[2][1], ck + sk, o[0], o[1], p.off);
        const double2 Jz2p = fm_ld2(J.c[2][2], ck + sk, o[0], o[1], p.off);
        double2 Rh = make_double2(0.0, 0.0);
        if (MODE != 1) Rh = fm_ld2(rhs, ck, o[0], o[1], p.off);
        const double2 Ji = fm_ld2(jinv, ck, o[0], o[1], p.off);
        const double2 Jx0 = fm_ld2(J.c[0][0], ck, gxo0, any, p.off);
        const double2 Jx1 = ZXY ? make_double2(0.0, 0.0) : fm_ld2(J.c[0][1], ck, gxo0, any, p.off);
        const double2 Jx2 = fm_ld2(J.c[0][2], ck, gxo0, any, p.off);
        const double2 Jy0 = ZXY ? make_double2(0.0, 0.0) : fm_ld2(J.c[1][0], ck, o[0], o[1], p.off);
        const double2 Jy1 = fm_ld2(J.c[1][1], ck, o[0], o[1], p.off);
        const double2 Jy2 = fm_ld2(J.c[1][2], ck, o[0], o[1], p.off);
        const double2 Jy0h = ZXY ? make_double2(0.0, 0.0) : fm_ld2(J.c[1][0], ck + sj, o[0], o[1], p.off);
        const double2 Jy1h = fm_ld2(J.c[1][1], ck + sj, o[0], o[1], p.off);
        const double2 Jy2h = fm_ld2(J.c[1][2], ck + sj, o[0], o[1], p.off);
        const double jx0n = __shfl_down(Jx0.x, 1, 64), jx1n = __shfl_down(Jx1.x, 1, 64), jx2n = __shfl_down(Jx2.x, 1, 64);

        store_plane(k + 1, vp, ve);
        __syncthreads();

        if (any) {
            const int sm = (k - 1) & (FM_S - 1), sc = k & (FM_S - 1), sp = (k + 1) & (FM_S - 1);
            double res[2] = {0.0, 0.0};
            bool o3 = true;
            const int csel = (p.lo[0] + li + gj + gk + color) & 1;   // MODE 2: the cell of the pair that has this colour
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int rc = ri + s;
                const double pc = SPx(sc, row, rc);
                if (MODE >= 2 && s != csel) { res[s] = pc; continue; }
                if (!o[s]) continue;
                if (MODE == 3 && min(min(min(li + s, p.n[0] - 1 - (li + s)), min(lj, p.n[1] - 1 - lj)), min(k, p.n[2] - 1 - k)) >= 3) {
                    o3 = false;   // three layers inside the box: k_full_fused relaxed this cell
                    continue;
                }
                const int gi = p.lo[0] + li + s;
                // neighbours: Pn(di,dj,dk) = phi, En(di,dj,dk) = E
#define Pn(di, dj, dk) SPx((dk) < 0 ? sm : ((dk) > 0 ? sp : sc), row + (dj), rc + (di))
#define En(di, dj, dk) SEx((dk) < 0 ? sm : ((dk) > 0 ? sp : sc), row + (dj), rc + (di))
                // coefficients on the low / high face of this cell
                const double jx0l = s ? Jx0.y : Jx0.x, jx0h = s ? jx0n : Jx0.y;
                const double jx1l = s ? Jx1.y : Jx1.x, jx1h = s ? jx1n : Jx1.y;
                const double jx2l = s ? Jx2.y : Jx2.x, jx2h = s ? jx2n : Jx2.y;
                const double jy0l = fm_pick(Jy0, s), jy0h = fm_pick(Jy0h, s);
                const double jy1l = fm_pick(Jy1, s), jy1h = fm_pick(Jy1h, s);
                const double jy2l = fm_pick(Jy2, s), jy2h = fm_pick(Jy2h, s);
                const double jz0l = fm_pick(Jz0c, s), jz0h = fm_pick(Jz0p, s);
                const double jz1l = fm_pick(Jz1c, s), jz1h = fm_pick(Jz1p, s);
                const double jz2l = fm_pick(Jz2c, s), jz2h = fm_pick(Jz2p, s);
                const double ji = fm_pick(Ji, s);
                if (MODE < 2) {
                    // flux19 (full19.hip) at the six faces: direction a, then b = a+1, c = a+2 (cyclic)
                    double fxl = a0 * jx0l * (pc - Pn(-1, 0, 0)) +
                                 q1 * jx1l * (En(0, 1, 0) - En(0, -1, 0) + En(-1, 1, 0) - En(-1, -1, 0)) +
                                 q2 * jx2l * (En(0, 0, 1) - En(0, 0, -1) + En(-1, 0, 1) - En(-1, 0, -1));
                    double fxh = a0 * jx0h * (Pn(1, 0, 0) - pc) +
                                 q1 * jx1h * (En(1, 1, 0) - En(1, -1, 0) + En(0, 1, 0) - En(0, -1, 0)) +
                                 q2 * jx2h * (En(1, 0, 1) - En(1, 0, -1) + En(0, 0, 1) - En(0, 0, -1));
                    double fyl = a1 * jy1l * (pc - Pn(0, -1, 0)) +
                                 q2 * jy2l * (En(0, 0, 1) - En(0, 0, -1) + En(0, -1, 1) - En(0, -1, -1)) +
                                 q0 * jy0l * (En(1, 0, 0) - En(-1, 0, 0) + En(1, -1, 0) - En(-1, -1, 0));
                    double fyh = a1 * jy1h * (Pn(0, 1, 0) - pc) +
                                 q2 * jy2h * (En(0, 1, 1) - En(0, 1, -1) + En(0, 0, 1) - En(0, 0, -1)) +
                                 q0 * jy0h * (En(1, 1, 0) - En(-1, 1, 0) + En(1, 0, 0) - En(-1, 0, 0));
                    double fzl = a2 * jz2l * (pc - Pn(0, 0, -1)) +
                                 q0 * jz0l * (En(1, 0, 0) - En(-1, 0, 0) + En(1, 0, -1) - En(-1, 0, -1)) +
                                 q1 * jz1l * (En(0, 1, 0) - En(0, -1, 0) + En(0, 1, -1) - En(0, -1, -1));
                    double fzh = a2 * jz2h * (Pn(0, 0, 1) - pc) +
                                 q0 * jz0h * (En(1, 0, 1) - En(-1, 0, 1) + En(1, 0, 0) - En(-1, 0, 0)) +
                                 q1 * jz1h * (En(0, 1, 1) - En(0, -1, 1) + En(0, 1, 0) - En(0, -1, 0));
                    // EllipticConstNeumBCFluxClass: boundary faces := 0 (homogeneous)
                    if (gi == P.dom_lo[0] && P.neum[0][0]) fxl = 0.0;
                    if (gi == P.dom_hi[0] && P.neum[0][1]) fxh = 0.0;
                    if (gj == P.dom_lo[1] && P.neum[1][0]) fyl = 0.0;
                    if (gj == P.dom_hi[1] && P.neum[1][1]) fyh = 0.0;
                    if (gk == P.dom_lo[2] && P.neum[2][0]) fzl = 0.0;
                    if (gk == P.dom_hi[2] && P.neum[2][1]) fzh = 0.0;
                    fxl *= P.beta; fxh *= P.beta; fyl *= P.beta; fyh *= P.beta; fzl *= P.beta; fzh *= P.beta;
                    double l = ji * ((fxh - fxl) * dxi0 + (fyh - fyl) * dxi1 + (fzh - fzl) * dxi2);
                    if (P.alpha != 0.0) l = P.alpha * pc + 1.0 * l;
                    res[s] = (MODE == 0) ? (fm_pick(Rh, s) - l) : l;
                } else {
                    const bool onb = (gi == P.dom_lo[0]) || (gi == P.dom_hi[0]) || (gj == P.dom_lo[1]) ||
                                     (gj == P.dom_hi[1]) || (gk == P.dom_lo[2]) || (gk == P.dom_hi[2]);
                    const double rh = fm_pick(Rh, s);
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
                        // lapDiag: FILLMAPPEDLAPDIAG3D's expression (k_lapdiag), bitwise the stored array's value
                        const double lapd = -ji * ((jx0h + jx0l) * xxScale + (jy1h + jy1l) * yyScale + (jz2h + jz2l) * zzScale);
                        res[s] = (rh - lphi) / (P.alpha + P.beta * lapd);
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
                        res[s] = (rh - lphi) / (P.alpha + P.beta * ld);
                    }
                }
#undef Pn
#undef En
            }
            double* dst = out + base + sk * k;
            if (MODE == 3) {
                if (o[csel] && o3) dst[csel] = res[csel];   // in place: only the shell cells of the colour
            } else if (o[0] && o[1]) *reinterpret_cast<double2*>(dst) = make_double2(res[0], res[1]);
            else if (o[0]) dst[0] = res[0];
            else dst[1] = res[1];
        }
        Jz0c = Jz0p;
        Jz1c = Jz1p;
        Jz2c = Jz2p;
    }
#undef SPx
#undef SEx
}

// NARROW: the tile table holds narrow lane classes; the instantiation without them is the one-body kernel with 64 KB of LDS
// (one 512-wide box: colour pass 3.02 ms against 3.24 with the three-body kernel on the same class-0 tiles).
template <int MODE, int FM_J, bool ZXY, bool NARROW>
// (an occupancy hint of 3 waves per SIMD on every instantiation changed the scheduling of those that need none: 512^3 colour pass
// 3.02 -> 3.29 ms, C5 53.1 -> 55.8 ms; only the three-body operator without zero planes would otherwise take 169 VGPRs)
__global__ __launch_bounds__(64 * FM_J, (NARROW && !ZXY && MODE == 1) ? 3 : 1) void k_full_march(const Tile* __restrict__ tiles,
                                                             const PatchDesc* __restrict__ patches,
                                                             double* __restrict__ out, const double* __restrict__ phi,
                                                             const double* __restrict__ psi,
                                                             const double* __restrict__ rhs, JgFullM J,
                                                             const double* __restrict__ jinv, StencilParams P, int color,
                                                          const double* __restrict__ phi2)
{
    // one slot = the largest class's region: FM_J rows of 128, or 16 FM_J rows of 8 + 2 columns
    __shared__ __attribute__((aligned(16))) double SP[FM_S * FM_J * (NARROW ? 160 : 128)];  // phi
    __shared__ __attribute__((aligned(16))) double SE[FM_S * FM_J * (NARROW ? 160 : 128)];  // E: phi inside the box, psi in its frame
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int cls = NARROW ? t.pad_[1] : 0;
    if (!NARROW || cls == 0) full_march_body<MODE, FM_J, ZXY, 0>(SP, SE, t, p, out, phi, psi, rhs, J, jinv, P, color, phi2);
    else if (cls == 1) full_march_body<MODE, FM_J, ZXY, NARROW ? 1 : 0>(SP, SE, t, p, out, phi, psi, rhs, J, jinv, P, color, phi2);
    else full_march_body<MODE, FM_J, ZXY, NARROW ? 4 : 0>(SP, SE, t, p, out, phi, psi, rhs, J, jinv, P, color, phi2);
}

int full_march_rows()
{
    static int rows = 0;
    if (!rows) {
        const char* e = getenv("SOMAR_FULL_ROWS");
        rows = (e && atoi(e) == 6) ? 6 : 8;
    }
    return rows;
}

static JgFullM jgfullm(const LevelDev& L)
{
    JgFullM J;
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) J.c[d][c] = L.jgf[d][c];
    return J;
}

// mode 0: out = rhs - L[phi], 1: out = L[phi], 2: one GSRB colour pass.  psi needs to be right only in the one-cell frame of every box.
template <int MODE>
static void launch_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out, const double* phi,
                         const double* psi, const double* rhs, int color, const double* phi2 = nullptr, bool narrow = false)
{
    const bool six = full_march_rows() == 6, z = L.P.zero_xy != 0, nw = L.narrowq != 0 || narrow;   // (6-row tables never hold narrow classes)
#define SOMAR_FM(ROWS, Z, N)                                                                                                   \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_full_march<MODE, ROWS, Z, N>), dim3(ntiles), dim3(64, ROWS, 1), 0, st, tiles, L.patches, \
                       out, phi, psi, rhs, jgfullm(L), L.jinv, L.P, color, phi2)
    if (six) { if (z) SOMAR_FM(6, true, false); else SOMAR_FM(6, false, false); }
    else if (nw) { if (z) SOMAR_FM(8, true, true); else SOMAR_FM(8, false, true); }
    else { if (z) SOMAR_FM(8, true, false); else SOMAR_FM(8, false, false); }
#undef SOMAR_FM
}

void launch_full_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out, const double* phi,
                       const double* psi, const double* rhs, int mode)
{
    if (ntiles == 0) return;
    if (mode == 0) launch_march<0>(st, tiles, ntiles, L, out, phi, psi, rhs, 0);
    else launch_march<1>(st, tiles, ntiles, L, out, phi, psi, rhs, 0);
}

// one colour pass of the 19-point GSRB: out = phi with the cells of `color` relaxed (out != phi)
void launch_gsrb_full_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out,
                            const double* phi, const double* psi, const double* rhs, int color)
{
    if (ntiles == 0) return;
    launch_march<2>(st, tiles, ntiles, L, out, phi, psi, rhs, color);
}

}  // namespace somar

namespace somar {
// the black cells of the three outer layers of every box after k_full_fused (full19_fused.hip): in place on phi, their same-colour
// neighbours from phi_in (the sweep's input); tiles = Level::d_stiles (box faces only, narrow classes on the x faces)
void launch_gsrb_full_shell(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* phi, const double* phi_in,
                            const double* psi, const double* rhs)
{
    if (ntiles == 0) return;
    SOMAR_CHECK(full_march_rows() == 8, "internal: the shell pass uses the 8-row marching kernels");
    launch_march<3>(st, tiles, ntiles, L, phi, phi, psi, rhs, 1, phi_in, true);
}
}  // namespace somar
