// somar_amd/csrc/gsrb_fused.hip -- one launch = one full red+black GSRB sweep (diagonal metric).
//
// Why: the two-pass form of LevelGSRB (GSRB.cpp:58-98) streams every coefficient line twice per
// sweep (each colour touches half of every 128-B line) and, measured with rocprofv3 PMC on
// MI355X, re-fetches phi 3x and Jg^zz 2x per pass because the k+-1 reuse distance (a whole
// j-slab of every resident workgroup) does not fit the 4 MiB per-XCD L2: 78 B/cell/pass against
// 32 B/cell algorithmic.  This kernel reads every array ONCE per sweep:
//
//   * a workgroup owns a (124 x 12) column of cells and MARCHES in k; per plane it stages phi in
//     LDS (3 rotating planes of 128 x 16 doubles, one wavefront per row, one double2 per lane) and
//     keeps the k-neighbours and the k-face coefficients in registers;
//   * red and black are pipelined one plane apart: step k computes red(k) from OLD black, then
//     black(k-1) from the NEW red of planes k-2, k-1, k -- exactly the values LevelGSRB produces,
//     because a red update only reads black cells and vice versa; one barrier per plane;
//   * the 1-cell ring of red cells around the tile is recomputed redundantly (from a 2-deep phi halo
//     and 1-deep coefficient/rhs halos) instead of being exchanged between the colours, so a level
//     needs ONE ghost exchange per sweep instead of two -- on a sharded level that also halves the
//     number of xGMI round trips;
//   * lapDiag is not loaded: it is recomputed from Jg/Jinv with FILLMAPPEDLAPDIAG3D's own expression
//     (MappedAMRPoissonOpF.ChF:266-271), which yields the identical bits and saves 8 B/cell;
//   * results go to a second array (ping-pong): neighbouring workgroups read each other's halo from
//     the untouched input.
//
// Traffic per cell and sweep: phi 8 in + 8 out, rhs 8, Jg 24, Jinv 8 = 56 B + halo overhead
// (phi x 128*16/(124*12), ring coefficients) ~ 63 B, versus 128-156 B for the two-pass form.
//
// Arithmetic: identical operation order to GSRBITER3DORTHO / GSRBBOUNDARYITER3DORTHO
// (GSRBF.ChF:545-701, 1362-1505) => bit-identical to the two-pass kernel and to the CPU oracle.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace somar {


__device__ __forceinline__ double pick(const double2& v, int s) { return s ? v.y : v.x; }

// load a[idx], a[idx+1] with per-element predicates (idx is 16-byte aligned by construction)
__device__ __forceinline__ double2 ld2(const double* __restrict__ a, long long idx, bool ok0, bool ok1,
                                          long long safe)
{
    // branch-free: always one aligned 16-byte load (from `safe`, any valid aligned element of the patch, when
    // neither element may be touched), then selects.  Straight-line loads let the compiler count outstanding
    // loads exactly (s_waitcnt vmcnt(N)) instead of draining everything at every predicated branch.
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

// the "load" of a uniform coefficient.  ld2 returns zeros where its predicate is off; those values only ever reach cells that
// are not computed (a computed cell's coefficients lie inside the frame by construction), so the constant may stand in
// unconditionally -- which makes every coefficient a wave-uniform loop invariant: the diagonal and the selects between the
// pair's two members fold away.
__device__ __forceinline__ double2 uni2(double c, bool, bool) { return make_double2(c, c); }

// One GSRB point update of cell (gi,gj,gk): interior form = GSRBITER3DORTHO, cells touching a domain
// face = GSRBBOUNDARYITER3DORTHO (a Neumann face contributes neither flux nor diagonal; a Dirichlet face contributes
// both, its ghost being -own).  own = the cell's value before this update.
// INT: the caller's tile (its red ring included) touches no domain face and no seam in x and y, so only the plane's position
// in z -- the same for every lane -- decides between the two forms: the per-lane classification (periodic wrap, six
// compares, a divergent branch) collapses into one scalar test per plane.
template <bool DIRI, bool INT = false>
__device__ __forceinline__ double gsrb_point(const StencilParams& P, double xxS, double yyS, double zzS, int gi,
                                             int gj, int gk, double pxl, double pxh, double pyl, double pyh,
                                             double pzl, double pzh, double gxl, double gxh, double gyl, double gyh,
                                             double gzl, double gzh, double Ji, double rhs, double own)
{
    // a ring cell may be the periodic image of a cell on the far side: classify the REAL cell
    if (!INT && P.periodic[0]) { const int n = P.dom_hi[0] - P.dom_lo[0] + 1; gi = gi < P.dom_lo[0] ? gi + n : (gi > P.dom_hi[0] ? gi - n : gi); }
    if (!INT && P.periodic[1]) { const int n = P.dom_hi[1] - P.dom_lo[1] + 1; gj = gj < P.dom_lo[1] ? gj + n : (gj > P.dom_hi[1] ? gj - n : gj); }
    if (P.periodic[2]) { const int n = P.dom_hi[2] - P.dom_lo[2] + 1; gk = gk < P.dom_lo[2] ? gk + n : (gk > P.dom_hi[2] ? gk - n : gk); }
    const bool onb = (!INT && ((gi == P.dom_lo[0]) || (gi == P.dom_hi[0]) || (gj == P.dom_lo[1]) || (gj == P.dom_hi[1]))) ||
                     (gk == P.dom_lo[2]) || (gk == P.dom_hi[2]);
    if (!onb) {
        const double JDxx = xxS * (gxh * pxh + gxl * pxl);
        const double JDyy = yyS * (gyh * pyh + gyl * pyl);
        const double JDzz = zzS * (gzh * pzh + gzl * pzl);
        const double lphi = P.beta * Ji * (JDxx + JDyy + JDzz);
        // lapDiag: FILLMAPPEDLAPDIAG3D's expression, bitwise the stored array's value
        const double lapd = -Ji * ((gxh + gxl) * xxS + (gyh + gyl) * yyS + (gzh + gzl) * zzS);
        return (rhs - lphi) / (P.alpha + P.beta * lapd);
    }
    const bool nxl = !INT && (gi == P.dom_lo[0]) && P.neum[0][0];
    const bool nxh = !INT && (gi == P.dom_hi[0]) && P.neum[0][1];
    const bool nyl = !INT && (gj == P.dom_lo[1]) && P.neum[1][0];
    const bool nyh = !INT && (gj == P.dom_hi[1]) && P.neum[1][1];
    const bool nzl = (gk == P.dom_lo[2]) && P.neum[2][0];
    const bool nzh = (gk == P.dom_hi[2]) && P.neum[2][1];
    // A Dirichlet face: the neighbour beyond it is the ghost ELLIPTICCONSTDIRIBCGHOST (order 1, homogeneous inside the
    // smoother) derives from this very cell, -phi(cell).  The cell keeps its value through the other colour's pass, so
    // the ghost LevelGSRB refills before each pass is -own in both; it never has to exist in memory.
    // (DIRI is a template parameter: the Neumann / periodic instantiations carry none of this.)
    if (DIRI) {   // (never together with INT)
        if ((gi == P.dom_lo[0]) && P.diri[0][0]) pxl = -own;
        if ((gi == P.dom_hi[0]) && P.diri[0][1]) pxh = -own;
        if ((gj == P.dom_lo[1]) && P.diri[1][0]) pyl = -own;
        if ((gj == P.dom_hi[1]) && P.diri[1][1]) pyh = -own;
        if ((gk == P.dom_lo[2]) && P.diri[2][0]) pzl = -own;
        if ((gk == P.dom_hi[2]) && P.diri[2][1]) pzh = -own;
    }
    double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0, ld = 0.0;
    if (!nxl) { JDloX = gxl * pxl; ld = ld - xxS * gxl; }
    if (!nyl) { JDloY = gyl * pyl; ld = ld - yyS * gyl; }
    if (!nzl) { JDloZ = gzl * pzl; ld = ld - zzS * gzl; }
    if (!nxh) { JDhiX = gxh * pxh; ld = ld - xxS * gxh; }
    if (!nyh) { JDhiY = gyh * pyh; ld = ld - yyS * gyh; }
    if (!nzh) { JDhiZ = gzh * pzh; ld = ld - zzS * gzh; }
    ld = ld * Ji;
    const double lphi = P.beta * Ji * ((JDloX + JDhiX) * xxS + (JDloY + JDhiY) * yyS + (JDloZ + JDhiZ) * zzS);
    return (rhs - lphi) / (P.alpha + P.beta * ld);
}

// blockDim = (64, FR_J): lane = i-pair of the region, threadIdx.y = region row (one wavefront each).
// FR_J = 16: 124 x 12 output columns per 1024-thread workgroup (one per CU);
// FR_J =  8: 124 x 4 per 512-thread workgroup, two resident per CU so one computes while the other waits.
// INMODE 0: plain.  1: phi_in is taken to be zero everywhere and is not read (first sweep on a zero correction).
// 2: every value of phi_in is read as (value - sums[0]/sums[1]): the mean removal of ZeroAvgConstInterpPS
//    (a_phiThisLevel[dit] -= avgPhi over the whole FAB, ProlongationStrategy.cpp:160-163) folded into the first
//    post-smoothing sweep instead of a separate 16 B/cell pass.
// 3: phi_in is read as value + crse(i / r): the prolongation (CONSTINTERPPS) folded into the first post-smoothing
//    sweep -- ghosts included, which needs crse exchanged one cell deep.  4: as 3, minus sums[0]/sums[1].
// INT (uniform-metric instantiations only): this tile and its red ring touch no domain face, periodic seam or coarse-fine face
// in x and y -- over 90 % of the tiles of a large level -- so everything that classifies a cell against those per lane
// is compiled out (gsrb_point<.., INT>, the ring's existence tests, the coarse-fine ghosts of x and y).  Same arithmetic on
// the same values: same bits.
// CLS: the tile's lane class (see full19_march.hip): a wavefront covers 2^CLS region rows of 128 >> CLS columns, all of the
// same parity, so the colour column c stays wave-uniform.
template <int FR_J, int INMODE, bool DIRI, bool UNI, bool INT, int CLS>
__device__ __forceinline__ void gsrb_fused_body(double* __restrict__ S, const Tile& t, const PatchDesc& p,
                                                double* __restrict__ phi_out,
                                                const double* __restrict__ phi_in,
                                                const double* __restrict__ rhs,
                                                const double* __restrict__ jgx,
                                                const double* __restrict__ jgy,
                                                const double* __restrict__ jgz,
                                                const double* __restrict__ jinv, const StencilParams& P,
                                                const double* __restrict__ sums,
                                                const PatchDesc* __restrict__ cpatches,
                                                const double* __restrict__ crse, int r0, int r1, int r2)
{
    const double avg = (INMODE == 2 || INMODE == 4) ? sums[0] / sums[1] : 0.0;
    constexpr int LPR = 64 >> CLS;                 // lanes per region row
    constexpr int NR = FR_J << CLS;                // region rows of the workgroup
    constexpr int PITCH = 2 * LPR + (CLS >= 2 ? 2 : 0);
#define Sx(slot, r, c) S[((slot) * NR + (r)) * PITCH + (c)]
    // class 0: the region row is the wavefront's index, a scalar -- everything derived from it stays in scalar registers
    const int lane = CLS == 0 ? (int)threadIdx.x : (int)(threadIdx.x & (LPR - 1));
    const int row = CLS == 0 ? (int)threadIdx.y
                             : (int)(((threadIdx.y >> 1) << (CLS + 1)) + (threadIdx.y & 1) + 2 * (threadIdx.x >> (6 - CLS)));
    const int ri = 2 * lane;       // region column of the pair's first cell
    const int li = t.i0 - 2 + ri;  // local i of the pair's first cell (even: rows are 16-byte aligned)
    // output columns of this tile (even, <= region width - 4); the region beyond wi + 4 columns is neither loaded nor computed
    const int wi = t.pad_[0] > 0 ? t.pad_[0] : 2 * LPR - 4;
    const int lj = t.j0 - 2 + row;
    // INMODE 3/4: where this pair's coarse parents live (floor division: ghosts map to coarse ghosts)
    long long cbase = 0, cpk = 0, coff = 0;
    int cstep = 0, csh2 = 0;
    if (INMODE >= 3) {
        const PatchDesc cp = cpatches[t.patch];
        auto fdiv = [](int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
        const int ci0 = fdiv(li, r0);
        cstep = fdiv(li + 1, r0) - ci0;  // 0 when both cells share a parent
        coff = cp.off;
        cbase = cp.off + ci0 + (long long)cp.pj * fdiv(lj, r1);
        cpk = cp.pk;
        csh2 = r2 >> 1;                  // multigrid ratios are 1 or 2 per direction (checked by the launcher)
    }
    auto ldphi = [&](int kp, bool ok0, bool ok1) {
        if (INMODE == 1) return make_double2(0.0, 0.0);
        double2 v = ld2(phi_in, p.off + li + (long long)p.pj * lj + p.pk * kp, ok0, ok1, p.off);
        if (INMODE >= 3) {
            // floor(kp / r2) as an arithmetic shift; both coarse loads are unconditional loads from a safe address (no branch,
            // no load that waits for another: a branch here made the compiler drain vmcnt(0) in every plane)
            const bool any = ok0 || ok1;
            const long long c = any ? cbase + cpk * (long long)(kp >> csh2) : coff;
            const long long c2 = any ? c + cstep : coff;
            const double c0 = crse[c];
            const double c1 = crse[c2];
            v.x = ok0 ? v.x + c0 : 0.0;
            v.y = ok1 ? v.y + c1 : 0.0;
        }
        if (INMODE == 2 || INMODE == 4) { v.x = v.x - avg; v.y = v.y - avg; }
        return v;
    };
    const int gj = p.lo[1] + lj;
    const double xxS = 1.0 / (P.dx[0] * P.dx[0]);
    const double yyS = 1.0 / (P.dx[1] * P.dx[1]);
    const double zzS = 1.0 / (P.dx[2] * P.dx[2]);

    // memory predicates: a cell may be touched only inside the patch's 2-cell frame
    const bool fj = (lj >= -FRAME) && (lj < p.n[1] + FRAME);
    const bool fjh = (lj + 1 >= -FRAME) && (lj + 1 < p.n[1] + FRAME);
    const bool f0 = fj && (li >= -FRAME) && (li < p.n[0] + FRAME) && (ri < wi + 4);
    const bool f1 = fj && (li + 1 >= -FRAME) && (li + 1 < p.n[0] + FRAME) && (ri + 1 < wi + 4);
    // the outermost region rows only supply phi to the red ring: they need no coefficients
    const bool cf = (row >= 1) && (row <= NR - 2);
    const bool c0 = f0 && cf, c1 = f1 && cf;
    // red is computed on the tile grown by one cell, restricted to cells that exist: a cell beyond a
    // non-periodic (Neumann) domain face does not; a periodic image or a neighbour box's cell does
    // (its phi was exchanged two deep, its coefficients one deep).  out_* = cells this tile owns.
    bool comp_ij[2], out_ij[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int l = li + s, g = p.lo[0] + l, r = ri + s;
        bool cmp = (l >= -1) && (l <= p.n[0]) && (r >= 1) && (r <= wi + 2) && (lj >= -1) && (lj <= p.n[1]) &&
                   (row >= 1) && (row <= NR - 2);
        if (!INT) {
            if ((g < P.dom_lo[0] && (P.neum[0][0] || (DIRI && P.diri[0][0]))) ||
                (g > P.dom_hi[0] && (P.neum[0][1] || (DIRI && P.diri[0][1]))))
                cmp = false;
            if ((gj < P.dom_lo[1] && (P.neum[1][0] || (DIRI && P.diri[1][0]))) ||
                (gj > P.dom_hi[1] && (P.neum[1][1] || (DIRI && P.diri[1][1]))))
                cmp = false;
            // beyond a coarse-fine face of this box there is no cell of this level either: the ghost there is an
            // interpolated value (filled before the sweep for the red phase, recomputed below for the black one)
            if ((l < 0 && (p.cf & 1)) || (l >= p.n[0] && (p.cf & 2))) cmp = false;
            if ((lj < 0 && (p.cf & 4)) || (lj >= p.n[1] && (p.cf & 8))) cmp = false;
        }
        comp_ij[s] = cmp;
        out_ij[s] = (l >= 0) && (l < p.n[0]) && (lj >= 0) && (lj < p.n[1]) && (r >= 2) && (r < wi + 2) &&
                    (row >= 2) && (row < NR - 2);
    }
    const long long sj = p.pj, sk = p.pk;
    const long long base = p.off + li + sj * lj;

    // ---- prologue: planes k0-2 and k0-1 ------------------------------------------------------------
    int k = t.k0 - 1;  // first red plane (the ring below the tile)
    bool fk = (k - 1 >= -FRAME) && (k - 1 < p.n[2] + FRAME);
    double2 Pm = ldphi(k - 1, f0 && fk, f1 && fk);
    fk = (k >= -FRAME) && (k < p.n[2] + FRAME);
    double2 Pc = ldphi(k, f0 && fk, f1 && fk);
    // Jg^zz on the LOW face of plane k
    double2 Gzc = UNI ? uni2(P.uc[2], c0 && fk, c1 && fk) : ld2(jgz, base + sk * k, c0 && fk, c1 && fk, p.off);
    // coefficients of the black cell of plane k-1 (column c), captured one step earlier
    double b_rhs = 0, b_ji = 1, b_gxl = 0, b_gxh = 0, b_gyl = 0, b_gyh = 0, b_gzl = 0;
    double redPrev1 = 0.0, redPrev2 = 0.0;
    // UNI: with the coefficient streams gone a plane's loads are 2 x 16 bytes per lane, too little in flight to cover the memory
    // latency a lock-stepped (one barrier per plane) workgroup exposes: phi and rhs are fetched ONE PLANE FURTHER AHEAD
    // (phi of plane k+2 and rhs of plane k+1 are issued in step k and consumed in step k+1)
    double2 Pn = make_double2(0.0, 0.0), Rn = make_double2(0.0, 0.0);
    if (UNI) {
        const bool fkp0 = (k + 1 >= -FRAME) && (k + 1 < p.n[2] + FRAME);
        Pn = ldphi(k + 1, f0 && fkp0, f1 && fkp0);
        Rn = ld2(rhs, base + sk * k, c0 && fk, c1 && fk, p.off);
    }

    const int kend = t.k0 + t.nk;  // last red plane (the ring above the tile)
    // One plane of the march.  The colour column c = (i + j + k) & 1 of the pair is the same for every lane of a wavefront
    // (li is even, gj and gk are wave-uniform) and alternates from plane to plane: the body is instantiated for c = 0 and
    // c = 1 and the loop below calls them in turn, so every select between the pair's two members is resolved at compile
    // time (a quarter of the loop's vector instructions were v_cndmask on a lane-varying-looking c).
    auto step = [&](auto CC) {
        constexpr int c = decltype(CC)::value;
        const int gk = p.lo[2] + k;
        fk = (k >= -FRAME) && (k < p.n[2] + FRAME);
        const bool fkp = (k + 1 >= -FRAME) && (k + 1 < p.n[2] + FRAME);
        // ---- this step's loads: phi and Jg^zz of plane k+1, cell coefficients of plane k --------
        double2 Pp, Rh;
        if (UNI) {
            Pp = Pn;
            Rh = Rn;
            const bool fkpp = (k + 2 >= -FRAME) && (k + 2 < p.n[2] + FRAME);
            Pn = ldphi(k + 2, f0 && fkpp, f1 && fkpp);
            Rn = ld2(rhs, base + sk * (k + 1), c0 && fkp, c1 && fkp, p.off);
        } else {
            Pp = ldphi(k + 1, f0 && fkp, f1 && fkp);
            Rh = ld2(rhs, base + sk * k, c0 && fk, c1 && fk, p.off);
        }
        const double2 Gzp = UNI ? uni2(P.uc[2], c0 && fkp, c1 && fkp) : ld2(jgz, base + sk * (k + 1), c0 && fkp, c1 && fkp, p.off);
        const double2 Ji = UNI ? uni2(P.uc[3], c0 && fk, c1 && fk) : ld2(jinv, base + sk * k, c0 && fk, c1 && fk, p.off);
        const double2 Gx = UNI ? uni2(P.uc[0], c0 && fk, c1 && fk) : ld2(jgx, base + sk * k, c0 && fk, c1 && fk, p.off);
        const double2 Gy = UNI ? uni2(P.uc[1], c0 && fk, c1 && fk) : ld2(jgy, base + sk * k, c0 && fk, c1 && fk, p.off);
        const double2 Gyh = UNI ? uni2(P.uc[1], c0 && fk && fjh, c1 && fk && fjh)
                                : ld2(jgy, base + sk * k + sj, c0 && fk && fjh, c1 && fk && fjh, p.off);
        // Jg^xx on the face right of the pair = first component of the next lane's pair
        const double gx_next = __shfl_down(Gx.x, 1, 64);

        // ---- stage plane k (old values) in LDS; ONE barrier per plane -----------------------------
        //   slot k%3 was last read two steps ago (black of plane k-3); every wave has passed the
        //   previous barrier since, so overwriting it is safe.
        const int slot = ((k % 3) + 3) % 3;
        *reinterpret_cast<double2*>(&Sx(slot, row, ri)) = Pc;
        __syncthreads();

        // ---- red(k) at column c: reads only black cells of the plane, writes only red ones ---------
        // red = pass 0 = even i+j+k (GSRBF.ChF:381-387): c is this plane's red column of the pair
        const int rc = ri + c;
        double red = pick(Pc, c);  // cells not computed here keep their old value
        {
            bool comp = comp_ij[c] && (k >= -1) && (k <= p.n[2]);
            if ((gk < P.dom_lo[2] && (P.neum[2][0] || (DIRI && P.diri[2][0]))) ||
                (gk > P.dom_hi[2] && (P.neum[2][1] || (DIRI && P.diri[2][1]))))
                comp = false;
            if ((k < 0 && (p.cf & 16)) || (k >= p.n[2] && (p.cf & 32))) comp = false;
            if (comp) {
                const double pxl = Sx(slot, row, rc - 1), pxh = Sx(slot, row, rc + 1);
                const double pyl = Sx(slot, row - 1, rc), pyh = Sx(slot, row + 1, rc);
                red = gsrb_point<DIRI, INT>(P, xxS, yyS, zzS, p.lo[0] + li + c, gj, gk, pxl, pxh, pyl, pyh, pick(Pm, c),
                                 pick(Pp, c), c ? Gx.y : Gx.x, c ? gx_next : Gx.y, pick(Gy, c), pick(Gyh, c),
                                 pick(Gzc, c), pick(Gzp, c), pick(Ji, c), pick(Rh, c), red);
                Sx(slot, row, rc) = red;  // visible to the black phase of the NEXT step (after its barrier)
            }
        }

        // ---- black(k-1) at the same column c: in-plane neighbours = new red of plane k-1 (LDS),
        //      below = red(c,k-2) and above = red(c,k), both computed by this very lane -------------
        {
            const int kb = k - 1;
            if ((kb >= t.k0) && (kb < t.k0 + t.nk) && (out_ij[0] || out_ij[1])) {
                const int sb = ((kb % 3) + 3) % 3;
                double black = 0.0;
                if (out_ij[c]) {
                    double pxl = Sx(sb, row, rc - 1), pxh = Sx(sb, row, rc + 1);
                    double pyl = Sx(sb, row - 1, rc), pyh = Sx(sb, row + 1, rc);
                    double pzl = redPrev2, pzh = red;
                    if (INT ? (p.cf & 48) != 0 : p.cf != 0) {
                        // homogeneousCFInterp between the colours (LevelGSRB refills the CF ghosts before the
                        // black pass): ghost = c1 * first valid cell (this black cell, old value) + c2 * second
                        // valid cell (its opposite neighbour, a NEW red value)
                        const double own = Sx(sb, row, rc);
                        const int l = li + c;
                        const double xl = pxl, xh = pxh, yl = pyl, yh = pyh, zl = pzl, zh = pzh;
                        if (!INT) {
                            if ((p.cf & 1) && l == 0) pxl = P.cf_c1[0] * own + P.cf_c2[0] * xh;
                            if ((p.cf & 2) && l == p.n[0] - 1) pxh = P.cf_c1[0] * own + P.cf_c2[0] * xl;
                            if ((p.cf & 4) && lj == 0) pyl = P.cf_c1[1] * own + P.cf_c2[1] * yh;
                            if ((p.cf & 8) && lj == p.n[1] - 1) pyh = P.cf_c1[1] * own + P.cf_c2[1] * yl;
                        }
                        if ((p.cf & 16) && kb == 0) pzl = P.cf_c1[2] * own + P.cf_c2[2] * zh;
                        if ((p.cf & 32) && kb == p.n[2] - 1) pzh = P.cf_c1[2] * own + P.cf_c2[2] * zl;
                    }
                    black = gsrb_point<DIRI, INT>(P, xxS, yyS, zzS, p.lo[0] + li + c, gj, p.lo[2] + kb, pxl, pxh, pyl, pyh,
                                       pzl, pzh, b_gxl, b_gxh, b_gyl, b_gyh, b_gzl, pick(Gzc, c), b_ji, b_rhs,
                                       Sx(sb, row, rc));
                }
                // plane k-1: column c is the new black, column c^1 is red(k-1) (= redPrev1)
                double* dst = phi_out + base + sk * kb;
                const double ox = c ? redPrev1 : black, oy = c ? black : redPrev1;
                if (out_ij[0] && out_ij[1]) *reinterpret_cast<double2*>(dst) = make_double2(ox, oy);
                else if (out_ij[0]) dst[0] = ox;
                else dst[1] = oy;
            }
        }

        // ---- rotate: the next black cell is column c^1 of plane k ------------------------------------
        {
            const int cb = c ^ 1;
            b_rhs = pick(Rh, cb);
            b_ji = pick(Ji, cb);
            b_gxl = cb ? Gx.y : Gx.x;
            b_gxh = cb ? gx_next : Gx.y;
            b_gyl = pick(Gy, cb);
            b_gyh = pick(Gyh, cb);
            b_gzl = pick(Gzc, cb);
        }
        redPrev2 = redPrev1;
        redPrev1 = red;
        Pm = Pc;
        Pc = Pp;
        Gzc = Gzp;
    };
    const int cfirst = __builtin_amdgcn_readfirstlane((p.lo[0] + li + gj + p.lo[2] + k) & 1);
    if (cfirst) {
        for (;;) {
            if (k > kend) break;
            step(std::integral_constant<int, 1>{});
            ++k;
            if (k > kend) break;
            step(std::integral_constant<int, 0>{});
            ++k;
        }
    } else {
        for (;;) {
            if (k > kend) break;
            step(std::integral_constant<int, 0>{});
            ++k;
            if (k > kend) break;
            step(std::integral_constant<int, 1>{});
            ++k;
        }
    }
#undef Sx
}

// NARROW: the tile table holds narrow lane classes (uniform-metric depths, or SOMAR_NARROW_7PT=1); the instantiation without
// them is the one-body kernel with 48 KB of LDS that the HBM-bound streaming sweep was tuned as.
template <int FR_J, int INMODE, bool DIRI = false, bool UNI = false, bool NARROW = false>
__global__ __launch_bounds__(64 * FR_J) void k_gsrb_fused(const Tile* __restrict__ tiles,
                                                     const PatchDesc* __restrict__ patches,
                                                     double* __restrict__ phi_out,
                                                     const double* __restrict__ phi_in,
                                                     const double* __restrict__ rhs,
                                                     const double* __restrict__ jgx,
                                                     const double* __restrict__ jgy,
                                                     const double* __restrict__ jgz,
                                                     const double* __restrict__ jinv, StencilParams P,
                                                     const double* __restrict__ sums,
                                                     const PatchDesc* __restrict__ cpatches,
                                                     const double* __restrict__ crse, int r0, int r1, int r2, int lean_ok)
{
    // one slot = the largest class's region: FR_J rows of 128, or 16 FR_J rows of 8 + 2
    __shared__ __attribute__((aligned(16))) double S[3 * FR_J * (NARROW ? 160 : 128)];
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int cls = NARROW ? t.pad_[1] : 0;
    bool lean = false;
    if (UNI && !DIRI) {
        // does the tile grown by its red ring stay clear of every domain face (periodic seams included: the boundary form of
        // the update applies there too) and of every coarse-fine face, in x and y?  One scalar test per workgroup.
        const int wi = t.pad_[0] > 0 ? t.pad_[0] : 124;
        const int ilo = t.i0 - 1, ihi = min(t.i0 + wi, p.n[0]), jlo = t.j0 - 1, jhi = min(t.j0 + (FR_J << cls) - 4, p.n[1]);
        lean = lean_ok && p.lo[0] + ilo > P.dom_lo[0] && p.lo[0] + ihi < P.dom_hi[0] && p.lo[1] + jlo > P.dom_lo[1] &&
               p.lo[1] + jhi < P.dom_hi[1] && !((p.cf & 1) && ilo < 0) && !((p.cf & 2) && ihi >= p.n[0]) &&
               !((p.cf & 4) && jlo < 0) && !((p.cf & 8) && jhi >= p.n[1]);
    }
#define SOMAR_FUSED_BODY(INT_, CLS_)                                                                                          \
    gsrb_fused_body<FR_J, INMODE, DIRI, UNI, INT_, CLS_>(S, t, p, phi_out, phi_in, rhs, jgx, jgy, jgz, jinv, P, sums, cpatches, \
                                                        crse, r0, r1, r2)
    if (!NARROW || cls == 0) {
        if (lean) SOMAR_FUSED_BODY(UNI && !DIRI, 0);
        else SOMAR_FUSED_BODY(false, 0);
    } else if (cls == 1) {
        if (lean) SOMAR_FUSED_BODY(UNI && !DIRI, NARROW ? 1 : 0);
        else SOMAR_FUSED_BODY(false, NARROW ? 1 : 0);
    } else {
        SOMAR_FUSED_BODY(false, NARROW ? 4 : 0);   // remainder columns sit on a box edge: never clear of it
    }
#undef SOMAR_FUSED_BODY
}

void launch_gsrb_fused(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* phi_out,
                       const double* phi_in, const double* rhs, int in_mode, const double* sums,
                       const LevelDev* C, const double* crse, const int* r)
{
    if (ntiles == 0) return;
    const PatchDesc* cpatches = C ? C->patches : nullptr;
    const int r0 = r ? r[0] : 1, r1 = r ? r[1] : 1, r2 = r ? r[2] : 1;
    static const int lean_ok = getenv("SOMAR_NO_LEAN_TILES") == nullptr;   // A/B switch: interior tiles take the general path too
    SOMAR_CHECK(in_mode < 3 || ((r0 == 1 || r0 == 2) && (r1 == 1 || r1 == 2) && (r2 == 1 || r2 == 2)),
                "internal: the prolongation folded into a sweep takes multigrid ratios of 1 or 2 per direction");
    auto go = [&](auto kern, int rows) {
        hipLaunchKernelGGL(kern, dim3(ntiles), dim3(64, rows, 1), 0, st, tiles, L.patches, phi_out, phi_in, rhs, L.jg[0], L.jg[1],
                           L.jg[2], L.jinv, L.P, sums, cpatches, crse, r0, r1, r2, lean_ok);
    };
    // (rows, mode, Dirichlet sides, uniform metric, narrow lane classes in the tile table)
#define SOMAR_FUSED_MODES(ROWS, D, U, N, M0, M1, M2, M3, M4)                    \
    switch (in_mode) {                                                          \
        case 1: go(k_gsrb_fused<ROWS, M1, D, U, N>, ROWS); break;               \
        case 2: go(k_gsrb_fused<ROWS, M2, D, U, N>, ROWS); break;               \
        case 3: go(k_gsrb_fused<ROWS, M3, D, U, N>, ROWS); break;               \
        case 4: go(k_gsrb_fused<ROWS, M4, D, U, N>, ROWS); break;               \
        default: go(k_gsrb_fused<ROWS, M0, D, U, N>, ROWS);                     \
    }
    const bool rows16 = fused_rows() == 16;
    const bool narrow = L.narrow7 != 0 && rows16;   // Level::build_march_tiles hands the 8-row kernels class-0 tiles only
    bool diri = false;
    for (int d = 0; d < 3; ++d) diri = diri || L.P.diri[d][0] || L.P.diri[d][1];
    if (diri) {
        // Dirichlet sides: no null space, hence never a mean removal (modes 2 / 4 fall back to 0 / 3 in the table below; checked)
        SOMAR_CHECK(in_mode == 0 || in_mode == 1 || in_mode == 3, "internal: mean removal on a level with Dirichlet sides");
        if (!rows16) { SOMAR_FUSED_MODES(8, true, false, false, 0, 1, 0, 3, 3) }
        else if (narrow) { SOMAR_FUSED_MODES(16, true, false, true, 0, 1, 0, 3, 3) }
        else { SOMAR_FUSED_MODES(16, true, false, false, 0, 1, 0, 3, 3) }
        return;
    }
    if (L.P.uniform && rows16) {
        // uniform metric: the four coefficient streams come from StencilParams
        if (narrow) { SOMAR_FUSED_MODES(16, false, true, true, 0, 1, 2, 3, 4) }
        else { SOMAR_FUSED_MODES(16, false, true, false, 0, 1, 2, 3, 4) }
        return;
    }
    if (!rows16) { SOMAR_FUSED_MODES(8, false, false, false, 0, 1, 2, 3, 4) }
    else if (narrow) { SOMAR_FUSED_MODES(16, false, false, true, 0, 1, 2, 3, 4) }
    else { SOMAR_FUSED_MODES(16, false, false, false, 0, 1, 2, 3, 4) }
#undef SOMAR_FUSED_MODES
}

// region rows per workgroup of the fused sweep (tile rows = rows - 4); SOMAR_FUSED_ROWS = 8 | 16
int fused_rows()
{
    static int rows = 0;
    if (!rows) {
        const char* e = getenv("SOMAR_FUSED_ROWS");
        rows = (e && atoi(e) == 8) ? 8 : 16;
    }
    return rows;
}

}  // namespace somar
