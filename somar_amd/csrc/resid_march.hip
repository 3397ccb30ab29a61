// somar_amd/csrc/resid_march.hip -- operator / residual of a large level as a k-marching, LDS-staged kernel.
//
// Why: k_op_ortho (kernels.hip) reads phi through L2 and, measured with rocprofv3 PMC at 512^3, moves 136 B
// per cell over the fabric against 56 algorithmic: the k+-1 (and j+-1) reuse distance of a 7-point stencil
// does not fit the 4 MiB per-XCD L2 when 256 workgroups stream different slabs.  Here a workgroup owns a
// (124 x 14) column of cells and marches in k, exactly like the fused GSRB sweep (gsrb_fused.hip):
//   * phi of plane k is staged in LDS (2 rotating planes of 128 x 16 doubles, one wavefront per row, one
//     double2 per lane, ONE barrier per plane); the k-neighbours and Jg^zz stay in registers;
//   * every array is read once: phi 8 (x 128*16/(124*14) halo), rhs 8, Jg 24, Jinv 8, out 8 = ~58 B/cell.
//
// Arithmetic: the expression order of k_op_ortho (= MAPPEDGETFLUXORTHO + zero Neumann flux + flux*=beta +
// MAPPEDFLUXDIVERGENCE3D + AXBYIP + SUBTRACTOP) => bit-identical to it and to the CPU oracle.
// MODE 0: out = rhs - L[phi]   MODE 1: out = L[phi]
#include "common.h"
#include "kernels.h"

namespace somar {

constexpr int RM_J = 16;   // region rows = 14-row tile + 1 low + 1 high

__device__ __forceinline__ double2 rm_ld2(const double* __restrict__ a, long long idx, bool ok0, bool ok1,
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

// unconditional: the zeros rm_ld2 returns under a false predicate only reach cells that are not written (see gsrb_fused.hip)
__device__ __forceinline__ double2 rm_uni2(double c, bool, bool) { return make_double2(c, c); }

// UNI: uniform metric, the four coefficient arrays are not read (StencilParams::uc)
// CLS: the tile's lane class (see full19_march.hip): a wavefront covers 2^CLS region rows of 128 >> CLS columns
template <int MODE, bool UNI, int CLS>
__device__ __forceinline__ void resid_march_body(double* __restrict__ S, double* __restrict__ T, const Tile& t,
                                                 const PatchDesc& p, double* __restrict__ out,
                                                 const double* __restrict__ phi, const double* __restrict__ rhs,
                                                 const double* __restrict__ jgx, const double* __restrict__ jgy,
                                                 const double* __restrict__ jgz, const double* __restrict__ jinv,
                                                 const StencilParams& P, const PatchDesc* __restrict__ cpatches, int r0,
                                                 int r1, int r2, double dxProduct, double* __restrict__ volsum)
{
    constexpr int LPR = 64 >> CLS;                 // lanes per region row
    constexpr int NR = RM_J << CLS;                // region rows of the workgroup
    constexpr int PITCH = 2 * LPR + (CLS >= 2 ? 2 : 0);
#define Sx(slot, r, c) S[((slot) * NR + (r)) * PITCH + (c)]
#define Tx(slot, r, l) (T + ((((slot) * NR + (r)) * LPR + (l)) << 2))
    // class 0: the region row is the wavefront's index, a scalar -- everything derived from it stays in scalar registers
    const int lane = CLS == 0 ? (int)threadIdx.x : (int)(threadIdx.x & (LPR - 1));
    const int row = CLS == 0 ? (int)threadIdx.y
                             : (int)(((threadIdx.y >> 1) << (CLS + 1)) + (threadIdx.y & 1) + 2 * (threadIdx.x >> (6 - CLS)));
    const int ri = 2 * lane;
    const int li = t.i0 - 2 + ri;  // even: rows are 16-byte aligned
    const int wi = t.pad_[0] > 0 ? t.pad_[0] : 2 * LPR - 4;  // output columns of this tile (see gsrb_fused.hip)
    const int lj = t.j0 - 1 + row;
    const int gj = p.lo[1] + lj;
    const double sx = 1.0 / P.dx[0], sy = 1.0 / P.dx[1], sz = 1.0 / P.dx[2];

    // phi may be touched inside the 1-cell ghost layer, coefficients only at the tile's own cells / faces
    const bool fj = (lj >= -1) && (lj <= p.n[1]);
    const bool f0 = fj && (li >= -1) && (li <= p.n[0]) && (ri < wi + 4);
    const bool f1 = fj && (li + 1 >= -1) && (li + 1 <= p.n[0]) && (ri + 1 < wi + 4);
    const bool own_j = (lj >= 0) && (lj < p.n[1]) && (row >= 1) && (row <= NR - 2) && (lj < t.j0 + (NR - 2));
    bool o[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int l = li + s, r = ri + s;
        o[s] = own_j && (l >= 0) && (l < p.n[0]) && (r >= 2) && (r < wi + 2);
    }
    // Jg^xx is also needed on the face right of the pair's second cell (next lane's first component): load the
    // pair if either of its cells, or the cell left of it, is an output cell
    // (the shuffle must run with all lanes active: keep it out of the short-circuit)
    const int left_o1 = __shfl_up((int)o[1], 1, 64);
    const bool gxo0 = o[0] || (left_o1 != 0);
    const bool any = o[0] || o[1];
    const long long sj = p.pj, sk = p.pk;
    const long long base = p.off + li + sj * lj;
    const bool zyl = (gj == P.dom_lo[1]) && P.neum[1][0];
    const bool zyh = (gj == P.dom_hi[1]) && P.neum[1][1];
    bool zxl[2], zxh[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int gi = p.lo[0] + li + s;
        zxl[s] = (gi == P.dom_lo[0]) && P.neum[0][0];
        zxh[s] = (gi == P.dom_hi[0]) && P.neum[0][1];
    }

    // MODE 2 state: this pair's (res/J, 1/J) of the previous plane, running sums of up to two coarse cells
    double pv[4] = {0.0, 0.0, 0.0, 0.0}, cs[2] = {0.0, 0.0}, cjs[2] = {0.0, 0.0};
    const bool leader = (MODE == 2) && any && (r1 == 1 || ((lj & 1) == 0));
    auto accumulate = [&](int kp) {
        // MAPPEDAVERAGE2's loop order: ii2 (planes), ii1 (rows), ii0 (cells of the pair)
        if (!leader) return;
        const bool first = (r2 == 1) || ((kp & 1) == 0);
        const bool last = (r2 == 1) || ((kp & 1) == 1);
        if (first) { cs[0] = cs[1] = 0.0; cjs[0] = cjs[1] = 0.0; }
        double q[4] = {0.0, 0.0, 0.0, 0.0};
        if (r1 == 2) {
            const double* src = Tx(kp & 1, row + 1, lane);
            q[0] = src[0]; q[1] = src[1]; q[2] = src[2]; q[3] = src[3];
        }
        if (r0 == 2) {
            cs[0] = cs[0] + pv[0]; cjs[0] = cjs[0] + pv[2];
            cs[0] = cs[0] + pv[1]; cjs[0] = cjs[0] + pv[3];
            if (r1 == 2) {
                cs[0] = cs[0] + q[0]; cjs[0] = cjs[0] + q[2];
                cs[0] = cs[0] + q[1]; cjs[0] = cjs[0] + q[3];
            }
        } else {
            cs[0] = cs[0] + pv[0]; cjs[0] = cjs[0] + pv[2];
            cs[1] = cs[1] + pv[1]; cjs[1] = cjs[1] + pv[3];
            if (r1 == 2) {
                cs[0] = cs[0] + q[0]; cjs[0] = cjs[0] + q[2];
                cs[1] = cs[1] + q[1]; cjs[1] = cjs[1] + q[3];
            }
        }
        if (last) {
            const PatchDesc cp = cpatches[t.patch];
            const long long c = cp.off + (li / r0) + (long long)cp.pj * (lj / r1) + cp.pk * (kp / r2);
            if (r0 == 2) {
                out[c] = cs[0] / cjs[0];
            } else {
                if (o[0]) out[c] = cs[0] / cjs[0];
                if (o[1]) out[c + 1] = cs[1] / cjs[1];
            }
        }
    };

    // MODE 2 with volsum: this block's sum of dvol * phi over its own cells (dvol = dxProduct / Jinv), the fine
    // half of the zero-average prolongation's mean (see PressureSolver::cycle); tree order
    double vsum = 0.0;

    int k = t.k0;
    double2 Pm = rm_ld2(phi, base + sk * (k - 1), o[0], o[1], p.off);
    double2 Pc = rm_ld2(phi, base + sk * k, f0, f1, p.off);
    double2 Gzc = UNI ? rm_uni2(P.uc[2], o[0], o[1]) : rm_ld2(jgz, base + sk * k, o[0], o[1], p.off);
    const int kend = t.k0 + t.nk;
    for (; k < kend; ++k) {
        const int gk = p.lo[2] + k;
        const bool more = (k + 1 < kend);
        // ---- this step's loads ----
        const double2 Pp = rm_ld2(phi, base + sk * (k + 1), more ? f0 : o[0], more ? f1 : o[1], p.off);
        const double2 Gzp = UNI ? rm_uni2(P.uc[2], o[0], o[1]) : rm_ld2(jgz, base + sk * (k + 1), o[0], o[1], p.off);
        double2 Rh = make_double2(0.0, 0.0);
        if (MODE != 1) Rh = rm_ld2(rhs, base + sk * k, o[0], o[1], p.off);
        const double2 Ji = UNI ? rm_uni2(P.uc[3], o[0], o[1]) : rm_ld2(jinv, base + sk * k, o[0], o[1], p.off);
        const double2 Gx = UNI ? rm_uni2(P.uc[0], gxo0, o[0] || o[1]) : rm_ld2(jgx, base + sk * k, gxo0, o[0] || o[1], p.off);
        const double2 Gy = UNI ? rm_uni2(P.uc[1], o[0], o[1]) : rm_ld2(jgy, base + sk * k, o[0], o[1], p.off);
        const double2 Gyh = UNI ? rm_uni2(P.uc[1], o[0], o[1]) : rm_ld2(jgy, base + sk * k + sj, o[0], o[1], p.off);
        const double gx_next = __shfl_down(Gx.x, 1, 64);

        // ---- stage plane k; slot k&1 was last read two steps ago, one barrier per plane suffices ----
        const int slot = k & 1;
        *reinterpret_cast<double2*>(&Sx(slot, row, ri)) = Pc;
        __syncthreads();
        if (MODE == 2 && k > t.k0) accumulate(k - 1);

        if (any) {
            const bool zzl = (gk == P.dom_lo[2]) && P.neum[2][0];
            const bool zzh = (gk == P.dom_hi[2]) && P.neum[2][1];
            double res[2] = {0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (!o[s]) continue;
                const int rc = ri + s;
                const double pc = s ? Pc.y : Pc.x;
                const double pxl = s ? Pc.x : Sx(slot, row, rc - 1);
                const double pxh = s ? Sx(slot, row, rc + 1) : Pc.y;
                const double pyl = Sx(slot, row - 1, rc), pyh = Sx(slot, row + 1, rc);
                const double gxl = s ? Gx.y : Gx.x, gxh = s ? gx_next : Gx.y;
                double fxl = gxl * sx * (pc - pxl);
                double fxh = gxh * sx * (pxh - pc);
                double fyl = (s ? Gy.y : Gy.x) * sy * (pc - pyl);
                double fyh = (s ? Gyh.y : Gyh.x) * sy * (pyh - pc);
                double fzl = (s ? Gzc.y : Gzc.x) * sz * (pc - (s ? Pm.y : Pm.x));
                double fzh = (s ? Gzp.y : Gzp.x) * sz * ((s ? Pp.y : Pp.x) - pc);
                if (zxl[s]) fxl = 0.0;
                if (zxh[s]) fxh = 0.0;
                if (zyl) fyl = 0.0;
                if (zyh) fyh = 0.0;
                if (zzl) fzl = 0.0;
                if (zzh) fzh = 0.0;
                fxl *= P.beta; fxh *= P.beta; fyl *= P.beta; fyh *= P.beta; fzl *= P.beta; fzh *= P.beta;
                double l = (s ? Ji.y : Ji.x) * ((fxh - fxl) * sx + (fyh - fyl) * sy + (fzh - fzl) * sz);
                if (P.alpha != 0.0) l = P.alpha * pc + 1.0 * l;
                res[s] = (MODE != 1) ? ((s ? Rh.y : Rh.x) - l) : l;
            }
            if (MODE == 2) {
                // coarseSum + fine/J, coarseCCJSum + 1.0/J  (MAPPEDAVERAGE2)
                pv[0] = o[0] ? res[0] / Ji.x : 0.0;
                pv[1] = o[1] ? res[1] / Ji.y : 0.0;
                pv[2] = o[0] ? 1.0 / Ji.x : 0.0;
                pv[3] = o[1] ? 1.0 / Ji.y : 0.0;
                if (volsum) {
                    if (o[0]) vsum = vsum + dxProduct * pv[2] * Pc.x;
                    if (o[1]) vsum = vsum + dxProduct * pv[3] * Pc.y;
                }
                if (r1 == 2) {
                    double* dstT = Tx(k & 1, row, lane);
                    *reinterpret_cast<double2*>(dstT) = make_double2(pv[0], pv[1]);
                    *reinterpret_cast<double2*>(dstT + 2) = make_double2(pv[2], pv[3]);
                }
            } else {
                double* dst = out + base + sk * k;
                if (o[0] && o[1]) *reinterpret_cast<double2*>(dst) = make_double2(res[0], res[1]);
                else if (o[0]) dst[0] = res[0];
                else dst[1] = res[1];
            }
        }
        Pm = Pc;
        Pc = Pp;
        Gzc = Gzp;
    }
    if (MODE == 2) {
        __syncthreads();
        accumulate(kend - 1);
        if (volsum) {
            __shared__ double red[RM_J];
            for (int o2 = 32; o2 > 0; o2 >>= 1) vsum += __shfl_down(vsum, o2, 64);
            if (threadIdx.x == 0) red[threadIdx.y] = vsum;
            __syncthreads();
            if (threadIdx.x == 0 && threadIdx.y == 0) {
                double tot = red[0];
                for (int q = 1; q < RM_J; ++q) tot = tot + red[q];
                volsum[t.pad_[2]] = tot;   // the tile's place in the level's full tile list: split launches (overlap) keep the order of the sum
            }
        }
    }
#undef Sx
#undef Tx
}

template <int MODE, bool UNI = false>
// uniform metric, plain output: 64 VGPRs, two workgroups per CU (the narrow classes' per-lane row arithmetic must not cost that)
__global__ __launch_bounds__(64 * RM_J, (UNI && MODE != 2) ? 8 : 4) void k_resid_march(const Tile* __restrict__ tiles,
                                                           const PatchDesc* __restrict__ patches,
                                                           double* __restrict__ out,
                                                           const double* __restrict__ phi,
                                                           const double* __restrict__ rhs,
                                                           const double* __restrict__ jgx,
                                                           const double* __restrict__ jgy,
                                                           const double* __restrict__ jgz,
                                                           const double* __restrict__ jinv, StencilParams P,
                                                           const PatchDesc* __restrict__ cpatches, int r0, int r1,
                                                           int r2, double dxProduct, double* __restrict__ volsum)
{
    __shared__ __attribute__((aligned(16))) double S[2 * RM_J * 160];   // one slot = the largest class's region (16 RM_J rows of 8 + 2)
    // MODE 2: (res/J, 1/J) of both cells of every pair, handed from the odd row of a coarse cell to the even one
    __shared__ __attribute__((aligned(16))) double T[MODE == 2 ? 2 * RM_J * 64 * 4 : 4];
    const Tile t = tiles[blockIdx.x];
    const PatchDesc p = patches[t.patch];
    const int cls = t.pad_[1];
    if (cls == 0) resid_march_body<MODE, UNI, 0>(S, T, t, p, out, phi, rhs, jgx, jgy, jgz, jinv, P, cpatches, r0, r1, r2, dxProduct, volsum);
    else if (cls == 1) resid_march_body<MODE, UNI, 1>(S, T, t, p, out, phi, rhs, jgx, jgy, jgz, jinv, P, cpatches, r0, r1, r2, dxProduct, volsum);
    else resid_march_body<MODE, UNI, 4>(S, T, t, p, out, phi, rhs, jgx, jgy, jgz, jinv, P, cpatches, r0, r1, r2, dxProduct, volsum);
}

void launch_resid_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out,
                        const double* phi, const double* rhs, int mode)
{
    if (ntiles == 0) return;
    if (L.P.uniform) {
        if (mode == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_resid_march<0, true>), dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, L.patches, out,
                               phi, rhs, L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, nullptr, 1, 1, 1, 0.0, nullptr);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_resid_march<1, true>), dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, L.patches, out,
                               phi, rhs, L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, nullptr, 1, 1, 1, 0.0, nullptr);
        return;
    }
    if (mode == 0)
        hipLaunchKernelGGL(k_resid_march<0>, dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, L.patches, out, phi, rhs,
                           L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, nullptr, 1, 1, 1, 0.0, nullptr);
    else
        hipLaunchKernelGGL(k_resid_march<1>, dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, L.patches, out, phi, rhs,
                           L.jg[0], L.jg[1], L.jg[2], L.jinv, L.P, nullptr, 1, 1, 1, 0.0, nullptr);
}

// restrictResidual in one pass: crse = J-weighted average (MAPPEDAVERAGE2) of rhs - L[phi]; the fine residual is
// never written.  Needs tiles whose k-extent is even (Level::hrtiles are) and a coarsenable layout.
// volsum (optional, ntiles doubles): per-block sums of (dxProduct / Jinv) * phi over the block's cells
void launch_resid_restrict(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& F, const LevelDev& C,
                           double* crse, const double* phi, const double* rhs, const int r[3], double dxProduct,
                           double* volsum)
{
    if (ntiles == 0) return;
    if (F.P.uniform) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_resid_march<2, true>), dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, F.patches, crse, phi,
                           rhs, F.jg[0], F.jg[1], F.jg[2], F.jinv, F.P, C.patches, r[0], r[1], r[2], dxProduct, volsum);
        return;
    }
    hipLaunchKernelGGL(k_resid_march<2>, dim3(ntiles), dim3(64, RM_J, 1), 0, st, tiles, F.patches, crse, phi, rhs,
                       F.jg[0], F.jg[1], F.jg[2], F.jinv, F.P, C.patches, r[0], r[1], r[2], dxProduct, volsum);
}

}  // namespace somar
