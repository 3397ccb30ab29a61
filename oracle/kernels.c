/*
 * oracle/kernels.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the Chombo-Fortran (.ChF) kernels on SOMAR's
 * pressure-projection hot path.  Every function follows one reference
 * subroutine (file:line given, paths relative to /root/reference/src) with the
 * same loop nest and the same floating-point operation order, and is compiled
 * with -ffp-contract=off so no FMA is formed.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library; the
 * product (somar_amd/) never does.
 *
 * Parity status: the reference ships no tests or golden vectors for this path
 * and cannot be built here (Chombo 3.1 + ChF preprocessor absent), so against
 * the reference's own tests parity is UNPINNED; the restatement is pinned by
 * the analytic known-answer tests in tests/test_oracle_kats.py (SURVEY.md 8c
 * k1..k9) and by oracle/_ref (utils/ThomasAlgorithm.f90 built with flang).
 *
 * Array convention = Chombo FRA: column-major, inclusive [lo,hi] bounds,
 * component slowest.  All boxes are passed as 3-vectors; 2-D problems use a
 * flat third direction (lo2 == hi2 == 0) and the *2D* routines.
 * Face-centred arrays: face index i is the LOW face of cell i.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    double *p;
    long s1, s2, sc; /* strides in j, k, comp (elements) */
    int lo[3], hi[3];
} fra_t;

static fra_t mk(double *p, const int *lo, const int *hi)
{
    fra_t a;
    a.p = p;
    for (int d = 0; d < 3; ++d) { a.lo[d] = lo[d]; a.hi[d] = hi[d]; }
    a.s1 = (long)(hi[0] - lo[0] + 1);
    a.s2 = a.s1 * (long)(hi[1] - lo[1] + 1);
    a.sc = a.s2 * (long)(hi[2] - lo[2] + 1);
    return a;
}
#define AT(A, i, j, k, n) \
    ((A).p[((long)(i) - (A).lo[0]) + (A).s1 * ((long)(j) - (A).lo[1]) + \
           (A).s2 * ((long)(k) - (A).lo[2]) + (A).sc * (long)(n)])

#define BC_NEUM 0 /* calculus/BCInterface/BCDescriptor.H:34-39 */

/* parity-shifted start of the i loop, GSRBF.ChF:381-387 */
static int imin_rb(int ilo, int j, int k, int redBlack)
{
    int indtot = ilo + j + k;
    return ilo + abs((indtot + redBlack) % 2);
}

/* ------------------------------------------------------------------------
 * K1  GSRBITER3DORTHO   RelaxationMethods/GSRBF.ChF:545-701 (general branch
 * :660-697; the alpha=0,beta=1 fast path :616-658 is bitwise the same numbers)
 * ---------------------------------------------------------------------- */
void orc_gsrbiter3dortho(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                         const double *rhs_, const int *rlo, const int *rhi,
                         const double *jgxx_, const int *xlo, const int *xhi,
                         const double *jgyy_, const int *ylo, const int *yhi,
                         const double *jgzz_, const int *zlo, const int *zhi,
                         const double *jinv_, const int *jlo, const int *jhi,
                         const double *lapd_, const int *dlo, const int *dhi,
                         const int *reglo, const int *reghi, const double *dx,
                         double alpha, double beta, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jgxx = mk((double *)jgxx_, xlo, xhi), Jgyy = mk((double *)jgyy_, ylo, yhi);
    fra_t Jgzz = mk((double *)jgzz_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    fra_t lapDiag = mk((double *)lapd_, dlo, dhi);
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const double zzScale = 1.0 / (dx[2] * dx[2]);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int imin = imin_rb(reglo[0], j, k, redBlack);
                for (int i = imin; i <= reghi[0]; i += 2) {
                    double JDxx = xxScale * (AT(Jgxx, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n) +
                                             AT(Jgxx, i, j, k, 0) * AT(phi, i - 1, j, k, n));
                    double JDyy = yyScale * (AT(Jgyy, i, j + 1, k, 0) * AT(phi, i, j + 1, k, n) +
                                             AT(Jgyy, i, j, k, 0) * AT(phi, i, j - 1, k, n));
                    double JDzz = zzScale * (AT(Jgzz, i, j, k + 1, 0) * AT(phi, i, j, k + 1, n) +
                                             AT(Jgzz, i, j, k, 0) * AT(phi, i, j, k - 1, n));
                    double lphi = beta * AT(Jinv, i, j, k, 0) * (JDxx + JDyy + JDzz);
                    AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) /
                                          (alpha + beta * AT(lapDiag, i, j, k, 0));
                }
            }
}

/* K3  GSRBITER2DORTHO   GSRBF.ChF:442-542 (k fixed at 0) */
void orc_gsrbiter2dortho(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                         const double *rhs_, const int *rlo, const int *rhi,
                         const double *jgxx_, const int *xlo, const int *xhi,
                         const double *jgyy_, const int *ylo, const int *yhi,
                         const double *jinv_, const int *jlo, const int *jhi,
                         const double *lapd_, const int *dlo, const int *dhi,
                         const int *reglo, const int *reghi, const double *dx,
                         double alpha, double beta, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jgxx = mk((double *)jgxx_, xlo, xhi), Jgyy = mk((double *)jgyy_, ylo, yhi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi), lapDiag = mk((double *)lapd_, dlo, dhi);
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    for (int n = 0; n < ncomp; ++n)
        for (int j = reglo[1]; j <= reghi[1]; ++j) {
            int imin = imin_rb(reglo[0], j, 0, redBlack);
            for (int i = imin; i <= reghi[0]; i += 2) {
                double JDxx = xxScale * (AT(Jgxx, i + 1, j, 0, 0) * AT(phi, i + 1, j, 0, n) +
                                         AT(Jgxx, i, j, 0, 0) * AT(phi, i - 1, j, 0, n));
                double JDyy = yyScale * (AT(Jgyy, i, j + 1, 0, 0) * AT(phi, i, j + 1, 0, n) +
                                         AT(Jgyy, i, j, 0, 0) * AT(phi, i, j - 1, 0, n));
                double lphi = beta * (JDxx + JDyy) * AT(Jinv, i, j, 0, 0);
                AT(phi, i, j, 0, n) = (AT(rhs, i, j, 0, n) - lphi) /
                                      (alpha + beta * AT(lapDiag, i, j, 0, 0));
            }
        }
}

/* ------------------------------------------------------------------------
 * K4  GSRBBOUNDARYITER3DORTHO   GSRBF.ChF:1362-1505
 * JDlo/JDhi are zero-initialised once per call (:1388-1393, quirk Q9).
 * ---------------------------------------------------------------------- */
void orc_gsrbboundaryiter3dortho(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                                 const double *rhs_, const int *rlo, const int *rhi,
                                 const double *jgxx_, const int *xlo, const int *xhi,
                                 const double *jgyy_, const int *ylo, const int *yhi,
                                 const double *jgzz_, const int *zlo, const int *zhi,
                                 const double *jinv_, const int *jlo, const int *jhi,
                                 const int *reglo, const int *reghi, const double *dx,
                                 double alpha, double beta, const int *stencil /*loX,hiX,loY,hiY,loZ,hiZ*/,
                                 int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jgxx = mk((double *)jgxx_, xlo, xhi), Jgyy = mk((double *)jgyy_, ylo, yhi);
    fra_t Jgzz = mk((double *)jgzz_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = stencil[0], hiX = stencil[1], loY = stencil[2], hiY = stencil[3],
              loZ = stencil[4], hiZ = stencil[5];
    double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0;
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const double zzScale = 1.0 / (dx[2] * dx[2]);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int imin = imin_rb(reglo[0], j, k, redBlack);
                for (int i = imin; i <= reghi[0]; i += 2) {
                    double lapDiag = 0.0;
                    if (loX != BC_NEUM) {
                        JDloX = AT(Jgxx, i, j, k, 0) * AT(phi, i - 1, j, k, n);
                        lapDiag = lapDiag - xxScale * AT(Jgxx, i, j, k, 0);
                    }
                    if (loY != BC_NEUM) {
                        JDloY = AT(Jgyy, i, j, k, 0) * AT(phi, i, j - 1, k, n);
                        lapDiag = lapDiag - yyScale * AT(Jgyy, i, j, k, 0);
                    }
                    if (loZ != BC_NEUM) {
                        JDloZ = AT(Jgzz, i, j, k, 0) * AT(phi, i, j, k - 1, n);
                        lapDiag = lapDiag - zzScale * AT(Jgzz, i, j, k, 0);
                    }
                    if (hiX != BC_NEUM) {
                        JDhiX = AT(Jgxx, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n);
                        lapDiag = lapDiag - xxScale * AT(Jgxx, i + 1, j, k, 0);
                    }
                    if (hiY != BC_NEUM) {
                        JDhiY = AT(Jgyy, i, j + 1, k, 0) * AT(phi, i, j + 1, k, n);
                        lapDiag = lapDiag - yyScale * AT(Jgyy, i, j + 1, k, 0);
                    }
                    if (hiZ != BC_NEUM) {
                        JDhiZ = AT(Jgzz, i, j, k + 1, 0) * AT(phi, i, j, k + 1, n);
                        lapDiag = lapDiag - zzScale * AT(Jgzz, i, j, k + 1, 0);
                    }
                    lapDiag = lapDiag * AT(Jinv, i, j, k, 0);
                    double lphi = beta * AT(Jinv, i, j, k, 0) *
                                  ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale +
                                   (JDloZ + JDhiZ) * zzScale);
                    AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) / (alpha + beta * lapDiag);
                }
            }
}

/* K4  GSRBBOUNDARYITER2DORTHO   GSRBF.ChF:1233-1359 (order loX,hiX,loY,hiY) */
void orc_gsrbboundaryiter2dortho(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                                 const double *rhs_, const int *rlo, const int *rhi,
                                 const double *jgxx_, const int *xlo, const int *xhi,
                                 const double *jgyy_, const int *ylo, const int *yhi,
                                 const double *jinv_, const int *jlo, const int *jhi,
                                 const int *reglo, const int *reghi, const double *dx,
                                 double alpha, double beta, const int *stencil, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jgxx = mk((double *)jgxx_, xlo, xhi), Jgyy = mk((double *)jgyy_, ylo, yhi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = stencil[0], hiX = stencil[1], loY = stencil[2], hiY = stencil[3];
    double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0;
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    for (int n = 0; n < ncomp; ++n)
        for (int j = reglo[1]; j <= reghi[1]; ++j) {
            int imin = imin_rb(reglo[0], j, 0, redBlack);
            for (int i = imin; i <= reghi[0]; i += 2) {
                double lapDiag = 0.0;
                if (loX != BC_NEUM) {
                    JDloX = AT(Jgxx, i, j, 0, 0) * AT(phi, i - 1, j, 0, n);
                    lapDiag = lapDiag - xxScale * AT(Jgxx, i, j, 0, 0);
                }
                if (hiX != BC_NEUM) {
                    JDhiX = AT(Jgxx, i + 1, j, 0, 0) * AT(phi, i + 1, j, 0, n);
                    lapDiag = lapDiag - xxScale * AT(Jgxx, i + 1, j, 0, 0);
                }
                if (loY != BC_NEUM) {
                    JDloY = AT(Jgyy, i, j, 0, 0) * AT(phi, i, j - 1, 0, n);
                    lapDiag = lapDiag - yyScale * AT(Jgyy, i, j, 0, 0);
                }
                if (hiY != BC_NEUM) {
                    JDhiY = AT(Jgyy, i, j + 1, 0, 0) * AT(phi, i, j + 1, 0, n);
                    lapDiag = lapDiag - yyScale * AT(Jgyy, i, j + 1, 0, 0);
                }
                lapDiag = lapDiag * AT(Jinv, i, j, 0, 0);
                double lphi = beta * AT(Jinv, i, j, 0, 0) *
                              ((JDloX + JDhiX) * xxScale + (JDloY + JDhiY) * yyScale);
                AT(phi, i, j, 0, n) = (AT(rhs, i, j, 0, n) - lphi) / (alpha + beta * lapDiag);
            }
        }
}

/* ------------------------------------------------------------------------
 * K2  GSRBITER3D (full 19-point)   GSRBF.ChF:283-439
 * Jg0/Jg1/Jg2 have 3 comps each; cross terms read `extrap`.
 * ---------------------------------------------------------------------- */
void orc_gsrbiter3d(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                    const double *ext_, const int *elo, const int *ehi,
                    const double *rhs_, const int *rlo, const int *rhi,
                    const double *jg0_, const int *xlo, const int *xhi,
                    const double *jg1_, const int *ylo, const int *yhi,
                    const double *jg2_, const int *zlo, const int *zhi,
                    const double *jinv_, const int *jlo, const int *jhi,
                    const double *lapd_, const int *dlo, const int *dhi,
                    const int *reglo, const int *reghi, const double *dx,
                    double alpha, double beta, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jg2 = mk((double *)jg2_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    fra_t lapDiag = mk((double *)lapd_, dlo, dhi);
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const double zzScale = 1.0 / (dx[2] * dx[2]);
    const double xyScale = 0.25 / (dx[0] * dx[1]);
    const double yzScale = 0.25 / (dx[1] * dx[2]);
    const double zxScale = 0.25 / (dx[2] * dx[0]);
#define E(a, b, c) AT(extrap, a, b, c, n)
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int imin = imin_rb(reglo[0], j, k, redBlack);
                for (int i = imin; i <= reghi[0]; i += 2) {
                    double pdx = E(i + 1, j, k) - E(i - 1, j, k);
                    double pdy = E(i, j + 1, k) - E(i, j - 1, k);
                    double pdz = E(i, j, k + 1) - E(i, j, k - 1);

                    double JDxx = AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n) +
                                  AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, n);
                    double JDxy = AT(Jg0, i + 1, j, k, 1) * (E(i + 1, j + 1, k) - E(i + 1, j - 1, k) + pdy) -
                                  AT(Jg0, i, j, k, 1) * (pdy + E(i - 1, j + 1, k) - E(i - 1, j - 1, k));
                    double JDxz = AT(Jg0, i + 1, j, k, 2) * (E(i + 1, j, k + 1) - E(i + 1, j, k - 1) + pdz) -
                                  AT(Jg0, i, j, k, 2) * (pdz + E(i - 1, j, k + 1) - E(i - 1, j, k - 1));

                    double JDyx = AT(Jg1, i, j + 1, k, 0) * (E(i + 1, j + 1, k) - E(i - 1, j + 1, k) + pdx) -
                                  AT(Jg1, i, j, k, 0) * (pdx + E(i + 1, j - 1, k) - E(i - 1, j - 1, k));
                    double JDyy = AT(Jg1, i, j + 1, k, 1) * AT(phi, i, j + 1, k, n) +
                                  AT(Jg1, i, j, k, 1) * AT(phi, i, j - 1, k, n);
                    double JDyz = AT(Jg1, i, j + 1, k, 2) * (E(i, j + 1, k + 1) - E(i, j + 1, k - 1) + pdz) -
                                  AT(Jg1, i, j, k, 2) * (pdz + E(i, j - 1, k + 1) - E(i, j - 1, k - 1));

                    double JDzx = AT(Jg2, i, j, k + 1, 0) * (E(i + 1, j, k + 1) - E(i - 1, j, k + 1) + pdx) -
                                  AT(Jg2, i, j, k, 0) * (pdx + E(i + 1, j, k - 1) - E(i - 1, j, k - 1));
                    double JDzy = AT(Jg2, i, j, k + 1, 1) * (E(i, j + 1, k + 1) - E(i, j - 1, k + 1) + pdy) -
                                  AT(Jg2, i, j, k, 1) * (pdy + E(i, j + 1, k - 1) - E(i, j - 1, k - 1));
                    double JDzz = AT(Jg2, i, j, k + 1, 2) * AT(phi, i, j, k + 1, n) +
                                  AT(Jg2, i, j, k, 2) * AT(phi, i, j, k - 1, n);

                    double lphi = beta * AT(Jinv, i, j, k, 0) *
                                  (JDxx * xxScale + JDyy * yyScale + JDzz * zzScale +
                                   (JDxy + JDyx) * xyScale + (JDyz + JDzy) * yzScale +
                                   (JDzx + JDxz) * zxScale);
                    AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) /
                                          (alpha + beta * AT(lapDiag, i, j, k, 0));
                }
            }
#undef E
}

/* K4  GSRBBOUNDARYITER3D (full)   GSRBF.ChF:1024-1230 */
void orc_gsrbboundaryiter3d(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                            const double *ext_, const int *elo, const int *ehi,
                            const double *rhs_, const int *rlo, const int *rhi,
                            const double *jg0_, const int *xlo, const int *xhi,
                            const double *jg1_, const int *ylo, const int *yhi,
                            const double *jg2_, const int *zlo, const int *zhi,
                            const double *jinv_, const int *jlo, const int *jhi,
                            const int *reglo, const int *reghi, const double *dx,
                            double alpha, double beta, const int *stencil, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jg2 = mk((double *)jg2_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = stencil[0], hiX = stencil[1], loY = stencil[2], hiY = stencil[3],
              loZ = stencil[4], hiZ = stencil[5];
    double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0, JDloZ = 0, JDhiZ = 0;
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const double zzScale = 1.0 / (dx[2] * dx[2]);
    const double xyScale = 0.25 / (dx[0] * dx[1]);
    const double yzScale = 0.25 / (dx[1] * dx[2]);
    const double zxScale = 0.25 / (dx[2] * dx[0]);
#define E(a, b, c) AT(extrap, a, b, c, n)
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int imin = imin_rb(reglo[0], j, k, redBlack);
                for (int i = imin; i <= reghi[0]; i += 2) {
                    double lapDiag = 0.0;
                    if (loX != BC_NEUM) {
                        JDloX = +xxScale * AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, n) -
                                xyScale * AT(Jg0, i, j, k, 1) *
                                    (E(i, j + 1, k) - E(i, j - 1, k) + E(i - 1, j + 1, k) - E(i - 1, j - 1, k)) -
                                zxScale * AT(Jg0, i, j, k, 2) *
                                    (E(i, j, k + 1) - E(i, j, k - 1) + E(i - 1, j, k + 1) - E(i - 1, j, k - 1));
                        lapDiag = lapDiag - xxScale * AT(Jg0, i, j, k, 0);
                    }
                    if (hiX != BC_NEUM) {
                        JDhiX = +xxScale * AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n) +
                                xyScale * AT(Jg0, i + 1, j, k, 1) *
                                    (E(i + 1, j + 1, k) - E(i + 1, j - 1, k) + E(i, j + 1, k) - E(i, j - 1, k)) +
                                zxScale * AT(Jg0, i + 1, j, k, 2) *
                                    (E(i + 1, j, k + 1) - E(i + 1, j, k - 1) + E(i, j, k + 1) - E(i, j, k - 1));
                        lapDiag = lapDiag - xxScale * AT(Jg0, i + 1, j, k, 0);
                    }
                    if (loY != BC_NEUM) {
                        JDloY = -xyScale * AT(Jg1, i, j, k, 0) *
                                    (E(i + 1, j, k) - E(i - 1, j, k) + E(i + 1, j - 1, k) - E(i - 1, j - 1, k)) +
                                yyScale * AT(Jg1, i, j, k, 1) * AT(phi, i, j - 1, k, n) -
                                yzScale * AT(Jg1, i, j, k, 2) *
                                    (E(i, j, k + 1) - E(i, j, k - 1) + E(i, j - 1, k + 1) - E(i, j - 1, k - 1));
                        lapDiag = lapDiag - yyScale * AT(Jg1, i, j, k, 1);
                    }
                    if (hiY != BC_NEUM) {
                        JDhiY = +xyScale * AT(Jg1, i, j + 1, k, 0) *
                                    (E(i + 1, j + 1, k) - E(i - 1, j + 1, k) + E(i + 1, j, k) - E(i - 1, j, k)) +
                                yyScale * AT(Jg1, i, j + 1, k, 1) * AT(phi, i, j + 1, k, n) +
                                yzScale * AT(Jg1, i, j + 1, k, 2) *
                                    (E(i, j + 1, k + 1) - E(i, j + 1, k - 1) + E(i, j, k + 1) - E(i, j, k - 1));
                        lapDiag = lapDiag - yyScale * AT(Jg1, i, j + 1, k, 1);
                    }
                    if (loZ != BC_NEUM) {
                        JDloZ = -zxScale * AT(Jg2, i, j, k, 0) *
                                    (E(i + 1, j, k) - E(i - 1, j, k) + E(i + 1, j, k - 1) - E(i - 1, j, k - 1)) -
                                yzScale * AT(Jg2, i, j, k, 1) *
                                    (E(i, j + 1, k) - E(i, j - 1, k) + E(i, j + 1, k - 1) - E(i, j - 1, k - 1)) +
                                zzScale * AT(Jg2, i, j, k, 2) * AT(phi, i, j, k - 1, n);
                        lapDiag = lapDiag - zzScale * AT(Jg2, i, j, k, 2);
                    }
                    if (hiZ != BC_NEUM) {
                        JDhiZ = +zxScale * AT(Jg2, i, j, k + 1, 0) *
                                    (E(i + 1, j, k + 1) - E(i - 1, j, k + 1) + E(i + 1, j, k) - E(i - 1, j, k)) +
                                yzScale * AT(Jg2, i, j, k + 1, 1) *
                                    (E(i, j + 1, k + 1) - E(i, j - 1, k + 1) + E(i, j + 1, k) - E(i, j - 1, k)) +
                                zzScale * AT(Jg2, i, j, k + 1, 2) * AT(phi, i, j, k + 1, n);
                        lapDiag = lapDiag - zzScale * AT(Jg2, i, j, k + 1, 2);
                    }
                    lapDiag = lapDiag * AT(Jinv, i, j, k, 0);
                    double lphi = beta * AT(Jinv, i, j, k, 0) *
                                  (JDloX + JDhiX + JDloY + JDhiY + JDloZ + JDhiZ);
                    AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) / (alpha + beta * lapDiag);
                }
            }
#undef E
}

/* ------------------------------------------------------------------------
 * K6  MAPPEDGETFLUXORTHO   AMRElliptic/MappedAMRPoissonOpOrthoF.ChF:33-82
 * flux = Jgaa * beta_dx * (phi(i) - phi(i - e_a)) over the face box.
 * ---------------------------------------------------------------------- */
void orc_mappedgetfluxortho(double *flux_, const int *flo, const int *fhi, int ncomp,
                            const double *phi_, const int *plo, const int *phi_hi,
                            const double *jgaa_, const int *glo, const int *ghi,
                            const int *reglo, const int *reghi, double beta_dx, int adir)
{
    fra_t flux = mk(flux_, flo, fhi), phi = mk((double *)phi_, plo, phi_hi);
    fra_t Jgaa = mk((double *)jgaa_, glo, ghi);
    const int ai = adir == 0, aj = adir == 1, ak = adir == 2;
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(flux, i, j, k, n) = AT(Jgaa, i, j, k, 0) * beta_dx *
                                           (AT(phi, i, j, k, n) - AT(phi, i - ai, j - aj, k - ak, n));
}

/* K6  MAPPEDGETFLUX (full)   AMRElliptic/MappedAMRPoissonOpF.ChF:335-427
 * Jga has SpaceDim comps.  bdir=(adir+1)%3, cdir=(adir+2)%3.  */
void orc_mappedgetflux(double *flux_, const int *flo, const int *fhi, int ncomp,
                       const double *phi_, const int *plo, const int *phi_hi,
                       const double *ext_, const int *elo, const int *ehi,
                       const double *jga_, const int *glo, const int *ghi,
                       const int *reglo, const int *reghi, double beta, const double *dx, int adir)
{
    fra_t flux = mk(flux_, flo, fhi), phi = mk((double *)phi_, plo, phi_hi);
    fra_t extrap = mk((double *)ext_, elo, ehi), Jga = mk((double *)jga_, glo, ghi);
    const int bdir = (adir + 1) % 3, cdir = (adir + 2) % 3;
    const int ai = adir == 0, aj = adir == 1, ak = adir == 2;
    const int bi = bdir == 0, bj = bdir == 1, bk = bdir == 2;
    const int ci = cdir == 0, cj = cdir == 1, ck = cdir == 2;
    const double aScale = beta / dx[adir];
    const double bScale = 0.25 * beta / dx[bdir];
    const double cScale = 0.25 * beta / dx[cdir];
#define E(a, b, c) AT(extrap, a, b, c, n)
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(flux, i, j, k, n) =
                        aScale * AT(Jga, i, j, k, adir) *
                            (AT(phi, i, j, k, n) - AT(phi, i - ai, j - aj, k - ak, n)) +
                        bScale * AT(Jga, i, j, k, bdir) *
                            (E(i + bi, j + bj, k + bk) - E(i - bi, j - bj, k - bk) +
                             E(i + bi - ai, j + bj - aj, k + bk - ak) -
                             E(i - bi - ai, j - bj - aj, k - bk - ak)) +
                        cScale * AT(Jga, i, j, k, cdir) *
                            (E(i + ci, j + cj, k + ck) - E(i - ci, j - cj, k - ck) +
                             E(i + ci - ai, j + cj - aj, k + ck - ak) -
                             E(i - ci - ai, j - cj - aj, k - ck - ak));
#undef E
}

/* ------------------------------------------------------------------------
 * K7  MAPPEDFLUXDIVERGENCE3D   DivCurlGrad/DivCurlGradF.ChF:1122-1215
 * ---------------------------------------------------------------------- */
void orc_mappedfluxdivergence3d(double *div_, const int *dlo, const int *dhi, int ncomp,
                                const double *f0_, const int *f0lo, const int *f0hi,
                                const double *f1_, const int *f1lo, const int *f1hi,
                                const double *f2_, const int *f2lo, const int *f2hi,
                                const double *jinv_, const int *jlo, const int *jhi,
                                const int *reglo, const int *reghi, const double *dx)
{
    fra_t div = mk(div_, dlo, dhi), flux0 = mk((double *)f0_, f0lo, f0hi);
    fra_t flux1 = mk((double *)f1_, f1lo, f1hi), flux2 = mk((double *)f2_, f2lo, f2hi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi);
    const double dxinv0 = 1.0 / dx[0], dxinv1 = 1.0 / dx[1], dxinv2 = 1.0 / dx[2];
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(div, i, j, k, n) =
                        AT(Jinv, i, j, k, 0) *
                        ((AT(flux0, i + 1, j, k, n) - AT(flux0, i, j, k, n)) * dxinv0 +
                         (AT(flux1, i, j + 1, k, n) - AT(flux1, i, j, k, n)) * dxinv1 +
                         (AT(flux2, i, j, k + 1, n) - AT(flux2, i, j, k, n)) * dxinv2);
}

/* K7  MAPPEDFLUXDIVERGENCE2D   DivCurlGradF.ChF:1034-1112 */
void orc_mappedfluxdivergence2d(double *div_, const int *dlo, const int *dhi, int ncomp,
                                const double *f0_, const int *f0lo, const int *f0hi,
                                const double *f1_, const int *f1lo, const int *f1hi,
                                const double *jinv_, const int *jlo, const int *jhi,
                                const int *reglo, const int *reghi, const double *dx)
{
    fra_t div = mk(div_, dlo, dhi), flux0 = mk((double *)f0_, f0lo, f0hi);
    fra_t flux1 = mk((double *)f1_, f1lo, f1hi), Jinv = mk((double *)jinv_, jlo, jhi);
    const double dxinv0 = 1.0 / dx[0], dxinv1 = 1.0 / dx[1];
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(div, i, j, k, n) =
                        AT(Jinv, i, j, k, 0) *
                        ((AT(flux0, i + 1, j, k, n) - AT(flux0, i, j, k, n)) * dxinv0 +
                         (AT(flux1, i, j + 1, k, n) - AT(flux1, i, j, k, n)) * dxinv1);
}

/* ------------------------------------------------------------------------
 * K8  SUBTRACTOP / AXBYIP / DIAGPRECOND   MappedAMRPoissonOpF.ChF:36-57, 62-85,
 * 284-328;  JACOBIITER  RelaxationMethods/JacobiF.ChF
 * ---------------------------------------------------------------------- */
void orc_subtractop(double *res_, const int *slo, const int *shi, int ncomp,
                    const double *a1_, const int *alo, const int *ahi,
                    const double *a2_, const int *blo, const int *bhi,
                    const int *reglo, const int *reghi)
{
    fra_t res = mk(res_, slo, shi), a1 = mk((double *)a1_, alo, ahi), a2 = mk((double *)a2_, blo, bhi);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(res, i, j, k, n) = AT(a1, i, j, k, n) - AT(a2, i, j, k, n);
}

void orc_axbyip(double *lhs_, const int *llo, const int *lhi, int ncomp,
                const double *phi_, const int *plo, const int *phi_hi,
                double alpha, double beta, const int *reglo, const int *reghi)
{
    fra_t lhs = mk(lhs_, llo, lhi), phi = mk((double *)phi_, plo, phi_hi);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(lhs, i, j, k, n) = alpha * AT(phi, i, j, k, n) + beta * AT(lhs, i, j, k, n);
}

void orc_diagprecond(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                     const double *rhs_, const int *rlo, const int *rhi,
                     const double *lapd_, const int *dlo, const int *dhi,
                     const int *reglo, const int *reghi, double alpha, double beta)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi);
    fra_t lapDiag = mk((double *)lapd_, dlo, dhi);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(phi, i, j, k, n) = AT(rhs, i, j, k, n) / (alpha + beta * AT(lapDiag, i, j, k, 0));
}

void orc_jacobiiter(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                    const double *res_, const int *rlo, const int *rhi,
                    const double *lapd_, const int *dlo, const int *dhi,
                    const int *reglo, const int *reghi, double alpha, double beta)
{
    fra_t phi = mk(phi_, plo, phi_hi), res = mk((double *)res_, rlo, rhi);
    fra_t lapDiag = mk((double *)lapd_, dlo, dhi);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(phi, i, j, k, n) = AT(phi, i, j, k, n) +
                                          0.5 * AT(res, i, j, k, n) / (alpha + beta * AT(lapDiag, i, j, k, 0));
}

/* ------------------------------------------------------------------------
 * K9  FILLMAPPEDLAPDIAG3D / 2D   MappedAMRPoissonOpF.ChF:215-274, 149-211
 * Jg0,Jg1,Jg2 carry SpaceDim comps; comp a of FAB a is used.
 * ---------------------------------------------------------------------- */
void orc_fillmappedlapdiag3d(double *lap_, const int *llo, const int *lhi,
                             const double *jg0_, const int *xlo, const int *xhi,
                             const double *jg1_, const int *ylo, const int *yhi,
                             const double *jg2_, const int *zlo, const int *zhi,
                             const double *jinv_, const int *jlo, const int *jhi,
                             const int *reglo, const int *reghi, const double *dx)
{
    fra_t lap = mk(lap_, llo, lhi), Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jg2 = mk((double *)jg2_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    const double s0 = 1.0 / (dx[0] * dx[0]), s1 = 1.0 / (dx[1] * dx[1]), s2 = 1.0 / (dx[2] * dx[2]);
    for (int k = reglo[2]; k <= reghi[2]; ++k)
        for (int j = reglo[1]; j <= reghi[1]; ++j)
            for (int i = reglo[0]; i <= reghi[0]; ++i)
                AT(lap, i, j, k, 0) = -AT(Jinv, i, j, k, 0) *
                                      ((AT(Jg0, i + 1, j, k, 0) + AT(Jg0, i, j, k, 0)) * s0 +
                                       (AT(Jg1, i, j + 1, k, 1) + AT(Jg1, i, j, k, 1)) * s1 +
                                       (AT(Jg2, i, j, k + 1, 2) + AT(Jg2, i, j, k, 2)) * s2);
}

void orc_fillmappedlapdiag2d(double *lap_, const int *llo, const int *lhi,
                             const double *jg0_, const int *xlo, const int *xhi,
                             const double *jg1_, const int *ylo, const int *yhi,
                             const double *jinv_, const int *jlo, const int *jhi,
                             const int *reglo, const int *reghi, const double *dx)
{
    fra_t lap = mk(lap_, llo, lhi), Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi);
    const double s0 = 1.0 / (dx[0] * dx[0]), s1 = 1.0 / (dx[1] * dx[1]);
    for (int k = reglo[2]; k <= reghi[2]; ++k)
        for (int j = reglo[1]; j <= reghi[1]; ++j)
            for (int i = reglo[0]; i <= reghi[0]; ++i)
                AT(lap, i, j, k, 0) = -AT(Jinv, i, j, k, 0) *
                                      ((AT(Jg0, i + 1, j, k, 0) + AT(Jg0, i, j, k, 0)) * s0 +
                                       (AT(Jg1, i, j + 1, k, 1) + AT(Jg1, i, j, k, 1)) * s1);
}

/* ------------------------------------------------------------------------
 * K10 MAPPEDAVERAGE2 / UNMAPPEDAVERAGEHARMONIC / UNMAPPEDAVERAGEFACE
 *     MappedChombo/MappedCoarseAverageF.ChF:132-167, 48-82, 182-221
 * `box` is the coarse box; bref loop = ii2 outer .. ii0 inner.
 * ---------------------------------------------------------------------- */
void orc_mappedaverage2(double *crse_, const int *clo, const int *chi, int ncomp,
                        const double *fine_, const int *flo, const int *fhi,
                        const double *fjinv_, const int *jlo, const int *jhi,
                        const int *boxlo, const int *boxhi, const int *refRatio)
{
    fra_t coarse = mk(crse_, clo, chi), fine = mk((double *)fine_, flo, fhi);
    fra_t fineCCJinv = mk((double *)fjinv_, jlo, jhi);
    for (int var = 0; var < ncomp; ++var)
        for (int ic2 = boxlo[2]; ic2 <= boxhi[2]; ++ic2)
            for (int ic1 = boxlo[1]; ic1 <= boxhi[1]; ++ic1)
                for (int ic0 = boxlo[0]; ic0 <= boxhi[0]; ++ic0) {
                    int ip0 = ic0 * refRatio[0], ip1 = ic1 * refRatio[1], ip2 = ic2 * refRatio[2];
                    double coarseSum = 0.0, coarseCCJSum = 0.0;
                    for (int ii2 = 0; ii2 < refRatio[2]; ++ii2)
                        for (int ii1 = 0; ii1 < refRatio[1]; ++ii1)
                            for (int ii0 = 0; ii0 < refRatio[0]; ++ii0) {
                                coarseSum = coarseSum + AT(fine, ip0 + ii0, ip1 + ii1, ip2 + ii2, var) /
                                                            AT(fineCCJinv, ip0 + ii0, ip1 + ii1, ip2 + ii2, 0);
                                coarseCCJSum = coarseCCJSum +
                                               1.0 / AT(fineCCJinv, ip0 + ii0, ip1 + ii1, ip2 + ii2, 0);
                            }
                    AT(coarse, ic0, ic1, ic2, var) = coarseSum / coarseCCJSum;
                }
}

void orc_unmappedaverageharmonic(double *crse_, const int *clo, const int *chi, int ncomp,
                                 const double *fine_, const int *flo, const int *fhi,
                                 const int *boxlo, const int *boxhi, const int *refRatio)
{
    fra_t coarse = mk(crse_, clo, chi), fine = mk((double *)fine_, flo, fhi);
    const double refScale = 1.0 / (double)(refRatio[0] * refRatio[1] * refRatio[2]);
    for (int var = 0; var < ncomp; ++var)
        for (int ic2 = boxlo[2]; ic2 <= boxhi[2]; ++ic2)
            for (int ic1 = boxlo[1]; ic1 <= boxhi[1]; ++ic1)
                for (int ic0 = boxlo[0]; ic0 <= boxhi[0]; ++ic0) {
                    int ip0 = ic0 * refRatio[0], ip1 = ic1 * refRatio[1], ip2 = ic2 * refRatio[2];
                    double coarseSum = 0.0;
                    for (int ii2 = 0; ii2 < refRatio[2]; ++ii2)
                        for (int ii1 = 0; ii1 < refRatio[1]; ++ii1)
                            for (int ii0 = 0; ii0 < refRatio[0]; ++ii0)
                                coarseSum = coarseSum + 1.0 / AT(fine, ip0 + ii0, ip1 + ii1, ip2 + ii2, var);
                    AT(coarse, ic0, ic1, ic2, var) = 1.0 / (coarseSum * refScale);
                }
}

/* refBox for face averaging is flat in `dir` (MappedCoarseAverageFace:
 * refbox.setBig(dir,0)), i.e. only fine faces lying ON the coarse face. */
void orc_unmappedaverageface(double *crse_, const int *clo, const int *chi, int ncomp,
                             const double *fine_, const int *flo, const int *fhi,
                             const int *boxlo, const int *boxhi, int dir, const int *refRatio)
{
    fra_t coarse = mk(crse_, clo, chi), fine = mk((double *)fine_, flo, fhi);
    const double refScale = (double)refRatio[dir] / (double)(refRatio[0] * refRatio[1] * refRatio[2]);
    int rb[3] = {refRatio[0], refRatio[1], refRatio[2]};
    rb[dir] = 1;
    for (int var = 0; var < ncomp; ++var)
        for (int ic2 = boxlo[2]; ic2 <= boxhi[2]; ++ic2)
            for (int ic1 = boxlo[1]; ic1 <= boxhi[1]; ++ic1)
                for (int ic0 = boxlo[0]; ic0 <= boxhi[0]; ++ic0) {
                    int ip0 = ic0 * refRatio[0], ip1 = ic1 * refRatio[1], ip2 = ic2 * refRatio[2];
                    double crseSum = 0.0;
                    for (int ii2 = 0; ii2 < rb[2]; ++ii2)
                        for (int ii1 = 0; ii1 < rb[1]; ++ii1)
                            for (int ii0 = 0; ii0 < rb[0]; ++ii0)
                                crseSum = crseSum + AT(fine, ip0 + ii0, ip1 + ii1, ip2 + ii2, var);
                    AT(coarse, ic0, ic1, ic2, var) = refScale * crseSum;
                }
}

/* ------------------------------------------------------------------------
 * K11 ConstInterpPS / ConstInterpWithAvgPS
 *     MGStrategies/ProlongationStrategyF.ChF:36-85, 97-160
 * The caller shifts fine and coarse boxes so fineRegion.lo == 0 (C++ side,
 * ProlongationStrategy.cpp:71-81), hence plain integer division i/m.
 * ---------------------------------------------------------------------- */
void orc_constinterpps(double *fine_, const int *flo, const int *fhi, int ncomp,
                       const double *crse_, const int *clo, const int *chi,
                       const int *reglo, const int *reghi, const int *m)
{
    fra_t fine = mk(fine_, flo, fhi), coarse = mk((double *)crse_, clo, chi);
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k) {
            int kk = k / m[2];
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int jj = j / m[1];
                for (int i = reglo[0]; i <= reghi[0]; ++i) {
                    int ii = i / m[0];
                    AT(fine, i, j, k, n) = AT(fine, i, j, k, n) + AT(coarse, ii, jj, kk, n);
                }
            }
        }
}

void orc_constinterpwithavgps(double *fine_, const int *flo, const int *fhi, int ncomp,
                              const double *crse_, const int *clo, const int *chi,
                              const int *reglo, const int *reghi, const int *m,
                              const double *jinv_, const int *jlo, const int *jhi,
                              double dxProduct, double *vol, double *sum)
{
    fra_t fine = mk(fine_, flo, fhi), coarse = mk((double *)crse_, clo, chi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi);
    double v = *vol, s = *sum;
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k) {
            int kk = k / m[2];
            for (int j = reglo[1]; j <= reghi[1]; ++j) {
                int jj = j / m[1];
                for (int i = reglo[0]; i <= reghi[0]; ++i) {
                    int ii = i / m[0];
                    AT(fine, i, j, k, n) = AT(fine, i, j, k, n) + AT(coarse, ii, jj, kk, n);
                    double dvol = dxProduct / AT(Jinv, i, j, k, 0);
                    s = s + dvol * AT(fine, i, j, k, n);
                    v = v + dvol;
                }
            }
        }
    *vol = v;
    *sum = s;
}

/* ------------------------------------------------------------------------
 * K12 EXTRAPOLATEFACENOEV   extrapolation/ExtrapolationUtilsF.ChF:35-138
 * ---------------------------------------------------------------------- */
int orc_extrapolatefacenoev(double *dest_, const int *dlo, const int *dhi, int ncomp,
                            const double *src_, const int *slo, const int *shi,
                            const int *reglo, const int *reghi, int dir, int sidesign, int order)
{
    fra_t dest = mk(dest_, dlo, dhi), src = mk((double *)src_, slo, shi);
    const int ii = sidesign * (dir == 0), jj = sidesign * (dir == 1), kk = sidesign * (dir == 2);
    if (order < 0 || order > 2) return -1;
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i) {
                    if (order == 0)
                        AT(dest, i, j, k, n) = AT(src, i - ii, j - jj, k - kk, n);
                    else if (order == 1)
                        AT(dest, i, j, k, n) = 2.0 * AT(src, i - ii, j - jj, k - kk, n) -
                                               AT(src, i - 2 * ii, j - 2 * jj, k - 2 * kk, n);
                    else
                        AT(dest, i, j, k, n) = 3.0 * (AT(src, i - ii, j - jj, k - kk, n) -
                                                      AT(src, i - 2 * ii, j - 2 * jj, k - 2 * kk, n)) +
                                               AT(src, i - 3 * ii, j - 3 * jj, k - 3 * kk, n);
                }
    return 0;
}

/* ------------------------------------------------------------------------
 * K13 ELLIPTICCONSTNEUMBCGHOSTORTHO / ELLIPTICCONSTNEUMBCGHOST
 *     BCInterface/EllipticBCUtilsF.ChF:415-455, 293-412
 * fsign = +1 on the high side (ghost = valid + e_dir), -1 on the low side.
 * ---------------------------------------------------------------------- */
void orc_ellipticconstneumbcghostortho(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                                       const double *nhat_, const int *nlo, const int *nhi,
                                       const int *glo, const int *ghi, double bcval, int fdir,
                                       int fsign, double dxDir)
{
    fra_t phi = mk(phi_, plo, phi_hi), nhat = mk((double *)nhat_, nlo, nhi);
    const double factor = bcval * dxDir;
    const int ai = fsign * (fdir == 0), aj = fsign * (fdir == 1), ak = fsign * (fdir == 2);
    const int foffset = (1 - fsign) / 2;
    const int fi = foffset * (fdir == 0), fj = foffset * (fdir == 1), fk = foffset * (fdir == 2);
    for (int n = 0; n < ncomp; ++n)
        for (int k = glo[2]; k <= ghi[2]; ++k)
            for (int j = glo[1]; j <= ghi[1]; ++j)
                for (int i = glo[0]; i <= ghi[0]; ++i)
                    AT(phi, i, j, k, n) = AT(phi, i - ai, j - aj, k - ak, n) +
                                          factor / AT(nhat, i + fi, j + fj, k + fk, 0);
}

void orc_ellipticconstneumbcghost(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                                  const double *ext_, const int *elo, const int *ehi,
                                  const double *nhat_, const int *nlo, const int *nhi,
                                  const int *glo, const int *ghi, double bcval, int fdir,
                                  int fsign, const double *dx)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t nhat = mk((double *)nhat_, nlo, nhi);
    const int adir = fdir, bdir = (fdir + 1) % 3, cdir = (fdir + 2) % 3;
    const int bi = bdir == 0, bj = bdir == 1, bk = bdir == 2;
    const int ci = cdir == 0, cj = cdir == 1, ck = cdir == 2;
    const int foffset = (1 - fsign) / 2;
    const int fio = foffset * (adir == 0), fjo = foffset * (adir == 1), fko = foffset * (adir == 2);
    const int vio = -fsign * (adir == 0), vjo = -fsign * (adir == 1), vko = -fsign * (adir == 2);
    const double idxb = -0.25 / dx[bdir], idxc = -0.25 / dx[cdir];
#define E(a, b, c) AT(extrap, a, b, c, n)
    for (int n = 0; n < ncomp; ++n)
        for (int gk = glo[2]; gk <= ghi[2]; ++gk)
            for (int gj = glo[1]; gj <= ghi[1]; ++gj)
                for (int gi = glo[0]; gi <= ghi[0]; ++gi) {
                    int fi = gi + fio, fj = gj + fjo, fk = gk + fko;
                    int vi = gi + vio, vj = gj + vjo, vk = gk + vko;
                    double cross =
                        (E(gi + bi, gj + bj, gk + bk) - E(gi - bi, gj - bj, gk - bk) +
                         E(vi + bi, vj + bj, vk + bk) - E(vi - bi, vj - bj, vk - bk)) *
                            AT(nhat, fi, fj, fk, bdir) * idxb +
                        (E(gi + ci, gj + cj, gk + ck) - E(gi - ci, gj - cj, gk - ck) +
                         E(vi + ci, vj + cj, vk + ck) - E(vi - ci, vj - cj, vk - ck)) *
                            AT(nhat, fi, fj, fk, cdir) * idxc;
                    AT(phi, gi, gj, gk, n) = AT(phi, vi, vj, vk, n) +
                                             (bcval - cross) * dx[adir] / AT(nhat, fi, fj, fk, adir);
                }
#undef E
}

/* ------------------------------------------------------------------------
 * utils/ThomasAlgorithm.f90:37-68  solve_tridiag (general tridiagonal, no
 * pivoting).  a = sub (n-1), b = diag (n), c = super (n-1), d = rhs.
 * Validated against oracle/_ref/libthomas_ref.so (flang build of that file).
 * ---------------------------------------------------------------------- */
void orc_solve_tridiag(const double *a, const double *b, const double *c, const double *d,
                       double *x, int n)
{
    double *cp = (double *)malloc(sizeof(double) * (size_t)n);
    double *dp = (double *)malloc(sizeof(double) * (size_t)n);
    double m;
    cp[0] = c[0] / b[0];
    dp[0] = d[0] / b[0];
    for (int i = 1; i < n - 1; ++i) {
        m = b[i] - cp[i - 1] * a[i - 1];
        cp[i] = c[i] / m;
        dp[i] = (d[i] - dp[i - 1] * a[i - 1]) / m;
    }
    int i = n - 1;
    m = b[i] - cp[i - 1] * a[i - 1];
    dp[i] = (d[i] - dp[i - 1] * a[i - 1]) / m;
    x[n - 1] = dp[n - 1];
    for (i = n - 2; i >= 0; --i) x[i] = dp[i] - cp[i] * x[i + 1];
    free(cp);
    free(dp);
}

/* ------------------------------------------------------------------------
 * K14 MAPPEDMACGRADORTHO   DivCurlGrad/DivCurlGradF.ChF:221-285
 * normal branch (gradDir == edgeDir): edgeGrad = dxinv * Jga(:,gradDir) * (phi(i) - phi(i - e))
 * transverse branch: edgeGrad = (0.25/dx) * Jga(:,gradDir) * centred differences of `extrap`
 * ---------------------------------------------------------------------- */
void orc_mappedmacgradortho(double *eg_, const int *elo, const int *ehi,
                            const double *phi_, const int *plo, const int *phi_hi,
                            const double *ext_, const int *xlo, const int *xhi,
                            const double *jga_, const int *glo, const int *ghi,
                            const int *reglo, const int *reghi, double dxDir, int gradDir, int edgeDir)
{
    fra_t edgeGrad = mk(eg_, elo, ehi), phi = mk((double *)phi_, plo, phi_hi);
    fra_t extrap = mk((double *)ext_, xlo, xhi), Jga = mk((double *)jga_, glo, ghi);
    const int gi = gradDir == 0, gj = gradDir == 1, gk = gradDir == 2;
    if (gradDir == edgeDir) {
        const double dxinv = 1.0 / dxDir;
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(edgeGrad, i, j, k, 0) = dxinv * AT(Jga, i, j, k, gradDir) *
                                               (AT(phi, i, j, k, 0) - AT(phi, i - gi, j - gj, k - gk, 0));
    } else {
        const int ei = edgeDir == 0, ej = edgeDir == 1, ek = edgeDir == 2;
        const double dxinv = 0.25 / dxDir;
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i)
                    AT(edgeGrad, i, j, k, 0) =
                        dxinv * AT(Jga, i, j, k, gradDir) *
                        (AT(extrap, i + gi, j + gj, k + gk, 0) - AT(extrap, i - gi, j - gj, k - gk, 0) +
                         AT(extrap, i + gi - ei, j + gj - ej, k + gk - ek, 0) -
                         AT(extrap, i - gi - ei, j - gj - ej, k - gk - ek, 0));
    }
}

/* ELLIPTICEXTRAPBCGHOST   BCInterface/EllipticBCUtilsF.ChF:115-205 (orders 0..2).
 * sidesign = +1: high side (ghost = f(ghost-1, ghost-2, ...)), -1: low side. */
int orc_ellipticextrapbcghost(double *st_, const int *slo, const int *shi, int ncomp,
                              const int *reglo, const int *reghi, int dir, int sidesign, int order)
{
    fra_t state = mk(st_, slo, shi);
    int ii0 = dir == 0, ii1 = dir == 1, ii2 = dir == 2;
    if (sidesign == 1) { ii0 = -ii0; ii1 = -ii1; ii2 = -ii2; }
    if (order < 0 || order > 2) return -1;
    for (int n = 0; n < ncomp; ++n)
        for (int k = reglo[2]; k <= reghi[2]; ++k)
            for (int j = reglo[1]; j <= reghi[1]; ++j)
                for (int i = reglo[0]; i <= reghi[0]; ++i) {
                    if (order == 0)
                        AT(state, i, j, k, n) = AT(state, i + ii0, j + ii1, k + ii2, n);
                    else if (order == 1)
                        AT(state, i, j, k, n) = 2.0 * AT(state, i + ii0, j + ii1, k + ii2, n) -
                                                AT(state, i + 2 * ii0, j + 2 * ii1, k + 2 * ii2, n);
                    else
                        AT(state, i, j, k, n) = 3.0 * (AT(state, i + ii0, j + ii1, k + ii2, n) -
                                                       AT(state, i + 2 * ii0, j + 2 * ii1, k + 2 * ii2, n)) +
                                                AT(state, i + 3 * ii0, j + 3 * ii1, k + 3 * ii2, n);
                }
    return 0;
}

/* ------------------------------------------------------------------------
 * LAPACK dgtsv (EXTERNAL, reference LAPACK 3.x dgtsv.f, NRHS = 1), restated for the case the path
 * produces: a diagonally dominant system, for which the partial-pivoting test |d(i)| >= |dl(i)| always
 * holds and no row interchange happens.  Returns INFO (0 ok, i > 0: zero pivot at i, -1: an interchange
 * would have been needed -- never seen on this path).  dl, d, du, b are overwritten like LAPACK does.
 * Call sites: RelaxationMethods/GSRBF.ChF:1709, 2022.
 * ---------------------------------------------------------------------- */
int orc_dgtsv_nopivot(int n, double *dl, double *d, double *du, double *b)
{
    if (n <= 0) return 0;
    for (int i = 0; i < n - 1; ++i) {
        if (fabs(d[i]) >= fabs(dl[i])) {
            if (d[i] == 0.0) return i + 1;
            double fact = dl[i] / d[i];
            d[i + 1] = d[i + 1] - fact * du[i];
            b[i + 1] = b[i + 1] - fact * b[i];
            dl[i] = 0.0;
        } else {
            return -1;
        }
    }
    if (d[n - 1] == 0.0) return n;
    b[n - 1] = b[n - 1] / d[n - 1];
    if (n > 1) b[n - 2] = (b[n - 2] - du[n - 2] * b[n - 1]) / d[n - 2];
    for (int i = n - 3; i >= 0; --i) b[i] = (b[i] - du[i] * b[i + 1] - dl[i] * b[i + 2]) / d[i];
    return 0;
}

/* ------------------------------------------------------------------------
 * K5  LineGSRBIter3D   RelaxationMethods/GSRBF.ChF:1730-2042
 * Vertical-line Gauss-Seidel, one colour: per (i,j) column assemble the tridiagonal system (equation
 * scaled by J: B = -lphi + rhs/Jinv, D = alpha/Jinv + lapDiag, DL_k = DU_k = Jg2^{22}_{k+1} zzScale) with the
 * horizontal and cross terms lagged, solve with dgtsv, overwrite the column.
 * DEVIATION from the reference: the 3-D routine reads uninitialised jmin/jmax
 * (:1764 vs :1831) and compares i with the parity-shifted imin (:1793-1797); as in the 2-D routine
 * (:1607-1611) the INTENDED test is used here: a cell drops the flux through a face iff it sits on the
 * region bound in that direction and that side's BC code is Neumann.
 * bc = {loX,hiX,loY,hiY,loZ,hiZ}.  Jg0/Jg1/Jg2 carry 3 comps.  Returns the worst dgtsv INFO.
 * ---------------------------------------------------------------------- */
int orc_linegsrbiter3d(double *phi_, const int *plo, const int *phi_hi,
                       const double *ext_, const int *elo, const int *ehi,
                       const double *rhs_, const int *rlo, const int *rhi,
                       const double *jg0_, const int *xlo, const int *xhi,
                       const double *jg1_, const int *ylo, const int *yhi,
                       const double *jg2_, const int *zlo, const int *zhi,
                       const double *jinv_, const int *jlo, const int *jhi,
                       const int *reglo, const int *reghi, const double *dx, double dzCrse,
                       double alpha, double beta, int redBlack, const int *bc)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jg2 = mk((double *)jg2_, zlo, zhi), Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = bc[0], hiX = bc[1], loY = bc[2], hiY = bc[3], loZ = bc[4], hiZ = bc[5];
    const int k0 = reglo[2], kmax = reghi[2], N = kmax - k0 + 1;
    const double xxScale = beta * 1.0 / (dx[0] * dx[0]);
    const double yyScale = beta * 1.0 / (dx[1] * dx[1]);
    const double zzScale = beta * 1.0 / (dx[2] * dx[2]);
    const double xyScale = beta * 0.25 / (dx[0] * dx[1]);
    const double yzScale = beta * 0.25 / (dx[1] * dx[2]);
    const double zxScale = beta * 0.25 / (dx[2] * dx[0]);
    double *D = (double *)malloc(sizeof(double) * (size_t)N), *B = (double *)malloc(sizeof(double) * (size_t)N);
    double *DL = (double *)malloc(sizeof(double) * (size_t)N), *DU = (double *)malloc(sizeof(double) * (size_t)N);
    int worst = 0;
#define E(a, b, c) AT(extrap, a, b, c, 0)
    for (int j = reglo[1]; j <= reghi[1]; ++j) {
        int imin = reglo[0] + abs((reglo[0] + j + redBlack) % 2);
        for (int i = imin; i <= reghi[0]; i += 2) {
            for (int k = k0; k <= kmax; ++k) {
                const int kdx = k - k0;
                double coeff1, lapDiag;
                if (k == k0) {
                    coeff1 = (loZ == BC_NEUM) ? 0.0 : (loZ == 1 /*Diri*/ ? 2.0 : (loZ == 3 /*CF*/ ? 2.0 * dx[2] / (dzCrse + dx[2]) : 0.0));
                    lapDiag = -zzScale * (AT(Jg2, i, j, k + 1, 2) + coeff1 * AT(Jg2, i, j, k, 2));
                    if (N == 1) {  /* degenerate single-cell column: both ends */
                        double c2 = (hiZ == BC_NEUM) ? 0.0 : (hiZ == 1 ? 2.0 : (hiZ == 3 ? 2.0 * dx[2] / (dzCrse + dx[2]) : 0.0));
                        lapDiag = -zzScale * (c2 * AT(Jg2, i, j, k + 1, 2) + coeff1 * AT(Jg2, i, j, k, 2));
                    }
                } else if (k == kmax) {
                    coeff1 = (hiZ == BC_NEUM) ? 0.0 : (hiZ == 1 ? 2.0 : (hiZ == 3 ? 2.0 * dx[2] / (dzCrse + dx[2]) : 0.0));
                    lapDiag = -zzScale * (coeff1 * AT(Jg2, i, j, k + 1, 2) + AT(Jg2, i, j, k, 2));
                } else {
                    lapDiag = -zzScale * (AT(Jg2, i, j, k, 2) + AT(Jg2, i, j, k + 1, 2));
                }
                double JDxx = 0.0, JDyy = 0.0;
                if ((loX != BC_NEUM) || (i != reglo[0])) {
                    JDxx = JDxx + AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, 0);
                    lapDiag = lapDiag - xxScale * AT(Jg0, i, j, k, 0);
                }
                if ((hiX != BC_NEUM) || (i != reghi[0])) {
                    JDxx = JDxx + AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, 0);
                    lapDiag = lapDiag - xxScale * AT(Jg0, i + 1, j, k, 0);
                }
                if ((loY != BC_NEUM) || (j != reglo[1])) {
                    JDyy = JDyy + AT(Jg1, i, j, k, 1) * AT(phi, i, j - 1, k, 0);
                    lapDiag = lapDiag - yyScale * AT(Jg1, i, j, k, 1);
                }
                if ((hiY != BC_NEUM) || (j != reghi[1])) {
                    JDyy = JDyy + AT(Jg1, i, j + 1, k, 1) * AT(phi, i, j + 1, k, 0);
                    lapDiag = lapDiag - yyScale * AT(Jg1, i, j + 1, k, 1);
                }
                double JDxy = AT(Jg0, i + 1, j, k, 1) * (E(i + 1, j + 1, k) - E(i + 1, j - 1, k) + E(i, j + 1, k) - E(i, j - 1, k)) -
                              AT(Jg0, i, j, k, 1) * (E(i, j + 1, k) - E(i, j - 1, k) + E(i - 1, j + 1, k) - E(i - 1, j - 1, k));
                double JDxz = AT(Jg0, i + 1, j, k, 2) * (E(i + 1, j, k + 1) - E(i + 1, j, k - 1) + E(i, j, k + 1) - E(i, j, k - 1)) -
                              AT(Jg0, i, j, k, 2) * (E(i, j, k + 1) - E(i, j, k - 1) + E(i - 1, j, k + 1) - E(i - 1, j, k - 1));
                double JDyx = AT(Jg1, i, j + 1, k, 0) * (E(i + 1, j + 1, k) - E(i - 1, j + 1, k) + E(i + 1, j, k) - E(i - 1, j, k)) -
                              AT(Jg1, i, j, k, 0) * (E(i + 1, j, k) - E(i - 1, j, k) + E(i + 1, j - 1, k) - E(i - 1, j - 1, k));
                double JDyz = AT(Jg1, i, j + 1, k, 2) * (E(i, j + 1, k + 1) - E(i, j + 1, k - 1) + E(i, j, k + 1) - E(i, j, k - 1)) -
                              AT(Jg1, i, j, k, 2) * (E(i, j, k + 1) - E(i, j, k - 1) + E(i, j - 1, k + 1) - E(i, j - 1, k - 1));
                double JDzx = AT(Jg2, i, j, k + 1, 0) * (E(i + 1, j, k + 1) - E(i - 1, j, k + 1) + E(i + 1, j, k) - E(i - 1, j, k)) -
                              AT(Jg2, i, j, k, 0) * (E(i + 1, j, k) - E(i - 1, j, k) + E(i + 1, j, k - 1) - E(i - 1, j, k - 1));
                double JDzy = AT(Jg2, i, j, k + 1, 1) * (E(i, j + 1, k + 1) - E(i, j - 1, k + 1) + E(i, j + 1, k) - E(i, j - 1, k)) -
                              AT(Jg2, i, j, k, 1) * (E(i, j + 1, k) - E(i, j - 1, k) + E(i, j + 1, k - 1) - E(i, j - 1, k - 1));
                double lphi = JDxx * xxScale + JDyy * yyScale + (JDyz + JDzy) * yzScale + (JDzx + JDxz) * zxScale +
                              (JDxy + JDyx) * xyScale;
                B[kdx] = -lphi + AT(rhs, i, j, k, 0) / AT(Jinv, i, j, k, 0);
                D[kdx] = alpha / AT(Jinv, i, j, k, 0) + lapDiag;
                if (k < kmax) DL[kdx] = AT(Jg2, i, j, k + 1, 2) * zzScale;
            }
            for (int q = 0; q < N - 1; ++q) DU[q] = DL[q];
            int info = orc_dgtsv_nopivot(N, DL, D, DU, B);
            if (info != 0 && info != N) { if (worst == 0) worst = info; }
            for (int k = k0; k <= kmax; ++k) AT(phi, i, j, k, 0) = B[k - k0];
        }
    }
#undef E
    free(D); free(B); free(DL); free(DU);
    return worst;
}

/* ------------------------------------------------------------------------
 * K5  LineGSRBIter2D   RelaxationMethods/GSRBF.ChF:1529-1724  (CH_SPACEDIM = 2: the vertical is direction 1)
 * One colour (columns i with i + redBlack even): per column assemble B = -lphi + rhs/Jinv, D = alpha/Jinv + lapDiag,
 * DL_j = DU_j = Jg1^{11}_{j+1} yyScale with the x and cross terms lagged, dgtsv, overwrite the column.  The x neighbour of
 * a column on the REGION bound is dropped iff that side's BC code is Neumann (region bounds, as written, :1607-1611).
 * The region must start at the vertical index 0 (:1561-1565).  bc = {loX, hiX, loY, hiY}; Jg0 / Jg1 carry 2 comps.
 * Arrays are the 3-D FRAs of this oracle with one cell in the third direction (index k0).
 * ---------------------------------------------------------------------- */
int orc_linegsrbiter2d(double *phi_, const int *plo, const int *phi_hi,
                       const double *ext_, const int *elo, const int *ehi,
                       const double *rhs_, const int *rlo, const int *rhi,
                       const double *jg0_, const int *xlo, const int *xhi,
                       const double *jg1_, const int *ylo, const int *yhi,
                       const double *jinv_, const int *jlo, const int *jhi,
                       const int *reglo, const int *reghi, const double *dx, double dzCrse,
                       double alpha, double beta, int redBlack, const int *bc)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi), Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = bc[0], hiX = bc[1], loY = bc[2], hiY = bc[3];
    const int k = reglo[2];
    if (reglo[1] != 0) return -2;   /* 'LineGSRBIter2D: region must have a vertical lower bound of zero' */
    const int jmax = reghi[1], N = jmax + 1;
    const double xxScale = beta * 1.0 / (dx[0] * dx[0]);
    const double yyScale = beta * 1.0 / (dx[1] * dx[1]);
    const double xyScale = beta * 0.25 / (dx[0] * dx[1]);
    double *D = (double *)malloc(sizeof(double) * (size_t)N), *B = (double *)malloc(sizeof(double) * (size_t)N);
    double *DL = (double *)malloc(sizeof(double) * (size_t)N), *DU = (double *)malloc(sizeof(double) * (size_t)N);
    int worst = 0;
#define E(a, b) AT(extrap, a, b, k, 0)
    int imin = reglo[0] + abs((reglo[0] + redBlack) % 2);
    for (int i = imin; i <= reghi[0]; i += 2) {
        for (int j = 0; j <= jmax; ++j) {
            double coeff1, lapDiag;
            if (j == 0) {
                coeff1 = (loY == BC_NEUM) ? 0.0 : (loY == 1 /*Diri*/ ? 2.0 : (loY == 3 /*CF*/ ? 2.0 * dx[1] / (dzCrse + dx[1]) : 0.0));
                lapDiag = -yyScale * (AT(Jg1, i, j + 1, k, 1) + coeff1 * AT(Jg1, i, j, k, 1));
            } else if (j == jmax) {
                coeff1 = (hiY == BC_NEUM) ? 0.0 : (hiY == 1 ? 2.0 : (hiY == 3 ? 2.0 * dx[1] / (dzCrse + dx[1]) : 0.0));
                lapDiag = -yyScale * (coeff1 * AT(Jg1, i, j + 1, k, 1) + AT(Jg1, i, j, k, 1));
            } else {
                lapDiag = -yyScale * (AT(Jg1, i, j, k, 1) + AT(Jg1, i, j + 1, k, 1));
            }
            double JDxx = 0.0;
            if ((loX != BC_NEUM) || (i != reglo[0])) {
                JDxx = JDxx + AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, 0);
                lapDiag = lapDiag - xxScale * AT(Jg0, i, j, k, 0);
            }
            if ((hiX != BC_NEUM) || (i != reghi[0])) {
                JDxx = JDxx + AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, 0);
                lapDiag = lapDiag - xxScale * AT(Jg0, i + 1, j, k, 0);
            }
            double JDxy = AT(Jg0, i + 1, j, k, 1) * (E(i + 1, j + 1) - E(i + 1, j - 1) + E(i, j + 1) - E(i, j - 1)) -
                          AT(Jg0, i, j, k, 1) * (E(i, j + 1) - E(i, j - 1) + E(i - 1, j + 1) - E(i - 1, j - 1));
            double JDyx = AT(Jg1, i, j + 1, k, 0) * (E(i + 1, j + 1) - E(i - 1, j + 1) + E(i + 1, j) - E(i - 1, j)) -
                          AT(Jg1, i, j, k, 0) * (E(i + 1, j) - E(i - 1, j) + E(i + 1, j - 1) - E(i - 1, j - 1));
            double lphi = JDxx * xxScale + (JDxy + JDyx) * xyScale;
            B[j] = -lphi + AT(rhs, i, j, k, 0) / AT(Jinv, i, j, k, 0);
            D[j] = alpha / AT(Jinv, i, j, k, 0) + lapDiag;
            if (j < jmax) DL[j] = AT(Jg1, i, j + 1, k, 1) * yyScale;
        }
        for (int q = 0; q < N - 1; ++q) DU[q] = DL[q];
        int info = orc_dgtsv_nopivot(N, DL, D, DU, B);
        if (info != 0 && info != N) { if (worst == 0) worst = info; }
        for (int j = 0; j <= jmax; ++j) AT(phi, i, j, k, 0) = B[j];
    }
#undef E
    free(D); free(B); free(DL); free(DU);
    return worst;
}

/* ------------------------------------------------------------------------
 * K18 TriDiagPoissonNN1DFAB   utils/TridiagUtilsF.ChF:85-166
 * Along every line that starts in bottomBox and runs Nx cells in direction dir: solves the homogeneous
 * Neumann-Neumann 1-D Poisson problem  D(sigma D phi) = rhs  (sigma face-centred in dir) with the
 * Thomas recurrence whose last row is special-cased, then removes the mean of the line.
 * Single component (the only way LevelLepticSolver.cpp:1364-1371 calls it).
 * ---------------------------------------------------------------------- */
void orc_tridiagpoissonnn1dfab(double *phi_, const int *plo, const int *phi_hi,
                               const double *rhs_, const int *rlo, const int *rhi,
                               const double *sig_, const int *slo, const int *shi,
                               const int *blo, const int *bhi, int Nx, double dx, int dir)
{
    fra_t phi = mk(phi_, plo, phi_hi), rhs = mk((double *)rhs_, rlo, rhi), sigma = mk((double *)sig_, slo, shi);
    const int ii[3] = {dir == 0, dir == 1, dir == 2};
    const double dxsq = dx * dx;
    double *x = (double *)malloc(sizeof(double) * (size_t)Nx);
    double *a = (double *)malloc(sizeof(double) * (size_t)Nx);
    double *b = (double *)malloc(sizeof(double) * (size_t)Nx);
    double *c = (double *)malloc(sizeof(double) * (size_t)Nx);
    double *gam = (double *)malloc(sizeof(double) * (size_t)Nx);
#define OFF(A, r) AT(A, i + (r) * ii[0], j + (r) * ii[1], k + (r) * ii[2], 0)
    for (int k = blo[2]; k <= bhi[2]; ++k)
        for (int j = blo[1]; j <= bhi[1]; ++j)
            for (int i = blo[0]; i <= bhi[0]; ++i) {
                double bet, avg;
                int r;
                c[0] = OFF(sigma, 1);
                a[0] = 1.2345e10;
                b[0] = -c[0];
                x[0] = OFF(rhs, 0) * dxsq;
                for (r = 1; r <= Nx - 2; ++r) {
                    a[r] = OFF(sigma, r);
                    c[r] = OFF(sigma, r + 1);
                    b[r] = -(a[r] + c[r]);
                    x[r] = OFF(rhs, r) * dxsq;
                }
                /* r == Nx-1 */
                a[Nx - 1] = OFF(sigma, r);
                b[Nx - 1] = -a[Nx - 1];
                x[Nx - 1] = OFF(rhs, r) * dxsq;

                bet = b[0];
                x[0] = x[0] / bet;
                gam[0] = c[0] / bet;
                for (r = 1; r <= Nx - 2; ++r) {
                    bet = b[r] - a[r] * gam[r - 1];
                    x[r] = (x[r] - a[r] * x[r - 1]) / bet;
                    gam[r] = c[r] / bet;
                }
                /* last index is a special case (as written in the reference: a(r-1), plain b(r)) */
                x[r] = (x[r] - a[r - 1] * x[r - 1]) / b[r];
                avg = x[r];
                for (r = Nx - 2; r >= 0; --r) {
                    x[r] = x[r] - gam[r] * x[r + 1];
                    avg = avg + x[r];
                }
                avg = avg / (double)Nx;
                for (r = 0; r <= Nx - 1; ++r) OFF(phi, r) = x[r] - avg;
            }
#undef OFF
    free(x); free(a); free(b); free(c); free(gam);
}

/* ========================================================================
 * Non-diagonal metric in 2-D (CH_SPACEDIM = 2, or a flat "horizontal" operator of a 3-D build): 9-point stencil.
 * Arrays are one cell thick in k (k = 0 on every index).
 * ====================================================================== */

/* GSRBITER2D   RelaxationMethods/GSRBF.ChF:155-281.  NOTE the cross sums are written out term by term
 * (a - b + c - d, left to right), unlike GSRBITER3D which goes through the pdx / pdy temporaries: the two round
 * differently, so the 2-D kernel is NOT the 3-D one with the z terms dropped. */
void orc_gsrbiter2d(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                    const double *ext_, const int *elo, const int *ehi,
                    const double *rhs_, const int *rlo, const int *rhi,
                    const double *jg0_, const int *xlo, const int *xhi,
                    const double *jg1_, const int *ylo, const int *yhi,
                    const double *jinv_, const int *jlo, const int *jhi,
                    const double *lapd_, const int *dlo, const int *dhi,
                    const int *reglo, const int *reghi, const double *dx,
                    double alpha, double beta, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi), lapDiag = mk((double *)lapd_, dlo, dhi);
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double xyScale = 0.25 / (dx[0] * dx[1]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const int k = reglo[2];
#define E(a, b) AT(extrap, a, b, k, n)
    for (int n = 0; n < ncomp; ++n)
        for (int j = reglo[1]; j <= reghi[1]; ++j) {
            int imin = reglo[0];
            imin = imin + abs((imin + j + redBlack) % 2);
            for (int i = imin; i <= reghi[0]; i += 2) {
                double JDxx = AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n) +
                              AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, n);
                double JDxy = AT(Jg0, i + 1, j, k, 1) * (E(i + 1, j + 1) - E(i + 1, j - 1) + E(i, j + 1) - E(i, j - 1)) -
                              AT(Jg0, i, j, k, 1) * (E(i, j + 1) - E(i, j - 1) + E(i - 1, j + 1) - E(i - 1, j - 1));
                double JDyx = AT(Jg1, i, j + 1, k, 0) * (E(i + 1, j + 1) - E(i - 1, j + 1) + E(i + 1, j) - E(i - 1, j)) -
                              AT(Jg1, i, j, k, 0) * (E(i + 1, j) - E(i - 1, j) + E(i + 1, j - 1) - E(i - 1, j - 1));
                double JDyy = AT(Jg1, i, j + 1, k, 1) * AT(phi, i, j + 1, k, n) +
                              AT(Jg1, i, j, k, 1) * AT(phi, i, j - 1, k, n);
                double lphi = beta * AT(Jinv, i, j, k, 0) * (JDxx * xxScale + JDyy * yyScale + (JDxy + JDyx) * xyScale);
                AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) / (alpha + beta * AT(lapDiag, i, j, k, 0));
            }
        }
#undef E
}

/* GSRBBOUNDARYITER2D   GSRBF.ChF:858-1022 */
void orc_gsrbboundaryiter2d(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                            const double *ext_, const int *elo, const int *ehi,
                            const double *rhs_, const int *rlo, const int *rhi,
                            const double *jg0_, const int *xlo, const int *xhi,
                            const double *jg1_, const int *ylo, const int *yhi,
                            const double *jinv_, const int *jlo, const int *jhi,
                            const int *reglo, const int *reghi, const double *dx,
                            double alpha, double beta, const int *stencil, int redBlack)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t rhs = mk((double *)rhs_, rlo, rhi);
    fra_t Jg0 = mk((double *)jg0_, xlo, xhi), Jg1 = mk((double *)jg1_, ylo, yhi);
    fra_t Jinv = mk((double *)jinv_, jlo, jhi);
    const int loX = stencil[0], hiX = stencil[1], loY = stencil[2], hiY = stencil[3];
    double JDloX = 0, JDhiX = 0, JDloY = 0, JDhiY = 0;
    const double xxScale = 1.0 / (dx[0] * dx[0]);
    const double yyScale = 1.0 / (dx[1] * dx[1]);
    const double xyScale = 0.25 / (dx[0] * dx[1]);
    const int k = reglo[2];
#define E(a, b) AT(extrap, a, b, k, n)
    for (int n = 0; n < ncomp; ++n)
        for (int j = reglo[1]; j <= reghi[1]; ++j) {
            int imin = reglo[0];
            imin = imin + abs((imin + j + redBlack) % 2);
            for (int i = imin; i <= reghi[0]; i += 2) {
                double lapDiag = 0.0;
                if (loX != BC_NEUM) {
                    JDloX = +xxScale * AT(Jg0, i, j, k, 0) * AT(phi, i - 1, j, k, n) -
                            xyScale * AT(Jg0, i, j, k, 1) * (E(i, j + 1) - E(i, j - 1) + E(i - 1, j + 1) - E(i - 1, j - 1));
                    lapDiag = lapDiag - xxScale * AT(Jg0, i, j, k, 0);
                }
                if (hiX != BC_NEUM) {
                    JDhiX = +xxScale * AT(Jg0, i + 1, j, k, 0) * AT(phi, i + 1, j, k, n) +
                            xyScale * AT(Jg0, i + 1, j, k, 1) * (E(i + 1, j + 1) - E(i + 1, j - 1) + E(i, j + 1) - E(i, j - 1));
                    lapDiag = lapDiag - xxScale * AT(Jg0, i + 1, j, k, 0);
                }
                if (loY != BC_NEUM) {
                    JDloY = -xyScale * AT(Jg1, i, j, k, 0) * (E(i + 1, j) - E(i - 1, j) + E(i + 1, j - 1) - E(i - 1, j - 1)) +
                            yyScale * AT(Jg1, i, j, k, 1) * AT(phi, i, j - 1, k, n);
                    lapDiag = lapDiag - yyScale * AT(Jg1, i, j, k, 1);
                }
                if (hiY != BC_NEUM) {
                    JDhiY = +xyScale * AT(Jg1, i, j + 1, k, 0) * (E(i + 1, j + 1) - E(i - 1, j + 1) + E(i + 1, j) - E(i - 1, j)) +
                            yyScale * AT(Jg1, i, j + 1, k, 1) * AT(phi, i, j + 1, k, n);
                    lapDiag = lapDiag - yyScale * AT(Jg1, i, j + 1, k, 1);
                }
                lapDiag = lapDiag * AT(Jinv, i, j, k, 0);
                double lphi = beta * AT(Jinv, i, j, k, 0) * (JDloX + JDhiX + JDloY + JDhiY);
                AT(phi, i, j, k, n) = (AT(rhs, i, j, k, n) - lphi) / (alpha + beta * lapDiag);
            }
        }
#undef E
}

/* MAPPEDGETFLUX with CH_SPACEDIM = 2 (MappedAMRPoissonOpF.ChF:335-427): bdir = mod(adir + 1, 2), no c term */
void orc_mappedgetflux2d(double *flux_, const int *flo, const int *fhi, int ncomp,
                         const double *phi_, const int *plo, const int *phi_hi,
                         const double *ext_, const int *elo, const int *ehi,
                         const double *jga_, const int *glo, const int *ghi,
                         const int *reglo, const int *reghi, double beta, const double *dx, int adir)
{
    fra_t flux = mk(flux_, flo, fhi), phi = mk((double *)phi_, plo, phi_hi);
    fra_t extrap = mk((double *)ext_, elo, ehi), Jga = mk((double *)jga_, glo, ghi);
    const int bdir = (adir + 1) % 2;
    const int ai = adir == 0, aj = adir == 1;
    const int bi = bdir == 0, bj = bdir == 1;
    const double aScale = beta / dx[adir];
    const double bScale = 0.25 * beta / dx[bdir];
    const int k = reglo[2];
#define E(a, b) AT(extrap, a, b, k, n)
    for (int n = 0; n < ncomp; ++n)
        for (int j = reglo[1]; j <= reghi[1]; ++j)
            for (int i = reglo[0]; i <= reghi[0]; ++i)
                AT(flux, i, j, k, n) =
                    aScale * AT(Jga, i, j, k, adir) * (AT(phi, i, j, k, n) - AT(phi, i - ai, j - aj, k, n)) +
                    bScale * AT(Jga, i, j, k, bdir) *
                        (E(i + bi, j + bj) - E(i - bi, j - bj) + E(i + bi - ai, j + bj - aj) - E(i - bi - ai, j - bj - aj));
#undef E
}

/* ELLIPTICCONSTNEUMBCGHOST with CH_SPACEDIM = 2 (BCInterface/EllipticBCUtilsF.ChF): one cross term */
void orc_ellipticconstneumbcghost2d(double *phi_, const int *plo, const int *phi_hi, int ncomp,
                                    const double *ext_, const int *elo, const int *ehi,
                                    const double *nhat_, const int *nlo, const int *nhi,
                                    const int *glo, const int *ghi, double bcval, int fdir,
                                    int fsign, const double *dx)
{
    fra_t phi = mk(phi_, plo, phi_hi), extrap = mk((double *)ext_, elo, ehi);
    fra_t nhat = mk((double *)nhat_, nlo, nhi);
    const int adir = fdir, bdir = (fdir + 1) % 2;
    const int bi = bdir == 0, bj = bdir == 1;
    const int foffset = (1 - fsign) / 2;
    const int fio = foffset * (adir == 0), fjo = foffset * (adir == 1);
    const int vio = -fsign * (adir == 0), vjo = -fsign * (adir == 1);
    const double idxb = -0.25 / dx[bdir];
    const int k = glo[2];
#define E(a, b) AT(extrap, a, b, k, n)
    for (int n = 0; n < ncomp; ++n)
        for (int gj = glo[1]; gj <= ghi[1]; ++gj)
            for (int gi = glo[0]; gi <= ghi[0]; ++gi) {
                int fi = gi + fio, fj = gj + fjo;
                int vi = gi + vio, vj = gj + vjo;
                double cross = (E(gi + bi, gj + bj) - E(gi - bi, gj - bj) + E(vi + bi, vj + bj) - E(vi - bi, vj - bj)) *
                               AT(nhat, fi, fj, k, bdir) * idxb;
                AT(phi, gi, gj, k, n) = AT(phi, vi, vj, k, n) + (bcval - cross) * dx[adir] / AT(nhat, fi, fj, k, adir);
            }
#undef E
}

/* ------------------------------------------------------------------------
 * ELLIPTICCONSTDIRIBCGHOST   BCInterface/EllipticBCUtilsF.ChF:29-110: ghost cells of a Dirichlet side.
 * order 0: bcval;  1: 2 bcval - near (-near when bcval == 0);  2: (8 bcval - 6 near + far) / 3
 * (far/3 - 2 near when bcval == 0).  near / far = first / second cell inside.
 * ---------------------------------------------------------------------- */
int orc_ellipticconstdiribcghost(double *st_, const int *slo, const int *shi, int ncomp,
                                 const int *glo, const int *ghi, double bcval, int fdir, int fsign, int order)
{
    fra_t state = mk(st_, slo, shi);
    int ii[3] = {fdir == 0, fdir == 1, fdir == 2};
    if (fsign == 1) { ii[0] = -ii[0]; ii[1] = -ii[1]; ii[2] = -ii[2]; }
    if (order < 0 || order > 2) return 1;
    for (int n = 0; n < ncomp; ++n)
        for (int k = glo[2]; k <= ghi[2]; ++k)
            for (int j = glo[1]; j <= ghi[1]; ++j)
                for (int i = glo[0]; i <= ghi[0]; ++i) {
                    if (order == 0) {
                        AT(state, i, j, k, n) = bcval;
                    } else if (order == 1) {
                        if (bcval == 0.0) AT(state, i, j, k, n) = -AT(state, i + ii[0], j + ii[1], k + ii[2], n);
                        else AT(state, i, j, k, n) = 2.0 * bcval - AT(state, i + ii[0], j + ii[1], k + ii[2], n);
                    } else {
                        if (bcval == 0.0)
                            AT(state, i, j, k, n) = (1.0 / 3.0) * AT(state, i + 2 * ii[0], j + 2 * ii[1], k + 2 * ii[2], n) -
                                                    2.0 * AT(state, i + ii[0], j + ii[1], k + ii[2], n);
                        else
                            AT(state, i, j, k, n) = (8.0 * bcval - 6.0 * AT(state, i + ii[0], j + ii[1], k + ii[2], n) +
                                                     AT(state, i + 2 * ii[0], j + 2 * ii[1], k + 2 * ii[2], n)) / 3.0;
                    }
                }
    return 0;
}

/* ------------------------------------------------------------------------
 * CRSEONESIDEGRAD   calculus/DivCurlGrad/DivCurlGradF.ChF:626-697
 * One-sided face gradients on the coarse side of a coarse-fine interface: the face between a coarse cell and the
 * region covered by the finer level takes the linear extrapolation of the two faces behind it (or a copy of the
 * one behind it), judged by the mask (MASKCOPY = 0: a cell of this level not covered by the finer one,
 * calculus/DivCurlGrad/Mask.cpp, MASKVAL.H:21).  edgeGrad is face-centred in `dir` (index i = low face of cell i).
 * ---------------------------------------------------------------------- */
void orc_crseonesidegrad(double *eg_, const int *elo, const int *ehi,
                         const int *mask_, const int *mlo, const int *mhi,
                         const int *lolo, const int *lohi, const int *hilo, const int *hihi,
                         int dir, int doLo, int doHi)
{
    fra_t eg = mk(eg_, elo, ehi);
    const long ms1 = (long)(mhi[0] - mlo[0] + 1), ms2 = ms1 * (long)(mhi[1] - mlo[1] + 1);
#define MASK(i, j, k) mask_[((long)(i) - mlo[0]) + ms1 * ((long)(j) - mlo[1]) + ms2 * ((long)(k) - mlo[2])]
    const int ii = dir == 0, jj = dir == 1, kk = dir == 2;
    if (doLo == 1)
        for (int k = lolo[2]; k <= lohi[2]; ++k)
            for (int j = lolo[1]; j <= lohi[1]; ++j)
                for (int i = lolo[0]; i <= lohi[0]; ++i) {
                    if (MASK(i - 2 * ii, j - 2 * jj, k - 2 * kk) == 0)
                        AT(eg, i, j, k, 0) = 2.0 * AT(eg, i - ii, j - jj, k - kk, 0) - AT(eg, i - 2 * ii, j - 2 * jj, k - 2 * kk, 0);
                    else if (MASK(i - ii, j - jj, k - kk) == 0)
                        AT(eg, i, j, k, 0) = AT(eg, i - ii, j - jj, k - kk, 0);
                }
    if (doHi == 1)
        for (int k = hilo[2]; k <= hihi[2]; ++k)
            for (int j = hilo[1]; j <= hihi[1]; ++j)
                for (int i = hilo[0]; i <= hihi[0]; ++i) {
                    if (MASK(i + ii, j + jj, k + kk) == 0)
                        AT(eg, i, j, k, 0) = 2.0 * AT(eg, i + ii, j + jj, k + kk, 0) - AT(eg, i + 2 * ii, j + 2 * jj, k + 2 * kk, 0);
                    else if (MASK(i, j, k) == 0)
                        AT(eg, i, j, k, 0) = AT(eg, i + ii, j + jj, k + kk, 0);
                }
#undef MASK
}

/* ------------------------------------------------------------------------
 * UNMAPPEDAVERAGE   MappedChombo/MappedCoarseAverageF.ChF:7-41  (bref loop = ii2 outer .. ii0 inner)
 * ---------------------------------------------------------------------- */
void orc_unmappedaverage(double *crse_, const int *clo, const int *chi, int ncomp,
                         const double *fine_, const int *flo, const int *fhi,
                         const int *boxlo, const int *boxhi, const int *refRatio)
{
    fra_t coarse = mk(crse_, clo, chi), fine = mk((double *)fine_, flo, fhi);
    const double refScale = 1.0 / (double)(refRatio[0] * refRatio[1] * refRatio[2]);
    for (int var = 0; var < ncomp; ++var)
        for (int ic2 = boxlo[2]; ic2 <= boxhi[2]; ++ic2)
            for (int ic1 = boxlo[1]; ic1 <= boxhi[1]; ++ic1)
                for (int ic0 = boxlo[0]; ic0 <= boxhi[0]; ++ic0) {
                    const int ip0 = ic0 * refRatio[0], ip1 = ic1 * refRatio[1], ip2 = ic2 * refRatio[2];
                    double coarseSum = 0.0;
                    for (int ii2 = 0; ii2 < refRatio[2]; ++ii2)
                        for (int ii1 = 0; ii1 < refRatio[1]; ++ii1)
                            for (int ii0 = 0; ii0 < refRatio[0]; ++ii0)
                                coarseSum = coarseSum + AT(fine, ip0 + ii0, ip1 + ii1, ip2 + ii2, var);
                    AT(coarse, ic0, ic1, ic2, var) = coarseSum * refScale;
                }
}

/* ------------------------------------------------------------------------
 * LEPTICVERTHORIZGRAD   calculus/LepticSolver/LevelLepticSolverF.ChF:59-99 (SpaceDim 3)
 * bcVals(i,j) on the vertical boundary face (index k = face): scale * J g^{z m} d(phi)/dx^m, m = x, y, averaged over
 * the ghost and the first valid layer.  isign = -1: bottom face (ghost cell k-1, valid cell k); +1: top face (ghost
 * cell k, valid cell k-1).  bcVals is a flat array over [lo0..hi0] x [lo1..hi1]; extrap = phi with ghosts filled.
 * ---------------------------------------------------------------------- */
void orc_lepticverthorizgrad(double *bc_, const int *blo, const int *bhi,
                             const double *ex_, const int *elo, const int *ehi,
                             const double *jgz_, const int *zlo, const int *zhi,
                             const int *flo, const int *fhi, int isign, const double *dx, double scale)
{
    fra_t ex = mk((double *)ex_, elo, ehi), Jgz = mk((double *)jgz_, zlo, zhi);
    const long bs1 = (long)(bhi[0] - blo[0] + 1);
    const int g = -(1 - isign) / 2, v = -(1 + isign) / 2;
    const double dxinv0 = scale * 0.25 / dx[0], dxinv1 = scale * 0.25 / dx[1];
    for (int k = flo[2]; k <= fhi[2]; ++k)
        for (int j = flo[1]; j <= fhi[1]; ++j)
            for (int i = flo[0]; i <= fhi[0]; ++i)
                bc_[(i - blo[0]) + bs1 * (j - blo[1])] =
                    AT(Jgz, i, j, k, 0) * dxinv0 *
                        (AT(ex, i + 1, j, k + g, 0) - AT(ex, i - 1, j, k + g, 0) + AT(ex, i + 1, j, k + v, 0) - AT(ex, i - 1, j, k + v, 0)) +
                    AT(Jgz, i, j, k, 1) * dxinv1 *
                        (AT(ex, i, j + 1, k + g, 0) - AT(ex, i, j - 1, k + g, 0) + AT(ex, i, j + 1, k + v, 0) - AT(ex, i, j - 1, k + v, 0));
}
