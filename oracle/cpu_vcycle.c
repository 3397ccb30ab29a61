/*
 * oracle/cpu_vcycle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The CPU baseline of bench.py: the reference's single-level multigrid V-cycle
 * (MappedMultiGrid<T>::cycle, calculus/AMRElliptic/MappedMultiGrid.H:555-653) for the
 * BASELINE C2 workload -- ONE box covering the domain, diagonal metric, homogeneous
 * Neumann on every side, no periodic direction, LevelGSRB smoother, BiCGStab bottom
 * solver -- orchestrated entirely in C over the kernels of kernels.c (the restated
 * Fortran), with OpenMP over k-slabs of each kernel's region standing in for Chombo's
 * one-MPI-rank-per-core box decomposition (SURVEY.md 8d "CPU reference timing").
 * The call sequence per operation is the one oracle/somar_oracle.py restates from the
 * reference (flux temporaries written and read back, ghost fill before every colour
 * pass, 26 boundary sub-box calls per pass); tests/test_oracle_cpu_vcycle.py checks the
 * result of one V-cycle against that Python orchestration bit for bit (1 thread) and
 * to round-off (several threads: only the zero-average sums associate differently).
 *
 * Not restated here (never reached by a one-box isotropic Neumann level): exchange,
 * coarse-fine ghosts, the factory's anisotropic fallback (MappedAMRPoissonOpFactory.cpp:504-550).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC cpu_vcycle.c -o liboracle_cpu.so -lm
 */
#include "kernels.c"

#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXDEPTH 16

typedef struct {
    int n[3];          /* cells */
    double dx[3];
    double *jg[3];     /* Jg^{aa} on faces(valid, a) */
    double *jinv, *lapd;
    int own_metric;
    double *res, *corr;         /* MG residual (no ghosts) / correction (1 ghost) of this depth */
    double *resfine;            /* restrictResidual's resFine */
    double *flux[3];            /* applyOpI's flux temporaries */
    int r[3];                   /* mgCrseRefRatio (to the next depth) */
    int zeroAvg;
    double dxProduct;
    /* boxes */
    int vlo[3], vhi[3];         /* valid */
    int glo[3], ghi[3];         /* valid grown by 1 */
    int flo[3][3], fhi[3][3];   /* faces(valid, a) */
    /* boundary sub-boxes of collectBoundaryData, in its order */
    int nb;
    int blo[26][3], bhi[26][3], bst[26][6];
} lev_t;

typedef struct {
    int depth, pre, post, bottom, nthreads;
    lev_t L[MAXDEPTH];
    /* BiCGStab parameters (utils/ProblemContext.cpp:1207-1231) */
    int b_imax, b_numRestarts, b_normType;
    double b_eps, b_reps, b_hang, b_small, b_metric;
    int b_iters, b_exit;
    double *bv[8];
} mg_t;

static long cells(const int *lo, const int *hi)
{
    return (long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
}

/* split [lo,hi] into slabs along direction sd; slab s of ns */
static void slab(const int *lo, const int *hi, int sd, int s, int ns, int *slo, int *shi)
{
    for (int d = 0; d < 3; ++d) { slo[d] = lo[d]; shi[d] = hi[d]; }
    long n = hi[sd] - lo[sd] + 1;
    slo[sd] = lo[sd] + (int)(n * s / ns);
    shi[sd] = lo[sd] + (int)(n * (s + 1) / ns) - 1;
}

/* how many slabs a region is cut into, and along which direction (the slowest one that is long enough) */
static int nslabs(const mg_t *M, const int *lo, const int *hi, int *sd)
{
    *sd = 2;
    if (hi[2] - lo[2] + 1 < 2 && hi[1] - lo[1] + 1 >= 2) *sd = 1;
    if (M->nthreads <= 1 || cells(lo, hi) < 4096) return 1;
    long n = hi[*sd] - lo[*sd] + 1;
    long ns = 4L * M->nthreads;
    return (int)(ns < n ? ns : n);
}

#define PAR_REGION(M, lo, hi, BODY)                                        \
    do {                                                                   \
        int sd_, ns_ = nslabs(M, lo, hi, &sd_);                            \
        _Pragma("omp parallel for schedule(static) if (ns_ > 1)")          \
        for (int s_ = 0; s_ < ns_; ++s_) {                                 \
            int slo[3], shi[3];                                            \
            slab(lo, hi, sd_, s_, ns_, slo, shi);                          \
            if (shi[sd_] < slo[sd_]) continue;                             \
            BODY;                                                          \
        }                                                                  \
    } while (0)

/* ---- boxes ---------------------------------------------------------------------------- */
static int box_and(const int *alo, const int *ahi, const int *blo, const int *bhi, int *lo, int *hi)
{
    int ok = 1;
    for (int d = 0; d < 3; ++d) {
        lo[d] = alo[d] > blo[d] ? alo[d] : blo[d];
        hi[d] = ahi[d] < bhi[d] ? ahi[d] : bhi[d];
        if (hi[d] < lo[d]) ok = 0;
    }
    return ok;
}
/* Chombo adjCellLo/Hi(box, dir, 1) with side: the 1-cell layer just OUTSIDE box */
static void adj_cell(const int *blo, const int *bhi, int d, int side, int *lo, int *hi)
{
    for (int e = 0; e < 3; ++e) { lo[e] = blo[e]; hi[e] = bhi[e]; }
    if (side == 0) { lo[d] = blo[d] - 1; hi[d] = blo[d] - 1; }
    else { lo[d] = bhi[d] + 1; hi[d] = bhi[d] + 1; }
}

/* RelaxationMethod::collectBoundaryData (RelaxationMethod.cpp:83-364), one box == the domain, all sides Neumann */
static void collect_boundary(lev_t *L)
{
    int ilo[3], ihi[3];
    for (int d = 0; d < 3; ++d) { ilo[d] = L->vlo[d] + 1; ihi[d] = L->vhi[d] - 1; }
    L->nb = 0;
#define PUSH(lo_, hi_)                                                                              \
    do {                                                                                            \
        int q = L->nb++;                                                                            \
        for (int d = 0; d < 3; ++d) { L->blo[q][d] = lo_[d]; L->bhi[q][d] = hi_[d]; }               \
        for (int d = 0; d < 3; ++d) {                                                               \
            L->bst[q][2 * d] = (lo_[d] == L->vlo[d]) ? BC_NEUM : -1;                                \
            L->bst[q][2 * d + 1] = (hi_[d] == L->vhi[d]) ? BC_NEUM : -1;                            \
        }                                                                                           \
    } while (0)
    for (int fdir = 0; fdir < 3; ++fdir)
        for (int fs = 0; fs < 2; ++fs) {
            int alo[3], ahi[3], flo[3], fhi[3];
            adj_cell(ilo, ihi, fdir, fs, alo, ahi);
            if (!box_and(alo, ahi, L->vlo, L->vhi, flo, fhi)) continue;
            PUSH(flo, fhi);
            for (int edir = fdir + 1; edir < 3; ++edir)
                for (int es = 0; es < 2; ++es) {
                    int elo[3], ehi[3];
                    adj_cell(flo, fhi, edir, es, alo, ahi);
                    if (!box_and(alo, ahi, L->vlo, L->vhi, elo, ehi)) continue;
                    PUSH(elo, ehi);
                    int vdir = 3 - fdir - edir;
                    if (vdir <= edir) continue;
                    for (int vs = 0; vs < 2; ++vs) {
                        int wlo[3], whi[3];
                        adj_cell(elo, ehi, vdir, vs, alo, ahi);
                        if (!box_and(alo, ahi, L->vlo, L->vhi, wlo, whi)) continue;
                        PUSH(wlo, whi);
                    }
                }
        }
#undef PUSH
}

static void set_boxes(lev_t *L)
{
    for (int d = 0; d < 3; ++d) {
        L->vlo[d] = 0; L->vhi[d] = L->n[d] - 1;
        L->glo[d] = -1; L->ghi[d] = L->n[d];
    }
    for (int a = 0; a < 3; ++a)
        for (int d = 0; d < 3; ++d) { L->flo[a][d] = 0; L->fhi[a][d] = L->n[d] - 1 + (a == d); }
    L->dxProduct = L->dx[0] * L->dx[1] * L->dx[2];
    collect_boundary(L);
}

static double *zalloc(long n)
{
    double *p = (double *)malloc(sizeof(double) * (size_t)n);
    if (!p) { fprintf(stderr, "cpu_vcycle: out of memory (%ld doubles)\n", n); abort(); }
    return p;
}
static void par_zero(const mg_t *M, double *p, long n)
{
    (void)M;
#pragma omp parallel for schedule(static) if (n > 65536)
    for (long i = 0; i < n; ++i) p[i] = 0.0;
}

/* ---- level operations -------------------------------------------------------------------- */
/* bc_set_ghosts with homogeneous Neumann on all six sides (EllipticConstNeumBCGhostClass, EllipticBCUtils.cpp:431-482) */
static void set_ghosts(const mg_t *M, const lev_t *L, double *phi)
{
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            int lo[3], hi[3];
            adj_cell(L->vlo, L->vhi, d, side, lo, hi);
            PAR_REGION(M, lo, hi,
                       orc_ellipticconstneumbcghostortho(phi, L->glo, L->ghi, 1, L->jg[d], L->flo[d], L->fhi[d], slo, shi,
                                                         0.0, d, side ? 1 : -1, L->dx[d]));
        }
}

/* LevelGSRB::relax (GSRB.cpp:58-98) */
static void relax_g(const mg_t *M, const lev_t *L, double *phi, const double *rhs, int rg, int iters)
{
    const int *rlo = rg ? L->glo : L->vlo, *rhi = rg ? L->ghi : L->vhi;   /* the box rhs is defined on */
    int ilo[3], ihi[3];
    for (int d = 0; d < 3; ++d) { ilo[d] = L->vlo[d] + 1; ihi[d] = L->vhi[d] - 1; }
    const int have_interior = ihi[0] >= ilo[0] && ihi[1] >= ilo[1] && ihi[2] >= ilo[2];
    for (int it = 0; it < iters; ++it)
        for (int pass = 0; pass < 2; ++pass) {
            set_ghosts(M, L, phi);
            if (have_interior)
                PAR_REGION(M, ilo, ihi,
                           orc_gsrbiter3dortho(phi, L->glo, L->ghi, 1, rhs, rlo, rhi, L->jg[0], L->flo[0], L->fhi[0],
                                               L->jg[1], L->flo[1], L->fhi[1], L->jg[2], L->flo[2], L->fhi[2], L->jinv,
                                               L->vlo, L->vhi, L->lapd, L->vlo, L->vhi, slo, shi, L->dx, 0.0, 1.0, pass));
            for (int q = 0; q < L->nb; ++q)
                PAR_REGION(M, L->blo[q], L->bhi[q],
                           orc_gsrbboundaryiter3dortho(phi, L->glo, L->ghi, 1, rhs, rlo, rhi, L->jg[0], L->flo[0],
                                                       L->fhi[0], L->jg[1], L->flo[1], L->fhi[1], L->jg[2], L->flo[2],
                                                       L->fhi[2], L->jinv, L->vlo, L->vhi, slo, shi, L->dx, 0.0, 1.0,
                                                       L->bst[q], pass));
        }
}

static void relax(const mg_t *M, const lev_t *L, double *phi, const double *rhs, int iters)
{
    relax_g(M, L, phi, rhs, 0, iters);
}

/* applyOpI, homogeneous (MappedAMRPoissonOp.cpp:772-898): ghosts, three face fluxes, zero boundary fluxes,
 * flux *= beta, divergence */
static void apply_op(const mg_t *M, const lev_t *L, double *lhs, double *phi)
{
    set_ghosts(M, L, phi);
    for (int d = 0; d < 3; ++d) {
        PAR_REGION(M, L->flo[d], L->fhi[d],
                   orc_mappedgetfluxortho(L->flux[d], L->flo[d], L->fhi[d], 1, phi, L->glo, L->ghi, L->jg[d], L->flo[d],
                                          L->fhi[d], slo, shi, 1.0 / L->dx[d], d));
    }
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            int lo[3], hi[3];
            for (int e = 0; e < 3; ++e) { lo[e] = L->flo[d][e]; hi[e] = L->fhi[d][e]; }
            if (side == 0) hi[d] = lo[d]; else lo[d] = hi[d];
            fra_t F = mk(L->flux[d], L->flo[d], L->fhi[d]);
            for (int k = lo[2]; k <= hi[2]; ++k)
                for (int j = lo[1]; j <= hi[1]; ++j)
                    for (int i = lo[0]; i <= hi[0]; ++i) AT(F, i, j, k, 0) = 0.0;
        }
    const double beta = 1.0;
    for (int d = 0; d < 3; ++d) {   /* fluxFB *= m_beta (:841) */
        double *f = L->flux[d];
        long n = cells(L->flo[d], L->fhi[d]);
#pragma omp parallel for schedule(static) if (n > 65536 && M->nthreads > 1)
        for (long i = 0; i < n; ++i) f[i] *= beta;
    }
    PAR_REGION(M, L->vlo, L->vhi,
               orc_mappedfluxdivergence3d(lhs, L->vlo, L->vhi, 1, L->flux[0], L->flo[0], L->fhi[0], L->flux[1], L->flo[1],
                                          L->fhi[1], L->flux[2], L->flo[2], L->fhi[2], L->jinv, L->vlo, L->vhi, slo, shi,
                                          L->dx));
}

static void residual(const mg_t *M, const lev_t *L, double *lhs, double *phi, const double *rhs)
{
    apply_op(M, L, lhs, phi);
    PAR_REGION(M, L->vlo, L->vhi, orc_subtractop(lhs, L->vlo, L->vhi, 1, rhs, L->vlo, L->vhi, lhs, L->vlo, L->vhi, slo, shi));
}

/* restrictResidual (MappedAMRPoissonOp.cpp:1281-1304) + FullWeightingPS::restrict */
static void restrict_residual(const mg_t *M, const lev_t *L, const lev_t *C, double *resCoarse, double *phiFine,
                              const double *rhsFine)
{
    residual(M, L, L->resfine, phiFine, rhsFine);
    PAR_REGION(M, C->vlo, C->vhi,
               orc_mappedaverage2(resCoarse, C->vlo, C->vhi, 1, L->resfine, L->vlo, L->vhi, L->jinv, L->vlo, L->vhi, slo, shi,
                                  L->r));
}

/* ConstInterpPS / ZeroAvgConstInterpPS::prolongIncrement (ProlongationStrategy.cpp:49-84, 91-164) */
static void prolong_increment(const mg_t *M, const lev_t *L, const lev_t *C, double *phiFine, const double *corrCoarse)
{
    if (!L->zeroAvg) {
        PAR_REGION(M, L->vlo, L->vhi,
                   orc_constinterpps(phiFine, L->glo, L->ghi, 1, corrCoarse, C->glo, C->ghi, slo, shi, L->r));
        return;
    }
    int sd, ns = nslabs(M, L->vlo, L->vhi, &sd);
    double *pv = (double *)calloc((size_t)ns * 2, sizeof(double));
#pragma omp parallel for schedule(static) if (ns > 1)
    for (int s = 0; s < ns; ++s) {
        int slo[3], shi[3];
        slab(L->vlo, L->vhi, sd, s, ns, slo, shi);
        if (shi[sd] < slo[sd]) continue;
        orc_constinterpwithavgps(phiFine, L->glo, L->ghi, 1, corrCoarse, C->glo, C->ghi, slo, shi, L->r, L->jinv, L->vlo,
                                 L->vhi, L->dxProduct, &pv[2 * s], &pv[2 * s + 1]);
    }
    double vol, sum;
    if (ns == 1) { vol = pv[0]; sum = pv[1]; }
    else { vol = 0.0; sum = 0.0; for (int s = 0; s < ns; ++s) { vol += pv[2 * s]; sum += pv[2 * s + 1]; } }
    free(pv);
    const double avg = sum / vol;
    long n = cells(L->glo, L->ghi);
#pragma omp parallel for schedule(static) if (n > 65536 && M->nthreads > 1)
    for (long i = 0; i < n; ++i) phiFine[i] -= avg;   /* the whole FAB, ghosts included */
}

/* ---- vector ops on the bottom level (LevelDataOps; sums in Fortran order) ------------------- */
static double dot_valid(const lev_t *L, const double *a, int ag, const double *b, int bg)
{
    fra_t A = ag ? mk((double *)a, L->glo, L->ghi) : mk((double *)a, L->vlo, L->vhi);
    fra_t B = bg ? mk((double *)b, L->glo, L->ghi) : mk((double *)b, L->vlo, L->vhi);
    double t = 0.0;
    for (int k = L->vlo[2]; k <= L->vhi[2]; ++k)
        for (int j = L->vlo[1]; j <= L->vhi[1]; ++j)
            for (int i = L->vlo[0]; i <= L->vhi[0]; ++i) t += AT(A, i, j, k, 0) * AT(B, i, j, k, 0);
    return t;
}
static double norm_valid(const lev_t *L, const double *a, int ag, int ord)
{
    fra_t A = ag ? mk((double *)a, L->glo, L->ghi) : mk((double *)a, L->vlo, L->vhi);
    double t = 0.0;
    for (int k = L->vlo[2]; k <= L->vhi[2]; ++k)
        for (int j = L->vlo[1]; j <= L->vhi[1]; ++j)
            for (int i = L->vlo[0]; i <= L->vhi[0]; ++i) {
                double v = fabs(AT(A, i, j, k, 0));
                if (ord == 0) t = v > t ? v : t;
                else if (ord == 1) t += v;
                else if (ord == 2) t += v * v;
                else t += pow(v, (double)ord);
            }
    if (ord == 0 || ord == 1) return t;
    return pow(t, 1.0 / ord);
}
/* dst (ghosts dg) += s * src (ghosts sg) on the intersection of their boxes */
static void incr(const lev_t *L, double *dst, int dg, const double *src, int sg, double s)
{
    const int g = dg && sg;
    const int *lo = g ? L->glo : L->vlo, *hi = g ? L->ghi : L->vhi;
    fra_t D = dg ? mk(dst, L->glo, L->ghi) : mk(dst, L->vlo, L->vhi);
    fra_t S = sg ? mk((double *)src, L->glo, L->ghi) : mk((double *)src, L->vlo, L->vhi);
    for (int k = lo[2]; k <= hi[2]; ++k)
        for (int j = lo[1]; j <= hi[1]; ++j)
            for (int i = lo[0]; i <= hi[0]; ++i) AT(D, i, j, k, 0) += s * AT(S, i, j, k, 0);
}
static void assign(const lev_t *L, double *dst, int dg, const double *src, int sg)
{
    const int g = dg && sg;
    const int *lo = g ? L->glo : L->vlo, *hi = g ? L->ghi : L->vhi;
    fra_t D = dg ? mk(dst, L->glo, L->ghi) : mk(dst, L->vlo, L->vhi);
    fra_t S = sg ? mk((double *)src, L->glo, L->ghi) : mk((double *)src, L->vlo, L->vhi);
    for (int k = lo[2]; k <= hi[2]; ++k)
        for (int j = lo[1]; j <= hi[1]; ++j)
            for (int i = lo[0]; i <= hi[0]; ++i) AT(D, i, j, k, 0) = AT(S, i, j, k, 0);
}
static void setval(double *a, long n, double v) { for (long i = 0; i < n; ++i) a[i] = v; }
static void scale(double *a, long n, double s) { for (long i = 0; i < n; ++i) a[i] *= s; }

/* preCond, DiagRelax (MappedAMRPoissonOp.cpp:684-734) */
static void pre_cond(const mg_t *M, const lev_t *L, double *phi, const double *rhs, int rg)
{
    orc_diagprecond(phi, L->glo, L->ghi, 1, rhs, rg ? L->glo : L->vlo, rg ? L->ghi : L->vhi, L->lapd, L->vlo, L->vhi,
                    L->vlo, L->vhi, 0.0, 1.0);
    relax_g(M, L, phi, rhs, rg, 2);
}

/* Chombo 3.1 BiCGStabSolver::solve (EXTERNAL; restated from the published source as in somar_oracle.BiCGStab) */
static void bicgstab(mg_t *M, const lev_t *L, double *phi, const double *rhs)
{
    const long nv = cells(L->vlo, L->vhi), ng = cells(L->glo, L->ghi);
    double *r = M->bv[0], *r_tilde = M->bv[1], *t = M->bv[2], *v = M->bv[3];          /* like rhs */
    double *e = M->bv[4], *p = M->bv[5], *p_tilde = M->bv[6], *s_tilde = M->bv[7];    /* like phi */
    setval(r, nv, 0.0); setval(r_tilde, nv, 0.0); setval(t, nv, 0.0); setval(v, nv, 0.0);
    setval(e, ng, 0.0); setval(p, ng, 0.0); setval(p_tilde, ng, 0.0); setval(s_tilde, ng, 0.0);
    int recount = 0;
    residual(M, L, r, phi, rhs);
    assign(L, r_tilde, 0, r, 0);
    int i = 0;
    double rho[4] = {0, 0, 0, 0};
    double norm[2];
    norm[0] = norm_valid(L, r, 0, M->b_normType);
    double initial_norm = norm[0];
    const double initial_rnorm = norm[0];
    norm[1] = norm[0];
    double alpha[2] = {0, 0}, beta[2] = {0, 0}, omega[2] = {0, 0};
    int init = 1, restarts = 0;
    if (M->b_metric > 0) initial_norm = M->b_metric;
    M->b_exit = -1;
    while ((i < M->b_imax && norm[0] > M->b_eps * norm[1]) && (norm[1] > 0)) {
        ++i;
        norm[1] = norm[0];
        alpha[1] = alpha[0]; beta[1] = beta[0]; omega[1] = omega[0];
        rho[3] = rho[2]; rho[2] = rho[1];
        rho[1] = dot_valid(L, r_tilde, 0, r, 0);
        if (rho[1] == 0.0) {
            incr(L, phi, 1, e, 1, 1.0);
            M->b_exit = 2; M->b_iters = i;
            return;
        }
        if (init) { assign(L, p, 1, r, 0); init = 0; }
        else {
            beta[1] = (rho[1] / rho[2]) * (alpha[1] / omega[1]);
            scale(p, ng, beta[1]);
            incr(L, p, 1, v, 0, -beta[1] * omega[1]);
            incr(L, p, 1, r, 0, 1.0);
        }
        pre_cond(M, L, p_tilde, p, 1);
        apply_op(M, L, v, p_tilde);
        double m = dot_valid(L, r_tilde, 0, v, 0);
        alpha[0] = rho[1] / m;
        if (fabs(m) > M->b_small * fabs(rho[1])) {
            incr(L, r, 0, v, 0, -alpha[0]);
            norm[0] = norm_valid(L, r, 0, M->b_normType);
            incr(L, e, 1, p_tilde, 1, alpha[0]);
        } else {
            setval(r, nv, 0.0);
            norm[0] = 0.0;
        }
        if (norm[0] > M->b_eps * initial_norm && norm[0] > M->b_reps * initial_rnorm) {
            pre_cond(M, L, s_tilde, r, 0);
            apply_op(M, L, t, s_tilde);
            omega[0] = dot_valid(L, t, 0, r, 0) / dot_valid(L, t, 0, t, 0);
            incr(L, e, 1, s_tilde, 1, omega[0]);
            incr(L, r, 0, t, 0, -omega[0]);
            norm[0] = norm_valid(L, r, 0, M->b_normType);
        }
        if (norm[0] <= M->b_eps * initial_norm || norm[0] <= M->b_reps * initial_rnorm) { M->b_exit = 1; break; }
        if (omega[0] == 0.0 || norm[0] > (1 - M->b_hang) * norm[1]) {
            if (recount == 0) recount = 1;
            else {
                recount = 0;
                incr(L, phi, 1, e, 1, 1.0);
                if (restarts == M->b_numRestarts) { M->b_exit = 3; M->b_iters = i; return; }
                residual(M, L, r, phi, rhs);
                norm[0] = norm_valid(L, r, 0, M->b_normType);
                rho[0] = rho[1] = rho[2] = rho[3] = 0.0;
                alpha[0] = beta[0] = omega[0] = 0.0;
                assign(L, r_tilde, 0, r, 0);
                setval(e, ng, 0.0);
                ++restarts;
                init = 1;
            }
        }
    }
    incr(L, phi, 1, e, 1, 1.0);
    M->b_iters = i;
}

/* MappedMultiGrid::cycle (MappedMultiGrid.H:555-653), V-cycle */
static void cycle(mg_t *M, int d, double *corr, const double *res)
{
    lev_t *L = &M->L[d];
    if (d == M->depth - 1) {
        if (cells(L->vlo, L->vhi) == 1) relax(M, L, corr, res, 1);
        else {
            relax(M, L, corr, res, M->bottom);
            bicgstab(M, L, corr, res);
        }
        return;
    }
    lev_t *C = &M->L[d + 1];
    relax(M, L, corr, res, M->pre);
    restrict_residual(M, L, C, C->res, corr, res);
    par_zero(M, C->corr, cells(C->glo, C->ghi));
    cycle(M, d + 1, C->corr, C->res);
    prolong_increment(M, L, C, corr, C->corr);
    relax(M, L, corr, res, M->post);
}

/* FILLMAPPEDLAPDIAG3D (MappedAMRPoissonOpF.ChF:215-274) on one-component face arrays: orc_fillmappedlapdiag3d's
 * expression, the component index of FAB a dropped */
static void lapdiag1(const lev_t *L, const int *reglo, const int *reghi)
{
    fra_t lap = mk(L->lapd, L->vlo, L->vhi), Jg0 = mk(L->jg[0], L->flo[0], L->fhi[0]), Jg1 = mk(L->jg[1], L->flo[1], L->fhi[1]);
    fra_t Jg2 = mk(L->jg[2], L->flo[2], L->fhi[2]), Jinv = mk(L->jinv, L->vlo, L->vhi);
    const double *dx = L->dx;
    const double s0 = 1.0 / (dx[0] * dx[0]), s1 = 1.0 / (dx[1] * dx[1]), s2 = 1.0 / (dx[2] * dx[2]);
    for (int k = reglo[2]; k <= reghi[2]; ++k)
        for (int j = reglo[1]; j <= reghi[1]; ++j)
            for (int i = reglo[0]; i <= reghi[0]; ++i)
                AT(lap, i, j, k, 0) = -AT(Jinv, i, j, k, 0) *
                                      ((AT(Jg0, i + 1, j, k, 0) + AT(Jg0, i, j, k, 0)) * s0 +
                                       (AT(Jg1, i, j + 1, k, 0) + AT(Jg1, i, j, k, 0)) * s1 +
                                       (AT(Jg2, i, j, k + 1, 0) + AT(Jg2, i, j, k, 0)) * s2);
}

/* ---- construction ------------------------------------------------------------------------------ */
/* semicoarsening rule (MappedAMRPoissonOpFactory.cpp:476-495) */
static void choose_ratio(const double *dx, int *r)
{
    double maxDx = dx[0];
    for (int d = 1; d < 3; ++d) if (dx[d] > maxDx) maxDx = dx[d];
    for (int d = 0; d < 3; ++d) r[d] = (dx[d] <= maxDx / 2.0) ? 2 : 1;
    if (r[0] * r[1] * r[2] == 1) r[0] = r[1] = r[2] = 2;
}

static void alloc_fields(lev_t *L)
{
    L->res = zalloc(cells(L->vlo, L->vhi));
    L->corr = zalloc(cells(L->glo, L->ghi));
    L->resfine = zalloc(cells(L->vlo, L->vhi));
    for (int d = 0; d < 3; ++d) L->flux[d] = zalloc(cells(L->flo[d], L->fhi[d]));
    L->lapd = zalloc(cells(L->vlo, L->vhi));
}

void *cpuvc_create(const int *n, const double *dx, double *jg0, double *jg1, double *jg2, double *jinv, int pre,
                   int post, int bottom, int nthreads)
{
    mg_t *M = (mg_t *)calloc(1, sizeof(mg_t));
    M->pre = pre; M->post = post; M->bottom = bottom;
    M->nthreads = nthreads < 1 ? 1 : nthreads;
#ifdef _OPENMP
    omp_set_num_threads(M->nthreads);
#else
    M->nthreads = 1;
#endif
    M->b_imax = 80; M->b_numRestarts = 5; M->b_normType = 2;
    M->b_eps = 1e-6; M->b_reps = 1e-12; M->b_hang = 1e-15; M->b_small = 1e-30; M->b_metric = -1.0;
    lev_t *L = &M->L[0];
    for (int d = 0; d < 3; ++d) { L->n[d] = n[d]; L->dx[d] = dx[d]; }
    L->jg[0] = jg0; L->jg[1] = jg1; L->jg[2] = jg2; L->jinv = jinv; L->own_metric = 0;
    set_boxes(L);
    alloc_fields(L);
    M->depth = 1;
    int coarsening[3] = {1, 1, 1};
    for (;;) {
        lev_t *F = &M->L[M->depth - 1];
        PAR_REGION(M, F->vlo, F->vhi, lapdiag1(F, slo, shi));
        /* null-space probe (MappedAMRPoissonOpFactory.cpp:659-693): rhs = L[0], res = rhs - L[1] */
        {
            long ng = cells(F->glo, F->ghi), nv = cells(F->vlo, F->vhi);
            par_zero(M, F->corr, ng);
            apply_op(M, F, F->res, F->corr);
            for (long i = 0; i < ng; ++i) F->corr[i] = 1.0;
            residual(M, F, F->resfine, F->corr, F->res);
            double mx = F->resfine[0];
            for (long i = 1; i < nv; ++i) if (F->resfine[i] > mx) mx = F->resfine[i];
            F->zeroAvg = fabs(mx) < 0.01 * 1e-6;
            par_zero(M, F->corr, ng);
        }
        if (M->depth == MAXDEPTH) break;
        int r[3];
        choose_ratio(F->dx, r);
        int ok = 1;
        for (int d = 0; d < 3; ++d)
            if (n[d] % (coarsening[d] * r[d] * 4) != 0) ok = 0;   /* coarsenable(grids, coarsening * s_maxCoarse) */
        if (!ok) break;   /* the factory's fallback branch (:504-550) is not restated: isotropic one-box levels never reach it */
        for (int d = 0; d < 3; ++d) { coarsening[d] *= r[d]; F->r[d] = r[d]; }
        lev_t *C = &M->L[M->depth];
        for (int d = 0; d < 3; ++d) { C->n[d] = F->n[d] / r[d]; C->dx[d] = F->dx[d] * r[d]; }
        set_boxes(C);
        alloc_fields(C);
        C->own_metric = 1;
        for (int d = 0; d < 3; ++d) {
            C->jg[d] = zalloc(cells(C->flo[d], C->fhi[d]));
            PAR_REGION(M, C->flo[d], C->fhi[d],
                       orc_unmappedaverageface(C->jg[d], C->flo[d], C->fhi[d], 1, F->jg[d], F->flo[d], F->fhi[d], slo, shi, d, r));
        }
        C->jinv = zalloc(cells(C->vlo, C->vhi));
        PAR_REGION(M, C->vlo, C->vhi,
                   orc_unmappedaverageharmonic(C->jinv, C->vlo, C->vhi, 1, F->jinv, F->vlo, F->vhi, slo, shi, r));
        M->depth++;
    }
    lev_t *B = &M->L[M->depth - 1];
    for (int q = 0; q < 4; ++q) M->bv[q] = zalloc(cells(B->vlo, B->vhi));
    for (int q = 4; q < 8; ++q) M->bv[q] = zalloc(cells(B->glo, B->ghi));
    return M;
}

int cpuvc_depth(void *h) { return ((mg_t *)h)->depth; }
int cpuvc_zero_avg(void *h, int d) { return ((mg_t *)h)->L[d].zeroAvg; }
int cpuvc_bottom_iters(void *h) { return ((mg_t *)h)->b_iters; }
void cpuvc_set_threads(void *h, int nthreads)
{
    mg_t *M = (mg_t *)h;
    M->nthreads = nthreads < 1 ? 1 : nthreads;
#ifdef _OPENMP
    omp_set_num_threads(M->nthreads);
#else
    M->nthreads = 1;
#endif
}
void cpuvc_set_bottom_metric(void *h, double metric, double eps)
{
    ((mg_t *)h)->b_metric = metric;
    ((mg_t *)h)->b_eps = eps;
}

/* one V-cycle from a zero correction: corr spans valid grown by 1, res spans valid */
void cpuvc_vcycle(void *h, double *corr, const double *res)
{
    mg_t *M = (mg_t *)h;
    par_zero(M, corr, cells(M->L[0].glo, M->L[0].ghi));
    cycle(M, 0, corr, res);
}

/* one residual + one red+black sweep on depth 0 (the north-star unit), for per-kernel CPU rates */
void cpuvc_relax(void *h, double *phi, const double *rhs, int iters) { relax((mg_t *)h, &((mg_t *)h)->L[0], phi, rhs, iters); }
void cpuvc_residual(void *h, double *out, double *phi, const double *rhs) { residual((mg_t *)h, &((mg_t *)h)->L[0], out, phi, rhs); }

void cpuvc_destroy(void *h)
{
    mg_t *M = (mg_t *)h;
    for (int d = 0; d < M->depth; ++d) {
        lev_t *L = &M->L[d];
        free(L->res); free(L->corr); free(L->resfine); free(L->lapd);
        for (int a = 0; a < 3; ++a) free(L->flux[a]);
        if (L->own_metric) { for (int a = 0; a < 3; ++a) free(L->jg[a]); free(L->jinv); }
    }
    for (int q = 0; q < 8; ++q) free(M->bv[q]);
    free(M);
}

/* STREAM triad a = b + s*c over n doubles per array, best of `reps`; returns GB/s (3 arrays x 8 B per element) */
double cpuvc_triad(long n, int nthreads, int reps)
{
#ifdef _OPENMP
    omp_set_num_threads(nthreads < 1 ? 1 : nthreads);
#endif
    double *a = zalloc(n), *b = zalloc(n), *c = zalloc(n);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
    double best = 0.0;
    for (int r = 0; r < reps; ++r) {
#ifdef _OPENMP
        double t0 = omp_get_wtime();
#else
        double t0 = 0.0;
#endif
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) a[i] = b[i] + 3.0 * c[i];
#ifdef _OPENMP
        double dt = omp_get_wtime() - t0;
#else
        double dt = 1.0;
#endif
        double gbs = 24.0 * (double)n / dt / 1e9;
        if (gbs > best) best = gbs;
    }
    volatile double sink = a[n / 2];
    (void)sink;
    free(a); free(b); free(c);
    return best;
}
