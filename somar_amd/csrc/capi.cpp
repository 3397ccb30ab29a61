// somar_amd/csrc/capi.cpp -- extern "C" boundary of libsomar_amd.so (include/somar_amd.h).
#include <cmath>
#include <random>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/somar_amd.h"
#include "amr.h"
#include "leptic.h"
#include "solver.h"

namespace somar {
void rccl_unique_id(unsigned char* id128);
Comm* rccl_create(const unsigned char* id128, int rank, int nranks, int device);
Comm* shm_create(const char* name, int rank, int nranks, size_t outbox_bytes);
}  // namespace somar

using namespace somar;

struct somar_solver {
    PressureSolver* ps = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool owned = true;  // false: a level of a somar_amr (somar_amr_level)
};

struct somar_leptic {
    LepticSolver* lep = nullptr;
    somar_solver* level = nullptr;
    somar_solver* parts[2] = {nullptr, nullptr};  // J-scaled operator + full multigrid, flat multigrid
};

struct somar_amr {
    AMRSolver* amr = nullptr;
    std::vector<somar_solver*> levels;
};

static thread_local std::string g_err;

#define API_BEGIN try {
#define API_END                              \
    }                                        \
    catch (const somar::Error& e)            \
    {                                        \
        g_err = e.what();                    \
        return e.code;                       \
    }                                        \
    catch (const std::exception& e)          \
    {                                        \
        g_err = e.what();                    \
        return -99;                          \
    }                                        \
    return 0;

static double* field_ptr(somar_solver* s, int field, int* depth_out = nullptr)
{
    const int which = field & 0xff, depth = field >> 8;
    PressureSolver& ps = *s->ps;
    SOMAR_CHECK(depth >= 0 && depth < ps.depth(), "field depth out of range");
    if (depth_out) *depth_out = depth;
    double* p = ps.field(depth, which);
    SOMAR_CHECK(p != nullptr, "no such resident field at this depth");
    return p;
}

extern "C" {

int somar_abi_version(void) { return SOMAR_AMD_ABI_VERSION; }
const char* somar_last_error(void) { return g_err.c_str(); }

int somar_host_fill_mt19937_64(double* out, long long n, unsigned long long seed, double lo, double hi)
{
    API_BEGIN
    SOMAR_CHECK(out && n >= 0 && lo < hi, "somar_host_fill_mt19937_64: bad arguments");
    std::mt19937_64 gen(seed);
    std::uniform_real_distribution<double> dist(lo, hi);
    for (long long i = 0; i < n; ++i) out[i] = dist(gen);
    API_END
}

// LedgeMap::fill_bathymetry (geometry/maps/LedgeMap.cpp:38-60 the coefficients, :98-164 the branches)
int somar_bathymetry_ledge(double* out, long long n, const double* x, const double* y, int order, double hl, double hr, double xl,
                           double xr)
{
    API_BEGIN
    SOMAR_CHECK(out && x && n >= 0, "somar_bathymetry_ledge: bad arguments");
    if (y) {   // the CH_SPACEDIM == 3 branch: "Alberto's Gaussian bump"
        for (long long i = 0; i < n; ++i) {
            const double a = (x[i] - xl) / hl, b = (y[i] - xr) / hr;
            const double R2 = a * a + b * b;
            out[i] = std::exp(-R2);
        }
    } else {
        SOMAR_CHECK(order == 1 || order == 3, "LedgeMap::m_transitionOrder must be 1 or 3");
        const double dh = hr - hl, dx = xr - xl;
        const double invdx3 = std::pow(dx, -3.0);
        double c0, c1, c2 = 0.0, c3 = 0.0;
        if (order == 1) {
            c0 = hr - xr * dh / dx;
            c1 = dh / dx;
        } else {
            c0 = hr + dh * (3.0 * xl - xr) * xr * xr * invdx3;
            c1 = -6.0 * dh * xl * xr * invdx3;
            c2 = 3.0 * dh * (xl + xr) * invdx3;
            c3 = -2.0 * dh * invdx3;
        }
        for (long long i = 0; i < n; ++i) {
            const double v = x[i];
            if (v < xl) out[i] = hl;
            else if (v > xr) out[i] = hr;
            else out[i] = order == 1 ? c0 + v * c1 : c0 + v * (c1 + v * (c2 + v * c3));
        }
    }
    API_END
}

// FILL_BeamGeneratorMapBATHYMETRY (geometry/maps/BeamGeneratorMapF.ChF:51-166)
int somar_bathymetry_beam_generator(double* out, long long n, const double* x, double Lx, double angle)
{
    API_BEGIN
    SOMAR_CHECK(out && x && n >= 0 && Lx > 0.0, "somar_bathymetry_beam_generator: bad arguments");
    const double lp = 0.009714, Bp = 0.01173, Pp = 0.0183542;   // Masoud's lab-scale ridge, the compiled-in PARAMETER set
    const double sa = std::sin(angle), ca = std::cos(angle), ta = std::tan(angle);
    const double l = lp * Lx, B = Bp * Lx, P = Pp * Lx;
    const double lstar = l + (B + P) / ca;
    const double C1 = -lstar * ca - B, C2 = -lstar * ca + B, C3 = -P, C4 = P, C5 = lstar * ca - B, C6 = lstar * ca + B;
    const double b0 = 0.25 * ta * (B + lstar * ca) * (B + lstar * ca) / B;
    const double b1 = -0.5 * ta * (B + lstar * ca) / B;
    const double b2 = 0.25 * ta / B;
    const double p0 = lstar * sa - 0.5 * ta * P;
    const double p2 = -0.5 * ta / P;
    for (long long i = 0; i < n; ++i) {
        const double v = x[i];
        if (v <= C1) out[i] = 0.0;
        else if (C1 < v && v < C2) out[i] = b2 * v * v - b1 * v + b0;
        else if (C2 <= v && v <= C3) out[i] = lstar * sa + ta * v;
        else if (C3 < v && v < C4) out[i] = p2 * v * v + p0;
        else if (C4 <= v && v <= C5) out[i] = lstar * sa - ta * v;
        else if (C5 < v && v < C6) out[i] = b2 * v * v + b1 * v + b0;
        else out[i] = 0.0;
    }
    API_END
}

// CubicSpline::solve (natural end conditions: lofbc = hifbc = 1e50) + CubicSpline::interp
// (calculus/interpolation/CubicSpline.cpp:57-143, CubicSplineF.ChF:46-112)
int somar_dem_cubic_spline(double* out, long long n, const double* x, int nd, const double* xd, const double* fd)
{
    API_BEGIN
    SOMAR_CHECK(out && x && xd && fd && n >= 0 && nd >= 2, "somar_dem_cubic_spline: bad arguments");
    std::vector<double> d2f(nd), u(nd);
    const int hi = nd - 1;
    u[0] = 0.0;
    d2f[0] = 0.0;
    for (int i = 1; i <= hi - 1; ++i) {
        const double dxl = xd[i] - xd[i - 1], dxr = xd[i + 1] - xd[i];
        const double dfl = fd[i] - fd[i - 1], dfr = fd[i + 1] - fd[i];
        const double sig = dxl / (dxr + dxl);
        const double p = sig * d2f[i - 1] + 2.0;
        d2f[i] = (sig - 1.0) / p;
        u[i] = (6.0 * (dfr / dxr - dfl / dxl) / (dxr + dxl) - sig * u[i - 1]) / p;
    }
    u[hi] = 0.0;
    d2f[hi] = 0.0;
    d2f[hi] = (u[hi] - d2f[hi] * u[hi - 1]) / (d2f[hi] * d2f[hi - 1] + 1.0);
    for (int i = hi - 1; i >= 0; --i) d2f[i] = d2f[i] * d2f[i + 1] + u[i];
    for (long long q = 0; q < n; ++q) {
        const double v = x[q];
        int klo = 0, khi = nd - 1;
        while (khi - klo > 1) {
            const int k = (khi + klo) >> 1;
            if (xd[k] > v) khi = k; else klo = k;
        }
        const double xlo = xd[klo], xhi = xd[khi], dx = xhi - xlo;
        SOMAR_CHECK(dx != 0.0, "somar_dem_cubic_spline: repeated abscissa");
        const double A = (xhi - v) / dx, B = (v - xlo) / dx;
        double f = A * fd[klo] + B * fd[khi];
        const double C = A * (A * A - 1.0), D = B * (B * B - 1.0);
        f += (C * d2f[klo] + D * d2f[khi]) * dx * dx / 6.0;
        out[q] = f;
    }
    API_END
}

// BilinearInterp2DF (calculus/interpolation/BilinearInterpF.ChF), xdir = 0, ydir = 1
int somar_dem_bilinear(double* out, long long n, const double* x, const double* y, int nx, int ny, const double* xd,
                       const double* yd, const double* fd)
{
    API_BEGIN
    SOMAR_CHECK(out && x && y && xd && yd && fd && n >= 0 && nx >= 1 && ny >= 1, "somar_dem_bilinear: bad arguments");
    for (long long q = 0; q < n; ++q) {
        const double xi = x[q], yj = y[q];
        int ilo = 0, ihi = nx - 1;
        while (ihi - ilo > 1) {
            const int i = (ihi + ilo) / 2;
            if (xd[i] > xi) ihi = i; else ilo = i;
        }
        int jlo = 0, jhi = ny - 1;
        while (jhi - jlo > 1) {
            const int j = (jhi + jlo) / 2;
            if (yd[j] > yj) jhi = j; else jlo = j;
        }
        const double xlo = xd[ilo], xhi = xd[ihi], ylo = yd[jlo], yhi = yd[jhi];
        const double fA = fd[ilo + (long long)nx * jlo], fB = fd[ihi + (long long)nx * jlo];
        const double fC = fd[ilo + (long long)nx * jhi], fD = fd[ihi + (long long)nx * jhi];
        const double u = xlo == xhi ? xi - xlo : (xi - xlo) / (xhi - xlo);
        const double v = ylo == yhi ? yj - ylo : (yj - ylo) / (yhi - ylo);
        out[q] = fA * (1 - u) * (1 - v) + fB * u * (1 - v) + fC * (1 - u) * v + fD * u * v;
    }
    API_END
}

// Create_Level_DEM_3D's derivative tables (DEMMap.cpp:222-289) + HermiteInterp2DF (HermiteInterpF.ChF:38-215), xdir = 0, ydir = 1
int somar_dem_hermite(double* out, long long n, const double* x, const double* y, int nx, int ny, const double* xd,
                      const double* yd, const double* fd)
{
    API_BEGIN
    SOMAR_CHECK(out && x && y && xd && yd && fd && n >= 0 && nx >= 2 && ny >= 2, "somar_dem_hermite: bad arguments");
    auto F = [&](int i, int j) { return fd[i + (long long)nx * j]; };
    std::vector<double> fx((size_t)nx * ny), fy((size_t)nx * ny);
    for (int j = 0; j < ny; ++j) {
        fx[0 + (size_t)nx * j] = (F(1, j) - F(0, j)) / (xd[1] - xd[0]);
        for (int i = 1; i < nx - 1; ++i) fx[i + (size_t)nx * j] = (F(i + 1, j) - F(i - 1, j)) / (xd[i + 1] - xd[i - 1]);
        fx[nx - 1 + (size_t)nx * j] = (F(nx - 1, j) - F(nx - 2, j)) / (xd[nx - 1] - xd[nx - 2]);
    }
    for (int i = 0; i < nx; ++i) {
        fy[i] = (F(i, 1) - F(i, 0)) / (yd[1] - yd[0]);
        for (int j = 1; j < ny - 1; ++j) fy[i + (size_t)nx * j] = (F(i, j + 1) - F(i, j - 1)) / (yd[j + 1] - yd[j - 1]);
        fy[i + (size_t)nx * (ny - 1)] = (F(i, ny - 1) - F(i, ny - 2)) / (yd[ny - 1] - yd[ny - 2]);
    }
    for (long long q = 0; q < n; ++q) {
        const double xi = x[q], yj = y[q];
        int ilo = 0, ihi = nx - 1;
        while (ihi - ilo > 1) {
            const int i = (ihi + ilo) / 2;
            if (xd[i] > xi) ihi = i; else ilo = i;
        }
        int jlo = 0, jhi = ny - 1;
        while (jhi - jlo > 1) {
            const int j = (jhi + jlo) / 2;
            if (yd[j] > yj) jhi = j; else jlo = j;
        }
        const double xlo = xd[ilo], xhi = xd[ihi], ylo = yd[jlo], yhi = yd[jhi];
        const size_t A = ilo + (size_t)nx * jlo, B = ihi + (size_t)nx * jlo, C = ilo + (size_t)nx * jhi, D = ihi + (size_t)nx * jhi;
        const double fA = fd[A], fB = fd[B], fC = fd[C], fD = fd[D];
        const double fxA = fx[A], fxB = fx[B], fxC = fx[C], fxD = fx[D];
        const double fyA = fy[A], fyB = fy[B], fyC = fy[C], fyD = fy[D];
        const double u = xlo == xhi ? xi - xlo : (xi - xlo) / (xhi - xlo);
        const double h4u = (u - 1.0) * u * u, h3u = ((u - 2.0) * u + 1.0) * u, h2u = (3.0 - 2.0 * u) * u * u, h1u = 1.0 - h2u;
        const double v = ylo == yhi ? yj - ylo : (yj - ylo) / (yhi - ylo);
        const double h4v = (v - 1.0) * v * v, h3v = ((v - 2.0) * v + 1.0) * v, h2v = (3.0 - 2.0 * v) * v * v, h1v = 1.0 - h2v;
        const double fAB = fA * h1u + fB * h2u + fxA * h3u + fxB * h4u;
        const double fCD = fC * h1u + fD * h2u + fxC * h3u + fxD * h4u;
        const double fAC = fA * h1v + fC * h2v + fyA * h3v + fyC * h4v;
        const double fBD = fB * h1v + fD * h2v + fyB * h3v + fyD * h4v;
        out[q] = fAB * h1v + fCD * h2v + fAC * h1u + fBD * h2u - fA * h1u * h1v - fB * h2u * h1v - fC * h1u * h2v - fD * h2u * h2v;
    }
    API_END
}

int somar_device_count(int* count)
{
    API_BEGIN
    int n = 0;
    SOMAR_HIP(hipGetDeviceCount(&n));
    *count = n;
    API_END
}

static void from_params(const SolverParams& d, somar_params_t* p)
{
    p->imin = d.imin; p->imax = d.imax; p->eps = d.eps; p->hang = d.hang; p->norm_thresh = d.normThresh;
    p->num_smooth_down = d.num_smooth_down; p->num_smooth_up = d.num_smooth_up;
    p->num_smooth_bottom = d.num_smooth_bottom; p->num_smooth_precond = d.num_smooth_precond;
    p->num_mg = d.numMG; p->max_depth = d.maxDepth; p->precond_mode = d.precondMode; p->relax_mode = d.relaxMode;
    p->verbosity = d.verbosity;
    p->bottom_imax = d.bottom_imax; p->bottom_num_restarts = d.bottom_numRestarts;
    p->bottom_norm_type = d.bottom_normType; p->bottom_verbosity = d.bottom_verbosity;
    p->bottom_eps = d.bottom_eps; p->bottom_reps = d.bottom_reps; p->bottom_hang = d.bottom_hang;
    p->bottom_small = d.bottom_small;
    p->space_dim = d.spaceDim;
}

int somar_params_default(somar_params_t* p)
{
    API_BEGIN
    from_params(SolverParams(), p);
    API_END
}

static SolverParams to_params(const somar_params_t* p)
{
    SolverParams d;
    if (!p) return d;
    d.imin = p->imin; d.imax = p->imax; d.eps = p->eps; d.hang = p->hang; d.normThresh = p->norm_thresh;
    d.num_smooth_down = p->num_smooth_down; d.num_smooth_up = p->num_smooth_up;
    d.num_smooth_bottom = p->num_smooth_bottom; d.num_smooth_precond = p->num_smooth_precond;
    d.numMG = p->num_mg; d.maxDepth = p->max_depth; d.precondMode = p->precond_mode; d.relaxMode = p->relax_mode;
    d.verbosity = p->verbosity;
    d.bottom_imax = p->bottom_imax; d.bottom_numRestarts = p->bottom_num_restarts;
    d.bottom_normType = p->bottom_norm_type; d.bottom_verbosity = p->bottom_verbosity;
    d.bottom_eps = p->bottom_eps; d.bottom_reps = p->bottom_reps; d.bottom_hang = p->bottom_hang;
    d.bottom_small = p->bottom_small;
    d.spaceDim = p->space_dim;
    return d;
}

int somar_solver_create(somar_solver_t** out, const int* domain_lo, const int* domain_hi, const int* periodic,
                        const double* dx, const int* bc_type, int nboxes, const int* boxes, const int* owner,
                        double alpha, double beta, const somar_params_t* prm, void* comm)
{
    API_BEGIN
    SOMAR_CHECK(out && domain_lo && domain_hi && periodic && dx && bc_type && boxes && nboxes > 0, "null/empty argument");
    IBox dom(domain_lo, domain_hi);
    bool per[3] = {periodic[0] != 0, periodic[1] != 0, periodic[2] != 0};
    int bct[3][2] = {{bc_type[0], bc_type[1]}, {bc_type[2], bc_type[3]}, {bc_type[4], bc_type[5]}};
    std::vector<IBox> bx;
    std::vector<int> own;
    for (int b = 0; b < nboxes; ++b) {
        IBox q(boxes + 6 * b, boxes + 6 * b + 3);
        SOMAR_CHECK(!q.empty(), "empty box");
        bx.push_back(q);
        own.push_back(owner ? owner[b] : 0);
    }
    // disjointness + containment (DisjointBoxLayout contract)
    for (size_t a = 0; a < bx.size(); ++a) {
        for (int d = 0; d < 3; ++d)
            SOMAR_CHECK(bx[a].lo[d] >= dom.lo[d] && bx[a].hi[d] <= dom.hi[d], "box outside the domain");
        for (size_t b = a + 1; b < bx.size(); ++b) SOMAR_CHECK((bx[a] & bx[b]).empty(), "boxes overlap");
    }
    somar_solver* s = new somar_solver;
    try {
        s->ps = new PressureSolver(static_cast<Comm*>(comm));
        s->ps->define(dom, per, dx, bct, bx, own, alpha, beta, to_params(prm));
        SOMAR_HIP(hipEventCreate(&s->ev0));
        SOMAR_HIP(hipEventCreate(&s->ev1));
    } catch (...) {
        delete s->ps;
        delete s;
        throw;
    }
    *out = s;
    API_END
}

int somar_solver_destroy(somar_solver_t* s)
{
    API_BEGIN
    if (s) {
        SOMAR_CHECK(s->owned, "this handle belongs to a somar_amr: destroy that instead");
        if (s->ev0) hipEventDestroy(s->ev0);
        if (s->ev1) hipEventDestroy(s->ev1);
        delete s->ps;
        delete s;
    }
    API_END
}

int somar_solver_num_local_patches(somar_solver_t* s, int* n)
{
    API_BEGIN
    *n = s->ps->level(0).npatches();
    API_END
}

int somar_solver_patch_box(somar_solver_t* s, int depth, int patch, int* box6, int* global_index)
{
    API_BEGIN
    SOMAR_CHECK(depth >= 0 && depth < s->ps->depth(), "depth out of range");
    Level& L = s->ps->level(depth);
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "patch out of range");
    const IBox& b = L.boxes[L.local[patch]];
    for (int d = 0; d < 3; ++d) { box6[d] = b.lo[d]; box6[3 + d] = b.hi[d]; }
    if (global_index) *global_index = L.local[patch];
    API_END
}

int somar_solver_set_metric_ortho(somar_solver_t* s, int patch, const double* jg0, const double* jg1,
                                  const double* jg2, const double* jinv)
{
    API_BEGIN
    SOMAR_CHECK(jg0 && jg1 && jinv && (jg2 || s->ps->prm.spaceDim == 2), "null metric pointer");
    s->ps->set_metric_ortho(patch, jg0, jg1, jg2, jinv);
    API_END
}

int somar_solver_set_bc_values(somar_solver_t* s, const double* values6)
{
    API_BEGIN
    SOMAR_CHECK(s && values6, "null argument");
    s->ps->set_bc_values(values6);
    API_END
}

int somar_solver_set_metric_full(somar_solver_t* s, int patch, const double* jg0, const double* jg1, const double* jg2,
                                 const double* jinv)
{
    API_BEGIN
    SOMAR_CHECK(jg0 && jg1 && jinv && (jg2 || s->ps->prm.spaceDim == 2), "null metric pointer");
    s->ps->set_metric_full(patch, jg0, jg1, jg2, jinv);
    API_END
}

int somar_solver_finalize(somar_solver_t* s)
{
    API_BEGIN
    s->ps->finalize();
    API_END
}

int somar_solver_depth(somar_solver_t* s, int* depth)
{
    API_BEGIN
    *depth = s->ps->depth();
    API_END
}

int somar_solver_mg_ref_ratio(somar_solver_t* s, int depth, int* r3)
{
    API_BEGIN
    SOMAR_CHECK(depth >= 0 && depth + 1 < s->ps->depth(), "no coarser depth");
    const std::array<int, 3> r = s->ps->ref_ratio(depth);
    for (int d = 0; d < 3; ++d) r3[d] = r[d];
    API_END
}

int somar_solver_metric_uniform(somar_solver_t* s, int depth, int* flag, double* c4)
{
    API_BEGIN
    SOMAR_CHECK(depth >= 0 && depth < s->ps->depth(), "depth out of range");
    const StencilParams& P = s->ps->level(depth).dev.P;
    *flag = P.uniform;
    if (c4)
        for (int a = 0; a < 4; ++a) c4[a] = P.uniform ? P.uc[a] : 0.0;
    API_END
}

int somar_solver_zero_avg(somar_solver_t* s, int depth, int* flag)
{
    API_BEGIN
    SOMAR_CHECK(depth >= 0 && depth < s->ps->depth(), "depth out of range");
    *flag = s->ps->level(depth).zeroAvg ? 1 : 0;
    API_END
}

int somar_solver_level_info(somar_solver_t* s, int depth, int* domain6, double* dx3, long long* cells,
                            long long* field_elems)
{
    API_BEGIN
    SOMAR_CHECK(depth >= 0 && depth < s->ps->depth(), "depth out of range");
    Level& L = s->ps->level(depth);
    for (int d = 0; d < 3; ++d) {
        if (domain6) { domain6[d] = L.domain.lo[d]; domain6[3 + d] = L.domain.hi[d]; }
        if (dx3) dx3[d] = L.dx[d];
    }
    if (cells) *cells = L.valid_cells_global;
    if (field_elems) *field_elems = L.field_elems;
    API_END
}

int somar_field_upload(somar_solver_t* s, int field, int patch, const double* host, const int* ghost)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    Level& L = s->ps->level(depth);
    SOMAR_CHECK(patch >= 0 && patch < L.npatches() && host && ghost, "bad patch / null pointer");
    for (int d = 0; d < 3; ++d) SOMAR_CHECK(ghost[d] >= 0 && ghost[d] <= FRAME, "ghost wider than device frame");
    const IBox valid = L.boxes[L.local[patch]];
    const IBox hb = valid.grow(ghost);
    L.upload(f, patch, host, hb, hb, s->ps->stream());
    s->ps->sync();
    API_END
}

int somar_field_download(somar_solver_t* s, int field, int patch, double* host, const int* ghost)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    SOMAR_CHECK(host && ghost, "null pointer");
    s->ps->download_field(f, depth, patch, host, ghost);
    API_END
}

int somar_field_set(somar_solver_t* s, int field, double value)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    launch_set(s->ps->stream(), f, s->ps->level(depth).field_elems, value);
    API_END
}

int somar_field_fill_hash(somar_solver_t* s, int field, unsigned long long seed)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    s->ps->fill_hash(depth, f, seed);
    API_END
}

int somar_field_remove_mean(somar_solver_t* s, int field)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    s->ps->remove_mean(depth, f);
    API_END
}

int somar_field_norm(somar_solver_t* s, int field, int ord, double* out)
{
    API_BEGIN
    int depth;
    double* f = field_ptr(s, field, &depth);
    *out = s->ps->norm(depth, f, ord);
    API_END
}

int somar_field_dot(somar_solver_t* s, int field_a, int field_b, double* out)
{
    API_BEGIN
    int da, db;
    double* a = field_ptr(s, field_a, &da);
    double* b = field_ptr(s, field_b, &db);
    SOMAR_CHECK(da == db, "fields of different depths");
    *out = s->ps->dot(da, a, b);
    API_END
}

static thread_local std::vector<double> g_history;   // the last solve's whole history (somar_last_history)

static void fill_stats(const SolveStats& st, somar_stats_t* o)
{
    g_history = st.history;
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->iters = st.iters;
    o->exit_status = st.exitStatus;
    o->status = st.status;
    o->bottom_iters = st.bottom_iters_last;
    o->bottom_exit = st.bottom_exit_last;
    o->initial_rnorm = st.initial_rnorm;
    o->final_rnorm = st.final_rnorm;
    o->nhistory = (int)std::min<size_t>(st.history.size(), SOMAR_MAX_HISTORY);
    for (int i = 0; i < o->nhistory; ++i) o->history[i] = st.history[i];
}

int somar_last_history(double* out, int capacity, int* n)
{
    API_BEGIN
    SOMAR_CHECK(n && capacity >= 0 && (out || capacity == 0), "somar_last_history: bad arguments");
    *n = (int)g_history.size();
    for (int i = 0; i < capacity && i < *n; ++i) out[i] = g_history[i];
    API_END
}

int somar_solver_solve(somar_solver_t* s, int zero_phi, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    s->ps->solve(zero_phi != 0, force_homogeneous != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_solver_solve_host(somar_solver_t* s, double* const* phi, const int* phi_ghost,
                            const double* const* rhs, const int* rhs_ghost, int l_max, int l_base,
                            int zero_phi, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(l_max == 0 && l_base == 0, "a somar_solver_t is ONE level (l_max = l_base = 0); hierarchies go through somar_amr_solve_host");
    SOMAR_CHECK(phi && rhs && phi_ghost && rhs_ghost, "null pointer");
    PressureSolver& ps = *s->ps;
    const int np = ps.level(0).npatches();
    for (int p = 0; p < np; ++p) {
        ps.upload_rhs(p, rhs[p], rhs_ghost);
        if (!zero_phi) ps.upload_phi(p, phi[p], phi_ghost);
    }
    SolveStats st;
    ps.solve(zero_phi != 0, force_homogeneous != 0, st);
    for (int p = 0; p < np; ++p) ps.download_phi(p, phi[p], phi_ghost);
    fill_stats(st, stats);
    API_END
}

int somar_level_relax(somar_solver_t* s, int depth, int phi_field, int rhs_field, int iters)
{
    API_BEGIN
    int d1, d2;
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(d1 == depth && d2 == depth, "field/depth mismatch");
    s->ps->relax(depth, phi, rhs, iters);
    API_END
}

int somar_level_residual(somar_solver_t* s, int depth, int out_field, int phi_field, int rhs_field)
{
    API_BEGIN
    int d0, d1, d2;
    double* out = field_ptr(s, out_field, &d0);
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(d0 == depth && d1 == depth && d2 == depth && out != rhs && out != phi, "field/depth mismatch or aliasing");
    s->ps->residual(depth, out, phi, rhs);
    API_END
}

int somar_level_apply_op(somar_solver_t* s, int depth, int out_field, int phi_field)
{
    API_BEGIN
    int d0, d1;
    double* out = field_ptr(s, out_field, &d0);
    double* phi = field_ptr(s, phi_field, &d1);
    SOMAR_CHECK(d0 == depth && d1 == depth && out != phi, "field/depth mismatch or aliasing");
    s->ps->apply_op(depth, out, phi);
    API_END
}

int somar_level_apply_op_bc(somar_solver_t* s, int out_field, int phi_field, int homogeneous)
{
    API_BEGIN
    int d0, d1;
    double* out = field_ptr(s, out_field, &d0);
    double* phi = field_ptr(s, phi_field, &d1);
    SOMAR_CHECK(d0 == 0 && d1 == 0 && out != phi, "depth-0 fields, no aliasing");
    s->ps->apply_op(0, out, phi, homogeneous != 0);
    API_END
}

int somar_level_residual_bc(somar_solver_t* s, int out_field, int phi_field, int rhs_field, int homogeneous)
{
    API_BEGIN
    int d0, d1, d2;
    double* out = field_ptr(s, out_field, &d0);
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(d0 == 0 && d1 == 0 && d2 == 0 && out != rhs && out != phi, "depth-0 fields, no aliasing");
    s->ps->residual(0, out, phi, rhs, homogeneous != 0);
    API_END
}

int somar_level_restrict_residual(somar_solver_t* s, int depth, int coarse_res_field, int phi_field, int rhs_field)
{
    API_BEGIN
    int dc, d1, d2;
    double* rc = field_ptr(s, coarse_res_field, &dc);
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(dc == depth + 1 && d1 == depth && d2 == depth, "field/depth mismatch");
    s->ps->restrict_residual(depth, rc, phi, rhs);
    API_END
}

int somar_level_prolong_increment(somar_solver_t* s, int depth, int phi_field, int coarse_corr_field)
{
    API_BEGIN
    int dc, d1;
    double* phi = field_ptr(s, phi_field, &d1);
    double* cc = field_ptr(s, coarse_corr_field, &dc);
    SOMAR_CHECK(dc == depth + 1 && d1 == depth, "field/depth mismatch");
    s->ps->prolong_increment(depth, phi, cc);
    API_END
}

int somar_level_precond(somar_solver_t* s, int depth, int phi_field, int rhs_field)
{
    API_BEGIN
    int d1, d2;
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(d1 == depth && d2 == depth && phi != rhs, "field/depth mismatch or aliasing");
    s->ps->pre_cond(depth, phi, rhs);
    API_END
}

int somar_vcycle(somar_solver_t* s, int corr_field, int res_field)
{
    API_BEGIN
    int d1, d2;
    double* e = field_ptr(s, corr_field, &d1);
    double* r = field_ptr(s, res_field, &d2);
    SOMAR_CHECK(d1 == 0 && d2 == 0 && e != r, "V-cycle starts at depth 0");
    s->ps->vcycle(e, r);
    API_END
}

int somar_mini_vcycle(somar_solver_t* s, int corr_field, int res_field)
{
    API_BEGIN
    int d1, d2;
    double* e = field_ptr(s, corr_field, &d1);
    double* r = field_ptr(s, res_field, &d2);
    SOMAR_CHECK(d1 == 0 && d2 == 0 && e != r, "the mini V-cycle starts at depth 0");
    s->ps->mini_vcycle(e, r);
    s->ps->sync();
    API_END
}

int somar_vcycle_from_zero(somar_solver_t* s, int corr_field, int res_field)
{
    API_BEGIN
    int d1, d2;
    double* c = field_ptr(s, corr_field, &d1);
    double* r = field_ptr(s, res_field, &d2);
    SOMAR_CHECK(d1 == 0 && d2 == 0, "V-cycles start at depth 0");
    s->ps->vcycle(c, r, true);
    API_END
}

int somar_bottom_solve(somar_solver_t* s, int phi_field, int rhs_field, int* iters, int* exit_code)
{
    API_BEGIN
    int d1, d2;
    double* phi = field_ptr(s, phi_field, &d1);
    double* rhs = field_ptr(s, rhs_field, &d2);
    SOMAR_CHECK(d1 == s->ps->depth() - 1 && d2 == d1 && phi != rhs, "bottom solve runs on the coarsest depth");
    s->ps->bottom_solve(phi, rhs);
    if (iters) *iters = s->ps->bottom_iters;
    if (exit_code) *exit_code = s->ps->bottom_exit;
    API_END
}

int somar_bottom_kind(somar_solver_t* s, int* kind)
{
    API_BEGIN
    *kind = s->ps->bottom_kind;
    API_END
}

int somar_solver_counters(somar_solver_t* s, long long* out4)
{
    API_BEGIN
    for (int q = 0; q < 4; ++q) out4[q] = s->ps->counters[q];
    API_END
}

int somar_solver_fused19_sweeps(somar_solver_t* s, long long* n)
{
    API_BEGIN
    *n = s->ps->counters[4];
    API_END
}

int somar_vel_upload(somar_solver_t* s, int dir, int patch, const double* host)
{
    API_BEGIN
    SOMAR_CHECK(host, "null pointer");
    s->ps->upload_vel(dir, patch, host);
    API_END
}

int somar_vel_download(somar_solver_t* s, int dir, int patch, double* host)
{
    API_BEGIN
    SOMAR_CHECK(host, "null pointer");
    s->ps->download_vel(dir, patch, host);
    API_END
}

int somar_solver_set_vel_bc(somar_solver_t* s, const int* kind, const double* value)
{
    API_BEGIN
    SOMAR_CHECK(s && kind && value, "null argument");
    s->ps->set_vel_bc(kind, value);
    API_END
}

int somar_vel_wall_bc(somar_solver_t* s)
{
    API_BEGIN
    s->ps->vel_wall_bc();
    API_END
}

int somar_level_divergence_mac(somar_solver_t* s, int out_field, double dt)
{
    API_BEGIN
    int d0;
    double* out = field_ptr(s, out_field, &d0);
    SOMAR_CHECK(d0 == 0, "the MAC divergence lives on depth 0");
    s->ps->divergence_mac(out, dt);
    API_END
}

int somar_level_mac_correct(somar_solver_t* s, int phi_field, double dt)
{
    API_BEGIN
    int d0;
    double* phi = field_ptr(s, phi_field, &d0);
    SOMAR_CHECK(d0 == 0, "the MAC correction lives on depth 0");
    s->ps->mac_correct(phi, dt);
    API_END
}

int somar_mac_project(somar_solver_t* s, double dt, int zero_pressure, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    s->ps->mac_project(dt, zero_pressure != 0, force_homogeneous != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_mac_project_host(somar_solver_t* s, double* const* u0, double* const* u1, double* const* u2, double dt,
                           int zero_pressure, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(u0 && u1 && u2, "null pointer");
    PressureSolver& ps = *s->ps;
    const int np = ps.level(0).npatches();
    double* const* u[3] = {u0, u1, u2};
    for (int p = 0; p < np; ++p)
        for (int d = 0; d < 3; ++d) ps.upload_vel(d, p, u[d][p]);
    SolveStats st;
    ps.mac_project(dt, zero_pressure != 0, force_homogeneous != 0, st);
    for (int p = 0; p < np; ++p)
        for (int d = 0; d < 3; ++d) ps.download_vel(d, p, u[d][p]);
    fill_stats(st, stats);
    API_END
}

int somar_solver_set_alpha_beta(somar_solver_t* s, double a, double b)
{
    API_BEGIN
    s->ps->set_alpha_beta(a, b);
    API_END
}

int somar_heat_step(somar_solver_t* s, int scheme, double dt, int zero_phi, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    s->ps->heat_step(scheme, dt, zero_phi != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_ccvel_upload(somar_solver_t* s, int patch, const double* host, const int* ghost)
{
    API_BEGIN
    SOMAR_CHECK(host && ghost, "null pointer");
    s->ps->upload_cc_vel(patch, host, ghost);
    API_END
}

int somar_ccvel_download(somar_solver_t* s, int patch, double* host, const int* ghost)
{
    API_BEGIN
    SOMAR_CHECK(host && ghost, "null pointer");
    s->ps->download_cc_vel(patch, host, ghost);
    API_END
}

int somar_level_divergence_cc(somar_solver_t* s, int out_field, double dt, int wall_bc)
{
    API_BEGIN
    int d0;
    double* out = field_ptr(s, out_field, &d0);
    SOMAR_CHECK(d0 == 0, "the divergence lives on depth 0");
    s->ps->divergence_cc(out, dt, wall_bc != 0);
    API_END
}

int somar_level_cc_correct(somar_solver_t* s, int phi_field, double dt)
{
    API_BEGIN
    int d0;
    double* phi = field_ptr(s, phi_field, &d0);
    SOMAR_CHECK(d0 == 0, "the correction lives on depth 0");
    s->ps->cc_correct(phi, dt);
    API_END
}

int somar_cc_project(somar_solver_t* s, double dt, int zero_pressure, int force_homogeneous, int wall_bc,
                     somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    s->ps->cc_project(dt, zero_pressure != 0, force_homogeneous != 0, wall_bc != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_cc_project_host(somar_solver_t* s, double* const* vel, const int* ghost, double dt, int zero_pressure,
                          int force_homogeneous, int wall_bc, somar_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(vel && ghost, "null pointer");
    PressureSolver& ps = *s->ps;
    const int np = ps.level(0).npatches();
    for (int p = 0; p < np; ++p) ps.upload_cc_vel(p, vel[p], ghost);
    SolveStats st;
    ps.cc_project(dt, zero_pressure != 0, force_homogeneous != 0, wall_bc != 0, st);
    for (int p = 0; p < np; ++p) ps.download_cc_vel(p, vel[p], ghost);
    fill_stats(st, stats);
    API_END
}

int somar_sync(somar_solver_t* s)
{
    API_BEGIN
    s->ps->sync();
    API_END
}

int somar_timer_start(somar_solver_t* s)
{
    API_BEGIN
    SOMAR_HIP(hipEventRecord(s->ev0, s->ps->stream()));
    API_END
}

int somar_timer_stop(somar_solver_t* s, double* milliseconds)
{
    API_BEGIN
    SOMAR_HIP(hipEventRecord(s->ev1, s->ps->stream()));
    SOMAR_HIP(hipEventSynchronize(s->ev1));
    float ms = 0.f;
    SOMAR_HIP(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    *milliseconds = ms;
    API_END
}

int somar_profile_enable(somar_solver_t* s, int on)
{
    API_BEGIN
    s->ps->profile_enable(on != 0);
    API_END
}

int somar_profile_get(somar_solver_t* s, int kernel, int* launches, double* total_ms)
{
    API_BEGIN
    s->ps->profile_get(kernel, launches, total_ms);
    API_END
}

int somar_plan_exchange(const int* domain_lo, const int* domain_hi, const int* periodic, int nboxes, const int* boxes,
                        const int* owner, int rank, int ghost, int max_items, int* n_local, int* local_items,
                        int* n_send, int* send_items, int* n_recv, int* recv_items)
{
    API_BEGIN
    IBox dom(domain_lo, domain_hi);
    bool per[3] = {periodic[0] != 0, periodic[1] != 0, periodic[2] != 0};
    int g[3] = {ghost, ghost, ghost};
    std::vector<IBox> bx;
    std::vector<int> own;
    for (int b = 0; b < nboxes; ++b) { bx.push_back(IBox(boxes + 6 * b, boxes + 6 * b + 3)); own.push_back(owner ? owner[b] : 0); }
    ExchangePlan plan = build_exchange_plan(dom, per, g, bx, own, rank);
    auto dump = [&](const std::vector<CopyItem>& v, const std::vector<int>* peer_of, int* n, int* out) {
        SOMAR_CHECK((int)v.size() <= max_items, "max_items too small");
        *n = (int)v.size();
        for (size_t i = 0; i < v.size(); ++i) {
            int* o = out + 12 * i;
            o[0] = v[i].src_patch; o[1] = v[i].dst_patch;
            for (int d = 0; d < 3; ++d) { o[2 + d] = v[i].src_lo[d]; o[5 + d] = v[i].dst_lo[d]; o[8 + d] = v[i].n[d]; }
            o[11] = peer_of ? (*peer_of)[i] : rank;
        }
    };
    // peer of each remote item: items are grouped per peer in plan.peers order
    std::vector<int> sp, rp;
    for (size_t q = 0; q < plan.peers.size(); ++q) {
        long long s_end = plan.soff[q] + plan.scount[q], r_end = plan.roff[q] + plan.rcount[q];
        for (size_t i = 0; i < plan.send_items.size(); ++i)
            if (plan.send_itemoff[i] >= plan.soff[q] && plan.send_itemoff[i] < s_end) sp.push_back(plan.peers[q]);
        for (size_t i = 0; i < plan.recv_items.size(); ++i)
            if (plan.recv_itemoff[i] >= plan.roff[q] && plan.recv_itemoff[i] < r_end) rp.push_back(plan.peers[q]);
    }
    dump(plan.local, nullptr, n_local, local_items);
    dump(plan.send_items, &sp, n_send, send_items);
    dump(plan.recv_items, &rp, n_recv, recv_items);
    API_END
}

// ---- several AMR levels ---------------------------------------------------------------------------------
int somar_amr_create(somar_amr_t** out, int nlevels, const int* domain_lo, const int* domain_hi, const int* periodic,
                     const double* dx0, const int* bc_type, const int* ref_ratios, const int* nboxes,
                     const int* boxes, const int* owner, double alpha, double beta, const somar_params_t* prm,
                     void* comm)
{
    API_BEGIN
    SOMAR_CHECK(out && nlevels >= 1 && domain_lo && domain_hi && periodic && dx0 && bc_type && nboxes && boxes &&
                    (nlevels == 1 || ref_ratios),
                "null/empty argument");
    IBox dom(domain_lo, domain_hi);
    bool per[3] = {periodic[0] != 0, periodic[1] != 0, periodic[2] != 0};
    int bct[3][2] = {{bc_type[0], bc_type[1]}, {bc_type[2], bc_type[3]}, {bc_type[4], bc_type[5]}};
    std::vector<std::array<int, 3>> ratios;
    for (int l = 0; l + 1 < nlevels; ++l) ratios.push_back({ref_ratios[3 * l], ref_ratios[3 * l + 1], ref_ratios[3 * l + 2]});
    std::vector<std::vector<IBox>> bx(nlevels);
    std::vector<std::vector<int>> own(nlevels);
    int cursor = 0;
    for (int l = 0; l < nlevels; ++l) {
        SOMAR_CHECK(nboxes[l] > 0, "level without boxes");
        for (int b = 0; b < nboxes[l]; ++b, ++cursor) {
            IBox q(boxes + 6 * cursor, boxes + 6 * cursor + 3);
            SOMAR_CHECK(!q.empty(), "empty box");
            bx[l].push_back(q);
            own[l].push_back(owner ? owner[cursor] : 0);
        }
        for (size_t a = 0; a < bx[l].size(); ++a)
            for (size_t b = a + 1; b < bx[l].size(); ++b) SOMAR_CHECK((bx[l][a] & bx[l][b]).empty(), "boxes overlap");
    }
    for (const IBox& b : bx[0])
        for (int d = 0; d < 3; ++d) SOMAR_CHECK(b.lo[d] >= dom.lo[d] && b.hi[d] <= dom.hi[d], "box outside the domain");
    somar_amr* a = new somar_amr;
    try {
        a->amr = new AMRSolver(static_cast<Comm*>(comm));
        a->amr->define(dom, per, dx0, bct, ratios, bx, own, alpha, beta, to_params(prm));
        for (int l = 0; l < nlevels; ++l) {
            somar_solver* s = new somar_solver;
            s->ps = &a->amr->level(l);
            s->owned = false;
            SOMAR_HIP(hipEventCreate(&s->ev0));
            SOMAR_HIP(hipEventCreate(&s->ev1));
            a->levels.push_back(s);
        }
    } catch (...) {
        for (somar_solver* s : a->levels) delete s;
        delete a->amr;
        delete a;
        throw;
    }
    *out = a;
    API_END
}

int somar_amr_destroy(somar_amr_t* a)
{
    API_BEGIN
    if (a) {
        for (somar_solver* s : a->levels) {
            if (s->ev0) hipEventDestroy(s->ev0);
            if (s->ev1) hipEventDestroy(s->ev1);
            delete s;
        }
        delete a->amr;
        delete a;
    }
    API_END
}

int somar_amr_level(somar_amr_t* a, int level, somar_solver_t** out)
{
    API_BEGIN
    SOMAR_CHECK(a && out && level >= 0 && level < (int)a->levels.size(), "bad AMR level");
    *out = a->levels[level];
    API_END
}

int somar_amr_finalize(somar_amr_t* a)
{
    API_BEGIN
    a->amr->finalize();
    API_END
}

int somar_amr_solve(somar_amr_t* a, int l_max, int l_base, int zero_phi, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    a->amr->solve(l_max, l_base, zero_phi != 0, force_homogeneous != 0, st);
    fill_stats(st, stats);
    API_END
}

// AMREllipticSolver<LevelData<FArrayBox>>::solve on caller-owned host data of SEVERAL levels
// (AMREllipticSolver.H:33-48, as AMRPressureSolver::solve / levelSolve call it, AMRPressureSolver.cpp:494-594)
int somar_amr_solve_host(somar_amr_t* a, double* const* const* phi, const int* phi_ghost, const double* const* const* rhs,
                         const int* rhs_ghost, int l_max, int l_base, int zero_phi, int force_homogeneous,
                         somar_stats_t* stats)
{
    API_BEGIN
    AMRSolver& A = *a->amr;
    SOMAR_CHECK(phi && rhs && phi_ghost && rhs_ghost, "null pointer");
    SOMAR_CHECK(0 <= l_base && l_base <= l_max && l_max < A.nlevels(), "bad level range");
    for (int l = l_base; l <= l_max; ++l) SOMAR_CHECK(phi[l] && rhs[l], "phi / rhs of a solved level is null");
    if (l_base > 0) SOMAR_CHECK(phi[l_base - 1], "phi of level l_base - 1 supplies the coarse-fine values and must be given");
    for (int l = (l_base > 0 ? l_base - 1 : 0); l <= l_max; ++l) {
        PressureSolver& ps = A.level(l);
        const int np = ps.level(0).npatches();
        for (int p = 0; p < np; ++p) {
            if (l >= l_base) ps.upload_rhs(p, rhs[l][p], rhs_ghost);
            if (l < l_base || !zero_phi) ps.upload_phi(p, phi[l][p], phi_ghost);
        }
    }
    SolveStats st;
    A.solve(l_max, l_base, zero_phi != 0, force_homogeneous != 0, st);
    for (int l = l_base; l <= l_max; ++l) {
        PressureSolver& ps = A.level(l);
        const int np = ps.level(0).npatches();
        for (int p = 0; p < np; ++p) ps.download_phi(p, phi[l][p], phi_ghost);
    }
    fill_stats(st, stats);
    API_END
}

// ---- kernel-level box-by-box hook: one colour pass of GSRBITER3DORTHO on host FABs, Fortran argument shapes ---------
extern "C++" {
namespace {
struct HostFab {
    const double* p;
    IBox box;
    int ncomp;
};
HostFab mkfab(const double* p, const int* l0, const int* l1, const int* l2, const int* h0, const int* h1, const int* h2,
              const int* nc)
{
    HostFab f;
    f.p = p;
    const int lo[3] = {*l0, *l1, *l2}, hi[3] = {*h0, *h1, *h2};
    f.box = IBox(lo, hi);
    f.ncomp = nc ? *nc : 1;
    return f;
}
}  // namespace
}  // extern "C++"

int somar_k_gsrbiter3dortho(double* phi, const int* iphilo0, const int* iphilo1, const int* iphilo2, const int* iphihi0,
                            const int* iphihi1, const int* iphihi2, const int* nphicomp, const double* rhs,
                            const int* irhslo0, const int* irhslo1, const int* irhslo2, const int* irhshi0,
                            const int* irhshi1, const int* irhshi2, const int* nrhscomp, const double* Jgxx,
                            const int* iJgxxlo0, const int* iJgxxlo1, const int* iJgxxlo2, const int* iJgxxhi0,
                            const int* iJgxxhi1, const int* iJgxxhi2, const double* Jgyy, const int* iJgyylo0,
                            const int* iJgyylo1, const int* iJgyylo2, const int* iJgyyhi0, const int* iJgyyhi1,
                            const int* iJgyyhi2, const double* Jgzz, const int* iJgzzlo0, const int* iJgzzlo1,
                            const int* iJgzzlo2, const int* iJgzzhi0, const int* iJgzzhi1, const int* iJgzzhi2,
                            const double* Jinv, const int* iJinvlo0, const int* iJinvlo1, const int* iJinvlo2,
                            const int* iJinvhi0, const int* iJinvhi1, const int* iJinvhi2, const double* lapDiag,
                            const int* ilapDiaglo0, const int* ilapDiaglo1, const int* ilapDiaglo2, const int* ilapDiaghi0,
                            const int* ilapDiaghi1, const int* ilapDiaghi2, const int* iregionlo0, const int* iregionlo1,
                            const int* iregionlo2, const int* iregionhi0, const int* iregionhi1, const int* iregionhi2,
                            const double* dx, const double* alpha, const double* beta, const int* redBlack)
{
    API_BEGIN
    HostFab fphi = mkfab(phi, iphilo0, iphilo1, iphilo2, iphihi0, iphihi1, iphihi2, nphicomp);
    HostFab frhs = mkfab(rhs, irhslo0, irhslo1, irhslo2, irhshi0, irhshi1, irhshi2, nrhscomp);
    HostFab fjg[3] = {mkfab(Jgxx, iJgxxlo0, iJgxxlo1, iJgxxlo2, iJgxxhi0, iJgxxhi1, iJgxxhi2, nullptr),
                      mkfab(Jgyy, iJgyylo0, iJgyylo1, iJgyylo2, iJgyyhi0, iJgyyhi1, iJgyyhi2, nullptr),
                      mkfab(Jgzz, iJgzzlo0, iJgzzlo1, iJgzzlo2, iJgzzhi0, iJgzzhi1, iJgzzhi2, nullptr)};
    HostFab fji = mkfab(Jinv, iJinvlo0, iJinvlo1, iJinvlo2, iJinvhi0, iJinvhi1, iJinvhi2, nullptr);
    HostFab fld = mkfab(lapDiag, ilapDiaglo0, ilapDiaglo1, ilapDiaglo2, ilapDiaghi0, ilapDiaghi1, ilapDiaghi2, nullptr);
    const int rlo[3] = {*iregionlo0, *iregionlo1, *iregionlo2}, rhi[3] = {*iregionhi0, *iregionhi1, *iregionhi2};
    const IBox region(rlo, rhi);
    if (region.empty()) return 0;
    SOMAR_CHECK(fphi.ncomp == frhs.ncomp && fphi.ncomp >= 1, "phi and rhs must carry the same number of components");
    const int one[3] = {1, 1, 1};
    const IBox halo = region.grow(one);
    for (int d = 0; d < 3; ++d)
        SOMAR_CHECK(halo.lo[d] >= fphi.box.lo[d] && halo.hi[d] <= fphi.box.hi[d],
                    "GSRBITER3DORTHO reads phi one cell around the region: the phi FAB must contain it");
    // a one-box level whose domain lies well outside the region, so every cell takes the interior (full-stencil) form
    const int four[3] = {4, 4, 4};
    const IBox dom = region.grow(four);
    const bool per[3] = {false, false, false};
    const int bct[3][2] = {{BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}};
    Level L;
    Comm self;
    L.define(dom, per, dx, bct, std::vector<IBox>{region}, std::vector<int>{0}, &self);
    L.alloc_metric();
    L.alpha = *alpha;
    L.beta = *beta;
    L.refresh_params();
    hipStream_t st = nullptr;
    double* dphi = L.alloc_field();
    double* drhs = L.alloc_field();
    for (int d = 0; d < 3; ++d) {
        IBox fr = region;
        fr.hi[d] += 1;  // faces(region, d)
        L.upload(L.dev.jg[d], 0, fjg[d].p, fjg[d].box, fr, st);
    }
    L.upload(L.dev.jinv, 0, fji.p, fji.box, region, st);
    L.upload(L.dev.lapdiag, 0, fld.p, fld.box, region, st);
    for (int n = 0; n < fphi.ncomp; ++n) {
        double* hp = phi + (size_t)n * fphi.box.numPts();
        const double* hr = rhs + (size_t)n * frhs.box.numPts();
        L.upload(dphi, 0, hp, fphi.box, halo, st);
        L.upload(drhs, 0, hr, frhs.box, region, st);
        launch_gsrb_ortho(st, L.dev, dphi, drhs, *redBlack);
        L.download(dphi, 0, hp, fphi.box, region, st);
    }
    SOMAR_HIP(hipStreamSynchronize(st));
    Level::free_field(dphi);
    Level::free_field(drhs);
    API_END
}

// FILLMAPPEDLAPDIAG3D (AMRElliptic/MappedAMRPoissonOpF.ChF:233-274; prototype MappedAMRPoissonOpF_F.H:139-146) on host FABs
int somar_k_fillmappedlapdiag3d(double* lapDiag, const int* ilapDiaglo0, const int* ilapDiaglo1, const int* ilapDiaglo2,
                                const int* ilapDiaghi0, const int* ilapDiaghi1, const int* ilapDiaghi2, const double* Jg0,
                                const int* iJg0lo0, const int* iJg0lo1, const int* iJg0lo2, const int* iJg0hi0,
                                const int* iJg0hi1, const int* iJg0hi2, const int* nJg0comp, const double* Jg1,
                                const int* iJg1lo0, const int* iJg1lo1, const int* iJg1lo2, const int* iJg1hi0,
                                const int* iJg1hi1, const int* iJg1hi2, const int* nJg1comp, const double* Jg2,
                                const int* iJg2lo0, const int* iJg2lo1, const int* iJg2lo2, const int* iJg2hi0,
                                const int* iJg2hi1, const int* iJg2hi2, const int* nJg2comp, const double* Jinv,
                                const int* iJinvlo0, const int* iJinvlo1, const int* iJinvlo2, const int* iJinvhi0,
                                const int* iJinvhi1, const int* iJinvhi2, const int* iregionlo0, const int* iregionlo1,
                                const int* iregionlo2, const int* iregionhi0, const int* iregionhi1, const int* iregionhi2,
                                const double* dx)
{
    API_BEGIN
    HostFab fld = mkfab(lapDiag, ilapDiaglo0, ilapDiaglo1, ilapDiaglo2, ilapDiaghi0, ilapDiaghi1, ilapDiaghi2, nullptr);
    HostFab fjg[3] = {mkfab(Jg0, iJg0lo0, iJg0lo1, iJg0lo2, iJg0hi0, iJg0hi1, iJg0hi2, nJg0comp),
                      mkfab(Jg1, iJg1lo0, iJg1lo1, iJg1lo2, iJg1hi0, iJg1hi1, iJg1hi2, nJg1comp),
                      mkfab(Jg2, iJg2lo0, iJg2lo1, iJg2lo2, iJg2hi0, iJg2hi1, iJg2hi2, nJg2comp)};
    HostFab fji = mkfab(Jinv, iJinvlo0, iJinvlo1, iJinvlo2, iJinvhi0, iJinvhi1, iJinvhi2, nullptr);
    const int rlo[3] = {*iregionlo0, *iregionlo1, *iregionlo2}, rhi[3] = {*iregionhi0, *iregionhi1, *iregionhi2};
    const IBox region(rlo, rhi);
    if (region.empty()) return 0;
    const int four[3] = {4, 4, 4};
    const bool per[3] = {false, false, false};
    const int bct[3][2] = {{BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}};
    Level L;
    Comm self;
    L.define(region.grow(four), per, dx, bct, std::vector<IBox>{region}, std::vector<int>{0}, &self);
    L.alloc_metric();
    L.refresh_params();
    hipStream_t st = nullptr;
    for (int d = 0; d < 3; ++d) {
        SOMAR_CHECK(fjg[d].ncomp > d, "Jg FAB of direction d must carry component d (the FluxBox layout)");
        IBox fr = region;
        fr.hi[d] += 1;
        L.upload(L.dev.jg[d], 0, fjg[d].p + (size_t)d * fjg[d].box.numPts(), fjg[d].box, fr, st);   // component d = J g^{dd}
    }
    L.upload(L.dev.jinv, 0, fji.p, fji.box, region, st);
    launch_lapdiag(st, L.dev);
    L.download(L.dev.lapdiag, 0, lapDiag, fld.box, region, st);
    SOMAR_HIP(hipStreamSynchronize(st));
    API_END
}

// MAPPEDAVERAGE2 (MappedChombo/MappedCoarseAverageF.ChF:132-167; prototype MappedCoarseAverageF_F.H:111-117) on host FABs:
// the J-weighted restriction of restrictResidual, coarse(ic) = sum(fine / Jinv) / sum(1 / Jinv) over the refRatio block
int somar_k_mappedaverage2(double* coarse, const int* icoarselo0, const int* icoarselo1, const int* icoarselo2,
                           const int* icoarsehi0, const int* icoarsehi1, const int* icoarsehi2, const int* ncoarsecomp,
                           const double* fine, const int* ifinelo0, const int* ifinelo1, const int* ifinelo2,
                           const int* ifinehi0, const int* ifinehi1, const int* ifinehi2, const int* nfinecomp,
                           const double* fineCCJinv, const int* ifineCCJinvlo0, const int* ifineCCJinvlo1,
                           const int* ifineCCJinvlo2, const int* ifineCCJinvhi0, const int* ifineCCJinvhi1,
                           const int* ifineCCJinvhi2, const int* iboxlo0, const int* iboxlo1, const int* iboxlo2,
                           const int* iboxhi0, const int* iboxhi1, const int* iboxhi2, const int* refRatio,
                           const int* ibreflo0, const int* ibreflo1, const int* ibreflo2, const int* ibrefhi0,
                           const int* ibrefhi1, const int* ibrefhi2)
{
    API_BEGIN
    HostFab fc = mkfab(coarse, icoarselo0, icoarselo1, icoarselo2, icoarsehi0, icoarsehi1, icoarsehi2, ncoarsecomp);
    HostFab ff = mkfab(fine, ifinelo0, ifinelo1, ifinelo2, ifinehi0, ifinehi1, ifinehi2, nfinecomp);
    HostFab fj = mkfab(fineCCJinv, ifineCCJinvlo0, ifineCCJinvlo1, ifineCCJinvlo2, ifineCCJinvhi0, ifineCCJinvhi1,
                       ifineCCJinvhi2, nullptr);
    const int blo[3] = {*iboxlo0, *iboxlo1, *iboxlo2}, bhi[3] = {*iboxhi0, *iboxhi1, *iboxhi2};
    const IBox cbox(blo, bhi);
    if (cbox.empty()) return 0;
    const int r[3] = {refRatio[0], refRatio[1], refRatio[2]};
    const int brl[3] = {*ibreflo0, *ibreflo1, *ibreflo2}, brh[3] = {*ibrefhi0, *ibrefhi1, *ibrefhi2};
    for (int d = 0; d < 3; ++d) {
        SOMAR_CHECK(r[d] == 1 || r[d] == 2 || r[d] == 4, "refinement ratios are 1, 2 or 4 per direction");
        SOMAR_CHECK(brl[d] == 0 && brh[d] == r[d] - 1, "bref must be the refRatio block [0, refRatio - 1]");
    }
    SOMAR_CHECK(fc.ncomp == ff.ncomp, "coarse and fine must carry the same number of components");
    int flo[3], fhi[3];
    for (int d = 0; d < 3; ++d) { flo[d] = cbox.lo[d] * r[d]; fhi[d] = (cbox.hi[d] + 1) * r[d] - 1; }
    const IBox fbox(flo, fhi);
    const int four[3] = {4, 4, 4};
    const bool per[3] = {false, false, false};
    const int bct[3][2] = {{BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}, {BC_NEUM, BC_NEUM}};
    const double dx1[3] = {1.0, 1.0, 1.0};
    Level LF, LC;
    Comm self;
    int g4r[3] = {4 * r[0], 4 * r[1], 4 * r[2]};
    LF.define(fbox.grow(g4r), per, dx1, bct, std::vector<IBox>{fbox}, std::vector<int>{0}, &self);
    LC.define(cbox.grow(four), per, dx1, bct, std::vector<IBox>{cbox}, std::vector<int>{0}, &self);
    LF.alloc_metric();
    LC.alloc_metric();
    hipStream_t st = nullptr;
    LF.upload(LF.dev.jinv, 0, fj.p, fj.box, fbox, st);
    double* dfine = LF.alloc_field();
    double* dcrse = LC.alloc_field();
    for (int n = 0; n < fc.ncomp; ++n) {
        LF.upload(dfine, 0, fine + (size_t)n * ff.box.numPts(), ff.box, fbox, st);
        launch_restrict(st, LC.dev, LF.dev, dcrse, dfine, r);
        LC.download(dcrse, 0, coarse + (size_t)n * fc.box.numPts(), fc.box, cbox, st);
    }
    SOMAR_HIP(hipStreamSynchronize(st));
    Level::free_field(dfine);
    Level::free_field(dcrse);
    API_END
}

int somar_solver_set_cc_j(somar_solver_t* s, int patch, const double* J, const double* Jinv, const int* ghost)
{
    API_BEGIN
    SOMAR_CHECK(J && Jinv && ghost, "null pointer");
    s->ps->set_scale_cc(0, patch, J, ghost);
    s->ps->set_scale_cc(1, patch, Jinv, ghost);
    API_END
}

int somar_solver_set_face_j(somar_solver_t* s, int dir, int patch, const double* J, const double* Jinv)
{
    API_BEGIN
    SOMAR_CHECK(J && Jinv, "null pointer");
    s->ps->set_scale_face(0, dir, patch, J);
    s->ps->set_scale_face(1, dir, patch, Jinv);
    API_END
}

int somar_vel_mult_by_j(somar_solver_t* s, int centring)
{
    API_BEGIN
    s->ps->scale_vel(centring, 0);
    API_END
}

int somar_vel_div_by_j(somar_solver_t* s, int centring)
{
    API_BEGIN
    s->ps->scale_vel(centring, 1);
    API_END
}

int somar_amr_set_inspector(somar_amr_t* a, somar_inspector_fn fn, void* user)
{
    API_BEGIN
    a->amr->set_inspector(fn, user);
    API_END
}

int somar_amr_cc_project(somar_amr_t* a, int l_min, int l_max, double dt, int zero_pressure, int force_homogeneous,
                         int wall_bc, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    a->amr->cc_project(l_min, l_max, dt, zero_pressure != 0, force_homogeneous != 0, wall_bc != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_amr_comp_divergence_cc(somar_amr_t* a, int level, int l_max, int out_field, int wall_bc)
{
    API_BEGIN
    SOMAR_CHECK(level >= 0 && level < a->amr->nlevels(), "bad level");
    int d0;
    double* out = field_ptr(a->levels[level], out_field, &d0);
    SOMAR_CHECK(d0 == 0, "the divergence is a depth-0 field");
    a->amr->comp_divergence_cc(level, l_max, out, wall_bc != 0);
    a->amr->sync();
    API_END
}

int somar_amr_comp_grad_correct_cc(somar_amr_t* a, int level, int l_max, int phi_field, double dt)
{
    API_BEGIN
    SOMAR_CHECK(level >= 0 && level < a->amr->nlevels(), "bad level");
    int d0;
    double* phi = field_ptr(a->levels[level], phi_field, &d0);
    SOMAR_CHECK(d0 == 0, "phi is a depth-0 field");
    a->amr->comp_grad_correct_cc(level, l_max, phi, dt);
    a->amr->sync();
    API_END
}

int somar_amr_average_down_ccvel(somar_amr_t* a, int level)
{
    API_BEGIN
    a->amr->average_down_ccvel(level);
    a->amr->sync();
    API_END
}

int somar_amr_level_project(somar_amr_t* a, int level, int centring, double dt, int zero_pressure, int force_homogeneous,
                            int wall_bc, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    a->amr->level_project(level, centring, dt, zero_pressure != 0, force_homogeneous != 0, wall_bc != 0, st);
    fill_stats(st, stats);
    API_END
}

static double* amr_field(somar_amr* a, int level, int field)
{
    SOMAR_CHECK(level >= 0 && level < (int)a->levels.size(), "bad AMR level");
    SOMAR_CHECK((field >> 8) == 0, "AMR operations act on depth-0 fields");
    return field_ptr(a->levels[level], field);
}

int somar_amr_interp_cf(somar_amr_t* a, int level, int fine_field, int coarse_field)
{
    API_BEGIN
    SOMAR_CHECK(level >= 1, "level 0 has no coarser level");
    a->amr->interp_cf(level, amr_field(a, level, fine_field), amr_field(a, level - 1, coarse_field));
    a->amr->sync();
    API_END
}

int somar_amr_residual_level(somar_amr_t* a, int l_max, int l_base, int ilev, int res_field, int phi_field,
                             int rhs_field)
{
    API_BEGIN
    const int n = a->amr->nlevels();
    SOMAR_CHECK(0 <= l_base && l_base <= ilev && ilev <= l_max && l_max < n, "bad level range");
    std::vector<double*> res(n, nullptr), phi(n, nullptr), rhs(n, nullptr);
    for (int l = (l_base > 0 ? l_base - 1 : 0); l <= l_max; ++l) {
        phi[l] = amr_field(a, l, phi_field);
        if (l >= l_base) { res[l] = amr_field(a, l, res_field); rhs[l] = amr_field(a, l, rhs_field); }
    }
    a->amr->compute_residual_level(res.data(), phi.data(), rhs.data(), l_max, l_base, ilev, true);
    a->amr->sync();
    API_END
}

int somar_amr_zero_covered(somar_amr_t* a, int level, int field)
{
    API_BEGIN
    SOMAR_CHECK(level >= 0 && level + 1 < a->amr->nlevels(), "level has no finer level");
    a->amr->zero_covered(level, amr_field(a, level, field));
    a->amr->sync();
    API_END
}

int somar_amr_vcycle(somar_amr_t* a, int l_max, int l_base)
{
    API_BEGIN
    const int n = a->amr->nlevels();
    SOMAR_CHECK(0 <= l_base && l_base <= l_max && l_max < n, "bad level range");
    std::vector<double*> corr(n, nullptr), res(n, nullptr);
    for (int l = (l_base > 0 ? l_base - 1 : 0); l <= l_max; ++l) {
        corr[l] = amr_field(a, l, SOMAR_F_CORR);
        res[l] = amr_field(a, l, SOMAR_F_RES);
    }
    a->amr->vcycle(corr.data(), res.data(), l_max, l_max, l_base);
    a->amr->sync();
    API_END
}

int somar_comm_unique_id(unsigned char* id128)
{
    API_BEGIN
    rccl_unique_id(id128);
    API_END
}

int somar_comm_create(void** comm, const unsigned char* id128, int rank, int nranks, int device)
{
    API_BEGIN
    *comm = rccl_create(id128, rank, nranks, device);
    API_END
}

int somar_comm_create_shm(void** comm, const char* name, int rank, int nranks, long long outbox_bytes)
{
    API_BEGIN
    SOMAR_CHECK(comm && name && outbox_bytes > 0, "null/empty argument");
    *comm = shm_create(name, rank, nranks, (size_t)outbox_bytes);
    API_END
}

// ---- AlteredMetric::fill_Jgup (implicit-gravity / Coriolis projection metric) ----
int somar_altered_jgup(long long n, double* dest, const double* nsq_fc, const double* dximu_dz, const double* dxinu_dz,
                       const double* ix, const double* jy, const double* iy, const double* jx, const double* gup,
                       const double* J, double dt_theta, double coriolis_f)
{
    API_BEGIN
    SOMAR_CHECK(n > 0 && dest && nsq_fc && dximu_dz && dxinu_dz && gup && J, "null/empty argument");
    const bool offdiag = ix || jy || iy || jx;
    SOMAR_CHECK(!offdiag || (ix && jy && iy && jx), "the four horizontal map derivatives come together (mu != nu) or not at all");
    const int nin = offdiag ? 9 : 5;
    const double* in[9] = {nsq_fc, dximu_dz, dxinu_dz, gup, J, ix, jy, iy, jx};
    double* dev = nullptr;
    SOMAR_HIP(hipMalloc(&dev, (size_t)(nin + 1) * n * sizeof(double)));
    try {
        for (int q = 0; q < nin; ++q)
            SOMAR_HIP(hipMemcpy(dev + (size_t)(q + 1) * n, in[q], (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        double* d[10];
        for (int q = 0; q <= 9; ++q) d[q] = q <= nin ? dev + (size_t)q * n : nullptr;
        SOMAR_HIP(hipDeviceSynchronize());
        launch_altered_jgup(nullptr, n, d[0], d[1], d[2], d[3], d[6], d[7], d[8], d[9], d[4], d[5], dt_theta, coriolis_f, offdiag);
        SOMAR_HIP(hipDeviceSynchronize());
        SOMAR_HIP(hipMemcpy(dest, dev, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    } catch (...) {
        hipFree(dev);
        throw;
    }
    hipFree(dev);
    API_END
}

// GeoSourceInterface::fill_Jgup's generic algebra on the device: host arrays in, host array out
int somar_metric_jgup_from_dxdxi(long long n, int mu, const double* dxdxi9, const double* detJ, double scale, double* jgup3)
{
    API_BEGIN
    SOMAR_CHECK(n > 0 && mu >= 0 && mu < 3 && dxdxi9 && detJ && jgup3, "null/empty argument");
    double* dev = nullptr;
    SOMAR_HIP(hipMalloc(&dev, (size_t)13 * n * sizeof(double)));
    try {
        SOMAR_HIP(hipMemcpy(dev, dxdxi9, (size_t)9 * n * sizeof(double), hipMemcpyHostToDevice));
        SOMAR_HIP(hipMemcpy(dev + (size_t)9 * n, detJ, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        const double* x9[9];
        for (int q = 0; q < 9; ++q) x9[q] = dev + (size_t)q * n;
        launch_jgup_from_dxdxi(nullptr, n, mu, x9, dev + (size_t)9 * n, scale, dev + (size_t)10 * n);
        SOMAR_HIP(hipDeviceSynchronize());
        SOMAR_HIP(hipMemcpy(jgup3, dev + (size_t)10 * n, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
    } catch (...) {
        hipFree(dev);
        throw;
    }
    hipFree(dev);
    API_END
}

// CartesianMap::fill_Jgup / fill_Jinv (geometry/maps/CartesianMap.cpp:230-280): constants, written on the device
int somar_solver_set_metric_map(somar_solver_t* s, int kind, const double* L, const double* depth, const int* depth_lo,
                                const int* depth_n)
{
    API_BEGIN
    SOMAR_CHECK(s && L, "null argument");
    const int z[2] = {0, 0};
    s->ps->set_metric_map(kind, L, depth, depth_lo ? depth_lo : z, depth_n ? depth_n : z);
    API_END
}

int somar_solver_set_metric_uniform(somar_solver_t* s, const double* c4)
{
    API_BEGIN
    SOMAR_CHECK(s && c4, "null argument");
    s->ps->set_metric_uniform(c4);
    API_END
}

// Diagnostics: the streaming rate of this device for a given stream mix, GB/s of algorithmic bytes (kind 0: 16 B/cell,
// 1: 8 B/cell, 2: 56 B/cell), best of the workgroup counts tried.  bench.py reports kind 2 next to the fused sweep's rate.
int somar_diag_stream_probe(int kind, long long cells, int reps, double* gbs)
{
    API_BEGIN
    SOMAR_CHECK(gbs && kind >= 0 && kind <= 2 && cells >= 1024 && cells % 2 == 0 && reps >= 1, "bad argument");
    double* buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int nin = kind == 2 ? 6 : 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    try {
        for (int q = 0; q < nin; ++q) {
            SOMAR_HIP(hipMalloc(&buf[q], cells * sizeof(double)));
            SOMAR_HIP(hipMemset(buf[q], 0, cells * sizeof(double)));
        }
        for (int q = nin; q < 6; ++q) buf[q] = buf[0];
        SOMAR_HIP(hipMalloc(&buf[6], cells * sizeof(double)));
        SOMAR_HIP(hipMemset(buf[6], 0, cells * sizeof(double)));
        SOMAR_HIP(hipEventCreate(&e0));
        SOMAR_HIP(hipEventCreate(&e1));
        const double bytes = (kind == 0 ? 16.0 : kind == 1 ? 8.0 : 56.0) * (double)cells;
        double best = 0.0;
        for (int wgs : {1024, 2048, 8192, 32768}) {
            for (int w = 0; w < 2; ++w) launch_stream_probe(nullptr, kind, wgs, buf, buf[6], cells);
            SOMAR_HIP(hipDeviceSynchronize());
            SOMAR_HIP(hipEventRecord(e0, nullptr));
            for (int r = 0; r < reps; ++r) launch_stream_probe(nullptr, kind, wgs, buf, buf[6], cells);
            SOMAR_HIP(hipEventRecord(e1, nullptr));
            SOMAR_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            SOMAR_HIP(hipEventElapsedTime(&ms, e0, e1));
            best = std::max(best, bytes * reps / (ms * 1e-3) * 1e-9);
        }
        *gbs = best;
    } catch (...) {
        for (int q = 0; q < nin; ++q) hipFree(buf[q]);
        hipFree(buf[6]);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
        throw;
    }
    for (int q = 0; q < nin; ++q) hipFree(buf[q]);
    hipFree(buf[6]);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    API_END
}

// ---- leptic level solver -----------------------------------------------------------------------------------
int somar_leptic_params_default(somar_leptic_params_t* p)
{
    API_BEGIN
    SOMAR_CHECK(p, "null argument");
    LepticParams d;
    p->max_order = d.maxOrder; p->norm_type = d.normType;
    p->hang = d.hang; p->horiz_rhs_tol = d.horizRhsTol; p->domain_height = d.domainHeight;
    from_params(d.horiz, &p->horiz);
    from_params(d.full, &p->full);
    API_END
}

static LepticParams to_leptic_params(const somar_leptic_params_t* lp)
{
    LepticParams P;
    if (lp) {
        P.maxOrder = lp->max_order; P.normType = lp->norm_type;
        P.hang = lp->hang; P.horizRhsTol = lp->horiz_rhs_tol; P.domainHeight = lp->domain_height;
        P.horiz = to_params(&lp->horiz);
        P.full = to_params(&lp->full);
    }
    return P;
}

static void fill_leptic_stats(const LepticStats& S, somar_leptic_stats_t* stats)
{
    std::memset(stats, 0, sizeof(*stats));
    stats->exit_status = S.exitStatus;
    stats->orders = S.orders;
    stats->horiz_solves = S.horizSolves;
    stats->used_full_solver = S.usedFullSolver;
    stats->nres = (int)std::min<size_t>(S.resNorms.size(), SOMAR_MAX_HISTORY);
    for (int i = 0; i < stats->nres; ++i) stats->res_norms[i] = S.resNorms[i];
    fill_stats(S.horizStats, &stats->horiz);
    fill_stats(S.fullStats, &stats->full);
}

// ---- level heat integrators on a level of a hierarchy (AMRParabolic/*.cpp) -------------------------------------------------
int somar_amr_set_alpha_beta(somar_amr_t* a, double alpha, double beta)
{
    API_BEGIN
    SOMAR_CHECK(a, "null argument");
    a->amr->set_alpha_beta(alpha, beta);
    API_END
}

int somar_amr_heat_step(somar_amr_t* a, int level, int scheme, double dt, int zero_phi, double old_time, double crse_old_time,
                        double crse_new_time, somar_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(a, "null argument");
    SolveStats st;
    a->amr->heat_step(level, scheme, dt, zero_phi != 0, old_time, crse_old_time, crse_new_time, st);
    fill_stats(st, stats);
    API_END
}

int somar_amr_tga_step(somar_amr_t* a, int l_max, int l_base, double dt, somar_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(a, "null argument");
    SolveStats st;
    a->amr->tga_step(l_max, l_base, dt, st);
    fill_stats(st, stats);
    API_END
}

int somar_heat_flux_download(somar_solver_t* s, int dir, int patch, double* host)
{
    API_BEGIN
    SOMAR_CHECK(s && host, "null pointer");
    s->ps->download_heat_flux(dir, patch, host);
    API_END
}

// ---- AMRLepticSolver on an AMR hierarchy (AMRLepticSolver.cpp) -------------------------------------------------------------
int somar_amr_enable_leptic(somar_amr_t* a, const somar_leptic_params_t* lp, int base_from_restricted)
{
    API_BEGIN
    SOMAR_CHECK(a, "null argument");
    a->amr->enable_leptic(to_leptic_params(lp), base_from_restricted != 0);
    API_END
}

int somar_amr_solve_leptic(somar_amr_t* a, int l_max, int l_base, int zero_phi, int force_homogeneous, somar_stats_t* stats)
{
    API_BEGIN
    SolveStats st;
    a->amr->solve_leptic(l_max, l_base, zero_phi != 0, force_homogeneous != 0, st);
    fill_stats(st, stats);
    API_END
}

int somar_amr_leptic_stats(somar_amr_t* a, int level, somar_leptic_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(a && stats && level >= 0 && level < a->amr->nlevels(), "bad argument");
    fill_leptic_stats(a->amr->leptic_stats(level), stats);
    API_END
}

int somar_leptic_create(somar_leptic_t** out, const int* domain_lo, const int* domain_hi, const int* periodic,
                        const double* dx, const int* bc_type, int nboxes, const int* boxes, const int* owner,
                        double alpha, double beta, const somar_params_t* level_prm, const somar_leptic_params_t* lp,
                        void* comm)
{
    API_BEGIN
    SOMAR_CHECK(out && domain_lo && domain_hi && periodic && dx && bc_type && boxes && nboxes > 0, "null/empty argument");
    IBox dom(domain_lo, domain_hi);
    bool per[3] = {periodic[0] != 0, periodic[1] != 0, periodic[2] != 0};
    int bct[3][2] = {{bc_type[0], bc_type[1]}, {bc_type[2], bc_type[3]}, {bc_type[4], bc_type[5]}};
    std::vector<IBox> bx;
    std::vector<int> own;
    for (int b = 0; b < nboxes; ++b) {
        IBox q(boxes + 6 * b, boxes + 6 * b + 3);
        SOMAR_CHECK(!q.empty(), "empty box");
        bx.push_back(q);
        own.push_back(owner ? owner[b] : 0);
    }
    for (size_t a = 0; a < bx.size(); ++a) {
        for (int d = 0; d < 3; ++d)
            SOMAR_CHECK(bx[a].lo[d] >= dom.lo[d] && bx[a].hi[d] <= dom.hi[d], "box outside the domain");
        for (size_t b = a + 1; b < bx.size(); ++b) SOMAR_CHECK((bx[a] & bx[b]).empty(), "boxes overlap");
    }
    const LepticParams P = to_leptic_params(lp);
    somar_leptic* h = new somar_leptic;
    try {
        h->lep = new LepticSolver(static_cast<Comm*>(comm));
        h->lep->define(dom, per, dx, bct, bx, own, alpha, beta, to_params(level_prm), P);
        h->level = new somar_solver;
        h->level->ps = &h->lep->orig();
        h->level->owned = false;
        SOMAR_HIP(hipEventCreate(&h->level->ev0));
        SOMAR_HIP(hipEventCreate(&h->level->ev1));
        for (int q = 0; q < 2; ++q) {
            h->parts[q] = new somar_solver;
            h->parts[q]->ps = q == 0 ? &h->lep->vert() : h->lep->horiz_ptr();
            h->parts[q]->owned = false;
            SOMAR_HIP(hipEventCreate(&h->parts[q]->ev0));
            SOMAR_HIP(hipEventCreate(&h->parts[q]->ev1));
        }
    } catch (...) {
        delete h->parts[0];
        delete h->parts[1];
        delete h->level;
        delete h->lep;
        delete h;
        throw;
    }
    *out = h;
    API_END
}

int somar_leptic_destroy(somar_leptic_t* h)
{
    API_BEGIN
    if (h) {
        for (somar_solver* q : {h->level, h->parts[0], h->parts[1]}) {
            if (!q) continue;
            if (q->ev0) hipEventDestroy(q->ev0);
            if (q->ev1) hipEventDestroy(q->ev1);
            delete q;
        }
        delete h->lep;
        delete h;
    }
    API_END
}

int somar_leptic_level(somar_leptic_t* h, somar_solver_t** out)
{
    API_BEGIN
    SOMAR_CHECK(h && out, "null argument");
    *out = h->level;
    API_END
}

int somar_leptic_part(somar_leptic_t* h, int which, somar_solver_t** out)
{
    API_BEGIN
    SOMAR_CHECK(h && out && (which == 1 || which == 2), "which: 1 (J-scaled 3-D solver) or 2 (flat solver)");
    SOMAR_CHECK(h->parts[which - 1]->ps, "this layout has no flat problem: no column is Neumann-Neumann "
                                         "(gatherVerticalBCTypes switched the horizontal solves off)");
    *out = h->parts[which - 1];
    API_END
}

int somar_leptic_finalize(somar_leptic_t* h)
{
    API_BEGIN
    SOMAR_CHECK(h, "null argument");
    h->lep->finalize();
    API_END
}

int somar_leptic_solve(somar_leptic_t* h, int homogeneous, somar_leptic_stats_t* stats)
{
    API_BEGIN
    SOMAR_CHECK(h, "null argument");
    LepticStats S;
    h->lep->solve(homogeneous != 0, S);
    if (stats) fill_leptic_stats(S, stats);
    API_END
}

// Pushes real traffic through a transport: an all-reduce (sum and max) of rank-dependent values and a ring
// neighbour exchange (rank -> rank+1; a self send/recv on one rank).  Returns an error if any value is wrong.
int somar_comm_selftest(void* comm)
{
    API_BEGIN
    SOMAR_CHECK(comm, "null communicator");
    Comm* c = static_cast<Comm*>(comm);
    const int n = 4096;
    hipStream_t st;
    SOMAR_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *d_red = nullptr, *d_send = nullptr, *d_recv = nullptr;
    SOMAR_HIP(hipMalloc(&d_red, 2 * sizeof(double)));
    SOMAR_HIP(hipMalloc(&d_send, n * sizeof(double)));
    SOMAR_HIP(hipMalloc(&d_recv, n * sizeof(double)));
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.0 * c->rank + i;
    SOMAR_HIP(hipMemcpyAsync(d_send, h.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    SOMAR_HIP(hipMemsetAsync(d_recv, 0, n * sizeof(double), st));
    double hr[2] = {double(c->rank + 1), double(c->rank + 1)};
    SOMAR_HIP(hipMemcpyAsync(d_red, hr, 2 * sizeof(double), hipMemcpyHostToDevice, st));
    SOMAR_HIP(hipStreamSynchronize(st));
    c->allreduce_raw(d_red, 1, 0, st);
    c->allreduce_raw(d_red + 1, 1, 1, st);
    const int nxt = (c->rank + 1) % c->size, prv = (c->rank + c->size - 1) % c->size;
    if (nxt == prv) {
        c->neighbor_exchange(d_send, d_recv, {nxt}, {0}, {n}, {0}, {n}, st);
    } else {
        // peers in ascending order, as the exchange plans list them
        const bool nf = nxt < prv;
        c->neighbor_exchange(d_send, d_recv, {nf ? nxt : prv, nf ? prv : nxt}, {0, 0},
                             {nf ? (long long)n : 0, nf ? 0 : (long long)n}, {0, 0},
                             {nf ? 0 : (long long)n, nf ? (long long)n : 0}, st);
    }
    std::vector<double> g(n);
    SOMAR_HIP(hipMemcpyAsync(g.data(), d_recv, n * sizeof(double), hipMemcpyDeviceToHost, st));
    SOMAR_HIP(hipMemcpyAsync(hr, d_red, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    SOMAR_HIP(hipStreamSynchronize(st));
    hipFree(d_red);
    hipFree(d_send);
    hipFree(d_recv);
    hipStreamDestroy(st);
    SOMAR_CHECK(hr[0] == 0.5 * c->size * (c->size + 1), "comm selftest: all-reduce(sum) returned a wrong value");
    SOMAR_CHECK(hr[1] == double(c->size), "comm selftest: all-reduce(max) returned a wrong value");
    for (int i = 0; i < n; ++i)
        SOMAR_CHECK(g[i] == 1000.0 * prv + i, "comm selftest: neighbour exchange delivered wrong data");
    API_END
}

int somar_comm_destroy(void* comm)
{
    API_BEGIN
    delete static_cast<Comm*>(comm);
    API_END
}

}  // extern "C"
