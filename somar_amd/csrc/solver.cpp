// somar_amd/csrc/solver.cpp -- see solver.h for the reference map.
#include "solver.h"

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace somar {

static const int S_MAX_COARSE = 4;  // MappedAMRPoissonOp::s_maxCoarse, MappedAMRPoissonOp.cpp:55
enum { SLOT_TMP = 0, SLOT_SUMS = 8, NSLOTS = 16 };

PressureSolver::PressureSolver(Comm* comm, hipStream_t shared) : comm_(comm ? comm : &self_)
{
    if (shared) { st_ = shared; own_stream_ = false; }
    else SOMAR_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    SOMAR_HIP(hipMalloc(&d_scalars, NSLOTS * sizeof(double)));
    SOMAR_HIP(hipMemset(d_scalars, 0, NSLOTS * sizeof(double)));
    SOMAR_HIP(hipDeviceSynchronize());
    SOMAR_HIP(hipHostMalloc(&h_scalars, (NSLOTS + 2) * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));
    h_seq_ = reinterpret_cast<unsigned long long*>(h_scalars + NSLOTS);
    *h_seq_ = 0;
    // levels smaller than this use the two-pass colour kernel (launch-latency bound anyway);
    // SOMAR_FUSED_MIN_CELLS=0 forces the fused sweep everywhere (tests), a huge value disables it.
    if (const char* e = getenv("SOMAR_FUSED_MIN_CELLS")) fused_min_cells_ = atoll(e);
    march_min_cells_ = fused_min_cells_;
    if (const char* e = getenv("SOMAR_MARCH_MIN_CELLS")) march_min_cells_ = atoll(e);
    // levels up to this many cells sum in the reference's serial order (k_reduce_ordered); tests raise it to
    // make whole solves reproduce the oracle's histories to the last bits
    if (const char* e = getenv("SOMAR_ORDERED_REDUCE_MAX")) ordered_max_cells_ = atoll(e);
    if (const char* e = getenv("SOMAR_FUSED_BOTTOM_MAX_CELLS")) fused_bottom_max_ = atoll(e);
    if (const char* e = getenv("SOMAR_BOX_BOTTOM")) box_bottom_on_ = atoi(e) != 0;
    if (const char* e = getenv("SOMAR_GHOST_STAGED")) ghost_box_on_ = atoi(e) == 0;
    if (const char* e = getenv("SOMAR_NO_OVERLAP")) overlap_on_ = atoi(e) == 0;
    if (const char* e = getenv("SOMAR_BOX_BOTTOM_MIN_CELLS")) box_min_cells_ = atoll(e);
    if (const char* e = getenv("SOMAR_AGGLOM_CELLS")) agglom_cells_ = atoll(e);
    if (const char* e = getenv("SOMAR_GRAPH_CELLS")) graph_cells_ = atoll(e);
}

PressureSolver::~PressureSolver()
{
    drop_graphs();
    for (GhostOp* q : d_diri_ops_) hipFree(q);
    for (double* q : f_flux) Level::free_field(q);
    for (double* q : f_ccvel) Level::free_field(q);
    for (double* q : f_heat) Level::free_field(q);
    hipFree(d_extrapbc_ops_);
    if (st_) hipStreamSynchronize(st_);
    if (st_comm_) { hipStreamSynchronize(st_comm_); hipStreamDestroy(st_comm_); }
    if (ev_ready_) hipEventDestroy(ev_ready_);
    if (ev_done_) hipEventDestroy(ev_done_);
    for (int w = 0; w < 2; ++w) { hipFree(f_sc_cc[w]); for (int d = 0; d < 3; ++d) hipFree(f_sc_face[w][d]); }
    for (double* f : f_res) hipFree(f);
    for (double* f : f_corr) hipFree(f);
    for (double* f : f_scratch) hipFree(f);
    for (double* f : f_pp) hipFree(f);
    for (double* f : f_vel) hipFree(f);
    for (double* f : f_amr) hipFree(f);
    for (double* f : f_psi) hipFree(f);
    for (double* f : f_W) hipFree(f);
    hipFree(d_fold);
    for (FullProgram& q : aux_prog_) free_program(q);
    for (double* f : f_heatflux) Level::free_field(f);
    for (auto& pr : full_prog_)
        for (auto& q : pr) free_program(q);
    for (auto& L : lev)
        for (int d = 0; d < 3; ++d)
            for (int c = 0; c < 3; ++c)
                if (c != d && L->dev.jgf[d][c]) { hipFree(L->dev.jgf[d][c]); L->dev.jgf[d][c] = nullptr; }
    hipFree(f_phi); hipFree(f_rhs); hipFree(f_uberRes); hipFree(f_uberCorr); hipFree(f_best);
    for (double* f : bicg) hipFree(f);
    hipFree(d_box_nb_); hipFree(d_box_cstart_); hipFree(d_box_sums_); hipFree(d_box_sync_); hipFree(d_box_fab_); hipFree(d_box_fabstart_);
    for (int w = 0; w < 2; ++w) { hipFree(d_box_ent_[w]); hipFree(d_box_entfirst_[w]); hipFree(d_box_stg_[w]); hipFree(d_box_stgfirst_[w]); hipFree(d_box_nfg_[w]); hipFree(d_box_nfgfirst_[w]); }
    hipFree(d_partials);
    hipFree(d_scalars);
    for (Prof& p : prof_) {
        for (hipEvent_t e : p.a) hipEventDestroy(e);
        for (hipEvent_t e : p.b) hipEventDestroy(e);
    }
    if (h_scalars) hipHostFree(h_scalars);
    hipFree(d_agglom_back_);
    coarse_.reset();
    lev.clear();
    if (st_ && own_stream_) hipStreamDestroy(st_);
}

void PressureSolver::sync() { SOMAR_HIP(hipStreamSynchronize(st_)); }

void PressureSolver::profile_enable(bool on)
{
    const int POOL = 4096;
    if (on)
        for (Prof& p : prof_) {
            while ((int)p.a.size() < POOL) {
                hipEvent_t e0, e1;
                SOMAR_HIP(hipEventCreate(&e0));
                SOMAR_HIP(hipEventCreate(&e1));
                p.a.push_back(e0);
                p.b.push_back(e1);
            }
            p.used = 0;
        }
    profiling_ = on;
}
void PressureSolver::prof_begin(int k)
{
    Prof& p = prof_[k];
    if (p.used < (int)p.a.size()) SOMAR_HIP(hipEventRecord(p.a[p.used], st_));
}
void PressureSolver::prof_end(int k)
{
    Prof& p = prof_[k];
    if (p.used < (int)p.a.size()) { SOMAR_HIP(hipEventRecord(p.b[p.used], st_)); ++p.used; }
}
void PressureSolver::profile_get(int kernel, int* count, double* total_ms)
{
    SOMAR_CHECK(kernel >= 0 && kernel < 4, "profile id: 0 GSRB, 1 operator / residual, 2 remote ghost exchanges, 3 replicated tail");
    sync();
    Prof& p = prof_[kernel];
    double tot = 0.0;
    for (int i = 0; i < p.used; ++i) {
        float ms = 0.f;
        SOMAR_HIP(hipEventElapsedTime(&ms, p.a[i], p.b[i]));
        tot += ms;
    }
    *count = p.used;
    *total_ms = tot;
    p.used = 0;
}

void PressureSolver::xchg(const Level& L, double* f)
{
    const bool timed = profiling_ && !L.plan.peers.empty();
    if (timed) prof_begin(2);
    L.exchange(f, st_);
    if (timed) prof_end(2);
}

double PressureSolver::fetch_scalar(int slot)
{
    fetch_scalars(slot, 1);
    return h_scalars[slot];
}

// d_scalars[slot .. slot+n) -> h_scalars[slot .. slot+n).  The stopping tests of BiCGStab and of the outer iteration need
// a handful of scalars per V-cycle on the host; a copy + stream synchronize costs ~25 us each.  Instead a one-thread
// kernel stores the values into coherent host memory followed by a sequence number, and the host spins on that
// number: the round trip drops to the latency of a PCIe write (SOMAR_POLL_FETCH=0 restores copy + synchronize).
void PressureSolver::fetch_scalars(int slot, int n)
{
    static const bool poll = !(getenv("SOMAR_POLL_FETCH") && atoi(getenv("SOMAR_POLL_FETCH")) == 0);
    if (!poll || capturing_) {
        SOMAR_HIP(hipMemcpyAsync(h_scalars + slot, d_scalars + slot, n * sizeof(double), hipMemcpyDeviceToHost, st_));
        SOMAR_HIP(hipStreamSynchronize(st_));
        return;
    }
    const unsigned long long want = ++fetch_seq_;
    launch_publish(st_, d_scalars + slot, n, h_scalars + slot, h_seq_, want);
    wait_published(want);
}

// the host side of a published scalar: spin on the sequence number the device stores last
void PressureSolver::wait_published(unsigned long long want)
{
    unsigned long long spins = 0;
    while (__atomic_load_n(h_seq_, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0xfffff) == 0) {  // every ~1M spins: has the stream died or drained without publishing?
            const hipError_t q = hipStreamQuery(st_);
            if (q == hipSuccess) {
                if (__atomic_load_n(h_seq_, __ATOMIC_ACQUIRE) == want) break;
                throw Error(-2, "scalar publish kernel finished without publishing");
            }
            if (q != hipErrorNotReady) SOMAR_HIP(q);
        }
    }
}

double* PressureSolver::work(int which)
{
    return which == 0 ? f_uberRes : (which == 1 ? f_uberCorr : f_best);
}

double* PressureSolver::field(int depth, int which)
{
    if (depth < 0 || depth >= (int)lev.size() || !finalized) return nullptr;
    switch (which) {
        case 0: return depth == 0 ? f_phi : nullptr;
        case 1: return depth == 0 ? f_rhs : nullptr;
        case 2: return depth == 0 ? f_uberRes : f_res[depth];
        case 3: return depth == 0 ? f_uberCorr : f_corr[depth];
        case 4: return depth == 0 ? f_best : nullptr;
        case 5: return f_scratch[depth];
        case 6: return depth == 0 ? amr_field(0) : nullptr;
        case 7: return depth == 0 ? amr_field(1) : nullptr;
        case 8: return depth == 0 ? heat_field(0) : nullptr;
        case 9: return depth == 0 ? heat_field(1) : nullptr;
        case 10: case 11: case 12: return (depth == 0 && which - 10 < prm.spaceDim) ? heat_flux(which - 10) : nullptr;
        default: return nullptr;
    }
}

// ------------------------------------------------------------------------------------
// definition
// ------------------------------------------------------------------------------------
void PressureSolver::define(const IBox& domain, const bool periodic[3], const double dx[3],
                            const int bc_type[3][2], const std::vector<IBox>& boxes,
                            const std::vector<int>& owner, double alpha, double beta, const SolverParams& p,
                            const double* dxCrse)
{
    SOMAR_CHECK(lev.empty(), "solver already defined");
    prm = p;
    hasCF_ = dxCrse != nullptr;
    if (hasCF_)
        for (int d = 0; d < 3; ++d) dxCrse_[d] = dxCrse[d];
    SOMAR_CHECK(prm.relaxMode == RELAX_LEVEL_GSRB || prm.relaxMode == RELAX_JACOBI || prm.relaxMode == RELAX_LINE_GSRB ||
                    prm.relaxMode == RELAX_LOOSE_GSRB,
                "relax_mode must be 0 (Jacobi), 1 (LevelGSRB), 2 (LooseGSRB) or 3 (LineGSRB)");
    SOMAR_CHECK(prm.precondMode == PRECOND_DIAG_RELAX || prm.precondMode == PRECOND_NONE ||
                    prm.precondMode == PRECOND_DIAG_LINE_RELAX,
                "bad precondMode");
    for (int d = 0; d < p.spaceDim; ++d)
        for (int s = 0; s < 2; ++s)
            SOMAR_CHECK(periodic[d] || bc_type[d][s] == BC_NEUM || bc_type[d][s] == BC_DIRI,
                        "physical BCs are Neumann (0) or Dirichlet (1)");
    SOMAR_CHECK(prm.spaceDim == 2 || prm.spaceDim == 3, "space_dim must be 2 or 3");
    diri_ = false;
    for (int d = 0; d < p.spaceDim; ++d)
        for (int s = 0; s < 2; ++s)
            if (!periodic[d] && bc_type[d][s] == BC_DIRI) diri_ = true;
    std::unique_ptr<Level> L(new Level);
    if (prm.spaceDim == 2) {
        SOMAR_CHECK(domain.size(2) == 1, "space_dim 2 wants a domain (and boxes) one cell thick in z");
        if (prm.relaxMode == RELAX_LINE_GSRB || prm.precondMode == PRECOND_DIAG_LINE_RELAX)
            for (const IBox& b : boxes)   // 'LineGSRBIter2D: region must have a vertical lower bound of zero' (GSRBF.ChF:1561-1565)
                SOMAR_CHECK(b.lo[1] == domain.lo[1],
                            "2-D line relaxation wants boxes that start at the bottom of the vertical (direction 1)");
        L->active[2] = 0;
    }
    L->alpha = alpha;
    L->beta = beta;
    L->define(domain, periodic, dx, bc_type, boxes, owner, comm_);
    L->alloc_metric();
    if (hasCF_) {
        L->define_cf(dxCrse_);
        // Line relaxation with coarse-fine boundaries at the ENDS of a column: LineGSRB::relax hands the Fortran the codes of
        // BCDescriptor::stencil (BCInterface/BCDescriptor.H:218-229), which is BCType::None for every box end inside the domain --
        // the BCType_CF row of LineGSRBIter3D / 2D (GSRBF.ChF:1804-1817, 1588-1590) is never reached from there -- so such an end
        // gets coeff1 = 0 ("low-order extrapolation") and the coarse-fine ghost is not read.  The kernels do exactly that.
    }
    lev.push_back(std::move(L));
}

void PressureSolver::set_metric_ortho(int patch, const double* jg0, const double* jg1, const double* jg2,
                                      const double* jinv)
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_metric before define / after finalize");
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    const IBox valid = L.boxes[L.local[patch]];
    const double* jg[3] = {jg0, jg1, jg2};
    for (int d = 0; d < prm.spaceDim; ++d) {
        SOMAR_CHECK(jg[d] != nullptr, "null metric array");
        IBox fb = valid;
        fb.hi[d] += 1;
        L.upload(L.dev.jg[d], patch, jg[d], fb, fb, st_);
    }
    L.upload(L.dev.jinv, patch, jinv, valid, valid, st_);
    sync();
}

// Semicoarsening rule + fallback, MappedAMRPoissonOpFactory.cpp:476-550
bool PressureSolver::build_coarser(int depth)
{
    if (prm.maxDepth >= 0 && depth > prm.maxDepth) {
        SOMAR_CHECK(depth > (int)forcedRatios.size(), "You must make the maxDepth large enough to accomodate the mini V-cycles");
        return false;
    }
    Level& F = *lev[depth - 1];
    int prev[3] = {1, 1, 1};
    for (auto& r : mgRefRatios)
        for (int d = 0; d < 3; ++d) prev[d] *= r[d];
    const int nd = prm.spaceDim;
    const int mmc[3] = {S_MAX_COARSE, S_MAX_COARSE, nd == 3 ? S_MAX_COARSE : 1};
    int r[3] = {1, 1, 1};
    const bool forced = depth <= (int)forcedRatios.size();  // the mini V-cycle's coarsening pattern, Factory.cpp:414-441
    if (forced) {
        for (int d = 0; d < 3; ++d) r[d] = forcedRatios[depth - 1][d];
        int tot[3];
        for (int d = 0; d < 3; ++d) tot[d] = prev[d] * r[d] * mmc[d];
        SOMAR_CHECK(coarsenable(lev[0]->boxes, tot),
                    "Could not coarsen grids for mini V-cycle: the block factor is too small for this refinement ratio");
    }
    double maxDx = 0.0;
    for (int d = 0; d < nd; ++d) maxDx = std::max(maxDx, F.dx[d]);
    for (int d = 0; d < nd && !forced; ++d)
        if (F.dx[d] <= maxDx / 2.0) r[d] = 2;
    if (!forced && r[0] * r[1] * r[2] == 1) { r[0] = r[1] = 2; r[2] = nd == 3 ? 2 : 1; }
    int tot[3];
    for (int d = 0; d < 3; ++d) tot[d] = prev[d] * r[d] * mmc[d];
    const std::vector<IBox>& base = lev[0]->boxes;
    if (!forced && !coarsenable(base, tot)) {
        int q[3] = {1, 1, 1};
        for (int d = 0; d < nd; ++d) {
            q[d] = 2;
            int t2[3];
            for (int e = 0; e < 3; ++e) t2[e] = prev[e] * mmc[e] * q[e];
            if (!coarsenable(base, t2)) q[d] = 1;
        }
        if (q[0] * q[1] * q[2] == 1) return false;
        int refDir = 0;
        while (refDir < nd && q[refDir] != 1) ++refDir;
        SOMAR_CHECK(refDir < nd, "semicoarsening fallback: no stuck direction");
        for (int d = 0; d < nd; ++d)
            if (q[d] > 1 && F.dx[d] / F.dx[refDir] > 0.5) q[d] = 1;
        if (q[0] * q[1] * q[2] == 1) return false;
        for (int d = 0; d < 3; ++d) { r[d] = q[d]; tot[d] = prev[d] * r[d] * mmc[d]; }
        if (!coarsenable(base, tot)) return false;
    }
    mgRefRatios.push_back({r[0], r[1], r[2]});
    for (int d = 0; d < 3; ++d) F.mgCrseRefRatio[d] = r[d];
    F.hasCoarser = true;

    std::unique_ptr<Level> C(new Level);
    C->active[2] = F.active[2];
    C->alpha = F.alpha;
    C->beta = F.beta;
    std::vector<IBox> cb;
    for (const IBox& b : F.boxes) cb.push_back(b.coarsen(r));
    double cdx[3];
    for (int d = 0; d < 3; ++d) cdx[d] = F.dx[d] * (double)r[d];
    C->define(F.domain.coarsen(r), F.periodic, cdx, F.bc_type, cb, F.owner, comm_);
    C->alloc_metric();
    if (hasCF_) C->define_cf(dxCrse_);  // CFRegion::coarsen + the AMR coarser level's spacing (Factory.cpp:596-600)
    // coarse metrics: fill_MGfields, MappedAMRPoissonOpFactory.cpp:1164-1234
    for (int pi = 0; pi < C->npatches(); ++pi)
        for (int d = 0; d < nd; ++d)
            launch_avg_face(st_, C->dev, F.dev, pi, C->hpatches[pi].n, C->dev.jg[d], F.dev.jg[d], d, r);
    if (full_) {
        alloc_full_metric(*C);
        for (int pi = 0; pi < C->npatches(); ++pi)
            for (int d = 0; d < nd; ++d)
                for (int c = 0; c < nd; ++c)
                    if (c != d)
                        launch_avg_face(st_, C->dev, F.dev, pi, C->hpatches[pi].n, C->dev.jgf[d][c], F.dev.jgf[d][c], d, r);
    }
    launch_avg_harmonic(st_, C->dev, F.dev, C->dev.jinv, F.dev.jinv, r);
    launch_lapdiag(st_, C->dev);
    fill_metric_ghosts(*C);
    lev.push_back(std::move(C));
    return true;
}

// The fused sweep recomputes its neighbours' red ring, so Jg and Jinv need ghost values wherever a
// neighbouring box or a periodic image exists.  NOTE: a face shared by two boxes (or by periodic
// images) is stored by both; this exchange makes the two copies identical (the low-side owner's
// value wins).  They already are identical when both come from one evaluation of the map.
void PressureSolver::fill_metric_ghosts(Level& L)
{
    // on a level with coarse-fine boundaries the recomputed red ring also needs a neighbour's coefficient on a face
    // that no box owns as a low face (see Copier::define_faces); the ordinary exchange then has the last word
    for (int d = 0; d < 3; ++d)
        if (L.cf_faces[d]) L.cf_faces[d]->run(L.dev.jg[d], L.dev.jg[d], st_);
    for (int d = 0; d < 3; ++d) xchg(L, L.dev.jg[d]);
    if (full_)
        for (int d = 0; d < 3; ++d)
            for (int c = 0; c < 3; ++c)
                if (c != d && L.dev.jgf[d][c]) xchg(L, L.dev.jgf[d][c]);
    xchg(L, L.dev.jinv);
}

// null-space probe, MappedAMRPoissonOpFactory.cpp:659-693
void PressureSolver::probe_null_space(int d)
{
    Level& L = *lev[d];
    double* phi = L.alloc_field();
    double* rhs0 = L.alloc_field();
    double* res = L.alloc_field();
    launch_set(st_, phi, L.field_elems, 0.0);
    apply_op(d, rhs0, phi);
    launch_set(st_, phi, L.field_elems, 1.0);
    residual(d, res, phi, rhs0);
    launch_reduce(st_, L.dev, res, nullptr, 3, d_partials, d_scalars + SLOT_TMP);
    comm_->allreduce(d_scalars + SLOT_TMP, 1, 1, st_);
    const double maxNorm = std::fabs(fetch_scalar(SLOT_TMP));
    L.zeroAvg = maxNorm < 0.01 * (probe_eps > 0.0 ? probe_eps : prm.eps);
    Level::free_field(phi);
    Level::free_field(rhs0);
    Level::free_field(res);
}

// A Cartesian map hands over constant arrays (CartesianMap.cpp:261-280): J g^{aa} and J^{-1} are then the same number on
// every face / in every cell of a depth -- and of every coarser depth, whose averages of equal numbers are exact.  The
// k-marching kernels take such coefficients from StencilParams instead of streaming them (same arithmetic, same bits).
// SOMAR_NO_UNIFORM=1 disables the detection (A/B measurements, parity tests of the streaming path on Cartesian inputs).
void PressureSolver::detect_uniform_metric()
{
    const char* e = getenv("SOMAR_NO_UNIFORM");
    const bool off = e && atoi(e) != 0;
    for (auto& Lp : lev) Lp->dev.P.uniform = 0;
    if (off || full_ || prm.spaceDim != 3) return;
    // one scratch buffer for the deepest patch table, released whatever happens below
    size_t nmax = 1;
    for (auto& Lp : lev) nmax = std::max(nmax, (size_t)std::max(Lp->npatches(), 1));
    const size_t cap = 2 * nmax * MM_CH;
    struct Scratch {
        double* p = nullptr;
        ~Scratch() { hipFree(p); }
    } scratch;
    SOMAR_HIP(hipMalloc(&scratch.p, sizeof(double) * (cap + 8)));
    double* d_mm = scratch.p;
    std::vector<double> mm(cap);
    for (auto& Lp : lev) {
        Level& L = *Lp;
        if (!(L.active[0] && L.active[1] && L.active[2])) continue;
        const int np = L.npatches();
        // (max, -min) of the four arrays; a rank without boxes on this depth contributes -inf
        double h[8];
        for (double& v : h) v = -HUGE_VAL;
        const size_t nmm = 2 * (size_t)std::max(np, 1) * MM_CH;
        for (int a = 0; a < 4; ++a) {
            if (!np) break;
            launch_minmax_valid(st_, L.dev, a < 3 ? L.dev.jg[a] : L.dev.jinv, a < 3 ? a : -1, d_mm);
            SOMAR_HIP(hipMemcpyAsync(mm.data(), d_mm, sizeof(double) * nmm, hipMemcpyDeviceToHost, st_));
            SOMAR_HIP(hipStreamSynchronize(st_));
            for (size_t q = 0; q < (size_t)np * MM_CH; ++q) {
                h[2 * a] = std::max(h[2 * a], mm[2 * q + 1]);
                h[2 * a + 1] = std::max(h[2 * a + 1], -mm[2 * q]);
            }
        }
        if (comm_->size > 1) {
            double* d8 = d_mm + cap;
            SOMAR_HIP(hipMemcpyAsync(d8, h, sizeof(h), hipMemcpyHostToDevice, st_));
            comm_->allreduce(d8, 8, 1, st_);
            SOMAR_HIP(hipMemcpyAsync(h, d8, sizeof(h), hipMemcpyDeviceToHost, st_));
            SOMAR_HIP(hipStreamSynchronize(st_));
        }
        bool uni = true;
        for (int a = 0; a < 4; ++a) uni = uni && std::isfinite(h[2 * a]) && h[2 * a] == -h[2 * a + 1];
        if (!uni) continue;
        L.dev.P.uniform = 1;
        for (int a = 0; a < 4; ++a) L.dev.P.uc[a] = h[2 * a];
    }
}

// Non-diagonal metric: are J g^{xy} (x-faces) and J g^{yx} (y-faces) identically zero on a depth?  True for every map of the
// form x = xi, y = eta, z = z(xi, eta, zeta) -- BathymetricBaseMap (geometry/maps/BathymetricBaseMap.cpp:133-313) and all its
// subclasses: the x-face normal has no eta component.  Coarse depths inherit it (averages of zeros).  SOMAR_NO_ZERO_PLANES=1
// switches the detection off (A/B; the kernels then multiply the stored zeros, same values).
void PressureSolver::detect_zero_planes()
{
    for (auto& Lp : lev) Lp->dev.P.zero_xy = 0;
    const char* e = getenv("SOMAR_NO_ZERO_PLANES");
    if ((e && atoi(e) != 0) || !full_ || prm.spaceDim != 3) return;
    size_t nmax = 1;
    for (auto& Lp : lev) nmax = std::max(nmax, (size_t)std::max(Lp->npatches(), 1));
    const size_t cap = 2 * nmax * MM_CH;
    struct Scratch {
        double* p = nullptr;
        ~Scratch() { hipFree(p); }
    } scratch;
    SOMAR_HIP(hipMalloc(&scratch.p, sizeof(double) * (cap + 1)));
    std::vector<double> mm(cap);
    for (auto& Lp : lev) {
        Level& L = *Lp;
        if (!(L.active[0] && L.active[1] && L.active[2])) continue;
        const int np = L.npatches();
        double nonzero = 0.0;   // max |value| over both planes
        const int pl[2][2] = {{0, 1}, {1, 0}};
        for (int w = 0; w < 2 && np; ++w) {
            const double* a = L.dev.jgf[pl[w][0]][pl[w][1]];
            if (!a) { nonzero = 1.0; break; }
            launch_minmax_valid(st_, L.dev, a, pl[w][0], scratch.p);
            SOMAR_HIP(hipMemcpyAsync(mm.data(), scratch.p, sizeof(double) * 2 * (size_t)np * MM_CH, hipMemcpyDeviceToHost, st_));
            SOMAR_HIP(hipStreamSynchronize(st_));
            for (size_t q = 0; q < (size_t)np * MM_CH; ++q) {
                if (std::isfinite(mm[2 * q])) nonzero = std::max(nonzero, std::fabs(mm[2 * q]));
                if (std::isfinite(mm[2 * q + 1])) nonzero = std::max(nonzero, std::fabs(mm[2 * q + 1]));
            }
        }
        if (comm_->size > 1) {
            double* d1 = scratch.p + cap;
            SOMAR_HIP(hipMemcpyAsync(d1, &nonzero, sizeof(double), hipMemcpyHostToDevice, st_));
            comm_->allreduce(d1, 1, 1, st_);
            SOMAR_HIP(hipMemcpyAsync(&nonzero, d1, sizeof(double), hipMemcpyDeviceToHost, st_));
            SOMAR_HIP(hipStreamSynchronize(st_));
        }
        L.dev.P.zero_xy = nonzero == 0.0 ? 1 : 0;
    }
}

void PressureSolver::finalize()
{
    SOMAR_CHECK(!lev.empty() && !finalized, "finalize before define / twice");
    // SOMAR_TIMING=1: wall time of the stages below on stderr (what a re-definition of the hierarchy costs)
    static const bool timing = getenv("SOMAR_TIMING") && atoi(getenv("SOMAR_TIMING")) != 0;
    auto now = [&]() { if (timing) hipDeviceSynchronize(); return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double>(b - a).count();
    };
    const auto t0 = now();
    launch_lapdiag(st_, lev[0]->dev);
    fill_metric_ghosts(*lev[0]);
    int depth = 1;
    while (build_coarser(depth)) {
        if (comm_->size > 1 && lev[depth]->valid_cells_global <= agglom_cells_ && !full_) {
            build_agglomerated_tail(depth);
            break;
        }
        ++depth;
    }
    const auto t1 = now();
    detect_uniform_metric();
    detect_zero_planes();
    {
        // the fused red+black 19-point sweep on levels of large boxes
        if (const char* e = getenv("SOMAR_FUSED19_MIN_BOX")) fused19_min_box_ = atoi(e);
        fused19_.assign(lev.size(), 0);
        for (size_t d = 0; d < lev.size(); ++d) {
            Level& L = *lev[d];
            bool ok = full_ && fused19_min_box_ >= 0 && full_march((int)d) && full_march_rows() == 8 && !L.boxes.empty();
            for (const IBox& b : L.boxes)
                for (int a = 0; a < 3; ++a) ok = ok && b.size(a) >= std::max(fused19_min_box_, 8) && (a != 0 || b.size(a) % 2 == 0);
            fused19_[d] = ok ? 1 : 0;
            if (ok != L.want_fused19_) {
                L.want_fused19_ = ok;
                L.build_march_tiles(L.narrow7_);
            }
        }
    }
    {
        // narrow lane classes for the 7-point marching kernels where the metric is uniform (Level::build_march_tiles says why);
        // SOMAR_NARROW_7PT = 0 | 1 forces never / always (A/B)
        const char* e = getenv("SOMAR_NARROW_7PT");
        for (auto& Lp : lev) {
            const bool want = e ? atoi(e) != 0 : Lp->dev.P.uniform != 0;
            if (want != Lp->narrow7_) Lp->build_march_tiles(want);
        }
    }
    const auto t2 = now();
    struct Report {
        bool on; std::chrono::steady_clock::time_point t0, t1, t2; long long cells;
        ~Report() {
            if (!on) return;
            hipDeviceSynchronize();
            const auto t3 = std::chrono::steady_clock::now();
            fprintf(stderr, "[somar timing] finalize %lld cells: coarse hierarchy %.3f s, uniform detection %.3f s, fields + probes %.3f s\n",
                    cells, std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(),
                    std::chrono::duration<double>(t3 - t2).count());
        }
    } report{timing, t0, t1, t2, lev[0]->valid_cells_global};
    (void)secs;
    int maxTiles = 1;
    for (auto& L : lev) maxTiles = std::max(std::max(maxTiles, L->dev.ntiles), std::max(L->nrtiles, L->nftiles));
    SOMAR_HIP(hipMalloc(&d_partials, (size_t)maxTiles * 2 * sizeof(double)));
    const int D = (int)lev.size();
    f_res.assign(D, nullptr);
    f_corr.assign(D, nullptr);
    f_scratch.assign(D, nullptr);
    f_pp.assign(D, nullptr);
    if (full_) {
        SOMAR_CHECK(prm.relaxMode == RELAX_LEVEL_GSRB || prm.relaxMode == RELAX_JACOBI || prm.relaxMode == RELAX_LINE_GSRB,
                    "the non-diagonal metric path offers LevelGSRB, LineGSRB and Jacobi");
        // (space_dim 2: LineGSRBIter2D with its two cross terms, line_gsrb.hip)
        f_psi.assign(D, nullptr);
        full_prog_.resize(D);
        for (int d = 0; d < D; ++d) {
            f_psi[d] = lev[d]->alloc_field();
            build_full_programs(d);
        }
    }
    d_diri_ops_.assign(D, nullptr);
    n_diri_ops_.assign(D, 0);
    if (diri_)
        for (int d = 0; d < D; ++d) build_diri_ops(d);
    for (int d = 0; d < D; ++d) {
        if (d > 0) { f_res[d] = lev[d]->alloc_field(); f_corr[d] = lev[d]->alloc_field(); }
        f_scratch[d] = lev[d]->alloc_field();
        f_pp[d] = lev[d]->alloc_field();
    }
    Level& L0 = *lev[0];
    f_phi = L0.alloc_field();
    f_rhs = L0.alloc_field();
    f_uberRes = L0.alloc_field();
    f_uberCorr = L0.alloc_field();
    f_best = L0.alloc_field();
    // folded prolongation: child volumes and total volume per depth
    f_W.assign(D, nullptr);
    sf_valid_.assign(D, 0);
    SOMAR_HIP(hipMalloc(&d_fold, (size_t)D * 8 * sizeof(double)));
    SOMAR_HIP(hipMemset(d_fold, 0, (size_t)D * 8 * sizeof(double)));
    SOMAR_HIP(hipDeviceSynchronize());
    for (int d = 0; d + 1 < D; ++d) {
        Level& F = *lev[d];
        Level& C = *lev[d + 1];
        f_W[d + 1] = C.alloc_field();
        launch_child_volume(st_, C.dev, F.dev, f_W[d + 1], F.mgCrseRefRatio, F.dxProduct);
        launch_reduce(st_, C.dev, f_W[d + 1], nullptr, 2, d_partials, d_fold + 8 * d + 2);  // V = sum of volumes (> 0)
        comm_->allreduce(d_fold + 8 * d + 2, 1, 0, st_);
    }
    for (int i = 0; i < 8; ++i) bicg[i] = lev[D - 1]->alloc_field();
    for (int d = 0; d < D; ++d) probe_null_space(d);
    // first depth whose V-cycle leg is launch-bound (see solver.h); one rank only, never inside a sharded region
    graph_from_ = -1;
    if (comm_->size == 1 && !coarse_ && graph_cells_ > 0)
        for (int d = 0; d <= D - 2; ++d)
            if (lev[d]->valid_cells_global <= graph_cells_) { graph_from_ = d; break; }
    sync();
    finalized = true;
}

// The hierarchy from `depth` on, replicated on every rank (see solver.h).  lev[depth] stays as the sharded
// landing layout of the restriction; its data are allgathered into depth 0 of a communicator-less solver that
// owns every box and builds the remaining depths itself with the same rules.
void PressureSolver::build_agglomerated_tail(int depth)
{
    Level& T = *lev[depth];
    SolverParams cp = prm;
    if (cp.maxDepth >= 0) cp.maxDepth = std::max(0, cp.maxDepth - depth);
    coarse_.reset(new PressureSolver(nullptr, st_));
    coarse_->probe_eps = probe_eps > 0.0 ? probe_eps : prm.eps;
    coarse_->graph_cells_ = 0;  // graph replay is exercised (and measured) on single-process runs only
    for (int a = 0; a < 3; ++a)
        for (int q = 0; q < 2; ++q) coarse_->bc_value_[a][q] = bc_value_[a][q];
    std::vector<int> own(T.boxes.size(), 0);
    coarse_->define(T.domain, T.periodic, T.dx, T.bc_type, T.boxes, own, T.alpha, T.beta, cp,
                    hasCF_ ? dxCrse_ : nullptr);
    Level& R = *coarse_->lev[0];
    // metric: faces of a box live at indices 0..n, so take one layer beyond the valid cells
    Copier metric;
    metric.define_allgather(T, R, 1, comm_);
    for (int d = 0; d < prm.spaceDim; ++d) metric.run(T.dev.jg[d], R.dev.jg[d], st_);
    metric.run(T.dev.jinv, R.dev.jinv, st_);
    sync();
    coarse_->finalize();
    agglom_depth_ = depth;
    agglom_gather_.define_allgather(T, R, 0, comm_);
    std::vector<CopyItem> back;
    for (int pi = 0; pi < T.npatches(); ++pi) {
        CopyItem it;
        std::memset(&it, 0, sizeof(it));
        it.src_patch = T.local[pi];  // replicated layout: patch index = global box index
        it.dst_patch = pi;
        for (int d = 0; d < 3; ++d) it.n[d] = T.hpatches[pi].n[d];
        back.push_back(it);
    }
    n_agglom_back_ = (int)back.size();
    if (n_agglom_back_) {
        SOMAR_HIP(hipMalloc(&d_agglom_back_, back.size() * sizeof(CopyItem)));
        SOMAR_HIP(hipMemcpy(d_agglom_back_, back.data(), back.size() * sizeof(CopyItem), hipMemcpyHostToDevice));
        SOMAR_HIP(hipDeviceSynchronize());
    }
}

void PressureSolver::agglom_cycle(double* corr, const double* res, bool corr_zero)
{
    Level& T = *lev[agglom_depth_];
    PressureSolver& C = *coarse_;
    double* rC = C.work(0);
    double* cC = C.work(1);
    if (profiling_) prof_begin(3);
    agglom_gather_.run(res, rC, st_);
    if (!corr_zero) agglom_gather_.run(corr, cC, st_);
    C.bottom_metric = bottom_metric;
    C.bottom_eps_eff = bottom_eps_eff;
    C.prm.num_smooth_down = prm.num_smooth_down;
    C.prm.num_smooth_up = prm.num_smooth_up;
    C.prm.num_smooth_bottom = prm.num_smooth_bottom;
    C.prm.numMG = prm.numMG;
    C.cycle_override_ = cycle_override_;   // inside an F-cycle's inner V-cycle the replicated tail runs a V-cycle too
    C.cycle(0, cC, rC, corr_zero);
    C.cycle_override_ = 0;
    bottom_iters = C.bottom_iters;
    bottom_exit = C.bottom_exit;
    launch_copy_items2(st_, C.lev[0]->dev.patches, T.dev.patches, d_agglom_back_, n_agglom_back_, cC, corr);
    if (profiling_) prof_end(3);
}

// ------------------------------------------------------------------------------------
// boundary transfers
// ------------------------------------------------------------------------------------
void PressureSolver::upload_phi(int patch, const double* host, const int ghost[3])
{
    Level& L = *lev[0];
    const IBox valid = L.boxes[L.local[patch]];
    L.upload(f_phi, patch, host, valid.grow(ghost), valid, st_);
}
void PressureSolver::upload_rhs(int patch, const double* host, const int ghost[3])
{
    Level& L = *lev[0];
    const IBox valid = L.boxes[L.local[patch]];
    L.upload(f_rhs, patch, host, valid.grow(ghost), valid, st_);
}
void PressureSolver::download_phi(int patch, double* host, const int ghost[3])
{
    download_field(f_phi, 0, patch, host, ghost);
}
void PressureSolver::download_field(const double* field, int depth, int patch, double* host, const int ghost[3])
{
    Level& L = *lev[depth];
    const IBox valid = L.boxes[L.local[patch]];
    int g[3];
    for (int d = 0; d < 3; ++d) {
        SOMAR_CHECK(ghost[d] >= 0 && ghost[d] <= FRAME, "host ghost wider than the device frame");
        g[d] = ghost[d];
    }
    const IBox hb = valid.grow(g);
    L.download(field, patch, host, hb, hb, st_);
    sync();
}

// ------------------------------------------------------------------------------------
// level operator
// ------------------------------------------------------------------------------------
bool PressureSolver::fused_relax(int d, int iters) const
{
    const Level& L = *lev[d];
    // levels with coarse-fine boundaries qualify when their layout allows it (Level::cf_fusable)
    return prm.relaxMode == RELAX_LEVEL_GSRB && L.valid_cells_global >= fused_min_cells_ && iters > 0 &&
           (L.ncf == 0 || L.cf_fusable) && L.active[2] && !no_cf_fused_(L) && !full_;  // Dirichlet sides: ghosts synthesized in the kernel
}

void PressureSolver::relax(int d, double* e, const double* res, int iters, bool e_zero, const double* e_shift,
                           const Level* e_plus_level, const double* e_plus)
{
    Level& L = *lev[d];
    const bool fused_path = fused_relax(d, iters);
    SOMAR_CHECK((!e_shift && !e_plus) || fused_path, "deferred mean removal / folded prolongation need the fused sweep");
    static const bool no_zero_start = getenv("SOMAR_NO_ZERO_START") != nullptr;  // A/B switch
    if (e_zero && (!fused_path || no_zero_start)) {
        launch_set(st_, e, L.field_elems, 0.0);
        e_zero = false;
    }
    if (fused_path) {
        // LevelGSRB::relax (GSRB.cpp:58-98) as ONE fused red+black launch per sweep (gsrb_fused.hip):
        // same values bit for bit, one ghost exchange per sweep instead of two, ping-pong buffers.
        xchg(L, const_cast<double*>(res));  // rhs ghosts: constant over the sweeps
        double* cur = e;
        double* alt = f_pp[d];
        for (int it = 0; it < iters; ++it) {
            const bool zin = e_zero && it == 0;  // zeros need neither an exchange nor a read
            int mode = 0;
            if (zin) mode = 1;
            else if (it == 0 && e_plus) mode = e_shift ? 4 : 3;
            else if (it == 0 && e_shift) mode = 2;
            if (!zin && fused_overlap(L)) {
                // the sweep's one exchange in flight on the second stream while the tiles that read none of its cells are swept
                // (the CF ghosts are other cells, computed from this rank's valid cells: filled before the first tiles run)
                L.cf_homog_ext(cur, st_);
                overlapped(L, cur, L.d_ftiles_own, L.nftiles_own, L.d_ftiles_rem, L.nftiles_rem, [&](Tile* tl, int nt) {
                    launch_gsrb_fused(st_, tl, nt, L.dev, alt, cur, res, mode, e_shift, e_plus_level ? &e_plus_level->dev : nullptr,
                                      e_plus, L.mgCrseRefRatio);
                });
                std::swap(cur, alt);
                continue;
            }
            if (!zin) {
                xchg(L, cur);
                L.cf_homog_ext(cur, st_);  // CF ghosts (faces + the edge ghosts the red ring reads), pre-sweep values
            }
            if (profiling_ && d == 0) prof_begin(0);
            launch_gsrb_fused(st_, L.d_ftiles, L.nftiles, L.dev, alt, cur, res, mode, e_shift,
                              e_plus_level ? &e_plus_level->dev : nullptr, e_plus, L.mgCrseRefRatio);
            if (profiling_ && d == 0) prof_end(0);
            std::swap(cur, alt);
        }
        if (cur != e) launch_copy(st_, e, cur, L.field_elems);
        return;
    }
    if (prm.relaxMode == RELAX_LEVEL_GSRB && full_march(d) && fused19(d)) {
        // LevelGSRB::relax with a non-diagonal metric on a level of large boxes: red everywhere and black three layers inside
        // every box in ONE marching launch (full19_fused.hip), the between-colour ghost work on its output exactly as in the
        // two-pass form below, then the black cells of the outer layers in place (k_full_march<3>).  Same values, same bits.
        double* cur = e;
        double* alt = f_pp[d];
        for (int it = 0; it < iters; ++it) {
            L.cf_homog(cur, st_);
            xchg(L, cur);
            run_full_program_frames(d, 1, cur);
            copy_frames(d, cur, alt);
            if (profiling_ && d == 0) prof_begin(0);
            launch_full_fused(st_, L.d_gtiles, L.ngtiles, L.dev, alt, cur, f_psi[d], res);
            if (profiling_ && d == 0) prof_end(0);
            L.cf_homog(alt, st_);
            xchg(L, alt);
            run_full_program_frames(d, 1, alt);
            if (profiling_ && d == 0) prof_begin(0);   // (slot 0 then holds two launches per sweep, as in the two-pass form)
            launch_gsrb_full_shell(st_, L.d_stiles, L.nstiles, L.dev, alt, cur, f_psi[d], res);
            if (profiling_ && d == 0) prof_end(0);
            ++counters[4];
            std::swap(cur, alt);
        }
        if (cur != e) launch_copy(st_, e, cur, L.field_elems);
        return;
    }
    if (prm.relaxMode == RELAX_LEVEL_GSRB && full_march(d)) {
        // LevelGSRB::relax with a non-diagonal metric on a large level: per colour the same exchange / CF fill / ghost
        // program, then ONE marching pass (full19_march.hip) that reads the pre-pass values everywhere (the snapshot
        // semantics of the reference's `extrap`) and writes the other buffer.
        double* cur = e;
        double* alt = f_pp[d];
        for (int it = 0; it < iters; ++it)
            for (int pass = 0; pass < 2; ++pass) {
                L.cf_homog(cur, st_);
                xchg(L, cur);
                run_full_program_frames(d, 1, cur);
                // the pass rewrites the valid cells only; the ghost frame travels with them (edge / vertex ghosts at
                // coarse-fine corners keep whatever the last ExtrapolateCFEV left there, as in the reference's in-place sweep)
                copy_frames(d, cur, alt);
                if (profiling_ && d == 0) prof_begin(0);
                launch_gsrb_full_march(st_, L.d_qtiles, L.nqtiles, L.dev, alt, cur, f_psi[d], res, pass);
                if (profiling_ && d == 0) prof_end(0);
                std::swap(cur, alt);
            }
        // an even number of passes: the result is back in e
        return;
    }
    for (int it = 0; it < iters; ++it) {
        if (prm.relaxMode == RELAX_LEVEL_GSRB) {
            // LevelGSRB::relax, GSRB.cpp:58-98.  The Neumann ghost fill of
            // fillGhostsAndExtrapolate is dead code for a diagonal metric (the boundary
            // stencil never reads a Neumann ghost), so only the exchange remains.
            for (int pass = 0; pass < 2; ++pass) {
                // pull exchange: the colour pass refreshes the ghosts it reads itself (coarse-fine and Dirichlet ghosts are
                // other cells, computed from valid cells only: their order against the exchange does not matter)
                const bool pull = !full_ && L.pull_ready();
                L.cf_homog(e, st_);  // homogeneousCFInterp (Relaxer::fillGhostsAndExtrapolate)
                if (!pull) xchg(L, e);
                if (diri_ && !full_) apply_diri(d, e, true);  // ... and its physical ghosts (doBCs); non-diagonal: in the program
                if (full_) run_full_program(d, 1, e);  // psi snapshot + extrapolation (order 1) + Neumann ghosts
                if (profiling_ && d == 0) prof_begin(0);
                if (full_) launch_gsrb_full(st_, L.dev, e, f_psi[d], res, pass);
                else launch_gsrb_ortho(st_, L.dev, e, res, pass, 0, pull);
                if (profiling_ && d == 0) prof_end(0);
            }
        } else if (prm.relaxMode == RELAX_LOOSE_GSRB) {
            // LooseGSRB::relax, GSRB.cpp:104-141: ONE exchange per sweep; red+black on the cells strictly inside
            // each box, then red+black on the box shells.  (The exchange "begun" before the interior phase only
            // carries shell cells, which that phase does not touch: completing it first changes nothing.)
            L.cf_homog(e, st_);
            xchg(L, e);
            if (diri_) apply_diri(d, e, true);
            launch_gsrb_ortho(st_, L.dev, e, res, 0, 1);
            launch_gsrb_ortho(st_, L.dev, e, res, 1, 1);
            launch_gsrb_ortho(st_, L.dev, e, res, 0, 2);
            launch_gsrb_ortho(st_, L.dev, e, res, 1, 2);
        } else if (prm.relaxMode == RELAX_LINE_GSRB) {
            line_relax(d, e, res);
        } else {
            // Jacobi::relax, Jacobi.cpp:54-90
            residual(d, f_scratch[d], e, res);
            launch_diag(st_, L.dev, e, f_scratch[d], 1);
        }
    }
}

// LineGSRB::relax, GSRB.cpp:148-330: per colour exchange, then every (i,j) column of that colour is solved
// exactly in z (line_gsrb.hip); f_pp[d] receives dgtsv's modified diagonal.
void PressureSolver::line_relax(int d, double* e, const double* res)
{
    Level& L = *lev[d];
    for (int pass = 0; pass < 2; ++pass) {
        xchg(L, e);
        L.cf_homog(e, st_);  // fillGhostsAndExtrapolate: homogeneous CF values in the lateral ghost cells
        if (diri_ && !full_) apply_diri(d, e, true);  // ... and the ghosts of lateral Dirichlet sides (the vertical ends are folded in)
        if (full_) run_full_program(d, 1, e);  // extrap copy (order 1) + physical ghosts: the cross terms are explicit
        if (prm.spaceDim == 2) {
            int maxN0 = 1;
            for (const PatchDesc& q : L.hpatches) maxN0 = std::max(maxN0, q.n[0]);
            launch_line_gsrb_2d(st_, L.dev, maxN0, e, res, f_pp[d], pass, full_ ? f_psi[d] : nullptr);
            continue;
        }
        launch_line_gsrb_ortho(st_, L.d_ctiles, L.nctiles, L.ctile_j, L.dev, e, res, f_pp[d], pass, full_ ? f_psi[d] : nullptr);
    }
}

void PressureSolver::residual(int d, double* out, double* phi, const double* rhs, bool homogeneous)
{
    lev[d]->cf_homog(phi, st_);  // interpCFGhosts(homogeneous), MappedAMRPoissonOp.cpp:628-640
    cf_ev(d, phi);
    residual_i(d, out, phi, rhs, homogeneous);
}

void PressureSolver::apply_op(int d, double* out, double* phi, bool homogeneous)
{
    lev[d]->cf_homog(phi, st_);
    cf_ev(d, phi);
    apply_op_i(d, out, phi, homogeneous);
}

void PressureSolver::residual_i(int d, double* out, double* phi, const double* rhs, bool homogeneous)
{
    Level& L = *lev[d];
    // small levels (direct-load operator): the kernel pulls the ghosts it reads, no copy launch
    const bool pull = !full_ && L.pull_ready() && !(L.valid_cells_global >= march_min_cells_ && L.active[2]);
    if (!pull) xchg(L, phi);  // exchangeComplete, MappedAMRPoissonOp.cpp:2222-2238
    if (diri_ && !full_) apply_diri(d, phi, homogeneous);  // m_bc.setGhosts, :822 (non-diagonal: inside the program)
    if (profiling_ && d == 0) prof_begin(1);
    if (full_march(d)) {
        run_full_program_frames(d, 0, phi, homogeneous);
        launch_full_march(st_, L.d_qtiles, L.nqtiles, L.dev, out, phi, f_psi[d], rhs, 0);
    } else if (full_) {
        // exchangeComplete, fillExtrap (order 2), physical ghosts (Neumann with cross terms / Dirichlet), then the 19-point fluxes
        run_full_program(d, 0, phi, homogeneous);
        launch_op_full(st_, L.dev, out, phi, f_psi[d], rhs, 0);
    } else if (L.valid_cells_global >= march_min_cells_ && L.active[2]) launch_resid_march(st_, L.d_rtiles, L.nrtiles, L.dev, out, phi, rhs, 0);  // Dirichlet sides: their ghosts were just written, the kernel only zeroes NEUMANN fluxes
    else launch_op_ortho(st_, L.dev, out, phi, rhs, 0, pull);
    if (profiling_ && d == 0) prof_end(1);
}

void PressureSolver::apply_op_i(int d, double* out, double* phi, bool homogeneous)
{
    Level& L = *lev[d];
    const bool pull = !full_ && L.pull_ready() && !(L.valid_cells_global >= march_min_cells_ && L.active[2]);
    if (!pull) xchg(L, phi);
    if (diri_ && !full_) apply_diri(d, phi, homogeneous);
    if (full_march(d)) {
        run_full_program_frames(d, 0, phi, homogeneous);
        launch_full_march(st_, L.d_qtiles, L.nqtiles, L.dev, out, phi, f_psi[d], nullptr, 1);
    } else if (full_) {
        run_full_program(d, 0, phi, homogeneous);
        launch_op_full(st_, L.dev, out, phi, f_psi[d], nullptr, 1);
    } else if (L.valid_cells_global >= march_min_cells_ && L.active[2]) launch_resid_march(st_, L.d_rtiles, L.nrtiles, L.dev, out, phi, nullptr, 1);
    else launch_op_ortho(st_, L.dev, out, phi, nullptr, 1, pull);
}

void PressureSolver::prolong_from(const LevelDev& C, const double* crse, const int r[3], double* fine)
{
    Level& F = *lev[0];
    const bool os = F.zeroAvg && ord_sharded(0);
    launch_prolong(st_, F.dev, C, fine, crse, r, F.zeroAvg && !os, F.dxProduct, d_partials, d_scalars + SLOT_SUMS,
                   F.field_elems, ordered(0));
    if (F.zeroAvg) {
        if (os) ordered_sums(0, fine, F.dev.jinv, 6, F.dxProduct, d_scalars + SLOT_SUMS);
        else comm_->allreduce(d_scalars + SLOT_SUMS, 2, 0, st_);
        launch_sub_mean(st_, fine, F.field_elems, d_scalars + SLOT_SUMS);
    }
}

double* PressureSolver::amr_field(int which)
{
    SOMAR_CHECK(which == 0 || which == 1, "bad AMR work field");
    if (!f_amr[which]) f_amr[which] = lev[0]->alloc_field();
    return f_amr[which];
}

bool PressureSolver::residual_restrict_i(const LevelDev& C, double* crse, double* phi, const double* rhs, const int r[3])
{
    Level& F = *lev[0];
    for (int d = 0; d < 3; ++d)
        if (r[d] != 1 && r[d] != 2) return false;
    if (!(F.valid_cells_global >= march_min_cells_ && F.active[2] && !full_)) return false;
    xchg(F, phi);  // exchangeComplete, as residual_i
    if (diri_) apply_diri(0, phi, true);
    launch_resid_restrict(st_, F.d_rtiles, F.nrtiles, F.dev, C, crse, phi, rhs, r, F.dxProduct, nullptr);
    return true;
}

void PressureSolver::restrict_residual(int d, double* resCoarse, double* phiFine, const double* rhsFine)
{
    // restrictResidual, MappedAMRPoissonOp.cpp:1281-1304
    Level& F = *lev[d];
    if (F.valid_cells_global >= march_min_cells_ && F.active[2] && !full_) {
        // large level: residual and J-weighted average in one marching pass, the fine residual is never stored
        F.cf_homog(phiFine, st_);
        const bool want = F.zeroAvg && !ordered(d);  // the fine half of the folded prolongation's mean
        if (resid_overlap(F)) {
            overlapped(F, phiFine, F.d_rtiles_own, F.nrtiles_own, F.d_rtiles_rem, F.nrtiles_rem, [&](Tile* tl, int nt) {
                launch_resid_restrict(st_, tl, nt, F.dev, lev[d + 1]->dev, resCoarse, phiFine, rhsFine, F.mgCrseRefRatio,
                                      F.dxProduct, want ? d_partials : nullptr);
            });
            if (want) launch_sum_partials(st_, d_partials, F.nrtiles, d_fold + 8 * d);
            sf_valid_[d] = want ? 1 : 0;
            return;
        }
        xchg(F, phiFine);
        if (diri_) apply_diri(d, phiFine, true);  // homogeneous Dirichlet ghosts, as residual() fills them
        if (profiling_ && d == 0) prof_begin(1);
        launch_resid_restrict(st_, F.d_rtiles, F.nrtiles, F.dev, lev[d + 1]->dev, resCoarse, phiFine, rhsFine,
                              F.mgCrseRefRatio, F.dxProduct, want ? d_partials : nullptr);
        if (want) launch_sum_partials(st_, d_partials, F.nrtiles, d_fold + 8 * d);
        sf_valid_[d] = want ? 1 : 0;
        if (profiling_ && d == 0) prof_end(1);
        return;
    }
    residual(d, f_scratch[d], phiFine, rhsFine);
    launch_restrict(st_, lev[d + 1]->dev, lev[d]->dev, resCoarse, f_scratch[d], lev[d]->mgCrseRefRatio);
}

const double* PressureSolver::prolong_increment(int d, double* phiFine, const double* corrCoarse, bool defer_mean)
{
    // ConstInterpPS / ZeroAvgConstInterpPS, ProlongationStrategy.cpp:49-164.  The two scalar
    // MPI_Allreduce calls of the reference become one 2-element device-side reduction.
    Level& F = *lev[d];
    const bool os = F.zeroAvg && ord_sharded(d);
    launch_prolong(st_, F.dev, lev[d + 1]->dev, phiFine, corrCoarse, F.mgCrseRefRatio, F.zeroAvg && !os, F.dxProduct,
                   d_partials, d_scalars + SLOT_SUMS, F.field_elems, ordered(d));
    if (F.zeroAvg) {
        if (os) ordered_sums(d, phiFine, F.dev.jinv, 6, F.dxProduct, d_scalars + SLOT_SUMS);
        else comm_->allreduce(d_scalars + SLOT_SUMS, 2, 0, st_);
        if (defer_mean) return d_scalars + SLOT_SUMS;
        launch_sub_mean(st_, phiFine, F.field_elems, d_scalars + SLOT_SUMS);
    }
    return nullptr;
}

void PressureSolver::pre_cond(int d, double* phi, const double* rhs)
{
    // preCond, MappedAMRPoissonOp.cpp:684-734
    Level& L = *lev[d];
    if (prm.num_smooth_precond == 0 || prm.precondMode == PRECOND_NONE) {
        launch_copy(st_, phi, rhs, L.field_elems);
        return;
    }
    launch_diag(st_, L.dev, phi, rhs, 0);
    if (prm.precondMode == PRECOND_DIAG_LINE_RELAX && prm.relaxMode != RELAX_LINE_GSRB) {
        for (int it = 0; it < prm.num_smooth_precond; ++it) line_relax(d, phi, rhs);  // m_precondRelaxPtr = LineGSRB
        return;
    }
    relax(d, phi, rhs, prm.num_smooth_precond);
}

double PressureSolver::norm(int d, const double* a, int ord)
{
    Level& L = *lev[d];
    if (ord == 0) {
        if (fused_publish(d)) {
            const ScalarPublish P{h_scalars + SLOT_TMP, h_seq_, ++fetch_seq_};
            launch_reduce(st_, L.dev, a, nullptr, 1, d_partials, d_scalars + SLOT_TMP, false, &P);
            wait_published(P.seq);
            return h_scalars[SLOT_TMP];
        }
        launch_reduce(st_, L.dev, a, nullptr, 1, d_partials, d_scalars + SLOT_TMP);
        comm_->allreduce(d_scalars + SLOT_TMP, 1, 1, st_);
        return fetch_scalar(SLOT_TMP);
    }
    if (ord == 1) {
        if (fused_publish(d)) return reduce_fetch(d, a, nullptr, 2);
        reduce_sum(d, a, nullptr, 2, d_scalars + SLOT_TMP);
        return fetch_scalar(SLOT_TMP);
    }
    SOMAR_CHECK(ord == 2, "norm order must be 0, 1 or 2");
    if (fused_publish(d)) return std::sqrt(reduce_fetch(d, a, a, 0));
    reduce_sum(d, a, a, 0, d_scalars + SLOT_TMP);
    return std::sqrt(fetch_scalar(SLOT_TMP));
}

// single-rank, polling fetch: the reduction's last kernel publishes its result itself (one launch less per scalar)
bool PressureSolver::fused_publish(int d) const
{
    static const bool poll = !(getenv("SOMAR_POLL_FETCH") && atoi(getenv("SOMAR_POLL_FETCH")) == 0);
    static const bool off = getenv("SOMAR_NO_FUSED_PUBLISH") != nullptr;
    return poll && !off && !capturing_ && comm_->size == 1 && !ord_sharded(d);
}

double PressureSolver::reduce_fetch(int d, const double* a, const double* b, int mode)
{
    const ScalarPublish P{h_scalars + SLOT_TMP, h_seq_, ++fetch_seq_};
    launch_reduce(st_, lev[d]->dev, a, b, mode, d_partials, d_scalars + SLOT_TMP, ordered(d), &P);
    wait_published(P.seq);
    return h_scalars[SLOT_TMP];
}

double PressureSolver::dot(int d, const double* a, const double* b)
{
    if (fused_publish(d)) return reduce_fetch(d, a, b, 0);
    reduce_sum(d, a, b, 0, d_scalars + SLOT_TMP);
    return fetch_scalar(SLOT_TMP);
}

void PressureSolver::reduce_sum(int d, const double* a, const double* b, int mode, double* out)
{
    if (ord_sharded(d)) {
        ordered_sums(d, a, b, mode, 0.0, out);
        return;
    }
    launch_reduce(st_, lev[d]->dev, a, b, mode, d_partials, out, ordered(d));
    comm_->allreduce(out, 1, 0, st_);
}

void PressureSolver::ordered_sums(int d, const double* a, const double* b, int mode, double dxProduct, double* out)
{
    Level& L = *lev[d];
    const int nb = (int)L.boxes.size();
    if ((int)d_ord_start_.size() <= d) { d_ord_start_.resize(lev.size(), nullptr); d_box_start_.resize(lev.size(), nullptr); }
    if (!d_box_start_[d]) {
        std::vector<long long> bs(nb + 1, 0), ls(std::max<size_t>(L.local.size(), 1), 0);
        for (int i = 0; i < nb; ++i) bs[i + 1] = bs[i] + L.boxes[i].numPts();
        for (size_t q = 0; q < L.local.size(); ++q) ls[q] = bs[L.local[q]];
        SOMAR_HIP(hipMalloc(&d_box_start_[d], bs.size() * sizeof(long long)));
        SOMAR_HIP(hipMalloc(&d_ord_start_[d], ls.size() * sizeof(long long)));
        SOMAR_HIP(hipMemcpy(d_box_start_[d], bs.data(), bs.size() * sizeof(long long), hipMemcpyHostToDevice));
        SOMAR_HIP(hipMemcpy(d_ord_start_[d], ls.data(), ls.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    const long long n = L.valid_cells_global;
    if (!d_ordbuf_) SOMAR_HIP(hipMalloc(&d_ordbuf_, (size_t)2 * ordered_max_cells_ * sizeof(double)));
    const int nvec = mode == 6 ? 2 : 1;
    double *X = d_ordbuf_, *Y = d_ordbuf_ + n;
    launch_set(st_, X, nvec * n, 0.0);
    launch_ord_fill(st_, L.dev, d_ord_start_[d], a, b, mode, dxProduct, X, Y);
    comm_->allreduce(X, (int)(nvec * n), 0, st_);
    launch_reduce_ordered_flat(st_, nb, d_box_start_[d], X, Y, mode, out);
}

// f -= sum(f*J)/sum(J): the J-weighted mean removal the callers apply to make an all-Neumann/periodic
// right-hand side solvable (computeMappedSum + setZeroAvg, MappedChombo/computeMappedSum.cpp)
void PressureSolver::remove_mean(int d, double* f)
{
    Level& L = *lev[d];
    launch_reduce(st_, L.dev, f, L.dev.jinv, 4, d_partials, d_scalars + SLOT_SUMS);
    launch_reduce(st_, L.dev, f, L.dev.jinv, 5, d_partials, d_scalars + SLOT_SUMS + 1);
    comm_->allreduce(d_scalars + SLOT_SUMS, 2, 0, st_);
    launch_sub_mean(st_, f, L.field_elems, d_scalars + SLOT_SUMS);
}

// ------------------------------------------------------------------------------------
// MAC level projection: BaseProjector<FluxBox>::project (projection/BaseProjectorI.H:176-299) with
// LevelMACProjector::computeDiv/computeGrad/applyCorrection (LevelMACProjector.cpp:156-241), velocity in
// flux form (a_velIsFlux = true).  Boundary-face values of the velocity are taken as given.
// ------------------------------------------------------------------------------------
double* PressureSolver::vel(int dir)
{
    SOMAR_CHECK(dir >= 0 && dir < 3 && finalized, "bad velocity direction / solver not finalized");
    if (!f_vel[dir]) f_vel[dir] = lev[0]->alloc_field();
    return f_vel[dir];
}

void PressureSolver::upload_vel(int dir, int patch, const double* host)
{
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    IBox fb = L.boxes[L.local[patch]];
    fb.hi[dir] += 1;
    L.upload(vel(dir), patch, host, fb, fb, st_);
    sync();
}

void PressureSolver::download_vel(int dir, int patch, double* host)
{
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    IBox fb = L.boxes[L.local[patch]];
    fb.hi[dir] += 1;
    L.download(vel(dir), patch, host, fb, fb, st_);
    sync();
}

void PressureSolver::divergence_mac(double* out, double dt)
{
    launch_div_mac(st_, lev[0]->dev, out, vel(0), vel(1), vel(2), dt);
}

void PressureSolver::mac_correct(double* phi, double dt)
{
    Level& L = *lev[0];
    xchg(L, phi);  // Copier excp(grids, grids, domain, ghost, true); a_phi.exchange(excp)  (Gradient.cpp:118-121)
    double* v[3] = {vel(0), vel(1), prm.spaceDim == 3 ? vel(2) : nullptr};
    if (full_) {
        // singleBoxMacGrad with a non-diagonal metric (Gradient.cpp:946-1101): extrap from the exchanged phi first, then
        // the extrapolation BC on phi's own physical ghosts, then MAPPEDMACGRAD == MAPPEDGETFLUX(beta = 1) on every face
        mac_grad_full(phi);
        launch_face_axpy(st_, L.dev, v, f_flux, dt == 0.0 ? -1.0 : -dt);
        return;
    }
    launch_mac_correct(st_, L.dev, v, phi, dt == 0.0 ? -1.0 : -dt);
}

void PressureSolver::set_scale_cc(int which, int patch, const double* host, const int ghost[3])
{
    Level& L = *lev[0];
    SOMAR_CHECK(finalized && (which == 0 || which == 1) && patch >= 0 && patch < L.npatches(), "set_scale_cc: bad argument");
    if (!f_sc_cc[which]) f_sc_cc[which] = L.alloc_field();
    const IBox valid = L.boxes[L.local[patch]];
    int one[3] = {0, 0, 0};
    for (int d = 0; d < prm.spaceDim; ++d) {
        SOMAR_CHECK(ghost[d] >= 1, "the cell-centred scale needs the velocity's ghost layer");
        one[d] = 1;
    }
    L.upload(f_sc_cc[which], patch, host, valid.grow(ghost), valid.grow(one), st_);
    sync();
}

void PressureSolver::set_scale_face(int which, int dir, int patch, const double* host)
{
    Level& L = *lev[0];
    SOMAR_CHECK(finalized && (which == 0 || which == 1) && dir >= 0 && dir < prm.spaceDim && patch >= 0 && patch < L.npatches(),
                "set_scale_face: bad argument");
    if (!f_sc_face[which][dir]) f_sc_face[which][dir] = L.alloc_field();
    IBox fb = L.boxes[L.local[patch]];
    fb.hi[dir] += 1;
    L.upload(f_sc_face[which][dir], patch, host, fb, fb, st_);
    sync();
}

void PressureSolver::scale_vel(int centring, int which)
{
    Level& L = *lev[0];
    SOMAR_CHECK(which == 0 || which == 1, "which: 0 = J, 1 = Jinv");
    if (centring == 1) {
        SOMAR_CHECK(f_sc_cc[which], "the cell-centred J / Jinv has not been set (somar_solver_set_cc_j)");
        for (int c = 0; c < prm.spaceDim; ++c) launch_mul(st_, cc_vel(c), f_sc_cc[which], L.field_elems);
    } else {
        for (int d = 0; d < prm.spaceDim; ++d) {
            SOMAR_CHECK(f_sc_face[which][d], "the face-centred J / Jinv has not been set (somar_solver_set_face_j)");
            launch_mul(st_, vel(d), f_sc_face[which][d], L.field_elems);
        }
    }
}

double* const* PressureSolver::mac_grad(double* phi)
{
    Level& L = *lev[0];
    if (full_) {
        mac_grad_full(phi);
        return f_flux;
    }
    for (int a = 0; a < 3; ++a)
        if (!f_flux[a]) f_flux[a] = L.alloc_field();
    // G = 0 + 1.0 * (dxinv * Jg * dphi): k_mac_correct's own expression, stored instead of subtracted (exact)
    double* g[3] = {f_flux[0], f_flux[1], prm.spaceDim == 3 ? f_flux[2] : nullptr};
    for (int a = 0; a < prm.spaceDim; ++a) launch_set(st_, f_flux[a], L.field_elems, 0.0);
    launch_mac_correct(st_, L.dev, g, phi, 1.0);
    return f_flux;
}

// uStarFuncBC without inflow / outflow sides on the face-centred velocity (LevelMACProjector::computeDiv hands &m_divBC to
// levelDivergenceMAC, Divergence.cpp:73-100, which overwrites the caller's boundary faces): solid walls, zero normal flux
void PressureSolver::vel_wall_bc()
{
    double* e[3] = {vel(0), vel(1), prm.spaceDim == 3 ? vel(2) : nullptr};
    if (velbc_default_) launch_face_wall(st_, lev[0]->dev, e);
    else launch_face_bc(st_, lev[0]->dev, e, velbc_kind_, velbc_value_);
}

// BasicVelocityBCGhostClass's inflow / outflow sides (EllipticBCUtils.cpp:1244-1327): what vel_wall_bc and the
// cell-centred divergence's face BC apply from now on
void PressureSolver::set_vel_bc(const int kind[6], const double value[6])
{
    velbc_default_ = true;
    for (int i = 0; i < 6; ++i) {
        SOMAR_CHECK(kind[i] >= 0 && kind[i] <= 2, "velocity BC kind: 0 solid wall, 1 prescribed normal velocity, 2 outflow");
        velbc_kind_[i] = kind[i];
        velbc_value_[i] = kind[i] == 1 ? value[i] : 0.0;
        if (kind[i] != 0) velbc_default_ = false;
    }
}

void PressureSolver::mac_project(double dt, bool zeroPressure, bool forceHomogeneous, SolveStats& s)
{
    divergence_mac(f_rhs, dt);
    solve(zeroPressure, forceHomogeneous, s);
    mac_correct(f_phi, dt);
    sync();
}

// ------------------------------------------------------------------------------------
// Viscous / diffusive Helmholtz solves (single level).
//   MappedAMRPoissonOp::setAlphaAndBeta (MappedAMRPoissonOp.cpp:582-619) on every op of the hierarchy, as
//   MappedBaseLevelHeatSolver::resetSolverAlphaAndBeta does (MappedBaseLevelHeatSolver.cpp:257-270): alpha = a * aCoef,
//   beta = b * bCoef with aCoef / bCoef the alpha / beta the solver was created with.  The reference refills lapDiag with
//   the same values (it depends on neither coefficient) and keeps the prolongation strategy the factory's null-space
//   probe chose at construction; so does this.
// ------------------------------------------------------------------------------------
void PressureSolver::set_alpha_beta(double a, double b, bool amr_member_ok)
{
    SOMAR_CHECK(finalized, "set_alpha_beta before finalize");
    SOMAR_CHECK(!amr_member_ || amr_member_ok,
                "set_alpha_beta on a level of an AMR hierarchy goes through the hierarchy (somar_amr_set_alpha_beta)");
    if (!coefs_saved_) {
        aCoef_ = lev[0]->alpha;
        bCoef_ = lev[0]->beta;
        coefs_saved_ = true;
    }
    for (auto& L : lev) {
        L->alpha = a * aCoef_;
        L->beta = b * bCoef_;
        L->refresh_params();
    }
    drop_graphs();  // captured launches carry the old coefficients in their kernel arguments
    if (coarse_) coarse_->set_alpha_beta(a, b);
}

double* PressureSolver::heat_flux(int dir)
{
    SOMAR_CHECK(dir >= 0 && dir < prm.spaceDim && finalized, "bad direction / solver not finalized");
    if (!f_heatflux[dir]) f_heatflux[dir] = lev[0]->alloc_field();
    return f_heatflux[dir];
}

void PressureSolver::download_heat_flux(int dir, int patch, double* host)
{
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    IBox fb = L.boxes[L.local[patch]];
    fb.hi[dir] += 1;
    L.download(heat_flux(dir), patch, host, fb, fb, st_);
    sync();
}

void PressureSolver::increment_heat_flux(double* phi, bool setToZero)
{
    Level& L = *lev[0];
    double* const* G = full_ ? flux_fields(phi) : mac_grad(phi);
    for (int d = 0; d < prm.spaceDim; ++d) {
        double* acc = heat_flux(d);
        if (setToZero) launch_set(st_, acc, L.field_elems, 0.0);
        launch_incr(st_, acc, G[d], 1.0, L.field_elems);   // thisFlux += tempFlux
    }
}

double* PressureSolver::heat_field(int which)
{
    SOMAR_CHECK(which >= 0 && which < 3 && finalized, "bad heat field / solver not finalized");
    if (!f_heat[which]) f_heat[which] = lev[0]->alloc_field();
    return f_heat[which];
}

// MappedLevelBackwardEuler::updateSoln (AMRParabolic/MappedLevelBackwardEuler.cpp:52-158) and
// MappedLevelCrankNicolson::updateSoln (MappedLevelCrankNicolson.cpp:52-152) on one level, with applyHelm / solveHelm
// (MappedBaseLevelHeatSolver.cpp:154-255).  phiNew = F_PHI (initial guess unless zeroPhi), phiOld / src = heat fields.
//   BE: rhs = phiOld (the source term is commented out in the reference);   solve (aCoef I - dt bCoef L) phiNew = rhs
//   CN: rhs = dt src + (aCoef I + dt/2 bCoef L) phiOld [inhomogeneous BCs]; solve (aCoef I - dt/2 bCoef L) phiNew = rhs
//   TGA: (I - mu1 dt L)(I - mu2 dt L) phiNew = (I + mu3 dt L) phiOld + (I + mu4 dt L) dt src; stats are the last solve's
void PressureSolver::heat_step(int scheme, double dt, bool zeroPhi, SolveStats& s)
{
    SOMAR_CHECK(scheme >= 0 && scheme <= 2, "heat scheme: 0 backward Euler, 1 Crank-Nicolson, 2 TGA");
    SOMAR_CHECK(dt >= 0.0, "negative time step");
    const long long n = lev[0]->field_elems;
    double* phiOld = heat_field(0);
    if (scheme == 0) {
        launch_copy(st_, f_rhs, phiOld, n);
        set_alpha_beta(1.0, -dt * 1.0);
    } else if (scheme == 1) {
        set_alpha_beta(1.0, 0.5 * dt);
        apply_op(0, f_scratch[0], phiOld, false);
        launch_copy(st_, f_rhs, heat_field(1), n);
        launch_scale(st_, f_rhs, dt, n);
        launch_incr(st_, f_rhs, f_scratch[0], 1.0, n);
        set_alpha_beta(1.0, -dt * 0.5);
    } else {
        // MappedLevelTGA::updateSolnWithTimeIndependentOp, MappedLevelTGA.cpp:231-387; coefficients: its constructor, :30-56
        const double tgaEpsilon = 1.e-12;
        const double a = 2.0 - std::sqrt(2.0) - tgaEpsilon;
        const double discr = std::sqrt(a * a - 4.0 * a + 2.0);
        const double mu1 = (a - discr) / 2.0, mu2 = (a + discr) / 2.0, mu3 = 1.0 - a, mu4 = 0.5 - a;
        double* tmp = heat_field(2);              // srct, then phis (the caller's initial guess for both solves)
        launch_copy(st_, tmp, heat_field(1), n);  // srct = dt * src
        launch_scale(st_, tmp, dt, n);
        set_alpha_beta(1.0, mu4 * dt);
        apply_op(0, f_rhs, tmp, true);            // rhst = (I + mu4 dt L) srct, homogeneous BCs
        set_alpha_beta(1.0, mu3 * dt);
        apply_op(0, f_scratch[0], phiOld, false);  // (I + mu3 dt L) phiOld with the inhomogeneous BCs
        launch_incr(st_, f_rhs, f_scratch[0], 1.0, n);
        if (!zeroPhi) launch_copy(st_, tmp, f_phi, n);
        set_alpha_beta(1.0, -dt * mu2);
        solve(zeroPhi, false, s);                 // (I - mu2 dt L) phi* = rhst
        launch_copy(st_, f_rhs, f_phi, n);        // assign(rhst, phiNew)
        if (!zeroPhi) launch_copy(st_, f_phi, tmp, n);
        set_alpha_beta(1.0, -dt * mu1);           // (I - mu1 dt L) phiNew = phi*
    }
    solve(zeroPhi, false, s);
    sync();
}

// ------------------------------------------------------------------------------------
// Cell-centred level projection: BaseProjector<FArrayBox>::project (projection/BaseProjectorI.H:176-299) with
// LevelCCProjector::computeDiv/computeGrad/applyCorrection (LevelCCProjector.cpp:163-255), velocity in flux form.
// The velocity's ghost layer is the caller's (the reference does not exchange it before CellToEdge either).
// ------------------------------------------------------------------------------------
double* PressureSolver::cc_vel(int comp)
{
    SOMAR_CHECK(comp >= 0 && comp < prm.spaceDim && finalized, "bad velocity component / solver not finalized");
    if (!f_ccvel[comp]) f_ccvel[comp] = lev[0]->alloc_field();
    return f_ccvel[comp];
}

void PressureSolver::upload_cc_vel(int patch, const double* host, const int ghost[3])
{
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    const IBox valid = L.boxes[L.local[patch]];
    int one[3] = {0, 0, 0};
    for (int d = 0; d < prm.spaceDim; ++d) {
        SOMAR_CHECK(ghost[d] >= 1, "the cell-centred velocity needs one ghost layer (CellToEdge reads it)");
        one[d] = 1;
    }
    const IBox hb = valid.grow(ghost);
    for (int c = 0; c < prm.spaceDim; ++c) L.upload(cc_vel(c), patch, host + (long long)c * hb.numPts(), hb, valid.grow(one), st_);
    sync();
}

void PressureSolver::download_cc_vel(int patch, double* host, const int ghost[3])
{
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    const IBox valid = L.boxes[L.local[patch]];
    const IBox hb = valid.grow(ghost);
    for (int c = 0; c < prm.spaceDim; ++c) L.download(cc_vel(c), patch, host + (long long)c * hb.numPts(), hb, valid, st_);
    sync();
}

void PressureSolver::divergence_cc(double* out, double dt, bool wall)
{
    double* e[3] = {vel(0), vel(1), vel(2)};
    double* c[3] = {cc_vel(0), cc_vel(1), prm.spaceDim == 3 ? cc_vel(2) : nullptr};
    launch_cell_to_edge(st_, lev[0]->dev, e, c, wall && velbc_default_);  // Divergence::levelDivergenceCC, Divergence.cpp:361-396
    if (wall && !velbc_default_) {
        double* eb[3] = {e[0], e[1], prm.spaceDim == 3 ? e[2] : nullptr};
        launch_face_bc(st_, lev[0]->dev, eb, velbc_kind_, velbc_value_);
    }
    divergence_mac(out, dt);
}

void PressureSolver::cc_correct(double* phi, double dt)
{
    Level& L = *lev[0];
    xchg(L, phi);
    double* c[3] = {cc_vel(0), cc_vel(1), prm.spaceDim == 3 ? cc_vel(2) : nullptr};
    const double dtScale = dt == 0.0 ? -1.0 : -dt;
    if (full_) {
        mac_grad_full(phi);  // singleBoxMacGrad's sequence, as in mac_correct
        launch_edge_to_cell_axpy(st_, L.dev, c, f_flux, dtScale);
        return;
    }
    launch_cc_correct(st_, L.dev, c, phi, dtScale);
}

void PressureSolver::cc_project(double dt, bool zeroPressure, bool forceHomogeneous, bool wall, SolveStats& s)
{
    divergence_cc(f_rhs, dt, wall);
    solve(zeroPressure, forceHomogeneous, s);
    cc_correct(f_phi, dt);
    sync();
}

void PressureSolver::fill_hash(int d, double* f, unsigned long long seed)
{
    launch_fill_hash(st_, lev[d]->dev, f, seed);
}

// ------------------------------------------------------------------------------------
// MappedMultiGrid::cycle, MappedMultiGrid.H:555-653 (V/W cycles; F-cycle not offered)
// ------------------------------------------------------------------------------------
void PressureSolver::vcycle(double* e, const double* res, bool e_zero) { cycle(0, e, res, e_zero); }

// corr_zero: the correction is to be taken as zero whatever the array holds (the reference zeroes it with
// setToZero right before: MappedMultiGrid.H:589, MappedAMRMultiGrid.H:1203); a smoother that knows this skips the
// memset and the read.  Only honoured when at least one smoothing sweep will overwrite the whole array.
void PressureSolver::cycle_bottom_relax(double* corr, const double* res, bool corr_zero)
{
    const int d = (int)lev.size() - 1;
    if (lev[d]->domain.numPts() == 1) {
        relax(d, corr, res, 1, corr_zero);
        return;
    }
    if (corr_zero && prm.num_smooth_bottom == 0) launch_set(st_, corr, lev[d]->field_elems, 0.0);
    relax(d, corr, res, prm.num_smooth_bottom, corr_zero && prm.num_smooth_bottom > 0);
}

void PressureSolver::cycle_down(int d, double* corr, const double* res, bool corr_zero)
{
    if (corr_zero && prm.num_smooth_down == 0) launch_set(st_, corr, lev[d]->field_elems, 0.0);
    relax(d, corr, res, prm.num_smooth_down, corr_zero && prm.num_smooth_down > 0);
    restrict_residual(d, f_res[d + 1], corr, res);
}

void PressureSolver::cycle_up(int d, double* corr, const double* res)
{
    if (fold_prolong(d)) {
        // Large level: neither the prolongation nor the zero-average mean removal gets a pass of its own -- the first
        // post-smoothing sweep reads corr + coarse(i/r) - mean.  mean = (S_f + S_c) / V with S_f = sum dvol * corr
        // (gathered by the fused restriction), S_c = sum over coarse cells of coarse * (volume of its children), V the
        // level's volume: the same number as ZeroAvgConstInterpPS's sum dvol * (corr + coarse) / sum dvol up to the
        // association of the (already tree-ordered) sum.
        Level& F = *lev[d];
        Level& C = *lev[d + 1];
        xchg(C, f_corr[d + 1]);
        const double* shift = nullptr;
        if (F.zeroAvg) {
            double* s = d_fold + 8 * d;
            launch_reduce(st_, C.dev, f_corr[d + 1], f_W[d + 1], 0, d_partials, s + 1);
            comm_->allreduce(s, 2, 0, st_);
            launch_combine_sums(st_, s + 3, s, s + 1, s + 2);
            shift = s + 3;
        }
        relax(d, corr, res, prm.num_smooth_up, false, shift, &C, f_corr[d + 1]);
        return;
    }
    // the zero-average mean is folded into the first post-smoothing sweep when that sweep is the fused kernel
    const double* shift = prolong_increment(d, corr, f_corr[d + 1], fused_relax(d, prm.num_smooth_up));
    relax(d, corr, res, prm.num_smooth_up, false, shift);
}

void PressureSolver::cycle(int d, double* corr, const double* res, bool corr_zero)
{
    if (coarse_ && d == agglom_depth_) {
        agglom_cycle(corr, res, corr_zero);
        return;
    }
    const int D = (int)lev.size();
    if (mini_depth_ > 0 && d == mini_depth_ - 1) {
        // Bottom of a mini V-cycle: the reference smooths here (m_bottom sweeps) and then calls the AMR solver's
        // NoOpSolver, whose solve() is Chombo 3.1's "m_op->setToZero(a_phi)" -- so what goes back up is ZERO and the
        // sweeps are dead work.  Reproduced as the reference behaves (not as its name suggests): just the zero.
        (void)corr_zero;
        launch_set(st_, corr, lev[d]->field_elems, 0.0);
        return;
    }
    if (d == D - 1) {
        cycle_bottom_relax(corr, res, corr_zero);
        if (lev[d]->domain.numPts() != 1) bottom_solve(corr, res);
        return;
    }
    const int ncyc = cycle_override_ ? cycle_override_ : prm.numMG;
    SOMAR_CHECK(ncyc != 0, "numMG must not be 0");
    if (ncyc < 0) {
        // F-cycle (MappedMultiGrid.H:577-619): a recursive F-cycle first, pre-smoothing, then |numMG| V-cycles ("hack to
        // get a V-cycle": m_cycle = 1 around the inner call), post-smoothing.  No folding, no graphs: plain passes.
        const int cycles = -ncyc;
        Level& L = *lev[d];
        if (corr_zero) launch_set(st_, corr, L.field_elems, 0.0);
        restrict_residual(d, f_res[d + 1], corr, res);
        cycle(d + 1, f_corr[d + 1], f_res[d + 1], true);
        prolong_increment(d, corr, f_corr[d + 1]);
        relax(d, corr, res, prm.num_smooth_down);
        for (int img = 0; img < cycles; ++img) {
            restrict_residual(d, f_res[d + 1], corr, res);
            const int saved = cycle_override_;
            cycle_override_ = 1;
            try {
                cycle(d + 1, f_corr[d + 1], f_res[d + 1], true);
            } catch (...) {
                cycle_override_ = saved;
                throw;
            }
            cycle_override_ = saved;
            prolong_increment(d, corr, f_corr[d + 1]);
        }
        relax(d, corr, res, prm.num_smooth_up);
        return;
    }
    if (mini_depth_ == 0 && graph_cycle(d, corr, res, corr_zero)) return;
    cycle_down(d, corr, res, corr_zero);
    for (int img = 0; img < ncyc; ++img) cycle(d + 1, f_corr[d + 1], f_res[d + 1], img == 0);
    cycle_up(d, corr, res);
}

// MappedAMRMultiGrid::relax on a level whose refinement ratio to the coarser AMR level has an entry > 2
// (MappedAMRMultiGrid.H:742-754): a V-cycle over the forced depths only, nothing solved at its bottom.
void PressureSolver::mini_vcycle(double* corr, const double* res)
{
    SOMAR_CHECK(!forcedRatios.empty() && (int)lev.size() > (int)forcedRatios.size(), "no forced MG depths on this level");
    SOMAR_CHECK(!coarse_ || agglom_depth_ > (int)forcedRatios.size(), "internal: forced depths inside the replicated tail");
    mini_depth_ = (int)forcedRatios.size() + 1;
    try {
        cycle(0, corr, res, false);
    } catch (...) {
        mini_depth_ = 0;
        throw;
    }
    mini_depth_ = 0;
}

void PressureSolver::set_bc_values(const double v[6])
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_bc_values before define / after finalize");
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) bc_value_[d][s] = v[2 * d + s];
}

// one GHOST_DIRI op per (patch, Dirichlet side it touches): the face-adjacent ghost layer, setSideDiriBC's destBox
void PressureSolver::build_diri_ops(int d)
{
    Level& L = *lev[d];
    std::vector<GhostOp> ops;
    for (int pi = 0; pi < L.npatches(); ++pi) {
        const IBox valid = L.boxes[L.local[pi]];
        for (int a = 0; a < 3; ++a) {
            if (!L.active[a] || L.periodic[a]) continue;
            for (int s = 0; s < 2; ++s) {
                if (L.bc_type[a][s] != BC_DIRI) continue;
                if ((s ? valid.hi[a] : valid.lo[a]) != (s ? L.domain.hi[a] : L.domain.lo[a])) continue;
                GhostOp op;
                std::memset(&op, 0, sizeof(op));
                op.patch = pi;
                op.type = GHOST_DIRI;
                for (int q = 0; q < 3; ++q) { op.lo[q] = 0; op.n[q] = valid.size(q); }
                op.lo[a] = s ? valid.size(a) : -1;
                op.n[a] = 1;
                op.dir = a;
                op.sgn = s ? 1 : -1;
                op.val = bc_value_[a][s];
                ops.push_back(op);
            }
        }
    }
    n_diri_ops_[d] = (int)ops.size();
    if (!ops.empty()) {
        SOMAR_HIP(hipMalloc(&d_diri_ops_[d], ops.size() * sizeof(GhostOp)));
        SOMAR_HIP(hipMemcpy(d_diri_ops_[d], ops.data(), ops.size() * sizeof(GhostOp), hipMemcpyHostToDevice));
    }
}

void PressureSolver::apply_diri(int d, double* phi, bool homogeneous)
{
    launch_ghost_ops(st_, lev[d]->dev, d_diri_ops_[d], n_diri_ops_[d], phi, phi, homogeneous);
}

void PressureSolver::drop_graphs()
{
    if (cg_.down) hipGraphExecDestroy(cg_.down);
    if (cg_.up) hipGraphExecDestroy(cg_.up);
    cg_ = CoarseGraph();
}

// The V-cycle from depth d down to the bottom and back as two graph replays around the bottom solve.  Returns
// false when this call is not the one the graphs describe (the caller then runs the launches one by one).
bool PressureSolver::graph_cycle(int d, double* corr, const double* res, bool corr_zero)
{
    const int D = (int)lev.size();
    if (graph_from_ < 0 || d != graph_from_ || capturing_ || !corr_zero || prm.numMG != 1 || comm_->size != 1 ||
        coarse_ || profiling_ || d > D - 2)
        return false;
    if (cg_.down && (cg_.d0 != d || cg_.corr != corr || cg_.res != res || cg_.pre != prm.num_smooth_down ||
                     cg_.post != prm.num_smooth_up || cg_.bottom != prm.num_smooth_bottom))
        drop_graphs();
    if (!cg_.down) {
        auto capture = [&](bool down) -> hipGraphExec_t {
            hipGraph_t g = nullptr;
            capturing_ = true;
            SOMAR_HIP(hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal));
            try {
                if (down) {
                    cycle_down(d, corr, res, true);
                    for (int q = d + 1; q <= D - 2; ++q) cycle_down(q, f_corr[q], f_res[q], true);
                    cycle_bottom_relax(f_corr[D - 1], f_res[D - 1], true);
                } else {
                    for (int q = D - 2; q > d; --q) cycle_up(q, f_corr[q], f_res[q]);
                    cycle_up(d, corr, res);
                }
            } catch (...) {
                hipStreamEndCapture(st_, &g);
                if (g) hipGraphDestroy(g);
                capturing_ = false;
                throw;
            }
            capturing_ = false;
            SOMAR_HIP(hipStreamEndCapture(st_, &g));
            hipGraphExec_t e = nullptr;
            const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            SOMAR_HIP(rc);
            return e;
        };
        cg_.down = capture(true);
        cg_.up = capture(false);
        cg_.d0 = d;
        cg_.corr = corr;
        cg_.res = res;
        cg_.pre = prm.num_smooth_down;
        cg_.post = prm.num_smooth_up;
        cg_.bottom = prm.num_smooth_bottom;
    }
    SOMAR_HIP(hipGraphLaunch(cg_.down, st_));
    if (lev[D - 1]->domain.numPts() != 1) bottom_solve(f_corr[D - 1], f_res[D - 1]);
    SOMAR_HIP(hipGraphLaunch(cg_.up, st_));
    return true;
}

bool PressureSolver::fold_prolong(int d) const
{
    static const bool off = getenv("SOMAR_NO_FOLD_PROLONG") != nullptr;  // A/B switch
    const Level& F = *lev[d];
    if (off || !fused_relax(d, prm.num_smooth_up) || ordered(d) || hasCF_) return false;
    return !F.zeroAvg || sf_valid_[d];
}

// ------------------------------------------------------------------------------------
// Chombo 3.1 BiCGStabSolver<T>::solve (EXTERNAL to the reference; restated from the published
// algorithm, same variable names).  Runs on the coarsest MG depth.
// ------------------------------------------------------------------------------------
// A bottom level small enough for the one-launch BiCGStab (k_tiny_bicgstab): all boxes on this rank,
// the 7-point operator, Neumann / periodic sides, no coarse-fine faces, point GSRB) and serial-order sums.
// SOMAR_FUSED_BOTTOM_MAX_CELLS (default 512, 0 = off) is the A/B switch.
bool PressureSolver::fused_bottom(int d) const
{
    const long long maxc = fused_bottom_max_;   // SOMAR_FUSED_BOTTOM_MAX_CELLS, read once at construction
    static const bool poll = !(getenv("SOMAR_POLL_FETCH") && atoi(getenv("SOMAR_POLL_FETCH")) == 0);
    const Level& L = *lev[d];
    return maxc > 0 && poll && !full_ && !diri_ && L.ncf == 0 && L.plan.peers.empty() && !profiling_ && !capturing_ &&
           comm_->size == 1 && L.valid_cells_global <= maxc && ordered(d) && !ord_sharded(d) && L.dev.ntiles > 0 &&
           L.dev.tile_j >= 1 && L.dev.tile_j <= 16 && (1024 % (64 * L.dev.tile_j)) == 0 &&
           prm.relaxMode == RELAX_LEVEL_GSRB && prm.precondMode != PRECOND_DIAG_LINE_RELAX && bicg[7] != nullptr;
}

// A bottom level for the persistent one-workgroup-per-box BiCGStab (k_box_bicgstab): every box on this rank, the boxes cover
// the whole domain (no coarse-fine faces), the 7-point operator with Neumann / periodic sides, point GSRB, at most BOX_MAX_WG
// boxes of at most BOX_MAX_CELLS cells.
bool PressureSolver::box_bottom(int d) const
{
    static const bool poll = !(getenv("SOMAR_POLL_FETCH") && atoi(getenv("SOMAR_POLL_FETCH")) == 0);
    const Level& L = *lev[d];
    if (!box_bottom_on_ || !poll || diri_ || L.ncf != 0 || !L.plan.peers.empty() || profiling_ || capturing_ ||
        comm_->size != 1 || prm.relaxMode != RELAX_LEVEL_GSRB || prm.precondMode == PRECOND_DIAG_LINE_RELAX || !bicg[7])
        return false;
    // (the 19-point operator has no single-workgroup kernel: its bottoms come here whatever their size)
    if ((!full_ && L.valid_cells_global < box_min_cells_) || L.valid_cells_global != L.domain.numPts()) return false;
    if (L.npatches() < 1 || L.npatches() > BOX_MAX_WG || L.field_elems > 0x7fffffffll) return false;
    for (const PatchDesc& p : L.hpatches)
        if ((long long)p.n[0] * p.n[1] * p.n[2] > BOX_MAX_CELLS) return false;
    if (full_) {
        // the 19-point variant: 3-D, boxes of at most 256 cells and at least 2 wide (order-2 extrapolation reads three cells
        // back from a ghost), the box grown by one cell and its two ghost programs within the kernel's LDS arrays
        if (!L.active[2] || hasCF_) return false;
        const int d0 = (int)lev.size() - 1;
        if (d != d0 || (int)full_prog_.size() <= d) return false;
        for (const PatchDesc& p : L.hpatches) {
            if ((long long)p.n[0] * p.n[1] * p.n[2] > 256 || p.n[0] < 2 || p.n[1] < 2 || p.n[2] < 2) return false;
            if ((long long)(p.n[0] + 2) * (p.n[1] + 2) * (p.n[2] + 2) > BOX_FAB_MAX) return false;
        }
        if (box_depth_ == d && !box_full_ok_) return false;   // (tables built: a program did not fit the kernel's LDS arrays)
    }
    return true;
}

// For every valid cell of depth d (boxes back to back, Fortran order inside a box) the field offsets of the six cells its
// stencil reads: the neighbour itself, or the valid cell the level's ghost exchange copies into that ghost cell (the exchange
// plan applied to an identity map) -- k_box_bicgstab reads neighbouring boxes directly and never fills a ghost cell.
void PressureSolver::build_box_tables(int d)
{
    if (box_depth_ == d) return;
    SOMAR_CHECK(box_depth_ < 0, "box bottom solver tables built for another depth");
    const Level& L = *lev[d];
    std::vector<int> G((size_t)L.field_elems);
    for (size_t q = 0; q < G.size(); ++q) G[q] = (int)q;
    auto at = [](const PatchDesc& p, int i, int j, int k) { return p.off + i + (long long)p.pj * j + p.pk * k; };
    for (const CopyItem& it : L.plan.local) {
        const PatchDesc& sp = L.hpatches[it.src_patch];
        const PatchDesc& dp = L.hpatches[it.dst_patch];
        for (int k = 0; k < it.n[2]; ++k)
            for (int j = 0; j < it.n[1]; ++j)
                for (int i = 0; i < it.n[0]; ++i)
                    G[(size_t)at(dp, it.dst_lo[0] + i, it.dst_lo[1] + j, it.dst_lo[2] + k)] =
                        (int)at(sp, it.src_lo[0] + i, it.src_lo[1] + j, it.src_lo[2] + k);
    }
    std::vector<int> cstart(L.hpatches.size() + 1, 0), nb;
    box_max_cells_ = 0;
    for (size_t b = 0; b < L.hpatches.size(); ++b) {
        const PatchDesc& p = L.hpatches[b];
        const int cells = p.n[0] * p.n[1] * p.n[2];
        cstart[b + 1] = cstart[b] + cells;
        box_max_cells_ = std::max(box_max_cells_, cells);
        for (int k = 0; k < p.n[2]; ++k)
            for (int j = 0; j < p.n[1]; ++j)
                for (int i = 0; i < p.n[0]; ++i) {
                    const long long c = at(p, i, j, k);
                    const long long off[6] = {c - 1, c + 1, c - p.pj, c + p.pj, c - p.pk, c + p.pk};
                    for (int s = 0; s < 6; ++s)   // (a flat level has no z frame: those two entries are never read)
                        nb.push_back(off[s] >= 0 && off[s] < L.field_elems ? G[(size_t)off[s]] : (int)c);
                }
    }
    SOMAR_HIP(hipMalloc(&d_box_nb_, nb.size() * sizeof(int)));
    SOMAR_HIP(hipMalloc(&d_box_cstart_, cstart.size() * sizeof(int)));
    SOMAR_HIP(hipMalloc(&d_box_sums_, ((size_t)4 * BOX_MAX_WG + 8) * sizeof(double)));   // + 6 debug counters (SOMAR_BOX_TIMING)
    SOMAR_HIP(hipMalloc(&d_box_sync_, (BOX_MAX_WG + 1) * sizeof(unsigned)));
    if (full_) {
        // the box grown by one cell: where each of its cells' values comes from
        std::vector<int> fab, fstart(L.hpatches.size() + 1, 0);
        for (size_t b = 0; b < L.hpatches.size(); ++b) {
            const PatchDesc& p = L.hpatches[b];
            fstart[b] = (int)fab.size();
            for (int k = -1; k <= p.n[2]; ++k)
                for (int j = -1; j <= p.n[1]; ++j)
                    for (int i = -1; i <= p.n[0]; ++i) {
                        const long long c = at(p, i, j, k);
                        const bool inside = i >= 0 && i < p.n[0] && j >= 0 && j < p.n[1] && k >= 0 && k < p.n[2];
                        fab.push_back(inside ? (int)c : (G[(size_t)c] != (int)c ? G[(size_t)c] : -1));
                    }
        }
        fstart[L.hpatches.size()] = (int)fab.size();
        SOMAR_HIP(hipMalloc(&d_box_fab_, fab.size() * sizeof(int)));
        SOMAR_HIP(hipMalloc(&d_box_fabstart_, fstart.size() * sizeof(int)));
        SOMAR_HIP(hipMemcpy(d_box_fab_, fab.data(), fab.size() * sizeof(int), hipMemcpyHostToDevice));
        SOMAR_HIP(hipMemcpy(d_box_fabstart_, fstart.data(), fstart.size() * sizeof(int), hipMemcpyHostToDevice));
        // the two ghost programs ([0] operator, [1] smoother), one entry per written cell, stage after stage: what the kernel
        // runs on its LDS copy with one thread per entry (decoding ops and their regions there cost 20 us per program)
        box_full_ok_ = true;
        for (int w = 0; w < 2; ++w) {
            const FullProgram& Pg = full_prog_[d][w];
            std::vector<BoxProgEntry> ent;
            std::vector<int> efirst(L.hpatches.size() + 1, 0), stg, sfirst(L.hpatches.size() + 1, 0), nfg, nfirst(L.hpatches.size() + 1, 0);
            for (size_t b = 0; b < L.hpatches.size(); ++b) {
                const PatchDesc& p = L.hpatches[b];
                const int m0 = p.n[0] + 2, m01 = m0 * (p.n[1] + 2);
                const int fs[3] = {1, m0, m01};
                const long long gst[3] = {1, (long long)p.pj, p.pk};
                efirst[b] = (int)ent.size();
                sfirst[b] = (int)stg.size();
                nfirst[b] = (int)nfg.size();
                int cur = -1;
                const int o0 = Pg.h_box_first.empty() ? 0 : Pg.h_box_first[b], o1 = Pg.h_box_first.empty() ? 0 : Pg.h_box_first[b + 1];
                for (int o = o0; o < o1; ++o) {
                    const GhostOp& op = Pg.h_box_ops[o];
                    const int stage = op.pad_ & 0xffff;
                    if (stage != cur) { stg.push_back((int)ent.size() - efirst[b]); cur = stage; }
                    SOMAR_CHECK(op.type != GHOST_DIRI, "k_box_bicgstab: Dirichlet ghosts are not compiled (levels with Dirichlet sides take the launch path)");
                    for (int k = 0; k < op.n[2]; ++k)
                        for (int j = 0; j < op.n[1]; ++j)
                            for (int i = 0; i < op.n[0]; ++i) {
                                const int l0 = op.lo[0] + i, l1 = op.lo[1] + j, l2 = op.lo[2] + k;
                                const int f = (l0 + 1) + m0 * (l1 + 1) + m01 * (l2 + 1);
                                BoxProgEntry en;
                                std::memset(&en, 0, sizeof(en));
                                en.dst = (unsigned short)f;
                                en.flags = (unsigned char)((op.dstf ? 1 : 0) | (op.srcf ? 2 : 0));
                                if (op.type == GHOST_COPY) {
                                    en.kind = 0;
                                } else if (op.type == GHOST_EXTRAP) {
                                    const int dd = -op.sgn * fs[op.dir];
                                    en.kind = (unsigned char)(1 + op.order);
                                    en.s1 = (unsigned short)(f + dd);
                                    en.s2 = (unsigned short)(f + 2 * dd);
                                    en.s3 = (unsigned short)(f + 3 * dd);
                                } else {   // GHOST_NEUM: writes phi, reads psi around the face and the first valid cell of phi
                                    en.kind = 4;
                                    en.flags = (unsigned char)((op.dir << 2) | (op.sgn > 0 ? 16 : 0));
                                    en.nslot = (unsigned short)((int)nfg.size() - nfirst[b]);
                                    const long long face = at(p, l0, l1, l2) + (op.sgn < 0 ? gst[op.dir] : 0);
                                    nfg.push_back((int)((face << 2) | op.dir));
                                }
                                ent.push_back(en);
                            }
                }
                stg.push_back((int)ent.size() - efirst[b]);
                if ((int)ent.size() - efirst[b] > BOX_MAX_ENT || (int)stg.size() - sfirst[b] > BOX_MAX_STAGES + 1 ||
                    (int)nfg.size() - nfirst[b] > BOX_MAX_NEUM || L.field_elems >= (1ll << 28))
                    box_full_ok_ = false;
            }
            efirst[L.hpatches.size()] = (int)ent.size();
            sfirst[L.hpatches.size()] = (int)stg.size();
            nfirst[L.hpatches.size()] = (int)nfg.size();
            auto up = [](auto*& dptr, const auto& v) {
                using T = typename std::remove_reference<decltype(v)>::type::value_type;
                const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
                SOMAR_HIP(hipMalloc(&dptr, bytes));
                if (!v.empty()) SOMAR_HIP(hipMemcpy(dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
            };
            up(d_box_ent_[w], ent);
            up(d_box_entfirst_[w], efirst);
            up(d_box_stg_[w], stg);
            up(d_box_stgfirst_[w], sfirst);
            up(d_box_nfg_[w], nfg);
            up(d_box_nfgfirst_[w], nfirst);
        }
    }
    SOMAR_HIP(hipMemcpy(d_box_nb_, nb.data(), nb.size() * sizeof(int), hipMemcpyHostToDevice));
    SOMAR_HIP(hipMemcpy(d_box_cstart_, cstart.data(), cstart.size() * sizeof(int), hipMemcpyHostToDevice));
    box_depth_ = d;
}

void PressureSolver::bottom_solve(double* phi, const double* rhs)
{
    ++counters[3];
    const int d = (int)lev.size() - 1;
    const long long n = lev[d]->field_elems;
    if (!fused_bottom(d) && box_bottom(d)) build_box_tables(d);   // (the 19-point tables may turn out not to fit: asked again below)
    if (!fused_bottom(d) && box_bottom(d)) {
        Level& L = *lev[d];
        BoxBicg A;
        std::memset(&A, 0, sizeof(A));
        A.nb = d_box_nb_;
        A.cstart = d_box_cstart_;
        A.phi = phi;
        A.rhs = rhs;
        A.z[0] = bicg[4];
        A.z[1] = bicg[5];
        A.imax = prm.bottom_imax;
        A.numRestarts = prm.bottom_numRestarts;
        A.normType = prm.bottom_normType;
        A.precondIters = (prm.num_smooth_precond == 0 || prm.precondMode == PRECOND_NONE) ? 0 : prm.num_smooth_precond;
        A.eps = bottom_eps_eff;
        A.reps = prm.bottom_reps;
        A.hang = prm.bottom_hang;
        A.small = prm.bottom_small;
        A.metric = bottom_metric;
        A.sums = d_box_sums_;
        A.sync = d_box_sync_;
        A.serial = ordered(d) ? 1 : 0;
        static const bool box_timing = getenv("SOMAR_BOX_TIMING") != nullptr;
        A.dbg = box_timing ? reinterpret_cast<long long*>(d_box_sums_ + 4 * BOX_MAX_WG) : nullptr;
        A.full = full_ ? 1 : 0;
        if (full_) {
            A.fab_src = d_box_fab_;
            A.fab_start = d_box_fabstart_;
            for (int w = 0; w < 2; ++w) {
                A.ent[w] = d_box_ent_[w]; A.ent_first[w] = d_box_entfirst_[w];
                A.stg[w] = d_box_stg_[w]; A.stg_first[w] = d_box_stgfirst_[w];
                A.nfg[w] = d_box_nfg_[w]; A.nfg_first[w] = d_box_nfgfirst_[w];
            }
        }
        A.info = d_scalars + SLOT_TMP;
        A.pub = ScalarPublish{h_scalars + SLOT_TMP, h_seq_, ++fetch_seq_};
        launch_box_bicgstab(st_, L.dev, box_max_cells_, A);
        wait_published(A.pub.seq);
        bottom_iters = (int)h_scalars[SLOT_TMP];
        bottom_exit = (int)h_scalars[SLOT_TMP + 1];
        bottom_kind = 2;
        if (box_timing) {
            long long t[6];
            SOMAR_HIP(hipMemcpy(t, A.dbg, sizeof(t), hipMemcpyDeviceToHost));
            fprintf(stderr, "[somar box timing] iterations %d: ticks staging %lld, program %lld, barrier %lld, sums %lld, stagings %lld (100 MHz s_memtime)\n",
                    bottom_iters, t[0], t[1], t[3], t[4], t[5]);
        }
        return;
    }
    if (fused_bottom(d)) {
        Level& L = *lev[d];
        TinyBicg A;
        std::memset(&A, 0, sizeof(A));
        A.phi = phi;
        A.rhs = rhs;
        for (int q = 0; q < 8; ++q) A.w[q] = bicg[q];
        A.imax = prm.bottom_imax;
        A.numRestarts = prm.bottom_numRestarts;
        A.normType = prm.bottom_normType;
        A.precondIters = (prm.num_smooth_precond == 0 || prm.precondMode == PRECOND_NONE) ? 0 : prm.num_smooth_precond;
        A.eps = bottom_eps_eff;
        A.reps = prm.bottom_reps;
        A.hang = prm.bottom_hang;
        A.small = prm.bottom_small;
        A.metric = bottom_metric;
        A.info = d_scalars + SLOT_TMP;
        A.pub = ScalarPublish{h_scalars + SLOT_TMP, h_seq_, ++fetch_seq_};
        launch_tiny_bicgstab(st_, L.dev, L.d_local_items, (int)L.plan.local.size(), n, A);
        wait_published(A.pub.seq);
        bottom_iters = (int)h_scalars[SLOT_TMP];
        bottom_exit = (int)h_scalars[SLOT_TMP + 1];
        bottom_kind = 1;
        return;
    }
    double *r = bicg[0], *r_tilde = bicg[1], *e = bicg[2], *p = bicg[3], *p_tilde = bicg[4], *s_tilde = bicg[5],
           *t = bicg[6], *v = bicg[7];
    const int nt = prm.bottom_normType;
    int recount = 0;
    bottom_kind = 0;
    residual(d, r, phi, rhs);
    launch_copy(st_, r_tilde, r, n);
    launch_set(st_, e, n, 0.0);
    launch_set(st_, p_tilde, n, 0.0);
    launch_set(st_, s_tilde, n, 0.0);
    int i = 0;
    double rho[4] = {0, 0, 0, 0};
    double nrm[2];
    nrm[0] = norm(d, r, nt);
    double initial_norm = nrm[0];
    const double initial_rnorm = nrm[0];
    nrm[1] = nrm[0];
    double alpha[2] = {0, 0}, beta[2] = {0, 0}, omega[2] = {0, 0};
    bool init = true;
    int restarts = 0;
    if (bottom_metric > 0) initial_norm = bottom_metric;
    const double eps = bottom_eps_eff;
    bottom_exit = -1;
    while ((i < prm.bottom_imax && nrm[0] > eps * nrm[1]) && (nrm[1] > 0)) {
        ++i;
        nrm[1] = nrm[0];
        alpha[1] = alpha[0]; beta[1] = beta[0]; omega[1] = omega[0];
        rho[3] = rho[2]; rho[2] = rho[1];
        rho[1] = dot(d, r_tilde, r);
        if (rho[1] == 0.0) {
            launch_incr(st_, phi, e, 1.0, n);
            bottom_exit = 2;
            bottom_iters = i;
            return;
        }
        if (init) {
            launch_copy(st_, p, r, n);
            init = false;
        } else {
            beta[1] = (rho[1] / rho[2]) * (alpha[1] / omega[1]);
            launch_bicg_p(st_, p, v, r, beta[1], -beta[1] * omega[1], n);   // scale, incr, incr in one pass (same roundings)
        }
        pre_cond(d, p_tilde, p);
        apply_op(d, v, p_tilde);
        const double m = dot(d, r_tilde, v);
        alpha[0] = rho[1] / m;
        if (std::fabs(m) > prm.bottom_small * std::fabs(rho[1])) {
            launch_incr2(st_, r, v, -alpha[0], e, p_tilde, alpha[0], n);   // r -= alpha v; e += alpha p~ (independent)
            nrm[0] = norm(d, r, nt);
        } else {
            launch_set(st_, r, n, 0.0);
            nrm[0] = 0.0;
        }
        if (nrm[0] > eps * initial_norm && nrm[0] > prm.bottom_reps * initial_rnorm) {
            pre_cond(d, s_tilde, r);
            apply_op(d, t, s_tilde);
            // (t,r) and (t,t) in one host round trip
            if (ord_sharded(d)) {
                ordered_sums(d, t, r, 0, 0.0, d_scalars + SLOT_TMP);
                ordered_sums(d, t, t, 0, 0.0, d_scalars + SLOT_TMP + 1);
            } else {
                launch_reduce(st_, lev[d]->dev, t, r, 0, d_partials, d_scalars + SLOT_TMP, ordered(d));
                launch_reduce(st_, lev[d]->dev, t, t, 0, d_partials, d_scalars + SLOT_TMP + 1, ordered(d));
                comm_->allreduce(d_scalars + SLOT_TMP, 2, 0, st_);
            }
            fetch_scalars(SLOT_TMP, 2);
            const double tr = h_scalars[SLOT_TMP], tt = h_scalars[SLOT_TMP + 1];
            omega[0] = tr / tt;
            launch_incr2(st_, e, s_tilde, omega[0], r, t, -omega[0], n);
            nrm[0] = norm(d, r, nt);
        }
        if (nrm[0] <= eps * initial_norm || nrm[0] <= prm.bottom_reps * initial_rnorm) {
            bottom_exit = 1;
            break;
        }
        if (omega[0] == 0.0 || nrm[0] > (1 - prm.bottom_hang) * nrm[1]) {
            if (recount == 0) {
                recount = 1;
            } else {
                recount = 0;
                launch_incr(st_, phi, e, 1.0, n);
                if (restarts == prm.bottom_numRestarts) {
                    bottom_exit = 3;
                    bottom_iters = i;
                    return;
                }
                residual(d, r, phi, rhs);
                nrm[0] = norm(d, r, nt);
                rho[0] = rho[1] = rho[2] = rho[3] = 0.0;
                alpha[0] = beta[0] = omega[0] = 0.0;
                launch_copy(st_, r_tilde, r, n);
                launch_set(st_, e, n, 0.0);
                ++restarts;
                init = true;
            }
        }
    }
    launch_incr(st_, phi, e, 1.0, n);
    bottom_iters = i;
}

// ------------------------------------------------------------------------------------
// MappedAMRMultiGrid::solveNoInitResid for l_base == l_max == 0, MappedAMRMultiGrid.H:979-1183
// ------------------------------------------------------------------------------------
void PressureSolver::solve(bool zeroPhi, bool forceHomogeneous, SolveStats& s)
{
    SOMAR_CHECK(finalized, "solve before finalize");
    Level& L = *lev[0];
    const long long n = L.field_elems;
    launch_set(st_, f_uberRes, n, 0.0);
    launch_set(st_, f_uberCorr, n, 0.0);
    if (zeroPhi) launch_set(st_, f_phi, n, 0.0);
    launch_copy(st_, f_best, f_phi, n);
    residual(0, f_uberRes, f_phi, f_rhs, forceHomogeneous);  // computeAMRResidual(..., a_forceHomogeneous), :1021
    double initial_rnorm = norm(0, f_uberRes, 0);
    double rnorm = initial_rnorm, norm_last = 2 * initial_rnorm, best_rnorm = rnorm;
    bool useBestPhi = false, somethingConverged = false;
    bottom_metric = initial_rnorm;          // setConvergenceMetrics(initial_rnorm, cushion*eps), :1047
    bottom_eps_eff = 1.0 * prm.eps;
    int iter = 0;
    s = SolveStats();
    s.history.push_back(rnorm);
    bool goNorm = rnorm > prm.normThresh;
    bool goRedu = rnorm > prm.eps * initial_rnorm;
    bool goIter = iter < prm.imax;
    bool goHang = iter < prm.imin || rnorm < (1 - prm.hang) * norm_last;
    while (goIter && goRedu && goHang && goNorm) {
        norm_last = rnorm;
        vcycle(f_uberCorr, f_uberRes, true);          // uberCorrection is zero here (setToZero, :1203)
        launch_incr(st_, f_phi, f_uberCorr, 1.0, n);   // postVCycleOps, :1189-1215
        residual(0, f_uberRes, f_phi, f_rhs, forceHomogeneous);
        rnorm = norm(0, f_uberRes, 0);
        ++iter;
        s.history.push_back(rnorm);
        if (rnorm <= best_rnorm) {
            best_rnorm = rnorm;
            launch_copy(st_, f_best, f_phi, n);
            useBestPhi = false;
            somethingConverged = true;
        } else {
            useBestPhi = true;
        }
        goNorm = rnorm > prm.normThresh;
        goRedu = rnorm > prm.eps * initial_rnorm;
        goIter = iter < prm.imax;
        goHang = iter < prm.imin || rnorm < (1 - prm.hang) * norm_last;
    }
    if (useBestPhi) {
        rnorm = best_rnorm;
        launch_copy(st_, f_phi, f_best, n);
    }
    s.status = 0;
    if (rnorm > 10. * initial_rnorm && rnorm > 10. * prm.eps) s.status = 1;                        // "kaboom" :1134
    else if (!somethingConverged && rnorm >= initial_rnorm && rnorm >= prm.eps) s.status = 2;      // :1141
    s.exitStatus = int(!goRedu) + int(!goIter) * 2 + int(!goHang) * 4 + int(!goNorm) * 8;
    s.iters = iter;
    s.initial_rnorm = initial_rnorm;
    s.final_rnorm = rnorm;
    s.bottom_iters_last = bottom_iters;
    s.bottom_exit_last = bottom_exit;
    sync();
}

}  // namespace somar
