// somar_amd/csrc/amr.cpp -- see amr.h for the reference map.
#include "amr.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace somar {

template <class T>
static T* to_device(const std::vector<T>& v)
{
    if (v.empty()) return nullptr;
    T* d = nullptr;
    SOMAR_HIP(hipMalloc(&d, v.size() * sizeof(T)));
    SOMAR_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

static IBox adj_cell(const IBox& b, int d, int s)
{
    IBox g = b;
    if (s == 0) { g.lo[d] = b.lo[d] - 1; g.hi[d] = b.lo[d] - 1; }
    else { g.lo[d] = b.hi[d] + 1; g.hi[d] = b.hi[d] + 1; }
    return g;
}
static bool contains(const IBox& b, const int iv[3])
{
    for (int d = 0; d < 3; ++d)
        if (iv[d] < b.lo[d] || iv[d] > b.hi[d]) return false;
    return true;
}

AMRLink::~AMRLink()
{
    Level::free_field(buf);
    Level::free_field(resC);
    hipFree(d_cc); hipFree(d_pts); hipFree(d_fc); hipFree(d_der); hipFree(d_cover); hipFree(d_reg); hipFree(d_regvals);
    hipFree(d_reflux); hipFree(d_A); hipFree(d_B); hipFree(d_sendidx); hipFree(d_sendbuf); hipFree(d_osg);
}

// ------------------------------------------------------------------------------------
// AMRSolver
// ------------------------------------------------------------------------------------
// (lives here for FillItem / launch_fill_items, which the AMR tables introduced)
// CartesianMap::fill_Jgup / fill_Jinv: the same regions set_metric_ortho uploads (valid faces / cells of every local patch),
// written by a fill kernel
void PressureSolver::set_metric_uniform(const double c4[4])
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_metric before define / after finalize");
    SOMAR_CHECK(!full_, "set_metric_uniform is the diagonal-metric (Cartesian) producer");
    Level& L = *lev[0];
    for (int a = 0; a < 4; ++a) {
        if (a < 3 && a >= prm.spaceDim) continue;
        std::vector<FillItem> items;
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox valid = L.boxes[L.local[pi]];
            FillItem it;
            std::memset(&it, 0, sizeof(it));
            it.patch = pi;
            for (int d = 0; d < 3; ++d) { it.lo[d] = 0; it.n[d] = valid.size(d) + ((a < 3 && d == a) ? 1 : 0); }
            items.push_back(it);
        }
        FillItem* d_items = to_device(items);
        launch_fill_items(st_, L.dev.patches, d_items, (int)items.size(), a < 3 ? L.dev.jg[a] : L.dev.jinv, c4[a]);
        sync();
        hipFree(d_items);
    }
}

AMRSolver::AMRSolver(Comm* comm) : comm_(comm ? comm : &self_)
{
    SOMAR_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
}

AMRSolver::~AMRSolver()
{
    if (st_) hipStreamSynchronize(st_);
    for (double* f : spare_) Level::free_field(f);
    links_.clear();
    leptic_.clear();
    S.clear();
    if (st_) hipStreamDestroy(st_);
}

void AMRSolver::define(const IBox& domain0, const bool periodic[3], const double dx0[3], const int bc_type[3][2],
                       const std::vector<std::array<int, 3>>& ratios, const std::vector<std::vector<IBox>>& boxes,
                       const std::vector<std::vector<int>>& owners, double alpha, double beta, const SolverParams& p)
{
    SOMAR_CHECK(S.empty(), "AMR solver already defined");
    const int n = (int)boxes.size();
    SOMAR_CHECK(n >= 1 && (int)ratios.size() >= n - 1 && (int)owners.size() == n, "bad level count");
    prm = p;
    SOMAR_CHECK(p.spaceDim == 3 || p.spaceDim == 2, "space_dim must be 2 or 3");
    if (p.spaceDim == 2)
        for (const auto& r : ratios) SOMAR_CHECK(r[2] == 1, "space_dim 2: the refinement ratio in z must be 1");
    ratios_ = ratios;
    IBox dom = domain0;
    double dx[3] = {dx0[0], dx0[1], dx0[2]}, dxc[3] = {0, 0, 0};
    for (int l = 0; l < n; ++l) {
        if (l > 0) {
            const int* r = ratios[l - 1].data();
            for (int d = 0; d < 3; ++d) {
                SOMAR_CHECK(r[d] == 1 || r[d] == 2 || r[d] == 4, "refinement ratios are 1, 2 or 4 per direction");
                dxc[d] = dx[d];
                dx[d] = dx[d] / (double)r[d];
            }
            dom = dom.refine(r);
            // proper nesting as far as the tables need it: every fine box coarsens exactly
            SOMAR_CHECK(coarsenable(boxes[l], r), "fine boxes must be coarsenable by the refinement ratio");
            for (const IBox& b : boxes[l])
                for (int d = 0; d < 3; ++d)
                    SOMAR_CHECK(b.lo[d] >= dom.lo[d] && b.hi[d] <= dom.hi[d], "fine box outside the domain");
        }
        std::unique_ptr<PressureSolver> ps(new PressureSolver(comm_, st_));
        ps->define(dom, periodic, dx, bc_type, boxes[l], owners[l], alpha, beta, p, l > 0 ? dxc : nullptr);
        ps->set_amr_member();
        if (l > 0) {
            // the mini V-cycle's coarsening pattern, MappedAMRMultiGrid.H:1455-1482 (anisotropic coarsening first)
            int r[3] = {ratios[l - 1][0], ratios[l - 1][1], ratios[l - 1][2]};
            std::vector<std::array<int, 3>> all;
            while (r[0] > 2 || r[1] > 2 || r[2] > 2) {
                std::array<int, 3> t = {1, 1, 1};
                for (int d = 0; d < 3; ++d)
                    if (r[d] > 2) { r[d] /= 2; t[d] = 2; }
                if (t[0] * t[1] * t[2] > 1) all.push_back(t);
            }
            ps->forcedRatios.assign(all.rbegin(), all.rend());
        }
        S.push_back(std::move(ps));
    }
}

void AMRSolver::finalize()
{
    SOMAR_CHECK(!S.empty() && !finalized_, "finalize before define / twice");
    const int n = nlevels();
    for (auto& s : S) s->finalize();
    corr_.assign(n, nullptr);
    res_.assign(n, nullptr);
    for (int l = 0; l < n; ++l) {
        corr_[l] = S[l]->amr_field(0);
        res_[l] = S[l]->amr_field(1);
    }
    {
        const char* e = getenv("SOMAR_AMR_PLAIN");
        lean_ = !(e && atoi(e) != 0);
    }
    spare_.assign(n, nullptr);
    rcur_ = res_;
    visits_.assign(n, 0);
    // the residual of an intermediate level ping-pongs between res_ and spare_ in AMRUpdateResidual (the finest level's
    // between the caller's uberResidual and res_; the base level's is never updated)
    if (lean_)
        for (int l = 1; l + 1 < n; ++l) spare_[l] = S[l]->level(0).alloc_field();
    links_.resize(n);
    for (int l = 1; l < n; ++l) build_link(l);
    SOMAR_HIP(hipDeviceSynchronize());  // tables were uploaded with plain hipMemcpy (null stream)
    finalized_ = true;
}

void AMRSolver::build_link(int l)
{
    const auto tl0 = std::chrono::steady_clock::now();
    std::unique_ptr<AMRLink> K(new AMRLink);
    Level& F = S[l]->level(0);
    Level& C = S[l - 1]->level(0);
    for (int d = 0; d < 3; ++d) K->r[d] = ratios_[l - 1][d];
    std::vector<IBox> cb;
    for (const IBox& b : F.boxes) cb.push_back(b.coarsen(K->r));
    K->cfl.reset(new Level);
    K->cfl->active[2] = C.active[2];
    K->cfl->define(C.domain, C.periodic, C.dx, C.bc_type, cb, F.owner, comm_);
    K->buf = K->cfl->alloc_field();
    K->resC = K->cfl->alloc_field();
    const int g2[3] = {2, 2, 2}, g0[3] = {0, 0, 0};
    K->gather.define(C.domain, C.periodic, C, *K->cfl, g2, comm_);
    K->gather_ring.define(C.domain, C.periodic, C, *K->cfl, g2, comm_, true);
    K->scatter.define(C.domain, C.periodic, *K->cfl, C, g0, comm_);
    // zeroCovered: the coarse cells under the fine level
    std::vector<FillItem> cover;
    for (int pi = 0; pi < C.npatches(); ++pi) {
        const IBox valid = C.boxes[C.local[pi]];
        for (const IBox& b : cb) {
            const IBox reg = b & valid;
            if (reg.empty()) continue;
            FillItem it;
            std::memset(&it, 0, sizeof(it));
            it.patch = pi;
            for (int d = 0; d < 3; ++d) { it.lo[d] = reg.lo[d] - valid.lo[d]; it.n[d] = reg.size(d); }
            cover.push_back(it);
        }
    }
    K->ncover = (int)cover.size();
    K->d_cover = to_device(cover);
    if (getenv("SOMAR_TIMING") && atoi(getenv("SOMAR_TIMING")) != 0)
        fprintf(stderr, "[somar timing] link %d: coarsened-fine layout + copiers %.3f s\n", l,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count());
    links_[l] = std::move(K);
    static const bool timing = getenv("SOMAR_TIMING") && atoi(getenv("SOMAR_TIMING")) != 0;
    const auto t0 = std::chrono::steady_clock::now();
    build_quad_tables(l);
    const auto t1 = std::chrono::steady_clock::now();
    build_reflux_tables(l);
    const auto t2 = std::chrono::steady_clock::now();
    if (timing)
        fprintf(stderr, "[somar timing] link %d: quad CF tables %.3f s, reflux tables %.3f s\n", l,
                std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
}

// ------------------------------------------------------------------------------------
// MappedQuadCFInterp::define + MappedQuadCFStencil::define/buildStencils as flat tables
// ------------------------------------------------------------------------------------
namespace {
struct Mask {
    IBox b;
    std::vector<char> m;
    void define(const IBox& box, char v)
    {
        b = box;
        m.assign((size_t)std::max<long long>(box.numPts(), 0), v);
    }
    size_t idx(const int iv[3]) const
    {
        return (size_t)(iv[0] - b.lo[0]) + (size_t)b.size(0) * ((size_t)(iv[1] - b.lo[1]) + (size_t)b.size(1) * (size_t)(iv[2] - b.lo[2]));
    }
    bool at(const int iv[3]) const { return contains(b, iv) && m[idx(iv)]; }
    void set_box(const IBox& r0, char v)
    {
        const IBox r = r0 & b;
        if (r.empty()) return;
        int iv[3];
        for (iv[2] = r.lo[2]; iv[2] <= r.hi[2]; ++iv[2])
            for (iv[1] = r.lo[1]; iv[1] <= r.hi[1]; ++iv[1])
                for (iv[0] = r.lo[0]; iv[0] <= r.hi[0]; ++iv[0]) m[idx(iv)] = v;
    }
};
}  // namespace

namespace {
// "is this cell inside one of these boxes" in O(1): the boxes binned into a uniform grid of buckets over their bounding box
// (table builders asked that once per surface cell with a loop over the whole layout: 1.7 s of C4's 2 s re-definition)
class BoxIndex {
public:
    explicit BoxIndex(const std::vector<IBox>& boxes) : boxes_(boxes)
    {
        if (boxes.empty()) return;
        bb_ = boxes[0];
        long long ext[3] = {0, 0, 0};
        for (const IBox& b : boxes) {
            for (int d = 0; d < 3; ++d) {
                bb_.lo[d] = std::min(bb_.lo[d], b.lo[d]);
                bb_.hi[d] = std::max(bb_.hi[d], b.hi[d]);
                ext[d] += b.size(d);
            }
        }
        for (int d = 0; d < 3; ++d) {
            w_[d] = (int)std::max<long long>(1, ext[d] / (long long)boxes.size());   // the mean box extent
            nb_[d] = (bb_.size(d) + w_[d] - 1) / w_[d];
        }
        bins_.resize((size_t)nb_[0] * nb_[1] * nb_[2]);
        for (int i = 0; i < (int)boxes.size(); ++i) {
            int lo[3], hi[3];
            for (int d = 0; d < 3; ++d) {
                lo[d] = (boxes[i].lo[d] - bb_.lo[d]) / w_[d];
                hi[d] = (boxes[i].hi[d] - bb_.lo[d]) / w_[d];
            }
            for (int c = lo[2]; c <= hi[2]; ++c)
                for (int b = lo[1]; b <= hi[1]; ++b)
                    for (int a = lo[0]; a <= hi[0]; ++a) bins_[a + (size_t)nb_[0] * (b + (size_t)nb_[1] * c)].push_back(i);
        }
    }
    int find(const int iv[3]) const
    {
        if (boxes_.empty() || !contains(bb_, iv)) return -1;
        const size_t q = (size_t)((iv[0] - bb_.lo[0]) / w_[0]) +
                         (size_t)nb_[0] * ((size_t)((iv[1] - bb_.lo[1]) / w_[1]) + (size_t)nb_[1] * (size_t)((iv[2] - bb_.lo[2]) / w_[2]));
        for (int i : bins_[q])
            if (contains(boxes_[i], iv)) return i;
        return -1;
    }
    bool has(const int iv[3]) const { return find(iv) >= 0; }

private:
    const std::vector<IBox>& boxes_;
    IBox bb_;
    int w_[3] = {1, 1, 1}, nb_[3] = {0, 0, 0};
    std::vector<std::vector<int>> bins_;
};
}  // namespace

void AMRSolver::build_quad_tables(int l)
{
    AMRLink& K = *links_[l];
    Level& F = S[l]->level(0);
    Level& C = S[l - 1]->level(0);
    const int* r = K.r;
    // m_level < 0-equivalent: the fine level covers the whole coarse level -> no CF interpolation
    long long nf = 0, nc = 0;
    for (const IBox& b : F.boxes) nf += b.numPts();
    for (const IBox& b : C.boxes) nc += b.numPts();
    K.hasCF = (nf / ((long long)r[0] * r[1] * r[2])) != nc;
    if (!K.hasCF) return;

    const IBox fdom = F.domain, cdom = C.domain;
    int gp1[3], gp6[3];
    for (int d = 0; d < 3; ++d) { gp1[d] = F.periodic[d] ? 1 : 0; gp6[d] = F.periodic[d] ? 6 : 0; }
    const IBox fdomG = fdom.grow(gp1), cdomG1 = cdom.grow(gp1), cdomG6 = cdom.grow(gp6);
    const auto fshifts = periodic_shifts(fdom, F.periodic);
    const auto cshifts = periodic_shifts(cdom, C.periodic);
    // coarsened images of every fine box, images of every coarse box
    std::vector<IBox> crseImgsOfFine, coarImgs;
    for (const auto& sh : fshifts)
        for (const IBox& b : F.boxes) crseImgsOfFine.push_back(b.shift(sh.data()).coarsen(r));
    for (const auto& sh : cshifts)
        for (const IBox& b : C.boxes) coarImgs.push_back(b.shift(sh.data()));

    std::vector<QCoarse> ccs;
    std::vector<QPoint> pts;
    std::vector<QFine> fcs;

    for (int pi = 0; pi < F.npatches(); ++pi) {
        const IBox grid = F.boxes[F.local[pi]];
        const PatchDesc& fp = F.hpatches[pi];
        const PatchDesc& cp = K.cfl->hpatches[pi];
        const long long fst[3] = {1, fp.pj, fp.pk};
        const long long cst[3] = {1, cp.pj, cp.pk};
        const IBox coarseGrid = grid.coarsen(r);
        // CH_SPACEDIM = 2 (MappedQuadCFInterp.cpp:300-310, 386-400): one tangential direction, no mixed derivative.  The
        // records keep their layout; the z slots (t2, mixed) are empty for non-standard cells and multiply the zero
        // offset x2 = 0 for standard ones (the buffer's z ghost planes hold zeros).
        const bool flat = !F.active[2];
        for (int dir = 0; dir < (flat ? 2 : 3); ++dir) {
            const int t1 = dir == 0 ? 1 : 0, t2 = dir == 2 ? 1 : 2;
            for (int s = 0; s < 2; ++s) {
                const IBox edge = adj_cell(grid, dir, s) & fdomG;
                if (edge.empty()) continue;
                const std::vector<IBox> fineIVS = uncovered(edge, F.boxes, fdom, F.periodic);
                if (fineIVS.empty()) continue;
                long long nunc = 0;
                for (const IBox& u : fineIVS) nunc += u.numPts();
                const bool packed = nunc == edge.numPts();
                // coarse IVS = coarsen(fine IVS) & (coarse domain grown 1 in periodic directions)
                Mask coar;
                coar.define(edge.coarsen(r) & cdomG1, 0);
                if (coar.b.empty()) continue;
                for (const IBox& u : fineIVS) coar.set_box(u.coarsen(r), 1);
                // "all good" coarse cells: not under the fine level, covered by the coarse level
                IBox g2 = adj_cell(coarseGrid, dir, s), g1 = g2;
                g2.lo[t1] -= 2; g2.hi[t1] += 2;
                g1.lo[t1] -= 1; g1.hi[t1] += 1;
                if (!flat) {
                    g2.lo[t2] -= 2; g2.hi[t2] += 2;
                    g1.lo[t2] -= 1; g1.hi[t2] += 1;
                }
                g2 = g2 & cdomG6;
                g1 = g1 & cdomG6;
                Mask good;
                good.define(g2, 0);
                for (const IBox& b : coarImgs) good.set_box(b, 1);
                for (const IBox& b : crseImgsOfFine) good.set_box(b, 0);
                // standard = good within g1, shrunk by one in each tangential direction
                Mask stdm;
                stdm.define(g2, 0);
                {
                    const IBox r1 = g1 & g2;
                    int iv[3];
                    for (iv[2] = r1.lo[2]; iv[2] <= r1.hi[2]; ++iv[2])
                        for (iv[1] = r1.lo[1]; iv[1] <= r1.hi[1]; ++iv[1])
                            for (iv[0] = r1.lo[0]; iv[0] <= r1.hi[0]; ++iv[0]) stdm.m[stdm.idx(iv)] = good.m[good.idx(iv)];
                    for (int t : {t1, t2}) {
                        if (flat && t == t2) continue;
                        Mask nxt = stdm;
                        for (iv[2] = g2.lo[2]; iv[2] <= g2.hi[2]; ++iv[2])
                            for (iv[1] = g2.lo[1]; iv[1] <= g2.hi[1]; ++iv[1])
                                for (iv[0] = g2.lo[0]; iv[0] <= g2.hi[0]; ++iv[0]) {
                                    int a[3] = {iv[0], iv[1], iv[2]}, b[3] = {iv[0], iv[1], iv[2]};
                                    a[t] -= 1;
                                    b[t] += 1;
                                    nxt.m[nxt.idx(iv)] = stdm.m[stdm.idx(iv)] && stdm.at(a) && stdm.at(b);
                                }
                        stdm = nxt;
                    }
                }
                // ---- coarse records ----
                std::map<long long, int> ccIndex;  // buffer offset -> record
                int iv[3];
                for (iv[2] = coar.b.lo[2]; iv[2] <= coar.b.hi[2]; ++iv[2])
                    for (iv[1] = coar.b.lo[1]; iv[1] <= coar.b.hi[1]; ++iv[1])
                        for (iv[0] = coar.b.lo[0]; iv[0] <= coar.b.hi[0]; ++iv[0]) {
                            if (!coar.m[coar.idx(iv)]) continue;
                            QCoarse q;
                            std::memset(&q, 0, sizeof(q));
                            q.boff = cp.off + (iv[0] - cp.lo[0]) + cst[1] * (iv[1] - cp.lo[1]) + cst[2] * (iv[2] - cp.lo[2]);
                            for (int d = 0; d < 3; ++d)
                                SOMAR_CHECK(iv[d] - cp.lo[d] >= -1 && iv[d] - cp.lo[d] <= cp.n[d], "coarse CF cell outside the buffer ring");
                            q.dir = dir;
                            q.s1 = (int)cst[t1];
                            q.s2 = (int)cst[t2];
                            q.p0 = (int)pts.size();
                            auto G = [&](int a, int b) {
                                int w[3] = {iv[0], iv[1], iv[2]};
                                w[t1] += a;
                                w[t2] += b;
                                return good.at(w);
                            };
                            if (stdm.at(iv)) {
                                q.flags = 1;
                            } else {
                                bool drop = false;
                                // mixed derivative first (it may set dropOrd), MappedCFStencil.cpp:1010-1075
                                std::vector<std::pair<std::pair<int, int>, double>> mix;
                                int nused = 0;
                                const int quad[4][2] = {{-1, 0}, {0, 0}, {0, -1}, {-1, -1}};
                                for (int qd = 0; qd < (flat ? 0 : 4); ++qd) {
                                    const int la = quad[qd][0], lb = quad[qd][1];
                                    if (!(G(la, lb) && G(la + 1, lb) && G(la, lb + 1) && G(la + 1, lb + 1))) continue;
                                    ++nused;
                                    for (int db = 0; db < 2; ++db)
                                        for (int da = 0; da < 2; ++da) {
                                            const double w = (da == db) ? -1.0 : 1.0;
                                            const std::pair<int, int> key(la + da, lb + db);
                                            bool found = false;
                                            for (auto& e : mix)
                                                if (e.first == key) { e.second += w; found = true; break; }
                                            if (!found) mix.push_back({key, w});
                                        }
                                }
                                if (nused == 0 && !flat) drop = true;
                                // first/second derivatives per tangential direction, :1077-1178
                                std::vector<QPoint> d1[2], d2[2];
                                for (int ti = 0; ti < (flat ? 1 : 2); ++ti) {
                                    const long long sst = ti == 0 ? cst[t1] : cst[t2];
                                    auto g = [&](int k) { return ti == 0 ? G(k, 0) : G(0, k); };
                                    auto P = [&](std::vector<QPoint>& v, int k, double w) { v.push_back({k * sst, w}); };
                                    if (drop) continue;  // dropped before this direction: no stencils
                                    if (g(-1) && g(0) && g(1)) {
                                        P(d2[ti], -1, 1.0); P(d2[ti], 0, -2.0); P(d2[ti], 1, 1.0);
                                        P(d1[ti], -1, -0.5); P(d1[ti], 0, 0.0); P(d1[ti], 1, 0.5);
                                    } else if (g(0) && g(1) && g(2)) {
                                        P(d2[ti], 0, 1.0); P(d2[ti], 1, -2.0); P(d2[ti], 2, 1.0);
                                        P(d1[ti], 0, -3.0 / 2.0); P(d1[ti], 1, 4.0 / 2.0); P(d1[ti], 2, -1.0 / 2.0);
                                    } else if (g(-2) && g(-1) && g(0)) {
                                        P(d2[ti], -2, 1.0); P(d2[ti], -1, -2.0); P(d2[ti], 0, 1.0);
                                        P(d1[ti], -2, 1.0 / 2.0); P(d1[ti], -1, -4.0 / 2.0); P(d1[ti], 0, 3.0 / 2.0);
                                    } else {
                                        drop = true;  // m_dropOrd(iv) = true
                                        if (g(1)) { P(d1[ti], 0, -1.0); P(d1[ti], 1, 1.0); }
                                        else if (g(-1)) { P(d1[ti], -1, -1.0); P(d1[ti], 0, 1.0); }
                                        else P(d1[ti], 0, 0.0);
                                    }
                                }
                                if (drop) { d2[0].clear(); d2[1].clear(); mix.clear(); }
                                q.np[0] = (int)d1[0].size(); q.np[1] = (int)d2[0].size();
                                q.np[2] = (int)d1[1].size(); q.np[3] = (int)d2[1].size();
                                q.np[4] = (int)mix.size();
                                for (auto& e : d1[0]) pts.push_back(e);
                                for (auto& e : d2[0]) pts.push_back(e);
                                for (auto& e : d1[1]) pts.push_back(e);
                                for (auto& e : d2[1]) pts.push_back(e);
                                for (auto& e : mix)
                                    pts.push_back({e.first.first * cst[t1] + e.first.second * cst[t2], e.second / (double)nused});
                            }
                            ccIndex[q.boff] = (int)ccs.size();
                            ccs.push_back(q);
                        }
                // ---- fine records ----
                for (const IBox& u : fineIVS)
                    for (iv[2] = u.lo[2]; iv[2] <= u.hi[2]; ++iv[2])
                        for (iv[1] = u.lo[1]; iv[1] <= u.hi[1]; ++iv[1])
                            for (iv[0] = u.lo[0]; iv[0] <= u.hi[0]; ++iv[0]) {
                                int ivc[3];
                                for (int d = 0; d < 3; ++d) ivc[d] = IBox::fdiv(iv[d], r[d]);
                                const long long boff = cp.off + (ivc[0] - cp.lo[0]) + cst[1] * (ivc[1] - cp.lo[1]) + cst[2] * (ivc[2] - cp.lo[2]);
                                auto itc = ccIndex.find(boff);
                                SOMAR_CHECK(itc != ccIndex.end(), "fine CF ghost cell without a coarse stencil");
                                QFine f;
                                std::memset(&f, 0, sizeof(f));
                                f.foff = fp.off + (iv[0] - fp.lo[0]) + fst[1] * (iv[1] - fp.lo[1]) + fst[2] * (iv[2] - fp.lo[2]);
                                f.stride = (int)(s ? fst[dir] : -fst[dir]);
                                f.cc = itc->second;
                                f.ivf1 = iv[t1]; f.ivf2 = iv[t2];
                                f.ivc1 = ivc[t1]; f.ivc2 = ivc[t2];
                                f.dirflags = dir | (packed ? 4 : 0);
                                fcs.push_back(f);
                            }
            }
        }
    }
    K.ncc = (int)ccs.size();
    K.nfc = (int)fcs.size();
    K.d_cc = to_device(ccs);
    K.d_pts = to_device(pts);
    K.d_fc = to_device(fcs);
    if (K.ncc) SOMAR_HIP(hipMalloc(&K.d_der, (size_t)K.ncc * 5 * sizeof(double)));
}

// ------------------------------------------------------------------------------------
// MappedLevelFluxRegister::define as flat tables
// ------------------------------------------------------------------------------------
void AMRSolver::build_reflux_tables(int l)
{
    AMRLink& K = *links_[l];
    Level& F = S[l]->level(0);
    Level& C = S[l - 1]->level(0);
    const int* r = K.r;
    const int me = comm_->rank;
    long long nc = 0, ncf = 0;
    for (const IBox& b : C.boxes) nc += b.numPts();
    for (const IBox& b : K.cfl->boxes) ncf += b.numPts();
    K.fluxDefined = (nc - ncf) != 0;  // the "temporary flux register optimization", MappedLevelFluxRegister.cpp:97-109
    if (!K.fluxDefined) return;
    const std::vector<IBox>& cf = K.cfl->boxes;
    const auto shifts = periodic_shifts(C.domain, C.periodic);
    const double beta = C.beta;
    K.beta_built = beta;
    for (int d = 0; d < 3; ++d) {
        const double scale = beta / C.dx[d];
        const double denom = (double)(r[0] * r[1] * r[2] / r[d]);
        K.sc_fine[d][0] = -1.0 * scale / denom;
        K.sc_fine[d][1] = 1.0 * scale / denom;
    }
    // ---- register cells of the local fine boxes: (patch, dir, side, slab cell in Fortran order) ----
    std::vector<FRegCell> reg;
    std::vector<std::array<std::array<int, 2>, 3>> regbase(F.npatches());
    for (int pi = 0; pi < F.npatches(); ++pi) {
        const PatchDesc& fp = F.hpatches[pi];
        const IBox fb = F.boxes[F.local[pi]];
        const IBox cb = cf[F.local[pi]];
        const long long st[3] = {1, fp.pj, fp.pk};
        for (int d = 0; d < 3; ++d)
            for (int s = 0; s < 2; ++s) {
                regbase[pi][d][s] = (int)reg.size();
                const IBox slab = adj_cell(cb, d, s);
                int c[3];
                for (c[2] = slab.lo[2]; c[2] <= slab.hi[2]; ++c[2])
                    for (c[1] = slab.lo[1]; c[1] <= slab.hi[1]; ++c[1])
                        for (c[0] = slab.lo[0]; c[0] <= slab.hi[0]; ++c[0]) {
                            int f[3];
                            for (int a = 0; a < 3; ++a) f[a] = c[a] * r[a];
                            f[d] = s == 0 ? fb.lo[d] : fb.hi[d] + 1;
                            FRegCell q;
                            std::memset(&q, 0, sizeof(q));
                            q.cell0 = fp.off + (f[0] - fp.lo[0]) + st[1] * (f[1] - fp.lo[1]) + st[2] * (f[2] - fp.lo[2]);
                            q.patch = pi;
                            q.dir = d;
                            q.side = s;
                            reg.push_back(q);
                        }
            }
    }
    K.nreg_local = (int)reg.size();
    std::vector<int> fpatch_of(F.boxes.size(), -1);
    for (int pi = 0; pi < F.npatches(); ++pi) fpatch_of[F.local[pi]] = pi;
    auto local_reg_index = [&](int fi, int d, int s, const int c[3]) {
        const int pi = fpatch_of[fi];
        const IBox slab = adj_cell(cf[fi], d, s);
        return regbase[pi][d][s] + (c[0] - slab.lo[0]) + slab.size(0) * ((c[1] - slab.lo[1]) + slab.size(1) * (c[2] - slab.lo[2]));
    };
    const BoxIndex cfIndex(cf);
    auto covered = [&](const int c[3]) { return cfIndex.has(c); };
    // ---- per coarse cell records ----
    struct Tmp { int patch; long long coff; std::vector<RefluxA> A; std::vector<std::pair<int, int>> B; };  // B: (peer or -1, index)
    std::vector<Tmp> tmp;
    std::map<std::pair<int, long long>, int> index;
    std::vector<int> cpatch_of(C.boxes.size(), -1);
    for (int pi = 0; pi < C.npatches(); ++pi) cpatch_of[C.local[pi]] = pi;
    auto record = [&](int pi, const int c[3]) -> Tmp& {
        const PatchDesc& p = C.hpatches[pi];
        const long long off = p.off + (c[0] - p.lo[0]) + (long long)p.pj * (c[1] - p.lo[1]) + p.pk * (c[2] - p.lo[2]);
        auto key = std::make_pair(pi, off);
        auto it = index.find(key);
        if (it == index.end()) {
            it = index.insert({key, (int)tmp.size()}).first;
            Tmp t;
            t.patch = pi;
            t.coff = off;
            tmp.push_back(t);
        }
        return tmp[it->second];
    };
    // coarse side (incrementCoarse): local coarse boxes only
    for (int pi = 0; pi < C.npatches(); ++pi) {
        const IBox cb = C.boxes[C.local[pi]];
        const PatchDesc& p = C.hpatches[pi];
        const long long st[3] = {1, p.pj, p.pk};
        for (int d = 0; d < 3; ++d)
            for (int s = 0; s < 2; ++s) {
                const double sc = -(s ? 1.0 : -1.0) * (beta / C.dx[d]);
                for (const auto& sh : shifts)
                    for (const IBox& fb : cf) {
                        const IBox b = adj_cell(fb.shift(sh.data()), d, s) & cb;
                        if (b.empty()) continue;
                        int c[3];
                        for (c[2] = b.lo[2]; c[2] <= b.hi[2]; ++c[2])
                            for (c[1] = b.lo[1]; c[1] <= b.hi[1]; ++c[1])
                                for (c[0] = b.lo[0]; c[0] <= b.hi[0]; ++c[0]) {
                                    if (covered(c)) continue;
                                    Tmp& t = record(pi, c);
                                    RefluxA a;
                                    std::memset(&a, 0, sizeof(a));
                                    // Lo: the cell sees the interface through its HIGH face = low face of c + e_d
                                    a.face = t.coff + (s == 0 ? st[d] : 0);
                                    a.sc = sc;
                                    a.dir = d;
                                    a.sgn = s == 0 ? 1 : -1;
                                    t.A.push_back(a);
                                }
                    }
            }
    }
    // fine side (the reverse copier): every rank walks (coarse box, shift, fine box, cell) in the same order
    std::map<int, std::vector<int>> sendTo;   // peer -> local register indices, in walk order
    std::map<int, int> recvCount;             // peer -> values expected
    for (size_t ci = 0; ci < C.boxes.size(); ++ci) {
        const int co = C.owner[ci];
        const IBox cb = C.boxes[ci];
        for (const auto& sh : shifts)
            for (size_t fi = 0; fi < cf.size(); ++fi) {
                const int fo = F.owner[fi];
                if (co != me && fo != me) continue;
                const int g1[3] = {1, 1, 1};
                const IBox ring = cf[fi].grow(g1).shift(sh.data()) & cb;
                if (ring.empty()) continue;
                int c[3];
                for (c[2] = ring.lo[2]; c[2] <= ring.hi[2]; ++c[2])
                    for (c[1] = ring.lo[1]; c[1] <= ring.hi[1]; ++c[1])
                        for (c[0] = ring.lo[0]; c[0] <= ring.hi[0]; ++c[0]) {
                            int u[3] = {c[0] - sh[0], c[1] - sh[1], c[2] - sh[2]};
                            int nout = 0, d = -1, s = 0;
                            for (int a = 0; a < 3; ++a) {
                                if (u[a] < cf[fi].lo[a]) { ++nout; d = a; s = 0; }
                                else if (u[a] > cf[fi].hi[a]) { ++nout; d = a; s = 1; }
                            }
                            if (nout != 1) continue;   // interior and edge/corner cells of the register hold zeros
                            if (covered(c)) continue;  // refluxed values under the fine level are never used
                            if (co == me) {
                                Tmp& t = record(cpatch_of[ci], c);
                                if (fo == me) t.B.push_back({-1, local_reg_index((int)fi, d, s, u)});
                                else t.B.push_back({fo, recvCount[fo]++});
                            } else {
                                sendTo[co].push_back(local_reg_index((int)fi, d, s, u));
                            }
                        }
            }
    }
    // peers, message layout
    for (auto& kv : sendTo) K.peers.push_back(kv.first);
    for (auto& kv : recvCount) K.peers.push_back(kv.first);
    std::sort(K.peers.begin(), K.peers.end());
    K.peers.erase(std::unique(K.peers.begin(), K.peers.end()), K.peers.end());
    std::vector<int> sendidx;
    std::map<int, long long> recvBase;
    long long rtot = 0;
    for (int q : K.peers) {
        K.soff.push_back((long long)sendidx.size());
        if (sendTo.count(q)) sendidx.insert(sendidx.end(), sendTo[q].begin(), sendTo[q].end());
        K.scount.push_back((long long)sendidx.size() - K.soff.back());
        K.roff.push_back(rtot);
        recvBase[q] = rtot;
        rtot += recvCount.count(q) ? recvCount[q] : 0;
        K.rcount.push_back(rtot - K.roff.back());
    }
    K.nsend = (long long)sendidx.size();
    K.nreg_recv = (int)rtot;
    K.d_sendidx = to_device(sendidx);
    if (K.nsend) SOMAR_HIP(hipMalloc(&K.d_sendbuf, K.nsend * sizeof(double)));
    // flatten
    std::vector<RefluxCell> cells;
    std::vector<RefluxA> A;
    std::vector<int> B;
    for (const Tmp& t : tmp) {
        RefluxCell c;
        std::memset(&c, 0, sizeof(c));
        c.coff = t.coff;
        c.patch = t.patch;
        c.a0 = (int)A.size();
        c.na = (int)t.A.size();
        c.b0 = (int)B.size();
        c.nb = (int)t.B.size();
        for (const RefluxA& a : t.A) A.push_back(a);
        for (const auto& b : t.B) B.push_back(b.first < 0 ? b.second : K.nreg_local + (int)recvBase[b.first] + b.second);
        cells.push_back(c);
    }
    K.nreflux = (int)cells.size();
    K.d_reg = to_device(reg);
    K.d_reflux = to_device(cells);
    K.d_A = to_device(A);
    K.nA = (long long)A.size();
    K.d_B = to_device(B);
    const size_t nvals = (size_t)std::max(1, K.nreg_local + K.nreg_recv);
    SOMAR_HIP(hipMalloc(&K.d_regvals, nvals * sizeof(double)));
    SOMAR_HIP(hipMemset(K.d_regvals, 0, nvals * sizeof(double)));
}

// ------------------------------------------------------------------------------------
// MappedAMRPoissonOp AMR* members
// ------------------------------------------------------------------------------------
void AMRSolver::interp_cf(int l, double* phiFine, const double* phiCoarse, bool ev)
{
    SOMAR_CHECK(l >= 1 && l < nlevels(), "interp_cf: level has no coarser level");
    AMRLink& K = *links_[l];
    if (!K.hasCF) return;
    Level& F = S[l]->level(0);
    double dxc[3];
    for (int d = 0; d < 3; ++d) dxc[d] = F.dx[d] * (double)K.r[d];  // m_dxCrse of MappedQuadCFInterp::define
    // MappedQuadCFInterp copies the coarse data onto the whole coarsened-fine layout and reads its ghost ring; here only the ring
    // travels (the stencils' points, K.d_pts / K.d_cc, all lie outside the coarsened fine boxes)
    static const bool full_gather = getenv("SOMAR_CF_FULL_GATHER") != nullptr;
    (full_gather ? K.gather : K.gather_ring).run(phiCoarse, K.buf, st_);
    launch_cf_slopes(st_, K.d_cc, K.ncc, K.d_pts, K.buf, K.d_der, dxc);
    launch_cf_quad(st_, K.d_fc, K.nfc, K.d_cc, K.d_der, K.buf, phiFine, F.dx, dxc, K.r);
    if (ev) S[l]->cf_ev(0, phiFine);  // ExtrapolateCFEV: non-diagonal metric only (interpCFGhosts, MappedAMRPoissonOp.cpp:2193-2216)
}

// Level projection on level l with the coarser level's data at the coarse-fine interface.
//   LevelMACProjector::computeDiv / LevelCCProjector::computeDiv  (LevelMACProjector.cpp:156-186, LevelCCProjector.cpp:163-197):
//       the cell-centred divergence starts with m_velCFInterp.coarseFineInterp(u, uCrse) (Divergence.cpp:372-375)
//   AMRPressureSolver::solve(lmin = lmax = l): the level solve, phi[l-1] as CF data
//   computeGrad -> levelGradientMAC(edgeGrad, phi, crsePhi, cfInterp) (Gradient.cpp:85-206): coarseFineInterp(phi, crsePhi),
//       exchange, extrapolation BC, MAC gradient [+ EdgeToCell]; applyCorrection
void AMRSolver::level_project(int l, int centring, double dt, bool zeroPressure, bool forceHomogeneous, bool wall,
                              SolveStats& st)
{
    SOMAR_CHECK(finalized_ && l >= 0 && l < nlevels(), "level_project: bad level / hierarchy not finalized");
    SOMAR_CHECK(centring == 0 || centring == 1, "centring: 0 MAC, 1 cell-centred");
    PressureSolver& P = *S[l];
    double* phi = P.field(0, 0);
    double* rhs = P.field(0, 1);
    if (centring == 1) {
        if (l > 0)
            for (int c = 0; c < prm.spaceDim; ++c) interp_cf(l, P.cc_vel(c), S[l - 1]->cc_vel(c), false);
        P.divergence_cc(rhs, dt, wall);
    } else {
        if (wall) P.vel_wall_bc();
        P.divergence_mac(rhs, dt);
    }
    solve(l, l, zeroPressure, forceHomogeneous, st);
    if (l > 0) interp_cf(l, phi, S[l - 1]->field(0, 0));
    if (centring == 1) P.cc_correct(phi, dt);
    else P.mac_correct(phi, dt);
    sync();
}

// ------------------------------------------------------------------------------------
// The COMPOSITE cell-centred projector (sync / initialisation / post-regrid projection)
// ------------------------------------------------------------------------------------
// compGradientCC's one-sided faces (Gradient.cpp:740-833) on level l-1 next to level l, as a face list: per coarse box,
// per coarsened fine box, per direction the loops of the reference, the mask of Mask::buildMask (Mask.cpp:16-59)
// evaluated on the host (MASKCOPY = a cell of a box of this level, unshifted, that no coarsened fine box covers).
// do_lo / do_hi keep every face that is read inside the faces of the box itself, so no exchange of the face gradient is
// needed.  Entries are staged in the reference's order: an entry that reads or rewrites a face an earlier entry wrote
// (fine boxes one or two coarse cells apart) goes to a later stage.
void AMRSolver::build_one_sided_tables(int l)
{
    AMRLink& K = *links_[l];
    if (K.osg_built) return;
    Level& C = S[l - 1]->level(0);
    Level& F = S[l]->level(0);
    std::vector<IBox> cfb;
    for (const IBox& b : F.boxes) cfb.push_back(b.coarsen(K.r));
    const BoxIndex crseIndex(C.boxes), cfbIndex(cfb);
    auto is_copy = [&](const int iv[3]) { return crseIndex.has(iv) && !cfbIndex.has(iv); };
    struct E { OneSided o; int stage; };
    std::vector<E> ents;
    std::map<std::pair<int, long long>, int> wrote;   // (dir, face) -> stage of its last writer
    for (int pi = 0; pi < C.npatches(); ++pi) {
        const IBox box = C.boxes[C.local[pi]];
        const PatchDesc& p = C.hpatches[pi];
        const long long st[3] = {1, (long long)p.pj, p.pk};
        const int one[3] = {1, 1, 1};
        for (const IBox& fb : cfb) {
            const IBox overlap = box & fb.grow(one);
            if (overlap.empty()) continue;
            for (int dir = 0; dir < 3; ++dir) {
                if (!C.active[dir]) continue;
                for (int side = 0; side < 2; ++side) {
                    // the coarse cells just outside the fine box on that side, inside this box
                    IBox adj = fb;
                    if (side == 0) { adj.lo[dir] = fb.lo[dir] - 1; adj.hi[dir] = fb.lo[dir] - 1; }
                    else { adj.lo[dir] = fb.hi[dir] + 1; adj.hi[dir] = fb.hi[dir] + 1; }
                    adj = adj & box;
                    if (adj.empty()) continue;
                    if (side == 0 && overlap.lo[dir] <= box.lo[dir]) continue;   // do_lo = 0
                    if (side == 1 && overlap.hi[dir] >= box.hi[dir]) continue;   // do_hi = 0
                    int iv[3];
                    for (iv[2] = adj.lo[2]; iv[2] <= adj.hi[2]; ++iv[2])
                        for (iv[1] = adj.lo[1]; iv[1] <= adj.hi[1]; ++iv[1])
                            for (iv[0] = adj.lo[0]; iv[0] <= adj.hi[0]; ++iv[0]) {
                                // lo side: the face is the HIGH face of cell iv (index iv + e); the mask is read at the cells
                                // face - 2e = iv - e and face - e = iv.  hi side: the face is the LOW face of iv (index iv); the
                                // mask is read at face + e = iv + e and face = iv.
                                int far[3] = {iv[0], iv[1], iv[2]}, near[3] = {iv[0], iv[1], iv[2]};
                                far[dir] += side == 0 ? -1 : 1;
                                int mode = 0;
                                if (is_copy(far)) mode = 2;
                                else if (is_copy(near)) mode = 1;
                                if (!mode) continue;
                                int fl[3] = {iv[0] - box.lo[0], iv[1] - box.lo[1], iv[2] - box.lo[2]};
                                if (side == 0) fl[dir] += 1;
                                OneSided o;
                                o.face = p.off + fl[0] + st[1] * fl[1] + st[2] * fl[2];
                                o.stride = (int)(side == 0 ? -st[dir] : st[dir]);
                                o.dirmode = dir | (mode << 2);
                                int stage = 0;
                                const long long rd[3] = {o.face, o.face + o.stride, o.face + 2LL * o.stride};
                                for (int q = 0; q < (mode == 2 ? 3 : 2); ++q) {
                                    auto it = wrote.find({dir, rd[q]});
                                    if (it != wrote.end()) stage = std::max(stage, it->second + 1);
                                }
                                wrote[{dir, o.face}] = stage;
                                ents.push_back({o, stage});
                            }
                }
            }
        }
    }
    int nst = 0;
    for (const E& e : ents) nst = std::max(nst, e.stage + 1);
    std::vector<OneSided> flat;
    K.osg_first.clear();
    K.osg_count.clear();
    for (int sgi = 0; sgi < nst; ++sgi) {
        K.osg_first.push_back((int)flat.size());
        for (const E& e : ents)
            if (e.stage == sgi) flat.push_back(e.o);
        K.osg_count.push_back((int)flat.size() - K.osg_first.back());
    }
    K.d_osg = to_device(flat);
    K.osg_built = true;
}

void AMRSolver::comp_divergence_cc(int l, int l_max, double* out, bool wall)
{
    SOMAR_CHECK(finalized_ && l >= 0 && l <= l_max && l_max < nlevels(), "comp_divergence_cc: bad level range");
    PressureSolver& P = *S[l];
    Level& L = P.level(0);
    const int nd = prm.spaceDim;
    for (int c = 0; c < nd; ++c) L.exchange(P.cc_vel(c), st_);   // "Just in case...", AMRCCProjector.cpp:241-243
    if (l > 0)
        for (int c = 0; c < nd; ++c) interp_cf(l, P.cc_vel(c), S[l - 1]->cc_vel(c), false);
    P.divergence_cc(out, 0.0, wall);   // CellToEdge (+ wall BC, in place on the face field) and the level divergence
    if (l == l_max) return;
    // coarse-fine mismatch: reflux the face velocities of level l+1 (Divergence.cpp:770-836)
    AMRLink& K = *links_[l + 1];
    PressureSolver& Q = *S[l + 1];
    Level& F = Q.level(0);
    for (int c = 0; c < nd; ++c) interp_cf(l + 1, Q.cc_vel(c), P.cc_vel(c), false);
    if (!K.fluxDefined) return;
    // the register's scales carry the operator's beta / dx_coarse; the divergence wants 1 / dx_coarse
    SOMAR_CHECK(L.beta == 1.0, "the composite projector needs the pressure operator's beta = 1 (AMRPressureSolver's)");
    sync_reflux_scales(l + 1);
    double* fe[3] = {Q.vel(0), Q.vel(1), Q.vel(2)};
    double* fc[3] = {Q.cc_vel(0), Q.cc_vel(1), nd == 3 ? Q.cc_vel(2) : nullptr};
    launch_cell_to_edge(st_, F.dev, fe, fc, false);   // only the box-side faces are used: cells astride a CF face
    double* ce[3] = {P.vel(0), P.vel(1), P.vel(2)};
    launch_fine_register(st_, K.d_reg, K.nreg_local, F.dev.patches, nullptr, F.dev.jg, F.dx, K.sc_fine, K.r, K.d_regvals, fe);
    if (!K.peers.empty()) {
        launch_gather(st_, K.d_sendidx, K.nsend, K.d_regvals, K.d_sendbuf);
        comm_->neighbor_exchange(K.d_sendbuf, K.d_regvals + K.nreg_local, K.peers, K.soff, K.scount, K.roff, K.rcount, st_);
    }
    launch_reflux(st_, K.d_reflux, K.nreflux, K.d_A, K.d_B, L.dev.patches, nullptr, L.dev.jg, L.dev.jinv, L.dx, K.d_regvals,
                  out, ce);
}

void AMRSolver::comp_grad_correct_cc(int l, int l_max, double* phi, double dt)
{
    SOMAR_CHECK(finalized_ && l >= 0 && l <= l_max && l_max < nlevels(), "comp_grad_correct_cc: bad level range");
    PressureSolver& P = *S[l];
    Level& L = P.level(0);
    const int nd = prm.spaceDim;
    L.exchange(phi, st_);                                         // Copier + CornerCopier, AMRCCProjector.cpp:303-313
    if (l > 0) interp_cf(l, phi, S[l - 1]->field(0, 0));          // levelGradientMAC, Gradient.cpp:104-114
    double* const* g = P.mac_grad(phi);                           // the face gradient, stored
    if (l < l_max) {
        build_one_sided_tables(l + 1);
        AMRLink& K = *links_[l + 1];
        for (size_t q = 0; q < K.osg_first.size(); ++q) launch_one_sided(st_, K.d_osg + K.osg_first[q], K.osg_count[q], g);
    }
    double* c[3] = {P.cc_vel(0), P.cc_vel(1), nd == 3 ? P.cc_vel(2) : nullptr};
    double* gg[3] = {g[0], g[1], g[2]};
    launch_edge_to_cell_axpy(st_, L.dev, c, gg, dt == 0.0 ? -1.0 : -dt);   // EdgeToCell + JVelFAB.plus(corrFAB, dtScale)
}

void AMRSolver::average_down_ccvel(int l)
{
    SOMAR_CHECK(finalized_ && l >= 0 && l + 1 < nlevels(), "average_down_ccvel: level has no finer level");
    AMRLink& K = *links_[l + 1];
    for (int c = 0; c < prm.spaceDim; ++c) {
        launch_avg_unweighted(st_, K.cfl->dev, S[l + 1]->level(0).dev, K.resC, S[l + 1]->cc_vel(c), K.r);
        K.scatter.run(K.resC, S[l]->cc_vel(c), st_);
    }
}

void AMRSolver::cc_project(int l_min, int l_max, double dt, bool zeroPressure, bool forceHomogeneous, bool wall,
                           SolveStats& st)
{
    SOMAR_CHECK(finalized_ && 0 <= l_min && l_min <= l_max && l_max < nlevels(), "cc_project: bad level range");
    for (int l = l_min; l <= l_max; ++l) comp_divergence_cc(l, l_max, S[l]->field(0, 1), wall);
    if (dt != 0.0)
        for (int l = l_min; l <= l_max; ++l) launch_divide(st_, S[l]->field(0, 1), dt, S[l]->level(0).field_elems);
    solve(l_max, l_min, zeroPressure, forceHomogeneous, st);
    // computeGrad on every level first (each level's gradient sees the UNcorrected pressure of its neighbours only), then
    // applyCorrection from the finest level down with the averaging (AMRCCProjector.cpp:334-377); the face gradient of a
    // level does not depend on any velocity, so gradient + correction per level, finest first, is the same arithmetic
    for (int l = l_max; l >= l_min; --l) {
        comp_grad_correct_cc(l, l_max, S[l]->field(0, 0), dt);
        if (l < l_max) average_down_ccvel(l);
    }
    sync();
}

void AMRSolver::amr_operator(int l, double* LofPhi, double* phiFine, double* phi, const double* phiCoarse,
                             bool homogeneous)
{
    if (phiCoarse) interp_cf(l, phi, phiCoarse);
    S[l]->apply_op_i(0, LofPhi, phi, homogeneous);
    if (phiFine) reflux(l, phiFine, phi, LofPhi);
}

void AMRSolver::amr_residual(int l, double* res, double* phiFine, double* phi, const double* phiCoarse,
                             const double* rhs, bool homogeneous)
{
    amr_operator(l, res, phiFine, phi, phiCoarse, homogeneous);
    launch_axby(st_, res, res, rhs, -1.0, 1.0, S[l]->level(0).field_elems);  // axby(res, res, rhs, -1, 1)
}

void AMRSolver::amr_residual_nf(int l, double* res, double* phi, const double* phiCoarse, const double* rhs,
                                bool homogeneous)
{
    if (phiCoarse) interp_cf(l, phi, phiCoarse);
    S[l]->residual_i(0, res, phi, rhs, homogeneous);
}

// MappedAMRPoissonOp::reflux takes its register scale m_beta / m_dx[idir] from the coarse operator when it runs
// (MappedAMRPoissonOp.cpp:1661, 1693), so after setAlphaAndBeta (the heat integrators' resetAlphaAndBeta) the tables built
// at define are rewritten with the same expressions: three numbers on the host, one pass over the coarse-side entries.
void AMRSolver::sync_reflux_scales(int lf)
{
    AMRLink& K = *links_[lf];
    if (!K.fluxDefined) return;
    const Level& C = S[lf - 1]->level(0);
    const double beta = C.beta;
    if (beta == K.beta_built) return;
    double sc[3];
    for (int d = 0; d < 3; ++d) {
        const double scale = beta / C.dx[d];
        const double denom = (double)(K.r[0] * K.r[1] * K.r[2] / K.r[d]);
        K.sc_fine[d][0] = -1.0 * scale / denom;
        K.sc_fine[d][1] = 1.0 * scale / denom;
        sc[d] = scale;
    }
    launch_reflux_rescale(st_, K.d_A, K.nA, sc);
    K.beta_built = beta;
}

void AMRSolver::reflux(int l, double* phiFine, double* phi, double* LofPhi)
{
    SOMAR_CHECK(l >= 0 && l + 1 < nlevels(), "reflux: level has no finer level");
    sync_reflux_scales(l + 1);
    AMRLink& K = *links_[l + 1];
    interp_cf(l + 1, phiFine, phi);
    if (!K.fluxDefined) return;
    Level& F = S[l + 1]->level(0);
    Level& C = S[l]->level(0);
    // non-diagonal metric: getFlux = fillExtrap + MAPPEDGETFLUX.  The register reads the fluxes of its own faces only, so they
    // are evaluated there (reg_flux19) from the level's extrapolated copy instead of filling three whole face fields per level
    // (k_flux_full: 1 ms per 16.7 M-cell level, 8 ms of a C5 cycle).  SOMAR_FLUX_FIELDS=1: the face fields, as before (A/B).
    static const bool fields = getenv("SOMAR_FLUX_FIELDS") != nullptr;
    double* const* flC = (fields && S[l]->is_full()) ? S[l]->flux_fields(phi) : nullptr;
    double* const* flF = (fields && S[l + 1]->is_full()) ? S[l + 1]->flux_fields(phiFine) : nullptr;
    FullFlux ffC, ffF;
    if (!fields && S[l]->is_full()) S[l]->flux_at_faces(phi, ffC);
    if (!fields && S[l + 1]->is_full()) S[l + 1]->flux_at_faces(phiFine, ffF);
    launch_fine_register(st_, K.d_reg, K.nreg_local, F.dev.patches, phiFine, F.dev.jg, F.dx, K.sc_fine, K.r, K.d_regvals,
                         flF, &ffF);
    if (!K.peers.empty()) {
        launch_gather(st_, K.d_sendidx, K.nsend, K.d_regvals, K.d_sendbuf);
        comm_->neighbor_exchange(K.d_sendbuf, K.d_regvals + K.nreg_local, K.peers, K.soff, K.scount, K.roff, K.rcount, st_);
    }
    launch_reflux(st_, K.d_reflux, K.nreflux, K.d_A, K.d_B, C.dev.patches, phi, C.dev.jg, C.dev.jinv, C.dx, K.d_regvals,
                  LofPhi, flC, &ffC);
}

void AMRSolver::amr_restrict(int l, double* residual, double* correction, const double* coarseCorrection,
                             double* scratch)
{
    AMRLink& K = *links_[l];
    if (lean_ && !lepticCycle_) {
        // residual and average in one pass where the marching kernel applies; the scratch (the caller's uberCorrection, rebuilt
        // at the end of the level's branch and not read before) then stays untouched
        if (coarseCorrection) interp_cf(l, correction, coarseCorrection);
        if (S[l]->residual_restrict_i(K.cfl->dev, K.resC, correction, residual, K.r)) return;
        S[l]->residual_i(0, scratch, correction, residual, true);
        launch_restrict(st_, K.cfl->dev, S[l]->level(0).dev, K.resC, scratch, K.r);
        return;
    }
    amr_residual_nf(l, scratch, correction, coarseCorrection, residual);
    launch_restrict(st_, K.cfl->dev, S[l]->level(0).dev, K.resC, scratch, K.r);
}

void AMRSolver::assign_coarse_residual(int l, double* coarseResidual)
{
    AMRLink& K = *links_[l];
    K.scatter.run(K.resC, coarseResidual, st_);
}

void AMRSolver::amr_prolong(int l, double* correction, const double* coarseCorrection)
{
    AMRLink& K = *links_[l];
    K.gather.run(coarseCorrection, K.buf, st_);
    S[l]->prolong_from(K.cfl->dev, K.buf, K.r, correction);
}

void AMRSolver::amr_update_residual(int l, double* residual, double* correction, const double* coarseCorrection)
{
    double* old = S[l]->field(0, 5);
    launch_copy(st_, old, residual, S[l]->level(0).field_elems);
    amr_residual_nf(l, residual, correction, coarseCorrection, old);
}

void AMRSolver::zero_covered(int l, double* f)
{
    AMRLink& K = *links_[l + 1];
    launch_fill_items(st_, S[l]->level(0).dev.patches, K.d_cover, K.ncover, f, 0.0);
}

// ------------------------------------------------------------------------------------
// MappedAMRMultiGrid
// ------------------------------------------------------------------------------------
void AMRSolver::compute_residual_level(double* const* resid, double* const* phi, double* const* rhs, int l_max,
                                       int l_base, int ilev, bool homogeneous)
{
    // homogeneous switches the PHYSICAL boundary values only (Dirichlet sides); the CF values always come from phi
    if (l_max != l_base) {
        if (ilev == l_max) amr_residual_nf(l_max, resid[l_max], phi[l_max], phi[l_max - 1], rhs[l_max], homogeneous);
        else if (ilev == l_base && l_base == 0) amr_residual(0, resid[0], phi[1], phi[0], nullptr, rhs[0], homogeneous);
        else amr_residual(ilev, resid[ilev], phi[ilev + 1], phi[ilev], phi[ilev - 1], rhs[ilev], homogeneous);
    } else {
        if (l_base == 0) S[0]->residual(0, resid[0], phi[0], rhs[0], homogeneous);
        else amr_residual_nf(l_max, resid[l_max], phi[l_max], phi[l_max - 1], rhs[l_max], homogeneous);
    }
}

// computeAMRResidual with a_computeNorm = false (MappedAMRMultiGrid.H:793-836): no zeroCovered, no norm
void AMRSolver::compute_residual_levels_only(double* const* resid, double* const* phi, double* const* rhs, int l_max,
                                             int l_base, bool homogeneous)
{
    for (int ilev = l_base; ilev <= l_max; ++ilev) compute_residual_level(resid, phi, rhs, l_max, l_base, ilev, homogeneous);
}

double AMRSolver::compute_residual(double* const* resid, double* const* phi, double* const* rhs, int l_max, int l_base,
                                   bool homogeneous)
{
    double rnorm = 0.0;
    for (int ilev = l_base; ilev <= l_max; ++ilev) {
        compute_residual_level(resid, phi, rhs, l_max, l_base, ilev, homogeneous);
        if (ilev != l_max) zero_covered(ilev, resid[ilev]);
        rnorm = std::max(S[ilev]->norm(0, resid[ilev], 0), rnorm);
    }
    return rnorm;
}

// MappedAMRMultiGrid::relax, MappedAMRMultiGrid.H:736-766
// corr_zero: corr is to be taken as all zeros whatever it holds (the fused sweep then neither reads nor needs it zeroed)
void AMRSolver::level_relax(int l, double* corr, const double* res, int iters, bool corr_zero)
{
    if (!S[l]->forcedRatios.empty()) {
        if (corr_zero) launch_set(st_, corr, S[l]->level(0).field_elems, 0.0);
        S[l]->prm.num_smooth_down = prm.num_smooth_down;
        S[l]->prm.num_smooth_up = prm.num_smooth_up;
        S[l]->prm.num_smooth_bottom = prm.num_smooth_bottom;
        S[l]->prm.numMG = prm.numMG;
        S[l]->mini_vcycle(corr, res);
    } else {
        S[l]->relax(0, corr, res, iters, corr_zero);
    }
}

void AMRSolver::vcycle(double* const* uberCorr, double* const* uberRes, int ilev, int l_max, int l_base)
{
    const bool lean = lean_ && !lepticCycle_ && l_max != l_base;
    if (ilev == l_max) {
        // m_residual := uberResidual, m_correction := 0 on every level (MappedAMRMultiGrid.H:1504-1510).  Lean: the copies of
        // the levels below the finest are dead stores (computeAMRResidualLevel + assignCopier rewrite m_residual before it
        // is read, the coarser m_correction is zeroed again before its level is entered); the finest level reads the
        // caller's uberResidual in place and starts its first sweep from an implicit zero.
        for (int l = l_base; l <= l_max; ++l) {
            rcur_[l] = res_[l];
            visits_[l] = 0;
            if (lean) continue;
            const long long n = S[l]->level(0).field_elems;
            launch_copy(st_, res_[l], uberRes[l], n);
            launch_set(st_, corr_[l], n, 0.0);
        }
        if (lean) rcur_[l_max] = uberRes[l_max];
    }
    const long long n = S[ilev]->level(0).field_elems;
    if (lepticCycle_) {
        // AMRLepticSolver::AMRVCycle, AMRLepticSolver.cpp:430-529
        auto lsolve = [&](int l, double* phi, const double* rhs) { leptic_[l]->solve_fields(phi, rhs, lepStats_[l]); };
        if (l_max == l_base) {
            lsolve(l_base, uberCorr[ilev], uberRes[ilev]);
        } else if (ilev == l_base) {
            if (lepticBaseFromRestricted_) lsolve(l_base, corr_[ilev], res_[ilev]);
            else lsolve(l_base, uberCorr[ilev], uberRes[ilev]);   // as written, :444-449
            launch_incr(st_, uberCorr[ilev], corr_[ilev], 1.0, n);
        } else {
            lsolve(ilev, corr_[ilev], res_[ilev]);
            launch_incr(st_, uberCorr[ilev], corr_[ilev], 1.0, n);
            launch_set(st_, corr_[ilev - 1], S[ilev - 1]->level(0).field_elems, 0.0);
            compute_residual_level(res_.data(), uberCorr, uberRes, l_max, l_base, ilev - 1, true);
            amr_restrict(ilev, res_[ilev], corr_[ilev], corr_[ilev - 1], uberCorr[ilev]);
            assign_coarse_residual(ilev, res_[ilev - 1]);
            for (int img = 0; img < prm.numMG; ++img) vcycle(uberCorr, uberRes, ilev - 1, l_max, l_base);
            amr_prolong(ilev, corr_[ilev], corr_[ilev - 1]);
            amr_update_residual(ilev, res_[ilev], corr_[ilev], corr_[ilev - 1]);
            double* dCorr = uberCorr[ilev];
            launch_set(st_, dCorr, n, 0.0);
            lsolve(ilev, dCorr, res_[ilev]);
            launch_incr(st_, corr_[ilev], dCorr, 1.0, n);
            launch_copy(st_, uberCorr[ilev], corr_[ilev], n);
        }
        return;
    }
    if (l_max == l_base) {
        S[l_base]->vcycle(uberCorr[ilev], uberRes[ilev]);
    } else if (ilev == l_base) {
        S[l_base]->vcycle(corr_[ilev], res_[ilev], true);  // m_correction was set to zero above
        launch_incr(st_, uberCorr[ilev], corr_[ilev], 1.0, n);
    } else {
        if (lean) {
            // the same operations on the same values, minus the whole-field passes that only move or clear data:
            //  - m_correction[ilev] is all zeros on the level's first visit in this cycle: the first sweep starts from an
            //    implicit zero (no memset, no read);
            //  - restriction: residual + average in one marching pass (amr_restrict);
            //  - AMRUpdateResidual writes the new residual into the level's other buffer instead of copying the old one;
            //  - dCorr (= uberCorrection[ilev]) likewise starts as an implicit zero;
            //  - m_correction += dCorr and uberCorrection := m_correction in one pass.
            double* R = rcur_[ilev];
            level_relax(ilev, corr_[ilev], R, prm.num_smooth_down, visits_[ilev] == 0);
            ++visits_[ilev];
            launch_incr(st_, uberCorr[ilev], corr_[ilev], 1.0, n);
            launch_set(st_, corr_[ilev - 1], S[ilev - 1]->level(0).field_elems, 0.0);
            rcur_[ilev - 1] = res_[ilev - 1];
            compute_residual_level(res_.data(), uberCorr, uberRes, l_max, l_base, ilev - 1, true);
            amr_restrict(ilev, R, corr_[ilev], corr_[ilev - 1], uberCorr[ilev]);
            assign_coarse_residual(ilev, res_[ilev - 1]);
            for (int img = 0; img < prm.numMG; ++img) vcycle(uberCorr, uberRes, ilev - 1, l_max, l_base);
            amr_prolong(ilev, corr_[ilev], corr_[ilev - 1]);
            double* Rn = (R == res_[ilev]) ? spare_[ilev] : res_[ilev];
            SOMAR_CHECK(Rn, "internal: no spare residual buffer on this level");
            amr_residual_nf(ilev, Rn, corr_[ilev], corr_[ilev - 1], R);   // AMRUpdateResidual
            rcur_[ilev] = Rn;
            double* dCorr = uberCorr[ilev];
            level_relax(ilev, dCorr, Rn, prm.num_smooth_up, true);
            launch_incr_copy(st_, corr_[ilev], dCorr, 1.0, n);
            return;
        }
        level_relax(ilev, corr_[ilev], res_[ilev], prm.num_smooth_down);
        launch_incr(st_, uberCorr[ilev], corr_[ilev], 1.0, n);
        launch_set(st_, corr_[ilev - 1], S[ilev - 1]->level(0).field_elems, 0.0);
        compute_residual_level(res_.data(), uberCorr, uberRes, l_max, l_base, ilev - 1, true);
        // the scratch of AMRRestrictS IS uberCorrection[ilev] (MappedAMRMultiGrid.H:1548-1552)
        amr_restrict(ilev, res_[ilev], corr_[ilev], corr_[ilev - 1], uberCorr[ilev]);
        assign_coarse_residual(ilev, res_[ilev - 1]);
        for (int img = 0; img < prm.numMG; ++img) vcycle(uberCorr, uberRes, ilev - 1, l_max, l_base);
        amr_prolong(ilev, corr_[ilev], corr_[ilev - 1]);
        amr_update_residual(ilev, res_[ilev], corr_[ilev], corr_[ilev - 1]);
        double* dCorr = uberCorr[ilev];
        launch_set(st_, dCorr, n, 0.0);
        level_relax(ilev, dCorr, res_[ilev], prm.num_smooth_up);
        launch_incr(st_, corr_[ilev], dCorr, 1.0, n);
        launch_copy(st_, uberCorr[ilev], corr_[ilev], n);
    }
}

void AMRSolver::set_alpha_beta(double a, double b)
{
    SOMAR_CHECK(finalized_, "set_alpha_beta before finalize");
    for (auto& s : S) s->set_alpha_beta(a, b, true);
}

void AMRSolver::heat_step(int l, int scheme, double dt, bool zeroPhi, double oldTime, double crseOldTime, double crseNewTime,
                          SolveStats& st)
{
    SOMAR_CHECK(finalized_ && l >= 0 && l < nlevels(), "bad level / hierarchy not finalized");
    SOMAR_CHECK(scheme >= 0 && scheme <= 2, "heat scheme: 0 backward Euler, 1 Crank-Nicolson, 2 TGA");
    SOMAR_CHECK(dt >= 0.0 && crseNewTime >= crseOldTime, "negative time step");
    PressureSolver& P = *S[l];
    const long long n = P.level(0).field_elems;
    double* phiNew = P.phi();
    double* rhst = P.rhs();
    double* phiOld = P.heat_field(0);
    double* src = P.heat_field(1);
    double* tmp = P.heat_field(2);
    double* scratch = P.field(0, 5);
    double* coarse = l > 0 ? S[l - 1]->heat_field(2) : nullptr;
    // timeInterp, MappedBaseLevelHeatSolver.cpp:273-300
    auto coarse_at = [&](double t) {
        if (l == 0) return;
        const long long nc = S[l - 1]->level(0).field_elems;
        const double* cOld = S[l - 1]->heat_field(0);
        const double* cNew = S[l - 1]->phi();
        launch_set(st_, coarse, nc, 0.0);
        const double diff = crseNewTime - crseOldTime;
        if (diff < 1.0e-10) {
            launch_incr(st_, coarse, cOld, 1.0, nc);
        } else {
            const double factor = (t - crseOldTime) / (crseNewTime - crseOldTime);
            launch_incr(st_, coarse, cOld, 1.0 - factor, nc);
            launch_incr(st_, coarse, cNew, factor, nc);
        }
    };
    // applyHelm: (I + mu dt L) phi, through AMROperatorNF when there is coarse data (:154-180)
    auto apply_helm = [&](double* ans, double* phi, bool withCoarse, double mu, bool homogeneous) {
        set_alpha_beta(1.0, mu * dt);
        if (!withCoarse || l == 0) P.apply_op(0, ans, phi, homogeneous);
        else amr_operator(l, ans, nullptr, phi, coarse, homogeneous);
    };
    // solveHelm: m_solver->solve(phi, rhs, l, l, zeroPhi) with alpha 1, beta -dt mu (:217-251)
    auto solve_helm = [&](double mu) {
        set_alpha_beta(1.0, -dt * mu);
        crse_override_ = coarse;
        try {
            solve(l, l, zeroPhi, false, st);
        } catch (...) {
            crse_override_ = nullptr;
            throw;
        }
        crse_override_ = nullptr;
    };
    if (scheme == 0) {
        launch_copy(st_, rhst, phiOld, n);
        coarse_at(oldTime);   // as written: a_oldTime (MappedLevelBackwardEuler.cpp:112-113)
        solve_helm(1.0);
        P.increment_heat_flux(phiNew, true);
    } else if (scheme == 1) {
        coarse_at(oldTime);
        apply_helm(scratch, phiOld, true, 0.5, false);
        launch_copy(st_, rhst, src, n);
        launch_scale(st_, rhst, dt, n);
        launch_incr(st_, rhst, scratch, 1.0, n);
        solve_helm(0.5);
        P.increment_heat_flux(phiNew, true);
    } else {
        const double tgaEpsilon = 1.e-12;
        const double a = 2.0 - std::sqrt(2.0) - tgaEpsilon;
        const double discr = std::sqrt(a * a - 4.0 * a + 2.0);
        const double mu1 = (a - discr) / 2.0, mu2 = (a + discr) / 2.0, mu3 = 1.0 - a, mu4 = 0.5 - a;
        const double r1 = (2.0 * a - 1.0) / (a + discr);
        double* phis = S[l]->field(0, 4);        // bestPhi's array: free between solves (each solve rewrites it first)
        launch_copy(st_, tmp, src, n);           // srct = dt * src
        launch_scale(st_, tmp, dt, n);
        if (!zeroPhi) launch_copy(st_, phis, phiNew, n);
        apply_helm(rhst, tmp, false, mu4, true);
        P.increment_heat_flux(tmp, true);
        coarse_at(oldTime);
        apply_helm(scratch, phiOld, true, mu3, false);
        P.increment_heat_flux(phiOld, false);
        launch_incr(st_, rhst, scratch, 1.0, n);
        coarse_at(oldTime + (1.0 - r1) * dt);
        double* keep = nullptr;
        if (!zeroPhi) {
            // the guess has to survive the first solve, whose bestPhi bookkeeping overwrites `phis`: park it in srct's array
            keep = tmp;
            launch_copy(st_, keep, phis, n);
            launch_copy(st_, phiNew, keep, n);
        }
        solve_helm(mu2);
        P.increment_heat_flux(phiNew, false);
        launch_copy(st_, rhst, phiNew, n);       // assign(rhst, phiNew)
        coarse_at(oldTime + dt);
        if (!zeroPhi) launch_copy(st_, phiNew, keep, n);
        solve_helm(mu1);
        P.increment_heat_flux(phiNew, false);
    }
    sync();
}

// MappedAMRTGA<T>::oneStep (AMRElliptic/MappedAMRTGA.H:417-497): the COMPOSITE TGA step over levels l_base..l_max at once --
// applyHelm = resetAlphaAndBeta(1, mu dt) + MappedAMRMultiGrid::computeAMROperator (:499-523; MappedAMRMultiGrid.H:862-878:
// the composite residual of a zero right-hand side, negated; covered cells are NOT zeroed because a_computeNorm is false),
// solveHelm = resetAlphaAndBeta(1, -mu dt) + solveNoInit(ans, rhs, l_max, l_base, zeroPhi = false) (:525-546).
// divideByIdentityCoef and diagonalScale are no-ops of MappedAMRPoissonOp (MappedAMRPoissonOp.cpp:814-825).
// Per level: phiNew = phi(), rhst = rhs(), phiOld = heat_field(0), source = heat_field(1), srct = heat_field(2).
void AMRSolver::tga_step(int l_max, int l_base, double dt, SolveStats& st)
{
    SOMAR_CHECK(finalized_ && l_base >= 0 && l_base <= l_max && l_max < nlevels(), "tga_step: bad level range");
    // createData allocates m_srct for l_base..l_max only (MappedAMRTGA.H:388-403); computeAMROperator on m_srct then reads
    // *m_srct[l_base - 1] for the coarse-fine values of level l_base (MappedAMRMultiGrid.H:907-909) -- a null pointer when
    // l_base > 0.  The reference has no defined behaviour there, so neither has this.
    SOMAR_CHECK(l_base == 0, "tga_step: l_base > 0 is undefined in the reference (MappedAMRTGA::oneStep reads the source "
                             "term of level l_base - 1, which createData never allocates)");
    SOMAR_CHECK(dt >= 0.0, "negative time step");
    const double tgaEpsilon = 1.e-12;
    const double a = 2.0 - std::sqrt(2.0) - tgaEpsilon;
    const double discr = std::sqrt(a * a - 4.0 * a + 2.0);
    const double mu1 = (a - discr) / 2.0, mu2 = (a + discr) / 2.0, mu3 = 1.0 - a, mu4 = 0.5 - a;
    const int nl = nlevels();
    std::vector<double*> phiNew(nl), rhst(nl), phiOld(nl), src(nl), srct(nl), zero(nl);
    std::vector<long long> n(nl);
    for (int l = 0; l < nl; ++l) {
        PressureSolver& P = *S[l];
        n[l] = P.level(0).field_elems;
        phiNew[l] = P.phi();
        rhst[l] = P.rhs();
        phiOld[l] = P.heat_field(0);
        src[l] = P.heat_field(1);
        srct[l] = P.heat_field(2);
        zero[l] = P.field(0, 5);   // m_residual, set to zero: computeAMROperator's right-hand side
    }
    auto apply_helm = [&](std::vector<double*>& ans, std::vector<double*>& phi, double mu, bool homogeneous) {
        set_alpha_beta(1.0, mu * dt);
        for (int l = l_base; l <= l_max; ++l) launch_set(st_, zero[l], n[l], 0.0);
        compute_residual_levels_only(ans.data(), phi.data(), zero.data(), l_max, l_base, homogeneous);
        for (int l = l_base; l <= l_max; ++l) launch_scale(st_, ans[l], -1.0, n[l]);
    };
    auto solve_helm = [&](double mu) {
        set_alpha_beta(1.0, -mu * dt);
        solve(l_max, l_base, false, false, st);
    };
    for (int l = l_base; l <= l_max; ++l) {
        launch_set(st_, srct[l], n[l], 0.0);
        launch_incr(st_, srct[l], src[l], 1.0, n[l]);
    }
    apply_helm(rhst, srct, mu4, true);
    for (int l = l_base; l <= l_max; ++l) launch_scale(st_, rhst[l], dt, n[l]);
    apply_helm(phiNew, phiOld, mu3, false);
    for (int l = l_base; l <= l_max; ++l) launch_incr(st_, rhst[l], phiNew[l], 1.0, n[l]);
    for (int l = l_base; l <= l_max; ++l) launch_copy(st_, phiNew[l], phiOld[l], n[l]);
    solve_helm(mu2);
    for (int l = l_base; l <= l_max; ++l) launch_copy(st_, rhst[l], phiNew[l], n[l]);
    for (int l = l_base; l <= l_max; ++l) launch_copy(st_, phiNew[l], phiOld[l], n[l]);
    solve_helm(mu1);
    sync();
}

void AMRSolver::enable_leptic(const LepticParams& lp, bool baseFromRestricted)
{
    SOMAR_CHECK(finalized_, "enable_leptic before finalize");
    lepticBaseFromRestricted_ = baseFromRestricted;
    leptic_.clear();
    lepStats_.assign(nlevels(), LepticStats());
    for (int l = 0; l < nlevels(); ++l) {
        leptic_.emplace_back(new LepticSolver(comm_, st_));
        leptic_.back()->attach(S[l].get(), lp);
    }
}

void AMRSolver::solve(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& s)
{
    lepticCycle_ = false;
    solve_impl(l_max, l_base, zeroPhi, forceHomogeneous, s);
}

void AMRSolver::solve_leptic(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& s)
{
    SOMAR_CHECK((int)leptic_.size() == nlevels(), "solve_leptic before enable_leptic");
    lepticCycle_ = true;
    try {
        solve_impl(l_max, l_base, zeroPhi, forceHomogeneous, s);
    } catch (...) {
        lepticCycle_ = false;
        throw;
    }
    lepticCycle_ = false;
}

void AMRSolver::solve_impl(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& s)
{
    SOMAR_CHECK(finalized_, "solve before finalize");
    SOMAR_CHECK(0 <= l_base && l_base <= l_max && l_max < nlevels(), "bad level range");
    const int n = nlevels();
    std::vector<double*> phi(n), rhs(n), uRes(n), uCorr(n), best(n);
    for (int l = 0; l < n; ++l) {
        phi[l] = (crse_override_ && l == l_base - 1) ? crse_override_ : S[l]->phi();
        rhs[l] = S[l]->rhs();
        uRes[l] = S[l]->work(0);
        uCorr[l] = S[l]->work(1);
        best[l] = S[l]->work(2);
    }
    const int lowlim = l_base > 0 ? l_base - 1 : l_base;
    for (int l = 0; l < n; ++l) {
        S[l]->prm.num_smooth_down = prm.num_smooth_down;
        S[l]->prm.num_smooth_up = prm.num_smooth_up;
        S[l]->prm.num_smooth_bottom = prm.num_smooth_bottom;
        S[l]->prm.numMG = prm.numMG;
    }
    for (int l = lowlim; l <= l_max; ++l) {
        const long long ne = S[l]->level(0).field_elems;
        launch_set(st_, uCorr[l], ne, 0.0);
        if (l >= l_base) launch_set(st_, uRes[l], ne, 0.0);
    }
    if (zeroPhi)
        for (int l = l_base; l <= l_max; ++l) launch_set(st_, phi[l], S[l]->level(0).field_elems, 0.0);
    for (int l = lowlim; l <= l_max; ++l) launch_copy(st_, best[l], phi[l], S[l]->level(0).field_elems);
    double initial_rnorm = compute_residual(uRes.data(), phi.data(), rhs.data(), l_max, l_base, forceHomogeneous);
    double rnorm = initial_rnorm, norm_last = 2 * initial_rnorm, best_rnorm = rnorm;
    bool useBestPhi = false, somethingConverged = false;
    S[l_base]->bottom_metric = initial_rnorm;  // setConvergenceMetrics(initial_rnorm, cushion * eps)
    S[l_base]->bottom_eps_eff = 1.0 * prm.eps;
    int iter = 0;
    s = SolveStats();
    s.history.push_back(rnorm);
    bool goNorm = rnorm > prm.normThresh;
    bool goRedu = rnorm > prm.eps * initial_rnorm;
    bool goIter = iter < prm.imax;
    bool goHang = iter < prm.imin || rnorm < (1 - prm.hang) * norm_last;
    while (goIter && goRedu && goHang && goNorm) {
        if (inspector_) { sync(); inspector_(inspector_user_, 0, iter, l_base, l_max); }   // recordResiduals, :1064-1065
        norm_last = rnorm;
        vcycle(uCorr.data(), uRes.data(), l_max, l_max, l_base);
        if (inspector_) { sync(); inspector_(inspector_user_, 1, iter, l_base, l_max); }   // recordCorrections, :1083-1084
        for (int l = l_base; l <= l_max; ++l) {  // postVCycleOps
            const long long ne = S[l]->level(0).field_elems;
            launch_incr(st_, phi[l], uCorr[l], 1.0, ne);
            launch_set(st_, uCorr[l], ne, 0.0);
        }
        rnorm = compute_residual(uRes.data(), phi.data(), rhs.data(), l_max, l_base, forceHomogeneous);
        ++iter;
        s.history.push_back(rnorm);
        if (rnorm <= best_rnorm) {
            best_rnorm = rnorm;
            for (int l = l_base; l <= l_max; ++l) launch_copy(st_, best[l], phi[l], S[l]->level(0).field_elems);
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
        for (int l = l_base; l <= l_max; ++l) launch_copy(st_, phi[l], best[l], S[l]->level(0).field_elems);
    }
    s.status = 0;
    if (rnorm > 10. * initial_rnorm && rnorm > 10. * prm.eps) s.status = 1;
    else if (!lepticCycle_ && !somethingConverged && rnorm >= initial_rnorm && rnorm >= prm.eps)
        s.status = 2;   // AMRLepticSolver only prints here (AMRLepticSolver.cpp:384-388)
    s.exitStatus = int(!goRedu) + int(!goIter) * 2 + int(!goHang) * 4 + int(!goNorm) * 8;
    s.iters = iter;
    s.initial_rnorm = initial_rnorm;
    s.final_rnorm = rnorm;
    s.bottom_iters_last = S[l_base]->bottom_iters;
    s.bottom_exit_last = S[l_base]->bottom_exit;
    sync();
}

}  // namespace somar
