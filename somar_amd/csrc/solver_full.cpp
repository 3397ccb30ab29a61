// somar_amd/csrc/solver_full.cpp -- host side of the non-diagonal (19-point) path of PressureSolver:
// metric storage with all J g^{ab} components, and the compilation of the reference's ghost-filling call
// sequences into flat per-level op lists ("programs") that full19.hip executes stage by stage.
//
//   fillExtrap                               calculus/AMRElliptic/MappedAMRPoissonOp.cpp:2244-2270
//   RelaxationMethod::fillGhostsAndExtrapolate  calculus/AMRElliptic/RelaxationMethods/RelaxationMethod.cpp:376-435
//   ExtrapolateFaceAndCopy / ExtrapolateFaceNoEV  calculus/extrapolation/ExtrapolationUtils.cpp:34-67, 109-155
//   EllipticConstNeumBCGhostClass, setSideNeumBC   calculus/BCInterface/EllipticBCUtils.cpp:128-214, 431-482
//
// The reference runs these per box on FABs with ONE ghost layer; regions are therefore clipped to the box grown by
// one cell although the device frame is two deep.  The ops of one box must run in order; the s-th ops of all boxes
// form stage s and go up in one launch.
#include <algorithm>
#include <cstring>

#include "solver.h"

namespace somar {

namespace {

IBox adj_cell(const IBox& b, int d, int side)
{
    IBox g = b;
    if (side == 0) { g.lo[d] = b.lo[d] - 1; g.hi[d] = b.lo[d] - 1; }
    else { g.lo[d] = b.hi[d] + 1; g.hi[d] = b.hi[d] + 1; }
    return g;
}
IBox grow_dir(const IBox& b, int d, int n)
{
    IBox g = b;
    g.lo[d] -= n;
    g.hi[d] += n;
    return g;
}

struct Builder {
    const IBox valid;   // the box
    const IBox fab;     // box grown by one cell in the active directions: the reference's FAB
    int patch;
    std::vector<GhostOp>& out;

    void push(int type, const IBox& region, int dir, int sgn, int order, int dstf, int srcf)
    {
        if (region.empty()) return;
        GhostOp op;
        std::memset(&op, 0, sizeof(op));
        op.patch = patch;
        op.type = type;
        for (int d = 0; d < 3; ++d) { op.lo[d] = region.lo[d] - valid.lo[d]; op.n[d] = region.size(d); }
        op.dir = dir;
        op.sgn = sgn;
        op.order = order;
        op.dstf = dstf;
        op.srcf = srcf;
        out.push_back(op);
    }
    // ExtrapolateFaceNoEV (cell-centred): fills adjCell(v, d, side) & srcBox
    void face_no_ev(int dstf, int srcf, const IBox& v, int d, int side, int order)
    {
        if (v.empty()) return;
        push(GHOST_EXTRAP, adj_cell(v, d, side) & fab, d, side ? 1 : -1, order, dstf, srcf);
    }
    // ExtrapolateFaceAndCopy(dest, src, valid v, d, side, order, numLayers = 1)
    void face_and_copy(int dstf, int srcf, const IBox& v, int d, int side, int order, const int active[3])
    {
        if (v.empty()) return;
        face_no_ev(dstf, srcf, v, d, side, order);
        IBox ghostBox = adj_cell(v, d, side) & fab;
        IBox nearBox = ghostBox;
        nearBox.lo[d] += side ? -1 : 1;
        nearBox.hi[d] += side ? -1 : 1;
        if (ghostBox.empty()) return;
        if (dstf != srcf) push(GHOST_COPY, nearBox, 0, 0, 0, dstf, srcf);
        for (int e = 0; e < 3; ++e) {
            if (e == d || !active[e]) continue;
            for (int es = 0; es < 2; ++es) {
                face_no_ev(dstf, dstf, ghostBox, e, es, order);
                face_no_ev(dstf, dstf, nearBox, e, es, order);
            }
            ghostBox = grow_dir(ghostBox, e, 1) & fab;
            nearBox = grow_dir(nearBox, e, 1) & fab;
        }
    }
};

}  // namespace

// Programs of one level.  which = 0: operator (fillExtrap order 2, then the Neumann ghosts of phi);
// which = 1: smoother (extrapolation order 1 from the domain box, then the Neumann ghosts of phi).
// The leading full copy psi := phi is done by the caller with one flat copy.
static std::vector<std::vector<GhostOp>> build_program(const Level& L, int which)
{
    std::vector<std::vector<GhostOp>> perPatch(L.npatches());
    int g1[3];
    for (int d = 0; d < 3; ++d) g1[d] = L.active[d] ? 1 : 0;
    int gper[3];
    for (int d = 0; d < 3; ++d) gper[d] = (L.periodic[d] && L.active[d]) ? 1 : 0;
    const IBox validDomain = L.domain.grow(gper);  // m_validDomain, MappedAMRPoissonOp.cpp:278-283
    for (int pi = 0; pi < L.npatches(); ++pi) {
        const IBox valid = L.boxes[L.local[pi]];
        Builder B{valid, valid.grow(g1), pi, perPatch[pi]};
        if (which == 0) {
            IBox validPhi = B.fab & validDomain;
            for (int fdir = 0; fdir < 3; ++fdir) {
                if (!L.active[fdir]) continue;
                B.face_and_copy(1, 1, validPhi, fdir, 0, 2, L.active);
                B.face_and_copy(1, 1, validPhi, fdir, 1, 2, L.active);
                validPhi = grow_dir(validPhi, fdir, 1) & B.fab;
            }
        } else {
            IBox domValid = L.domain & B.fab;
            for (int fdir = 0; fdir < 3; ++fdir) {
                if (!L.active[fdir]) continue;
                B.face_and_copy(1, 1, domValid, fdir, 0, 1, L.active);
                B.face_and_copy(1, 1, domValid, fdir, 1, 1, L.active);
                domValid = grow_dir(domValid, fdir, 1);
            }
        }
        // bc_set_ghosts: for every non-periodic direction and side with a Neumann BC, where the box touches the domain
        for (int d = 0; d < 3; ++d) {
            if (!L.active[d] || L.periodic[d]) continue;
            for (int side = 0; side < 2; ++side) {
                if (L.bc_type[d][side] != BC_NEUM) continue;
                const int vend = side ? valid.hi[d] : valid.lo[d];
                const int dend = side ? L.domain.hi[d] : L.domain.lo[d];
                if (vend != dend) continue;
                const IBox ghostBox = adj_cell(valid, d, side) & B.fab;
                if (ghostBox.empty()) continue;
                B.face_and_copy(1, 0, valid, d, side, 2, L.active);  // ex <- extrapolation of phi, order 2
                B.push(GHOST_NEUM, ghostBox, d, side ? 1 : -1, 0, 0, 1);
            }
        }
    }
    // transpose into stages
    size_t ns = 0;
    for (auto& v : perPatch) ns = std::max(ns, v.size());
    std::vector<std::vector<GhostOp>> stages(ns);
    for (auto& v : perPatch)
        for (size_t s = 0; s < v.size(); ++s) stages[s].push_back(v[s]);
    return stages;
}

void PressureSolver::build_full_programs(int d)
{
    Level& L = *lev[d];
    for (int which = 0; which < 2; ++which) {
        FullProgram& P = full_prog_[d][which];
        const auto stages = build_program(L, which);
        std::vector<GhostOp> flat;
        P.first.clear();
        P.count.clear();
        for (const auto& s : stages) {
            P.first.push_back((int)flat.size());
            P.count.push_back((int)s.size());
            flat.insert(flat.end(), s.begin(), s.end());
        }
        hipFree(P.d_ops);
        P.d_ops = nullptr;
        if (!flat.empty()) {
            SOMAR_HIP(hipMalloc(&P.d_ops, flat.size() * sizeof(GhostOp)));
            SOMAR_HIP(hipMemcpy(P.d_ops, flat.data(), flat.size() * sizeof(GhostOp), hipMemcpyHostToDevice));
        }
    }
    SOMAR_HIP(hipDeviceSynchronize());
}

void PressureSolver::run_full_program(int d, int which, double* phi)
{
    Level& L = *lev[d];
    launch_copy(st_, f_psi[d], phi, L.field_elems);  // psi := phi (valid cells and exchanged ghosts)
    const FullProgram& P = full_prog_[d][which];
    for (size_t s = 0; s < P.first.size(); ++s)
        launch_ghost_ops(st_, L.dev, P.d_ops + P.first[s], P.count[s], phi, f_psi[d]);
}

void PressureSolver::set_metric_full(int patch, const double* jg0, const double* jg1, const double* jg2,
                                     const double* jinv)
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_metric before define / after finalize");
    SOMAR_CHECK(!hasCF_, "the non-diagonal metric path is implemented for one AMR level");
    Level& L = *lev[0];
    SOMAR_CHECK(patch >= 0 && patch < L.npatches(), "bad patch index");
    full_ = true;
    alloc_full_metric(L);  // SpaceDim 2: the planes with a z index stay zero, the 3-D operator / ghost kernels see 0 * finite
    const IBox valid = L.boxes[L.local[patch]];
    const double* jg[3] = {jg0, jg1, jg2};
    const int nd = prm.spaceDim;
    for (int d = 0; d < nd; ++d) {
        SOMAR_CHECK(jg[d] != nullptr, "null metric array");
        IBox fb = valid;
        fb.hi[d] += 1;
        const long long plane = fb.numPts();
        for (int c = 0; c < nd; ++c) L.upload(L.dev.jgf[d][c], patch, jg[d] + c * plane, fb, fb, st_);  // comp slowest
    }
    L.upload(L.dev.jinv, patch, jinv, valid, valid, st_);
    sync();
}

void PressureSolver::alloc_full_metric(Level& L)
{
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) {
            if (c == d) L.dev.jgf[d][c] = L.dev.jg[d];
            else if (!L.dev.jgf[d][c]) L.dev.jgf[d][c] = L.alloc_field();
        }
}

}  // namespace somar
