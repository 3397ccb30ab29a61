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

// The ops of one box must take effect in the order the reference issues them -- but only where they touch the same
// cells.  Each op gets the earliest stage after every earlier op of its box it conflicts with (read-after-write,
// write-after-read, write-after-write on the same field); ops of one stage then go up in ONE launch.  The reference's
// ~130 sequential calls per box and application collapse to about a dozen launches.
static std::vector<std::vector<GhostOp>> schedule_stages(const std::vector<std::vector<GhostOp>>& perPatch)
{
    struct Acc { IBox box; int field; };
    auto reads_of = [](const GhostOp& op, std::vector<Acc>& rd) {
        IBox r;
        for (int d = 0; d < 3; ++d) { r.lo[d] = op.lo[d]; r.hi[d] = op.lo[d] + op.n[d] - 1; }
        if (op.type == GHOST_COPY) rd.push_back({r, op.srcf});
        else if (op.type == GHOST_EXTRAP || op.type == GHOST_DIRI) {
            IBox b = r;   // 1 .. 3 steps back along dir
            const int far = op.type == GHOST_DIRI ? 1 : (op.order == 2 ? 3 : (op.order == 1 ? 2 : 1));
            if (op.sgn > 0) { b.lo[op.dir] = r.lo[op.dir] - far; b.hi[op.dir] = r.hi[op.dir] - 1; }
            else { b.lo[op.dir] = r.lo[op.dir] + 1; b.hi[op.dir] = r.hi[op.dir] + far; }
            rd.push_back({b, op.srcf});
        } else {  // GHOST_NEUM: psi on the ghost layer and the first valid layer at +-1 along each tangential direction (no diagonal:
                  // ghost_op_body), phi on the valid layer
            IBox b = r;
            if (op.sgn > 0) b.lo[op.dir] -= 1; else b.hi[op.dir] += 1;
            for (int t = 1; t <= 2; ++t) {
                const int d = (op.dir + t) % 3;
                for (int sg = -1; sg <= 1; sg += 2) {
                    IBox q = b;
                    q.lo[d] += sg;
                    q.hi[d] += sg;
                    rd.push_back({q, 1});
                }
            }
            rd.push_back({b, 0});
        }
    };
    size_t ns = 0;
    std::vector<std::vector<int>> stageOf(perPatch.size());
    for (size_t pi = 0; pi < perPatch.size(); ++pi) {
        const auto& ops = perPatch[pi];
        std::vector<std::vector<Acc>> rds(ops.size());
        std::vector<Acc> wrs(ops.size());
        stageOf[pi].assign(ops.size(), 0);
        for (size_t q = 0; q < ops.size(); ++q) {
            const GhostOp& op = ops[q];
            IBox w;
            for (int d = 0; d < 3; ++d) { w.lo[d] = op.lo[d]; w.hi[d] = op.lo[d] + op.n[d] - 1; }
            wrs[q] = {w, (op.type == GHOST_NEUM || op.type == GHOST_DIRI) ? 0 : op.dstf};
            reads_of(op, rds[q]);
            int st = 0;
            for (size_t e = 0; e < q; ++e) {
                bool dep = wrs[e].field == wrs[q].field && !(wrs[e].box & wrs[q].box).empty();            // WAW
                for (const Acc& r : rds[q]) dep = dep || (r.field == wrs[e].field && !(r.box & wrs[e].box).empty());   // RAW
                for (const Acc& r : rds[e]) dep = dep || (r.field == wrs[q].field && !(r.box & wrs[q].box).empty());   // WAR
                if (dep) st = std::max(st, stageOf[pi][e] + 1);
            }
            stageOf[pi][q] = st;
            ns = std::max(ns, (size_t)st + 1);
        }
    }
    std::vector<std::vector<GhostOp>> stages(ns);
    for (size_t pi = 0; pi < perPatch.size(); ++pi)
        for (size_t q = 0; q < perPatch[pi].size(); ++q) stages[stageOf[pi][q]].push_back(perPatch[pi][q]);
    return stages;
}

// Dead ops of a program.  (Frame programs -- the marching kernels' form -- read psi in the boxes' one-cell frames only, and "psi"
// inside a box is phi; the other programs leave all of psi to the direct kernels.)  The reference's sequence writes some regions more than once before anything reads them -- on a box under a Neumann
// wall the smoother's own order-1 extrapolation of the whole ghost face is overwritten by the order-2 extrapolation the Neumann
// ghost wants, and the copy of the first valid layer into psi lands where nobody looks -- and each such op is a launch (a stage)
// on a large level.  Backward liveness over the cells of each box's FAB: at the end phi is live everywhere, psi in the frame only
// (frame programs) or everywhere;
// an op none of whose cells is live is dropped; a kept op kills what it writes and revives what it reads.  Same final values.
namespace {
struct ReadOff { int o[3]; int field; };
void reads_offsets(const GhostOp& op, std::vector<ReadOff>& rd)
{
    rd.clear();
    if (op.type == GHOST_COPY) rd.push_back({{0, 0, 0}, op.srcf});
    else if (op.type == GHOST_EXTRAP || op.type == GHOST_DIRI) {
        const int far = op.type == GHOST_DIRI ? 1 : (op.order == 2 ? 3 : (op.order == 1 ? 2 : 1));
        for (int k = 1; k <= far; ++k) {
            ReadOff r{{0, 0, 0}, op.srcf};
            r.o[op.dir] = -op.sgn * k;
            rd.push_back(r);
        }
    } else {  // GHOST_NEUM (ghost_op_body): psi on the ghost layer and the first valid layer at +-1 along each tangential direction
              // (no diagonal), phi on the valid layer
        const int a = op.dir;
        for (int da = 0; da < 2; ++da)
            for (int t = 1; t <= 2; ++t)
                for (int sg = -1; sg <= 1; sg += 2) {
                    ReadOff r{{0, 0, 0}, 1};
                    r.o[a] = da ? -op.sgn : 0;
                    r.o[(a + t) % 3] = sg;
                    rd.push_back(r);
                }
        ReadOff v{{0, 0, 0}, 0};
        v.o[a] = -op.sgn;
        rd.push_back(v);
    }
}
}  // namespace

static void drop_dead_ops(const Level& L, std::vector<std::vector<GhostOp>>& perPatch, bool frames)
{
    std::vector<ReadOff> rd;
    for (int pi = 0; pi < L.npatches(); ++pi) {
        const IBox valid = L.boxes[L.local[pi]];
        // the FAB grown once more: reads may reach one cell beyond it (they are clipped away below)
        int n[3], g[3];
        for (int d = 0; d < 3; ++d) { g[d] = L.active[d] ? 1 : 0; n[d] = valid.size(d) + 2 * g[d]; }
        const long long tot = (long long)n[0] * n[1] * n[2];
        auto inside = [&](int i, int j, int k) {   // local coordinates relative to the valid low corner
            return i >= -g[0] && i < n[0] - g[0] && j >= -g[1] && j < n[1] - g[1] && k >= -g[2] && k < n[2] - g[2];
        };
        auto at = [&](int i, int j, int k) { return (long long)(i + g[0]) + (long long)n[0] * ((j + g[1]) + (long long)n[1] * (k + g[2])); };
        std::vector<char> live[2];
        live[0].assign(tot, 1);                    // phi: anything may be read later
        live[1].assign(tot, frames ? 0 : 1);       // psi: the frame only (frame programs) / everything (the direct kernels read it all)
        for (int k = -g[2]; frames && k < n[2] - g[2]; ++k)
            for (int j = -g[1]; j < n[1] - g[1]; ++j)
                for (int i = -g[0]; i < n[0] - g[0]; ++i) {
                    const bool in_valid = i >= 0 && i < valid.size(0) && j >= 0 && j < valid.size(1) && k >= 0 && k < valid.size(2);
                    // ... minus its eight corner cells: a 19-point stencil reads no body diagonal
                    const bool corner = (i < 0 || i >= valid.size(0)) && (j < 0 || j >= valid.size(1)) && (k < 0 || k >= valid.size(2));
                    if (!in_valid && !corner) live[1][at(i, j, k)] = 1;
                }
        std::vector<GhostOp>& ops = perPatch[pi];
        std::vector<char> keep(ops.size(), 1);
        for (size_t q = ops.size(); q-- > 0;) {
            const GhostOp& op = ops[q];
            const int wf = (op.type == GHOST_NEUM || op.type == GHOST_DIRI) ? 0 : op.dstf;
            bool any = false;
            for (int k = op.lo[2]; k < op.lo[2] + op.n[2] && !any; ++k)
                for (int j = op.lo[1]; j < op.lo[1] + op.n[1] && !any; ++j)
                    for (int i = op.lo[0]; i < op.lo[0] + op.n[0]; ++i)
                        if (inside(i, j, k) && live[wf][at(i, j, k)]) { any = true; break; }
            if (!any) { keep[q] = 0; continue; }
            reads_offsets(op, rd);
            // kill first (an op never reads what it writes), then revive the inputs
            for (int k = op.lo[2]; k < op.lo[2] + op.n[2]; ++k)
                for (int j = op.lo[1]; j < op.lo[1] + op.n[1]; ++j)
                    for (int i = op.lo[0]; i < op.lo[0] + op.n[0]; ++i)
                        if (inside(i, j, k)) live[wf][at(i, j, k)] = 0;
            for (const ReadOff& r : rd)
                for (int k = op.lo[2]; k < op.lo[2] + op.n[2]; ++k)
                    for (int j = op.lo[1]; j < op.lo[1] + op.n[1]; ++j)
                        for (int i = op.lo[0]; i < op.lo[0] + op.n[0]; ++i) {
                            const int a = i + r.o[0], b = j + r.o[1], c = k + r.o[2];
                            if (!inside(a, b, c)) continue;
                            // a read of psi inside the box is a read of phi (k_ghost_ops<true>)
                            const bool in_valid = a >= 0 && a < valid.size(0) && b >= 0 && b < valid.size(1) && c >= 0 && c < valid.size(2);
                            live[(frames && r.field == 1 && in_valid) ? 0 : r.field][at(a, b, c)] = 1;
                        }
        }
        std::vector<GhostOp> kept;
        for (size_t q = 0; q < ops.size(); ++q) if (keep[q]) kept.push_back(ops[q]);
        ops.swap(kept);
    }
}

// Programs of one level.  which = 0: operator (fillExtrap order 2, then the Neumann ghosts of phi);
// which = 1: smoother (extrapolation order 1 from the domain box, then the Neumann ghosts of phi).
// The leading full copy psi := phi is done by the caller with one flat copy.
static std::vector<std::vector<GhostOp>> build_program(const Level& L, int which, const double bcv[3][2], bool frames = false)
{
    std::vector<std::vector<GhostOp>> perPatch(L.npatches());
    if (which == 3) {
        // ExtrapolateCFEV (LevelData version, ExtrapolationUtils.cpp:165-372): per box, per direction, low then high
        // side: on the bounding box of that side's CF cells (cfivs.minBox & FAB) -- SpaceDim 2: the two cells beyond
        // its ends; SpaceDim 3: per tangential direction and side the edge cells (EXTRAPOLATEFACENOEV along that
        // direction), then the two vertex cells beyond the ends of that edge.  All of order 2, in place.
        int g1[3], gp[3];
        for (int d = 0; d < 3; ++d) { g1[d] = L.active[d] ? 1 : 0; gp[d] = (L.periodic[d] && L.active[d]) ? 1 : 0; }
        const IBox domG = L.domain.grow(gp);
        const bool flat = !L.active[2];
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox valid = L.boxes[L.local[pi]];
            Builder B{valid, valid.grow(g1), pi, perPatch[pi]};
            auto point_extrap = [&](const IBox& cell, int vdir, int sgn) {
                // the cell one step beyond `cell` in direction vdir (sign sgn), from cell, cell -+ 1, cell -+ 2
                IBox dst = cell;
                dst.lo[vdir] += sgn;
                dst.hi[vdir] += sgn;
                if ((dst & B.fab).empty()) return;
                B.push(GHOST_EXTRAP, dst, vdir, sgn, 2, 0, 0);
            };
            for (int dir = 0; dir < 3; ++dir) {
                if (!L.active[dir]) continue;
                for (int s = 0; s < 2; ++s) {
                    const IBox gb = adj_cell(valid, dir, s) & domG;
                    if (gb.empty()) continue;
                    const std::vector<IBox> un = uncovered(gb, L.boxes, L.domain, L.periodic);
                    if (un.empty()) continue;
                    IBox face = un[0];
                    for (const IBox& u : un)
                        for (int q = 0; q < 3; ++q) { face.lo[q] = std::min(face.lo[q], u.lo[q]); face.hi[q] = std::max(face.hi[q], u.hi[q]); }
                    face = face & B.fab;
                    if (face.empty()) continue;
                    auto lo_cell = [](const IBox& b) { IBox c = b; for (int q = 0; q < 3; ++q) c.hi[q] = c.lo[q]; return c; };
                    auto hi_cell = [](const IBox& b) { IBox c = b; for (int q = 0; q < 3; ++q) c.lo[q] = c.hi[q]; return c; };
                    if (flat) {
                        const int vdir = 1 - dir;
                        if (!L.active[vdir]) continue;
                        point_extrap(lo_cell(face), vdir, -1);
                        point_extrap(hi_cell(face), vdir, +1);
                        continue;
                    }
                    for (int edir = 0; edir < 3; ++edir) {
                        if (edir == dir || !L.active[edir]) continue;
                        for (int es = 0; es < 2; ++es) {
                            const IBox edge = adj_cell(face, edir, es) & B.fab;
                            if (edge.empty()) continue;
                            B.push(GHOST_EXTRAP, edge, edir, es ? 1 : -1, 2, 0, 0);
                            const int vdir = 3 - edir - dir;
                            if (!L.active[vdir]) continue;
                            point_extrap(lo_cell(edge), vdir, -1);
                            point_extrap(hi_cell(edge), vdir, +1);
                        }
                    }
                }
            }
        }
        return schedule_stages(perPatch);
    }
    int g1[3];
    for (int d = 0; d < 3; ++d) g1[d] = L.active[d] ? 1 : 0;
    int gper[3];
    for (int d = 0; d < 3; ++d) gper[d] = (L.periodic[d] && L.active[d]) ? 1 : 0;
    const IBox validDomain = L.domain.grow(gper);  // m_validDomain, MappedAMRPoissonOp.cpp:278-283
    for (int pi = 0; pi < L.npatches(); ++pi) {
        const IBox valid = L.boxes[L.local[pi]];
        Builder B{valid, valid.grow(g1), pi, perPatch[pi]};
        if (frames) {
            // the frame copy psi := phi as the FIRST ops of the box, cut into the frame's 26 pieces (faces, edges, corners): the
            // pieces a later op overwrites before anything reads them are dead (drop_dead_ops), the others conflict with nothing
            // and share the first launch with the first real ops -- no launch of its own for the copy
            const IBox fab = valid.grow(g1);
            for (int c = 0; c < 27; ++c) {
                const int t[3] = {c % 3, (c / 3) % 3, c / 9};   // 0: low ghost layer, 1: valid range, 2: high ghost layer
                if (t[0] == 1 && t[1] == 1 && t[2] == 1) continue;
                IBox piece = valid;
                bool ok = true;
                for (int d = 0; d < 3; ++d) {
                    if (t[d] == 1) continue;
                    if (!L.active[d]) { ok = false; break; }
                    piece.lo[d] = piece.hi[d] = t[d] == 0 ? valid.lo[d] - 1 : valid.hi[d] + 1;
                }
                if (ok) B.push(GHOST_COPY, piece & fab, 0, 0, 0, 1, 0);
            }
        }
        if (which == 0 || which == 2) {
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
        for (int d = 0; d < 3 && which != 2; ++d) {
            if (!L.active[d] || L.periodic[d]) continue;
            for (int side = 0; side < 2; ++side) {
                const int vend = side ? valid.hi[d] : valid.lo[d];
                const int dend = side ? L.domain.hi[d] : L.domain.lo[d];
                if (vend != dend) continue;
                const IBox ghostBox = adj_cell(valid, d, side) & B.fab;
                if (ghostBox.empty()) continue;
                if (L.bc_type[d][side] == BC_DIRI) {
                    // setSideDiriBC (order 1) needs no extrap: ghost = 2 value - first cell, value 0 when homogeneous
                    B.push(GHOST_DIRI, ghostBox, d, side ? 1 : -1, 1, 0, 0);
                    B.out.back().val = bcv[d][side];
                    continue;
                }
                if (L.bc_type[d][side] != BC_NEUM) continue;
                B.face_and_copy(1, 0, valid, d, side, 2, L.active);  // ex <- extrapolation of phi, order 2
                B.push(GHOST_NEUM, ghostBox, d, side ? 1 : -1, 0, 0, 1);
            }
        }
    }
    static const bool dce = getenv("SOMAR_NO_GHOST_DCE") == nullptr;   // A/B switch
    if (dce && frames) {
        // every op cut along the faces of its box (low ghost layer | valid range | high ghost layer, per direction): dependences
        // and liveness are then judged piece by piece -- an edge strip no longer waits for the strip that only feeds its corner
        // cell, and the corner pieces, which nothing reads, die
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox valid = L.boxes[L.local[pi]];
            std::vector<GhostOp> cut;
            for (const GhostOp& op : perPatch[pi]) {
                int lo[3][3], hi[3][3], np[3];
                for (int d = 0; d < 3; ++d) {
                    // local coordinates: the valid range is [0, n)
                    const int a = op.lo[d], b = op.lo[d] + op.n[d] - 1, n = valid.size(d);
                    np[d] = 0;
                    if (a < 0) { lo[d][np[d]] = a; hi[d][np[d]] = std::min(b, -1); ++np[d]; }
                    if (b >= 0 && a < n) { lo[d][np[d]] = std::max(a, 0); hi[d][np[d]] = std::min(b, n - 1); ++np[d]; }
                    if (b >= n) { lo[d][np[d]] = std::max(a, n); hi[d][np[d]] = b; ++np[d]; }
                }
                for (int z = 0; z < np[2]; ++z)
                    for (int y = 0; y < np[1]; ++y)
                        for (int x = 0; x < np[0]; ++x) {
                            GhostOp o = op;
                            const int q[3] = {x, y, z};
                            for (int d = 0; d < 3; ++d) { o.lo[d] = lo[d][q[d]]; o.n[d] = hi[d][q[d]] - lo[d][q[d]] + 1; }
                            cut.push_back(o);
                        }
            }
            perPatch[pi].swap(cut);
        }
    }
    if (dce) drop_dead_ops(L, perPatch, frames);
    return schedule_stages(perPatch);
}

// copy_frames: psi := phi in the one-cell frame of every box (all six slabs of all boxes in ONE launch: they overlap only in
// edge / corner cells, where they write the same values).  (The frame programs carry their own copy, piece by piece.)
static std::vector<GhostOp> frame_copy_stage(const Level& L)
{
    std::vector<GhostOp> out;
    for (int pi = 0; pi < L.npatches(); ++pi) {
        const IBox valid = L.boxes[L.local[pi]];
        int g1[3];
        for (int d = 0; d < 3; ++d) g1[d] = L.active[d] ? 1 : 0;
        const IBox fab = valid.grow(g1);
        for (int a = 0; a < 3; ++a) {
            if (!L.active[a]) continue;
            for (int side = 0; side < 2; ++side) {
                IBox slab = fab;
                if (side == 0) slab.hi[a] = slab.lo[a];
                else slab.lo[a] = slab.hi[a];
                GhostOp op;
                std::memset(&op, 0, sizeof(op));
                op.patch = pi;
                op.type = GHOST_COPY;
                for (int q = 0; q < 3; ++q) { op.lo[q] = slab.lo[q] - valid.lo[q]; op.n[q] = slab.size(q); }
                op.dstf = 1;
                op.srcf = 0;
                out.push_back(op);
            }
        }
    }
    return out;
}

void PressureSolver::free_program(FullProgram& P)
{
    hipFree(P.d_ops);
    hipFree(P.d_box_ops);
    hipFree(P.d_box_first);
    P.d_ops = nullptr;
    P.d_box_ops = nullptr;
    P.d_box_first = nullptr;
}

// stages (the ops of stage s may run together, stage s + 1 after them) -> the device tables of both forms
void PressureSolver::upload_program(FullProgram& P, const std::vector<std::vector<GhostOp>>& stages, int npatches)
{
    free_program(P);
    std::vector<GhostOp> flat;
    P.first.clear();
    P.count.clear();
    P.h_box_ops.clear();
    P.h_box_first.assign(npatches + 1, 0);
    std::vector<std::vector<GhostOp>> byBox(npatches);
    for (size_t s = 0; s < stages.size(); ++s) {
        P.first.push_back((int)flat.size());
        P.count.push_back((int)stages[s].size());
        for (GhostOp op : stages[s]) {
            op.pad_ = (int)s;
            flat.push_back(op);
            byBox[op.patch].push_back(op);
        }
    }
    if (flat.empty()) return;
    if (getenv("SOMAR_TIMING") && atoi(getenv("SOMAR_TIMING")) > 1) {
        fprintf(stderr, "[somar timing] ghost program: %zu stages, %zu ops, %d boxes; ops per stage:", stages.size(), flat.size(), npatches);
        for (size_t q = 0; q < stages.size(); ++q) {
            long long cells = 0;
            for (const GhostOp& o : stages[q]) cells += (long long)o.n[0] * o.n[1] * o.n[2];
            fprintf(stderr, " %zu (%lld cells)", stages[q].size(), cells);
        }
        fprintf(stderr, "\n");
    }
    std::vector<GhostOp> sorted;
    std::vector<int> first(npatches + 1, 0);
    for (int b = 0; b < npatches; ++b) {
        first[b] = (int)sorted.size();
        // stage order is kept (stages were walked in order).  pad_ of the box-sorted copy = stage | (how many ops of the same
        // stage FOLLOW this one in the box's list) << 16: a kernel finds the end of a stage without walking the list
        std::vector<GhostOp>& v = byBox[b];
        for (size_t q = 0; q < v.size(); ++q) {
            size_t e = q + 1;
            while (e < v.size() && v[e].pad_ == v[q].pad_) ++e;
            SOMAR_CHECK(v[q].pad_ < 65536 && e - q - 1 < 32768, "ghost program too long for its stage encoding");
            const int st = v[q].pad_;
            for (size_t r = q; r < e; ++r) v[r].pad_ = st | (int)((e - r - 1) << 16);
            q = e - 1;
        }
        sorted.insert(sorted.end(), v.begin(), v.end());
    }
    first[npatches] = (int)sorted.size();
    P.max_box_ops = 0;
    for (int b = 0; b < npatches; ++b) P.max_box_ops = std::max(P.max_box_ops, first[b + 1] - first[b]);
    SOMAR_HIP(hipMalloc(&P.d_ops, flat.size() * sizeof(GhostOp)));
    SOMAR_HIP(hipMemcpy(P.d_ops, flat.data(), flat.size() * sizeof(GhostOp), hipMemcpyHostToDevice));
    SOMAR_HIP(hipMalloc(&P.d_box_ops, sorted.size() * sizeof(GhostOp)));
    SOMAR_HIP(hipMemcpy(P.d_box_ops, sorted.data(), sorted.size() * sizeof(GhostOp), hipMemcpyHostToDevice));
    P.h_box_ops = sorted;
    P.h_box_first = first;
    SOMAR_HIP(hipMalloc(&P.d_box_first, first.size() * sizeof(int)));
    SOMAR_HIP(hipMemcpy(P.d_box_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
}

// One application of a ghost program: on small levels ONE launch (a workgroup per box walks the box's ops, a workgroup barrier
// at every stage boundary), on large ones a launch per stage.  copy_all: psi := phi first.
void PressureSolver::run_program(int d, const FullProgram& P, double* phi, double* psi, bool homogeneous, bool redirect,
                                 bool copy_all)
{
    Level& L = *lev[d];
    if (box_program(d) && (P.d_box_ops || copy_all)) {
        if (!P.d_box_ops) { launch_copy(st_, psi, phi, L.field_elems); return; }
        launch_ghost_program(st_, L.dev, P.d_box_ops, P.d_box_first, phi, psi, homogeneous, redirect, copy_all);
        ++counters[1];
        return;
    }
    if (copy_all) launch_copy(st_, psi, phi, L.field_elems);  // psi := phi (valid cells and exchanged ghosts)
    ++counters[2];
    for (size_t s = 0; s < P.first.size(); ++s)
        launch_ghost_ops(st_, L.dev, P.d_ops + P.first[s], P.count[s], phi, psi, homogeneous, redirect);
}

void PressureSolver::build_full_programs(int d)
{
    Level& L = *lev[d];
    for (int which = 0; which < 6; ++which) {
        if ((which == 2 || which == 3) && !(hasCF_ || d == 0)) continue;  // [2] serves the flux register (depth 0), [3] needs CF faces
        if (which == 3 && !hasCF_) continue;
        if (which >= 4 && !full_march(d)) continue;   // [4] / [5]: [0] / [1] for the marching kernels (psi in frames only)
        FullProgram& P = full_prog_[d][which];
        auto stages = build_program(L, which >= 4 ? which - 4 : which, bc_value_, which >= 4);
        upload_program(P, stages, L.npatches());
    }
    if (full_march(d)) upload_program(full_prog_[d][6], {frame_copy_stage(L)}, L.npatches());
    SOMAR_HIP(hipDeviceSynchronize());
}

void PressureSolver::run_full_program(int d, int which, double* phi, bool homogeneous)
{
    run_program(d, full_prog_[d][which], phi, f_psi[d], homogeneous, false, true);
}

// The same call sequences for the marching kernels (full19_march.hip): psi is written in the boxes' one-cell frames
// only -- stage 0 copies phi's frame, the ops read "psi" inside a box's valid region from phi (k_ghost_ops<true>) --
// so the whole-field copy psi := phi (16 B/cell per application) is gone.  which: 0 operator, 1 smoother.
void PressureSolver::run_full_program_frames(int d, int which, double* phi, bool homogeneous)
{
    run_program(d, full_prog_[d][which + 4], phi, f_psi[d], homogeneous, true, false);
}

// the frame copy (six slabs per box, one launch) with (src, dst) in the roles of (phi, psi)
void PressureSolver::copy_frames(int d, const double* src, double* dst)
{
    Level& L = *lev[d];
    const FullProgram& P = full_prog_[d][6];
    if (P.first.empty()) return;
    launch_ghost_ops(st_, L.dev, P.d_ops + P.first[0], P.count[0], const_cast<double*>(src), dst, true, false);
}

void PressureSolver::make_full()
{
    SOMAR_CHECK(!lev.empty() && !finalized, "make_full before define / after finalize");
    full_ = true;
    alloc_full_metric(*lev[0]);
}

void PressureSolver::run_aux_program(int which, double* phi)
{
    SOMAR_CHECK(finalized && (which == 0 || which == 1), "run_aux_program: bad argument");
    Level& L = *lev[0];
    FullProgram& P = aux_prog_[which];
    if (!aux_built_[which]) {
        std::vector<std::vector<GhostOp>> perPatch(L.npatches());
        int g1[3];
        for (int d = 0; d < 3; ++d) g1[d] = L.active[d] ? 1 : 0;
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox valid = L.boxes[L.local[pi]];
            Builder B{valid, valid.grow(g1), pi, perPatch[pi]};
            if (which == 0) {
                IBox v = valid;
                for (int d = 0; d < 3; ++d) {
                    if (!L.active[d]) continue;
                    B.face_no_ev(0, 0, v, d, 0, 2);
                    B.face_no_ev(0, 0, v, d, 1, 2);
                    v = grow_dir(v, d, 1);
                }
            } else {
                const IBox domValid = B.fab & L.domain;
                int vd = 2;
                while (vd > 0 && !L.active[vd]) --vd;
                B.face_and_copy(0, 0, domValid, vd, 0, 2, L.active);
                B.face_and_copy(0, 0, domValid, vd, 1, 2, L.active);
            }
        }
        upload_program(P, schedule_stages(perPatch), L.npatches());
        SOMAR_HIP(hipDeviceSynchronize());
        aux_built_[which] = true;
    }
    // these ops only use phi (dst = src = field 0); the metric planes are not read (no NEUM op)
    run_program(0, P, phi, phi, true, false, false);
}

void PressureSolver::cf_ev(int d, double* phi)
{
    if (!full_ || !hasCF_) return;
    run_program(d, full_prog_[d][3], phi, phi, true, false, false);
}

double* const* PressureSolver::flux_fields(double* phi)
{
    SOMAR_CHECK(full_ && finalized, "flux fields are for the non-diagonal path");
    Level& L = *lev[0];
    for (int a = 0; a < prm.spaceDim; ++a)
        if (!f_flux[a]) f_flux[a] = L.alloc_field();
    run_program(0, full_prog_[0][2], phi, f_psi[0], true, false, true);
    launch_flux_full(st_, L.dev, f_flux, phi, f_psi[0]);
    return f_flux;
}

void PressureSolver::flux_at_faces(double* phi, FullFlux& ff)
{
    SOMAR_CHECK(full_ && finalized, "flux_at_faces is for the non-diagonal path");
    Level& L = *lev[0];
    run_program(0, full_prog_[0][2], phi, f_psi[0], true, false, true);   // psi := phi, then fillExtrap (order 2)
    ff.psi = f_psi[0];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) ff.J[a][b] = L.dev.jgf[a][b];
        ff.dxi[a] = 1.0 / L.dx[a];
    }
}

void PressureSolver::mac_grad_full(double* phi)
{
    SOMAR_CHECK(full_ && finalized, "mac_grad_full is the non-diagonal path");
    // On a refined level the caller has filled the coarse-fine ghosts (quadratic interpolation + ExtrapolateCFEV,
    // Gradient.cpp:104-114); they lie inside the domain, so fillExtrap copies them like any exchanged ghost.
    Level& L = *lev[0];
    if (!d_extrapbc_ops_) {
        // EllipticExtrapBCGhostClass -> setSideExtrapBC, order 2: the face-adjacent ghost layer of every box side on a
        // non-periodic domain boundary (BCInterface/EllipticBCUtils.cpp:1011-1050, 224-312)
        std::vector<GhostOp> ops;
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox valid = L.boxes[L.local[pi]];
            for (int a = 0; a < 3; ++a) {
                if (!L.active[a] || L.periodic[a]) continue;
                for (int s = 0; s < 2; ++s) {
                    if ((s ? valid.hi[a] : valid.lo[a]) != (s ? L.domain.hi[a] : L.domain.lo[a])) continue;
                    GhostOp op;
                    std::memset(&op, 0, sizeof(op));
                    op.patch = pi;
                    op.type = GHOST_EXTRAP;
                    for (int q = 0; q < 3; ++q) { op.lo[q] = 0; op.n[q] = valid.size(q); }
                    op.lo[a] = s ? valid.size(a) : -1;
                    op.n[a] = 1;
                    op.dir = a;
                    op.sgn = s ? 1 : -1;
                    op.order = 2;
                    ops.push_back(op);
                }
            }
        }
        n_extrapbc_ops_ = (int)ops.size();
        if (!ops.empty()) {
            SOMAR_HIP(hipMalloc(&d_extrapbc_ops_, ops.size() * sizeof(GhostOp)));
            SOMAR_HIP(hipMemcpy(d_extrapbc_ops_, ops.data(), ops.size() * sizeof(GhostOp), hipMemcpyHostToDevice));
            SOMAR_HIP(hipDeviceSynchronize());
        }
    }
    for (int a = 0; a < prm.spaceDim; ++a)
        if (!f_flux[a]) f_flux[a] = L.alloc_field();
    run_program(0, full_prog_[0][2], phi, f_psi[0], true, false, true);
    // one op per (box, side): different sides of a box write different cells and read only valid ones
    launch_ghost_ops(st_, L.dev, d_extrapbc_ops_, n_extrapbc_ops_, phi, phi);
    launch_flux_full(st_, L.dev, f_flux, phi, f_psi[0]);
}

void PressureSolver::set_metric_full(int patch, const double* jg0, const double* jg1, const double* jg2,
                                     const double* jinv)
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_metric before define / after finalize");
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

// LevelGeometry's FC J g^{ab} / CC J^{-1} of a coordinate map evaluated on the device into the level's resident arrays
// (maps.hip): what set_metric_ortho / set_metric_full upload, for every local patch at once.  Only the nodal depth of a
// bathymetric map (O(N^2)) crosses PCIe.
void PressureSolver::set_metric_map(int kind, const double Lc[3], const double* depth, const int dlo[2], const int dn[2])
{
    SOMAR_CHECK(!lev.empty() && !finalized, "set_metric before define / after finalize");
    SOMAR_CHECK(prm.spaceDim == 3, "the map producers restate the CH_SPACEDIM = 3 algebra (GeoSourceInterface.cpp:236-291)");
    SOMAR_CHECK(kind >= 1 && kind <= 4, "map kind: 1 cylindrical, 2 bathymetric, 3 twisted (type 0), 4 twisted (type 1)");
    Level& L = *lev[0];
    SOMAR_CHECK(L.npatches() < 65536, "too many local patches for one launch");
    double* d_depth = nullptr;
    int lo[2] = {0, 0}, n[2] = {0, 0};
    if (kind == 2) {
        SOMAR_CHECK(depth && dn[0] > 0 && dn[1] > 0, "the bathymetric map needs the nodal depth");
        SOMAR_CHECK(Lc[0] > 0.0 && Lc[1] > 0.0 && Lc[2] > 0.0, "the bathymetric map needs the domain lengths");
        // FILL_BATHYDZDXI after CONVERTFAB reaches nodes lo-1 .. hi+2 of a box (BathymetricBaseMap.cpp:190-196)
        for (int pi = 0; pi < L.npatches(); ++pi) {
            const IBox b = L.boxes[L.local[pi]];
            for (int d = 0; d < 2; ++d)
                SOMAR_CHECK(dlo[d] <= b.lo[d] - 1 && dlo[d] + dn[d] - 1 >= b.hi[d] + 2,
                            "nodal depth must cover nodes lo-1 .. hi+2 of every local box in both horizontal directions");
        }
        const size_t bytes = (size_t)dn[0] * dn[1] * sizeof(double);
        SOMAR_HIP(hipMalloc(&d_depth, bytes));
        SOMAR_HIP(hipMemcpy(d_depth, depth, bytes, hipMemcpyHostToDevice));
        for (int d = 0; d < 2; ++d) { lo[d] = dlo[d]; n[d] = dn[d]; }
        full_ = true;
        alloc_full_metric(L);
    } else if (kind == 3 || kind == 4) {
        // TwistedMap, m_twistType 0 / 1: Lc = the amplitudes m_pert; 2 pi |pert| < 1 keeps the Jacobian positive.  Type 1 takes its
        // derivatives and its Jacobian from differences of the coordinate functions (GeoSourceInterface's defaults) over the domain
        // lengths m_L = dx * cells
        full_ = true;
        alloc_full_metric(L);
    } else {
        SOMAR_CHECK(!full_, "the cylindrical map is diagonal: the level already holds a non-diagonal metric");
        // r = dXi0 (i + offset) > 0 on every face / cell the level owns
        SOMAR_CHECK(L.domain.lo[0] >= 0, "the cylindrical map needs r >= 0: domain index 0 starts below zero");
    }
    double domLen[3];
    for (int d = 0; d < 3; ++d) domLen[d] = L.dx[d] * (double)L.domain.size(d);
    launch_map_metric(st_, L.dev, kind, L.dx, Lc, d_depth, lo, n, kind == 1, domLen);
    sync();
    if (d_depth) hipFree(d_depth);
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
