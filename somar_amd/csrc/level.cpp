// somar_amd/csrc/level.cpp -- layout tables, HBM allocation, ghost-exchange plan.
#include "level.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>

namespace somar {

bool coarsenable(const std::vector<IBox>& boxes, const int* r)
{
    // Chombo: refine(coarsen(b, r), r) == b for every box
    for (const IBox& b : boxes)
        if (!(b.coarsen(r).refine(r) == b)) return false;
    return true;
}

template <class T>
static T* to_device(const std::vector<T>& v)
{
    if (v.empty()) return nullptr;
    T* d = nullptr;
    SOMAR_HIP(hipMalloc(&d, v.size() * sizeof(T)));
    SOMAR_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

// Pure host logic (no GPU): the ghost-exchange plan of rank `myrank` for the layout -- Chombo's
// Copier(grids, grids, domain, ghost, exchange=true).  Items carry GLOBAL box indices in
// src_patch/dst_patch.  Both sides of a rank pair enumerate (dst box, src box, periodic shift) in the
// same order, so the i-th item of a send message is the i-th item of the matching receive message.
ExchangePlan build_exchange_plan(const IBox& domain, const bool periodic[3], const int ghost[3],
                                 const std::vector<IBox>& boxes, const std::vector<int>& owner, int myrank)
{
    ExchangePlan plan;
    std::vector<std::array<int, 3>> shifts;
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b)
            for (int cc = -1; cc <= 1; ++cc) {
                if ((a && !periodic[0]) || (b && !periodic[1]) || (cc && !periodic[2])) continue;
                shifts.push_back({a * domain.size(0), b * domain.size(1), cc * domain.size(2)});
            }
    struct Remote { int peer; CopyItem it; };
    std::vector<Remote> sends, recvs;
    for (size_t di = 0; di < boxes.size(); ++di) {
        const IBox gbox = boxes[di].grow(ghost);
        for (size_t si = 0; si < boxes.size(); ++si) {
            const bool dl = owner[di] == myrank, sl = owner[si] == myrank;
            if (!dl && !sl) continue;
            for (const auto& sh : shifts) {
                if (si == di && sh[0] == 0 && sh[1] == 0 && sh[2] == 0) continue;
                const IBox img = boxes[si].shift(sh.data());
                const IBox r = gbox & img;
                if (r.empty()) continue;
                CopyItem it;
                std::memset(&it, 0, sizeof(it));
                it.src_patch = (int)si;
                it.dst_patch = (int)di;
                for (int d = 0; d < 3; ++d) {
                    it.n[d] = r.size(d);
                    it.dst_lo[d] = r.lo[d] - boxes[di].lo[d];
                    it.src_lo[d] = r.lo[d] - sh[d] - boxes[si].lo[d];
                }
                if (dl && sl) plan.local.push_back(it);
                else if (sl) sends.push_back({owner[di], it});
                else recvs.push_back({owner[si], it});
            }
        }
    }
    for (auto& r : sends) plan.peers.push_back(r.peer);
    for (auto& r : recvs) plan.peers.push_back(r.peer);
    std::sort(plan.peers.begin(), plan.peers.end());
    plan.peers.erase(std::unique(plan.peers.begin(), plan.peers.end()), plan.peers.end());
    for (int q : plan.peers) {
        plan.soff.push_back(plan.send_total);
        for (auto& r : sends)
            if (r.peer == q) {
                plan.send_items.push_back(r.it);
                plan.send_itemoff.push_back(plan.send_total);
                plan.send_total += (long long)r.it.n[0] * r.it.n[1] * r.it.n[2];
            }
        plan.scount.push_back(plan.send_total - plan.soff.back());
        plan.roff.push_back(plan.recv_total);
        for (auto& r : recvs)
            if (r.peer == q) {
                plan.recv_items.push_back(r.it);
                plan.recv_itemoff.push_back(plan.recv_total);
                plan.recv_total += (long long)r.it.n[0] * r.it.n[1] * r.it.n[2];
            }
        plan.rcount.push_back(plan.recv_total - plan.roff.back());
    }
    return plan;
}

// a minus b as a list of disjoint boxes (at most 6)
void box_subtract(const IBox& a, const IBox& b, std::vector<IBox>& out)
{
    const IBox c = a & b;
    if (c.empty()) { out.push_back(a); return; }
    IBox rest = a;
    for (int d = 2; d >= 0; --d) {
        if (rest.lo[d] < c.lo[d]) { IBox p = rest; p.hi[d] = c.lo[d] - 1; out.push_back(p); rest.lo[d] = c.lo[d]; }
        if (rest.hi[d] > c.hi[d]) { IBox p = rest; p.lo[d] = c.hi[d] + 1; out.push_back(p); rest.hi[d] = c.hi[d]; }
    }
}

std::vector<std::array<int, 3>> periodic_shifts(const IBox& domain, const bool periodic[3])
{
    // the unshifted image first, then the others in (x, y, z) lexicographic order of (-n, 0, +n)
    std::vector<std::array<int, 3>> shifts;
    shifts.push_back({0, 0, 0});
    for (int a = -1; a <= 1; ++a)
        for (int b = -1; b <= 1; ++b)
            for (int cc = -1; cc <= 1; ++cc) {
                if ((a && !periodic[0]) || (b && !periodic[1]) || (cc && !periodic[2])) continue;
                if (!a && !b && !cc) continue;
                shifts.push_back({a * domain.size(0), b * domain.size(1), cc * domain.size(2)});
            }
    return shifts;
}

// `region` minus every box of `boxes` and every periodic image of them
std::vector<IBox> uncovered(const IBox& region, const std::vector<IBox>& boxes, const IBox& domain,
                            const bool periodic[3])
{
    std::vector<IBox> cur{region};
    const auto shifts = periodic_shifts(domain, periodic);
    for (const IBox& b : boxes)
        for (const auto& sh : shifts) {
            const IBox img = b.shift(sh.data());
            if ((img & region).empty()) continue;
            std::vector<IBox> nxt;
            for (const IBox& c : cur) box_subtract(c, img, nxt);
            cur.swap(nxt);
            if (cur.empty()) return cur;
        }
    return cur;
}

static bool contains_cell(const IBox& b, const int c[3])
{
    for (int d = 0; d < 3; ++d)
        if (c[d] < b.lo[d] || c[d] > b.hi[d]) return false;
    return true;
}

// CFRegion of this level + homogeneousCFInterp weights (HomogeneousCFInterp.cpp:56, 72-73)
void Level::define_cf(const double dxCrse[3])
{
    hcf.clear();
    int g[3];
    for (int d = 0; d < 3; ++d) g[d] = periodic[d] ? 1 : 0;
    const IBox dom = domain.grow(g);
    for (int pi = 0; pi < npatches(); ++pi) {
        const PatchDesc& p = hpatches[pi];
        const IBox valid = boxes[local[pi]];
        const long long st[3] = {1, p.pj, p.pk};
        for (int d = 0; d < 3; ++d) {
            if (!active[d]) continue;
            for (int s = 0; s < 2; ++s) {
                IBox gb = valid;
                if (s == 0) { gb.lo[d] = valid.lo[d] - 1; gb.hi[d] = valid.lo[d] - 1; }
                else { gb.lo[d] = valid.hi[d] + 1; gb.hi[d] = valid.hi[d] + 1; }
                gb = gb & dom;
                if (gb.empty()) continue;
                for (const IBox& u : uncovered(gb, boxes, domain, periodic))
                    for (int k = u.lo[2]; k <= u.hi[2]; ++k)
                        for (int j = u.lo[1]; j <= u.hi[1]; ++j)
                            for (int i = u.lo[0]; i <= u.hi[0]; ++i) {
                                CFCell c;
                                c.off = p.off + (i - p.lo[0]) + st[1] * (j - p.lo[1]) + st[2] * (k - p.lo[2]);
                                c.stride = (int)(s ? st[d] : -st[d]);
                                c.dir = d | (p.n[d] == 1 ? 4 : 0);
                                hcf.push_back(c);
                            }
            }
        }
    }
    hipFree(d_cf);
    d_cf = to_device(hcf);
    ncf = (int)hcf.size();
    for (int d = 0; d < 3; ++d) {
        const double Df = dx[d], Dc = dxCrse[d];
        cf_c1[d] = 2.0 * (Dc - Df) / (Dc + Df);
        cf_c2[d] = -(Dc - Df) / (Dc + 3.0 * Df);
        cf_fac[d] = 1.0 - 2.0 * Df / (Df + Dc);
    }

    // ---- what the fused sweep needs (see level.h) ----------------------------------------------------
    cf_fusable = true;
    hcfx = hcf;
    const auto shifts = periodic_shifts(domain, periodic);
    std::vector<IBox> near;  // box images around the patch at hand
    auto covered1 = [&](const int c[3]) {
        for (const IBox& b : near)
            if (contains_cell(b, c)) return true;
        return false;
    };
    for (int pi = 0; pi < npatches(); ++pi) {
        PatchDesc& p = hpatches[pi];
        const IBox valid = boxes[local[pi]];
        const long long st[3] = {1, p.pj, p.pk};
        p.cf = 0;
        near.clear();
        {
            const int g2[3] = {2, 2, 2};
            const IBox around = valid.grow(g2);
            for (const IBox& b : boxes)
                for (const auto& sh : shifts) {
                    const IBox img = b.shift(sh.data());
                    if (!(img & around).empty()) near.push_back(img);
                }
        }
        for (int d = 0; d < 3; ++d) {
            if (!active[d]) continue;
            for (int s = 0; s < 2; ++s) {
                IBox gb = valid;
                if (s == 0) { gb.lo[d] = gb.hi[d] = valid.lo[d] - 1; }
                else { gb.lo[d] = gb.hi[d] = valid.hi[d] + 1; }
                gb = gb & dom;
                if (gb.empty()) continue;
                long long nunc = 0;
                for (const IBox& u : uncovered(gb, boxes, domain, periodic)) nunc += u.numPts();
                if (nunc == gb.numPts()) p.cf |= 1 << (2 * d + s);
                else if (nunc != 0) cf_fusable = false;  // a face that is only partly coarse-fine
                if (nunc != 0 && p.n[d] < 2) cf_fusable = false;
            }
        }
        // edge ghosts: two coordinates one cell outside the box
        for (int d1 = 0; d1 < 3 && cf_fusable; ++d1)
            for (int d2 = d1 + 1; d2 < 3 && cf_fusable; ++d2) {
                if (!active[d1] || !active[d2]) continue;
                const int d3 = 3 - d1 - d2;
                for (int s1 = 0; s1 < 2; ++s1)
                    for (int s2 = 0; s2 < 2; ++s2)
                        for (int t = valid.lo[d3]; t <= valid.hi[d3] && cf_fusable; ++t) {
                            int g[3];
                            g[d1] = s1 ? valid.hi[d1] + 1 : valid.lo[d1] - 1;
                            g[d2] = s2 ? valid.hi[d2] + 1 : valid.lo[d2] - 1;
                            g[d3] = t;
                            if (!contains_cell(dom, g) || covered1(g)) continue;
                            // candidates: directions along which the cell next to g (towards the box) is a real
                            // cell of a neighbouring box, which then owns g as one of ITS face ghosts
                            int ncand = 0, cd = -1, cs = 0;
                            const int dd[2] = {d1, d2}, ss[2] = {s1, s2};
                            for (int q = 0; q < 2; ++q) {
                                int h[3] = {g[0], g[1], g[2]};
                                h[dd[q]] += ss[q] ? -1 : 1;
                                if (!covered1(h)) continue;
                                int h2[3] = {h[0], h[1], h[2]};
                                h2[dd[q]] += ss[q] ? -1 : 1;
                                if (!covered1(h2)) { cf_fusable = false; break; }  // neighbour box one cell wide
                                ++ncand;
                                cd = dd[q];
                                cs = ss[q];
                            }
                            if (ncand == 2) cf_fusable = false;  // re-entrant corner of the refined region
                            if (ncand != 1 || !cf_fusable) continue;
                            CFCell c;
                            c.off = p.off + (g[0] - p.lo[0]) + st[1] * (g[1] - p.lo[1]) + st[2] * (g[2] - p.lo[2]);
                            c.stride = (int)(cs ? st[cd] : -st[cd]);
                            c.dir = cd;
                            hcfx.push_back(c);
                        }
            }
    }
    if (!cf_fusable) hcfx = hcf;
    for (int d = 0; d < 3; ++d) {
        cf_faces[d].reset();
        if (active[d] && ((cf_fusable && ncf > 0) || (!periodic[d] && bc_type[d][1] == BC_DIRI))) {
            int gh[3];
            for (int e = 0; e < 3; ++e) gh[e] = active[e] ? FRAME : 0;
            cf_faces[d].reset(new Copier);
            cf_faces[d]->define_faces(domain, periodic, *this, d, gh, comm);
        }
    }
    hipFree(d_cfx);
    d_cfx = to_device(hcfx);
    ncfx = (int)hcfx.size();
    hipFree(d_patches);
    d_patches = to_device(hpatches);  // the flag bits
    dev.patches = d_patches;
    refresh_params();
    SOMAR_HIP(hipDeviceSynchronize());
}

// ------------------------------------------------------------------------------------
// Copier between two layouts
// ------------------------------------------------------------------------------------
// Pure host logic.  Items carry GLOBAL box indices; every rank enumerates (dst box, src box, shift) in the
// same order, so the i-th item of a send message is the i-th item of the matching receive message.
ExchangePlan build_copy_plan(const IBox& domain, const bool periodic[3], const std::vector<IBox>& srcBoxes,
                             const std::vector<int>& srcOwner, const std::vector<IBox>& dstBoxes,
                             const std::vector<int>& dstOwner, const int ghost[3], int myrank, bool ring_only)
{
    ExchangePlan plan;
    const auto shifts = periodic_shifts(domain, periodic);
    struct Remote { int peer; CopyItem it; };
    std::vector<Remote> sends, recvs;
    for (size_t di = 0; di < dstBoxes.size(); ++di) {
        const IBox gbox = dstBoxes[di].grow(ghost);
        for (size_t si = 0; si < srcBoxes.size(); ++si) {
            const bool dl = dstOwner[di] == myrank, sl = srcOwner[si] == myrank;
            if (!dl && !sl) continue;
            for (const auto& sh : shifts) {
                const IBox r0 = gbox & srcBoxes[si].shift(sh.data());
                if (r0.empty()) continue;
                // ring_only: the part of the region outside the destination box (its ghost ring), as up to six slabs in a fixed
                // order -- every rank derives the same pieces, so sends and receives still pair up
                std::vector<IBox> pieces;
                if (!ring_only) pieces.push_back(r0);
                else {
                    IBox rest = r0;
                    const IBox& V = dstBoxes[di];
                    for (int d = 2; d >= 0 && !rest.empty(); --d) {
                        IBox lo = rest, hi = rest;
                        lo.hi[d] = std::min(rest.hi[d], V.lo[d] - 1);
                        hi.lo[d] = std::max(rest.lo[d], V.hi[d] + 1);
                        if (!lo.empty()) pieces.push_back(lo);
                        if (!hi.empty()) pieces.push_back(hi);
                        rest.lo[d] = std::max(rest.lo[d], V.lo[d]);
                        rest.hi[d] = std::min(rest.hi[d], V.hi[d]);
                    }
                }
                for (const IBox& r : pieces) {
                    CopyItem it;
                    std::memset(&it, 0, sizeof(it));
                    it.src_patch = (int)si;
                    it.dst_patch = (int)di;
                    for (int d = 0; d < 3; ++d) {
                        it.n[d] = r.size(d);
                        it.dst_lo[d] = r.lo[d] - dstBoxes[di].lo[d];
                        it.src_lo[d] = r.lo[d] - sh[d] - srcBoxes[si].lo[d];
                    }
                    if (dl && sl) plan.local.push_back(it);
                    else if (sl) sends.push_back({dstOwner[di], it});
                    else recvs.push_back({srcOwner[si], it});
                }
            }
        }
    }
    for (auto& r : sends) plan.peers.push_back(r.peer);
    for (auto& r : recvs) plan.peers.push_back(r.peer);
    std::sort(plan.peers.begin(), plan.peers.end());
    plan.peers.erase(std::unique(plan.peers.begin(), plan.peers.end()), plan.peers.end());
    for (int q : plan.peers) {
        plan.soff.push_back(plan.send_total);
        for (auto& r : sends)
            if (r.peer == q) {
                plan.send_items.push_back(r.it);
                plan.send_itemoff.push_back(plan.send_total);
                plan.send_total += (long long)r.it.n[0] * r.it.n[1] * r.it.n[2];
            }
        plan.scount.push_back(plan.send_total - plan.soff.back());
        plan.roff.push_back(plan.recv_total);
        for (auto& r : recvs)
            if (r.peer == q) {
                plan.recv_items.push_back(r.it);
                plan.recv_itemoff.push_back(plan.recv_total);
                plan.recv_total += (long long)r.it.n[0] * r.it.n[1] * r.it.n[2];
            }
        plan.rcount.push_back(plan.recv_total - plan.roff.back());
    }
    return plan;
}

Copier::~Copier()
{
    hipFree(d_local); hipFree(d_send); hipFree(d_recv); hipFree(d_soff); hipFree(d_roff); hipFree(d_sbuf); hipFree(d_rbuf);
}

void Copier::define(const IBox& domain, const bool periodic[3], const Level& src, const Level& dst, const int ghost[3],
                    Comm* comm, bool ring_only)
{
    src_ = &src;
    dst_ = &dst;
    comm_ = comm;
    for (int d = 0; d < 3; ++d) SOMAR_CHECK(ghost[d] <= FRAME, "copier ghost wider than the device frame");
    plan = build_copy_plan(domain, periodic, src.boxes, src.owner, dst.boxes, dst.owner, ghost, comm ? comm->rank : 0, ring_only);
    std::vector<int> sp(src.boxes.size(), -1), dp(dst.boxes.size(), -1);
    for (int pi = 0; pi < (int)src.local.size(); ++pi) sp[src.local[pi]] = pi;
    for (int pi = 0; pi < (int)dst.local.size(); ++pi) dp[dst.local[pi]] = pi;
    for (CopyItem& it : plan.local) { it.src_patch = sp[it.src_patch]; it.dst_patch = dp[it.dst_patch]; }
    for (CopyItem& it : plan.send_items) it.src_patch = sp[it.src_patch];
    for (CopyItem& it : plan.recv_items) it.dst_patch = dp[it.dst_patch];
    upload_tables();
}

// Face-centred data of direction `dir`: the source of box b is faces(b, dir) -- the valid cells plus the layer
// that holds the box's HIGH face -- so a neighbour's frame also receives the coefficient of a face that no box
// owns as a low face (a box face on a coarse-fine boundary).  Run it BEFORE the ordinary exchange: where a real
// low-face owner exists its value then overwrites this one.
void Copier::define_faces(const IBox& domain, const bool periodic[3], const Level& L, int dir, const int ghost[3],
                          Comm* comm)
{
    src_ = &L;
    dst_ = &L;
    comm_ = comm;
    std::vector<IBox> fb = L.boxes;
    for (IBox& b : fb) b.hi[dir] += 1;
    plan = build_copy_plan(domain, periodic, fb, L.owner, L.boxes, L.owner, ghost, comm ? comm->rank : 0, false);
    std::vector<int> lp(L.boxes.size(), -1);
    for (int pi = 0; pi < (int)L.local.size(); ++pi) lp[L.local[pi]] = pi;
    for (CopyItem& it : plan.local) { it.src_patch = lp[it.src_patch]; it.dst_patch = lp[it.dst_patch]; }
    for (CopyItem& it : plan.send_items) it.src_patch = lp[it.src_patch];
    for (CopyItem& it : plan.recv_items) it.dst_patch = lp[it.dst_patch];
    // drop the copies of a box onto itself (same cells)
    std::vector<CopyItem> keep;
    for (const CopyItem& it : plan.local)
        if (!(it.src_patch == it.dst_patch && it.src_lo[0] == it.dst_lo[0] && it.src_lo[1] == it.dst_lo[1] &&
              it.src_lo[2] == it.dst_lo[2]))
            keep.push_back(it);
    plan.local.swap(keep);
    upload_tables();
}

void Copier::upload_tables()
{
    d_local = to_device(plan.local);
    d_send = to_device(plan.send_items);
    d_recv = to_device(plan.recv_items);
    d_soff = to_device(plan.send_itemoff);
    d_roff = to_device(plan.recv_itemoff);
    if (plan.send_total) SOMAR_HIP(hipMalloc(&d_sbuf, plan.send_total * sizeof(double)));
    if (plan.recv_total) SOMAR_HIP(hipMalloc(&d_rbuf, plan.recv_total * sizeof(double)));
    SOMAR_HIP(hipDeviceSynchronize());
}

// Every rank ends up with every box: my boxes are copied locally and sent to every other rank, the others'
// boxes are received from their owners.  Both sides walk the boxes in layout order.
ExchangePlan build_allgather_plan(const std::vector<IBox>& boxes, const std::vector<int>& owner, int grow, int myrank,
                                  int nranks)
{
    ExchangePlan plan;
    auto item = [&](size_t b) {
        CopyItem it;
        std::memset(&it, 0, sizeof(it));
        it.src_patch = (int)b;
        it.dst_patch = (int)b;
        for (int d = 0; d < 3; ++d) {
            it.n[d] = boxes[b].size(d) + 2 * grow;
            it.src_lo[d] = -grow;
            it.dst_lo[d] = -grow;
        }
        return it;
    };
    for (size_t b = 0; b < boxes.size(); ++b)
        if (owner[b] == myrank) plan.local.push_back(item(b));
    for (int q = 0; q < nranks; ++q) {
        if (q == myrank) continue;
        plan.peers.push_back(q);
        plan.soff.push_back(plan.send_total);
        for (size_t b = 0; b < boxes.size(); ++b)
            if (owner[b] == myrank) {
                const CopyItem it = item(b);
                plan.send_items.push_back(it);
                plan.send_itemoff.push_back(plan.send_total);
                plan.send_total += (long long)it.n[0] * it.n[1] * it.n[2];
            }
        plan.scount.push_back(plan.send_total - plan.soff.back());
        plan.roff.push_back(plan.recv_total);
        for (size_t b = 0; b < boxes.size(); ++b)
            if (owner[b] == q) {
                const CopyItem it = item(b);
                plan.recv_items.push_back(it);
                plan.recv_itemoff.push_back(plan.recv_total);
                plan.recv_total += (long long)it.n[0] * it.n[1] * it.n[2];
            }
        plan.rcount.push_back(plan.recv_total - plan.roff.back());
    }
    return plan;
}

void Copier::define_allgather(const Level& src, const Level& dst, int grow, Comm* comm)
{
    src_ = &src;
    dst_ = &dst;
    comm_ = comm;
    SOMAR_CHECK(grow >= 0 && grow <= FRAME, "allgather: ghost wider than the device frame");
    SOMAR_CHECK(dst.boxes.size() == src.boxes.size() && dst.local.size() == dst.boxes.size(),
                "allgather: the destination layout must hold every box locally");
    plan = build_allgather_plan(src.boxes, src.owner, grow, comm ? comm->rank : 0, comm ? comm->size : 1);
    std::vector<int> sp(src.boxes.size(), -1);
    for (int pi = 0; pi < (int)src.local.size(); ++pi) sp[src.local[pi]] = pi;
    for (CopyItem& it : plan.local) it.src_patch = sp[it.src_patch];   // dst patch index = global box index
    for (CopyItem& it : plan.send_items) it.src_patch = sp[it.src_patch];
    upload_tables();
}

void Copier::run(const double* s, double* d, hipStream_t st) const
{
    const bool remote = !plan.peers.empty();
    if (remote) {
        launch_pack(st, src_->dev, d_send, d_soff, (int)plan.send_items.size(), const_cast<double*>(s), d_sbuf, true);
        comm_->neighbor_exchange(d_sbuf, d_rbuf, plan.peers, plan.soff, plan.scount, plan.roff, plan.rcount, st);
    }
    launch_copy_items2(st, src_->dev.patches, dst_->dev.patches, d_local, (int)plan.local.size(), s, d);
    if (remote) launch_pack(st, dst_->dev, d_recv, d_roff, (int)plan.recv_items.size(), d, d_rbuf, false);
}

Level::~Level()
{
    hipFree(d_cf);
    hipFree(d_cfx);
    hipFree(d_patches);
    hipFree(d_tiles);
    hipFree(d_ftiles);
    hipFree(d_ftiles_own);
    hipFree(d_ftiles_rem);
    hipFree(d_rtiles_own);
    hipFree(d_rtiles_rem);
    hipFree(d_rtiles);
    hipFree(d_qtiles);
    hipFree(d_gtiles);
    hipFree(d_stiles);
    hipFree(d_ctiles);
    hipFree(d_local_items);
    hipFree(d_red_counter);
    hipFree(d_tile_items);
    hipFree(d_tile_item_start);
    hipFree(d_send_items);
    hipFree(d_recv_items);
    hipFree(d_send_off);
    hipFree(d_recv_off);
    hipFree(d_sendbuf);
    hipFree(d_recvbuf);
    for (int d = 0; d < 3; ++d) hipFree(dev.jg[d]);
    hipFree(dev.jinv);
    hipFree(dev.lapdiag);
}

// Tile tables of the k-marching kernels (fused 7-point sweep, 7-point operator / residual, 19-point kernels) and, on a sharded
// level, their split into tiles that read no remote ghost cell and the rest.  narrow7: the 7-point tables use the narrow lane
// classes for remainder columns.  That pays where the kernels are latency-bound -- a uniform metric, two or three streams
// (C4: 158 -> 112 ms per AMR V-cycle, c2_cartesian 134 -> 159 V-cycles/s) -- and costs where six streams are HBM-bound
// (512^3 stretched: residual 1.52 -> 2.00 ms: 64-byte row segments of six arrays), so PressureSolver::finalize asks for it
// on the depths whose metric it found uniform; the 19-point tables always use the classes.
void Level::build_march_tiles(bool narrow7)
{
    for (Tile** q : {&d_ftiles, &d_ftiles_own, &d_ftiles_rem, &d_rtiles, &d_rtiles_own, &d_rtiles_rem, &d_qtiles, &d_gtiles, &d_stiles}) {
        hipFree(*q);
        *q = nullptr;
    }
    nftiles_own = nftiles_rem = nrtiles_own = nrtiles_rem = 0;
    narrow7_ = narrow7;
    // ---- tiles of the k-marching kernels: (FT_I x FT_J) columns, k split into chunks so that the launch
    // fills the 256 CUs evenly (one 1024-thread workgroup per CU at a time); `halo` = planes a chunk reads
    // beyond its own (fused red-black sweep: 3, marching residual: 2) ----------------------------------
    // Tile columns and their lane class (Tile::pad_[1], see full19_march.hip): class 0 = 124 output columns, one region row
    // per wavefront; class 1 = 60 columns, two rows per wavefront; class 4 = 4 columns, sixteen rows per wavefront.  A box
    // is cut into 124-wide columns and its remainder into the narrow classes (128 -> 124 + 4, 64 -> 60 + 4, 512 -> 4 x 124 +
    // 4 x 4), so that a remainder column costs a sixteenth (a half) of a workgroup-march instead of a whole one.
    // `classes` off (SOMAR_NO_NARROW_TILES, the 6-/8-row A/B variants): columns of equal width, all class 0 (128 -> 2 x 64,
    // 512 -> 5 x 104).  region_rows = blockDim.y of the kernel, hrows = region rows that are halo.
    struct ColSpec { int i0, w, cls; };
    static const bool narrow_on = getenv("SOMAR_NO_NARROW_TILES") == nullptr;   // A/B switch
    auto march_tiles = [&](int FT_I, int region_rows, int hrows, double halo, int slots, bool classes, double min_gain = 1.0) {
        classes = classes && narrow_on;
        static const bool balanced = getenv("SOMAR_NO_BALANCED_TILES") == nullptr;  // A/B switch
        // min_gain: the narrow decomposition is taken only where it costs at most that fraction of the equal columns' workgroup-
        // marches (a class-4 workgroup counted as a quarter: it runs long; a class-1 one as a half).  1.0 = always.
        auto columns = [&](int n0) {
            std::vector<ColSpec> eq, v;
            {
                int w = FT_I;
                if (balanced) {
                    const int ncol = (n0 + FT_I - 1) / FT_I;
                    w = (n0 + ncol - 1) / ncol;
                    w += w & 1;
                    w = std::min(w, FT_I);
                }
                for (int i0 = 0; i0 < n0; i0 += w) eq.push_back({i0, w, 0});
            }
            if (!classes || (n0 & 1)) return eq;
            int i0 = 0, rem = n0;
            double cost = 0.0;
            while (rem >= FT_I) { v.push_back({i0, FT_I, 0}); i0 += FT_I; rem -= FT_I; cost += 1.0; }
            if (rem > 76) { v.push_back({i0, rem, 0}); rem = 0; cost += 1.0; }
            if (rem > 16) { const int w = std::min(rem, 60); v.push_back({i0, w, 1}); i0 += w; rem -= w; cost += 0.5; }
            while (rem > 0) { const int w = std::min(rem, 4); v.push_back({i0, w, 4}); i0 += w; rem -= w; cost += 0.25; }
            return cost <= min_gain * (double)eq.size() ? v : eq;
        };
        auto rows_of = [&](int cls) { return (region_rows << cls) - hrows; };
        long long cols = 0;
        int maxn2 = 1;
        for (const PatchDesc& p : hpatches) {
            for (const ColSpec& c : columns(p.n[0])) cols += (p.n[1] + rows_of(c.cls) - 1) / rows_of(c.cls);
            maxn2 = std::max(maxn2, p.n[2]);
        }
        int best = 1;
        double best_eff = -1.0;
        for (int nch = 1; nch <= 64 && (nch == 1 || maxn2 / nch >= 4); ++nch) {  // small levels: short chunks, more workgroups
            const long long blocks = cols * nch;
            const long long rounds = (blocks + slots - 1) / slots;
            const double nk = (double)maxn2 / nch;
            const double eff = (double)blocks / (double)(rounds * slots) * nk / (nk + halo);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = nch; }
        }
        std::vector<Tile> fnat;
        for (int pi = 0; pi < (int)hpatches.size(); ++pi) {
            const PatchDesc& p = hpatches[pi];
            int nk = (p.n[2] + best - 1) / best;
            nk += nk & 1;  // even chunks: a chunk never splits the two planes of a coarse cell (fused restriction)
            const std::vector<ColSpec> cs = columns(p.n[0]);
            auto add = [&](const ColSpec& c, int j0, int k0) {
                Tile t;
                std::memset(&t, 0, sizeof(t));
                t.patch = pi; t.i0 = c.i0; t.j0 = j0; t.k0 = k0;
                t.nk = std::min(nk, p.n[2] - k0);
                t.pad_[0] = c.w;
                t.pad_[1] = c.cls;
                fnat.push_back(t);
            };
            for (int k0 = 0; k0 < p.n[2]; k0 += nk) {
                // rows outside, columns inside (the natural order: a 512-wide box's 19-point colour pass takes 3.07 ms so, 3.23 ms
                // with the columns outside); a narrow column's tall tile goes where its first row is
                const size_t first = fnat.size();
                for (const ColSpec& c : cs)
                    for (int j0 = 0; j0 < p.n[1]; j0 += rows_of(c.cls)) add(c, j0, k0);
                std::stable_sort(fnat.begin() + first, fnat.end(), [](const Tile& a, const Tile& b) {
                    return a.j0 != b.j0 ? a.j0 < b.j0 : a.i0 < b.i0;
                });
            }
        }
        const int NF = (int)fnat.size();
        std::vector<Tile> perm(NF);
        const int NX = 8;
        int start[NX + 1];
        start[0] = 0;
        for (int x = 0; x < NX; ++x) start[x + 1] = start[x] + (NF - x + NX - 1) / NX;
        for (int b = 0; b < NF; ++b) perm[b] = fnat[start[b % NX] + b / NX];
        return perm;
    };
    hftiles = march_tiles(124, fused_rows(), 4, 3.0, 256, narrow7 && fused_rows() == 16, 0.85);
    d_ftiles = to_device(hftiles);
    nftiles = (int)hftiles.size();
    hrtiles = march_tiles(124, 16, 2, 2.0, 256, narrow7, 0.85);
    for (size_t q = 0; q < hrtiles.size(); ++q) hrtiles[q].pad_[2] = (int)q;   // its slot in per-tile partial sums (k_resid_march<2>)
    d_rtiles = to_device(hrtiles);
    nrtiles = (int)hrtiles.size();
    // the three-body instantiation of the fused sweep only where the table really holds a narrow tile (on class-0 tiles alone it
    // is the slower kernel: c2_cartesian 151 -> 140 V-cycles/s)
    dev.narrow7 = 0;
    for (const Tile& t : hftiles) if (t.pad_[1]) dev.narrow7 = 1;
    // every table: narrow classes where they save at least 15 % of the workgroup-marches (64-wide boxes: C5 62.8 -> 53.1 ms per
    // AMR V-cycle; 128-wide: C4 136.8 -> 111.6).  One 512-wide box, 5 equal columns against 4 + 4 narrow, is a wash or worse:
    // 19-point colour pass 3.02 -> 3.07 ms, residual 3.17 -> 3.42; c2_cartesian 151 -> 149-158 V-cycles/s depending on tile order)
    hqtiles = march_tiles(124, full_march_rows(), 2, 2.0, full_march_rows() == 6 ? 512 : 256, full_march_rows() == 8, 0.85);   // 6 rows: two workgroups per CU
    d_qtiles = to_device(hqtiles);
    nqtiles = (int)hqtiles.size();
    dev.narrowq = 0;
    for (const Tile& t : hqtiles) if (t.pad_[1]) dev.narrowq = 1;
    ngtiles = nstiles = 0;
    if (want_fused19_) {
        const std::vector<Tile> g = march_tiles(124, 8, 4, 3.0, 256, false);
        d_gtiles = to_device(g);
        ngtiles = (int)g.size();
        // shell pass: per box the two x faces (4-wide class-4 columns, 126 rows each), the two y faces (one 6-row band of class-0
        // columns each) in k-chunks of 32 planes, and the two z faces (3 planes over every class-0 tile).  The sets overlap at
        // the box edges: a cell relaxed twice gets the same value twice (its inputs are not written by the pass).
        std::vector<Tile> sh;
        const int KC = 32;
        for (int pi = 0; pi < (int)hpatches.size(); ++pi) {
            const PatchDesc& p = hpatches[pi];
            auto add = [&](int i0, int w, int cls, int j0, int k0, int nk) {
                Tile t;
                std::memset(&t, 0, sizeof(t));
                t.patch = pi; t.i0 = i0; t.j0 = j0; t.k0 = k0; t.nk = nk;
                t.pad_[0] = w; t.pad_[1] = cls;
                sh.push_back(t);
            };
            int wcol = 124;
            {
                const int ncol = (p.n[0] + 123) / 124;
                wcol = (p.n[0] + ncol - 1) / ncol;
                wcol += wcol & 1;
                wcol = std::min(wcol, 124);
            }
            for (int k0 = 0; k0 < p.n[2]; k0 += KC) {
                const int nk = std::min(KC, p.n[2] - k0);
                for (int xi : {0, std::max(0, p.n[0] - 4)})
                    for (int j0 = 0; j0 < p.n[1]; j0 += 126) add(xi, 4, 4, j0, k0, nk);
                for (int yj : {0, std::max(0, p.n[1] - 6)})
                    for (int i0 = 0; i0 < p.n[0]; i0 += wcol) add(i0, wcol, 0, yj, k0, nk);
            }
            for (int zk : {0, std::max(0, p.n[2] - 3)})
                for (int j0 = 0; j0 < p.n[1]; j0 += 6)
                    for (int i0 = 0; i0 < p.n[0]; i0 += wcol) add(i0, wcol, 0, j0, zk, std::min(3, p.n[2] - zk));
        }
        d_stiles = to_device(sh);
        nstiles = (int)sh.size();
    }

    if (!plan.peers.empty()) {
        // which marching tiles read a ghost cell that arrives from another rank?  The fused sweep reads phi two cells around
        // its columns and planes (the recomputed red ring) -- FRAME deep, as deep as the exchange fills; the operator one cell.
        std::vector<std::vector<const CopyItem*>> byDst(hpatches.size());
        for (const CopyItem& it : plan.recv_items) byDst[it.dst_patch].push_back(&it);
        auto split = [&](const std::vector<Tile>& all, int region_rows, int hrows, int halo, std::vector<Tile>& own, std::vector<Tile>& rem) {
            for (const Tile& t : all) {
                const PatchDesc& p = hpatches[t.patch];
                const int w = t.pad_[0] > 0 ? t.pad_[0] : 124;
                const int rows = (region_rows << t.pad_[1]) - hrows;
                const int lo[3] = {t.i0 - halo, t.j0 - halo, t.k0 - halo};
                const int hi[3] = {std::min(t.i0 + w, p.n[0]) - 1 + halo, std::min(t.j0 + rows, p.n[1]) - 1 + halo, t.k0 + t.nk - 1 + halo};
                bool hit = false;
                for (const CopyItem* it : byDst[t.patch]) {
                    bool ov = true;
                    for (int d = 0; d < 3; ++d)
                        ov = ov && std::max(lo[d], it->dst_lo[d]) <= std::min(hi[d], it->dst_lo[d] + it->n[d] - 1);
                    if (ov) { hit = true; break; }
                }
                (hit ? rem : own).push_back(t);
            }
        };
        std::vector<Tile> own, rem;
        split(hftiles, fused_rows(), 4, FRAME, own, rem);
        nftiles_own = (int)own.size();
        nftiles_rem = (int)rem.size();
        d_ftiles_own = to_device(own);
        d_ftiles_rem = to_device(rem);
        own.clear();
        rem.clear();
        split(hrtiles, 16, 2, 1, own, rem);
        nrtiles_own = (int)own.size();
        nrtiles_rem = (int)rem.size();
        d_rtiles_own = to_device(own);
        d_rtiles_rem = to_device(rem);
    }
}

void Level::define(const IBox& dom, const bool per[3], const double dx_[3], const int bct[3][2],
                   const std::vector<IBox>& bx, const std::vector<int>& own, Comm* c)
{
    domain = dom;
    comm = c;
    const int myrank = c ? c->rank : 0;
    for (int d = 0; d < 3; ++d) {
        periodic[d] = per[d];
        dx[d] = dx_[d];
        bc_type[d][0] = bct[d][0];
        bc_type[d][1] = bct[d][1];
    }
    boxes = bx;
    owner = own;
    if (owner.empty()) owner.assign(boxes.size(), 0);
    SOMAR_CHECK(owner.size() == boxes.size(), "owner/box count mismatch");

    // ---- patches -------------------------------------------------------------------
    local.clear();
    hpatches.clear();
    long long cursor = 0;
    valid_cells_global = 0;
    for (size_t b = 0; b < boxes.size(); ++b) {
        SOMAR_CHECK(!boxes[b].empty(), "empty box in layout");
        valid_cells_global += boxes[b].numPts();
        if (owner[b] != myrank) continue;
        PatchDesc p;
        std::memset(&p, 0, sizeof(p));
        for (int d = 0; d < 3; ++d) { p.lo[d] = boxes[b].lo[d]; p.n[d] = boxes[b].size(d); }
        p.pj = ((p.n[0] + 2 * FRAME + 1) / 2) * 2;
        p.pk = (long long)p.pj * (p.n[1] + 2 * FRAME);
        const long long elems = p.pk * (p.n[2] + 2 * FRAME);
        p.off = cursor + FRAME * p.pk + (long long)FRAME * p.pj + FRAME;
        cursor += ((elems + 31) / 32) * 32;  // 256-byte aligned patch starts
        hpatches.push_back(p);
        local.push_back((int)b);
    }
    field_elems = cursor;

    // ---- tiles, XCD-contiguous: block b runs on XCD (b % 8) under round-robin dispatch, so
    // give each XCD one contiguous slab of the natural (patch,k,j,i) tile order: neighbouring
    // tiles then share their j/k halo rows through the same 4 MiB L2. ------------------
    // Tile shape per level: big bricks (8 rows x 32 planes) keep the j/k halo re-reads of a
    // stencil sweep at (8+2)/8 * (32+2)/32 of the ideal; shrink k then j until the level still
    // yields >= 1024 workgroups (4 per CU) so coarse levels keep the chip busy.
    int tj = TILE_J_MAX, tk = TILE_K_MAX;
    auto count_tiles = [&](int tj_, int tk_) {
        long long c = 0;
        for (const PatchDesc& p : hpatches)
            c += (long long)((p.n[0] + TILE_I - 1) / TILE_I) * ((p.n[1] + tj_ - 1) / tj_) * ((p.n[2] + tk_ - 1) / tk_);
        return c;
    };
    while (count_tiles(tj, tk) < 1024 && tk > 1) tk /= 2;
    while (count_tiles(tj, tk) < 1024 && tj > 1) tj /= 2;
    dev.tile_j = tj;
    std::vector<Tile> nat;
    for (int pi = 0; pi < (int)hpatches.size(); ++pi) {
        const PatchDesc& p = hpatches[pi];
        for (int k0 = 0; k0 < p.n[2]; k0 += tk)
            for (int j0 = 0; j0 < p.n[1]; j0 += tj)
                for (int i0 = 0; i0 < p.n[0]; i0 += TILE_I) {
                    Tile t;
                    std::memset(&t, 0, sizeof(t));
                    t.patch = pi; t.i0 = i0; t.j0 = j0; t.k0 = k0;
                    t.nk = std::min(tk, p.n[2] - k0);
                    nat.push_back(t);
                }
    }
    const int N = (int)nat.size();
    htiles.assign(N, Tile());
    {
        const int NX = 8;
        int start[NX + 1];
        start[0] = 0;
        for (int x = 0; x < NX; ++x) start[x + 1] = start[x] + (N - x + NX - 1) / NX;  // #b with b%8==x
        for (int b = 0; b < N; ++b) htiles[b] = nat[start[b % NX] + b / NX];
    }

    // ---- whole-column tiles for line relaxation (one lane per (i-pair, j) column) ---------------
    {
        for (int pi = 0; pi < (int)hpatches.size(); ++pi) {
            const PatchDesc& p = hpatches[pi];
            for (int j0 = 0; j0 < p.n[1]; j0 += ctile_j)
                for (int i0 = 0; i0 < p.n[0]; i0 += TILE_I) {
                    Tile t;
                    std::memset(&t, 0, sizeof(t));
                    t.patch = pi; t.i0 = i0; t.j0 = j0; t.k0 = 0; t.nk = p.n[2];
                    hctiles.push_back(t);
                }
        }
        d_ctiles = to_device(hctiles);
        nctiles = (int)hctiles.size();
    }

    // ---- exchange plan ----------------------------------------------------------------
    // Ghost depth 2 (= FRAME): the fused red-black sweep recomputes the red ring of its neighbours
    // and therefore needs phi two deep; every other consumer reads at most one layer.
    int ghost[3];
    for (int d = 0; d < 3; ++d) ghost[d] = active[d] ? FRAME : 0;
    plan = build_exchange_plan(domain, periodic, ghost, boxes, owner, myrank);
    {
        std::vector<int> patch_of(boxes.size(), -1);
        for (int pi = 0; pi < (int)local.size(); ++pi) patch_of[local[pi]] = pi;
        for (CopyItem& it : plan.local) { it.src_patch = patch_of[it.src_patch]; it.dst_patch = patch_of[it.dst_patch]; }
        for (CopyItem& it : plan.send_items) it.src_patch = patch_of[it.src_patch];
        for (CopyItem& it : plan.recv_items) it.dst_patch = patch_of[it.dst_patch];
    }

    // ---- device tables ----------------------------------------------------------------
    d_patches = to_device(hpatches);
    d_tiles = to_device(htiles);
    d_local_items = to_device(plan.local);
    d_send_items = to_device(plan.send_items);
    d_recv_items = to_device(plan.recv_items);
    d_send_off = to_device(plan.send_itemoff);
    d_recv_off = to_device(plan.recv_itemoff);
    build_march_tiles(false);
    if (plan.send_total) SOMAR_HIP(hipMalloc(&d_sendbuf, plan.send_total * sizeof(double)));
    if (plan.recv_total) SOMAR_HIP(hipMalloc(&d_recvbuf, plan.recv_total * sizeof(double)));

    dev.tiles = d_tiles;
    dev.ntiles = N;
    dev.patches = d_patches;
    dev.npatches = (int)hpatches.size();
    SOMAR_HIP(hipMalloc(&d_red_counter, sizeof(unsigned int)));
    SOMAR_HIP(hipMemset(d_red_counter, 0, sizeof(unsigned int)));
    dev.red_counter = d_red_counter;
    // ---- pull exchange: per tile, the local copy items clipped to the tile's one-cell halo (k_gsrb_ortho / k_op_ortho) ----
    {
        static const long long pullMax = getenv("SOMAR_PULL_MAX_CELLS") ? atoll(getenv("SOMAR_PULL_MAX_CELLS")) : 262144;
        long long cells = 0;
        for (const IBox& b : boxes) cells += b.numPts();
        if (plan.peers.empty() && cells <= pullMax && N > 0) {
            std::vector<std::vector<int>> byDst(hpatches.size());
            for (int q = 0; q < (int)plan.local.size(); ++q) byDst[plan.local[q].dst_patch].push_back(q);
            std::vector<CopyItem> titems;
            std::vector<int> tstart(N + 1, 0);
            for (int b = 0; b < N; ++b) {
                const Tile& t = htiles[b];
                const PatchDesc& p = hpatches[t.patch];
                int lo[3] = {t.i0, t.j0, t.k0};
                int hi[3] = {std::min(t.i0 + TILE_I, p.n[0]) - 1, std::min(t.j0 + dev.tile_j, p.n[1]) - 1, t.k0 + t.nk - 1};
                for (int d = 0; d < 3; ++d)
                    if (active[d]) { lo[d] -= 1; hi[d] += 1; }
                tstart[b] = (int)titems.size();
                for (int q : byDst[t.patch]) {
                    const CopyItem& it = plan.local[q];
                    CopyItem c = it;
                    bool empty = false;
                    for (int d = 0; d < 3; ++d) {
                        const int a = std::max(lo[d], it.dst_lo[d]), e = std::min(hi[d], it.dst_lo[d] + it.n[d] - 1);
                        if (e < a) { empty = true; break; }
                        c.dst_lo[d] = a;
                        c.src_lo[d] = it.src_lo[d] + (a - it.dst_lo[d]);
                        c.n[d] = e - a + 1;
                    }
                    if (!empty) titems.push_back(c);
                }
            }
            tstart[N] = (int)titems.size();
            if (titems.empty()) titems.push_back(CopyItem());   // keep the table pointer non-null (one box, no ghosts to move)
            d_tile_items = to_device(titems);
            d_tile_item_start = to_device(tstart);
            dev.tile_items = d_tile_items;
            dev.tile_item_start = d_tile_item_start;
        }
    }
    {
        long long face = 0;
        for (const PatchDesc& q : hpatches) {
            const long long a = (long long)(q.n[0] + 2 * FRAME), b = (long long)(q.n[1] + 2 * FRAME), c = (long long)(q.n[2] + 2 * FRAME);
            face = std::max(face, std::max(a * b, std::max(a * c, b * c)));
        }
        dev.ghost_gy = (int)std::min<long long>(256, std::max<long long>(16, (face + 1023) / 1024));
    }
    // A Dirichlet wall on the HIGH side of direction d: the fused sweep's recomputed red ring needs, in a neighbouring
    // box's frame, the coefficient of that box's top face -- a face no box owns as a low face, so the cell-shaped
    // exchange never carries it (with a Neumann wall the face's flux is dropped and the value is never read).
    for (int d = 0; d < 3; ++d) {
        cf_faces[d].reset();
        if (active[d] && !periodic[d] && bc_type[d][1] == BC_DIRI) {
            int gh[3];
            for (int e = 0; e < 3; ++e) gh[e] = active[e] ? FRAME : 0;
            cf_faces[d].reset(new Copier);
            cf_faces[d]->define_faces(domain, periodic, *this, d, gh, comm);
        }
    }
    refresh_params();
    // the tables above went up with plain hipMemcpy (null stream); the solver's stream is non-blocking and
    // would not wait for them
    SOMAR_HIP(hipDeviceSynchronize());
}

void Level::refresh_params()
{
    StencilParams& P = dev.P;
    for (int d = 0; d < 3; ++d) {
        P.dom_lo[d] = domain.lo[d];
        P.dom_hi[d] = domain.hi[d];
        P.active[d] = active[d];
        P.periodic[d] = periodic[d] ? 1 : 0;
        P.dx[d] = dx[d];
        for (int s = 0; s < 2; ++s) {
            P.neum[d][s] = (!periodic[d] && bc_type[d][s] == BC_NEUM) ? 1 : 0;
            P.diri[d][s] = (!periodic[d] && bc_type[d][s] == BC_DIRI) ? 1 : 0;
        }
    }
    P.alpha = alpha;
    P.beta = beta;
    for (int d = 0; d < 3; ++d) { P.cf_c1[d] = cf_c1[d]; P.cf_c2[d] = cf_c2[d]; }
    dxProduct = active[2] ? dx[0] * dx[1] * dx[2] : dx[0] * dx[1];
}

double* Level::alloc_field() const
{
    double* f = nullptr;
    const long long n = field_elems > 0 ? field_elems : 1;
    SOMAR_HIP(hipMalloc(&f, n * sizeof(double)));
    SOMAR_HIP(hipMemset(f, 0, n * sizeof(double)));
    SOMAR_HIP(hipDeviceSynchronize());  // null-stream memset vs the solver's non-blocking stream
    return f;
}
void Level::free_field(double* f) { hipFree(f); }

void Level::alloc_metric()
{
    for (int d = 0; d < 3; ++d)
        if (!dev.jg[d]) dev.jg[d] = alloc_field();
    if (!dev.jinv) dev.jinv = alloc_field();
    if (!dev.lapdiag) dev.lapdiag = alloc_field();
}

void Level::exchange_remote(double* f, hipStream_t st) const
{
    if (plan.peers.empty()) return;
    launch_pack(st, dev, d_send_items, d_send_off, (int)plan.send_items.size(), f, d_sendbuf, true);
    comm->neighbor_exchange(d_sendbuf, d_recvbuf, plan.peers, plan.soff, plan.scount, plan.roff, plan.rcount, st);
    launch_pack(st, dev, d_recv_items, d_recv_off, (int)plan.recv_items.size(), f, d_recvbuf, false);
}

void Level::exchange_local(double* f, hipStream_t st) const
{
    launch_copy_items(st, dev, d_local_items, (int)plan.local.size(), f);
}

void Level::exchange(double* f, hipStream_t st) const
{
    // remote first so the wire time overlaps the local copies
    const bool remote = !plan.peers.empty();
    if (remote) {
        launch_pack(st, dev, d_send_items, d_send_off, (int)plan.send_items.size(), f, d_sendbuf, true);
        comm->neighbor_exchange(d_sendbuf, d_recvbuf, plan.peers, plan.soff, plan.scount, plan.roff, plan.rcount, st);
    }
    launch_copy_items(st, dev, d_local_items, (int)plan.local.size(), f);
    if (remote)
        launch_pack(st, dev, d_recv_items, d_recv_off, (int)plan.recv_items.size(), f, d_recvbuf, false);
}

static void copy3d(void* dst, size_t dpitchB, size_t dheight, int dx0, int dy0, int dz0, const void* src,
                   size_t spitchB, size_t sheight, int sx0, int sy0, int sz0, const int n[3], hipMemcpyKind kind,
                   hipStream_t st)
{
    hipMemcpy3DParms p;
    std::memset(&p, 0, sizeof(p));
    p.srcPtr = make_hipPitchedPtr(const_cast<void*>(src), spitchB, spitchB / sizeof(double), sheight);
    p.dstPtr = make_hipPitchedPtr(dst, dpitchB, dpitchB / sizeof(double), dheight);
    p.srcPos = make_hipPos((size_t)sx0 * sizeof(double), sy0, sz0);
    p.dstPos = make_hipPos((size_t)dx0 * sizeof(double), dy0, dz0);
    p.extent = make_hipExtent((size_t)n[0] * sizeof(double), n[1], n[2]);
    p.kind = kind;
    SOMAR_HIP(hipMemcpy3DAsync(&p, st));
}

void Level::upload(double* field, int patch, const double* host, const IBox& hostbox, const IBox& region,
                   hipStream_t st) const
{
    if (region.empty()) return;
    const PatchDesc& p = hpatches[patch];
    // device "volume": origin at the frame corner of the patch
    double* base = field + (p.off - FRAME * p.pk - (long long)FRAME * p.pj - FRAME);
    int n[3] = {region.size(0), region.size(1), region.size(2)};
    for (int d = 0; d < 3; ++d)
        SOMAR_CHECK(region.lo[d] - p.lo[d] >= -FRAME && region.hi[d] - p.lo[d] < p.n[d] + FRAME &&
                        region.lo[d] >= hostbox.lo[d] && region.hi[d] <= hostbox.hi[d],
                    "upload region outside patch frame or host box");
    copy3d(base, (size_t)p.pj * 8, (size_t)(p.pk / p.pj), region.lo[0] - p.lo[0] + FRAME,
           region.lo[1] - p.lo[1] + FRAME, region.lo[2] - p.lo[2] + FRAME, host, (size_t)hostbox.size(0) * 8,
           (size_t)hostbox.size(1), region.lo[0] - hostbox.lo[0], region.lo[1] - hostbox.lo[1],
           region.lo[2] - hostbox.lo[2], n, hipMemcpyHostToDevice, st);
}

void Level::download(const double* field, int patch, double* host, const IBox& hostbox, const IBox& region,
                     hipStream_t st) const
{
    if (region.empty()) return;
    const PatchDesc& p = hpatches[patch];
    const double* base = field + (p.off - FRAME * p.pk - (long long)FRAME * p.pj - FRAME);
    int n[3] = {region.size(0), region.size(1), region.size(2)};
    for (int d = 0; d < 3; ++d)
        SOMAR_CHECK(region.lo[d] - p.lo[d] >= -FRAME && region.hi[d] - p.lo[d] < p.n[d] + FRAME &&
                        region.lo[d] >= hostbox.lo[d] && region.hi[d] <= hostbox.hi[d],
                    "download region outside patch frame or host box");
    copy3d(host, (size_t)hostbox.size(0) * 8, (size_t)hostbox.size(1), region.lo[0] - hostbox.lo[0],
           region.lo[1] - hostbox.lo[1], region.lo[2] - hostbox.lo[2], base, (size_t)p.pj * 8,
           (size_t)(p.pk / p.pj), region.lo[0] - p.lo[0] + FRAME, region.lo[1] - p.lo[1] + FRAME,
           region.lo[2] - p.lo[2] + FRAME, n, hipMemcpyDeviceToHost, st);
}

}  // namespace somar
