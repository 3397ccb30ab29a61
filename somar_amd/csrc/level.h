// somar_amd/csrc/level.h -- device-resident level data: the MI355X counterpart of
// Chombo's DisjointBoxLayout + LevelData<FArrayBox> + Copier for ONE (AMR level, MG depth).
//
// Design (not a port): the caller's box layout is honoured semantically (it decides the MG
// depth exactly like the reference's coarsenable() tests) but physically every field of a
// level is ONE HBM allocation holding all local patches back to back, every field shares
// the same patch table, and ghost exchange is one kernel launch over a precomputed list of
// box-to-box copies (plus one packed message per neighbouring rank when the layout is
// sharded over GPUs).
#pragma once
#include <array>
#include <memory>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace somar {

struct IBox {
    int lo[3], hi[3];
    IBox() : lo{0, 0, 0}, hi{-1, -1, -1} {}
    IBox(const int* l, const int* h) { for (int d = 0; d < 3; ++d) { lo[d] = l[d]; hi[d] = h[d]; } }
    bool empty() const { return hi[0] < lo[0] || hi[1] < lo[1] || hi[2] < lo[2]; }
    int size(int d) const { return hi[d] - lo[d] + 1; }
    long long numPts() const { return empty() ? 0 : (long long)size(0) * size(1) * size(2); }
    IBox grow(const int* g) const { IBox b = *this; for (int d = 0; d < 3; ++d) { b.lo[d] -= g[d]; b.hi[d] += g[d]; } return b; }
    IBox shift(const int* s) const { IBox b = *this; for (int d = 0; d < 3; ++d) { b.lo[d] += s[d]; b.hi[d] += s[d]; } return b; }
    IBox operator&(const IBox& o) const {
        IBox b;
        for (int d = 0; d < 3; ++d) { b.lo[d] = lo[d] > o.lo[d] ? lo[d] : o.lo[d]; b.hi[d] = hi[d] < o.hi[d] ? hi[d] : o.hi[d]; }
        return b;
    }
    bool operator==(const IBox& o) const {
        for (int d = 0; d < 3; ++d) if (lo[d] != o.lo[d] || hi[d] != o.hi[d]) return false;
        return true;
    }
    // Chombo coarsen(): floor division
    static int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
    IBox coarsen(const int* r) const { IBox b; for (int d = 0; d < 3; ++d) { b.lo[d] = fdiv(lo[d], r[d]); b.hi[d] = fdiv(hi[d], r[d]); } return b; }
    IBox refine(const int* r) const { IBox b; for (int d = 0; d < 3; ++d) { b.lo[d] = lo[d] * r[d]; b.hi[d] = (hi[d] + 1) * r[d] - 1; } return b; }
};

bool coarsenable(const std::vector<IBox>& boxes, const int* r);
void box_subtract(const IBox& a, const IBox& b, std::vector<IBox>& out);
std::vector<std::array<int, 3>> periodic_shifts(const IBox& domain, const bool periodic[3]);
std::vector<IBox> uncovered(const IBox& region, const std::vector<IBox>& boxes, const IBox& domain,
                            const bool periodic[3]);

// Inter-GPU transport used by a sharded level (one process per GPU).  The default is the
// single-rank no-op; comm_rccl.cpp provides the RCCL/xGMI implementation.
struct Comm {
    int rank = 0, size = 1;
    virtual ~Comm() {}
    // in-place on DEVICE memory, enqueued on st.  op: 0 sum, 1 max
    virtual void allreduce(double* dbuf, int n, int op, hipStream_t st) { (void)dbuf; (void)n; (void)op; (void)st; }
    // the same without any single-rank shortcut (somar_comm_selftest)
    virtual void allreduce_raw(double* dbuf, int n, int op, hipStream_t st) { allreduce(dbuf, n, op, st); }
    // grouped neighbour exchange: for every peer q: send sendbuf+soff[q] (scount[q] doubles) and
    // receive into recvbuf+roff[q] (rcount[q] doubles), enqueued on st.
    virtual void neighbor_exchange(const double* sendbuf, double* recvbuf, const std::vector<int>& peers,
                                   const std::vector<long long>& soff, const std::vector<long long>& scount,
                                   const std::vector<long long>& roff, const std::vector<long long>& rcount,
                                   hipStream_t st)
    {
        (void)sendbuf; (void)recvbuf; (void)peers; (void)soff; (void)scount; (void)roff; (void)rcount; (void)st;
    }
};

// Host-side plan of the ghost exchange of one level ("Copier(grids,grids,domain,ghost,true)").
struct ExchangePlan {
    std::vector<CopyItem> local;      // both boxes on this rank
    // remote part, grouped by peer rank in ascending order
    std::vector<int> peers;
    std::vector<CopyItem> send_items, recv_items;           // concatenated per peer
    std::vector<long long> send_itemoff, recv_itemoff;      // buffer offset of each item
    std::vector<long long> soff, scount, roff, rcount;      // per peer (doubles)
    long long send_total = 0, recv_total = 0;
};

ExchangePlan build_exchange_plan(const IBox& domain, const bool periodic[3], const int ghost[3],
                                 const std::vector<IBox>& boxes, const std::vector<int>& owner, int myrank);

class Level {
public:
    // layout
    IBox domain;
    bool periodic[3] = {false, false, false};
    double dx[3] = {1, 1, 1};
    int active[3] = {1, 1, 1};
    int bc_type[3][2];
    std::vector<IBox> boxes;   // the whole DisjointBoxLayout, in the caller's order
    std::vector<int> owner;    // rank of each box
    std::vector<int> local;    // global box index of each local patch
    std::vector<PatchDesc> hpatches;
    std::vector<Tile> htiles;
    void build_march_tiles(bool narrow7);   // (re)builds the marching kernels' tile tables below
    bool narrow7_ = false;
    std::vector<Tile> hftiles;   // tiles of the fused red-black sweep (gsrb_fused.hip)
    Tile* d_ftiles = nullptr;
    int nftiles = 0;
    // sharded levels: hftiles split into the tiles that read no ghost cell another rank fills (they may run while the
    // messages are in flight) and the rest (fused_overlap in solver.cpp); both lists keep the XCD-contiguous order
    Tile* d_ftiles_own = nullptr;
    Tile* d_ftiles_rem = nullptr;
    int nftiles_own = 0, nftiles_rem = 0;
    Tile* d_rtiles_own = nullptr;   // the same split of the marching operator / residual's tiles (they read phi one cell around)
    Tile* d_rtiles_rem = nullptr;
    int nrtiles_own = 0, nrtiles_rem = 0;
    std::vector<Tile> hrtiles;   // tiles of the k-marching operator/residual (resid_march.hip): 124 x 14 columns
    Tile* d_rtiles = nullptr;
    int nrtiles = 0;
    std::vector<Tile> hqtiles;   // tiles of the k-marching 19-point kernels (full19_march.hip): 124 x (rows - 2) columns
    Tile* d_qtiles = nullptr;
    int nqtiles = 0;
    // fused red+black 19-point sweep (full19_fused.hip), built when PressureSolver asks (want_fused19_): its 124 x 4 column tiles
    // and the tiles of the shell pass that follows it (the three outer cell layers of every box)
    bool want_fused19_ = false;
    Tile* d_gtiles = nullptr;
    Tile* d_stiles = nullptr;
    int ngtiles = 0, nstiles = 0;
    std::vector<Tile> hctiles;   // whole-column tiles (line relaxation): 128 x ctile_j columns, all of k
    Tile* d_ctiles = nullptr;
    int nctiles = 0, ctile_j = 2;
    long long field_elems = 0;
    long long valid_cells_global = 0;
    ExchangePlan plan;
    Comm* comm = nullptr;

    // device tables
    PatchDesc* d_patches = nullptr;
    Tile* d_tiles = nullptr;
    CopyItem* d_local_items = nullptr;
    // pull exchange tables (LevelDev::tile_items): built for single-rank levels of at most PULL_MAX_CELLS cells
    unsigned int* d_red_counter = nullptr;
    CopyItem* d_tile_items = nullptr;
    int* d_tile_item_start = nullptr;
    bool pull_ready() const { return d_tile_item_start != nullptr; }
    CopyItem* d_send_items = nullptr;
    CopyItem* d_recv_items = nullptr;
    long long* d_send_off = nullptr;
    long long* d_recv_off = nullptr;
    double* d_sendbuf = nullptr;
    double* d_recvbuf = nullptr;
    LevelDev dev;  // kernel view (tables + metric planes)

    // operator state
    double alpha = 0.0, beta = 1.0;
    int mgCrseRefRatio[3] = {1, 1, 1};
    bool hasCoarser = false;
    bool zeroAvg = false;
    double dxProduct = 1.0;

    // coarse-fine ghost cells (only on levels that sit on a coarser AMR level): the cells of the 1-deep
    // ghost layer that lie inside the (periodically extended) domain and are covered by no box of this level
    std::vector<CFCell> hcf;
    CFCell* d_cf = nullptr;
    int ncf = 0;
    double cf_c1[3] = {0, 0, 0}, cf_c2[3] = {0, 0, 0}, cf_fac[3] = {0, 0, 0};
    // The fused red-black sweep on a level with CF boundaries needs (a) every box face to be CF either entirely
    // or not at all (flag bits in PatchDesc.cf), boxes at least two cells wide, and (b) the pre-sweep CF values
    // also in the EDGE ghosts that the recomputed red ring of a neighbouring box reads -- hcfx = hcf + those.  A
    // re-entrant corner of the refined region (one edge ghost wanted with two different values) rules it out.
    std::vector<CFCell> hcfx;
    CFCell* d_cfx = nullptr;
    int ncfx = 0;
    bool cf_fusable = true;
    std::unique_ptr<class Copier> cf_faces[3];  // high-face coefficients of neighbouring boxes (Copier::define_faces): CF levels, Dirichlet hi walls
    void cf_homog_ext(double* phi, hipStream_t st) const
    {
        launch_cf_homog(st, d_cfx, ncfx, phi, cf_c1, cf_c2, cf_fac);
    }
    // builds hcf/d_cf and the interpolation weights for a coarser-level spacing dxCrse
    void define_cf(const double dxCrse[3]);
    void cf_homog(double* phi, hipStream_t st) const
    {
        launch_cf_homog(st, d_cf, ncf, phi, cf_c1, cf_c2, cf_fac);
    }

    Level() {}
    ~Level();
    Level(const Level&) = delete;
    Level& operator=(const Level&) = delete;

    // Build tables for a layout.  bc_type codes: BC_NEUM / BC_DIRI per non-periodic side.
    void define(const IBox& dom, const bool per[3], const double dx_[3], const int bct[3][2],
                const std::vector<IBox>& bx, const std::vector<int>& own, Comm* c);
    // allocate coefficient planes (zero-filled) and fill dev view
    void alloc_metric();
    void refresh_params();

    double* alloc_field() const;        // zero-initialised, field_elems doubles
    static void free_field(double* f);

    // ghost exchange of one field (faces, edges and corners, 1 cell deep in active dirs)
    void exchange(double* f, hipStream_t st) const;
    // its two halves: what other ranks send (pack, one grouped send / receive per neighbour, unpack) and the box-to-box
    // copies inside this rank; they write disjoint ghost cells and may run on different streams
    void exchange_remote(double* f, hipStream_t st) const;
    void exchange_local(double* f, hipStream_t st) const;

    // host<->device transfer of one patch in Chombo FRA layout.  `hostbox` is the box the host
    // array is defined on (valid grown by the caller's ghosts, or a face box); `region` is
    // what to move.  comp selects a component of a multi-comp host FAB.
    void upload(double* field, int patch, const double* host, const IBox& hostbox, const IBox& region,
                hipStream_t st) const;
    void download(const double* field, int patch, double* host, const IBox& hostbox, const IBox& region,
                  hipStream_t st) const;

    int npatches() const { return (int)hpatches.size(); }
};

// Copier between two layouts of one index space: valid cells of `src` boxes (and their periodic images) into
// the cells of grow(dst box, ghost).  define_allgather: every rank receives EVERY box of `src` (grown by `grow`
// cells) into a replicated layout `dst` that holds all boxes locally (coarse-level agglomeration).
class Copier {
public:
    ~Copier();
    // ring_only: only the part of each destination box's grown region that lies OUTSIDE the box (its ghost ring)
    void define(const IBox& domain, const bool periodic[3], const Level& src, const Level& dst, const int ghost[3],
                Comm* comm, bool ring_only = false);
    void define_allgather(const Level& src, const Level& dst, int grow, Comm* comm);
    void define_faces(const IBox& domain, const bool periodic[3], const Level& L, int dir, const int ghost[3],
                      Comm* comm);
    void run(const double* s, double* d, hipStream_t st) const;
    ExchangePlan plan;

private:
    void upload_tables();
    const Level* src_ = nullptr;
    const Level* dst_ = nullptr;
    Comm* comm_ = nullptr;
    CopyItem *d_local = nullptr, *d_send = nullptr, *d_recv = nullptr;
    long long *d_soff = nullptr, *d_roff = nullptr;
    double *d_sbuf = nullptr, *d_rbuf = nullptr;
};

ExchangePlan build_copy_plan(const IBox& domain, const bool periodic[3], const std::vector<IBox>& srcBoxes,
                             const std::vector<int>& srcOwner, const std::vector<IBox>& dstBoxes,
                             const std::vector<int>& dstOwner, const int ghost[3], int myrank, bool ring_only = false);
// every rank gets every box (grown): items carry GLOBAL box indices on both sides
ExchangePlan build_allgather_plan(const std::vector<IBox>& boxes, const std::vector<int>& owner, int grow, int myrank,
                                  int nranks);

}  // namespace somar
