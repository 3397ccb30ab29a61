// somar_amd/csrc/kernels.h -- launcher prototypes for kernels.hip
#pragma once
#include "common.h"

namespace somar {

// Device-side view of one (AMR level, MG depth): tables + coefficient planes in HBM.
struct LevelDev {
    const Tile* tiles = nullptr;        // XCD-contiguous order
    int ntiles = 0;
    int tile_j = 4;                     // blockDim.y of every tile kernel on this level
    // pull exchange (small single-rank levels): per tile, the local ghost copies that land in the tile's one-cell halo; a
    // stencil kernel refreshes exactly the ghosts it is about to read, so the separate copy launch disappears (null: not built)
    const CopyItem* tile_items = nullptr;
    const int* tile_item_start = nullptr;
    unsigned int* red_counter = nullptr;   // arrival counter of the single-launch tree reduction (k_reduce_valid's last workgroup)
    int ghost_gy = 16;                  // workgroups per ghost op (k_ghost_ops): the largest box face / 1024, within [16, 256]
    const PatchDesc* patches = nullptr;
    int npatches = 0;
    double* jg[3] = {nullptr, nullptr, nullptr};  // J g^{aa} on a-faces
    double* jinv = nullptr;
    double* lapdiag = nullptr;
    // non-diagonal metric: all components J g^{ab} on a-faces, jgf[a][b]; jgf[a][a] aliases jg[a]
    double* jgf[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    StencilParams P;
    int narrow7 = 0;                    // the 7-point marching kernels' tile tables hold narrow lane classes (Level::build_march_tiles)
    int narrowq = 0;                    // the 19-point marching kernels' tile table does
};

// operands of MAPPEDGETFLUX with a non-diagonal metric, evaluated where a flux register needs it (amr_kernels.hip: reg_flux19)
struct FullFlux { const double* psi = nullptr; const double* J[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}; double dxi[3] = {0, 0, 0}; };

// 19-point path (full19.hip)
// redirect: a read of psi that lands INSIDE the box's valid region takes phi instead (psi is then maintained in the
// boxes' frames only -- the frame-only programs of the marching 19-point kernels, full19_march.hip)
void launch_ghost_ops(hipStream_t st, const LevelDev& L, const GhostOp* ops, int nops, double* phi, double* psi,
                      bool bc_homog = true, bool redirect = false);
// a whole ghost program in one launch, one workgroup per box (ops sorted by box, then stage = GhostOp::pad_; box b owns
// box_ops[box_first[b] .. box_first[b + 1])); copy_all: psi := phi on every box grown by one cell first
void launch_ghost_program(hipStream_t st, const LevelDev& L, const GhostOp* box_ops, const int* box_first, double* phi,
                          double* psi, bool bc_homog, bool redirect, bool copy_all);
// k-marching 19-point operator / residual (mode 0: out = rhs - L[phi], 1: out = L[phi]) and one GSRB colour pass
// (out = phi with the colour's cells relaxed; out != phi); psi valid in the one-cell frames only (full19_march.hip)
void launch_full_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out, const double* phi,
                       const double* psi, const double* rhs, int mode);
void launch_gsrb_full_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out,
                            const double* phi, const double* psi, const double* rhs, int color);
int full_march_rows();
// red + black in one launch for the cells three layers inside their box (full19_fused.hip), then -- after the between-colour
// ghost work on `out` -- the black cells of the outer layers in place (full19_march.hip, MODE 3)
void launch_full_fused(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out, const double* phi,
                       const double* psi, const double* rhs);
void launch_gsrb_full_shell(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* phi, const double* phi_in,
                            const double* psi, const double* rhs);
void launch_flux_full(hipStream_t st, const LevelDev& L, double* const out[3], const double* phi, const double* psi);
void launch_op_full(hipStream_t st, const LevelDev& L, double* out, const double* phi, const double* psi,
                    const double* rhs, int mode);
void launch_gsrb_full(hipStream_t st, const LevelDev& L, double* phi, const double* psi, const double* rhs, int color);

// loose: 0 LevelGSRB pass; 1 / 2: interior / box-shell phase of LooseGSRB (see k_gsrb_ortho)
// pull: refresh the ghosts the pass reads inside the kernel (LevelDev::tile_items; LevelGSRB passes only)
void launch_gsrb_ortho(hipStream_t st, const LevelDev& L, double* phi, const double* rhs, int color, int loose = 0,
                       bool pull = false);
// one full red+black sweep, phi_in -> phi_out (gsrb_fused.hip); needs phi_in ghosts 2 deep and
// rhs / Jg / Jinv ghosts 1 deep wherever a neighbouring box or periodic image exists
// in_mode 0: plain; 1: phi_in is taken to be all zeros and is not read; 2: phi_in is read as
// (value - sums[0]/sums[1]) (deferred mean removal of the zero-average prolongation); 3: phi_in is read as
// value + crse(i / r) (prolongation folded in; C = coarse level, crse exchanged one deep); 4: 3 and 2 together
void launch_gsrb_fused(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* phi_out,
                       const double* phi_in, const double* rhs, int in_mode = 0, const double* sums = nullptr,
                       const LevelDev* C = nullptr, const double* crse = nullptr, const int* r = nullptr);
int fused_rows();
// k-marching operator/residual of a large level (resid_march.hip); mode 0: out = rhs - L[phi], 1: out = L[phi]
void launch_resid_march(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& L, double* out,
                        const double* phi, const double* rhs, int mode);
void launch_resid_restrict(hipStream_t st, const Tile* tiles, int ntiles, const LevelDev& F, const LevelDev& C,
                           double* crse, const double* phi, const double* rhs, const int r[3], double dxProduct = 0.0,
                           double* volsum = nullptr);
// crse(ic) = sum over the children of ic of dxProduct / fjinv(child): the volume a coarse value is spread over
void launch_child_volume(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const int r[3],
                         double dxProduct);
void launch_sum_partials(hipStream_t st, const double* partials, int n, double* out);  // out[0] = sum (fixed order)
// out[0] = a[0] + b[0]; out[1] = c[0]
void launch_combine_sums(hipStream_t st, double* out, const double* a, const double* b, const double* c);
// one colour of vertical-line GSRB (line_gsrb.hip); ctiles = whole-column tiles (k0 = 0, nk = n2)
// LineGSRBIter2D (space_dim 2: lines along y); maxN0: the widest local box in x
void launch_line_gsrb_2d(hipStream_t st, const LevelDev& L, int maxN0, double* phi, const double* rhs, double* dmod, int color,
                         const double* psi);
void launch_line_gsrb_ortho(hipStream_t st, const Tile* ctiles, int nctiles, int tile_j, const LevelDev& L,
                            double* phi, const double* rhs, double* dmod, int color, const double* psi = nullptr);
void launch_op_ortho(hipStream_t st, const LevelDev& L, double* out, const double* phi, const double* rhs, int mode,
                     bool pull = false);
void launch_lapdiag(hipStream_t st, const LevelDev& L);
void launch_diag(hipStream_t st, const LevelDev& L, double* phi, const double* r, int mode);
void launch_restrict(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine,
                     const int r[3]);
void launch_prolong(hipStream_t st, const LevelDev& F, const LevelDev& C, double* fine, const double* crse,
                    const int r[3], bool zeroAvg, double dxProduct, double* partials, double* sums,
                    long long fieldElems, bool ordered = false);
void launch_sub_mean(hipStream_t st, double* f, long long n, const double* sums);
void launch_avg_harmonic(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine,
                         const int r[3]);
void launch_avg_face(hipStream_t st, const LevelDev& C, const LevelDev& F, int patch, const int cn[3], double* crse,
                     const double* fine, int dir, const int r[3]);
void launch_copy_items(hipStream_t st, const LevelDev& L, const CopyItem* items, int nitems, double* f);
void launch_pack(hipStream_t st, const LevelDev& L, const CopyItem* items, const long long* bufoff, int nitems,
                 double* f, double* buf, bool pack);
void launch_set(hipStream_t st, double* a, long long n, double v);
// n device values -> coherent host memory, then the sequence number (system-scope release)
void launch_publish(hipStream_t st, const double* src, int n, double* host_dst, unsigned long long* host_seq,
                    unsigned long long seq);
void launch_copy(hipStream_t st, double* d, const double* s, long long n);
void launch_incr(hipStream_t st, double* y, const double* x, double a, long long n);
void launch_incr2(hipStream_t st, double* y1, const double* x1, double a1, double* y2, const double* x2, double a2, long long n);
void launch_bicg_p(hipStream_t st, double* p, const double* v, const double* r, double beta, double bw, long long n);  // p = (p*beta + bw*v) + r
void launch_incr_copy(hipStream_t st, double* y, double* x, double a, long long n);  // y += a*x; x = y
void launch_scale(hipStream_t st, double* y, double a, long long n);
void launch_mul(hipStream_t st, double* y, const double* x, long long n);   // y *= x elementwise
void launch_axby(hipStream_t st, double* z, const double* x, const double* y, double a, double b, long long n);
// where a reduction's result is also stored for the host: coherent host memory + a sequence number the host spins on
// (PressureSolver::fetch_scalars); fused into the reduction's last kernel it saves the separate one-thread publish launch
struct ScalarPublish { double* host_dst; unsigned long long* host_seq; unsigned long long seq; };
// the whole BiCGStab bottom solve of a tiny level in one single-workgroup launch (k_tiny_bicgstab, kernels.hip); the caller
// fills phi, rhs, w (r, r~, e, p, p~, s~, t, v), the parameters, info (device, 2 doubles: iterations, exit code) and pub
struct TinyBicg {
    const Tile* tiles; int ntiles; int tile_j;
    const PatchDesc* patches; int npatches;
    const CopyItem* items; int nitems;
    long long field_elems;
    const double* jg[3]; const double* jinv; const double* lapd;
    StencilParams P;
    double* phi; const double* rhs;
    double* w[8];
    int imax, numRestarts, normType, precondIters;
    double eps, reps, hang, small, metric;
    double* info;
    ScalarPublish pub;
};
void launch_tiny_bicgstab(hipStream_t st, const LevelDev& L, const CopyItem* items, int nitems, long long field_elems, TinyBicg A);
// the BiCGStab bottom solve of a level of up to BOX_MAX_WG boxes of up to BOX_MAX_CELLS cells as ONE persistent launch, one
// workgroup per box (k_box_bicgstab, kernels.hip): device-wide barriers where the launch-by-launch path has kernel boundaries.
// nb: for every valid cell (boxes back to back, Fortran order inside a box; box b starts at cstart[b]) the field offsets of the
// six cells its stencil reads (x-, x+, y-, y+, z-, z+): the neighbour itself, or -- across a box edge -- the valid cell the ghost
// exchange would have copied from (built on the host from the level's exchange plan), so no ghost cell is ever filled.
constexpr int BOX_MAX_WG = 128, BOX_MAX_CELLS = 2048;
// 19-point variant: cells of a box grown by one cell; per box and ghost program the entries (one per written cell), stages and
// cross-term Neumann ghost cells the kernel keeps in LDS
constexpr int BOX_FAB_MAX = 2048, BOX_MAX_ENT = 2048, BOX_MAX_STAGES = 63, BOX_MAX_NEUM = 512;
// One written cell of a ghost program, compiled on the host from the box's GhostOps (PressureSolver::build_box_tables): indices
// into the LDS copy of the box grown by one cell.  kind 0: dst := src (the other field); 1 / 2 / 3: extrapolation of order
// 0 / 1 / 2 from s1, s2, s3; 4: the cross-term Neumann ghost (flags: direction and side; nslot: its coefficient triple).
struct alignas(4) BoxProgEntry {
    unsigned short dst, s1, s2, s3;
    unsigned char kind, flags;   // flags: bit 0 dst field (1 = psi), bit 1 src field, bits 2-3 direction, bit 4 high side
    unsigned short nslot;
};
struct BoxBicg {
    const PatchDesc* patches; int npatches;
    const int* nb; const int* cstart;
    const double* jg[3]; const double* jinv; const double* lapd;
    StencilParams P;
    double* phi; const double* rhs;
    double* z[2];        // p~ and s~ in the level's layout: the only fields a workgroup reads outside its own box
    int imax, numRestarts, normType, precondIters;
    double eps, reps, hang, small, metric;
    double* sums;        // 4 * BOX_MAX_WG doubles: per-box partial results of the running reduction(s), double-buffered
    unsigned* sync;      // BOX_MAX_WG + 1: per workgroup the number of the last barrier it reached; abort flag (a barrier gave up); zeroed per launch
    // 19-point variant (non-diagonal metric, 3-D, boxes of at most 256 cells): all J g^{ab}; for every cell of box b grown by one
    // cell (x fastest, box b starts at fab_start[b]) the field offset its value comes from -- the cell itself, the valid cell
    // the ghost exchange copies there, or -1 (a ghost beyond a wall: the program fills it); the ghost programs [0] operator,
    // [1] smoother compiled into per-cell entries: box b owns ent[w][ent_first[w][b] .. ent_first[w][b + 1]), sorted by stage;
    // stg[w][stg_first[w][b] ..]: where each of its stages starts (relative to the box's first entry; one more than stages);
    // nfg[w][nfg_first[w][b] ..]: the boundary face (field offset) of each of its cross-term Neumann ghost cells
    int full;
    const double* jgf[3][3];
    const int* fab_src; const int* fab_start;
    const BoxProgEntry* ent[2]; const int* ent_first[2];
    const int* stg[2]; const int* stg_first[2];
    const int* nfg[2]; const int* nfg_first[2];
    long long* dbg;      // optional (SOMAR_BOX_TIMING=1): workgroup 0's clock ticks in {staging loads, ghost program, stencil, barrier, sums, count}
    int serial;          // sums in the reference's serial order (levels of at most ordered_max cells) or by a fixed tree per box
    double* info;        // device: iterations, exit code
    ScalarPublish pub;
};
void launch_box_bicgstab(hipStream_t st, const LevelDev& L, int max_box_cells, BoxBicg A);
// streaming probe (diagnostics): kind 0 copy, 1 read, 2 six reads + one write; in6: six arrays of `cells` doubles
void launch_stream_probe(hipStream_t st, int kind, int workgroups, double* const* in6, double* out, long long cells);
// (min, max) per (patch, k-chunk) of a over the valid cells (dir < 0) or valid dir-faces: out[2 * npatches * MM_CH] (device)
constexpr int MM_CH = 64;
void launch_minmax_valid(hipStream_t st, const LevelDev& L, const double* a, int dir, double* out);
// mode 0: sum a*b, 1: max|a|, 2: sum|a|, 3: signed max a  -> out[0] (device)
// ordered: reference-ordered serial sum (modes 0 and 2; meant for small levels, see k_reduce_ordered)
void launch_reduce(hipStream_t st, const LevelDev& L, const double* a, const double* b, int mode, double* partials,
                   double* out, bool ordered = false, const ScalarPublish* pub = nullptr);
// sharded small level: the rank's per-cell terms into their slot of the serial sequence (mode 0: a*b, 2: |a|, 6: X = dxProduct/b*a,
// Y = dxProduct/b), then -- after a sum-allreduce of X (and Y) -- the walk in the reference's order (k_reduce_ordered_flat)
void launch_ord_fill(hipStream_t st, const LevelDev& L, const long long* start, const double* a, const double* b,
                     int mode, double dxProduct, double* X, double* Y);
void launch_reduce_ordered_flat(hipStream_t st, int nboxes, const long long* box_start, const double* X, const double* Y,
                                int mode, double* out);
void launch_div_mac(hipStream_t st, const LevelDev& L, double* out, const double* u0, const double* u1,
                    const double* u2, double dt);
void launch_jgup_from_dxdxi(hipStream_t st, long long n, int mu, const double* const x9[9], const double* J, double scale,
                            double* out);
void launch_altered_jgup(hipStream_t st, long long n, double* dest, const double* nsq, const double* dmu,
                         const double* dnu, const double* ix, const double* jy, const double* iy, const double* jx,
                         const double* gup, const double* J, double theta, double coriolisF, bool offdiag);
void launch_face_axpy(hipStream_t st, const LevelDev& L, double* const vel[3], double* const grad[3], double s);
void launch_mac_correct(hipStream_t st, const LevelDev& L, double* const vel[3], const double* phi, double dtScale);
// cell-centred level projection: CellToEdge (+ zero normal flux on walls), EdgeToCell + correction (null cc[a] = skip a)
void launch_face_wall(hipStream_t st, const LevelDev& L, double* const edge[3]);  // zero normal flux on solid walls (null = skip)
// coordinate maps' metric producers (maps.hip): kind 1 cylindrical (diagonal), 2 bathymetric (non-diagonal, nodal depth)
void launch_map_metric(hipStream_t st, const LevelDev& L, int kind, const double dXi[3], const double Lc[3],
                       const double* d_depth, const int dlo[2], const int dn[2], bool diagonal, const double* domLen = nullptr);
void launch_face_bc(hipStream_t st, const LevelDev& L, double* const edge[3], const int kind[6], const double value[6]);
void launch_cell_to_edge(hipStream_t st, const LevelDev& L, double* const edge[3], double* const cc[3], bool wall);
void launch_cc_correct(hipStream_t st, const LevelDev& L, double* const cc[3], const double* phi, double dtScale);
void launch_edge_to_cell_axpy(hipStream_t st, const LevelDev& L, double* const cc[3], double* const grad[3], double dtScale);
void launch_cf_homog(hipStream_t st, const CFCell* cells, int n, double* phi, const double c1[3], const double c2[3],
                     const double fac[3]);
void launch_copy_items2(hipStream_t st, const PatchDesc* spatches, const PatchDesc* dpatches, const CopyItem* items,
                        int nitems, const double* src, double* dst);
void launch_fill_hash(hipStream_t st, const LevelDev& L, double* f, unsigned long long seed);

}  // namespace somar
