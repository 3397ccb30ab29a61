// somar_amd/csrc/solver.h -- host-side multigrid driver (C++), the MI355X counterpart of
//   MappedAMRPoissonOpFactory::MGnewOp      calculus/AMRElliptic/MappedAMRPoissonOpFactory.cpp:363-702
//   MappedMultiGrid<T>::define/init/cycle   calculus/AMRElliptic/MappedMultiGrid.H:328-434, 555-653
//   MappedAMRMultiGrid<T>::solveNoInitResid calculus/AMRElliptic/MappedAMRMultiGrid.H:979-1183
//   MappedAMRPoissonOp (level operator)     calculus/AMRElliptic/MappedAMRPoissonOp.cpp
//   LevelGSRB / Jacobi                      calculus/AMRElliptic/RelaxationMethods/GSRB.cpp:58-98, Jacobi.cpp:54-90
//   Chombo 3.1 BiCGStabSolver (EXTERNAL)    restated from its published algorithm
// All level data stay resident in HBM for the life of the solver; the host only sequences
// kernel launches on one HIP stream and reads back the few scalars the stopping tests need.
#pragma once
#include <memory>
#include <vector>

#include "level.h"

namespace somar {

enum { RELAX_JACOBI = 0, RELAX_LEVEL_GSRB = 1, RELAX_LOOSE_GSRB = 2, RELAX_LINE_GSRB = 3 };  // ProblemContext.H:322-340
enum { PRECOND_NONE = -1, PRECOND_DIAG_RELAX = 0, PRECOND_DIAG_LINE_RELAX = 1 };

// AMRPressureSolver::setAMRMGParameters / setBottomParameters (projection/AMRPressureSolver.H:53-77);
// defaults utils/ProblemContext.cpp:1147-1236
struct SolverParams {
    int imin = 5, imax = 20;
    double eps = 1e-6, hang = 1e-15, normThresh = 1e-30;
    int num_smooth_down = 2, num_smooth_up = 2, num_smooth_bottom = 2, num_smooth_precond = 2;
    int numMG = 1, maxDepth = -1, precondMode = PRECOND_DIAG_RELAX, relaxMode = RELAX_LEVEL_GSRB;
    int verbosity = 0;
    int spaceDim = 3;  // 2: the reference built with CH_SPACEDIM = 2 (boxes one cell thick in z, which is inactive)
    int bottom_imax = 80, bottom_numRestarts = 5, bottom_normType = 2, bottom_verbosity = 0;
    double bottom_eps = 1e-6, bottom_reps = 1e-12, bottom_hang = 1e-15, bottom_small = 1e-30;
};

struct SolveStats {
    int iters = 0;
    int exitStatus = 0;         // !goRedu + 2*!goIter + 4*!goHang + 8*!goNorm (MappedAMRMultiGrid.H:1148)
    int status = 0;             // 0 ok, 1 "kaboom", 2 "solver blew up"
    double initial_rnorm = 0, final_rnorm = 0;
    std::vector<double> history;  // max-norm residual after each V-cycle (history[0] = initial)
    int bottom_iters_last = 0, bottom_exit_last = 0;
};

class PressureSolver {
public:
    // shared: run on the caller's stream (the levels of an AMR hierarchy share one) instead of an own one
    explicit PressureSolver(Comm* comm = nullptr, hipStream_t shared = nullptr);
    ~PressureSolver();

    // ---- definition (MappedAMRPoissonOpFactory::define + MappedAMRMultiGrid::define) -----
    void define(const IBox& domain, const bool periodic[3], const double dx[3], const int bc_type[3][2],
                const std::vector<IBox>& boxes, const std::vector<int>& owner, double alpha, double beta,
                const SolverParams& prm, const double* dxCrse = nullptr);
    bool has_cf() const { return hasCF_; }
    const double* dx_crse() const { return hasCF_ ? dxCrse_ : nullptr; }
    // Diagonal metric of one LOCAL patch in Chombo FRA layout: Jg_aa on faces(valid,a), Jinv on valid.
    void set_metric_ortho(int patch, const double* jg0, const double* jg1, const double* jg2, const double* jinv);
    // Non-diagonal metric of one LOCAL patch: jgD = J g^{Db} on faces(valid, D), 3 components, component slowest
    // (LevelGeometry::getFCJgup's FluxBox layout).  Switches the solver to the 19-point kernels (full19.hip).
    void set_metric_full(int patch, const double* jg0, const double* jg1, const double* jg2, const double* jinv);
    // a Cartesian map's constants (CartesianMap::fill_Jgup / fill_Jinv) into every local patch, on the device
    void set_metric_uniform(const double c4[4]);
    bool is_full() const { return full_; }
    // before finalize: switch this solver to the non-diagonal path with every cross plane allocated (zero) -- for callers that
    // fill the metric planes of level(0).dev.jgf on the device themselves (the leptic solver's J-scaled and flat operators)
    void make_full();
    // ghost-op lists built with the reference's helpers for other callers (leptic): which 0 = extrapAllGhosts(phi, order 2)
    // (ExtrapolationUtils.cpp:388-420), 1 = ExtrapolateFaceAndCopy(phi, phi, FAB & domain, vertical dir, lo then hi, order 2)
    // (levelVertHorizGradient, LevelLepticSolver.cpp:1107-1176); run on depth 0, in place on phi
    void run_aux_program(int which, double* phi);
    void set_amr_member() { amr_member_ = true; }  // a level of an AMRSolver hierarchy (whose reflux tables carry beta)
    void finalize();  // builds the semicoarsened hierarchy, coarse metrics, lapDiag, null-space probes

    // ---- data movement across the boundary (host FABs, caller-owned) ---------------------
    void upload_phi(int patch, const double* host, const int ghost[3]);
    void upload_rhs(int patch, const double* host, const int ghost[3]);
    void download_phi(int patch, double* host, const int ghost[3]);
    void download_field(const double* field, int depth, int patch, double* host, const int ghost[3]);

    // ---- AMREllipticSolver::solve on the resident phi/rhs (AMREllipticSolver.H:33-48) -----
    void solve(bool zeroPhi, bool forceHomogeneous, SolveStats& st);

    // ---- pieces, exposed for benchmarks and kernel-level parity tests -----------------------
    // MG depths of the whole hierarchy (the agglomerated tail included)
    int depth() const { return coarse_ ? agglom_depth_ + coarse_->depth() : (int)lev.size(); }
    // Coarse-level agglomeration (sharded runs only): from the first depth with at most SOMAR_AGGLOM_CELLS cells
    // on, EVERY rank holds all boxes and runs the rest of the cycle redundantly -- no halo exchange, no
    // allreduce and no host round trip per BiCGStab scalar where the work is microseconds and the wire is not.
    // One allgather of the restricted residual goes down; nothing comes back (every rank already has the
    // correction).  Same arithmetic, same order (in fact the serial order), so results do not change.
    int agglom_depth() const { return coarse_ ? agglom_depth_ : -1; }
    PressureSolver* coarse_solver() { return coarse_.get(); }
    // depth d of the whole hierarchy; depths inside the agglomerated tail (d > agglom_depth()) are the replicated
    // solver's, depth == agglom_depth() is the sharded landing layout
    Level& level(int d) { return (coarse_ && d > agglom_depth_) ? coarse_->level(d - agglom_depth_) : *lev[d]; }
    std::array<int, 3> ref_ratio(int d) const
    {
        return (coarse_ && d >= agglom_depth_) ? coarse_->ref_ratio(d - agglom_depth_) : mgRefRatios.at(d);
    }
    double* phi() { return f_phi; }
    double* rhs() { return f_rhs; }
    double* work(int which);  // 0 uberResidual 1 uberCorrection 2 bestPhi
    double* field(int depth, int which);  // SOMAR_F_* handle -> device pointer (nullptr if absent)
    // e_zero: e is to be taken as all zeros, whatever it holds (saves the memset and the first sweep's read)
    // e_shift: device pointer to (sum, volume): e is to be read as e - sum/volume (deferred mean removal); only
    //          legal when fused_relax(d, iters) holds
    // e_plus: (coarse level, coarse correction): e is to be read as e + coarse(i / r) -- the prolongation folded
    //         into the first sweep; the coarse field must have been exchanged
    void relax(int d, double* e, const double* res, int iters, bool e_zero = false, const double* e_shift = nullptr,
               const Level* e_plus_level = nullptr, const double* e_plus = nullptr);
    bool fused_relax(int d, int iters) const;
    static bool no_cf_fused_(const Level& L)  // A/B switch: SOMAR_NO_CF_FUSED=1 keeps CF levels on the two-pass kernel
    {
        static const bool off = getenv("SOMAR_NO_CF_FUSED") != nullptr;
        return off && L.ncf > 0;
    }
    // homogeneous: physical BC values taken as zero (only Dirichlet sides can carry a value here)
    void residual(int d, double* out, double* phi, const double* rhs, bool homogeneous = true);   // homogeneous CF ghosts, then residual_i
    void apply_op(int d, double* out, double* phi, bool homogeneous = true);
    void residual_i(int d, double* out, double* phi, const double* rhs, bool homogeneous = true); // residualI: CF ghosts as they are
    // residualI of depth 0 and its J-weighted average (MAPPEDAVERAGE2) onto the layout C coarsened by r, in ONE marching pass
    // (the fine residual is never stored); CF ghosts of phi as they are.  false: not available here (non-diagonal metric, small
    // level, a ratio entry other than 1 or 2) -- the caller then runs residual_i + launch_restrict
    bool residual_restrict_i(const LevelDev& C, double* crse, double* phi, const double* rhs, const int r[3]);
    void apply_op_i(int d, double* out, double* phi, bool homogeneous = true);
    // Dirichlet sides (EllipticConstDiriBCGhostClass, BCInterface/EllipticBCUtils.cpp:382-424): values per
    // {loX,hiX,loY,hiY,loZ,hiZ}, before finalize.  Such a solver runs the two-pass / direct-load kernels.
    void set_bc_values(const double v[6]);
    // Non-diagonal metric on a level with coarse-fine boundaries:
    //   cf_ev: ExtrapolateCFEV(phi, cfregion, 2) -- the edge / vertex ghosts next to the CF faces, after a CF fill
    //   (interpCFGhosts, MappedAMRPoissonOp.cpp:2193-2216);  flux_fields: getFlux (fillExtrap + MAPPEDGETFLUX, beta = 1)
    //   on every face of depth 0, for the flux register (reflux, :1615-1707)
    void cf_ev(int d, double* phi);
    double* const* flux_fields(double* phi);
    void flux_at_faces(double* phi, FullFlux& ff);   // getFlux's operands for the register kernels: fillExtrap run, nothing stored
    void mac_grad_full(double* phi);  // f_flux := MAC gradient of phi, non-diagonal metric (phi exchanged)
    // the MAC gradient G^a = J g^{ab} d_b(phi) on every a-face of depth 0, STORED (either metric; phi exchanged, CF ghosts
    // filled by the caller): levelGradientMAC's per-box part (Gradient.cpp:124-160)
    double* const* mac_grad(double* phi);
    bool has_diri() const { return diri_; }
    bool bc_values_zero() const
    {
        for (int d = 0; d < 3; ++d)
            for (int s = 0; s < 2; ++s)
                if (bc_value_[d][s] != 0.0) return false;
        return true;
    }
    // ConstInterpPS / ZeroAvgConstInterpPS of depth 0 from a coarse field living on layout C (AMRProlong)
    void prolong_from(const LevelDev& C, const double* crse, const int r[3], double* fine);
    double* amr_field(int which);  // 0 m_correction, 1 m_residual of MappedAMRMultiGrid (allocated on first use)
    void restrict_residual(int d, double* resCoarse, double* phiFine, const double* rhsFine);
    // defer_mean: leave the zero-average mean (if any) to the caller; returns the device (sum, volume) pair to
    //             subtract, or nullptr if nothing is pending
    const double* prolong_increment(int d, double* phiFine, const double* corrCoarse, bool defer_mean = false);
    void pre_cond(int d, double* phi, const double* rhs);
    void vcycle(double* e, const double* res, bool e_zero = false);  // MappedMultiGrid::oneCycle
    // MG ratios imposed on the first depths (set before finalize): the coarsening pattern of the mini V-cycle of a
    // level refined by more than 2 (MappedAMRMultiGrid.H:1455-1482), and that mini V-cycle itself (:742-754)
    std::vector<std::array<int, 3>> forcedRatios;
    void mini_vcycle(double* corr, const double* res);
    void bottom_solve(double* phi, const double* rhs);
    double norm(int d, const double* a, int ord);
    double dot(int d, const double* a, const double* b);
    void fill_hash(int d, double* f, unsigned long long seed);
    // ---- MAC level projection pieces (LevelMACProjector / BaseProjector::project), depth 0 -------------
    double* vel(int dir);   // resident face field J*u^dir (allocated on first use)
    void upload_vel(int dir, int patch, const double* host);     // host FAB over faces(valid, dir)
    void download_vel(int dir, int patch, double* host);
    void divergence_mac(double* out, double dt);                 // out = div(vel) [/ dt]
    void mac_correct(double* phi, double dt);                    // vel -= dt * G(phi)
    void mac_project(double dt, bool zeroPressure, bool forceHomogeneous, SolveStats& st);
    void set_metric_map(int kind, const double Lc[3], const double* depth, const int dlo[2], const int dn[2]);
    void set_vel_bc(const int kind[6], const double value[6]);   // inflow / outflow sides of BasicVelocityBCGhostClass
    void vel_wall_bc();   // the velocity BC levelDivergenceMAC applies through a_fluxBC, solid walls: zero wall-normal faces of vel()
    // viscous / diffusive Helmholtz solves through the same operator
    // amr_member_ok: the caller (AMRSolver::set_alpha_beta) looks after the flux-register scales, which carry beta
    void set_alpha_beta(double a, double b, bool amr_member_ok = false);
    // a_flux of the level heat integrators: thisFlux (+)= getFlux(phi) = J Grad(phi), NO beta (MappedBaseLevelHeatSolver::
    // incrementFlux, MappedBaseLevelHeatSolver.cpp:183-209; MappedAMRPoissonOp::getFlux(FluxBox&,...), :2129-2151); phi's ghosts
    // as the last operation left them
    void increment_heat_flux(double* phi, bool setToZero);
    double* heat_flux(int dir);
    void download_heat_flux(int dir, int patch, double* host);   // host: faces(valid, dir), Fortran order   // MappedAMRPoissonOp::setAlphaAndBeta on every depth: alpha = a*aCoef, beta = b*bCoef
    double* heat_field(int which);             // 0: phiOld, 1: src (depth 0, allocated on first use)
    void heat_step(int scheme, double dt, bool zeroPhi, SolveStats& st);   // 0 backward Euler, 1 Crank-Nicolson, 2 TGA
    // cell-centred level projection (LevelCCProjector): velocity J*u, SpaceDim comps, resident with one ghost layer
    double* cc_vel(int comp);
    void upload_cc_vel(int patch, const double* host, const int ghost[3]);   // host FAB: SpaceDim comps, comp slowest
    void download_cc_vel(int patch, double* host, const int ghost[3]);       // valid cells only are written
    void divergence_cc(double* out, double dt, bool wall);                   // CellToEdge [+ wall BC] -> vel(); div [/ dt]
    void cc_correct(double* phi, double dt);                                 // cc_vel -= dt * EdgeToCell(G(phi))
    void cc_project(double dt, bool zeroPressure, bool forceHomogeneous, bool wall, SolveStats& st);
    // LevelGeometry::multByJ / divByJ on the resident velocities (geometry/LevelGeometryUtil.cpp:287-339, 372-420, 456-...):
    // data *= J resp. data *= Jinv, the scale arrays being the caller's (getCCJ / getCCJinv over valid.grow(ghost), fill_J /
    // fill_Jinv on the face boxes).  which: 0 = J, 1 = Jinv.  centring 0: the MAC velocity (set per direction), 1: cell-centred.
    void set_scale_cc(int which, int patch, const double* host, const int ghost[3]);
    void set_scale_face(int which, int dir, int patch, const double* host);
    void scale_vel(int centring, int which);
    void remove_mean(int d, double* f);
    void sync();
    // per-kernel HIP-event timing of the depth-0 launches (0 = GSRB colour pass, 1 = operator/residual)
    void profile_enable(bool on);
    void profile_get(int kernel, int* count, double* total_ms);
    hipStream_t stream() const { return st_; }

    SolverParams prm;
    // The null-space probe compares against ProblemContext's GLOBAL AMRMG.eps (MappedAMRPoissonOpFactory.cpp:679), not
    // against this solver's own eps; a solver set up by another one (leptic: horizontal / full multigrid) inherits it.
    double probe_eps = -1.0;
    // bottom-solver state shared with MappedAMRMultiGrid (setConvergenceMetrics)
    double bottom_metric = -1.0, bottom_eps_eff = 1e-6;
    int bottom_iters = 0, bottom_exit = 0;
    long long counters[5] = {0, 0, 0, 0, 0};   // overlapped sweeps, one-launch ghost programs, staged ghost programs, bottom solves, fused 19-point sweeps
    int bottom_kind = 0;   // how the last bottom solve ran: 0 launch by launch, 1 k_tiny_bicgstab, 2 k_box_bicgstab
    std::vector<std::array<int, 3>> mgRefRatios;

private:
    void cycle(int d, double* corr, const double* res, bool corr_zero = false);
    double fetch_scalar(int slot);
    void fetch_scalars(int slot, int n);
    unsigned long long fetch_seq_ = 0;
    unsigned long long* h_seq_ = nullptr;  // in the coherent host block behind h_scalars
    bool build_coarser(int depth);
    void probe_null_space(int d);
    void fill_metric_ghosts(Level& L);
    void detect_zero_planes();     // sets StencilParams::zero_xy per depth (non-diagonal metric, 3-D)
    void detect_uniform_metric();  // sets StencilParams::uniform / uc per depth (diagonal metric, 3-D)
    void line_relax(int d, double* e, const double* res);
    double* f_vel[3] = {nullptr, nullptr, nullptr};
    double* f_ccvel[3] = {nullptr, nullptr, nullptr};
    double* f_heat[3] = {nullptr, nullptr, nullptr};
    double* f_sc_cc[2] = {nullptr, nullptr};
    double* f_sc_face[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    double aCoef_ = 0.0, bCoef_ = 1.0;  // the factory's alpha / beta (MappedAMRPoissonOpFactory.cpp:585-586)
    bool coefs_saved_ = false;
    bool velbc_default_ = true;                      // every non-periodic side a solid wall
    int velbc_kind_[6] = {0, 0, 0, 0, 0, 0};
    double velbc_value_[6] = {0, 0, 0, 0, 0, 0};
    bool amr_member_ = false;
    double* f_amr[2] = {nullptr, nullptr};
    bool hasCF_ = false;
    double* f_heatflux[3] = {nullptr, nullptr, nullptr};
    bool own_stream_ = true;
    double dxCrse_[3] = {0, 0, 0};
    std::vector<double*> f_pp;  // per-depth ping-pong buffer of the fused sweep
    bool fused_bottom(int d) const;   // the whole BiCGStab bottom solve in one single-workgroup launch (k_tiny_bicgstab)
    // the BiCGStab bottom solve of a multi-box bottom level as one persistent launch, one workgroup per box (k_box_bicgstab);
    // SOMAR_BOX_BOTTOM=0 is the A/B switch (read once, at construction).  The neighbour table is built at the first use.
    bool box_bottom(int d) const;
    void build_box_tables(int d);
    bool box_bottom_on_ = true;
    long long fused_bottom_max_ = 512;   // k_tiny_bicgstab takes bottoms of at most this many cells (0: off)
    long long box_min_cells_ = 513;      // smaller single-workgroup bottoms stay with k_tiny_bicgstab
    int box_depth_ = -1, box_max_cells_ = 0;
    int *d_box_nb_ = nullptr, *d_box_cstart_ = nullptr, *d_box_fab_ = nullptr, *d_box_fabstart_ = nullptr;
    // 19-point variant: the two ghost programs compiled into per-cell entries (kernels.h: BoxProgEntry), per box
    BoxProgEntry* d_box_ent_[2] = {nullptr, nullptr};
    int *d_box_entfirst_[2] = {nullptr, nullptr}, *d_box_stg_[2] = {nullptr, nullptr}, *d_box_stgfirst_[2] = {nullptr, nullptr};
    int *d_box_nfg_[2] = {nullptr, nullptr}, *d_box_nfgfirst_[2] = {nullptr, nullptr};
    bool box_full_ok_ = false;   // the compiled programs fit the kernel's LDS tables
    double* d_box_sums_ = nullptr;
    unsigned* d_box_sync_ = nullptr;
    long long fused_min_cells_ = 262144;
    long long march_min_cells_ = 262144;  // levels at least this big use the k-marching operator/residual
    long long ordered_max_cells_ = 4096;
    bool ordered(int d) const { return lev[d]->valid_cells_global <= ordered_max_cells_; }
    // A small level that is SHARDED keeps the serial order too: every rank contributes the per-cell terms of its boxes to
    // one vector in the serial (box after box) sequence, a sum-allreduce completes it, one wavefront walks it
    // (k_ord_fill / k_reduce_ordered_flat) -- so a sharded solve adds the same numbers in the same order as one rank.
    bool ord_sharded(int d) const { return ordered(d) && comm_->size > 1; }
    double* d_ordbuf_ = nullptr;                    // 2 * ordered_max_cells_ doubles
    std::vector<long long*> d_ord_start_, d_box_start_;  // per depth: serial start of each LOCAL patch / of every box (+ end)
    void ordered_sums(int d, const double* a, const double* b, int mode, double dxProduct, double* out);
    // out[0] = sum over the level (mode 0: a*b, 2: |a|), every rank: serial order on small levels, tree order on large ones
    bool fused_publish(int d) const;
    double reduce_fetch(int d, const double* a, const double* b, int mode);   // reduction + host fetch, published by the reduction
    void wait_published(unsigned long long want);
    void reduce_sum(int d, const double* a, const double* b, int mode, double* out);

    Comm* comm_;
    Comm self_;
    hipStream_t st_ = nullptr;
    std::vector<std::unique_ptr<Level>> lev;
    std::vector<double*> f_res, f_corr, f_scratch;  // per depth
    double *f_phi = nullptr, *f_rhs = nullptr, *f_uberRes = nullptr, *f_uberCorr = nullptr, *f_best = nullptr;
    double* bicg[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* d_partials = nullptr;
    double* d_scalars = nullptr;  // device scalar slots
    double* h_scalars = nullptr;  // pinned
    bool finalized = false;
    // ---- prolongation folded into the first post-smoothing sweep (large levels) ----
    // f_W[d+1]: per coarse cell the volume of its children on depth d; d_fold: per depth {S_f, S_c, V, sum, V}
    std::vector<double*> f_W;
    double* d_fold = nullptr;
    std::vector<char> sf_valid_;
    bool fold_prolong(int d) const;
    // ---- non-diagonal metric (19-point) path, solver_full.cpp ----
    // d_ops: the ops stage by stage (first / count per stage) for the staged form, one launch per stage; d_box_ops / d_box_first:
    // the same ops sorted by box, then stage (GhostOp::pad_), for the one-launch form (k_ghost_program, small levels)
    struct FullProgram { GhostOp* d_ops = nullptr; std::vector<int> first, count; GhostOp* d_box_ops = nullptr; int* d_box_first = nullptr; int max_box_ops = 0;
                         std::vector<GhostOp> h_box_ops; std::vector<int> h_box_first; };   // host copy of the box-sorted list (k_box_bicgstab's tables)
    void upload_program(FullProgram& P, const std::vector<std::vector<GhostOp>>& stages, int npatches);
    static void free_program(FullProgram& P);
    // small levels run a ghost program as ONE launch, one workgroup per box (SOMAR_GHOST_STAGED=1: always stage by stage)
    bool box_program(int d) const { return ghost_box_on_ && !full_march(d) && lev[d]->npatches() > 0; }
    bool ghost_box_on_ = true;
    void run_program(int d, const FullProgram& P, double* phi, double* psi, bool homogeneous, bool redirect, bool copy_all);
    bool full_ = false;
    std::vector<double*> f_psi;                            // per depth: the extrapolated copy of phi
    // per depth: [0] operator, [1] smoother, [2] fillExtrap alone (getFlux of the flux register), [3] ExtrapolateCFEV
    FullProgram aux_prog_[2];
    bool aux_built_[2] = {false, false};
    std::vector<std::array<FullProgram, 7>> full_prog_;   // + [4] / [5]: [0] / [1] writing psi in the boxes' frames only (marching kernels); [6]: the plain frame copy (copy_frames)
    double* f_flux[3] = {nullptr, nullptr, nullptr};  // face fluxes of depth 0 (refluxing with a non-diagonal metric)
    void alloc_full_metric(Level& L);
    void build_full_programs(int d);
    void run_full_program(int d, int which, double* phi, bool homogeneous = true);
    void run_full_program_frames(int d, int which, double* phi, bool homogeneous = true);
    void copy_frames(int d, const double* src, double* dst);   // dst := src in the one-cell frame of every box
    // large 3-D levels of the non-diagonal path run the k-marching 19-point kernels (full19_march.hip)
    bool full_march(int d) const
    {
        const Level& L = *lev[d];
        return full_ && L.active[2] && L.valid_cells_global >= march_min_cells_;
    }
    // ... or, where every box of the level is at least fused19_min_box_ cells wide in every direction, red + black in one launch
    // plus a shell pass (full19_fused.hip).  OFF by default (negative): measured on one MI355X it loses at every box size --
    // 512^3 in one box: 9.1 ms (fused) + 0.38 ms (shell) against 2 x 2.85 ms; boxes of 128: 14.9 against 6.6 ms; of 64: 16.0
    // against 6.0 (profiles/r03_fused19.txt).  SOMAR_FUSED19_MIN_BOX = 0 forces it onto every marching level (tests, A/B).
    int fused19_min_box_ = -1;
    std::vector<char> fused19_;   // per depth, decided in finalize
    bool fused19(int d) const { return d < (int)fused19_.size() && fused19_[d] != 0; }
    std::unique_ptr<PressureSolver> coarse_;   // replicated tail of the hierarchy (agglomeration)
    int agglom_depth_ = -1;
    long long agglom_cells_ = 2097152;  // 128^3: below this a level costs less to replicate (~0.2 ms of sweeps) than to exchange (~8 x 60 us)
    Copier agglom_gather_;                      // sharded depth agglom_depth_ -> replicated depth 0 of coarse_
    CopyItem* d_agglom_back_ = nullptr;         // my boxes of the replicated correction -> sharded layout
    int n_agglom_back_ = 0;
    void build_agglomerated_tail(int depth);
    void agglom_cycle(double* corr, const double* res, bool corr_zero);
    // ---- launch-bound coarse depths replayed as HIP graphs ----
    // From the first depth with at most graph_cells_ cells on, one V-cycle is ~20 microsecond-sized launches per
    // depth; the legs on either side of the bottom solve (whose loop needs host decisions) are captured once and
    // replayed.  Only where nothing but kernels is enqueued: one rank (which includes the replicated tail).
    struct CoarseGraph {
        hipGraphExec_t down = nullptr, up = nullptr;
        int d0 = -1, pre = -1, post = -1, bottom = -1;
        const double *corr = nullptr, *res = nullptr;
    };
    CoarseGraph cg_;
    long long graph_cells_ = 262144;
    int graph_from_ = -1;
    bool capturing_ = false;
    bool diri_ = false;
    double bc_value_[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    std::vector<GhostOp*> d_diri_ops_;
    std::vector<int> n_diri_ops_;
    GhostOp* d_extrapbc_ops_ = nullptr;  // order-2 extrapolation BC on the physical ghosts of depth 0 (gradient BC)
    int n_extrapbc_ops_ = 0;
    void build_diri_ops(int d);
    void apply_diri(int d, double* phi, bool homogeneous);
    int mini_depth_ = 0;  // > 0 while a mini V-cycle runs: the depth count it is limited to
    int cycle_override_ = 0;  // != 0 inside an F-cycle's inner V-cycles: the effective numMG ("m_cycle = 1" hack, MappedMultiGrid.H:603-605)
    void cycle_down(int d, double* corr, const double* res, bool corr_zero);  // pre-smoothing + restriction
    void cycle_up(int d, double* corr, const double* res);                    // prolongation + post-smoothing
    void cycle_bottom_relax(double* corr, const double* res, bool corr_zero);
    bool graph_cycle(int d, double* corr, const double* res, bool corr_zero);
    void drop_graphs();
    struct Prof { std::vector<hipEvent_t> a, b; int used = 0; };
    Prof prof_[4];   // 0: GSRB launches of depth 0, 1: its operator / residual launches, 2: ghost exchanges with other ranks (any
                     // depth), 3: the replicated tail of a sharded hierarchy (agglom_cycle)
    // ghost exchange of a level of this solver (timed in a profiled pass when other ranks take part)
    void xchg(const Level& L, double* f);
    // Sharded large levels: the messages of a sweep's one ghost exchange travel on a second stream while the tiles that read no
    // remote ghost cell are swept (the LooseGSRB idea, GSRB.cpp:122-140, without its change of the iteration: the fused sweep's
    // tiles are independent, so this is the same sweep bit for bit).  SOMAR_NO_OVERLAP=1 is the A/B switch.
    bool overlap_on_ = true;
    hipStream_t st_comm_ = nullptr;
    hipEvent_t ev_ready_ = nullptr, ev_done_ = nullptr;
    bool resid_overlap(const Level& L) const
    {
        return overlap_on_ && !L.plan.peers.empty() && L.nrtiles_own > 0 && !capturing_ && !profiling_ && !diri_;
    }
    // exchange of f with its remote half on the second stream; run(tiles, n) is issued for the tiles that read no remote ghost
    // first, for the others once the messages have landed
    template <class Run>
    void overlapped(const Level& L, double* f, Tile* own, int nown, Tile* rem, int nrem, Run run)
    {
        if (!st_comm_) {
            SOMAR_HIP(hipStreamCreateWithFlags(&st_comm_, hipStreamNonBlocking));
            SOMAR_HIP(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming));
            SOMAR_HIP(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
        }
        SOMAR_HIP(hipEventRecord(ev_ready_, st_));
        SOMAR_HIP(hipStreamWaitEvent(st_comm_, ev_ready_, 0));
        L.exchange_local(f, st_);
        run(own, nown);
        L.exchange_remote(f, st_comm_);   // (a host-staged transport may block here: the tiles above are already queued)
        SOMAR_HIP(hipEventRecord(ev_done_, st_comm_));
        SOMAR_HIP(hipStreamWaitEvent(st_, ev_done_, 0));
        run(rem, nrem);
        ++counters[0];
    }
    bool fused_overlap(const Level& L) const
    {
        return overlap_on_ && !L.plan.peers.empty() && L.nftiles_own > 0 && !capturing_ && !profiling_;
    }
    bool profiling_ = false;
    void prof_begin(int k);
    void prof_end(int k);
};

}  // namespace somar
