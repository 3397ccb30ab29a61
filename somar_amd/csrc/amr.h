// somar_amd/csrc/amr.h -- several AMR levels: the MI355X counterpart of
//   MappedAMRMultiGrid<T> (multi-level part)  calculus/AMRElliptic/MappedAMRMultiGrid.H:736-927, 933-1215, 1320-1598
//   MappedAMRPoissonOp AMR* members + reflux  calculus/AMRElliptic/MappedAMRPoissonOp.cpp:1311-1707
//   MappedQuadCFInterp / MappedQuadCFStencil  MappedChombo/MappedQuadCFInterp.cpp, MappedCFStencil.cpp:831-1232
//   MappedLevelFluxRegister                   MappedChombo/MappedLevelFluxRegister.cpp
//
// Design (not a port): every level keeps its fields resident in HBM (one PressureSolver per level, all on
// one HIP stream).  Everything that crosses levels is precomputed on the host into flat tables --
//   * a Copier (box-to-box copy list, packed messages between ranks) coarse level -> "coarsened fine"
//     buffer layout and back,
//   * per coarse-fine ghost cell records for the quadratic interpolation (stencil weights included),
//   * per coarse cell records for refluxing (which coarse faces to subtract, which fine register sums to add)
// -- so that at solve time an inter-level operation is one or two kernel launches over a table, no host
// geometry, no atomics (each output cell is owned by exactly one thread, contributions are added in the
// reference's order => bit-identical results).
#pragma once
#include <memory>
#include <vector>

#include "leptic.h"
#include "solver.h"

namespace somar {

// ---- device records (amr_kernels.hip) ---------------------------------------------------------------
struct QPoint {       // one point of a derivative stencil
    long long off;    // element offset relative to the coarse cell, in the buffer field
    double w;
};
struct QCoarse {      // one coarse cell under the CF ghost layer of (fine box, dir, side)
    long long boff;   // offset of the cell in the buffer field
    int flags;        // bit0: standard (centred) stencils
    int dir;          // normal direction
    int s1, s2;       // buffer element strides of the two tangential directions (ascending)
    int p0;           // first QPoint
    int np[5];        // points of D1_t1, D2_t1, D1_t2, D2_t2, Dmixed
};
struct QFine {        // one fine CF ghost cell
    long long foff;   // its offset in the fine field
    int stride;       // signed element stride pointing OUT of the box: foff - stride is the first valid cell
    int cc;           // QCoarse index
    int ivf1, ivf2, ivc1, ivc2;  // global fine / coarse indices in the two tangential directions
    int dirflags;     // dir | packed << 2
    int pad_;
};
struct FRegCell {     // one cell of the fine flux register: sum over the fine faces of one coarse face
    long long cell0;  // offset (fine field) of the cell whose LOW face is the first fine face
    int patch;        // fine patch
    int dir, side;
    int pad_;
};
struct RefluxA {      // coarse-side contribution: coar += sc * F_c(face)
    long long face;   // offset (coarse field) of the cell whose LOW face it is
    double sc;
    int dir;
    int sgn;          // sc = sgn * (beta / dx[dir]) of the coarse operator (MappedAMRPoissonOp::reflux, :1661)
};
struct RefluxCell {   // one coarse cell next to the fine level
    long long coff;
    int patch;
    int a0, na;       // RefluxA range
    int b0, nb;       // register-value index range (into the rank's register array)
    int pad_;
};
struct OneSided {     // one face of CRSEONESIDEGRAD: g[face] = 2 g[face + stride] - g[face + 2 stride]  (mode 2) or g[face + stride] (mode 1)
    long long face;   // offset (coarse field) of the cell whose LOW face it is
    int stride;       // signed element stride pointing AWAY from the finer level
    int dirmode;      // dir | mode << 2
};
struct FillItem {     // a region of one patch
    int patch;
    int lo[3];        // local start
    int n[3];
    int pad_;
};

void launch_one_sided(hipStream_t st, const OneSided* e, int n, double* const grad[3]);
// crse(ic) = (sum over the children of ic of fine) * (1 / prod r): UNMAPPEDAVERAGE (MappedCoarseAverageF.ChF:7-41)
void launch_avg_unweighted(hipStream_t st, const LevelDev& C, const LevelDev& F, double* crse, const double* fine, const int r[3]);
void launch_divide(hipStream_t st, double* a, double d, long long n);   // a[i] = a[i] / d
void launch_fill_items(hipStream_t st, const PatchDesc* patches, const FillItem* items, int nitems, double* f, double v);
void launch_cf_slopes(hipStream_t st, const QCoarse* cc, int ncc, const QPoint* pts, const double* buf, double* der,
                      const double dxc[3]);
void launch_cf_quad(hipStream_t st, const QFine* fc, int nfc, const QCoarse* cc, const double* der, const double* buf,
                    double* fine, const double dxf[3], const double dxc[3], const int r[3]);
// Non-diagonal metric: the face fluxes getFlux would fill a whole FluxBox with (fillExtrap + MAPPEDGETFLUX, beta = 1;
// MappedAMRPoissonOp.cpp:2048-2122) are evaluated AT THE REGISTER'S FACES ONLY, inside the register kernels: psi = the level's
// extrapolated copy of phi (ready), J[a][b] = J g^{ab} on a-faces, dxi = 1 / dx of that level.  psi == nullptr: not used.
void launch_fine_register(hipStream_t st, const FRegCell* cells, int n, const PatchDesc* fpatches, const double* phi,
                          double* const jg[3], const double dxf[3], const double sc[3][2], const int r[3], double* out,
                          double* const* fluxes = nullptr, const FullFlux* ff = nullptr);
void launch_gather(hipStream_t st, const int* idx, long long n, const double* src, double* dst);
void launch_reflux_rescale(hipStream_t st, RefluxA* A, long long n, const double scale[3]);
void launch_reflux(hipStream_t st, const RefluxCell* cells, int n, const RefluxA* A, const int* B,
                   const PatchDesc* cpatches, const double* phi, double* const jg[3], const double* jinv,
                   const double dxc[3], const double* freg, double* LofPhi, double* const* fluxes = nullptr,
                   const FullFlux* ff = nullptr);

// ---- host side -------------------------------------------------------------------------------------
// What ties level l (fine) to level l-1 (coarse).
struct AMRLink {
    int r[3] = {1, 1, 1};
    std::unique_ptr<Level> cfl;  // coarsened fine layout, owned by the fine boxes' ranks
    double* buf = nullptr;       // coarse values under and around the fine boxes (2 ghosts)
    double* resC = nullptr;      // restricted fine residual
    Copier gather;               // coarse level -> buf (valid + 2 ghosts, periodic)
    Copier gather_ring;          // the same onto the 2-cell ring around each coarsened fine box only: all the coarse-fine
                                 // interpolation reads (a 64 x 64 x 128 box: 86 K of 610 K cells; SOMAR_CF_FULL_GATHER=1: A/B)
    Copier scatter;              // resC valid -> coarse level valid
    // quadratic CF interpolation
    bool hasCF = false;          // false: the fine level covers the whole domain
    int ncc = 0, nfc = 0;
    QCoarse* d_cc = nullptr;
    QPoint* d_pts = nullptr;
    QFine* d_fc = nullptr;
    double* d_der = nullptr;
    // zeroCovered on the coarse level
    int ncover = 0;
    FillItem* d_cover = nullptr;
    // flux register
    bool fluxDefined = false;
    int nreg_local = 0, nreg_recv = 0, nreflux = 0;
    FRegCell* d_reg = nullptr;
    double* d_regvals = nullptr;       // [local | received]
    RefluxCell* d_reflux = nullptr;
    RefluxA* d_A = nullptr;
    long long nA = 0;
    int* d_B = nullptr;
    // register values that travel between ranks
    std::vector<int> peers;
    std::vector<long long> soff, scount, roff, rcount;
    int* d_sendidx = nullptr;
    long long nsend = 0;
    double* d_sendbuf = nullptr;
    double sc_fine[3][2];
    double beta_built = 0.0;   // the coarse operator's beta the register scales were built with
    // compGradientCC's one-sided faces on the COARSE level next to this (fine) level, in stages (entries of one stage are
    // independent; normally there is one stage)
    bool osg_built = false;
    OneSided* d_osg = nullptr;
    std::vector<int> osg_first, osg_count;
    ~AMRLink();
};

class AMRSolver {
public:
    explicit AMRSolver(Comm* comm = nullptr);
    ~AMRSolver();

    // levels[l]: boxes + owners in level-l index space; domain/dx are level 0's, refined by ratios
    void define(const IBox& domain0, const bool periodic[3], const double dx0[3], const int bc_type[3][2],
                const std::vector<std::array<int, 3>>& ratios, const std::vector<std::vector<IBox>>& boxes,
                const std::vector<std::vector<int>>& owners, double alpha, double beta, const SolverParams& prm);
    void finalize();  // after every level's metric is set
    int nlevels() const { return (int)S.size(); }
    PressureSolver& level(int l) { return *S[l]; }

    // MappedAMRMultiGrid::solve on the resident phi/rhs of levels l_base..l_max (phi of l_base-1 supplies the
    // CF values when l_base > 0)
    void solve(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& st);
    // AMRLepticSolver (calculus/LepticSolver/AMRLepticSolver.cpp): the same composite iteration with LevelLepticSolver::solve
    // in place of relax / the base level's multigrid cycle.  enable_leptic (after finalize) builds one leptic level solver per
    // level on the level's own operator (init, :185-195).  baseFromRestricted = false is the reference: the base level solves
    // a_uberCorrection from a_uberResidual and its m_correction stays zero (:444-449); true (NOT the reference) feeds it the
    // restricted residual as MappedAMRMultiGrid's V-cycle does.
    // MappedBaseLevelHeatSolver::resetSolverAlphaAndBeta on every op of every level (MappedBaseLevelHeatSolver.cpp:257-270).
    // The flux-register scales follow: reflux() rewrites them when the coarse operator's beta differs from the one they hold
    // (sync_reflux_scales), as MappedAMRPoissonOp::reflux takes m_beta / m_dx when it runs.
    void set_alpha_beta(double a, double b);
    // MappedAMRTGA<T>::oneStep (AMRElliptic/MappedAMRTGA.H:417-497): one composite TGA step over levels l_base..l_max;
    // phiNew = PHI, phiOld = HEAT_OLD, source = HEAT_SRC of every level; st = the LAST solve's
    void tga_step(int l_max, int l_base, double dt, SolveStats& st);
    // MappedLevelBackwardEuler / MappedLevelCrankNicolson / MappedLevelTGA::updateSoln on level l of the hierarchy
    // (AMRParabolic/*.cpp): phiNew = level l's PHI, phiOld / src = its HEAT_OLD / HEAT_SRC; for l > 0 the coarse-fine values
    // are timeInterp(level l-1's HEAT_OLD at crseOldTime, level l-1's PHI at crseNewTime); a_flux accumulates in heat_flux(d)
    void heat_step(int l, int scheme, double dt, bool zeroPhi, double oldTime, double crseOldTime, double crseNewTime,
                   SolveStats& st);
    void enable_leptic(const LepticParams& lp, bool baseFromRestricted);
    void solve_leptic(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& st);
    const LepticStats& leptic_stats(int l) const { return lepStats_.at(l); }

    // BaseProjector<T>::levelProject -> project(lmin = lmax = l) (projection/BaseProjectorI.H:176-366) on the resident
    // velocity of level l: centring 0 = LevelMACProjector (vel()), 1 = LevelCCProjector (cc_vel(), level l-1's supplies the
    // velocity's coarse-fine values); phi of level l-1 supplies the pressure's.  Refined levels: diagonal metric.
    void level_project(int l, int centring, double dt, bool zeroPressure, bool forceHomogeneous, bool wall, SolveStats& st);

    // ---- the COMPOSITE cell-centred projector (sync / init / regrid projection) on the levels' resident cc velocities ----
    //   BaseProjector<FArrayBox>::project over l_min..l_max   projection/BaseProjectorI.H:176-299
    //   AMRCCProjector::computeDiv/computeGrad/applyCorrection projection/AMRCCProjector.cpp:204-377
    //   Divergence::compDivergenceCC (refluxed)                calculus/DivCurlGrad/Divergence.cpp:697-838
    //   Gradient::compGradientCC (one-sided CF faces)          calculus/DivCurlGrad/Gradient.cpp:707-842
    void cc_project(int l_min, int l_max, double dt, bool zeroPressure, bool forceHomogeneous, bool wall, SolveStats& st);
    // pieces: out = compDivergenceCC of level l (finer level l+1 refluxed in when l < l_max; NOT divided by dt);
    // cc_vel(l) += dtScale * compGradientCC(phi) with phi = field `phi` of level l (PHI of l-1 / l+1 as coarse / fine data);
    // the valid cells of level l under level l+1 := plain average of level l+1's cc velocity
    void comp_divergence_cc(int l, int l_max, double* out, bool wall);
    void comp_grad_correct_cc(int l, int l_max, double* phi, double dt);
    void average_down_ccvel(int l);

    // pieces (parity tests)
    // interpCFGhosts(phi, &phiCoarse, false); ev = false: MappedQuadCFInterp::coarseFineInterp alone (no ExtrapolateCFEV)
    void interp_cf(int l, double* phiFine, const double* phiCoarse, bool ev = true);
    // homogeneous: physical boundary values taken as zero (Dirichlet sides); the CF values always come from phiCoarse
    void amr_operator(int l, double* LofPhi, double* phiFine, double* phi, const double* phiCoarse,
                      bool homogeneous = true);
    void amr_residual(int l, double* res, double* phiFine, double* phi, const double* phiCoarse, const double* rhs,
                      bool homogeneous = true);
    void amr_residual_nf(int l, double* res, double* phi, const double* phiCoarse, const double* rhs,
                         bool homogeneous = true);
    void reflux(int l, double* phiFine, double* phi, double* LofPhi);
    void sync_reflux_scales(int lf);
    void compute_residual_levels_only(double* const* resid, double* const* phi, double* const* rhs, int l_max, int l_base,
                                      bool homogeneous);
    void amr_restrict(int l, double* residual, double* correction, const double* coarseCorrection, double* scratch);
    void assign_coarse_residual(int l, double* coarseResidual);  // assignCopier of link l's resC
    void amr_prolong(int l, double* correction, const double* coarseCorrection);
    void amr_update_residual(int l, double* residual, double* correction, const double* coarseCorrection);
    void zero_covered(int l, double* f);
    void compute_residual_level(double* const* resid, double* const* phi, double* const* rhs, int l_max, int l_base,
                                int ilev, bool homogeneous);
    double compute_residual(double* const* resid, double* const* phi, double* const* rhs, int l_max, int l_base,
                            bool homogeneous);
    void vcycle(double* const* uberCorr, double* const* uberRes, int ilev, int l_max, int l_base);
    void level_relax(int l, double* corr, const double* res, int iters, bool corr_zero = false);
    double* corr(int l) { return corr_[l]; }
    double* res(int l) { return res_[l]; }
    AMRLink& link(int l) { return *links_[l]; }
    hipStream_t stream() const { return st_; }
    void sync() { SOMAR_HIP(hipStreamSynchronize(st_)); }

    SolverParams prm;
    // MappedAMRMultiGridInspector (MappedAMRMultiGrid.H:260-298): called from solve() with the solver's stream synchronized,
    // kind 0 = recordResiduals (uberResidual, before every V-cycle), 1 = recordCorrections (uberCorrection, after it)
    typedef void (*Inspector)(void* user, int kind, int iter, int l_min, int l_max);
    void set_inspector(Inspector f, void* user) { inspector_ = f; inspector_user_ = user; }

private:
    std::vector<std::unique_ptr<LepticSolver>> leptic_;
    std::vector<LepticStats> lepStats_;   // of each level's last leptic solve
    bool lepticCycle_ = false, lepticBaseFromRestricted_ = false;
    double* crse_override_ = nullptr;   // solve_impl: stands in for phi of level l_base - 1 (the heat integrators' coarseData)
    void solve_impl(int l_max, int l_base, bool zeroPhi, bool forceHomogeneous, SolveStats& st);
    Inspector inspector_ = nullptr;
    void* inspector_user_ = nullptr;
    void build_link(int l);
    void build_quad_tables(int l);
    void build_reflux_tables(int l);
    void build_one_sided_tables(int l);   // link l: the one-sided faces on level l-1
    Comm* comm_;
    Comm self_;
    hipStream_t st_ = nullptr;
    std::vector<std::unique_ptr<PressureSolver>> S;
    std::vector<std::unique_ptr<AMRLink>> links_;  // links_[l] ties l to l-1 (links_[0] unused)
    std::vector<std::array<int, 3>> ratios_;
    std::vector<double*> corr_, res_;
    // lean V-cycle (default; SOMAR_AMR_PLAIN=1 restores the reference's pass-by-pass structure): where level l's residual
    // currently lives (res_[l], spare_[l], or -- finest level -- the caller's uberResidual, which is then never written)
    std::vector<double*> spare_, rcur_;
    std::vector<int> visits_;
    bool lean_ = true;
    bool finalized_ = false;
};

}  // namespace somar
