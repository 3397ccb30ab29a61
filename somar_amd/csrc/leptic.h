// somar_amd/csrc/leptic.h -- the leptic level solver: the MI355X counterpart of
//   LevelLepticSolver::define / solve     calculus/LepticSolver/LevelLepticSolver.cpp:147-437, 646-956
//   LepticOperator (what it asks of the level operator)   calculus/LepticSolver/LepticOperator.H:33-45
//
// Design (not a port): the level's boxes are vertically complete columns -- the layout the reference re-creates
// with LepticBoxUtils::createVerticalSolverGrids before every solve path; here the caller hands it over once, a
// 288 GB HBM holds such columns whole -- so the reference's four re-layout Copiers (orig <-> vertical, flat <->
// horizontal) disappear.  Three PressureSolvers share one stream and the layout: the level's own operator (for
// the initial residual), the J-scaled operator (J^{-1} := 1, alpha 0, beta 1) with the full 3-D multigrid used
// as fallback, and the flat 2-D multigrid on the vertically averaged metric.  Everything in between is column
// kernels (leptic_kernels.hip); the host sequences launches and reads back one norm per order.
//
// Scope: diagonal or non-diagonal metric; coarse-fine boundaries on the lateral sides of the columns (the level of an AMR
// hierarchy refined by (r, r, 1): attach()); columns that are ALL Neumann-Neumann (horizontal solves) or ALL ended by a
// Dirichlet wall / a coarse-fine interface (LepticLapackVerticalSolver + dptsv, no horizontal solves; layouts mixing the two
// kinds raise); homogeneous-Neumann lateral boundaries,
// non-periodic directions (the reference leaves the averaged gradient on a periodic horizontal boundary face
// unset, LevelLepticSolver.cpp:997-1001, and refuses a periodic vertical, :1315).
#pragma once
#include <memory>
#include <vector>

#include "solver.h"

namespace somar {

struct LepticParams {
    // setDefaultParameters, LevelLepticSolver.cpp:461-508
    int maxOrder = 4;
    double hang = 1e-15;
    int normType = 0;
    double horizRhsTol = 1e-14;
    double domainHeight = 0.0;  // LevelGeometry::getDomainLength(SpaceDim-1); 0: dz * Nz
    SolverParams horiz, full;
    LepticParams();
};

struct LepticStats {
    int exitStatus = -1;  // LevelLepticSolver::ExitStatus: 0 converge, 1 iter, 2 hang, 3 diverge, 4 kaboom
    int orders = 0;
    int horizSolves = 0;
    int usedFullSolver = 0;
    std::vector<double> resNorms;  // [0] initial |J res|, [k] after order k-1
    SolveStats horizStats, fullStats;
};

class LepticSolver {
public:
    // shared: run on the caller's stream (an AMR hierarchy's) instead of an own one
    explicit LepticSolver(Comm* comm = nullptr, hipStream_t shared = nullptr);
    ~LepticSolver();
    void define(const IBox& domain, const bool periodic[3], const double dx[3], const int bc_type[3][2],
                const std::vector<IBox>& boxes, const std::vector<int>& owner, double alpha, double beta,
                const SolverParams& prmOrig, const LepticParams& lp, const double* dxCrse = nullptr);
    // LevelLepticSolver::define(opPtr, homogeneous) on the FINALIZED operator of an AMR level (AMRLepticSolver::init,
    // AMRLepticSolver.cpp:185-195): layout, spacing, boundary types and dxCrse are the operator's; not owned
    void attach(PressureSolver* orig, const LepticParams& lp);
    // the level's own operator: metric, phi and rhs go through it (somar_solver_* entry points)
    PressureSolver& orig() { return *orig_; }
    PressureSolver& vert() { return *vert_; }
    PressureSolver& horiz() { return *horiz_; }
    PressureSolver* horiz_ptr() { return horiz_.get(); }   // null: no column is Neumann-Neumann, there is no flat problem
    void finalize();  // after the metric of orig() is set: finalizes all three solvers
    // LevelLepticSolver::solve(phi, rhs) on orig()'s resident phi / rhs: phi += leptic correction
    void solve(bool homogeneous, LepticStats& S);
    // the same on any two fields of the level's layout: phi += leptic correction for rhs - L[phi] (homogeneous CF / BC values)
    void solve_fields(double* phi, const double* rhs, LepticStats& S, bool homogeneous = true);
    bool do_horiz_solve() const { return doHorizSolve_; }
    void sync() { SOMAR_HIP(hipStreamSynchronize(st_)); }
    LepticParams prm;

private:
    void set_zero_avg(double* hphi);
    void solve_fields_no_horiz(double* phi, const double* rhs, LepticStats& S, bool homogeneous);
    Comm* comm_;
    hipStream_t st_ = nullptr;
    void define_inner(const IBox& domain, const bool periodic[3], const double dx[3], const int bc_type[3][2],
                      const std::vector<IBox>& boxes, const std::vector<int>& owner, const double* dxCrse, double probeEps);
    PressureSolver* orig_ = nullptr;
    std::unique_ptr<PressureSolver> own_orig_, vert_, horiz_;
    bool own_stream_ = true, hasCF_ = false;
    double dx_[3] = {1, 1, 1};
    double H_ = 1.0;
    bool horizRemoveAvg_ = false;
    long long horizCells_ = 0;
    bool finalized_ = false;
    bool full_ = false;   // non-diagonal metric (decided when orig()'s metric has been set)
    // 3-D work fields (vertical layout) and flat ones (horizontal layout)
    double *f_total = nullptr, *f_rhsA = nullptr, *f_rhsB = nullptr, *f_gam = nullptr;
    double *h_excess = nullptr, *h_bcLo = nullptr, *h_bcHi = nullptr, *h_gx = nullptr, *h_gy = nullptr;
    double* d_avg = nullptr;  // (sum, count) for setZeroAvg
    // columns ending at Dirichlet walls / coarse-fine interfaces (gatherVerticalBCTypes): per box (lo, hi) 0 Neum, 1 Diri, 2 CF
    std::vector<int> vbc_;
    bool doHorizSolve_ = true;
    double dzCrse_ = 0.0;
    double* f_efac = nullptr;
    int* d_vbc = nullptr;
    int* d_bad = nullptr;
};

void launch_lep_avg_metric(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H);
// non-diagonal metric: the horizontal block averaged, MAPPEDMACGRAD with cross terms, LEPTICVERTHORIZGRAD
void launch_lep_avg_metric_full(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H);
void launch_lep_hgrad_full(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                           const double* phi, double* gx, double* gy);
void launch_lep_vhgrad(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       const double* phi, double* bcLo, double* bcHi, double scale);
void launch_lep_excess(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       const double* rhs, const double* bcLo, const double* bcHi, double* excess, double dzScale);
void launch_lep_vsolve(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                       double* phi, double* rhs, double* gam, const double* bcLo, const double* bcHi, double dz);
void launch_lep_vsolve_lapack(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* phi,
                              const double* rhs, double* dfac, double* efac, const int* vbc, double dz, double dzCrse, int* bad);
void launch_lep_hgrad(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                      const double* phi, double* gx, double* gy);
void launch_lep_hrhs(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                     const double* gx, const double* gy, const double* excess, double* hrhs, double sx, double sy,
                     double negInvH, bool useExcess);
void launch_lep_extrude(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, const LevelDev& H,
                        double* phi, const double* flat);
void launch_lep_divide(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* y, const double* x,
                       const double* b);
void launch_lep_axpy(hipStream_t st, const Tile* ct, int nct, int tj, const LevelDev& V, double* y, const double* x,
                     double a);

}  // namespace somar
