/*
 * include/somar_amd.h -- C ABI of libsomar_amd.so: MI355X-native (gfx950, HIP) drop-in for
 * the pressure-projection hot path of UNC-CFD/somar.
 *
 * Plain pointers and sizes only; no C++ or torch types.  Every entry point returns 0 on
 * success and a negative code on failure (somar_last_error() gives the text); nothing
 * aborts the process -- the reference's MayDay::Error sites ("kaboom",
 * MappedAMRMultiGrid.H:1136,1144) are reported through somar_stats_t.status instead.
 *
 * What each group replaces in the reference (paths relative to /root/reference/src):
 *
 *  somar_solver_create / _set_metric_ortho / _finalize
 *      AMRPressureSolver::levelDefine / define          projection/AMRPressureSolver.cpp:163-314, 321-463
 *      MappedAMRPoissonOpFactory::define                calculus/AMRElliptic/MappedAMRPoissonOpFactory.cpp:95-205
 *      MappedAMRMultiGrid<T>::define                    calculus/AMRElliptic/MappedAMRMultiGrid.H:1407-1492
 *      (metric arrays = LevelGeometry::getFCJgupPtr / getCCJinvPtr, geometry/LevelGeometry.H:139-286)
 *  somar_params_t
 *      AMRPressureSolver::setAMRMGParameters / setBottomParameters   projection/AMRPressureSolver.H:53-77
 *  somar_solver_solve / somar_solver_solve_host
 *      AMREllipticSolver<LevelData<FArrayBox>>::solve(phi, rhs, l_max, l_base, zeroPhi, forceHomogeneous)
 *                                                       calculus/AMRElliptic/AMREllipticSolver.H:33-48
 *      as called from AMRPressureSolver::solve          projection/AMRPressureSolver.cpp:494-561
 *  somar_stats_t.exit_status
 *      MappedAMRMultiGrid<T>::m_exitStatus              calculus/AMRElliptic/MappedAMRMultiGrid.H:545, 1148
 *  somar_level_relax / _residual / _apply_op / _restrict_residual / _prolong_increment / _precond
 *      RelaxationMethod::relax                          calculus/AMRElliptic/RelaxationMethods/RelaxationMethod.H:34-52
 *      MappedAMRPoissonOp::residual/applyOp/restrictResidual/preCond
 *                                                       calculus/AMRElliptic/MappedAMRPoissonOp.cpp:628-765, 1281-1304, 684-734
 *      ProlongationStrategy::prolongIncrement           calculus/AMRElliptic/MGStrategies/ProlongationStrategy.H:35-48
 *  somar_vcycle
 *      MappedMultiGrid<T>::oneCycle                     calculus/AMRElliptic/MappedMultiGrid.H:528-548
 *
 * Host arrays use the Chombo BaseFab layout the Fortran kernels see (FORT_PROTO FRA):
 * column-major, inclusive [lo,hi] box, i fastest; a face-centred array in direction a spans
 * faces(valid, a) = valid with hi[a]+1, face i being the LOW face of cell i.
 * Ownership: host buffers stay caller-owned; device buffers belong to the handle.
 * Threading: one host thread drives one handle; distinct handles are independent.
 */
#ifndef SOMAR_AMD_H
#define SOMAR_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

/* 4: + cell-centred level projection, level projections on hierarchy levels, Helmholtz coefficients and heat
 *    integrators, operator / residual with the BC flag, MAC wall BC (additions only: version-3 callers are unaffected)
 * 5: + somar_amr_solve_host (multi-level host boundary), somar_k_gsrbiter3dortho (box-by-box kernel hook); additions only
 * 6: + somar_amr_tga_step (composite MappedAMRTGA::oneStep); composite operations with heat coefficients installed no
 *    longer fail (the flux-register scales follow beta); somar_solver_set_vel_bc (inflow / outflow sides); somar_solver_set_metric_map (cylindrical and bathymetric
 *    metric producers on the device); somar_k_fillmappedlapdiag3d, somar_k_mappedaverage2 (kernel-level hooks); additions only */
#define SOMAR_AMD_ABI_VERSION 11

/* BCType codes, calculus/BCInterface/BCDescriptor.H:34-39 */
#define SOMAR_BC_NONE (-1)
#define SOMAR_BC_NEUM 0
#define SOMAR_BC_DIRI 1

/* RelaxMode / PrecondMode, utils/ProblemContext.H:322-340 */
#define SOMAR_RELAX_JACOBI 0
#define SOMAR_RELAX_LEVEL_GSRB 1
#define SOMAR_RELAX_LINE_GSRB 3
#define SOMAR_PRECOND_NONE (-1)
#define SOMAR_PRECOND_DIAG_RELAX 0
#define SOMAR_PRECOND_DIAG_LINE_RELAX 1

typedef struct somar_solver somar_solver_t; /* opaque */

typedef struct somar_params {
    /* AMRMG.* (utils/ProblemContext.cpp:1147-1201) */
    int imin, imax;
    double eps, hang, norm_thresh;
    int num_smooth_down, num_smooth_up, num_smooth_bottom, num_smooth_precond;
    int num_mg, max_depth, precond_mode, relax_mode, verbosity;
    /* bottom.* (utils/ProblemContext.cpp:1207-1231) */
    int bottom_imax, bottom_num_restarts, bottom_norm_type, bottom_verbosity;
    double bottom_eps, bottom_reps, bottom_hang, bottom_small;
    /* CH_SPACEDIM of the build being replaced: 3, or 2 (domain and boxes one cell thick in z, z inactive;
     * the 2-D Fortran kernels GSRBITER2DORTHO / MAPPEDFLUXDIVERGENCE2D / FILLMAPPEDLAPDIAG2D are reproduced) */
    int space_dim;
} somar_params_t;

#define SOMAR_MAX_HISTORY 64
typedef struct somar_stats {
    int iters;
    int exit_status; /* !goRedu + 2*!goIter + 4*!goHang + 8*!goNorm */
    int status;      /* 0 ok, 1 "kaboom" (residual grew 10x), 2 "solver blew up" */
    int bottom_iters, bottom_exit;
    int nhistory;
    double initial_rnorm, final_rnorm;
    double history[SOMAR_MAX_HISTORY]; /* max-norm residual: [0] initial, [k] after V-cycle k; TRUNCATED at SOMAR_MAX_HISTORY
                                        * entries (nhistory = min(iters + 1, SOMAR_MAX_HISTORY)): iters, initial_rnorm and
                                        * final_rnorm are exact whatever imax is */
} somar_stats_t;

/* The COMPLETE residual history of the calling thread's last solve (any entry point that fills a somar_stats_t), for callers
 * whose AMRMG.imax exceeds SOMAR_MAX_HISTORY - 1: copies min(capacity, n) entries into out (may be NULL with capacity 0) and
 * stores the full length in *n.  (MappedAMRMultiGrid keeps no history at all; it prints each norm, MappedAMRMultiGrid.H:1150-1157.) */
int somar_last_history(double* out, int capacity, int* n);

/* resident field handles: which | (depth << 8) */
#define SOMAR_F_PHI 0     /* depth 0: the solution        */
#define SOMAR_F_RHS 1     /* depth 0: the right-hand side */
#define SOMAR_F_RES 2     /* residual   at this depth (depth 0: uberResidual)   */
#define SOMAR_F_CORR 3    /* correction at this depth (depth 0: uberCorrection) */
#define SOMAR_F_BEST 4    /* depth 0: bestPhi */
#define SOMAR_F_SCRATCH 5 /* per-depth scratch (fine residual before restriction) */
#define SOMAR_FIELD(depth, which) (((depth) << 8) | (which))

int somar_abi_version(void);
const char* somar_last_error(void);
int somar_device_count(int* count);
/* Synthetic inputs of the benchmark configurations: out[0 .. n) = successive draws of
 * std::uniform_real_distribution<double>(lo, hi) from std::mt19937_64(seed).  Host memory, host arithmetic (no GPU): the
 * SAME array then goes to the GPU path and to the CPU baseline of bench.py. */
int somar_host_fill_mt19937_64(double* out, long long n, unsigned long long seed, double lo, double hi);
int somar_params_default(somar_params_t* p);

/* boxes: nboxes*6 ints {lo0,lo1,lo2,hi0,hi1,hi2}; owner: rank per box or NULL (all rank 0);
 * bc_type: {loX,hiX,loY,hiY,loZ,hiZ}; comm: handle from somar_comm_* or NULL (single GPU). */
int somar_solver_create(somar_solver_t** out, const int* domain_lo, const int* domain_hi, const int* periodic,
                        const double* dx, const int* bc_type, int nboxes, const int* boxes, const int* owner,
                        double alpha, double beta, const somar_params_t* prm, void* comm);
int somar_solver_destroy(somar_solver_t* s);
int somar_solver_num_local_patches(somar_solver_t* s, int* n);
int somar_solver_patch_box(somar_solver_t* s, int depth, int patch, int* box6, int* global_index);
/* Diagonal metric of local patch `patch`: Jg^{aa} over faces(valid,a) (1 comp), Jinv over valid. */
int somar_solver_set_metric_ortho(somar_solver_t* s, int patch, const double* jg0, const double* jg1,
                                  const double* jg2, const double* jinv);
/* Values of the Dirichlet sides (bc_type SOMAR_BC_DIRI), {loX,hiX,loY,hiY,loZ,hiZ}, before finalize: the constants of
 * EllipticConstDiriBCGhostClass (BCInterface/EllipticBCUtils.H:114-147; ghost = 2 value - first cell, order 1,
 * EllipticBCUtilsF.ChF:71-84).  They enter the residuals of somar_solver_solve unless force_homogeneous is set; the
 * corrections always see zero.  Entries of Neumann / periodic sides are ignored.  A solver with Dirichlet sides runs
 * the two-pass GSRB and direct-load operator kernels (line relaxation included: Dirichlet vertical ends are folded into
 * the column systems); with a non-diagonal metric the Dirichlet ghosts are steps of the ghost programs. */
int somar_solver_set_bc_values(somar_solver_t* s, const double* values6);
/* Non-diagonal metric (LevelGeometry::isDiagonal() == false): jgD holds J g^{Db}, b = 0..2, over faces(valid, D),
 * component slowest (the FluxBox layout of LevelGeometry::getFCJgupPtr).  Selects the 19-point kernels
 * (GSRBITER3D, GSRBBOUNDARYITER3D, MAPPEDGETFLUX, fillExtrap / ExtrapolateFaceAndCopy, the cross-term Neumann
 * ghost of EllipticBCUtils.cpp:128-214).  One AMR level, LevelGSRB or Jacobi.  space_dim 2: jgD holds 2 components over
 * faces(valid, D), jg2 = NULL; the 9-point kernels GSRBITER2D / GSRBBOUNDARYITER2D (GSRBF.ChF:155-281, 858-1022) and
 * MAPPEDGETFLUX / ELLIPTICCONSTNEUMBCGHOST with CH_SPACEDIM = 2 are reproduced. */
int somar_solver_set_metric_full(somar_solver_t* s, int patch, const double* jg0, const double* jg1, const double* jg2,
                                 const double* jinv);
int somar_solver_finalize(somar_solver_t* s);
int somar_solver_depth(somar_solver_t* s, int* depth);
int somar_solver_mg_ref_ratio(somar_solver_t* s, int depth, int* r3); /* depth -> depth+1 */
int somar_solver_zero_avg(somar_solver_t* s, int depth, int* flag);
/* 1 when J g^{aa} and J^{-1} of this depth were found constant at finalize (a Cartesian map, CartesianMap.cpp:261-280):
 * the sweep / residual kernels then take the four values from their parameter block instead of streaming four arrays;
 * c4 (optional) = {J g^xx, J g^yy, J g^zz, J^{-1}}.  Environment SOMAR_NO_UNIFORM=1 turns the detection off. */
int somar_solver_metric_uniform(somar_solver_t* s, int depth, int* flag, double* c4);
int somar_solver_level_info(somar_solver_t* s, int depth, int* domain6, double* dx3, long long* cells,
                            long long* field_elems);

/* host <-> HBM, one local patch; host array spans valid.grow(ghost). */
int somar_field_upload(somar_solver_t* s, int field, int patch, const double* host, const int* ghost);
int somar_field_download(somar_solver_t* s, int field, int patch, double* host, const int* ghost);
int somar_field_set(somar_solver_t* s, int field, double value);
int somar_field_fill_hash(somar_solver_t* s, int field, unsigned long long seed);
/* f -= sum(f J)/sum(J) over the level: makes an all-Neumann/periodic rhs solvable
 * (computeMappedSum / setZeroAvg, MappedChombo/computeMappedSum.cpp) */
int somar_field_remove_mean(somar_solver_t* s, int field);
int somar_field_norm(somar_solver_t* s, int field, int ord, double* out);
int somar_field_dot(somar_solver_t* s, int field_a, int field_b, double* out);

/* the solve on resident PHI/RHS */
int somar_solver_solve(somar_solver_t* s, int zero_phi, int force_homogeneous, somar_stats_t* stats);
/* AMREllipticSolver::solve on caller-owned host LevelData (one pointer per local patch). */
int somar_solver_solve_host(somar_solver_t* s, double* const* phi, const int* phi_ghost,
                            const double* const* rhs, const int* rhs_ghost, int l_max, int l_base,
                            int zero_phi, int force_homogeneous, somar_stats_t* stats);

/* level-operator pieces on resident fields (all fields must belong to `depth`) */
int somar_level_relax(somar_solver_t* s, int depth, int phi_field, int rhs_field, int iters);
int somar_level_residual(somar_solver_t* s, int depth, int out_field, int phi_field, int rhs_field);
int somar_level_apply_op(somar_solver_t* s, int depth, int out_field, int phi_field);
/* the same two on depth 0 with the physical BCs' homogeneous flag exposed: applyOp(lhs, phi, a_homogeneous) /
 * residual(lhs, phi, rhs, a_homogeneous) (MappedAMRPoissonOp.cpp:740-765, 628-677).  With homogeneous = 0 and Dirichlet
 * sides this is also one component of VelocityAMRPoissonOp::applyOpI with viscous solid walls, where every component's
 * VelBCHolder entry is a constant Dirichlet value (AMRElliptic/VelocityAMRPoissonOp.cpp:64-166,
 * BasicVelocityBCGhostClass EllipticBCUtils.cpp:1284-1306): the explicit viscous source is one call per component. */
int somar_level_apply_op_bc(somar_solver_t* s, int out_field, int phi_field, int homogeneous);
int somar_level_residual_bc(somar_solver_t* s, int out_field, int phi_field, int rhs_field, int homogeneous);
int somar_level_restrict_residual(somar_solver_t* s, int depth, int coarse_res_field, int phi_field, int rhs_field);
int somar_level_prolong_increment(somar_solver_t* s, int depth, int phi_field, int coarse_corr_field);
int somar_level_precond(somar_solver_t* s, int depth, int phi_field, int rhs_field);
int somar_vcycle(somar_solver_t* s, int corr_field, int res_field);
/* the same cycle started from a ZERO correction, as MappedAMRMultiGrid::solveNoInitResid does every iteration
 * (uberCorrection is setToZero'ed in postVCycleOps, MappedAMRMultiGrid.H:1203): the contents of corr_field are
 * ignored and overwritten, which saves the memset and the first sweep's read */
int somar_vcycle_from_zero(somar_solver_t* s, int corr_field, int res_field);
/* MappedAMRMultiGrid::relax on a level refined by more than 2 against its coarser level (MappedAMRMultiGrid.H:742-754):
 * one V-cycle over the forced MG depths only (define :1455-1482), smoothing but no solve at its bottom.  Only on
 * such a level of a somar_amr hierarchy. */
int somar_mini_vcycle(somar_solver_t* s, int corr_field, int res_field);
int somar_bottom_solve(somar_solver_t* s, int phi_field, int rhs_field, int* iters, int* exit_code);
/* how the LAST bottom solve ran (Chombo BiCGStabSolver as AMRPressureSolver.cpp:253-265 configures it): 0 = launch by launch,
 * 1 = one single-workgroup launch (bottoms of at most 512 cells), 2 = one persistent launch with one workgroup per box and
 * device-wide barriers (multi-box bottoms; SOMAR_BOX_BOTTOM=0 switches it off).  Same iterates on every path. */
int somar_bottom_kind(somar_solver_t* s, int* kind);
/* which of the library's alternative execution paths have run on this solver so far (tests assert that the path under test
 * is the one that ran): out4 = {fused sweeps whose ghost exchange travelled on the second stream under their interior tiles,
 * ghost programs executed as one launch (a workgroup per box), ghost programs executed stage by stage, bottom solves} */
int somar_solver_counters(somar_solver_t* s, long long* out4);
/* sweeps of LevelGSRB with a non-diagonal metric (GSRB.cpp:58-98 with GSRBITER3D) that ran as ONE red+black marching launch plus
 * a shell pass (levels of large boxes; csrc/full19_fused.hip) instead of two colour passes */
int somar_solver_fused19_sweeps(somar_solver_t* s, long long* n);

/* MAC level projection of a face-centred velocity given in flux form (J u^a on a-faces, one host array per
 * local patch spanning faces(valid, a)):  rhs = div(U)/dt ; solve ; U -= dt * Jg^{aa} d_a(phi).
 *   BaseProjector<FluxBox>::project, a_velIsFlux = true      projection/BaseProjectorI.H:176-299
 *   LevelMACProjector::computeDiv/computeGrad/applyCorrection  projection/LevelMACProjector.cpp:156-241
 *   Divergence::levelDivergenceMAC                              calculus/DivCurlGrad/Divergence.cpp:44-127
 *   Gradient::levelGradientMAC (order-2 extrapolated ghosts)    calculus/DivCurlGrad/Gradient.cpp:85-206
 * Boundary-face velocities are taken as given; somar_vel_wall_bc applies what levelDivergenceMAC applies through its
 * a_fluxBC = uStarFuncBC when no side is an inflow / outflow side (BCutil/PhysBCUtil.cpp:793-801, 1261-1276): solid walls,
 * BasicVelocityBCGhostClass -> setSideDiriBC(0) on the wall-normal faces of the resident velocity, in place as in the
 * reference (Divergence.cpp:73-100, EllipticBCUtils.cpp:1284-1327, 96-100).  Call it between upload and projection. */
int somar_vel_wall_bc(somar_solver_t* s);
/* Inflow / outflow sides of BasicVelocityBCGhostClass (calculus/BCInterface/EllipticBCUtils.cpp:1244-1327; chosen by
 * PhysBCUtil::basicVelFuncBC, BCutil/PhysBCUtil.cpp:1261-1276) for the face-centred velocity: kind[2*dir + side] = 0 solid
 * wall (setSideDiriBC(0), the default), 1 prescribed normal velocity value[2*dir + side] (setSideDiriBC(inflowVel), an
 * inflow side, or the class's viscous walls of the inflow component), 2 outflow (setSideExtrapBC order 0: the boundary face
 * takes the next face inside, EllipticBCUtilsF.ChF:148-154).  From then on somar_vel_wall_bc and the face BC inside the
 * cell-centred divergence (somar_level_divergence_cc / somar_cc_project / the AMR projections with wall_bc = 1) apply these
 * instead of solid walls on every side.  Periodic directions are never touched.  Values are in the velocity's own form
 * (flux form J u^a when the projector runs with velIsFlux = true), exactly as the reference writes inflowVel into it. */
int somar_solver_set_vel_bc(somar_solver_t* s, const int* kind, const double* value);
int somar_vel_upload(somar_solver_t* s, int dir, int patch, const double* host);
int somar_vel_download(somar_solver_t* s, int dir, int patch, double* host);
int somar_level_divergence_mac(somar_solver_t* s, int out_field, double dt);
int somar_level_mac_correct(somar_solver_t* s, int phi_field, double dt);
int somar_mac_project(somar_solver_t* s, double dt, int zero_pressure, int force_homogeneous, somar_stats_t* stats);
int somar_mac_project_host(somar_solver_t* s, double* const* u0, double* const* u1, double* const* u2, double dt,
                           int zero_pressure, int force_homogeneous, somar_stats_t* stats);

/* Velocities that are NOT in flux form: BaseProjector::project(..., a_velIsFlux = false) multiplies the velocity by J before
 * the projection and divides it by J afterwards (projection/BaseProjectorI.H:235-241, 291-297) with
 * LevelGeometry::multByJ / divByJ (geometry/LevelGeometryUtil.cpp:287-339, 372-420, 456-...): data *= J, resp. data *= Jinv
 * (a multiplication by the cached 1/J, not a division).  Here the scale arrays become resident once --
 *   somar_solver_set_cc_j   J and Jinv at cell centres over valid.grow(ghost), ghost >= 1 (getCCJ / getCCJinv)
 *   somar_solver_set_face_j J and Jinv on faces(valid, dir) (what fill_J / fill_Jinv put on the face box)
 * -- and somar_vel_mult_by_j / somar_vel_div_by_j scale the resident velocity on the device (centring 0: the MAC velocity,
 * every direction; 1: the cell-centred velocity, every component, ghost layer included), so a non-flux velocity costs
 * two launches around the projection instead of a round trip through the host.  After finalize. */
int somar_solver_set_cc_j(somar_solver_t* s, int patch, const double* J, const double* Jinv, const int* ghost);
int somar_solver_set_face_j(somar_solver_t* s, int dir, int patch, const double* J, const double* Jinv);
int somar_vel_mult_by_j(somar_solver_t* s, int centring);
int somar_vel_div_by_j(somar_solver_t* s, int centring);

/* Viscous / diffusive Helmholtz solves through the same operator (single level).
 * somar_solver_set_alpha_beta = MappedAMRPoissonOp::setAlphaAndBeta (AMRElliptic/MappedAMRPoissonOp.cpp:582-619) applied to
 * every op of the hierarchy as MappedBaseLevelHeatSolver::resetSolverAlphaAndBeta does (AMRParabolic/
 * MappedBaseLevelHeatSolver.cpp:257-270): alpha = a * aCoef, beta = b * bCoef, aCoef / bCoef being the alpha / beta
 * handed to somar_solver_create (the factory's, MappedAMRPoissonOpFactory.cpp:585-586).  The prolongation strategy
 * chosen by the null-space probe at finalize is kept, as in the reference.
 * somar_heat_step = one level time step of d(phi)/dt = L[phi] + src:
 *   scheme 0  MappedLevelBackwardEuler::updateSoln   (AMRParabolic/MappedLevelBackwardEuler.cpp:52-158)
 *             (aCoef I - dt bCoef L) phiNew = phiOld          (the reference leaves the source out of this scheme)
 *   scheme 1  MappedLevelCrankNicolson::updateSoln   (AMRParabolic/MappedLevelCrankNicolson.cpp:52-152)
 *             (aCoef I - dt/2 bCoef L) phiNew = dt src + (aCoef I + dt/2 bCoef L) phiOld
 *   scheme 2  MappedLevelTGA::updateSolnWithTimeIndependentOp (AMRParabolic/MappedLevelTGA.cpp:231-387, coefficients :30-56)
 *             (I - mu1 dt L)(I - mu2 dt L) phiNew = (I + mu3 dt L) phiOld + (I + mu4 dt L) dt src (source half with
 *             homogeneous BCs); two solves, both from the caller's initial guess; stats are the second solve's
 * phiNew is SOMAR_F_PHI (the initial guess unless zero_phi), phiOld SOMAR_F_HEAT_OLD, src SOMAR_F_HEAT_SRC; boundary
 * conditions are the solver's (viscousSolveFuncBC = constant Dirichlet values on solid walls: bc_type SOMAR_BC_DIRI +
 * somar_solver_set_bc_values; BCutil/PhysBCUtil.cpp:822-826).  The flux-register increments that follow the solve in the
 * reference only matter with a finer or coarser level and are not part of this single-level entry point. */
#define SOMAR_F_HEAT_OLD 8 /* depth 0: phi at the old time */
#define SOMAR_F_HEAT_SRC 9 /* depth 0: the source term */
int somar_solver_set_alpha_beta(somar_solver_t* s, double a, double b);
int somar_heat_step(somar_solver_t* s, int scheme, double dt, int zero_phi, somar_stats_t* stats);

/* Cell-centred level projection of a velocity given in flux form (J u at cell centres; ONE host FArrayBox per local
 * patch with SpaceDim components, component slowest, defined on valid grown by ghost[] >= 1 in every active direction;
 * the ghost layer is used as the caller filled it -- the reference does not exchange it either):
 *   rhs = div(CellToEdge(U))/dt with zero normal flux on solid walls ; solve ; U -= dt * EdgeToCell(Jg^{ab} d_b(phi)).
 *   BaseProjector<FArrayBox>::project, a_velIsFlux = true      projection/BaseProjectorI.H:176-299
 *   LevelCCProjector::computeDiv/computeGrad/applyCorrection    projection/LevelCCProjector.cpp:163-255
 *   Divergence::levelDivergenceCC (CellToEdge + levelDivergenceMAC)   calculus/DivCurlGrad/Divergence.cpp:361-396
 *   Gradient::levelGradientCC (levelGradientMAC + EdgeToCell)         calculus/DivCurlGrad/Gradient.cpp:469-495
 * wall_bc != 0: the velocity BC of uStarFuncBC with no inflow/outflow side (BCutil/PhysBCUtil.cpp:793-801, 1261-1276):
 * BasicVelocityBCGhostClass's solid wall = setSideDiriBC(0) on the averaged normal faces of every non-periodic side
 * (calculus/BCInterface/EllipticBCUtils.cpp:1284-1327, 96-100).  wall_bc == 0: levelDivergenceCC with a_fluxBC = NULL.
 * Only the valid cells of the host arrays are written back.  Single level (no coarse-fine velocity interpolation). */
int somar_ccvel_upload(somar_solver_t* s, int patch, const double* host, const int* ghost);
int somar_ccvel_download(somar_solver_t* s, int patch, double* host, const int* ghost);
int somar_level_divergence_cc(somar_solver_t* s, int out_field, double dt, int wall_bc);
int somar_level_cc_correct(somar_solver_t* s, int phi_field, double dt);
int somar_cc_project(somar_solver_t* s, double dt, int zero_pressure, int force_homogeneous, int wall_bc,
                     somar_stats_t* stats);
int somar_cc_project_host(somar_solver_t* s, double* const* vel, const int* ghost, double dt, int zero_pressure,
                          int force_homogeneous, int wall_bc, somar_stats_t* stats);

/* Kernel-level, box-by-box hook with the argument shapes the reference's Fortran exports: one colour
 * pass of GSRBITER3DORTHO (RelaxationMethods/GSRBF.ChF:545-701; prototype RelaxationMethods/GSRBF_F.H:216-232) on HOST FABs.
 * Chombo's FORT_PROTO expansion: every argument by pointer; CHFp_FRA(a) = Real* a, const int* ialo0, ialo1, ialo2, iahi0,
 * iahi1, iahi2, const int* nacomp; CHFp_CONST_FRA1(a) the same without the component count; CHFp_BOX(b) = six const int*;
 * CHFp_CONST_REALVECT = const Real* (3 values); CHFp_CONST_REAL / _INT = const Real* / const int*.  Arrays column-major,
 * inclusive bounds, component slowest.  phi is updated in place on `region` (cells of the colour only); it must be
 * defined one cell around the region, rhs / Jinv / lapDiag on the region, Jg^{aa} on its a-faces.  This is a parity-test
 * hook (it moves the box to the GPU and back per call), not a production path: production keeps levels resident. */
int somar_k_gsrbiter3dortho(double* phi, const int* iphilo0, const int* iphilo1, const int* iphilo2, const int* iphihi0,
                            const int* iphihi1, const int* iphihi2, const int* nphicomp, const double* rhs,
                            const int* irhslo0, const int* irhslo1, const int* irhslo2, const int* irhshi0,
                            const int* irhshi1, const int* irhshi2, const int* nrhscomp, const double* Jgxx,
                            const int* iJgxxlo0, const int* iJgxxlo1, const int* iJgxxlo2, const int* iJgxxhi0,
                            const int* iJgxxhi1, const int* iJgxxhi2, const double* Jgyy, const int* iJgyylo0,
                            const int* iJgyylo1, const int* iJgyylo2, const int* iJgyyhi0, const int* iJgyyhi1,
                            const int* iJgyyhi2, const double* Jgzz, const int* iJgzzlo0, const int* iJgzzlo1,
                            const int* iJgzzlo2, const int* iJgzzhi0, const int* iJgzzhi1, const int* iJgzzhi2,
                            const double* Jinv, const int* iJinvlo0, const int* iJinvlo1, const int* iJinvlo2,
                            const int* iJinvhi0, const int* iJinvhi1, const int* iJinvhi2, const double* lapDiag,
                            const int* ilapDiaglo0, const int* ilapDiaglo1, const int* ilapDiaglo2, const int* ilapDiaghi0,
                            const int* ilapDiaghi1, const int* ilapDiaghi2, const int* iregionlo0, const int* iregionlo1,
                            const int* iregionlo2, const int* iregionhi0, const int* iregionhi1, const int* iregionhi2,
                            const double* dx, const double* alpha, const double* beta, const int* redBlack);
/* Two more of the Fortran exports, same conventions (parity-test hooks): FILLMAPPEDLAPDIAG3D (AMRElliptic/
 * MappedAMRPoissonOpF.ChF:233-274, prototype MappedAMRPoissonOpF_F.H:139-146: lapDiag on `region` from component d of the
 * direction-d FluxBox FAB and 1/J) and MAPPEDAVERAGE2 (MappedChombo/MappedCoarseAverageF.ChF:132-167, prototype
 * MappedCoarseAverageF_F.H:111-117: the J-weighted restriction coarse = sum(fine / Jinv) / sum(1 / Jinv) over the refRatio
 * block of every cell of `box`; bref = [0, refRatio - 1]). */
int somar_k_fillmappedlapdiag3d(double* lapDiag, const int* ilapDiaglo0, const int* ilapDiaglo1, const int* ilapDiaglo2,
                                const int* ilapDiaghi0, const int* ilapDiaghi1, const int* ilapDiaghi2, const double* Jg0,
                                const int* iJg0lo0, const int* iJg0lo1, const int* iJg0lo2, const int* iJg0hi0,
                                const int* iJg0hi1, const int* iJg0hi2, const int* nJg0comp, const double* Jg1,
                                const int* iJg1lo0, const int* iJg1lo1, const int* iJg1lo2, const int* iJg1hi0,
                                const int* iJg1hi1, const int* iJg1hi2, const int* nJg1comp, const double* Jg2,
                                const int* iJg2lo0, const int* iJg2lo1, const int* iJg2lo2, const int* iJg2hi0,
                                const int* iJg2hi1, const int* iJg2hi2, const int* nJg2comp, const double* Jinv,
                                const int* iJinvlo0, const int* iJinvlo1, const int* iJinvlo2, const int* iJinvhi0,
                                const int* iJinvhi1, const int* iJinvhi2, const int* iregionlo0, const int* iregionlo1,
                                const int* iregionlo2, const int* iregionhi0, const int* iregionhi1, const int* iregionhi2,
                                const double* dx);
int somar_k_mappedaverage2(double* coarse, const int* icoarselo0, const int* icoarselo1, const int* icoarselo2,
                           const int* icoarsehi0, const int* icoarsehi1, const int* icoarsehi2, const int* ncoarsecomp,
                           const double* fine, const int* ifinelo0, const int* ifinelo1, const int* ifinelo2,
                           const int* ifinehi0, const int* ifinehi1, const int* ifinehi2, const int* nfinecomp,
                           const double* fineCCJinv, const int* ifineCCJinvlo0, const int* ifineCCJinvlo1,
                           const int* ifineCCJinvlo2, const int* ifineCCJinvhi0, const int* ifineCCJinvhi1,
                           const int* ifineCCJinvhi2, const int* iboxlo0, const int* iboxlo1, const int* iboxlo2,
                           const int* iboxhi0, const int* iboxhi1, const int* iboxhi2, const int* refRatio,
                           const int* ibreflo0, const int* ibreflo1, const int* ibreflo2, const int* ibrefhi0,
                           const int* ibrefhi1, const int* ibrefhi2);


/* stream control + HIP-event timing on the solver's own stream */
int somar_sync(somar_solver_t* s);
int somar_timer_start(somar_solver_t* s);
int somar_timer_stop(somar_solver_t* s, double* milliseconds);
/* per-launch HIP events: kernel 0 = the depth-0 GSRB launches, 1 = the depth-0 operator / residual launches, 2 = ghost exchanges
 * with other ranks on any depth (pack + grouped send/recv + unpack; LevelData::exchange, MappedAMRPoissonOp.cpp:2222-2238),
 * 3 = the replicated coarse tail of a sharded hierarchy.  While profiling nothing is overlapped or graph-replayed. */
int somar_profile_enable(somar_solver_t* s, int on);
int somar_profile_get(somar_solver_t* s, int kernel, int* launches, double* total_ms);

/* Pure host planning, no GPU needed: the ghost-exchange plan of `rank` for a sharded layout =
 * Chombo's Copier(grids, grids, domain, ghost, exchange=true) (MappedAMRPoissonOpFactory.cpp:161-164).
 * Each item is 12 ints {src_box, dst_box, src_lo[3], dst_lo[3], n[3], peer_rank} (box-local starts).
 * Send and receive items are grouped per peer in ascending peer order; within a pair both sides use the
 * same item order.  Exposed so the N>1 data path can be rehearsed on CPUs (tests/test_multirank_cpu.py). */
int somar_plan_exchange(const int* domain_lo, const int* domain_hi, const int* periodic, int nboxes, const int* boxes,
                        const int* owner, int rank, int ghost, int max_items, int* n_local, int* local_items,
                        int* n_send, int* send_items, int* n_recv, int* recv_items);

/* ------------------------------------------------------------------------------------------------
 * Several AMR levels (refinement ratios with entries 1 or 2).
 *   somar_amr_create/_finalize    AMRPressureSolver::define (projection/AMRPressureSolver.cpp:272-491) ->
 *                                 MappedAMRPoissonOpFactory::define + MappedAMRMultiGrid::define
 *                                 (calculus/AMRElliptic/MappedAMRMultiGrid.H:1407-1490)
 *   somar_amr_level               the per-level operator (MappedAMRMultiGrid::levelOp, :784): a BORROWED
 *                                 somar_solver_t for metric upload, field I/O and the single-level pieces
 *   somar_amr_solve               MappedAMRMultiGrid::solve(phi, rhs, l_max, l_base, zeroPhi, forceHomogeneous)
 *                                 (:933-1183) on the levels' resident PHI/RHS; with l_base > 0 the PHI of level
 *                                 l_base-1 supplies the coarse-fine boundary values (AMRPressureSolver::levelSolve,
 *                                 projection/AMRPressureSolver.cpp:567-594)
 *   somar_amr_interp_cf           MappedAMRPoissonOp::interpCFGhosts(phi, &phiCoarse, false)  (:2170-2216) =
 *                                 MappedQuadCFInterp::coarseFineInterp (MappedChombo/MappedQuadCFInterp.cpp:579-622)
 *   somar_amr_residual_level      MappedAMRMultiGrid::computeAMRResidualLevel (:884-927) incl. refluxing
 *                                 (MappedAMRPoissonOp::reflux :1615-1707, MappedLevelFluxRegister)
 *   somar_amr_zero_covered        MappedAMRLevelOp::zeroCovered
 *   somar_amr_vcycle              MappedAMRMultiGrid::AMRVCycle (:1498-1597) on every level's CORR / RES
 * Level l's boxes are given in level-l index space; `boxes`/`owner` list level 0 first. */
typedef struct somar_amr somar_amr_t; /* opaque */
#define SOMAR_F_AMR_CORR 6 /* depth 0: m_correction of MappedAMRMultiGrid */
#define SOMAR_F_AMR_RES 7  /* depth 0: m_residual   of MappedAMRMultiGrid */
int somar_amr_create(somar_amr_t** out, int nlevels, const int* domain_lo, const int* domain_hi, const int* periodic,
                     const double* dx0, const int* bc_type, const int* ref_ratios, const int* nboxes,
                     const int* boxes, const int* owner, double alpha, double beta, const somar_params_t* prm,
                     void* comm);
int somar_amr_destroy(somar_amr_t* a);
int somar_amr_level(somar_amr_t* a, int level, somar_solver_t** out);
int somar_amr_finalize(somar_amr_t* a);
int somar_amr_solve(somar_amr_t* a, int l_max, int l_base, int zero_phi, int force_homogeneous, somar_stats_t* stats);
/* The same solve on CALLER-OWNED host data of several levels: the primary boundary for AMR hierarchies,
 *   AMREllipticSolver<LevelData<FArrayBox>>::solve(Vector<T*>& phi, const Vector<T*>& rhs, l_max, l_base, zeroPhi,
 *   forceHomogeneous)  (calculus/AMRElliptic/AMREllipticSolver.H:33-48), reached from AMRPressureSolver::solve
 *   (projection/AMRPressureSolver.cpp:494-561; note its call order solve(phi, rhs, a_lmax, a_lmin, ...) :529-534) and
 *   levelSolve (:567-594).
 * phi[l] / rhs[l]: one pointer per LOCAL patch of level l (host FABs over valid.grow(ghost)); entries of levels outside
 * [l_base, l_max] may be NULL, except phi[l_base-1] when l_base > 0 (it supplies the coarse-fine boundary values, as
 * the reference's a_phi[l_base-1] does).  Uploads rhs (and phi unless zero_phi; phi[l_base-1] always), runs
 * somar_amr_solve, writes phi of l_base..l_max back (valid cells and the ghost layer the solver leaves).  Same stats,
 * exit status and best-phi semantics as somar_amr_solve. */
int somar_amr_solve_host(somar_amr_t* a, double* const* const* phi, const int* phi_ghost, const double* const* const* rhs,
                         const int* rhs_ghost, int l_max, int l_base, int zero_phi, int force_homogeneous,
                         somar_stats_t* stats);
int somar_amr_interp_cf(somar_amr_t* a, int level, int fine_field, int coarse_field);
/* Level projection on level `level` of the hierarchy, the coarser level supplying the coarse-fine values:
 * BaseProjector<T>::levelProject -> project(lmin = lmax = level) (projection/BaseProjectorI.H:176-366).
 *   centring 0  LevelMACProjector (projection/LevelMACProjector.cpp:156-241): the level's somar_vel_upload'ed face velocity
 *   centring 1  LevelCCProjector  (projection/LevelCCProjector.cpp:163-255): the level's somar_ccvel_upload'ed velocity; level-1's
 *               cell-centred velocity feeds m_velCFInterp.coarseFineInterp (calculus/DivCurlGrad/Divergence.cpp:372-375)
 * SOMAR_F_PHI of level-1 feeds the level solve (AMRPressureSolver::levelSolve's a_crsePhiPtr, AMRPressureSolver.cpp:567-594)
 * and levelGradientMAC's coarseFineInterp (Gradient.cpp:106-113).  Velocities in flux form.  On a refined level the
 * metric must be diagonal (error otherwise; level 0 takes either). */
int somar_amr_level_project(somar_amr_t* a, int level, int centring, double dt, int zero_pressure, int force_homogeneous,
                            int wall_bc, somar_stats_t* stats);
/* Solver inspector: MappedAMRMultiGridInspector<T>::recordResiduals / recordCorrections (calculus/AMRElliptic/
 * MappedAMRMultiGrid.H:260-298, called at :1064-1065 and :1083-1084).  somar_amr_solve[_host] calls fn(user, kind, iter,
 * l_min, l_max) with every stream drained: kind 0 before V-cycle `iter` -- each level's SOMAR_F_RES holds uberResidual --
 * and kind 1 after it -- SOMAR_F_CORR holds uberCorrection.  Inside the callback the fields can be read through the level
 * handles (somar_amr_level + somar_field_download): that is what OutputMappedAMRMultiGridInspector (:305-362) hands to
 * outputAMR / HDF5, so a reference build can write the same "name.residual.iter.N.hdf5" files from this solver's data, and
 * tools/inspect_solve.py dumps them as .npz for diffing against such files.  fn = NULL removes the inspector. */
typedef void (*somar_inspector_fn)(void* user, int kind, int iter, int l_min, int l_max);
int somar_amr_set_inspector(somar_amr_t* a, somar_inspector_fn fn, void* user);

/* The COMPOSITE cell-centred projection over levels l_min..l_max -- SOMAR's sync / initialisation / post-regrid projection
 * (NavierStokes/AMRNavierStokesSync.cpp:280-295), velocities in flux form (J u at cell centres, the levels'
 * somar_ccvel_upload'ed fields; level l_min-1's velocity and pressure, if that level exists, supply coarse-fine values):
 *   BaseProjector<FArrayBox>::project(lmin, lmax)                  projection/BaseProjectorI.H:176-299
 *   AMRCCProjector::computeDiv / computeGrad / applyCorrection     projection/AMRCCProjector.cpp:204-377
 *   Divergence::compDivergenceCC: CF interpolation of the velocity, CellToEdge, wall BC, level divergence, and the
 *       coarse-fine mismatch refluxed through MappedLevelFluxRegister      calculus/DivCurlGrad/Divergence.cpp:697-838
 *   Gradient::compGradientCC: levelGradientMAC, one-sided faces next to the finer level (CRSEONESIDEGRAD,
 *       DivCurlGradF.ChF:626-697, mask of calculus/DivCurlGrad/Mask.cpp), EdgeToCell     Gradient.cpp:707-842
 *   correction from the finest level down, each level then averaged onto the next coarser one
 *       (MappedCoarseAverage::averageToCoarse, unweighted: UNMAPPEDAVERAGE)  AMRCCProjector.cpp:334-377
 * rhs = compDiv / dt (not divided when dt == 0) goes to SOMAR_F_RHS, the pressure is left in SOMAR_F_PHI of each level.
 * The pieces are exposed for parity tests: somar_amr_comp_divergence_cc writes compDivergenceCC of one level (NOT divided
 * by dt) to out_field; somar_amr_comp_grad_correct_cc does ccvel(level) += (dt == 0 ? -1 : -dt) * compGradientCC(phi_field
 * of that level; PHI of level-1 / level+1 as coarse / fine data); somar_amr_average_down_ccvel overwrites the cells of
 * `level` under level+1 with the plain average of level+1's velocity.  The pressure operator must have beta = 1. */
int somar_amr_cc_project(somar_amr_t* a, int l_min, int l_max, double dt, int zero_pressure, int force_homogeneous,
                         int wall_bc, somar_stats_t* stats);
int somar_amr_comp_divergence_cc(somar_amr_t* a, int level, int l_max, int out_field, int wall_bc);
int somar_amr_comp_grad_correct_cc(somar_amr_t* a, int level, int l_max, int phi_field, double dt);
int somar_amr_average_down_ccvel(somar_amr_t* a, int level);
int somar_amr_residual_level(somar_amr_t* a, int l_max, int l_base, int ilev, int res_field, int phi_field,
                             int rhs_field);
int somar_amr_zero_covered(somar_amr_t* a, int level, int field);
int somar_amr_vcycle(somar_amr_t* a, int l_max, int l_base);

/* ---- AlteredMetric ------------------------------------------------------------------------------------------------
 * Replaces the algebra of AlteredMetric::fill_Jgup (projection/AlteredMetric.cpp:82-198), the FillJgupInterface that
 * the implicit-gravity / Coriolis projections install in the operator factory:
 *   dest = J * ( g^{mu nu} / (1 + f~^2) + ( f~^2/(1+f~^2) - w^2/(1+w^2) ) dXi^mu/dz dXi^nu/dz [+ f~/(1+f~^2) (ix jy - iy jx)] )
 * with w^2 = (dt theta)^2 N^2, f~ = f dt theta, one value per face of the destination face box (n faces, flat arrays).
 * The caller evaluates its GeoSourceInterface / background scalar on that box (map evaluation is out of scope):
 * nsq_fc = N^2 averaged to the faces (Chombo CellToEdge of :114-137's NsqFAB), dximu_dz / dxinu_dz = fill_dXidx(mu|nu,
 * SpaceDim-1), gup = fill_gup(mu, nu), J = fill_J(scale); ix, jy, iy, jx = fill_dXidx(mu,0), (nu,1), (mu,1), (nu,0),
 * all four NULL when mu == nu.  The result feeds somar_solver_set_metric_ortho / _full. */
int somar_altered_jgup(long long n, double* dest, const double* nsq_fc, const double* dximu_dz, const double* dxinu_dz,
                       const double* ix, const double* jy, const double* iy, const double* jx, const double* gup,
                       const double* J, double dt_theta, double coriolis_f);

/* ---- leptic level solver -------------------------------------------------------------------------------------
 * Replaces LevelLepticSolver (calculus/LepticSolver/LevelLepticSolver.H:53-347): define(op) :147-437 and
 * solve(phi, rhs) :646-956, for an operator that offers what LepticOperator asks (LepticOperator.H:33-45).
 * The level's boxes must not be split in the vertical (the layout LepticBoxUtils::createVerticalSolverGrids makes): every
 * column ends at a physical boundary or, on a level of a hierarchy, at a coarse-fine interface.  gatherVerticalBCTypes
 * (:1523-1640) decides from the ends: columns that are Neumann at both ends take the Neumann-Neumann line solver and the
 * horizontal (flat) problem; when NO column is Neumann-Neumann -- a Dirichlet (free-surface) top or bottom, or columns ending
 * under the coarser level -- every order is one LepticLapackVerticalSolver + dptsv pass (LevelLepticSolverF.ChF:161-283) and
 * there is no flat problem; a layout mixing the two kinds is refused.  Diagonal or non-diagonal metric
 * (somar_solver_set_metric_full on the level handle), homogeneous-Neumann lateral boundaries, no periodic direction;
 * coarse-fine boundaries through somar_amr_enable_leptic.
 * somar_leptic_params_t = setParameters / setHorizMGParameters / setHorizBottomParameters / setFullMGParameters /
 * setFullBottomParameters (:516-640); defaults = setDefaultParameters (:461-508). */
typedef struct somar_leptic somar_leptic_t; /* opaque */
typedef struct somar_leptic_params {
    int max_order, norm_type;
    double hang, horiz_rhs_tol;
    double domain_height; /* LevelGeometry::getDomainLength(SpaceDim-1); 0: dz * Nz */
    somar_params_t horiz; /* flat (SpaceDim-1) multigrid + its BiCGStab */
    somar_params_t full;  /* full 3-D multigrid used when the last order hangs + its BiCGStab */
} somar_leptic_params_t;
typedef struct somar_leptic_stats {
    int exit_status; /* LevelLepticSolver::ExitStatus: 0 CONVERGE 1 ITER 2 HANG 3 DIVERGE 4 KABOOM */
    int orders, horiz_solves, used_full_solver;
    int nres;
    double res_norms[SOMAR_MAX_HISTORY]; /* m_resNorms: [0] |J * initial residual|, [k] after order k-1 */
    somar_stats_t horiz, full;           /* last horizontal / full multigrid solve */
} somar_leptic_stats_t;
int somar_leptic_params_default(somar_leptic_params_t* p);
/* level_prm: parameters of the level's own operator/multigrid (the op handed to LevelLepticSolver::define) */
int somar_leptic_create(somar_leptic_t** out, const int* domain_lo, const int* domain_hi, const int* periodic,
                        const double* dx, const int* bc_type, int nboxes, const int* boxes, const int* owner,
                        double alpha, double beta, const somar_params_t* level_prm, const somar_leptic_params_t* lp,
                        void* comm);
int somar_leptic_destroy(somar_leptic_t* h);
/* the level's own operator: set its metric (somar_solver_set_metric_ortho) and move phi / rhs through it; do not
 * finalize or destroy it -- somar_leptic_finalize / _destroy do */
int somar_leptic_level(somar_leptic_t* h, somar_solver_t** level);
/* read-only views of the two internal solvers after a solve (tests, diagnostics): which = 1 the J-scaled operator
 * with the full 3-D multigrid (m_opPtr / m_mgSolverPtr; its PHI holds the last order's correction), which = 2 the
 * flat multigrid (m_horizSolverPtr; PHI / RHS = last horizontal solution / right-hand side) */
int somar_leptic_part(somar_leptic_t* h, int which, somar_solver_t** solver);
int somar_leptic_finalize(somar_leptic_t* h);
/* phi += leptic correction for L[phi] = rhs on the level's resident phi / rhs */
int somar_leptic_solve(somar_leptic_t* h, int homogeneous, somar_leptic_stats_t* stats);

/* The level heat integrators on level `level` of a hierarchy (multi-level part):
 * MappedLevelBackwardEuler / MappedLevelCrankNicolson::updateSoln, MappedLevelTGA::updateSolnWithTimeIndependentOp
 * (AMRParabolic/MappedLevelBackwardEuler.cpp:52-158, MappedLevelCrankNicolson.cpp:52-152, MappedLevelTGA.cpp:231-387) with
 * applyHelm -> AMROperatorNF, solveHelm -> m_solver->solve(phi, rhs, level, level), timeInterp
 * (MappedBaseLevelHeatSolver.cpp:154-300).  Level handles (somar_amr_level): phiNew = SOMAR_F_PHI (the guess unless
 * zero_phi), phiOld = SOMAR_F_HEAT_OLD, src = SOMAR_F_HEAT_SRC of `level`; for level > 0 the coarse-fine values are the
 * linear interpolation in time of level-1's SOMAR_F_HEAT_OLD (a_crsePhiOldPtr, at crse_old_time) and SOMAR_F_PHI
 * (a_crsePhiNewPtr, at crse_new_time) -- i.e. step level-1 first, then `level`, as the subcycled advance does.  scheme 0 / 1 /
 * 2 as somar_heat_step; stats = the LAST solve's.  a_flux (incrementFlux: J Grad(phi) of every stage, without beta,
 * MappedAMRPoissonOp.cpp:2129-2151) accumulates on the level and is read with somar_heat_flux_download, so the adapter can
 * run MappedLevelFluxRegister::incrementCoarse / incrementFine on its own registers as before; faces on the domain boundary
 * are not meaningful (no register reads them).
 * somar_amr_set_alpha_beta = resetSolverAlphaAndBeta on every op of every level.  The hierarchy's own flux-register scales
 * follow the coarse operator's beta the next time a composite operation refluxes (MappedAMRPoissonOp::reflux takes
 * m_beta / m_dx when it runs, MappedAMRPoissonOp.cpp:1661, 1693), so composite solves work with any coefficients installed.
 * somar_amr_tga_step = MappedAMRTGA<T>::oneStep (AMRElliptic/MappedAMRTGA.H:417-497), the COMPOSITE TGA step over levels
 * l_base..l_max in one call: applyHelm = computeAMROperator with (1, mu dt) (MappedAMRMultiGrid.H:862-878), solveHelm =
 * solveNoInit(..., zeroPhi = false) with (1, -mu dt), the guess of both solves is phiOld (:473-476, 493-496).  phiNew =
 * SOMAR_F_PHI, phiOld = SOMAR_F_HEAT_OLD, source = SOMAR_F_HEAT_SRC of every level in the range.  l_base must be 0: for
 * l_base > 0 the reference reads *m_srct[l_base - 1], which createData never allocates (:388-403) -- undefined there, refused
 * here.  stats = the LAST solve's.
 * (No driver of the reference calls MappedAMRTGA -- AMRNavierStokes steps level by level with the integrators above -- it
 * is provided because the class ships with the operator.) */
int somar_amr_set_alpha_beta(somar_amr_t* a, double alpha, double beta);
int somar_amr_tga_step(somar_amr_t* a, int l_max, int l_base, double dt, somar_stats_t* stats);
int somar_amr_heat_step(somar_amr_t* a, int level, int scheme, double dt, int zero_phi, double old_time, double crse_old_time,
                        double crse_new_time, somar_stats_t* stats);
int somar_heat_flux_download(somar_solver_t* s, int dir, int patch, double* host); /* faces(valid, dir), Fortran order */

/* AMRLepticSolver (calculus/LepticSolver/AMRLepticSolver.cpp:68-672; AMRPressureSolver.cpp:383-403, 542-550 when
 * s_useAMRLepticSolver is set): the composite iteration of somar_amr_solve with LevelLepticSolver::solve in place of
 * relax and of the base level's multigrid cycle.  somar_amr_enable_leptic (after somar_amr_finalize) defines one leptic level
 * solver per level on the level's own operator, homogeneous, no coarse phi (init, :185-195); levels must consist of
 * vertically complete columns (refinement ratios (r, r, 1)).  base_from_restricted = 0 is the reference as written: the
 * base level solves a_uberCorrection from a_uberResidual and contributes nothing to the finer levels (:444-449), so a
 * multi-level solve drifts after its first cycles; 1 (NOT the reference) feeds the base level the restricted residual, as
 * MappedAMRMultiGrid::AMRVCycle does.  Stats: exit_status bitfield and history as somar_amr_solve; status 2 ("blew up") is
 * never raised, the reference only prints there (:384-388). */
int somar_amr_enable_leptic(somar_amr_t* a, const somar_leptic_params_t* lp, int base_from_restricted);
int somar_amr_solve_leptic(somar_amr_t* a, int l_max, int l_base, int zero_phi, int force_homogeneous, somar_stats_t* stats);
int somar_amr_leptic_stats(somar_amr_t* a, int level, somar_leptic_stats_t* stats); /* of the level's LAST leptic solve */

/* Metric producers.  The coordinate maps themselves stay with the caller (LevelGeometry /
 * GeoSourceInterface subclasses evaluate dx/dXi); what the device takes over is
 *  - GeoSourceInterface::fill_Jgup's generic algebra (geometry/GeoSourceInterface.cpp:200-450: fill_dXidx by cofactors,
 *    fill_gup, times det J, times scale) in 3-D: dxdxi9 = dx^rho/dXi^sigma at [(3 rho + sigma) n + i], detJ, out
 *    jgup3[nu n + i] = scale J g^{mu nu} -- the layout somar_solver_set_metric_full takes for face direction mu;
 *  - a Cartesian map (CartesianMap::fill_Jgup / fill_Jinv, geometry/maps/CartesianMap.cpp:230-280): c4 = {J g^xx, J g^yy,
 *    J g^zz, J^{-1}} written into every local patch on the device, nothing crosses PCIe (before somar_solver_finalize). */
int somar_metric_jgup_from_dxdxi(long long n, int mu, const double* dxdxi9, const double* detJ, double scale, double* jgup3);
int somar_solver_set_metric_uniform(somar_solver_t* s, const double* c4);
/*  - the coordinate maps whose evaluation is closed-form or needs only a 2-D input, evaluated on the device into EVERY local
 *    patch of the level (faces J g^{ab}, cells J^{-1}; before somar_solver_finalize; CH_SPACEDIM = 3):
 *      SOMAR_MAP_CYLINDRICAL  CylindricalMap::fill_dxdXi / fill_J (geometry/maps/CylindricalMap.cpp:125-190,
 *          CylindricalMapF.ChF), diagonal: x = xi cos(eta), y = xi sin(eta), z = zeta; L, depth unused;
 *      SOMAR_MAP_BATHYMETRIC  BathymetricBaseMap::fill_dxdXi / fill_J (geometry/maps/BathymetricBaseMap.cpp:133-313,
 *          BathymetricBaseMapF.ChF) with CONVERTFAB's centring averages (ConvertFABF.ChF:32-150), non-diagonal:
 *          z = d + (1 - d / H) zeta; L = domain lengths (H = L[2]); depth = the NODAL depth d the subclass's
 *          fill_bathymetry returns (DEMMap, LedgeMap, BeamGeneratorMap ...), host array over nodes
 *          [depth_lo, depth_lo + depth_n) of this level's index space, i fastest, covering nodes lo-1 .. hi+2 of every
 *          local box in both horizontal directions.  AVG3IX's misprinted eighth term (AddlFortranMacros.H:88) is reproduced.
 *      SOMAR_MAP_TWISTED      TwistedMap with m_twistType 0 (geometry/maps/TwistedMap.cpp:160-260, TwistedMapF.ChF:
 *          TWISTED0_FILL_DXDXI, TWISTED0_FILL_J), non-diagonal: x^mu = xi^mu + pert_mu sin(2 pi xi^nu) sin(2 pi xi^sigma);
 *          L = the three amplitudes m_pert, depth unused.
 *      SOMAR_MAP_TWISTED1     TwistedMap with m_twistType 1: analytic coordinate functions (TWISTED1_FILL_PHYSCOOR,
 *          TwistedMapF.ChF:356-430; domain lengths m_L = dx * cells of the level's domain), dx/dXi and J from
 *          GeoSourceInterface's defaults: staggered differences of the coordinates (GeoSourceInterface.cpp:65-113,
 *          SIMPLECCDERIV / SIMPLEFCDERIV) and DEFAULT_FILL_J_3D with Chombo's CellToEdge average on faces (:122-202).
 *          L = the three amplitudes m_pert, depth unused.
 *    All run GeoSourceInterface::fill_Jgup / fill_Jinv's generic algebra per point at the point's own centring, as
 *    LevelGeometry does.  Maps not listed (NewBeamGenerator's spline, the DEM interpolation itself) stay with the caller
 *    through somar_solver_set_metric_full or somar_metric_jgup_from_dxdxi. */
#define SOMAR_MAP_CYLINDRICAL 1
#define SOMAR_MAP_BATHYMETRIC 2
#define SOMAR_MAP_TWISTED 3
#define SOMAR_MAP_TWISTED1 4
int somar_solver_set_metric_map(somar_solver_t* s, int kind, const double* L, const double* depth, const int* depth_lo,
                                const int* depth_n);
/* The nodal "depth" of the reference's analytic bathymetric maps, for SOMAR_MAP_BATHYMETRIC (host arithmetic, no GPU; x, y: the
 * Cartesian coordinates of the nodes, for these maps x = dXi0 * i, y = dXi1 * j):
 *   LedgeMap::fill_bathymetry            geometry/maps/LedgeMap.cpp:38-60, 98-164   order 1 / 3: h_l left of x_l, h_r right of x_r,
 *        a linear / cubic transition between (y = NULL); the 3-D build's Gaussian bump exp(-((x - x_l) / h_l)^2 - ((y - x_r) / h_r)^2)
 *        when y is given (order ignored, as in the reference's CH_SPACEDIM == 3 branch)
 *   BeamGeneratorMap::fill_bathymetry    geometry/maps/BeamGeneratorMap.cpp:73-93, BeamGeneratorMapF.ChF:51-166   the smoothed
 *        triangular ridge of critical slope `angle` centred at x = 0 (Masoud's lab-scale proportions, the compiled-in set), Lx = the
 *        domain length in x */
int somar_bathymetry_ledge(double* out, long long n, const double* x, const double* y, int order, double hl, double hr, double xl,
                           double xr);
int somar_bathymetry_beam_generator(double* out, long long n, const double* x, double Lx, double angle);
/* DEMMap's interpolators (geometry/maps/DEMMap.cpp:197-462 turns a digital elevation model /X, /Y, /Depth into the nodal depth of
 * every level with them), host arithmetic:
 *   CubicSpline::solve + interp   calculus/interpolation/CubicSpline.cpp:57-143, CubicSplineF.ChF:46-112   the NATURAL cubic spline
 *        through (xd[q], fd[q]), q < nd (xd ascending), evaluated at x[0 .. n)   (Create_Level_DEM_2D)
 *   BilinearInterp2D              calculus/interpolation/BilinearInterp.cpp:37-119, BilinearInterpF.ChF   f on the tensor grid
 *        xd[0 .. nx) x yd[0 .. ny) (fd[i + nx * j]) evaluated at the points (x[q], y[q])   (Create_Level_DEM_3D, interpOrder 0)
 *   HermiteInterp2D               calculus/interpolation/HermiteInterp.cpp:28-146, HermiteInterpF.ChF:38-215   the same grid with
 *        interpOrder > 0: df/dx, df/dy by the centred (one-sided at the ends) differences of Create_Level_DEM_3D
 *        (DEMMap.cpp:222-289), then the "simplified Hermite interpolant" as written (the nodal derivatives enter the cubic basis
 *        functions unscaled by the cell width)
 * The HDF5 reader stays with the caller. */
int somar_dem_cubic_spline(double* out, long long n, const double* x, int nd, const double* xd, const double* fd);
int somar_dem_bilinear(double* out, long long n, const double* x, const double* y, int nx, int ny, const double* xd,
                       const double* yd, const double* fd);
int somar_dem_hermite(double* out, long long n, const double* x, const double* y, int nx, int ny, const double* xd,
                      const double* yd, const double* fd);

/* Diagnostics, no reference counterpart: what this device streams for a given mix of streams, in GB/s of algorithmic bytes
 * -- kind 0 copy (16 B/cell), 1 read (8 B/cell), 2 six reads + one write (56 B/cell: the fused GSRB sweep's mix without
 * stencil or halo).  bench.py reports kind 2 as the ceiling its roofline fraction is to be read against. */
int somar_diag_stream_probe(int kind, long long cells, int reps, double* gbs);

/* one-process-per-GPU transport (RCCL over xGMI).  The unique id is created on rank 0 and
 * distributed by the launcher (torch.distributed store / MPI / file). */
#define SOMAR_COMM_ID_BYTES 128
int somar_comm_unique_id(unsigned char* id128);
int somar_comm_create(void** comm, const unsigned char* id128, int rank, int nranks, int device);
/* host-staged transport over a POSIX shared-memory segment `name` (single node; every rank passes the same name,
 * rank 0 creates it): a way to run the sharded path where RCCL cannot be used -- several ranks on ONE GPU of a
 * development box -- with the same message plans.  Synchronous; not for production. */
int somar_comm_create_shm(void** comm, const char* name, int rank, int nranks, long long outbox_bytes);
/* Transport self-test: all-reduce (sum, max) of rank-dependent values and a ring neighbour exchange
 * (rank -> rank+1; a self send/recv on one rank), checked on the host.  Collective over the communicator. */
int somar_comm_selftest(void* comm);
int somar_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* SOMAR_AMD_H */
