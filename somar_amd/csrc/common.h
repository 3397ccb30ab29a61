// somar_amd/csrc/common.h -- shared host/device declarations for the MI355X (gfx950)
// implementation of SOMAR's pressure-projection multigrid.  HIP only; no CUDA paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace somar {

// Error convention of the C ABI (include/somar_amd.h): 0 ok, <0 failure.  Inside the
// library failures are C++ exceptions, translated at the ABI boundary.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define SOMAR_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            throw ::somar::Error(-2, std::string("HIP error ") + hipGetErrorString(e_) + " at " \
                                         + __FILE__ + ":" + std::to_string(__LINE__));          \
    } while (0)

#define SOMAR_CHECK(cond, msg)                                                      \
    do {                                                                            \
        if (!(cond))                                                                \
            throw ::somar::Error(-1, std::string(msg) + " (" #cond ") at " + __FILE__ + \
                                         ":" + std::to_string(__LINE__));         \
    } while (0)

// BCType codes, reference calculus/BCInterface/BCDescriptor.H:34-39
enum { BC_UNDEFINED = -2, BC_NONE = -1, BC_NEUM = 0, BC_DIRI = 1, BC_PERIODIC = 2, BC_CF = 3 };

// Ghost frame allocated around every patch, in cells.  All fields of a level (cell- and
// face-centred alike) share one shape so a single offset addresses every array; face
// index i is the LOW face of cell i (reference convention, MappedAMRPoissonOpF.ChF:412-415).
constexpr int FRAME = 2;

// One box of a level as stored in HBM.  Element (i,j,k) (local, 0-based from the valid
// low corner; ghosts are negative) of any field lives at  base[off + i + pj*j + pk*k].
struct PatchDesc {
    int lo[3];  // global index of the valid low corner
    int n[3];   // valid cells per direction
    int pj;     // j pitch (elements), even => 16-byte aligned rows
    int cf;     // bit (2*dir + side): that whole face of the box is a coarse-fine boundary (levels on a coarser one)
    long long pk;   // k pitch
    long long off;  // offset of local cell (0,0,0)
};

// Work item of a sweep kernel: a (128 x 4 x nk) brick of one patch.
struct Tile {
    int patch;
    int i0, j0, k0;  // local start
    int nk;
    int pad_[3];
};
constexpr int TILE_I = 128;  // cells in i per block (64 lanes x double2)
constexpr int TILE_J_MAX = 8;   // rows per block (blockDim.y), chosen per level
constexpr int TILE_K_MAX = 32;  // planes marched by one block, chosen per level

// One box-to-box ghost copy ("motion item" of a Chombo Copier).
struct CopyItem {
    int src_patch, dst_patch;
    int src_lo[3];  // local start in the source patch
    int dst_lo[3];  // local start in the destination patch
    int n[3];
    int pad_;
};

// One coarse-fine ghost cell of a level that sits on a coarser one (homogeneous CF interpolation).
struct CFCell {
    long long off;  // the ghost cell
    int stride;     // signed element stride pointing OUT of the box: off - stride is the first valid cell
    int dir;        // direction | 4 if the box is one cell wide in that direction
};

// One step of a ghost "program" of the non-diagonal (19-point) path (full19.hip): a region of one patch.
enum { GHOST_COPY = 0, GHOST_EXTRAP = 1, GHOST_NEUM = 2, GHOST_DIRI = 3 };
struct GhostOp {
    int patch, type;
    int lo[3];       // local start of the region
    int n[3];
    int dir, sgn;    // EXTRAP / NEUM: direction and side sign (+1 high, -1 low)
    int order;       // EXTRAP: 0, 1 or 2
    int dstf, srcf;  // 0 = phi, 1 = psi (the extrapolated copy)
    int pad_;
    double val;      // DIRI: the boundary value of that side
};

// Per-level constants handed to the stencil kernels by value.
struct StencilParams {
    int dom_lo[3], dom_hi[3];  // domain box at this depth
    int neum[3][2];            // 1 = homogeneous-Neumann physical face (no ghost, no flux)
    int active[3];             // activeDirs
    int periodic[3];
    double dx[3];
    double alpha, beta;
    double cf_c1[3], cf_c2[3];  // homogeneous CF interpolation: ghost = c1 * first valid + c2 * second valid
    int bc_homog;               // ghost programs: Dirichlet sides take the value 0 instead of their own
    int pad2_;
    int diri[3][2];             // 1 = Dirichlet physical face (ghost = 2 value - first cell)
    // uniform metric (a Cartesian map: CartesianMap.cpp:261-280 fills constants): J g^{aa} == uc[a] on every face and
    // J^{-1} == uc[3] in every cell of this depth, found by PressureSolver::detect_uniform_metric.  The k-marching kernels
    // then take the four values from here instead of streaming four arrays of them (56 -> 24 B/cell on a sweep).
    int uniform = 0;
    // non-diagonal metric: J g^{xy} on x-faces and J g^{yx} on y-faces are zero everywhere on this depth (any map with x = xi,
    // y = eta: BathymetricBaseMap and its DEM / Ledge / BeamGenerator subclasses); the marching 19-point kernels then stream
    // seven coefficient planes instead of nine (PressureSolver::detect_zero_planes)
    int zero_xy = 0;
    double uc[4] = {0.0, 0.0, 0.0, 0.0};
};

}  // namespace somar
