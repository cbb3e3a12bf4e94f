// Device-resident linear solvers for  mass * x - diffusion * laplace(x) = rhs  that can stand in for the Jacobi
// sweeps of the split-Bregman loop (reference: utils/linear_solvers/mg.py, cg.py; plug-in contract solver.py:12-72).
// All pointers are device planes in the pitched (n0, n1*n2) layout of `Geom`.  gfx950 only.
#pragma once

#include "dxh_kernels.h"

namespace dxh {

struct DeviceSolver {
    virtual ~DeviceSolver() {}
    // x0, rhs, out: distinct device planes of the level-0 geometry; `iters` = cycles / iterations performed
    virtual int solve(const void* x0, const void* rhs, void* out, int* iters) = 0;
    // scalar or device-plane coefficients (a plane pointer wins over the scalar); copied, not retained
    virtual int set_params(double mass, const void* mass_plane, double diffusion, const void* diffusion_plane) = 0;
    // current coefficient planes (heterogeneous MG: as mutated by the V-cycles)
    virtual int get_params(void* mass_plane, void* diffusion_plane) = 0;
    int64_t launches = 0;
};

// mg.py:13-270.  `heterogeneous` mirrors the reference's constructor-time flag (mg.py:72-75).
DeviceSolver* make_mg_solver(int dtype, const Geom& g0, int depth, int smoother_iterations, int maxiter, double tol,
                             bool heterogeneous, hipStream_t stream, int* rc);
// cg.py:16-59; h <= 0 means "no grid spacing" (h=None)
DeviceSolver* make_cg_solver(int dtype, const Geom& g0, int maxiter, double tol, double h, hipStream_t stream, int* rc);

}  // namespace dxh
