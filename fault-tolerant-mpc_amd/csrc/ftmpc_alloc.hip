// ftmpc_alloc.hip -- batched thruster allocation, the reference's second stage
// (ControlAllocator.get_physical_input, ft_mpc/controllers/tools/control_allocator.py:27-40,65-94):
//     min |u|^2   s.t.  D u = tau,  0 <= u <= ub          (NT <= 16 thrusters, 6 wrench components)
// The thruster-space MPC path does not need it (its QP has the allocation inside); it is the
// standalone operator for callers that, like the reference, hold a generalized force.
//
// One LANE per instance, float64.  Dual form: u(lambda) = clip(D' lambda, 0, ub), and lambda is
// the minimiser of the convex piecewise-quadratic dual
//     q(lambda) = sum_i [ v_i u_i - u_i^2 / 2 ] - lambda' tau,   v = D' lambda,   grad q = D u(lambda) - tau,
// found by a semismooth Newton method (generalised Hessian D_A D_A', A = thrusters strictly inside their
// bounds) with Armijo backtracking on q.  The dual is piecewise quadratic, so the iteration terminates
// in a handful of steps; an infeasible request (tau outside the attainable set) leaves a residual
// and is reported per instance (the reference prints and exit()s, control_allocator.py:88-93).
#include <hip/hip_runtime.h>

#include "ftmpc_common.h"

namespace ftmpc {

struct AllocParams {
    int64_t B;
    const double* tau;   // [B*6]
    const double* ub;    // [B*NT]  0 for a broken thruster
    double* out_u;       // [B*NT]
    int32_t* status;     // [B] or nullptr: 0 solved, 1 iteration cap, 2 infeasible (residual left)
    int32_t* iters;      // [B] or nullptr
    int32_t max_iters;
    double tol;          // |D u - tau|_inf <= tol (1 + |tau|_inf)
    // list mode (the instances kernel 11 handed over, allocated after kernel 13 has re-solved them): lane i serves list[i], i < *count
    const int32_t* list = nullptr;
    const int32_t* count = nullptr;
};

namespace {

// Dm: the allocation matrix, 6 x MAX_NT, in LDS (ninety-six kernel-argument doubles do not fit the scalar registers beside
// the rest: the unrolled loops spilled 381 of them)
__device__ inline double alloc_dual(const double* Dm, int NT, const double lam[6], const double tau[6], const double* ubv,
                                    double u[MAX_NT], double F[6]) {
    double q = 0.0;
#pragma unroll
    for (int g = 0; g < 6; ++g) F[g] = -tau[g];
#pragma unroll
    for (int i = 0; i < MAX_NT; ++i) {
        u[i] = 0.0;
        if (i < NT) {
            double v = 0.0;
#pragma unroll
            for (int g = 0; g < 6; ++g) v += Dm[g * MAX_NT + i] * lam[g];
            const double ui = fmin(fmax(v, 0.0), ubv[i]);
            u[i] = ui;
            q += v * ui - 0.5 * ui * ui;
#pragma unroll
            for (int g = 0; g < 6; ++g) F[g] += Dm[g * MAX_NT + i] * ui;
        }
    }
#pragma unroll
    for (int g = 0; g < 6; ++g) q -= lam[g] * tau[g];
    return q;
}

// solves (J + delta I) x = b for the symmetric positive semi-definite 6x6 J (Cholesky, in place)
__device__ inline void solve6(double J[6][6], const double b[6], double x[6]) {
    double tr = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) tr += J[i][i];
    const double delta = 1e-12 * tr + 1e-300;
    double L[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double d = J[j][j] + delta;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < j) d -= L[j][k] * L[j][k];
        d = fmax(d, delta);
        const double inv = 1.0 / sqrt(d);
        L[j][j] = d * inv;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i > j) {
                double s = J[i][j];
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if (k < j) s -= L[i][k] * L[j][k];
                L[i][j] = s * inv;
            }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < i) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k > i) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}

}  // namespace

__global__ void __launch_bounds__(64) ftmpc_allocate_kernel(const DeviceConsts C, const AllocParams P) {
    __shared__ double Dm[6 * MAX_NT];
    for (int i = threadIdx.x; i < 6 * MAX_NT; i += 64) Dm[i] = C.D[i];
    __syncthreads();
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    if (P.list) {
        if (b >= *P.count) return;
        b = P.list[b];
    }
    const int NT = C.NT;
    double tau[6], ubv[MAX_NT];
    double tmax = 0.0;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        tau[g] = P.tau[b * 6 + g];
        tmax = fmax(tmax, fabs(tau[g]));
    }
#pragma unroll
    for (int i = 0; i < MAX_NT; ++i) ubv[i] = (i < NT) ? P.ub[b * NT + i] : 0.0;
    // start: unconstrained least-norm multiplier (D D') lambda = tau over the healthy thrusters
    double lam[6], u[MAX_NT], F[6];
    {
        double J[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < MAX_NT; ++i)
                    if (i < NT && ubv[i] > 0.0) s += Dm[r * MAX_NT + i] * Dm[c * MAX_NT + i];
                J[r][c] = s;
            }
        solve6(J, tau, lam);
    }
    double q = alloc_dual(Dm, NT, lam, tau, ubv, u, F);
    const double tol = P.tol * (1.0 + tmax);
    // A tau on the boundary of the attainable set (an MPC solution with active hull rows: several thrusters exactly at a
    // bound) leaves the dual flat and the Newton iteration stalls a few 1e-7 short.  Polish: thrusters within 1e-6 f_max
    // of a bound are put ON it, the rest take the least-norm share of what is left; accepted if the residual passes.
    auto polish = [&]() -> bool {
        double ubmax = 0.0;
#pragma unroll
        for (int i = 0; i < MAX_NT; ++i) ubmax = fmax(ubmax, ubv[i]);
        const double band = 1e-6 * ubmax;
        double up[MAX_NT], r[6], J[6][6], lf[6];
        bool fr[MAX_NT];
#pragma unroll
        for (int g = 0; g < 6; ++g) r[g] = tau[g];
#pragma unroll
        for (int r6 = 0; r6 < 6; ++r6)
#pragma unroll
            for (int c = 0; c < 6; ++c) J[r6][c] = 0.0;
        bool any_free = false;
#pragma unroll
        for (int i = 0; i < MAX_NT; ++i) {
            const bool healthy = i < NT && ubv[i] > 0.0;
            const bool lo = healthy && u[i] <= band, hi = healthy && u[i] >= ubv[i] - band;
            fr[i] = healthy && !lo && !hi;
            up[i] = hi ? ubv[i] : 0.0;
            any_free = any_free || fr[i];
            if (hi) {
#pragma unroll
                for (int g = 0; g < 6; ++g) r[g] -= Dm[g * MAX_NT + i] * ubv[i];
            }
            if (fr[i]) {
#pragma unroll
                for (int r6 = 0; r6 < 6; ++r6)
#pragma unroll
                    for (int c = 0; c < 6; ++c) J[r6][c] += Dm[r6 * MAX_NT + i] * Dm[c * MAX_NT + i];
            }
        }
        if (any_free) {
            solve6(J, r, lf);
            // the fixed set is a GUESS read off an iterate that has not converged: it is the minimum-norm allocation only if the
            // multiplier of the free set agrees with it -- D_i' lf >= ub_i for a thruster put on its upper bound, <= 0 on the
            // lower (the KKT signs of min |u|^2, D u = tau, 0 <= u <= ub).  Otherwise this is a feasible point that is not the
            // answer: refused, the Newton iteration goes on.
            bool signs = true;
#pragma unroll
            for (int i = 0; i < MAX_NT; ++i) {
                const bool healthy = i < NT && ubv[i] > 0.0;
                double v = 0.0;
#pragma unroll
                for (int g = 0; g < 6; ++g) v += Dm[g * MAX_NT + i] * lf[g];
                if (fr[i]) up[i] = fmin(fmax(v, 0.0), ubv[i]);
                else if (healthy) signs = signs && ((up[i] > 0.0) ? v >= ubv[i] - band : v <= band);
            }
            if (!signs) return false;
        }
        double res = 0.0;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            double t = -tau[g];
#pragma unroll
            for (int i = 0; i < MAX_NT; ++i)
                if (i < NT) t += Dm[g * MAX_NT + i] * up[i];
            res = fmax(res, fabs(t));
        }
        if (res > tol) return false;
#pragma unroll
        for (int i = 0; i < MAX_NT; ++i) u[i] = up[i];
        return true;
    };
    // (the polish is also tried EARLY, once, when the residual has stopped halving over three iterations: left to the iteration
    // cap, the boundary wrenches the wrench-space MPC hands over cost fifty Newton steps with a line search each before it runs,
    // and one such lane holds up its whole wave)
    double fh[3] = {1e300, 1e300, 1e300};
    bool tried = false;
    int status = 1, it = 0;
    for (; it < P.max_iters; ++it) {
        double fmaxabs = 0.0;
#pragma unroll
        for (int g = 0; g < 6; ++g) fmaxabs = fmax(fmaxabs, fabs(F[g]));
        if (fmaxabs <= tol) {
            status = 0;
            break;
        }
        if (!tried && fmaxabs > 0.5 * fh[0] && fmaxabs <= 1e-3 * (1.0 + tmax)) {
            tried = true;
            if (polish()) {
                status = 0;
                break;
            }
        }
        fh[0] = fh[1];
        fh[1] = fh[2];
        fh[2] = fmaxabs;
        // an unattainable tau makes the dual unbounded below: the multiplier runs away
        double lmax = 0.0;
#pragma unroll
        for (int g = 0; g < 6; ++g) lmax = fmax(lmax, fabs(lam[g]));
        if (lmax > 1e9 * (1.0 + tmax)) {
            status = 2;
            break;
        }
        // generalised Hessian over the thrusters strictly inside their bounds
        double J[6][6];
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) J[r][c] = 0.0;
#pragma unroll
        for (int i = 0; i < MAX_NT; ++i)
            if (i < NT && u[i] > 0.0 && u[i] < ubv[i]) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c) J[r][c] += Dm[r * MAX_NT + i] * Dm[c * MAX_NT + i];
            }
        double nF[6], dl[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) nF[g] = -F[g];
        solve6(J, nF, dl);
        double slope = 0.0;
#pragma unroll
        for (int g = 0; g < 6; ++g) slope += F[g] * dl[g];
        if (!(slope < 0.0)) {   // no descent along the Newton direction: steepest descent
#pragma unroll
            for (int g = 0; g < 6; ++g) dl[g] = -F[g];
            slope = 0.0;
#pragma unroll
            for (int g = 0; g < 6; ++g) slope -= F[g] * F[g];
        }
        double t = 1.0;
        double ln[6], Fn[6], qn = q;
        bool moved = false;
        for (int ls = 0; ls < 40; ++ls) {
#pragma unroll
            for (int g = 0; g < 6; ++g) ln[g] = lam[g] + t * dl[g];
            qn = alloc_dual(Dm, NT, ln, tau, ubv, u, Fn);      // (u follows the trial point: restored below if nothing is accepted)
            // (rounding slack: for a wrench on the boundary of the attainable set -- what the polished MPC solution hands over --
            // the decrease of a good step is below 1e-16 |q|, and without the slack one such wrench in ~10^4 stalled here)
            if (qn <= q + 1e-4 * t * slope + 1e-13 * (1.0 + fabs(q))) {
                moved = true;
                break;
            }
            t *= 0.5;
        }
        if (!moved) {        // stationary for the line search: the residual that is left is infeasibility
            q = alloc_dual(Dm, NT, lam, tau, ubv, u, F);
            break;
        }
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            lam[g] = ln[g];
            F[g] = Fn[g];
        }
        q = qn;
    }
    if (status == 1) {
        double fmaxabs = 0.0;
#pragma unroll
        for (int g = 0; g < 6; ++g) fmaxabs = fmax(fmaxabs, fabs(F[g]));
        status = (fmaxabs <= tol) ? 0 : (it >= P.max_iters ? 1 : 2);
    }
    if (status != 0 && polish()) status = 0;
#pragma unroll
    for (int i = 0; i < MAX_NT; ++i)
        if (i < NT) P.out_u[b * NT + i] = u[i];
    if (P.status) P.status[b] = status;
    if (P.iters) P.iters[b] = it;
}

// wrench the HEALTHY thrusters have to produce: tau_0 - D stuck (the reference's u_res = u + u_nom + u_comp,
// spiraling_mpc.py:301-302, i.e. the total wrench minus the uncontrollable part D f_fault)
__global__ void __launch_bounds__(256) ftmpc_healthy_wrench_kernel(const DeviceConsts C, int64_t B, const double* tau0, const double* stuck,
                                                                  double* out, const int32_t* list = nullptr, const int32_t* count = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 6) return;
    int64_t b = i / 6;
    const int g = (int)(i - 6 * b);
    if (list) {      // (list mode: see AllocParams)
        if (b >= *count) return;
        b = list[b];
    }
    double t = tau0[b * 6 + g];
    for (int k = 0; k < C.NT; ++k) t -= C.D[g * MAX_NT + k] * stuck[b * C.NT + k];
    out[b * 6 + g] = t;
}

}  // namespace ftmpc
