// ftmpc_sim.hip -- the caller side of the MPC step on the device:
// plant integration + measurement noise + quaternion renormalisation
//   (reference: SimulationEnvironment.step, ft_mpc/simulation/sim_env.py:77-99, with
//    SystemModel.dx_dt / rk4_integrator, ft_mpc/models/sys_model.py:138-226)
// and the warm-start shift (ft_mpc/controllers/spiraling_mpc.py:324-334), so that a Monte-Carlo
// fault campaign runs T closed-loop steps without host round trips.
#include <hip/hip_runtime.h>

#include "ftmpc_common.h"

namespace ftmpc {

namespace {
// counter-based uniform in [0,1): splitmix64 of (seed, index); oracle/closed_loop.py mirrors it
__device__ __forceinline__ double u01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void plant_f(const DeviceConsts& C, const double* x, const double* gen, double* dx) {
    const double *v = x + 3, *q = x + 6, *w = x + 10;
    dx[0] = v[0]; dx[1] = v[1]; dx[2] = v[2];
    // v' = Rot(q)^T F / m   (sys_model.py:215)
    const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const double R00 = qx * qx - qy * qy - qz * qz + qw * qw, R01 = 2 * (qx * qy + qz * qw), R02 = 2 * (qx * qz - qy * qw);
    const double R10 = 2 * (qx * qy - qz * qw), R11 = -qx * qx + qy * qy - qz * qz + qw * qw, R12 = 2 * (qy * qz + qx * qw);
    const double R20 = 2 * (qx * qz + qy * qw), R21 = 2 * (qy * qz - qx * qw), R22 = -qx * qx - qy * qy + qz * qz + qw * qw;
    const double f0 = gen[0] * C.inv_mass, f1 = gen[1] * C.inv_mass, f2 = gen[2] * C.inv_mass;
    dx[3] = R00 * f0 + R10 * f1 + R20 * f2;
    dx[4] = R01 * f0 + R11 * f1 + R21 * f2;
    dx[5] = R02 * f0 + R12 * f1 + R22 * f2;
    // q' = 1/2 Omega(w) q   (sys_model.py:18-29,218)
    dx[6] = 0.5 * (w[2] * qy - w[1] * qz + w[0] * qw);
    dx[7] = 0.5 * (-w[2] * qx + w[0] * qz + w[1] * qw);
    dx[8] = 0.5 * (w[1] * qx - w[0] * qy + w[2] * qw);
    dx[9] = 0.5 * (-w[0] * qx - w[1] * qy - w[2] * qz);
    // w' = J^-1 (tau - w x J w)   (sys_model.py:221-224)
    double Jw[3], t[3];
    for (int i = 0; i < 3; ++i) Jw[i] = C.J[3 * i] * w[0] + C.J[3 * i + 1] * w[1] + C.J[3 * i + 2] * w[2];
    t[0] = gen[3] - (w[1] * Jw[2] - w[2] * Jw[1]);
    t[1] = gen[4] - (w[2] * Jw[0] - w[0] * Jw[2]);
    t[2] = gen[5] - (w[0] * Jw[1] - w[1] * Jw[0]);
    for (int i = 0; i < 3; ++i) dx[10 + i] = C.Jinv[3 * i] * t[0] + C.Jinv[3 * i + 1] * t[1] + C.Jinv[3 * i + 2] * t[2];
}
}  // namespace

struct SimParams {
    int64_t B;
    double* x;            // [B*13] in/out
    const double* u0;     // [B*NT] command of this step
    const double* ub;
    const double* stuck;
    double noise[4];      // amplitudes: position, velocity, orientation, angular velocity (U(0, a), sim_env.py:25-30)
    unsigned long long seed;
    int64_t step;
    double* u_hist;       // nullptr or [T*B*NT]
    const int32_t* status;
    int32_t* bad_count;   // nullptr or [T]
};

__global__ void __launch_bounds__(64) ftmpc_plant_step_kernel(const DeviceConsts C, const SimParams S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    const int NT = C.NT;
    double gen[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < NT; ++i) {
        const double u = S.u0[b * NT + i];
        if (S.u_hist) S.u_hist[(S.step * S.B + b) * NT + i] = u;
        const double t = (S.ub[b * NT + i] > 0.0 ? u : 0.0) + S.stuck[b * NT + i];   // sys_model.py:198-208
        for (int g = 0; g < 6; ++g) gen[g] += C.D[g * MAX_NT + i] * t;
    }
    double x[13], k1[13], k2[13], k3[13], k4[13], s[13];
    for (int i = 0; i < 13; ++i) x[i] = S.x[b * 13 + i];
    const double dt = C.dt;
    plant_f(C, x, gen, k1);
    for (int i = 0; i < 13; ++i) s[i] = x[i] + 0.5 * dt * k1[i];
    plant_f(C, s, gen, k2);
    for (int i = 0; i < 13; ++i) s[i] = x[i] + 0.5 * dt * k2[i];
    plant_f(C, s, gen, k3);
    for (int i = 0; i < 13; ++i) s[i] = x[i] + dt * k3[i];
    plant_f(C, s, gen, k4);
    for (int i = 0; i < 13; ++i) x[i] += dt / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    // one-sided uniform measurement noise (sim_env.py:88-91), then quaternion renormalisation (:93)
    for (int i = 0; i < 13; ++i) {
        const double a = i < 3 ? S.noise[0] : (i < 6 ? S.noise[1] : (i < 10 ? S.noise[2] : S.noise[3]));
        if (a > 0.0) x[i] += a * u01(S.seed, (unsigned long long)((S.step * S.B + b) * 13 + i));
    }
    const double qn = 1.0 / sqrt(x[6] * x[6] + x[7] * x[7] + x[8] * x[8] + x[9] * x[9]);
    for (int i = 6; i < 10; ++i) x[i] *= qn;
    for (int i = 0; i < 13; ++i) S.x[b * 13 + i] = x[i];
    if (S.bad_count && S.status && S.status[b] != 0) atomicAdd(&S.bad_count[S.step], 1);
}

// warm[b][k] = U[b][k+1] (k < N-1), warm[b][N-1] = 0     (spiraling_mpc.py:327-329)
// last stage: zero (the thruster sequences, spiraling_mpc.py:327-329) or, repeat_last != 0, the previous last stage again
// (the wrench sequences of the two-stage structure)
__global__ void __launch_bounds__(256) ftmpc_shift_warm_kernel(int64_t B, int N, int NT, const double* U, double* warm, int repeat_last) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)N * NT;
    if (i >= B * per) return;
    const int64_t r = i % per;
    warm[i] = (r < per - NT) ? U[i + NT] : (repeat_last ? U[i] : 0.0);
}

__global__ void __launch_bounds__(256) ftmpc_count_nonzero_kernel(int64_t B, const int32_t* flags, int32_t* count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < B && flags[i] != 0;
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (int32_t)__popcll(m));
}


// ---------------------------------------------------------------------------------------------------------
// Line-search SQP towards the reference's nonlinear program, on the device (reference
// spiraling_mpc.py:87-238 solved by IPOPT :346).  ft_mpc_amd.BatchedMPC.solve_sqp is the host mirror of exactly this
// bookkeeping; here nothing crosses PCIe between the first upload and the last download.
//   per instance:  J (cost at U), Jt (cost at the trial point), alpha, flags {active, todo, improved}, counters
// ---------------------------------------------------------------------------------------------------------
struct SqpState {
    int64_t B;
    int32_t N, NT;
    const double* ub;       // [B*NT]
    double* U;              // [B*N*NT] current iterate
    const double* Uq;       // [B*N*NT] solution of the QP linearised about U
    double* Ut;             // [B*N*NT] trial point U + alpha (clip(Uq) - U)
    double* J;              // [B]
    const double* Jt;       // [B]
    double* alpha;          // [B] current trial step; after a success: the accepted step
    int32_t* active;        // [B]
    int32_t* todo;          // [B]
    int32_t* improved;      // [B]
    int32_t* nmajor;        // [B]
    int32_t* ipm;           // [B]
    int32_t* status;        // [B]
    const int32_t* qstatus; // [B] of the last QP
    const int32_t* qiters;  // [B]
    double tol;
    const double* Jall = nullptr;   // [B*ntrial] costs of all trial points of a line search (ftmpc_cost_kernel, ntrial > 0)
    int32_t ntrial = 0;
};

// U = clip(warm, 0, ub) (or 0); active = 1; counters = 0
__global__ void ftmpc_sqp_init_kernel(const SqpState S, const double* warm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nw = (int64_t)S.N * S.NT;
    if (i < S.B * nw) {
        const int64_t b = i / nw;
        const int t = (int)(i % S.NT);
        const double ub = S.ub[b * S.NT + t];
        S.U[i] = warm ? fmin(fmax(warm[i], 0.0), ub) : 0.0;
    }
    if (i < S.B) {
        S.active[i] = 1;
        S.todo[i] = 0;
        S.improved[i] = 0;
        S.nmajor[i] = 0;
        S.ipm[i] = 0;
        S.status[i] = 0;
        S.alpha[i] = 1.0;
    }
}
// after the QP of a major iteration: account for it, open the line search
__global__ void ftmpc_sqp_open_kernel(const SqpState S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    const int act = S.active[b];
    if (act) {
        S.ipm[b] += S.qiters[b];
        S.status[b] = S.qstatus[b];
    }
    S.todo[b] = act && S.qstatus[b] != 2;
    S.improved[b] = 0;
    S.alpha[b] = 1.0;
}
// trial point Ut = U + alpha (clip(Uq, 0, ub) - U)   (the fp32 kernels return ub rounded to float32: hence the clip)
__global__ void ftmpc_sqp_trial_kernel(const SqpState S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nw = (int64_t)S.N * S.NT;
    if (i >= S.B * nw) return;
    const int64_t b = i / nw;
    const int t = (int)(i % S.NT);
    const double ub = S.ub[b * S.NT + t];
    const double step = fmin(fmax(S.Uq[i], 0.0), ub) - S.U[i];
    S.Ut[i] = S.U[i] + S.alpha[b] * step;
}
// accept the trial point where the TRUE cost decreased enough, else halve the step
__global__ void ftmpc_sqp_decide_kernel(const SqpState S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B || !S.todo[b]) return;
    const double J = S.J[b], Jt = S.Jt[b];
    if (Jt < J - S.tol * (1.0 + fabs(J))) {
        S.J[b] = Jt;
        S.improved[b] = 1;
        S.todo[b] = 0;       // alpha keeps the accepted step
    } else {
        S.alpha[b] *= 0.5;
    }
}
// the whole line search at once: the costs of the trial points alpha = 1, 1/2, ... are all there (S.Jall); the first that decreases the
// TRUE cost enough is accepted -- what `backtracks` rounds of trial / cost / decide arrive at, in one launch instead of 3 x backtracks
__global__ void ftmpc_sqp_pick_kernel(const SqpState S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B || !S.todo[b]) return;
    const double J = S.J[b];
    double alpha = S.alpha[b];
    for (int j = 0; j < S.ntrial; ++j) {
        const double Jt = S.Jall[b * S.ntrial + j];
        if (Jt < J - S.tol * (1.0 + fabs(J))) {
            S.J[b] = Jt;
            S.improved[b] = 1;
            S.todo[b] = 0;
            break;
        }
        alpha *= 0.5;
    }
    S.alpha[b] = alpha;
}
// close the line search: U += alpha step where a trial point was accepted; an instance without progress stops
__global__ void ftmpc_sqp_close_kernel(const SqpState S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nw = (int64_t)S.N * S.NT;
    if (i < S.B * nw) {
        const int64_t b = i / nw;
        if (S.improved[b]) {
            const int t = (int)(i % S.NT);
            const double ub = S.ub[b * S.NT + t];
            const double step = fmin(fmax(S.Uq[i], 0.0), ub) - S.U[i];
            S.Ut[i] = S.U[i] + S.alpha[b] * step;      // (U is read by the other threads of this launch: the new iterate goes to Ut)
        } else {
            S.Ut[i] = S.U[i];
        }
    }
}
__global__ void ftmpc_sqp_count_kernel(const SqpState S) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    S.nmajor[b] += S.improved[b];
    S.active[b] = S.active[b] && S.improved[b];
}

}  // namespace ftmpc
