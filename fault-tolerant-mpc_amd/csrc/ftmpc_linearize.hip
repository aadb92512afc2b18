// ftmpc_linearize.hip -- kernel 1 of the MPC QP-step path: nonlinear rollout + RK4 Jacobians.
//
// One LANE per instance, float64 arithmetic.  A lane's record (152 words per stage) is contiguous per
// instance because kernel 2 reads it that way, so storing word by word from the lanes would touch 64
// different cache lines per store instruction (measured: 4.6 GB written + 2.4 GB read-modify-write
// for a 1.6 GB payload).  The words are therefore staged through LDS ([word][lane], XOR-swizzled so
// that both sides are bank-conflict free) and written out by the whole wave as contiguous runs.
// For each horizon stage it evaluates the orbit-centre dynamics
//   (reference: SpiralModel.dx_dt, ft_mpc/models/spiral_model.py:44-76)
// at the four RK4 stage points (SystemModel.rk4_integrator, ft_mpc/models/sys_model.py:138-162),
// and propagates forward-mode tangents for the 13 non-trivial input directions
// (omega0, q0, F, tau) -- the Jacobians the reference obtains from CasADi AD
// (ft_mpc/controllers/spiraling_mpc.py:217-230).  SURVEY.md Appendix A lists the formulas.
//
// Work per instance ~ N * (4 dynamics evaluations + 13 * 4 tangent steps) ~ 0.1 MFLOP: below
// 1 % of the step, so this kernel is latency/occupancy-trivial and not the roofline kernel.
#include <hip/hip_runtime.h>

#include "ftmpc_common.h"

namespace ftmpc {

namespace {

struct StageBlk {
    double Fww[9];   // d wdot / d w
    double Fqw[12];  // d qdot / d w   (4x3)
    double Fqq[16];  // d qdot / d q   (4x4)
    double Vw[9];    // d vdot / d w
    double Vq[12];   // d vdot / d q   (3x4)
    double RT[9];    // Rot(q)^T  (body -> world), ft_mpc/util/utils.py:15-19 transposed
};

__device__ inline void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline void mat3vec(const double M[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; ++i) o[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
}

// skew(a) * M  (3x3)
__device__ inline void skew_mul(const double a[3], const double M[9], double o[9]) {
    for (int j = 0; j < 3; ++j) {
        o[0 + j] = -a[2] * M[3 + j] + a[1] * M[6 + j];
        o[3 + j] = a[2] * M[0 + j] - a[0] * M[6 + j];
        o[6 + j] = -a[1] * M[0 + j] + a[0] * M[3 + j];
    }
}

// Rot(q)^T for q = [x,y,z,w]; no unit-norm assumption (utils.py:15-19)
__device__ inline void rotT(const double q[4], double RT[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    // R rows (world->body); RT[i][j] = R[j][i]
    const double R00 = x * x - y * y - z * z + w * w, R01 = 2 * (x * y + z * w), R02 = 2 * (x * z - y * w);
    const double R10 = 2 * (x * y - z * w), R11 = -x * x + y * y - z * z + w * w, R12 = 2 * (y * z + x * w);
    const double R20 = 2 * (x * z + y * w), R21 = 2 * (y * z - x * w), R22 = -x * x - y * y + z * z + w * w;
    RT[0] = R00; RT[1] = R10; RT[2] = R20;
    RT[3] = R01; RT[4] = R11; RT[5] = R21;
    RT[6] = R02; RT[7] = R12; RT[8] = R22;
}

// one evaluation of the centre dynamics at (w, q) with total wrench (F, tau), plus its blocks
__device__ inline void stage_eval(const DeviceConsts& C, const double w[3], const double q[4],
                                  const double F[3], const double tau[3], double dw[3], double dq[4],
                                  double dv[3], StageBlk& S) {
    double Jw[3], wxJw[3], t[3];
    mat3vec(C.J, w, Jw);
    cross3(w, Jw, wxJw);
    for (int i = 0; i < 3; ++i) t[i] = tau[i] - wxJw[i];
    mat3vec(C.Jinv, t, dw);  // spiral_model.py:63-65
    // qdot = 1/2 Omega(w) q  (sys_model.py:18-29)
    dq[0] = 0.5 * (w[2] * q[1] - w[1] * q[2] + w[0] * q[3]);
    dq[1] = 0.5 * (-w[2] * q[0] + w[0] * q[2] + w[1] * q[3]);
    dq[2] = 0.5 * (w[1] * q[0] - w[0] * q[1] + w[2] * q[3]);
    dq[3] = 0.5 * (-w[0] * q[0] - w[1] * q[1] - w[2] * q[2]);
    // a_b = F/m + wdot x r + w x (w x r)   (spiral_model.py:67-71)
    double wxr[3], wxwxr[3], dwxr[3], ab[3];
    cross3(w, C.r, wxr);
    cross3(w, wxr, wxwxr);
    cross3(dw, C.r, dwxr);
    for (int i = 0; i < 3; ++i) ab[i] = F[i] * C.inv_mass + dwxr[i] + wxwxr[i];
    rotT(q, S.RT);
    mat3vec(S.RT, ab, dv);
    // ---- blocks ----
    // Fww = -Jinv ( [w]x J - [Jw]x )
    double M1[9], M2[9];
    skew_mul(w, C.J, M1);
    M2[0] = 0; M2[1] = -Jw[2]; M2[2] = Jw[1];
    M2[3] = Jw[2]; M2[4] = 0; M2[5] = -Jw[0];
    M2[6] = -Jw[1]; M2[7] = Jw[0]; M2[8] = 0;
    for (int i = 0; i < 9; ++i) M1[i] -= M2[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            S.Fww[3 * i + j] = -(C.Jinv[3 * i] * M1[j] + C.Jinv[3 * i + 1] * M1[3 + j] + C.Jinv[3 * i + 2] * M1[6 + j]);
    // d a_b / d w = -[r]x Fww - [w x r]x - [w]x [r]x
    double rF[9], wr[9], dab[9];
    skew_mul(C.r, S.Fww, rF);
    const double SR[9] = {0, -C.r[2], C.r[1], C.r[2], 0, -C.r[0], -C.r[1], C.r[0], 0};
    skew_mul(w, SR, wr);
    const double SW[9] = {0, -wxr[2], wxr[1], wxr[2], 0, -wxr[0], -wxr[1], wxr[0], 0};
    for (int i = 0; i < 9; ++i) dab[i] = -rF[i] - SW[i] - wr[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            S.Vw[3 * i + j] = S.RT[3 * i] * dab[j] + S.RT[3 * i + 1] * dab[3 + j] + S.RT[3 * i + 2] * dab[6 + j];
    // d (Rot(q)^T a_b) / d q   (3x4), entries linear in q
    {
        const double x = q[0], y = q[1], z = q[2], ww = q[3];
        const double d0[12] = {x, -y, -z, ww, y, x, ww, z, z, -ww, x, -y};
        const double d1[12] = {y, x, -ww, -z, -x, y, -z, ww, ww, z, y, x};
        const double d2[12] = {z, ww, x, y, -ww, z, y, -x, -x, -y, z, ww};
        for (int i = 0; i < 12; ++i) S.Vq[i] = 2.0 * (ab[0] * d0[i] + ab[1] * d1[i] + ab[2] * d2[i]);
    }
    // Fqw = 1/2 Xi(q) (4x3), Fqq = 1/2 Omega(w) (4x4)
    {
        const double x = q[0], y = q[1], z = q[2], ww = q[3];
        const double xi[12] = {ww, -z, y, z, ww, -x, -y, x, ww, -x, -y, -z};
        for (int i = 0; i < 12; ++i) S.Fqw[i] = 0.5 * xi[i];
        const double om[16] = {0, w[2], -w[1], w[0], -w[2], 0, w[0], w[1], w[1], -w[0], 0, w[2], -w[0], -w[1], -w[2], 0};
        for (int i = 0; i < 16; ++i) S.Fqq[i] = 0.5 * om[i];
    }
}

}  // namespace

constexpr int STG_WORDS = (REC_BPF > REC_STRIDE - REC_BPF) ? REC_BPF : REC_STRIDE - REC_BPF;   // 79: 40 448 B of LDS per wave

// DIRS = false: the whole record of 64 instances per wave (full batches).  DIRS = true: blockIdx.y picks a contiguous
// share of the 13 tangent directions (gridDim.y = 13, 4 or 2 shares; y = 0 also writes the words that are not tangents,
// the work lists and e_N); every block repeats the cheap nonlinear rollout and stage blocks, so a small batch spreads
// over more waves and each wave does a fraction of the work -- same arithmetic per direction, same record bits
// (tests/test_gpu_parity.py).  The host takes it where the full-record kernel would leave most of the device idle.
template <typename OutT, int SHARE>   // 0: full records; 1: one direction per block; 2: gridDim.y shares of the 13 directions
__global__ void __launch_bounds__(64) ftmpc_linearize_kernel(const DeviceConsts C, const LinParams P) {
    static_assert(sizeof(OutT) == 8, "records are float64");
    constexpr bool DIRS = SHARE != 0;
    __shared__ double stg[DIRS ? 1 : STG_WORDS * 64];
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    // lanes past the end of the batch redo the last instance (they take part in the cooperative stores)
    const int64_t b = (b0 + lane < P.B) ? b0 + lane : P.B - 1;
    const int N = C.N, NT = C.NT;
    OutT* const recw = reinterpret_cast<OutT*>(P.rec);
    const int jlo = SHARE == 0 ? 0 : (SHARE == 1 ? (int)blockIdx.y : 13 * (int)blockIdx.y / (int)gridDim.y);
    const int jhi = SHARE == 0 ? 13 : (SHARE == 1 ? jlo + 1 : 13 * ((int)blockIdx.y + 1) / (int)gridDim.y);
    const bool writes_rest = !DIRS || blockIdx.y == 0;
    int pk = 0, pbase = 0;     // DIRS: stage and half (0 / REC_BPF) the next put belongs to
    // stage one record word of this lane (word index relative to the current half)
    auto put = [&](int wd, double v) {
        if constexpr (DIRS) {
            if (b0 + lane < P.B) recw[(b * N + pk) * (int64_t)REC_STRIDE + pbase + wd] = (OutT)v;
        } else {
            stg[wd * 64 + (lane ^ (wd & 63))] = v;
        }
    };
    // the wave writes words [base, base + W) of stage k for its 64 instances: contiguous W-word runs
    auto flush = [&](int k, int base, int W) {
        if constexpr (!DIRS) {
            __syncthreads();
            for (int idx = lane; idx < 64 * W; idx += 64) {
                const int inst = idx / W, wd = idx - inst * W;
                const double v = stg[wd * 64 + (inst ^ (wd & 63))];
                if (b0 + inst < P.B) recw[((b0 + inst) * N + k) * (int64_t)REC_STRIDE + base + wd] = (OutT)v;
            }
            __syncthreads();
        }
    };

    // ---- robot -> orbit-centre state (spiral_model.py:91-109) ----
    double x[13];
    for (int i = 0; i < 13; ++i) x[i] = P.x0[b * 13 + i];
    double pos[3], vel[3], w[3], q[4];
    for (int i = 0; i < 4; ++i) q[i] = x[6 + i];
    for (int i = 0; i < 3; ++i) w[i] = x[10 + i];
    {
        double RT[9], wxr[3], a[3], c[3];
        rotT(q, RT);
        cross3(w, C.r, wxr);
        mat3vec(RT, C.r, a);
        mat3vec(RT, wxr, c);
        for (int i = 0; i < 3; ++i) {
            pos[i] = x[i] + a[i];
            vel[i] = x[3 + i] + c[i];
        }
    }
    double ubv[MAX_NT], stk[MAX_NT];
    for (int i = 0; i < NT; ++i) {
        ubv[i] = P.ub[b * NT + i];
        stk[i] = P.stuck[b * NT + i];
    }
    // ---- work lists of the fp32 solve instantiations: one atomic per wave and list ----
    if (P.qlist && writes_rest) {
        int na = 0;
        for (int i = 0; i < NT; ++i) na += (ubv[i] > 0.0) ? 1 : 0;
        const int nbk = (N * na + 15) >> 4;
        int v = (na == 0 || nbk <= 8) ? 0 : nbk - 8;      // 0..2: one-wave kernels NB = 8 / 9 / 10; 3: workgroup kernel
        v = v > P.qvmax ? P.qvmax : v;
        const bool real = b0 + lane < P.B;
        for (int vv = 0; vv <= P.qvmax; ++vv) {
            const unsigned long long m = __ballot(real && v == vv);
            if (m == 0ull) continue;
            const int leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(P.qcount + vv, __popcll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            if (real && v == vv) P.qlist[(int64_t)vv * P.B + base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)b;
        }
    }
    const double* xref = P.xref + b * P.xref_stride;
    const double* uref = P.uref ? P.uref + b * P.uref_stride : nullptr;
    const double dt = C.dt;
    const double acoef[4] = {0.0, 0.5 * dt, 0.5 * dt, dt};
    const double wcoef[4] = {1.0, 2.0, 2.0, 1.0};

    for (int k = 0; k < N; ++k) {
        // total wrench gen = D (ubar + stuck), ubar = clip(warm, 0, ub) (0 for broken thrusters)
        double gen[6] = {0, 0, 0, 0, 0, 0}, rut[6];
        for (int i = 0; i < NT; ++i) {
            double u = 0.0;
            if (P.warmU && ubv[i] > 0.0) {
                u = P.warmU[(b * N + k) * NT + i];
                u = fmin(fmax(u, 0.0), ubv[i]);
            }
            const double t = u + stk[i];
            for (int g = 0; g < 6; ++g) gen[g] += C.D[g * MAX_NT + i] * t;
        }
        if (P.warmG) {
            for (int g = 0; g < 6; ++g) gen[g] = P.warmG[(b * N + k) * 6 + g];
        }
        // R .* (gen - ur - [f_virt;0]),  ur = [Rot(q_k)^T uref[0:3]; uref[3:6]]  (spiraling_mpc.py:156-171)
        {
            double ur[6] = {0, 0, 0, 0, 0, 0};
            if (uref) {
                double RT[9], f3[3] = {uref[6 * k], uref[6 * k + 1], uref[6 * k + 2]}, o[3];
                rotT(q, RT);
                mat3vec(RT, f3, o);
                ur[0] = o[0]; ur[1] = o[1]; ur[2] = o[2];
                ur[3] = uref[6 * k + 3]; ur[4] = uref[6 * k + 4]; ur[5] = uref[6 * k + 5];
            }
            for (int g = 0; g < 6; ++g) {
                const double fv = g < 3 ? C.fvirt[g] : 0.0;
                rut[g] = C.R[g] * (gen[g] - ur[g] - fv);
            }
        }
        const double* F = gen;
        const double* tau = gen + 3;

        // ---- RK4 stage points + blocks ----
        StageBlk S[4];
        double kw[4][3], kq[4][4], kv[4][3];
        for (int i = 0; i < 4; ++i) {
            double wi[3], qi[4];
            for (int j = 0; j < 3; ++j) wi[j] = w[j] + (i ? acoef[i] * kw[i - 1][j] : 0.0);
            for (int j = 0; j < 4; ++j) qi[j] = q[j] + (i ? acoef[i] * kq[i - 1][j] : 0.0);
            stage_eval(C, wi, qi, F, tau, kw[i], kq[i], kv[i], S[i]);
        }

        // ---- tangents for the 13 input directions [w0(3), q0(4), F(3), tau(3)] ----
        pk = k;
        for (int j = jlo; j < jhi; ++j) {
            pbase = (j < 7) ? 0 : REC_BPF;
            double tw0[3] = {0, 0, 0}, tq0[4] = {0, 0, 0, 0}, tF[3] = {0, 0, 0}, tT[3] = {0, 0, 0};
            if (j < 3) tw0[j] = 1.0;
            else if (j < 7) tq0[j - 3] = 1.0;
            else if (j < 10) tF[j - 7] = 1.0;
            else tT[j - 10] = 1.0;
            // wrench contributions common to all stage points
            double jt[3], at[3];
            mat3vec(C.Jinv, tT, jt);   // d wdot / d tau * tT
            mat3vec(C.ArT, tT, at);    // d a_b / d tau * tT
            for (int i = 0; i < 3; ++i) at[i] += tF[i] * C.inv_mass;
            double tkw[3] = {0, 0, 0}, tkq[4] = {0, 0, 0, 0};
            double sw[3] = {0, 0, 0}, sq[4] = {0, 0, 0, 0}, sv[3] = {0, 0, 0}, sp[3] = {0, 0, 0};
            for (int i = 0; i < 4; ++i) {
                double tw[3], tq[4];
                for (int a = 0; a < 3; ++a) tw[a] = tw0[a] + acoef[i] * tkw[a];
                for (int a = 0; a < 4; ++a) tq[a] = tq0[a] + acoef[i] * tkq[a];
                const StageBlk& B = S[i];
                double nkw[3], nkq[4], nkv[3], rv[3];
                for (int a = 0; a < 3; ++a)
                    nkw[a] = B.Fww[3 * a] * tw[0] + B.Fww[3 * a + 1] * tw[1] + B.Fww[3 * a + 2] * tw[2] + jt[a];
                for (int a = 0; a < 4; ++a)
                    nkq[a] = B.Fqw[3 * a] * tw[0] + B.Fqw[3 * a + 1] * tw[1] + B.Fqw[3 * a + 2] * tw[2] +
                             B.Fqq[4 * a] * tq[0] + B.Fqq[4 * a + 1] * tq[1] + B.Fqq[4 * a + 2] * tq[2] + B.Fqq[4 * a + 3] * tq[3];
                mat3vec(B.RT, at, rv);
                for (int a = 0; a < 3; ++a)
                    nkv[a] = B.Vw[3 * a] * tw[0] + B.Vw[3 * a + 1] * tw[1] + B.Vw[3 * a + 2] * tw[2] +
                             B.Vq[4 * a] * tq[0] + B.Vq[4 * a + 1] * tq[1] + B.Vq[4 * a + 2] * tq[2] + B.Vq[4 * a + 3] * tq[3] + rv[a];
                for (int a = 0; a < 3; ++a) { tkw[a] = nkw[a]; sw[a] += wcoef[i] * nkw[a]; }
                for (int a = 0; a < 4; ++a) { tkq[a] = nkq[a]; sq[a] += wcoef[i] * nkq[a]; }
                for (int a = 0; a < 3; ++a) { sv[a] += wcoef[i] * nkv[a]; if (i < 3) sp[a] += nkv[a]; }
            }
            const double c6 = dt / 6.0, c26 = dt * dt / 6.0;
            if (j < 3) {
                for (int a = 0; a < 3; ++a) {
                    put(REC_APW + 3 * a + j, c26 * sp[a]);
                    put(REC_AVW + 3 * a + j, c6 * sv[a]);
                    put(REC_AWW + 3 * a + j, tw0[a] + c6 * sw[a]);
                }
                for (int a = 0; a < 4; ++a) put(REC_AQW + 3 * a + j, c6 * sq[a]);
            } else if (j < 7) {
                const int jj = j - 3;
                for (int a = 0; a < 3; ++a) {
                    put(REC_APQ + 4 * a + jj, c26 * sp[a]);
                    put(REC_AVQ + 4 * a + jj, c6 * sv[a]);
                }
                for (int a = 0; a < 4; ++a) put(REC_AQQ + 4 * a + jj, tq0[a] + c6 * sq[a]);
            } else if (j < 10) {
                const int jj = j - 7;
                for (int a = 0; a < 3; ++a) {
                    put(REC_BPF - REC_BPF + 3 * a + jj, c26 * sp[a]);
                    put(REC_BVF - REC_BPF + 3 * a + jj, c6 * sv[a]);
                }
            } else {
                const int jj = j - 10;
                for (int a = 0; a < 3; ++a) {
                    put(REC_BPT - REC_BPF + 3 * a + jj, c26 * sp[a]);
                    put(REC_BVT - REC_BPF + 3 * a + jj, c6 * sv[a]);
                    put(REC_BWT - REC_BPF + 3 * a + jj, c6 * sw[a]);
                }
                for (int a = 0; a < 4; ++a) put(REC_BQT - REC_BPF + 3 * a + jj, c6 * sq[a]);
            }
            if (!DIRS && j == 6) flush(k, 0, REC_BPF);   // the A blocks (words 0..78) are complete
        }

        // ---- advance the nonlinear rollout (no quaternion renormalisation, sys_model.py:152-158) ----
        for (int a = 0; a < 3; ++a) {
            pos[a] += dt * vel[a] + dt * dt / 6.0 * (kv[0][a] + kv[1][a] + kv[2][a]);
            vel[a] += dt / 6.0 * (kv[0][a] + 2 * kv[1][a] + 2 * kv[2][a] + kv[3][a]);
        }
        for (int a = 0; a < 3; ++a) w[a] += dt / 6.0 * (kw[0][a] + 2 * kw[1][a] + 2 * kw[2][a] + kw[3][a]);
        for (int a = 0; a < 4; ++a) q[a] += dt / 6.0 * (kq[0][a] + 2 * kq[1][a] + 2 * kq[2][a] + kq[3][a]);

        // ---- the linearisation trajectory itself, for the state-bound rows (xlb <= c_{k+1} <= xub, spiraling_mpc.py:179-185) ----
        if (P.out_cbar && writes_rest && b0 + lane < P.B) {
            double* cb = P.out_cbar + (b * N + k) * 13;
            for (int a = 0; a < 3; ++a) {
                cb[a] = pos[a];
                cb[3 + a] = vel[a];
                cb[6 + a] = w[a];
            }
            for (int a = 0; a < 4; ++a) cb[9 + a] = q[a];
        }
        // ---- weighted tracking error of stage k+1: W (c[0:9] - xref) ----
        pbase = REC_BPF;
        if (writes_rest) {
            double e[9];
            for (int a = 0; a < 3; ++a) {
                e[a] = pos[a] - xref[9 * (k + 1) + a];
                e[3 + a] = vel[a] - xref[9 * (k + 1) + 3 + a];
                e[6 + a] = w[a] - xref[9 * (k + 1) + 6 + a];
            }
            if (k + 1 < N) {
                for (int a = 0; a < 9; ++a) put(REC_WE - REC_BPF + a, C.Q[a] * e[a]);
            } else {
                if (P.out_eN && b0 + lane < P.B) {
                    for (int a = 0; a < 9; ++a) P.out_eN[b * 9 + a] = e[a];
                }
                double gnq[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (P.tcost) (void)term_cost_nq(*P.tcost, e, gnq);      // exact gradient of the non-quadratic terminal terms
                for (int a = 0; a < 9; ++a) {
                    double s = 0.0;
                    for (int c = 0; c < 9; ++c) s += C.P[9 * a + c] * e[c];
                    put(REC_WE - REC_BPF + a, s + 0.5 * gnq[a]);
                }
            }
            for (int g = 0; g < 6; ++g) put(REC_RUT - REC_BPF + g, rut[g]);
            put(REC_USED - REC_BPF, 0.0);
        }
        flush(k, REC_BPF, REC_STRIDE - REC_BPF);   // B blocks, W e, R ut (words 79..151)
    }
}

template __global__ void ftmpc_linearize_kernel<double, 0>(const DeviceConsts, const LinParams);
template __global__ void ftmpc_linearize_kernel<double, 1>(const DeviceConsts, const LinParams);
template __global__ void ftmpc_linearize_kernel<double, 2>(const DeviceConsts, const LinParams);

// ---------------------------------------------------------------------------------------------------------
// Cost of the nonlinear program for given thruster sequences (merit function of the line-search SQP; include/ftmpc.h
// ftmpc_eval_cost_batch).  One lane per instance, float64.
// ---------------------------------------------------------------------------------------------------------
struct CostParams {
    int64_t B;
    const double* x0;
    const double* ub;
    const double* stuck;
    const double* xref;
    int64_t xref_stride;
    const double* uref;
    int64_t uref_stride;
    const double* U;        // [B*N*NT]
    const TermCost* tcost;  // or nullptr
    double* out;            // [B]  (ntrial > 0: [B*ntrial])
    // line search of the SQP, all trial points in one launch (ntrial > 0): lane b * ntrial + j evaluates the point
    // U + 2^-j (clip(Uq, 0, ub) - U) of instance b -- what ftmpc_sqp_trial_kernel would store for alpha = 2^-j -- for the instances with todo set
    const double* Uq = nullptr;
    const int32_t* todo = nullptr;
    int32_t ntrial = 0;
};

namespace {
// centre dynamics (spiral_model.py:44-76): s = [p, v, w, q]
__device__ inline void centre_rhs(const DeviceConsts& C, const double s[13], const double gen[6], double ds[13]) {
    const double* v = s + 3;
    const double* w = s + 6;
    const double* q = s + 9;
    double Jw[3], wxJw[3], t[3], dw[3];
    mat3vec(C.J, w, Jw);
    cross3(w, Jw, wxJw);
    for (int i = 0; i < 3; ++i) t[i] = gen[3 + i] - wxJw[i];
    mat3vec(C.Jinv, t, dw);
    double wxr[3], wxwxr[3], dwxr[3], ab[3], RT[9], dv[3];
    cross3(w, C.r, wxr);
    cross3(w, wxr, wxwxr);
    cross3(dw, C.r, dwxr);
    for (int i = 0; i < 3; ++i) ab[i] = gen[i] * C.inv_mass + dwxr[i] + wxwxr[i];
    rotT(q, RT);
    mat3vec(RT, ab, dv);
    for (int i = 0; i < 3; ++i) {
        ds[i] = v[i];
        ds[3 + i] = dv[i];
        ds[6 + i] = dw[i];
    }
    ds[9] = 0.5 * (w[2] * q[1] - w[1] * q[2] + w[0] * q[3]);
    ds[10] = 0.5 * (-w[2] * q[0] + w[0] * q[2] + w[1] * q[3]);
    ds[11] = 0.5 * (w[1] * q[0] - w[0] * q[1] + w[2] * q[3]);
    ds[12] = 0.5 * (-w[0] * q[0] - w[1] * q[1] - w[2] * q[2]);
}
}  // namespace

__global__ void __launch_bounds__(64) ftmpc_cost_kernel(const DeviceConsts C, const CostParams P) {
    const int64_t lane_id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = P.ntrial > 0 ? lane_id / P.ntrial : lane_id;
    if (b >= P.B) return;
    const int jt = P.ntrial > 0 ? (int)(lane_id - b * P.ntrial) : 0;
    if (P.ntrial > 0 && !P.todo[b]) return;
    const double alpha = ldexp(1.0, -jt);      // (a power of two: alpha * step is exact, the point is the one the trial kernel stores)
    const int N = C.N, NT = C.NT;
    double s[13];
    {
        double x[13];
        for (int i = 0; i < 13; ++i) x[i] = P.x0[b * 13 + i];
        double RT[9], wxr[3], a[3], c[3];
        rotT(x + 6, RT);
        cross3(x + 10, C.r, wxr);
        mat3vec(RT, C.r, a);
        mat3vec(RT, wxr, c);
        for (int i = 0; i < 3; ++i) {
            s[i] = x[i] + a[i];
            s[3 + i] = x[3 + i] + c[i];
            s[6 + i] = x[10 + i];
        }
        for (int i = 0; i < 4; ++i) s[9 + i] = x[6 + i];
    }
    const double* xref = P.xref + b * P.xref_stride;
    const double* uref = P.uref ? P.uref + b * P.uref_stride : nullptr;
    const double dt = C.dt;
    double cost = 0.0;
    for (int k = 0; k < N; ++k) {
        double gen[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < NT; ++i) {
            const bool healthy = P.ub[b * NT + i] > 0.0;
            double u = healthy ? P.U[(b * N + k) * NT + i] : 0.0;
            if (P.ntrial > 0 && healthy) u = u + alpha * (fmin(fmax(P.Uq[(b * N + k) * NT + i], 0.0), P.ub[b * NT + i]) - u);
            cost += C.rho * u * u;
            const double t = u + P.stuck[b * NT + i];
            for (int g = 0; g < 6; ++g) gen[g] += C.D[g * MAX_NT + i] * t;
        }
        double ur[6] = {0, 0, 0, 0, 0, 0};
        if (uref) {
            double RT[9], f3[3] = {uref[6 * k], uref[6 * k + 1], uref[6 * k + 2]}, o[3];
            rotT(s + 9, RT);
            mat3vec(RT, f3, o);
            ur[0] = o[0]; ur[1] = o[1]; ur[2] = o[2];
            ur[3] = uref[6 * k + 3]; ur[4] = uref[6 * k + 4]; ur[5] = uref[6 * k + 5];
        }
        for (int g = 0; g < 6; ++g) {
            const double ut = gen[g] - ur[g] - (g < 3 ? C.fvirt[g] : 0.0);
            cost += C.R[g] * ut * ut;
        }
        // RK4 (sys_model.py:152-158), no quaternion renormalisation
        double k1[13], k2[13], k3[13], k4[13], t[13];
        centre_rhs(C, s, gen, k1);
        for (int i = 0; i < 13; ++i) t[i] = s[i] + 0.5 * dt * k1[i];
        centre_rhs(C, t, gen, k2);
        for (int i = 0; i < 13; ++i) t[i] = s[i] + 0.5 * dt * k2[i];
        centre_rhs(C, t, gen, k3);
        for (int i = 0; i < 13; ++i) t[i] = s[i] + dt * k3[i];
        centre_rhs(C, t, gen, k4);
        for (int i = 0; i < 13; ++i) s[i] += dt / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        double e[9];
        for (int a = 0; a < 9; ++a) e[a] = s[a] - xref[9 * (k + 1) + a];
        if (k + 1 < N) {
            for (int a = 0; a < 9; ++a) cost += C.Q[a] * e[a] * e[a];
        } else {
            for (int a = 0; a < 9; ++a)
                for (int c = 0; c < 9; ++c) cost += e[a] * C.P[9 * a + c] * e[c];
            if (P.tcost) cost += term_cost_nq(*P.tcost, e, nullptr);
        }
    }
    P.out[lane_id] = cost;
}

}  // namespace ftmpc
