// ftmpc_solve_ric.hip -- kernel 12: the float64 thruster-space box QP with its Newton systems solved by the RICCATI
// RECURSION on the stage structure, ONE WAVE per instance.
//
// Kernel 9 (ftmpc_solve_ws64.hip) assembles and factorises K = I + L' S L (6N x 6N) in every interior-point iteration:
// O(N^3) work and, at N = 40, 79 MB of fabric traffic per instance (512 resident workgroups with 544 KB of tiles each do not
// fit the Infinity Cache).  But (H + Sigma) x = r, H = Bbar' (2 Wbar) Bbar + Rt (the condensed Hessian: Bbar the stacked
// sensitivities, Rt = 2 (D_a' R D_a + rho I) per stage), is the optimality system of the linear-quadratic problem
//     min  sum_k 1/2 x_{k+1}' Qt_{k+1} x_{k+1} + 1/2 u_k' (Rt + Sigma_k) u_k - r_k' u_k,   x_{k+1} = A_k x_k + Bt_k u_k,  x_0 = 0
// (Qt = 2 diag(Q) on the nine costed states, 2 P at the end; Bt_k = B_k D_a), whose solution is the backward Riccati sweep
//     S = Qt_{k+1} + P_{k+1},  Ruu = Rt + Sigma_k + Bt' S Bt,  Rux = Bt' S A,  Ruu = L L', W = L^-1, Y = W Rux,
//     P_k = A' S A - Y' Y                                                                         (P_N = 0)
// once per iteration, and per right-hand side a backward vector sweep (p_k = A' s - Y' w, w = W (Bt' s - r_k), s = p_{k+1})
// and a forward one (u_k = -W' (Y x_k + w), x_{k+1} = A x_k + Bt u_k).  Every matrix is ONE 16 x 16 float64 tile (13 states,
// at most 16 thrusters), the work is O(N), nothing is condensed and no Hessian exists: SAME QP, SAME Mehrotra iteration in
// the thruster variables (bounds, slacks, duals, step rules, gradient by recurrence through the Newton identity) as kernels
// 3 / 9 -- only the linear algebra of the Newton step changes, so oracle/ftmpc_oracle.c stays the checker.
//
// Tiles live in the float64 MFMA C/D layout (lane (q, col): rows q + 4 s of column col), in which X' Y is four
// v_mfma_f64_16x16x4 on the registers as they stand (hullk::mm_tn64) -- every product of the sweep above is of that form
// (S is symmetric; W' comes from a 16 x 17 LDS transpose).  Per stage the factors W_k, Y_k (2 x 2 KB) go to a per-wave global
// slot and are streamed back by the four vector sweeps of the iteration, a stage ahead of their use; A_k and Bt_k are rebuilt
// from the float64 stage record (1.2 KB) wherever they are needed.  Matrix-vector products run on the VALU with the two
// reductions the layout offers (DPP row sums for M x, permlane swaps for M' x).
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354; oracle/qp_oracle.py:ipm_box is the mirror.
#include <hip/hip_runtime.h>

#include "ftmpc_common.h"

namespace ftmpc {

namespace rick {
using f64k::f64x4;
using f64k::v64pos;

struct DAdd { static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct DMin { static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };
struct DMax { static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); } };

// reduction over the 16 lanes of a DPP row (float64: the two halves travel separately), result in all of them
template <class Op>
__device__ __forceinline__ double row_red16(double x) {
#define FTMPC_RIC_DPP(ctrl)                                                                                                          \
    {                                                                                                                                \
        const unsigned long long b_ = __builtin_bit_cast(unsigned long long, x);                                                     \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b_, ctrl, 0xf, 0xf, false);                     \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b_ >> 32), ctrl, 0xf, 0xf, false);             \
        x = Op::f(x, __builtin_bit_cast(double, ((unsigned long long)hi_ << 32) | lo_));                                             \
    }
    FTMPC_RIC_DPP(0x128)      // row_ror:8
    FTMPC_RIC_DPP(0x124)      // row_ror:4
    FTMPC_RIC_DPP(0x122)      // row_ror:2
    FTMPC_RIC_DPP(0x121)      // row_ror:1
#undef FTMPC_RIC_DPP
    return x;
}
// reduction over the four lanes that share lane & 15 (ftmpc_solve.hip quad_sum_d with any operation)
template <class Op>
__device__ __forceinline__ double quad_red(double x) {
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, x);
    unsigned la = (unsigned)bits, lb = la, ha = (unsigned)(bits >> 32), hb = ha;
    swap32(la, lb);
    swap32(ha, hb);
    const double s = Op::f(__builtin_bit_cast(double, ((unsigned long long)ha << 32) | la), __builtin_bit_cast(double, ((unsigned long long)hb << 32) | lb));
    const unsigned long long sb = __builtin_bit_cast(unsigned long long, s);
    unsigned lc = (unsigned)sb, ld = lc, hc = (unsigned)(sb >> 32), hd = hc;
    swap16(lc, ld);
    swap16(hc, hd);
    return Op::f(__builtin_bit_cast(double, ((unsigned long long)hc << 32) | lc), __builtin_bit_cast(double, ((unsigned long long)hd << 32) | ld));
}
template <class Op>
__device__ __forceinline__ double wave_red(double x) { return quad_red<Op>(row_red16<Op>(x)); }

// y = M x for a tile in the C/D layout: x in COLUMN layout (x[col] in every lane of column col), y in ROW layout
// (y[rr] = element q + 4 rr, in every lane of row group q)
__device__ __forceinline__ f64x4 mv(const f64x4& M, double xc) {
    return f64x4{row_red16<DAdd>(M.x * xc), row_red16<DAdd>(M.y * xc), row_red16<DAdd>(M.z * xc), row_red16<DAdd>(M.w * xc)};
}
// y = M x + N z
__device__ __forceinline__ f64x4 mv2(const f64x4& M, double xc, const f64x4& Nn, double zc) {
    return f64x4{row_red16<DAdd>(M.x * xc + Nn.x * zc), row_red16<DAdd>(M.y * xc + Nn.y * zc), row_red16<DAdd>(M.z * xc + Nn.z * zc),
                 row_red16<DAdd>(M.w * xc + Nn.w * zc)};
}
// partial products of y = M' x (x in row layout); quad_red<DAdd> of the result is y in column layout
__device__ __forceinline__ double mvt_part(const f64x4& M, const f64x4& xr) { return (M.x * xr.x + M.y * xr.y) + (M.z * xr.z + M.w * xr.w); }

// per-wave global slot, in doubles: per stage the tiles W_k | Y_k, then the interior-point state of the thruster variables
// (sl, su, zl, zu, grad, da: NSTATE arrays of NV x 64 doubles, variable (stage 4 v + q, thruster a) at v * 64 + lane)
// with state bounds eight more: slack / dual / carried primal residual of the upper and of the lower row of (stage, state
// component), the state-space part psi of the gradient and the predictor's state step
constexpr int NSTATE = 20;      // (+ six: the interior-point iterate kept while the early polish runs)
__host__ __device__ constexpr int64_t state_off(int N) { return (int64_t)N * 2 * 256; }
__host__ __device__ constexpr int64_t slot_doubles(int N) { return state_off(N) + (int64_t)NSTATE * ((N + 3) / 4) * 64; }
}  // namespace rick

struct SolveRicParams {
    SolveParams base;     // rec is double; qhead: shared instance cursor (zeroed by the host) or nullptr (static stride)
    double* slot;         // [gridDim.x][slot_doubles]: per stage the tiles W_k | Y_k in register order, then the interior-point state
    int64_t slot_doubles;
    // state bounds (template SB; reference spiraling_mpc.py:129-130,179-185): xlb <= c_j <= xub on the orbit-centre state of the
    // stages j = 1 .. N-1; |bound| >= 1e299 = no row.  cbar: [B*N*13] the linearisation trajectory c_1 .. c_N (ftmpc_linearize.hip)
    double xlb[13], xub[13];
    const double* cbar;
};

#ifndef FTMPC_RIC_WAVES
#define FTMPC_RIC_WAVES 2      // resident waves per SIMD (register budget 512 / FTMPC_RIC_WAVES)
#endif
// NV: N <= 4 NV.  Variable (stage k, thruster a) belongs to lane 16 (k & 3) + a, slot k >> 2 of that lane.  The interior-point state
// of the variables (six doubles each) lives in the per-wave GLOBAL slot, not in registers: it is touched only by the element-wise
// passes between the sweeps (1.5 % of the time), and out of the register file it leaves room for a second wave per SIMD -- every
// phase of this kernel is a dependent chain (MFMA accumulation, pivots through LDS, DPP reductions) that a second wave fills.
// SB: the reference's optional STATE BOUNDS as two more families of rows  +-dx_j <= h.  In this un-condensed form they are what
// box bounds on the states are in any optimal-control interior point: a diagonal barrier term Sx_j on the state weight of their
// stage and a state-linear term q_j of the Newton problem -- no dense rows through the sensitivities, no extra factorisation
// work.  The objective gradient is carried as  grad + Gbar' psi  (grad per thruster variable as before, psi_j per state of stage
// j, never condensed: the sweeps apply Gbar' implicitly), and both parts follow the step through the Newton identity:
//     grad += alpha (rhs_u - Sigma dd),     psi_j += alpha (-q_j - Sx_j dx_j).
template <int NV, bool SB = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FTMPC_RIC_WAVES, FTMPC_RIC_WAVES))) ftmpc_solve_ric64_kernel(const DeviceConsts C, const SolveRicParams Q) {
    using namespace rick;
    constexpr int NS = 4 * NV;
    __shared__ __attribute__((aligned(32))) double qxv[SB ? NS * 16 : 4];   // per stage the state-linear term q_j of the Newton problem (position 4 q + rr = element q + 4 rr)
    const SolveParams& P = Q.base;
    __shared__ __attribute__((aligned(32))) double recbuf[2][REC_STRIDE];
    __shared__ __attribute__((aligned(32))) double rvec[NS * 16];      // per stage 16 doubles: right-hand side in, solution out (natural order)
    __shared__ __attribute__((aligned(32))) double wst[NS * 16];       // per stage a row-layout vector (position 4 q + rr = element q + 4 rr)
    __shared__ __attribute__((aligned(32))) double tsc[16 * 17];       // transpose scratch
    __shared__ __attribute__((aligned(32))) double pcs[32];            // pivot column | row of E (f64k::potrf_inv16_lds)
    __shared__ __attribute__((aligned(32))) double vsc[2][16];         // vector layout conversions
    __shared__ double sDa[6 * MAX_NT];                                 // healthy columns of D
    __shared__ int s_act[MAX_NT];
    __shared__ int s_nat;

    const int N = C.N, NT = C.NT;
    const double rho = C.rho;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    double* const slot = Q.slot + (int64_t)blockIdx.x * Q.slot_doubles;

    // ---- per-lane structure of the transition matrix A (13 x 13) and of B (13 x 6) in the record: element (row q + 4 rr, col li) ----
    const int lane0 = threadIdx.x;
    int aoff[4];          // offset of A[row][col] in the record, or -1 (constant acst)
    double acst[4];
    int bfo[4], bto[4];   // offsets of B[row][0..2] (force part) and B[row][3..5] (torque part) in the record, or -1
    {
        const int li = lane0 & 15, lq = lane0 >> 4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = lq + 4 * rr, c = li;
            int off = -1;
            double cst = 0.0;
            if (r < 3) {              // p' = p + dt v + Apw w + Apq q
                if (c < 3) cst = (c == r) ? 1.0 : 0.0;
                else if (c < 6) cst = (c - 3 == r) ? C.dt : 0.0;
                else if (c < 9) off = REC_APW + 3 * r + (c - 6);
                else if (c < 13) off = REC_APQ + 4 * r + (c - 9);
            } else if (r < 6) {       // v' = v + Avw w + Avq q
                if (c >= 3 && c < 6) cst = (c == r) ? 1.0 : 0.0;
                else if (c >= 6 && c < 9) off = REC_AVW + 3 * (r - 3) + (c - 6);
                else if (c >= 9 && c < 13) off = REC_AVQ + 4 * (r - 3) + (c - 9);
            } else if (r < 9) {       // w' = Aww w
                if (c >= 6 && c < 9) off = REC_AWW + 3 * (r - 6) + (c - 6);
            } else if (r < 13) {      // q' = Aqw w + Aqq q
                if (c >= 6 && c < 9) off = REC_AQW + 3 * (r - 9) + (c - 6);
                else if (c >= 9 && c < 13) off = REC_AQQ + 4 * (r - 9) + (c - 9);
            }
            aoff[rr] = off;
            acst[rr] = cst;
            bfo[rr] = (r < 3) ? REC_BPF + 3 * r : (r < 6 ? REC_BVF + 3 * (r - 3) : -1);
            bto[rr] = (r < 3) ? REC_BPT + 3 * r : (r < 6 ? REC_BVT + 3 * (r - 3) : (r < 9 ? REC_BWT + 3 * (r - 6) : (r < 13 ? REC_BQT + 3 * (r - 9) : -1)));
        }
    }

    auto pull = [&]() -> int64_t {
        int i = 0;
        if (lane0 == 0) i = atomicAdd(P.qhead, 1);
        return (int64_t)__builtin_amdgcn_readfirstlane(i);
    };
    int64_t inst = P.qhead ? pull() : (int64_t)blockIdx.x;
    for (; inst < P.B; inst = P.qhead ? pull() : inst + gridDim.x) {
        wave_lds_fence();
        const int lane = lane_now();
        const int li = lane & 15, lq = lane >> 4;
        S64_DECL;
        S64_START();
        // ---------------- prologue ----------------
        if (lane == 0) {
            int na0 = 0;
            for (int i = 0; i < NT; ++i)
                if (P.ub[inst * NT + i] > 0.0) s_act[na0++] = i;
            for (int i = na0; i < MAX_NT; ++i) s_act[i] = 0;
            s_nat = na0;
        }
        wave_lds_fence();
        const int nat = s_nat;
        const int nt = N * nat;
        if (nat == 0 || N > NS) {      // nothing to optimise / shape not served by this instantiation (the host does not send it)
            for (int i = lane; i < NT; i += 64) P.out_u0[inst * NT + i] = 0.0;
            if (P.out_U)
                for (int i = lane; i < N * NT; i += 64) P.out_U[inst * (int64_t)N * NT + i] = 0.0;
            if (lane == 0) {
                if (P.status) P.status[inst] = (nat == 0) ? 0 : 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        for (int i = lane; i < 6 * MAX_NT; i += 64) {
            const int g = i / MAX_NT, a = i % MAX_NT;
            sDa[i] = (a < nat) ? C.D[g * MAX_NT + s_act[a]] : 0.0;
        }
        wave_lds_fence();
        double dac[6];      // column li of D_a
#pragma unroll
        for (int g = 0; g < 6; ++g) dac[g] = sDa[g * MAX_NT + li];
        // Rt = 2 (D_a' R D_a + rho I) on the healthy thrusters, identity on the padding (the diagonal blocks are factorised as 16 x 16)
        f64x4 Rt;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = lq + 4 * rr;
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < 6; ++g) t += sDa[g * MAX_NT + r] * C.R[g] * dac[g];
            t = 2.0 * (t + ((r == li) ? rho : 0.0));
            Rt[rr] = (r < nat && li < nat) ? t : ((r == li) ? 1.0 : 0.0);
        }
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        // this lane's thruster variables: (stage 4 v + lq, thruster li), v < NV
        const bool athr = li < nat;
        const int tact = s_act[li];
        const double ubl = athr ? P.ub[inst * NT + tact] : 1.0;      // upper bound of thruster li (the same for every stage)
        auto tvalid = [&](int v) { return athr && 4 * v + lq < N; };
        auto ubar_of = [&](int v) -> double {
            return (tvalid(v) && P.warmU) ? fmin(fmax(P.warmU[(inst * N + 4 * v + lq) * NT + tact], 0.0), ubl) : 0.0;
        };
        double* const st = slot + state_off(N);
        const int nv = (N + 3) >> 2;
        auto sref = [&](int arr, int v) -> double& { return st[(int64_t)(arr * nv + v) * 64 + lane]; };
        enum { S_SL = 0, S_SU = 1, S_ZL = 2, S_ZU = 3, S_GRAD = 4, S_DA = 5, X_SU = 6, X_ZU = 7, X_RU = 8, X_SL = 9, X_ZL = 10, X_RL = 11, X_PSI = 12, X_DXA = 13, K_SL = 14, K_SU = 15, K_ZL = 16, K_ZU = 17, K_GRAD = 18, K_DD = 19 };
        // state-bound rows of this lane: (stage 4 v + lq in 1 .. N-1, component li < 13), upper and lower
        const bool xhu = SB && li < 13 && Q.xub[li < 13 ? li : 0] < 1e299, xhl = SB && li < 13 && Q.xlb[li < 13 ? li : 0] > -1e299;
        auto xrow = [&](int v) { return li < 13 && 4 * v + lq >= 1 && 4 * v + lq <= N - 1; };
        // ---- stage record k (three 8-byte loads per lane) and, in the vector sweeps, the factors W_k, Y_k: requested TWO stages
        // ahead of their use into one of two register sets (a stage of a vector sweep is ~1.5 k cycles, a round trip to the
        // Infinity Cache / HBM under load more), committed to the LDS record buffer when their stage begins ----
        struct Pre {
            f64x4 W, Y;
            double r0, r1, r2;
        };
        auto request = [&](Pre& p, int k, bool tiles) {
            const double* r = recg + (int64_t)k * REC_STRIDE;
            p.r0 = r[lane];
            p.r1 = r[64 + lane];
            p.r2 = (128 + lane < REC_STRIDE) ? r[128 + lane] : 0.0;
            if (tiles) {
                p.W = *reinterpret_cast<const f64x4*>(slot + (int64_t)(2 * k) * 256 + 4 * lane);
                p.Y = *reinterpret_cast<const f64x4*>(slot + (int64_t)(2 * k + 1) * 256 + 4 * lane);
            }
        };
        auto commit = [&](const Pre& p, int buf) {
            wave_lds_fence();
            recbuf[buf][lane] = p.r0;
            recbuf[buf][64 + lane] = p.r1;
            if (128 + lane < REC_STRIDE) recbuf[buf][128 + lane] = p.r2;
            wave_lds_fence();
        };
        // A_k and Bt_k = B_k D_a as tiles, from the record in LDS
        auto stage_tiles = [&](const double* rb, f64x4& A, f64x4& Bt) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                A[rr] = (aoff[rr] >= 0) ? rb[aoff[rr]] : acst[rr];
                double b = 0.0;
                if (bfo[rr] >= 0) b += rb[bfo[rr]] * dac[0] + rb[bfo[rr] + 1] * dac[1] + rb[bfo[rr] + 2] * dac[2];
                if (bto[rr] >= 0) b += rb[bto[rr]] * dac[3] + rb[bto[rr] + 1] * dac[4] + rb[bto[rr] + 2] * dac[5];
                Bt[rr] = b;
            }
        };
        // weight tile of stage k + 1: 2 diag(Q) on the nine costed states, 2 P at the end of the horizon
        auto weight_tile = [&](bool terminal) -> f64x4 {
            f64x4 w = zero4;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = lq + 4 * rr;
                if (r < 9 && li < 9) w[rr] = terminal ? 2.0 * C.P[9 * r + li] : ((r == li) ? 2.0 * C.Q[r] : 0.0);
            }
            return w;
        };
        // vector layout conversions through LDS (one round trip)
        auto col2row = [&](double xc, int slotv) -> f64x4 {      // x[li] in every lane of column li  ->  x[q + 4 rr]
            wave_lds_fence();
            if (lq == 0) vsc[slotv][v64pos(li)] = xc;
            wave_lds_fence();
            return *reinterpret_cast<const f64x4*>(&vsc[slotv][4 * lq]);
        };
        auto row2col = [&](const f64x4& xr, int slotv) -> double {
            wave_lds_fence();
            if (li == 0) *reinterpret_cast<f64x4*>(&vsc[slotv][4 * lq]) = xr;
            wave_lds_fence();
            return vsc[slotv][v64pos(li)];
        };

        // ---- the Newton problem  min 1/2 x'(Qt + Sx) x [+ q'x] + 1/2 u'(Rt + Sigma) u - r'u  over the stored factors: r in rvec [q in qxv]
        // on entry, the minimiser u in rvec on exit [and, SB, its states x_{k+1} in wst[k]] ----
        auto ric_solve = [&]() {
            Pre p0, p1;
            // backward: s = p_{k+1} (row layout), w_k = W (Bt' s - r_k) kept for the forward sweep
            f64x4 s = zero4;
            auto bstage = [&](int k, Pre& p) {
                commit(p, k & 1);
                const f64x4 Wk = p.W, Yk = p.Y;
                if (k >= 2) request(p, k - 2, true);
                f64x4 A, Bt;
                stage_tiles(recbuf[k & 1], A, Bt);
                if constexpr (SB) {      // s = p_{k+1} + q_{k+1}
                    if (k + 1 <= N - 1) s += *reinterpret_cast<const f64x4*>(&qxv[(k + 1) * 16 + 4 * lq]);
                }
                const double ru = quad_red<DAdd>(mvt_part(Bt, s)) - rvec[k * 16 + li];
                const f64x4 w = mv(Wk, ru);
                if (li == 0) *reinterpret_cast<f64x4*>(&wst[k * 16 + 4 * lq]) = w;
                const double pc = quad_red<DAdd>(mvt_part(A, s) - mvt_part(Yk, w));
                s = col2row(pc, k & 1);
            };
            request(p0, N - 1, true);
            if (N >= 2) request(p1, N - 2, true);
            for (int k = N - 1; k >= 0; k -= 2) {
                bstage(k, p0);
                if (k >= 1) bstage(k - 1, p1);
            }
            S64(2);
            // forward: u_k = -W' (Y x_k + w_k),  x_{k+1} = A x_k + Bt u_k
            double xc = 0.0;
            auto fstage = [&](int k, Pre& p) {
                commit(p, k & 1);
                const f64x4 Wk = p.W, Yk = p.Y;
                if (k + 2 < N) request(p, k + 2, true);
                f64x4 A, Bt;
                stage_tiles(recbuf[k & 1], A, Bt);
                const f64x4 w = *reinterpret_cast<const f64x4*>(&wst[k * 16 + 4 * lq]);
                const f64x4 v = mv(Yk, xc) + w;
                const double uc = -quad_red<DAdd>(mvt_part(Wk, v));
                if (lq == 0) rvec[k * 16 + li] = uc;
                const f64x4 xn = mv2(A, xc, Bt, uc);
                xc = row2col(xn, k & 1);
                if constexpr (SB) {
                    if (lq == 0) wst[k * 16 + li] = xc;      // x_{k+1} of the solution, natural order (w_k has been consumed)
                }
            };
            request(p0, 0, true);
            if (N >= 2) request(p1, 1, true);
            for (int k = 0; k < N; k += 2) {
                fstage(k, p0);
                if (k + 1 < N) fstage(k + 1, p1);
            }
            wave_lds_fence();
            S64(3);
        };
        // ---------------- start point (box centre) and its gradient ----------------
        //   g_k = Bt_k' lam_{k+1} + Rt d_k + 2 D_a' (R ut_k) + 2 rho ubar_k,   lam_{k+1} = Qt_{k+1} dx_{k+1} + 2 W e_{k+1} + A_{k+1}' lam_{k+2}
        wave_lds_fence();
        for (int v = 0; v < nv; ++v) {      // d = box centre - ubar
            const bool ok = tvalid(v);
            rvec[(4 * v + lq) * 16 + li] = ok ? 0.5 * ubl - ubar_of(v) : 0.0;
        }
        wave_lds_fence();
        double hs = 0.0;
        {
            Pre p0, p1;
            // forward: dx_{k+1} = A dx_k + Bt d_k, kept in natural order in wst (stage k + 1 at slot k)
            double xc = 0.0;
            auto gf = [&](int k, Pre& p) {
                commit(p, k & 1);
                if (k + 2 < N) request(p, k + 2, false);
                f64x4 A, Bt;
                stage_tiles(recbuf[k & 1], A, Bt);
                const f64x4 xn = mv2(A, xc, Bt, rvec[k * 16 + li]);
                xc = row2col(xn, k & 1);
                if (lq == 0) wst[k * 16 + li] = xc;
            };
            request(p0, 0, false);
            if (N >= 2) request(p1, 1, false);
            for (int k = 0; k < N; k += 2) {
                gf(k, p0);
                if (k + 1 < N) gf(k + 1, p1);
            }
            // backward (alongside: the open-loop weight S_j = Qt_j + A_j' S_{j+1} A_j, whose diag(Rt + Bt' S Bt) is the diagonal of the
            // condensed Hessian -- the scale of the polish's penalty)
            f64x4 mu = zero4, So = zero4;
            auto gb = [&](int k, Pre& p) {
                commit(p, k & 1);
                if (k >= 2) request(p, k - 2, false);
                const double* rb = recbuf[k & 1];
                f64x4 A, Bt;
                stage_tiles(rb, A, Bt);
                const f64x4 Wt = weight_tile(k + 1 == N);
                f64x4 lam = mv(Wt, wst[k * 16 + li]) + mu;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = lq + 4 * rr;
                    if (r < 9) lam[rr] += 2.0 * rb[REC_WE + r];
                }
                // d_k in row layout for Rt d_k (Rt is symmetric: Rt' d = Rt d)
                f64x4 drow;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) drow[rr] = rvec[k * 16 + lq + 4 * rr];
                double gl = 0.0;
#pragma unroll
                for (int g = 0; g < 6; ++g) gl += dac[g] * rb[REC_RUT + g];
                const double gk = quad_red<DAdd>(mvt_part(Bt, lam) + mvt_part(Rt, drow)) + 2.0 * gl;
                const double muc = quad_red<DAdd>(mvt_part(A, lam));
                mu = col2row(muc, k & 1);
                if constexpr (!SB) {
                    const f64x4 S = So + Wt;
                    const f64x4 SBt = hullk::mm_tn64(S, Bt, zero4), SA = hullk::mm_tn64(S, A, zero4);
                    const f64x4 Hd = hullk::mm_tn64(Bt, SBt, Rt);
                    So = hullk::mm_tn64(A, SA, zero4);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (lq + 4 * rr == li && li < nat) hs = fmax(hs, Hd[rr]);
                }
                wave_lds_fence();
                if (lq == 0) rvec[k * 16 + li] = gk;      // (d_k has been consumed)
            };
            request(p0, N - 1, false);
            if (N >= 2) request(p1, N - 2, false);
            for (int k = N - 1; k >= 0; k -= 2) {
                gb(k, p0);
                if (k >= 1) gb(k - 1, p1);
            }
            wave_lds_fence();
        }
        hs = wave_red<DMax>(hs);
        S64(0);

        // ---------------- interior-point iterations (thruster space; the iteration of kernels 3 / 9) ----------------
        int status = 1, nit = 0;
        double mrows = (double)(2 * nt);
        {
            // state at the start: slacks at the box centre, gradient from the sweep above, duals on the central path
            double gm = 0.0;
            for (int v = 0; v < nv; ++v) {
                const bool ok = tvalid(v);
                const double g = ok ? rvec[(4 * v + lq) * 16 + li] + 2.0 * rho * ubar_of(v) : 0.0;
                sref(S_GRAD, v) = g;
                sref(S_SL, v) = 0.5 * ubl;
                sref(S_SU, v) = 0.5 * ubl;
                gm = fmax(gm, fabs(g));
            }
            gm = wave_red<DMax>(gm);
            double wm = athr ? ubl : 0.0;
            if constexpr (SB) {      // state rows: slack max(residual, 0.1) and the primal residual the steps shrink (as the terminal rows of kernel 3)
                double cnt = 0.0;
                for (int v = 0; v < nv; ++v) {
                    const int j = 4 * v + lq;
                    const bool in = xrow(v);
                    const double dx = in ? wst[(j - 1) * 16 + li] : 0.0;      // dx_j at the start point (forward sweep above)
                    const double cb = in ? Q.cbar[(inst * N + (j - 1)) * 13 + li] : 0.0;
                    double su = 1.0, ru = 0.0, sl = 1.0, rl = 0.0;
                    if (in && xhu) {
                        const double res = Q.xub[li] - cb - dx;
                        su = fmax(res, 0.1);
                        ru = su - res;
                        wm = fmax(wm, su);
                        cnt += 1.0;
                    }
                    if (in && xhl) {
                        const double res = cb + dx - Q.xlb[li];
                        sl = fmax(res, 0.1);
                        rl = sl - res;
                        wm = fmax(wm, sl);
                        cnt += 1.0;
                    }
                    sref(X_SU, v) = su;
                    sref(X_RU, v) = ru;
                    sref(X_SL, v) = sl;
                    sref(X_RL, v) = rl;
                    sref(X_PSI, v) = 0.0;
                }
                mrows += wave_red<DAdd>(cnt);
            }
            wm = wave_red<DMax>(wm);
            const double mu0 = fmax(0.02 * gm * wm, 1e-3);
            for (int v = 0; v < nv; ++v) {
                const double z = tvalid(v) ? mu0 / (0.5 * ubl) : 0.0;
                sref(S_ZL, v) = z;
                sref(S_ZU, v) = z;
                if constexpr (SB) {
                    sref(X_ZU, v) = (xrow(v) && xhu) ? mu0 / sref(X_SU, v) : 0.0;
                    sref(X_ZL, v) = (xrow(v) && xhl) ? mu0 / sref(X_SL, v) : 0.0;
                }
            }
        }
        // ---- backward Riccati sweep with Sigma in rvec [Sx in wst]: the factors W_k, Y_k of every stage -> global slot ----
        auto ric_factor = [&]() -> bool {
            bool ok = true;
            {
                f64x4 Pm = zero4;
                Pre p0, p1;
                request(p0, N - 1, false);
                if (N >= 2) request(p1, N - 2, false);
                auto fac = [&](int k, Pre& p) {
                    commit(p, k & 1);
                    if (k >= 2) request(p, k - 2, false);
                    f64x4 A, Bt;
                    stage_tiles(recbuf[k & 1], A, Bt);
                    f64x4 S = Pm + weight_tile(k + 1 == N);
                    if constexpr (SB) {
                        if (k + 1 <= N - 1) {
                            const double sx = wst[(k + 1) * 16 + li];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr)
                                if (lq + 4 * rr == li) S[rr] += sx;
                        }
                    }
                    const f64x4 SA = hullk::mm_tn64(S, A, zero4), SBt = hullk::mm_tn64(S, Bt, zero4);      // (S is symmetric)
                    f64x4 Ruu = hullk::mm_tn64(Bt, SBt, Rt);
                    const f64x4 Rux = hullk::mm_tn64(Bt, SA, zero4);
                    const f64x4 PA = hullk::mm_tn64(A, SA, zero4);
                    const double sg = rvec[k * 16 + li];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (lq + 4 * rr == li) Ruu[rr] += sg;
                    double c[4] = {Ruu.x, Ruu.y, Ruu.z, Ruu.w}, w[4], l[4];
                    ok = f64k::potrf_inv16_lds(c, pcs, pcs + 16, lq, li, w, l, nat) && ok;      // (padding beyond the healthy thrusters: no pivot steps)
                    const f64x4 Wk = {w[0], w[1], w[2], w[3]};
                    wave_lds_fence();
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) tsc[(lq + 4 * rr) * 17 + li] = w[rr];
                    wave_lds_fence();
                    f64x4 Wt;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) Wt[rr] = tsc[li * 17 + lq + 4 * rr];
                    wave_lds_fence();
                    const f64x4 Yk = hullk::mm_tn64(Wt, Rux, zero4);      // W Rux
                    Pm = PA - hullk::mm_tn64(Yk, Yk, zero4);
                    *reinterpret_cast<f64x4*>(slot + (int64_t)(2 * k) * 256 + 4 * lane) = Wk;
                    *reinterpret_cast<f64x4*>(slot + (int64_t)(2 * k + 1) * 256 + 4 * lane) = Yk;
                };
                for (int k = N - 1; k >= 0; k -= 2) {
                    fac(k, p0);
                    if (k >= 1) fac(k - 1, p1);
                }
            }
            wave_global_fence();
            return __all(ok);
        };
        const double inv2n = 1.0 / mrows;
        bool polish_tried = false;
        for (int it = 0; it <= C.max_iters; ++it) {
            // mu [and the carried primal residual], and the barrier weights of this iterate: Sigma -> rvec [Sx -> wst] (the Riccati sweep reads them per stage)
            double csum = 0.0, rpn = 0.0;
            wave_lds_fence();
            for (int v = 0; v < nv; ++v) {
                const bool ok = tvalid(v);
                const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v);
                if (ok) csum += sl * zl + su * zu;
                rvec[(4 * v + lq) * 16 + li] = ok ? zl / sl + zu / su : 0.0;
                if constexpr (SB) {
                    double sx = 0.0;
                    if (xrow(v) && xhu) {
                        const double s = sref(X_SU, v), z = sref(X_ZU, v);
                        csum += s * z;
                        sx += z / s;
                        rpn = fmax(rpn, fabs(sref(X_RU, v)));
                    }
                    if (xrow(v) && xhl) {
                        const double s = sref(X_SL, v), z = sref(X_ZL, v);
                        csum += s * z;
                        sx += z / s;
                        rpn = fmax(rpn, fabs(sref(X_RL, v)));
                    }
                    wst[(4 * v + lq) * 16 + li] = sx;
                }
            }
            wave_lds_fence();
            const double mu = wave_red<DAdd>(csum) * inv2n;
            if constexpr (SB) rpn = wave_red<DMax>(rpn);
            if (!(mu == mu) || !(rpn == rpn)) {
                status = 2;
                break;
            }
            if (!(mu >= C.mu_stop) && !(rpn >= 1e-9)) {
                status = 0;
                break;
            }
            if constexpr (!SB) {
                // ---- EARLY ACTIVE-SET POLISH (box rows): once mu < 1e-7 the bounds with z > s are taken as active and the problem on that
                // set is solved exactly by multiplier steps with the penalty 1e6 hs on the active bounds (the Newton matrix's own shape:
                // Sigma = penalty on the active variables, 0 elsewhere), signs verified, at most three rounds -- as polish_general does for
                // the general rows.  Verified: done, with the exact solution and 3-4 interior-point iterations saved (12.9 -> 9.4 on
                // config 5).  Not verified: the iterate kept aside is restored and the iteration runs on to mu_stop as before.
                if (!polish_tried && mu < 1e-7) {
                    polish_tried = true;
                    const double pw = 1e6 * hs;
                    unsigned actl = 0u, actu = 0u;
                    for (int v = 0; v < nv; ++v) {
                        const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v);
                        sref(K_SL, v) = sl;
                        sref(K_SU, v) = su;
                        sref(K_ZL, v) = zl;
                        sref(K_ZU, v) = zu;
                        sref(K_GRAD, v) = sref(S_GRAD, v);
                        const bool al = tvalid(v) && zl > sl, au = tvalid(v) && zu > su;
                        actl |= al ? 1u << v : 0u;
                        actu |= au ? 1u << v : 0u;
                        sref(S_ZL, v) = al ? zl : 0.0;      // multipliers of the inactive bounds: 0
                        sref(S_ZU, v) = au ? zu : 0.0;
                    }
                    bool verified = false;
                    int rounds = 0;
                    for (int rd = 0; rd < 3 && !verified; ++rd) {
                        wave_lds_fence();
                        for (int v = 0; v < nv; ++v)
                            rvec[(4 * v + lq) * 16 + li] = ((actl >> v & 1u) ? pw : 0.0) + ((actu >> v & 1u) ? pw : 0.0);
                        wave_lds_fence();
                        if (__builtin_amdgcn_readfirstlane(!ric_factor())) break;
                        ++rounds;
                        for (int in = 0; in < 2; ++in) {
                            // (H + Sigma_A) dd = -grad + C_A' (W s_A - lam):  lower row c = -e, upper row c = +e
                            wave_lds_fence();
                            for (int v = 0; v < nv; ++v) {
                                double r = 0.0;
                                if (tvalid(v)) {
                                    r = -sref(S_GRAD, v);
                                    if (actl >> v & 1u) r -= pw * sref(S_SL, v) - sref(S_ZL, v);
                                    if (actu >> v & 1u) r += pw * sref(S_SU, v) - sref(S_ZU, v);
                                }
                                rvec[(4 * v + lq) * 16 + li] = r;
                                sref(K_DD, v) = r;
                            }
                            wave_lds_fence();
                            ric_solve();
                            for (int v = 0; v < nv; ++v)
                                if (tvalid(v)) {
                                    const bool al = actl >> v & 1u, au = actu >> v & 1u;
                                    const double sl = sref(S_SL, v), su = sref(S_SU, v);
                                    const double dd = rvec[(4 * v + lq) * 16 + li];
                                    sref(S_GRAD, v) += sref(K_DD, v) - ((al ? pw : 0.0) + (au ? pw : 0.0)) * dd;      // + H dd (Newton identity)
                                    if (al) sref(S_ZL, v) += pw * (-dd - sl);
                                    if (au) sref(S_ZU, v) += pw * (dd - su);
                                    sref(S_SL, v) = sl + dd;
                                    sref(S_SU, v) = su - dd;
                                }
                        }
                        bool changed = false;
                        for (int v = 0; v < nv; ++v)
                            if (tvalid(v)) {
                                // an active bound whose multiplier came out negative leaves; an inactive bound that is violated enters
                                if (actl >> v & 1u) {
                                    if (sref(S_ZL, v) < 0.0) { sref(S_ZL, v) = 0.0; actl &= ~(1u << v); changed = true; }
                                } else if (sref(S_SL, v) < -1e-10) { actl |= 1u << v; changed = true; }
                                if (actu >> v & 1u) {
                                    if (sref(S_ZU, v) < 0.0) { sref(S_ZU, v) = 0.0; actu &= ~(1u << v); changed = true; }
                                } else if (sref(S_SU, v) < -1e-10) { actu |= 1u << v; changed = true; }
                            }
                        verified = !__any(changed);
                    }
                    nit += rounds;
                    if (verified) {
                        status = 0;
                        break;
                    }
                    for (int v = 0; v < nv; ++v) {      // not settled: back to the interior-point iterate
                        sref(S_SL, v) = sref(K_SL, v);
                        sref(S_SU, v) = sref(K_SU, v);
                        sref(S_ZL, v) = sref(K_ZL, v);
                        sref(S_ZU, v) = sref(K_ZU, v);
                        sref(S_GRAD, v) = sref(K_GRAD, v);
                    }
                    wave_lds_fence();
                    for (int v = 0; v < nv; ++v) {
                        const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v);
                        rvec[(4 * v + lq) * 16 + li] = tvalid(v) ? zl / sl + zu / su : 0.0;
                    }
                    wave_lds_fence();
                }
            }
            if (it == C.max_iters) break;
            ++nit;
            S64(4);
            const bool fok = ric_factor();
            S64(1);
            if (__builtin_amdgcn_readfirstlane(!fok)) {
                // (state bounds: Sx ~ 1 / mu on an active row enters S, and P_k = A'SA - Y'Y is then a difference of numbers of that
                // size: the recursion runs out of digits near mu ~ 1e-11.  As kernel 3's general-constraint modes: a breakdown
                // once mu < 1e-7 with the primal residual closed ends the iteration as converged)
                status = (SB && mu < 1e-7 && rpn < 1e-9) ? 0 : 2;
                --nit;
                break;
            }
            // predictor: (H + Sig) da = -grad
            wave_lds_fence();
            for (int v = 0; v < nv; ++v) {
                rvec[(4 * v + lq) * 16 + li] = tvalid(v) ? -sref(S_GRAD, v) : 0.0;
                if constexpr (SB) {      // q = psi + Wxu rp_u - Wxl rp_l  (predictor: no second-order term)
                    double q = 0.0;
                    if (xrow(v)) {
                        q = sref(X_PSI, v);
                        if (xhu) q += sref(X_ZU, v) / sref(X_SU, v) * sref(X_RU, v);
                        if (xhl) q -= sref(X_ZL, v) / sref(X_SL, v) * sref(X_RL, v);
                    }
                    qxv[(4 * v + lq) * 16 + v64pos(li)] = q;
                }
            }
            wave_lds_fence();
            ric_solve();
            double ap = 1.0, ad = 1.0;
            for (int v = 0; v < nv; ++v) {
                const bool okv = tvalid(v);
                const double da = okv ? rvec[(4 * v + lq) * 16 + li] : 0.0;
                sref(S_DA, v) = da;
                if (okv) {
                    const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v);
                    const double dzl_a = -zl - zl * da / sl;
                    const double dzu_a = -zu + zu * da / su;
                    if (da < 0.0) ap = fmin(ap, -sl / da);
                    if (da > 0.0) ap = fmin(ap, su / da);
                    if (dzl_a < 0.0) ad = fmin(ad, -zl / dzl_a);
                    if (dzu_a < 0.0) ad = fmin(ad, -zu / dzu_a);
                }
                if constexpr (SB) {      // ds = -rp -+ dx,  dz = -z - z ds / s
                    double dxa = 0.0;
                    if (xrow(v)) {
                        dxa = wst[(4 * v + lq - 1) * 16 + li];
                        if (xhu) {
                            const double s = sref(X_SU, v), z = sref(X_ZU, v), ds = -sref(X_RU, v) - dxa, dz = -z - z * ds / s;
                            if (ds < 0.0) ap = fmin(ap, -s / ds);
                            if (dz < 0.0) ad = fmin(ad, -z / dz);
                        }
                        if (xhl) {
                            const double s = sref(X_SL, v), z = sref(X_ZL, v), ds = -sref(X_RL, v) + dxa, dz = -z - z * ds / s;
                            if (ds < 0.0) ap = fmin(ap, -s / ds);
                            if (dz < 0.0) ad = fmin(ad, -z / dz);
                        }
                    }
                    sref(X_DXA, v) = dxa;
                }
            }
            ap = wave_red<DMin>(ap);
            ad = wave_red<DMin>(ad);
            csum = 0.0;
            for (int v = 0; v < nv; ++v) {
                if (tvalid(v)) {
                    const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v), da = sref(S_DA, v);
                    const double dzl_a = -zl - zl * da / sl;
                    const double dzu_a = -zu + zu * da / su;
                    csum += (sl + ap * da) * (zl + ad * dzl_a) + (su - ap * da) * (zu + ad * dzu_a);
                }
                if constexpr (SB) {
                    if (xrow(v)) {
                        const double dxa = sref(X_DXA, v);
                        if (xhu) {
                            const double s = sref(X_SU, v), z = sref(X_ZU, v), ds = -sref(X_RU, v) - dxa, dz = -z - z * ds / s;
                            csum += (s + ap * ds) * (z + ad * dz);
                        }
                        if (xhl) {
                            const double s = sref(X_SL, v), z = sref(X_ZL, v), ds = -sref(X_RL, v) + dxa, dz = -z - z * ds / s;
                            csum += (s + ap * ds) * (z + ad * dz);
                        }
                    }
                }
            }
            const double mu_aff = wave_red<DAdd>(csum) * inv2n;
            double sigma = mu_aff / mu;
            sigma = fmin(fmax(sigma * sigma * sigma, 0.0), 1.0);
            const double sm = sigma * mu;
            // corrector: rc = s z + ds_a dz_a - sigma mu;  rhs = -(grad - zl + zu) - rcl / sl + rcu / su
            wave_lds_fence();
            for (int v = 0; v < nv; ++v) {
                double rhs = 0.0;
                if (tvalid(v)) {
                    const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v), da = sref(S_DA, v);
                    const double dzl_a = -zl - zl * da / sl;
                    const double dzu_a = -zu + zu * da / su;
                    const double rcl = sl * zl + da * dzl_a - sigma * mu;
                    const double rcu = su * zu - da * dzu_a - sigma * mu;
                    rhs = -(sref(S_GRAD, v) - zl + zu) - rcl / sl + rcu / su;
                }
                rvec[(4 * v + lq) * 16 + li] = rhs;
                if constexpr (SB) {      // q = psi + Wxu rp_u - Wxl rp_l - kappa_xu + kappa_xl,  kappa = (ds_a dz_a - sigma mu) / s
                    double q = 0.0;
                    if (xrow(v)) {
                        const double dxa = sref(X_DXA, v);
                        q = sref(X_PSI, v);
                        if (xhu) {
                            const double s = sref(X_SU, v), z = sref(X_ZU, v), rp = sref(X_RU, v), ds = -rp - dxa, dz = -z - z * ds / s;
                            q += z / s * rp - (ds * dz - sm) / s;
                        }
                        if (xhl) {
                            const double s = sref(X_SL, v), z = sref(X_ZL, v), rp = sref(X_RL, v), ds = -rp + dxa, dz = -z - z * ds / s;
                            q -= z / s * rp - (ds * dz - sm) / s;
                        }
                    }
                    qxv[(4 * v + lq) * 16 + v64pos(li)] = q;
                }
            }
            wave_lds_fence();
            ric_solve();
            ap = 1e300;
            ad = 1e300;
            for (int v = 0; v < nv; ++v)
                if (tvalid(v)) {
                    const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v), da = sref(S_DA, v);
                    const double dd = rvec[(4 * v + lq) * 16 + li];
                    const double dzl_a = -zl - zl * da / sl;
                    const double dzu_a = -zu + zu * da / su;
                    const double rcl = sl * zl + da * dzl_a - sigma * mu;
                    const double rcu = su * zu - da * dzu_a - sigma * mu;
                    const double dzl = (-rcl - zl * dd) / sl;
                    const double dzu = (-rcu + zu * dd) / su;
                    if (dd < 0.0) ap = fmin(ap, -sl / dd);
                    if (dd > 0.0) ap = fmin(ap, su / dd);
                    if (dzl < 0.0) ad = fmin(ad, -zl / dzl);
                    if (dzu < 0.0) ad = fmin(ad, -zu / dzu);
                }
            if constexpr (SB) {
                for (int v = 0; v < nv; ++v)
                    if (xrow(v)) {
                        const double dx = wst[(4 * v + lq - 1) * 16 + li], dxa = sref(X_DXA, v);
                        if (xhu) {
                            const double s = sref(X_SU, v), z = sref(X_ZU, v), rp = sref(X_RU, v), dsa = -rp - dxa, dza = -z - z * dsa / s;
                            const double ds = -rp - dx, dz = (-(s * z + dsa * dza - sm) - z * ds) / s;
                            if (ds < 0.0) ap = fmin(ap, -s / ds);
                            if (dz < 0.0) ad = fmin(ad, -z / dz);
                        }
                        if (xhl) {
                            const double s = sref(X_SL, v), z = sref(X_ZL, v), rp = sref(X_RL, v), dsa = -rp + dxa, dza = -z - z * dsa / s;
                            const double ds = -rp + dx, dz = (-(s * z + dsa * dza - sm) - z * ds) / s;
                            if (ds < 0.0) ap = fmin(ap, -s / ds);
                            if (dz < 0.0) ad = fmin(ad, -z / dz);
                        }
                    }
            }
            ap = fmin(1.0, 0.9995 * wave_red<DMin>(ap));
            ad = fmin(1.0, 0.9995 * wave_red<DMin>(ad));
            if constexpr (SB) {      // rows and psi follow the step: psi += alpha (-q - Sx dx)
                for (int v = 0; v < nv; ++v)
                    if (xrow(v)) {
                        const double dx = wst[(4 * v + lq - 1) * 16 + li], dxa = sref(X_DXA, v);
                        double q = sref(X_PSI, v), sx = 0.0;
                        if (xhu) {
                            const double s = sref(X_SU, v), z = sref(X_ZU, v), rp = sref(X_RU, v), dsa = -rp - dxa, dza = -z - z * dsa / s;
                            const double ds = -rp - dx, dz = (-(s * z + dsa * dza - sm) - z * ds) / s;
                            q += z / s * rp - (dsa * dza - sm) / s;
                            sx += z / s;
                            sref(X_SU, v) = s + ap * ds;
                            sref(X_ZU, v) = z + ad * dz;
                            sref(X_RU, v) = (1.0 - ap) * rp;
                        }
                        if (xhl) {
                            const double s = sref(X_SL, v), z = sref(X_ZL, v), rp = sref(X_RL, v), dsa = -rp + dxa, dza = -z - z * dsa / s;
                            const double ds = -rp + dx, dz = (-(s * z + dsa * dza - sm) - z * ds) / s;
                            q -= z / s * rp - (dsa * dza - sm) / s;
                            sx += z / s;
                            sref(X_SL, v) = s + ap * ds;
                            sref(X_ZL, v) = z + ad * dz;
                            sref(X_RL, v) = (1.0 - ap) * rp;
                        }
                        sref(X_PSI, v) += ap * (-q - sx * dx);
                    }
            }
            for (int v = 0; v < nv; ++v)
                if (tvalid(v)) {
                    const double sl = sref(S_SL, v), su = sref(S_SU, v), zl = sref(S_ZL, v), zu = sref(S_ZU, v), da = sref(S_DA, v);
                    const double dd = rvec[(4 * v + lq) * 16 + li];
                    const double dzl_a = -zl - zl * da / sl;
                    const double dzu_a = -zu + zu * da / su;
                    const double rcl = sl * zl + da * dzl_a - sigma * mu;
                    const double rcu = su * zu - da * dzu_a - sigma * mu;
                    const double rhs = -(sref(S_GRAD, v) - zl + zu) - rcl / sl + rcu / su;
                    const double Sig = zl / sl + zu / su;
                    sref(S_GRAD, v) += ap * (rhs - Sig * dd);   // + ap H dd
                    sref(S_SL, v) = sl + ap * dd;
                    sref(S_SU, v) = su - ap * dd;
                    sref(S_ZL, v) = zl + ad * (-rcl - zl * dd) / sl;
                    sref(S_ZU, v) = zu + ad * (-rcu + zu * dd) / su;
                }
            S64(4);
        }
        // ---------------- outputs ----------------
        wave_lds_fence();
        double* ubuf = rvec;      // N * NT <= NS * 16 doubles
        for (int i = lane; i < N * NT; i += 64) ubuf[i] = 0.0;
        wave_lds_fence();
        for (int v = 0; v < nv; ++v)
            if (tvalid(v)) {
                const double sl = sref(S_SL, v), su = sref(S_SU, v);
                double u = fmin(fmax((sl < su) ? sl : ubl - su, 0.0), ubl);      // (a polished active bound sits at 0 +- rounding)
                if (status == 2) u = ubar_of(v);
                ubuf[(4 * v + lq) * NT + tact] = u;
            }
        wave_lds_fence();
        if (lane < NT) P.out_u0[inst * NT + lane] = ubuf[lane];
        if (P.out_U)
            for (int i = lane; i < N * NT; i += 64) P.out_U[inst * (int64_t)N * NT + i] = ubuf[i];
        if (lane == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        wave_lds_fence();
        S64(5);
#ifdef FTMPC_STAMPS
        if (lane == 0 && inst < 512 && P.dbg_H) {      // (diagnostic build) phase cycles: 0 prologue + start gradient, 1 Riccati sweep, 2 / 3 backward / forward vector sweeps, 4 element-wise, 5 output
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(P.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = s64_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_ric64_kernel<4>(const DeviceConsts, const SolveRicParams);    // N <= 16 (the reference vehicle's horizon)
template __global__ void ftmpc_solve_ric64_kernel<6>(const DeviceConsts, const SolveRicParams);    // N <= 24
template __global__ void ftmpc_solve_ric64_kernel<10>(const DeviceConsts, const SolveRicParams);   // N <= 40 (BASELINE config 5)
template __global__ void ftmpc_solve_ric64_kernel<6, true>(const DeviceConsts, const SolveRicParams);    // with state bounds (N <= 24)
template __global__ void ftmpc_solve_ric64_kernel<10, true>(const DeviceConsts, const SolveRicParams);   // with state bounds (N <= 40)

}  // namespace ftmpc
