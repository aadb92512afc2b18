// ftmpc_solve_ricw.hip -- kernel 13: the reference's OWN formulation (6-D generalized force per stage under the input hull,
// spiraling_mpc.py:133-137,175-177; input_bounds.py:43-76) in float64 by the RICCATI RECURSION, one wave per instance.
//
// Kernel 11 (fp32, one wave) keeps the condensed 6N x 6N Hessian in LDS and its factor in registers: N <= 16 and at most 32 hull
// rows per stage; everything else went to the dense float64 kernel (247 k QP-steps/s at BASELINE's horizon N = 20, and the
// 112-facet hull of a generic 8-thruster vehicle not at all).  In the stage structure the problem has 6 inputs and 13 states per
// stage whatever N is, and the hull rows A tau_k <= b of a stage -- however many -- enter the Newton system as ONE 6 x 6 block
// G_k = A' diag(z / s) A on that stage's input weight:
//     Ruu = 2 R + G_k + B' S B,   Rux = B' S A,   W = chol(Ruu)^-1,   Y = W Rux,   P_k = A' S A - Y' Y,   S = Qt_{k+1} + P_{k+1}
// (kernel 12's backward sweep with a six-pivot potrf), two vector sweeps per right-hand side, O(N) work, nothing condensed.
// The iteration is the float64 kernel's general-constraint Mehrotra iteration (ftmpc_solve_f64.hip MODE 1;
// oracle/qp_oracle.py:ipm_general is the mirror): start at the hull centre, duals mu0 / s, rows strictly feasible throughout;
// then the same ACTIVE-SET POLISH (polish_general): penalty W = 1e6 hs / |a|^2 on the rows with z > s, two multiplier steps per
// round, signs verified.  The gradient is never condensed either: its input part c_k + 2 R d_k is element-wise, its state part
// 2 W e_j + Qt_j dx_j enters the sweeps as the state-linear term, with the state deviations dx_j of the iterate carried along.
// hs = max diag(H) comes from one open-loop sweep S_j = Qt_j + A_j' S_{j+1} A_j (diag(2 R + B' S B)).
#include <hip/hip_runtime.h>

#include "ftmpc_common.h"

namespace ftmpc {

namespace rickw {
using namespace rick;
constexpr int MHMAX = 128;       // hull rows per stage
constexpr int NVAR = 3;          // per wrench variable: d | c | d of the interior-point iterate (kept while the polish runs)
constexpr int NROW = 5;          // per hull row: s | z | ds_a | dz_a | active
__host__ __device__ constexpr int mhs_of(int MH) { return (MH + 15) / 16; }
__host__ __device__ constexpr int64_t var_off(int N) { return (int64_t)N * 2 * 256; }
__host__ __device__ constexpr int64_t row_off(int N) { return var_off(N) + (int64_t)NVAR * ((N + 3) / 4) * 64; }
__host__ __device__ constexpr int64_t xdev_off(int N, int MH) { return row_off(N) + (int64_t)NROW * ((N + 3) / 4) * mhs_of(MH) * 64; }
// terminal-set rows (row i = lane + 64 j, j < 2): s | z | carried primal residual | ds_a | dz_a | active | penalty
constexpr int NTROW = 8;          // (+ the carried residual of the iterate kept while the early polish runs)
__host__ __device__ constexpr int64_t trow_off(int N, int MH) { return xdev_off(N, MH) + (int64_t)N * 16; }
__host__ __device__ constexpr int64_t xdev0_off(int N, int MH) { return trow_off(N, MH) + (int64_t)NTROW * 2 * 64; }      // dx of that iterate
__host__ __device__ constexpr int64_t slot_doubles(int N, int MH) { return xdev0_off(N, MH) + (int64_t)N * 16; }
}  // namespace rickw

struct SolveRicwParams {
    SolveParams base;          // rec (double), ub, stuck, status, iters, qhead (shared instance cursor or nullptr)
    double* slot;              // [gridDim.x][slot_doubles]
    int64_t slot_doubles;
    const double* warmG;       // [B*N*6] previous wrench solution (already shifted) or nullptr: linearise about D stuck
    const double* hullA;       // [n_sets][hull_rows*6] facet normals
    const int32_t* hull_set;   // [B] table number or nullptr (table 0)
    const double* hullb;       // [B*hull_rows] facet offsets
    int32_t hull_rows;         // <= rickw::MHMAX
    double* out_tau0;          // [B*6]
    double* out_G;             // [B*N*6] or nullptr
    // terminal set (template TS): rows term_A (e_N + dx_N) <= term_b on the terminal tracking error (spiraling_mpc.py:199-202)
    const double* termA;       // [term_rows*9]
    const double* termb;       // [term_rows]
    const double* eN;          // [B*9] terminal tracking error at the linearisation point (ftmpc_linearize.hip)
    int32_t term_rows;         // <= 80
};

// TS: with the terminal set.  Its rows act on the terminal state only: a 9 x 9 term A_T' diag(z / s) A_T on the terminal weight S_N
// and a 9-vector on the terminal state-linear term -- where the condensed kernels carry a rank-9 dense update of every tile.
template <int NV, bool TS = false>      // N <= 4 NV
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FTMPC_RIC_WAVES, FTMPC_RIC_WAVES))) ftmpc_solve_ricw64_kernel(const DeviceConsts C, const SolveRicwParams Q) {
    using namespace rickw;
    constexpr int NS = 4 * NV;
    constexpr int MTP = 80;
    __shared__ double s_tA[TS ? MTP * 9 : 2];                             // rows of term_A
    __shared__ __attribute__((aligned(32))) double m9[TS ? 81 : 2];      // A_T' W A_T
    __shared__ __attribute__((aligned(32))) double qT[TS ? 16 : 4];      // terminal state-linear term of the rows (position 4 q + rr = element q + 4 rr)
    __shared__ double x9[TS ? 81 : 2];                                    // (GN GN')[0:9, 0:9]: the norms |A_T,i GN|^2 of the polish's penalties
    const SolveParams& P = Q.base;
    __shared__ __attribute__((aligned(32))) double recbuf[2][REC_STRIDE];
    __shared__ __attribute__((aligned(32))) double vecs[2 * NS * 16];
    double* const rvec = vecs;                 // per stage 16 doubles: right-hand side in, solution out (natural order; 6 used)
    double* const wst = vecs + NS * 16;        // per stage a row-layout vector in the sweeps; afterwards x_{k+1} of the solution, natural order
    double* const gblk = vecs;                 // (between the weight pass and the Riccati sweep) per stage the 21 entries of G_k at 24 k
    __shared__ __attribute__((aligned(32))) double tsc[16 * 17];
    __shared__ __attribute__((aligned(32))) double pcs[32];
    __shared__ __attribute__((aligned(32))) double vsc[2][16];
    __shared__ double s_hA[MHMAX * 6];
    __shared__ double s_ctr[12];               // hull centre D (ub / 2 + stuck) | D stuck

    const int N = C.N, NT = C.NT;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
    double* const slot = Q.slot + (int64_t)blockIdx.x * Q.slot_doubles;
    const int MH = Q.hull_rows, MHS = mhs_of(MH);

    // ---- per-lane structure of A (13 x 13) and B (13 x 6) in the record: element (row q + 4 rr, col li) ----
    const int lane0 = threadIdx.x;
    int aoff[4], boff[4], gidx[4];      // record offsets (or -1); index of G_k's entry (row, col) in its 21-entry block (or -1)
    double acst[4];
    {
        const int li = lane0 & 15, lq = lane0 >> 4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = lq + 4 * rr, c = li;
            int off = -1;
            double cst = 0.0;
            if (r < 3) {
                if (c < 3) cst = (c == r) ? 1.0 : 0.0;
                else if (c < 6) cst = (c - 3 == r) ? C.dt : 0.0;
                else if (c < 9) off = REC_APW + 3 * r + (c - 6);
                else if (c < 13) off = REC_APQ + 4 * r + (c - 9);
            } else if (r < 6) {
                if (c >= 3 && c < 6) cst = (c == r) ? 1.0 : 0.0;
                else if (c >= 6 && c < 9) off = REC_AVW + 3 * (r - 3) + (c - 6);
                else if (c >= 9 && c < 13) off = REC_AVQ + 4 * (r - 3) + (c - 9);
            } else if (r < 9) {
                if (c >= 6 && c < 9) off = REC_AWW + 3 * (r - 6) + (c - 6);
            } else if (r < 13) {
                if (c >= 6 && c < 9) off = REC_AQW + 3 * (r - 9) + (c - 6);
                else if (c >= 9 && c < 13) off = REC_AQQ + 4 * (r - 9) + (c - 9);
            }
            aoff[rr] = off;
            acst[rr] = cst;
            int bo = -1;      // B[row][col]: force part for col < 3, torque part for col 3..5
            if (c < 3) bo = (r < 3) ? REC_BPF + 3 * r + c : (r < 6 ? REC_BVF + 3 * (r - 3) + c : -1);
            else if (c < 6) bo = (r < 3) ? REC_BPT + 3 * r + (c - 3) : (r < 6 ? REC_BVT + 3 * (r - 3) + (c - 3) : (r < 9 ? REC_BWT + 3 * (r - 6) + (c - 3) : (r < 13 ? REC_BQT + 3 * (r - 9) + (c - 3) : -1)));
            boff[rr] = bo;
            const int g = r > c ? r : c, hh = r > c ? c : r;
            gidx[rr] = (r < 6 && c < 6) ? g * (g + 1) / 2 + hh : -1;
        }
    }

    auto pull = [&]() -> int64_t {
        int i = 0;
        if (lane0 == 0) i = atomicAdd(P.qhead, 1);
        return (int64_t)__builtin_amdgcn_readfirstlane(i);
    };
    // instances: the whole batch, or (P.qlist != nullptr) the *P.qcount entries of the list kernel 11 wrote (what it does not certify)
    const int64_t n_inst = P.qlist ? (int64_t)*P.qcount : P.B;
    int64_t qi = P.qhead ? pull() : (int64_t)blockIdx.x;
    for (; qi < n_inst; qi = P.qhead ? pull() : qi + gridDim.x) {
        const int64_t inst = P.qlist ? (int64_t)P.qlist[qi] : qi;
        wave_lds_fence();
        const int lane = lane_now();
        const int li = lane & 15, lq = lane >> 4;
        if (N > NS || MH < 1 || MH > MHMAX) {      // (the host does not send such shapes)
            if (lane == 0) {
                if (P.status) P.status[inst] = 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        // ---------------- prologue ----------------
        if (lane < 12) {      // hull centre (the start point: strictly inside every row) and the thrusters-off wrench (cold linearisation point)
            const int g = lane % 6;
            double acc = 0.0;
            for (int i = 0; i < NT; ++i) acc += C.D[g * MAX_NT + i] * ((lane < 6 ? 0.5 * P.ub[inst * NT + i] : 0.0) + P.stuck[inst * NT + i]);
            s_ctr[lane] = acc;
        }
        {
            const int64_t set = Q.hull_set ? Q.hull_set[inst] : 0;
            for (int i = lane; i < MH * 6; i += 64) s_hA[i] = Q.hullA[set * MH * 6 + i];
        }
        wave_lds_fence();
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        const int nv = (N + 3) >> 2;
        const bool wcomp = li < 6;                     // this lane's wrench variables: (stage 4 v + lq, component li), v < nv
        auto wvalid = [&](int v) { return wcomp && 4 * v + lq < N; };
        auto tbar_of = [&](int v) -> double { return wvalid(v) ? (Q.warmG ? Q.warmG[(inst * N + 4 * v + lq) * 6 + li] : s_ctr[6 + li]) : 0.0; };
        const double r2 = wcomp ? 2.0 * C.R[li < 6 ? li : 0] : 0.0;
        double* const vst = slot + var_off(N);
        double* const rst = slot + row_off(N);
        double* const xdev = slot + xdev_off(N, MH);      // dx_{k+1} of the iterate at 16 k, natural order
        auto vref = [&](int arr, int v) -> double& { return vst[(int64_t)(arr * nv + v) * 64 + lane]; };
        auto rref = [&](int arr, int v, int c) -> double& { return rst[(int64_t)((arr * nv + v) * MHS + c) * 64 + lane]; };
        enum { V_D = 0, V_CL = 1, V_D0 = 2, R_S = 0, R_Z = 1, R_DSA = 2, R_DZA = 3, R_ACT = 4 };
        auto rvalid = [&](int v, int c) { return 4 * v + lq < N && 16 * c + li < MH; };
        const int MT = TS ? Q.term_rows : 0;
        double* const tst = slot + trow_off(N, MH);
        auto tref = [&](int arr, int j) -> double& { return tst[(int64_t)(arr * 2 + j) * 64 + lane]; };
        enum { T_S = 0, T_Z = 1, T_RP = 2, T_DSA = 3, T_DZA = 4, T_ACT = 5, T_W = 6, T_RP0 = 7 };
        auto tvalid = [&](int j) { return TS && lane + 64 * j < MT; };
        if constexpr (TS) {
            for (int i = lane; i < MTP * 9; i += 64) s_tA[i] = (i < MT * 9) ? Q.termA[i] : 0.0;
            if (lane < 16) qT[lane] = 0.0;
            wave_lds_fence();
        }
        // A_T,i . x for a 9-vector in LDS (natural order)
        auto term_dot = [&](int j, const double* x) -> double {
            const double* a = s_tA + 9 * (tvalid(j) ? lane + 64 * j : 0);
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) t += a[r] * x[r];
            return t;
        };
        // m9 = A_T' diag(w) A_T for per-row weights, qT = -A_T' t for per-row values (the LQ convention: + q'x)
        auto term_blocks = [&](auto wof) {
            if constexpr (TS) {
                double acc[45];
#pragma unroll
                for (int p = 0; p < 45; ++p) acc[p] = 0.0;
                for (int j = 0; j < 2; ++j) {
                    const double w = tvalid(j) ? wof(j) : 0.0;
                    const double* a = s_tA + 9 * (tvalid(j) ? lane + 64 * j : 0);
                    int p = 0;
#pragma unroll
                    for (int r1 = 0; r1 < 9; ++r1) {
                        const double wa = w * a[r1];
#pragma unroll
                        for (int r2 = 0; r2 <= r1; ++r2) acc[p++] += wa * a[r2];
                    }
                }
                wave_lds_fence();
                int p = 0;
#pragma unroll
                for (int r1 = 0; r1 < 9; ++r1)
#pragma unroll
                    for (int r2 = 0; r2 <= r1; ++r2) {
                        const double t = wave_red<DAdd>(acc[p++]);
                        if (lane == 0) {
                            m9[9 * r1 + r2] = t;
                            m9[9 * r2 + r1] = t;
                        }
                    }
                wave_lds_fence();
            }
        };
        auto term_linear = [&](auto tof) {
            if constexpr (TS) {
                double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int j = 0; j < 2; ++j) {
                    const double t = tvalid(j) ? tof(j) : 0.0;
                    const double* a = s_tA + 9 * (tvalid(j) ? lane + 64 * j : 0);
#pragma unroll
                    for (int r = 0; r < 9; ++r) acc[r] += a[r] * t;
                }
                wave_lds_fence();
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const double t = wave_red<DAdd>(acc[r]);
                    if (lane == 0) qT[v64pos(r)] = -t;
                }
                wave_lds_fence();
            }
        };
        // Rt = 2 R on the six wrench components, identity on the padding
        f64x4 Rt;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = lq + 4 * rr;
            Rt[rr] = (r == li) ? (r < 6 ? 2.0 * C.R[r < 6 ? r : 0] : 1.0) : 0.0;
        }

        struct Pre {
            f64x4 W, Y;
            double r0, r1, r2, xd;
        };
        auto request = [&](Pre& p, int k, bool tiles) {
            const double* r = recg + (int64_t)k * REC_STRIDE;
            p.r0 = r[lane];
            p.r1 = r[64 + lane];
            p.r2 = (128 + lane < REC_STRIDE) ? r[128 + lane] : 0.0;
            if (tiles) {
                p.W = *reinterpret_cast<const f64x4*>(slot + (int64_t)(2 * k) * 256 + 4 * lane);
                p.Y = *reinterpret_cast<const f64x4*>(slot + (int64_t)(2 * k + 1) * 256 + 4 * lane);
                p.xd = xdev[k * 16 + li];
            }
        };
        auto commit = [&](const Pre& p, int buf) {
            wave_lds_fence();
            recbuf[buf][lane] = p.r0;
            recbuf[buf][64 + lane] = p.r1;
            if (128 + lane < REC_STRIDE) recbuf[buf][128 + lane] = p.r2;
            wave_lds_fence();
        };
        auto stage_tiles = [&](const double* rb, f64x4& A, f64x4& Bt) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                A[rr] = (aoff[rr] >= 0) ? rb[aoff[rr]] : acst[rr];
                Bt[rr] = (boff[rr] >= 0) ? rb[boff[rr]] : 0.0;
            }
        };
        auto weight_tile = [&](bool terminal) -> f64x4 {
            f64x4 w = zero4;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = lq + 4 * rr;
                if (r < 9 && li < 9) w[rr] = terminal ? 2.0 * C.P[9 * r + li] : ((r == li) ? 2.0 * C.Q[r] : 0.0);
            }
            return w;
        };
        auto col2row = [&](double xc, int slotv) -> f64x4 {
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

        // ---- the Newton problem  min 1/2 x'S x + q'x + 1/2 u'Ruu u - r'u  over the stored factors: r in rvec on entry; the state-linear
        // term q_{k+1} = 2 W e_{k+1} + Qt_{k+1} dx_{k+1} is formed here from the records and the iterate's state deviations (xdev);
        // on exit the minimiser u in rvec and its states x_{k+1} in wst[k] (natural order) ----
        auto ric_solve = [&]() {
            Pre p0, p1;
            f64x4 s = zero4;
            auto bstage = [&](int k, Pre& p) {
                commit(p, k & 1);
                const f64x4 Wk = p.W, Yk = p.Y;
                const double xd = p.xd;
                if (k >= 2) request(p, k - 2, true);
                f64x4 A, Bt;
                const double* rb = recbuf[k & 1];
                stage_tiles(rb, A, Bt);
                s += mv(weight_tile(k + 1 == N), xd);      // s = p_{k+1} + q_{k+1}
                if constexpr (TS) {
                    if (k + 1 == N) s += *reinterpret_cast<const f64x4*>(&qT[4 * lq]);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (lq + 4 * rr < 9) s[rr] += 2.0 * rb[REC_WE + lq + 4 * rr];
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
                if (lq == 0) wst[k * 16 + li] = xc;
            };
            request(p0, 0, true);
            if (N >= 2) request(p1, 1, true);
            for (int k = 0; k < N; k += 2) {
                fstage(k, p0);
                if (k + 1 < N) fstage(k + 1, p1);
            }
            wave_lds_fence();
        };
        // ---- backward Riccati sweep with the stage blocks G_k in gblk: the factors W_k, Y_k of every stage -> global slot ----
        auto ric_factor = [&]() -> bool {
            bool ok = true;
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
                if constexpr (TS) {
                    if (k + 1 == N) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            if (lq + 4 * rr < 9 && li < 9) S[rr] += m9[9 * (lq + 4 * rr) + li];
                    }
                }
                const f64x4 SA = hullk::mm_tn64(S, A, zero4), SBt = hullk::mm_tn64(S, Bt, zero4);
                f64x4 Ruu = hullk::mm_tn64(Bt, SBt, Rt);
                const f64x4 Rux = hullk::mm_tn64(Bt, SA, zero4);
                const f64x4 PA = hullk::mm_tn64(A, SA, zero4);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (gidx[rr] >= 0) Ruu[rr] += gblk[k * 24 + gidx[rr]];
                double c[4] = {Ruu.x, Ruu.y, Ruu.z, Ruu.w}, w[4], l[4];
                ok = f64k::potrf_inv16_lds(c, pcs, pcs + 16, lq, li, w, l, 6) && ok;
                const f64x4 Wk = {w[0], w[1], w[2], w[3]};
                wave_lds_fence();
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tsc[(lq + 4 * rr) * 17 + li] = w[rr];
                wave_lds_fence();
                f64x4 Wt;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Wt[rr] = tsc[li * 17 + lq + 4 * rr];
                wave_lds_fence();
                const f64x4 Yk = hullk::mm_tn64(Wt, Rux, zero4);
                Pm = PA - hullk::mm_tn64(Yk, Yk, zero4);
                *reinterpret_cast<f64x4*>(slot + (int64_t)(2 * k) * 256 + 4 * lane) = Wk;
                *reinterpret_cast<f64x4*>(slot + (int64_t)(2 * k + 1) * 256 + 4 * lane) = Yk;
            };
            for (int k = N - 1; k >= 0; k -= 2) {
                fac(k, p0);
                if (k >= 1) fac(k - 1, p1);
            }
            wave_global_fence();
            return __all(ok);
        };
        // ---- stage blocks G_k = sum_r w_r a_r a_r' -> gblk, for row weights given by `wof(v, c)` ----
        auto form_blocks = [&](auto wof) {
            wave_lds_fence();
            for (int v = 0; v < nv; ++v) {
                double acc[21];
#pragma unroll
                for (int p = 0; p < 21; ++p) acc[p] = 0.0;
                for (int c = 0; c < MHS; ++c) {
                    const double w = rvalid(v, c) ? wof(v, c) : 0.0;
                    const double* a = s_hA + 6 * ((16 * c + li < MH) ? 16 * c + li : 0);
                    int p = 0;
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        const double wg = w * a[g];
#pragma unroll
                        for (int hh = 0; hh <= g; ++hh) acc[p++] += wg * a[hh];
                    }
                }
#pragma unroll
                for (int p = 0; p < 21; ++p) acc[p] = row_red16<DAdd>(acc[p]);
                if (li == 0 && 4 * v + lq < NS) {
#pragma unroll
                    for (int p = 0; p < 21; ++p) gblk[(4 * v + lq) * 24 + p] = acc[p];
                }
            }
            wave_lds_fence();
        };
        // ---- C x for the rows of slot (v, c): a_r . x_k, x_k = the six leading entries of rvec at stage k ----
        auto row_dot = [&](int v, int c) -> double {
            const double* a = s_hA + 6 * ((16 * c + li < MH) ? 16 * c + li : 0);
            const double* x = rvec + (4 * v + lq) * 16;
            return (a[0] * x[0] + a[1] * x[1]) + (a[2] * x[2] + a[3] * x[3]) + (a[4] * x[4] + a[5] * x[5]);
        };
        // ---- rvec_k[g] = base(v) + (C' t)_k[g] for per-row values t = tof(v, c) (zero for absent rows); base per wrench variable ----
        auto rhs_with_rows = [&](auto base, auto tof) {
            wave_lds_fence();
            for (int v = 0; v < nv; ++v) {
                double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                for (int c = 0; c < MHS; ++c) {
                    const double t = rvalid(v, c) ? tof(v, c) : 0.0;
                    const double* a = s_hA + 6 * ((16 * c + li < MH) ? 16 * c + li : 0);
#pragma unroll
                    for (int g = 0; g < 6; ++g) acc[g] += a[g] * t;
                }
#pragma unroll
                for (int g = 0; g < 6; ++g) acc[g] = row_red16<DAdd>(acc[g]);
                double tg = acc[0];
#pragma unroll
                for (int g = 1; g < 6; ++g) tg = (li == g) ? acc[g] : tg;
                rvec[(4 * v + lq) * 16 + li] = wvalid(v) ? base(v) + tg : 0.0;      // (this lane's own slot: nobody else reads it in here)
            }
            wave_lds_fence();
        };

        // ---------------- start point (hull centre), its state deviations and the condensed gradient's norm ----------------
        wave_lds_fence();
        for (int v = 0; v < nv; ++v) {
            const double d0 = wvalid(v) ? s_ctr[li] - tbar_of(v) : 0.0;
            vref(V_D, v) = d0;
            if (4 * v + lq < NS) rvec[(4 * v + lq) * 16 + li] = d0;
        }
        wave_lds_fence();
        double gm = 0.0, hs = 0.0;
        {
            Pre p0, p1;
            double xc = 0.0;
            f64x4 Xg = zero4;      // (TS) GN GN' by X_{k+1} = A X_k A' + B B'
            auto transpose = [&](const f64x4& M) -> f64x4 {
                wave_lds_fence();
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tsc[(lq + 4 * rr) * 17 + li] = M[rr];
                wave_lds_fence();
                f64x4 T;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) T[rr] = tsc[li * 17 + lq + 4 * rr];
                wave_lds_fence();
                return T;
            };
            auto gf = [&](int k, Pre& p) {      // dx_{k+1} = A dx_k + B d_k  -> xdev, wst (natural order)
                commit(p, k & 1);
                if (k + 2 < N) request(p, k + 2, false);
                f64x4 A, Bt;
                stage_tiles(recbuf[k & 1], A, Bt);
                if constexpr (TS) {
                    const f64x4 At = transpose(A), Btr = transpose(Bt);
                    const f64x4 T1 = hullk::mm_tn64(Xg, At, zero4);          // X A'   (X symmetric)
                    Xg = hullk::mm_tn64(At, T1, hullk::mm_tn64(Btr, Btr, zero4));      // A X A' + B B'
                }
                const f64x4 xn = mv2(A, xc, Bt, rvec[k * 16 + li]);
                xc = row2col(xn, k & 1);
                if (lq == 0) {
                    wst[k * 16 + li] = xc;
                    xdev[k * 16 + li] = xc;
                }
            };
            request(p0, 0, false);
            if (N >= 2) request(p1, 1, false);
            for (int k = 0; k < N; k += 2) {
                gf(k, p0);
                if (k + 1 < N) gf(k + 1, p1);
            }
            if constexpr (TS) {
                wave_lds_fence();
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (lq + 4 * rr < 9 && li < 9) x9[9 * (lq + 4 * rr) + li] = Xg[rr];
                wave_lds_fence();
            }
            // backward: lam_{k+1} = Qt dx_{k+1} + 2 W e_{k+1} + A_{k+1}' lam_{k+2};  g_k = B' lam_{k+1} + 2 R d_k + c_k;  the open-loop
            // weight S_j = Qt_j + A_j' S_{j+1} A_j alongside: diag(2 R + B' S B) is the diagonal of the condensed Hessian
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
                for (int rr = 0; rr < 4; ++rr)
                    if (lq + 4 * rr < 9) lam[rr] += 2.0 * rb[REC_WE + lq + 4 * rr];
                const double cl = wcomp ? 2.0 * rb[REC_RUT + (li < 6 ? li : 0)] : 0.0;      // c_k = 2 R (tbar_k - ur_k - [f_virt; 0])
                const double gk = quad_red<DAdd>(mvt_part(Bt, lam)) + r2 * rvec[k * 16 + li] + cl;
                if (wcomp) gm = fmax(gm, fabs(gk));
                const double muc = quad_red<DAdd>(mvt_part(A, lam));
                mu = col2row(muc, k & 1);
                const f64x4 S = So + Wt;
                const f64x4 SBt = hullk::mm_tn64(S, Bt, zero4), SA = hullk::mm_tn64(S, A, zero4);
                const f64x4 Hd = hullk::mm_tn64(Bt, SBt, Rt);
                So = hullk::mm_tn64(A, SA, zero4);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (lq + 4 * rr == li && li < 6) hs = fmax(hs, Hd[rr]);
                wave_lds_fence();
                if (lq == 0) rvec[k * 16 + li] = cl;      // (d_k has been consumed: the constant c_k for the pass below)
            };
            request(p0, N - 1, false);
            if (N >= 2) request(p1, N - 2, false);
            for (int k = N - 1; k >= 0; k -= 2) {
                gb(k, p0);
                if (k >= 1) gb(k - 1, p1);
            }
            wave_lds_fence();
        }
        gm = wave_red<DMax>(gm);
        hs = wave_red<DMax>(hs);
        for (int v = 0; v < nv; ++v) vref(V_CL, v) = wvalid(v) ? rvec[(4 * v + lq) * 16 + li] : 0.0;
        // rows: slack b - A centre (the same for every stage), duals on the central path
        double mrows = 0.0;
        {
            double smax = 0.0;
            for (int c = 0; c < MHS; ++c) {
                const int r = 16 * c + li;
                double s0 = 1.0;
                if (r < MH) {
                    s0 = Q.hullb[inst * MH + r];
#pragma unroll
                    for (int g = 0; g < 6; ++g) s0 -= s_hA[6 * r + g] * s_ctr[g];
                    smax = fmax(smax, s0);
                }
                for (int v = 0; v < nv; ++v) rref(R_S, v, c) = s0;
            }
            if constexpr (TS) {      // terminal rows: residual at the start point, slack max(residual, 0.1) and the primal residual the steps shrink
                for (int j = 0; j < 2; ++j) {
                    double st = 1.0, rp = 0.0;
                    if (tvalid(j)) {
                        const int i = lane + 64 * j;
                        double res = Q.termb[i] - term_dot(j, wst + (N - 1) * 16);
#pragma unroll
                        for (int r = 0; r < 9; ++r) res -= s_tA[9 * i + r] * Q.eN[inst * 9 + r];
                        st = fmax(res, 0.1);
                        rp = st - res;
                        smax = fmax(smax, st);
                        mrows += 1.0;
                    }
                    tref(T_S, j) = st;
                    tref(T_RP, j) = rp;
                }
            }
            smax = wave_red<DMax>(smax);
            const double mu0 = fmax(0.02 * gm * smax, 1e-3);
            for (int v = 0; v < nv; ++v)
                for (int c = 0; c < MHS; ++c) {
                    const bool ok = rvalid(v, c);
                    rref(R_Z, v, c) = ok ? mu0 / rref(R_S, v, c) : 0.0;
                    mrows += ok ? 1.0 : 0.0;
                }
            if constexpr (TS)
                for (int j = 0; j < 2; ++j) tref(T_Z, j) = tvalid(j) ? mu0 / tref(T_S, j) : 0.0;
            mrows = wave_red<DAdd>(mrows);
        }
        const double inv_m = 1.0 / mrows;
        wave_global_fence();

        // ---------------- interior-point iterations over the hull rows ----------------
        // The iteration is LEFT EARLY, at mu 1e-7, for the active-set polish below (as kernels 11 and 12 do): verified there, the
        // instance is done with the exact solution on its active set and the last interior-point passes are saved; not verified, the
        // iterate kept aside (rows in the predictor's arrays, which the polish does not use) is taken up again and run to mu_stop,
        // where the polish is tried once more.
        int status = 1, nit = 0, it = 0;
        bool verified = false;
        double mu_target = fmax(C.mu_stop, 1e-7);
        double* const xdev0 = slot + xdev0_off(N, MH);
        for (int phase = 0; phase < 2; ++phase) {
        for (; it <= C.max_iters; ++it) {
            double csum = 0.0;
            for (int v = 0; v < nv; ++v)
                for (int c = 0; c < MHS; ++c)
                    if (rvalid(v, c)) csum += rref(R_S, v, c) * rref(R_Z, v, c);
            double rpn = 0.0;
            if constexpr (TS) {
                for (int j = 0; j < 2; ++j)
                    if (tvalid(j)) {
                        csum += tref(T_S, j) * tref(T_Z, j);
                        rpn = fmax(rpn, fabs(tref(T_RP, j)));
                    }
                rpn = wave_red<DMax>(rpn);
            }
            const double mu = wave_red<DAdd>(csum) * inv_m;
            if (!(mu == mu) || !(rpn == rpn)) {
                status = 2;
                break;
            }
            if (mu < mu_target && rpn < 1e-9) {
                status = 0;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            form_blocks([&](int v, int c) { return rref(R_Z, v, c) / rref(R_S, v, c); });
            term_blocks([&](int j) { return tref(T_Z, j) / tref(T_S, j); });
            if (__builtin_amdgcn_readfirstlane(!ric_factor())) {
                status = (mu < 1e-7 && rpn < 1e-9) ? 0 : 2;      // (as the float64 kernel: a breakdown this close to the solution ends the iteration as converged)
                --nit;
                break;
            }
            // predictor: the hull rows are strictly feasible and on s z = rc: no row term; the terminal rows carry t = -z rp / s
            term_linear([&](int j) { return -tref(T_Z, j) * tref(T_RP, j) / tref(T_S, j); });
            wave_lds_fence();
            for (int v = 0; v < nv; ++v)
                if (4 * v + lq < NS) rvec[(4 * v + lq) * 16 + li] = wvalid(v) ? -(vref(V_CL, v) + r2 * vref(V_D, v)) : 0.0;
            wave_lds_fence();
            ric_solve();
            double ap = 1.0, ad = 1.0;
            if constexpr (TS) {
                for (int j = 0; j < 2; ++j) {
                    double ds = 0.0, dz = 0.0;
                    if (tvalid(j)) {
                        const double s_ = tref(T_S, j), z = tref(T_Z, j);
                        ds = -tref(T_RP, j) - term_dot(j, wst + (N - 1) * 16);
                        dz = -z - z * ds / s_;
                        if (ds < 0.0) ap = fmin(ap, -s_ / ds);
                        if (dz < 0.0) ad = fmin(ad, -z / dz);
                    }
                    tref(T_DSA, j) = ds;
                    tref(T_DZA, j) = dz;
                }
            }
            for (int v = 0; v < nv; ++v)
                for (int c = 0; c < MHS; ++c) {
                    double ds = 0.0, dz = 0.0;
                    if (rvalid(v, c)) {
                        const double s = rref(R_S, v, c), z = rref(R_Z, v, c);
                        ds = -row_dot(v, c);
                        dz = -z - z * ds / s;
                        if (ds < 0.0) ap = fmin(ap, -s / ds);
                        if (dz < 0.0) ad = fmin(ad, -z / dz);
                    }
                    rref(R_DSA, v, c) = ds;
                    rref(R_DZA, v, c) = dz;
                }
            ap = wave_red<DMin>(ap);
            ad = wave_red<DMin>(ad);
            csum = 0.0;
            for (int v = 0; v < nv; ++v)
                for (int c = 0; c < MHS; ++c)
                    if (rvalid(v, c)) csum += (rref(R_S, v, c) + ap * rref(R_DSA, v, c)) * (rref(R_Z, v, c) + ad * rref(R_DZA, v, c));
            if constexpr (TS)
                for (int j = 0; j < 2; ++j)
                    if (tvalid(j)) csum += (tref(T_S, j) + ap * tref(T_DSA, j)) * (tref(T_Z, j) + ad * tref(T_DZA, j));
            const double mu_aff = wave_red<DAdd>(csum) * inv_m;
            double sigma = mu_aff / mu;
            sigma = fmin(fmax(sigma * sigma * sigma, 0.0), 1.0);
            const double sm = sigma * mu;
            // corrector: rc = s z + ds_a dz_a - sigma mu,  t = -z + rc / s = (ds_a dz_a - sigma mu) / s  (terminal rows: - z rp / s more)
            term_linear([&](int j) { return (tref(T_DSA, j) * tref(T_DZA, j) - sm - tref(T_Z, j) * tref(T_RP, j)) / tref(T_S, j); });
            rhs_with_rows([&](int v) { return -(vref(V_CL, v) + r2 * vref(V_D, v)); },
                          [&](int v, int c) { return (rref(R_DSA, v, c) * rref(R_DZA, v, c) - sm) / rref(R_S, v, c); });
            ric_solve();
            ap = 1e300;
            ad = 1e300;
            for (int v = 0; v < nv; ++v)
                for (int c = 0; c < MHS; ++c)
                    if (rvalid(v, c)) {
                        const double s = rref(R_S, v, c), z = rref(R_Z, v, c);
                        const double ds = -row_dot(v, c);
                        const double rc = s * z + rref(R_DSA, v, c) * rref(R_DZA, v, c) - sm;
                        const double dz = (-rc - z * ds) / s;
                        if (ds < 0.0) ap = fmin(ap, -s / ds);
                        if (dz < 0.0) ad = fmin(ad, -z / dz);
                    }
            double tds[2] = {0.0, 0.0}, tdz[2] = {0.0, 0.0};
            if constexpr (TS) {
                for (int j = 0; j < 2; ++j)
                    if (tvalid(j)) {
                        const double s_ = tref(T_S, j), z = tref(T_Z, j), rp = tref(T_RP, j);
                        tds[j] = -rp - term_dot(j, wst + (N - 1) * 16);
                        tdz[j] = (-(s_ * z + tref(T_DSA, j) * tref(T_DZA, j) - sm) - z * tds[j]) / s_;
                        if (tds[j] < 0.0) ap = fmin(ap, -s_ / tds[j]);
                        if (tdz[j] < 0.0) ad = fmin(ad, -z / tdz[j]);
                    }
            }
            ap = fmin(1.0, 0.9995 * wave_red<DMin>(ap));
            ad = fmin(1.0, 0.9995 * wave_red<DMin>(ad));
            if constexpr (TS) {
                for (int j = 0; j < 2; ++j)
                    if (tvalid(j)) {
                        tref(T_S, j) += ap * tds[j];
                        tref(T_Z, j) += ad * tdz[j];
                        tref(T_RP, j) *= (1.0 - ap);
                    }
            }
            for (int v = 0; v < nv; ++v) {
                for (int c = 0; c < MHS; ++c)
                    if (rvalid(v, c)) {
                        const double s = rref(R_S, v, c), z = rref(R_Z, v, c);
                        const double ds = -row_dot(v, c);
                        const double rc = s * z + rref(R_DSA, v, c) * rref(R_DZA, v, c) - sm;
                        rref(R_S, v, c) = s + ap * ds;
                        rref(R_Z, v, c) = z + ad * (-rc - z * ds) / s;
                    }
                if (wvalid(v)) vref(V_D, v) += ap * rvec[(4 * v + lq) * 16 + li];
            }
            for (int i = lane; i < N * 16; i += 64) xdev[i] += ap * wst[i];      // the state deviations follow the step
            wave_global_fence();
        }

        // ---------------- active-set polish (oracle/qp_oracle.py:polish_general; ftmpc_solve_f64.hip MODE 1) ----------------
        if (status == 0) {
            constexpr double PW0 = 1e6, PRES_TOL = 1e-10;
            const double pw = PW0 * hs;
            for (int i = lane; i < N * 16; i += 64) xdev0[i] = xdev[i];
            for (int v = 0; v < nv; ++v) {
                vref(V_D0, v) = vref(V_D, v);
                for (int c = 0; c < MHS; ++c) {
                    rref(R_DSA, v, c) = rref(R_S, v, c);
                    rref(R_DZA, v, c) = rref(R_Z, v, c);
                    const bool act = rvalid(v, c) && rref(R_Z, v, c) > rref(R_S, v, c);
                    rref(R_ACT, v, c) = act ? 1.0 : 0.0;
                    if (!act) rref(R_Z, v, c) = 0.0;
                }
            }
            auto wrow = [&](int c) -> double {      // penalty of the rows 16 c + li
                const double* a = s_hA + 6 * ((16 * c + li < MH) ? 16 * c + li : 0);
                const double a2 = (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]) + (a[4] * a[4] + a[5] * a[5]);
                return pw / fmax(a2, 1e-300);
            };
            if constexpr (TS) {      // terminal rows: active set, true slack (the carried residual has closed), penalty pw / |A_T,i GN|^2
                for (int j = 0; j < 2; ++j) {
                    tref(T_DSA, j) = tref(T_S, j);
                    tref(T_DZA, j) = tref(T_Z, j);
                    tref(T_RP0, j) = tref(T_RP, j);
                    const bool act = tvalid(j) && tref(T_Z, j) > tref(T_S, j);
                    tref(T_ACT, j) = act ? 1.0 : 0.0;
                    if (!act) tref(T_Z, j) = 0.0;
                    tref(T_S, j) -= tref(T_RP, j);
                    tref(T_RP, j) = 0.0;
                    double c2 = 0.0;
                    const double* a = s_tA + 9 * (tvalid(j) ? lane + 64 * j : 0);
                    for (int r1 = 0; r1 < 9; ++r1)
                        for (int r2 = 0; r2 < 9; ++r2) c2 += a[r1] * x9[9 * r1 + r2] * a[r2];
                    tref(T_W, j) = pw / fmax(c2, 1e-300);
                }
            }
            for (int rd = 0; rd < 3 && !verified; ++rd) {
                form_blocks([&](int v, int c) { return rref(R_ACT, v, c) != 0.0 ? wrow(c) : 0.0; });
                term_blocks([&](int j) { return tref(T_ACT, j) != 0.0 ? tref(T_W, j) : 0.0; });
                if (__builtin_amdgcn_readfirstlane(!ric_factor())) break;
                ++nit;
                for (int in = 0; in < 2; ++in) {
                    // (H + C_A' W C_A) dd = -grad + C_A' (W s_A - lam);  lam += W (C_A dd - s_A);  s -= C dd;  d += dd;  dx += dx
                    term_linear([&](int j) { return tref(T_ACT, j) != 0.0 ? tref(T_W, j) * tref(T_S, j) - tref(T_Z, j) : 0.0; });
                    rhs_with_rows([&](int v) { return -(vref(V_CL, v) + r2 * vref(V_D, v)); },
                                  [&](int v, int c) { return rref(R_ACT, v, c) != 0.0 ? wrow(c) * rref(R_S, v, c) - rref(R_Z, v, c) : 0.0; });
                    ric_solve();
                    if constexpr (TS) {
                        for (int j = 0; j < 2; ++j)
                            if (tvalid(j)) {
                                const double s_ = tref(T_S, j), ch = term_dot(j, wst + (N - 1) * 16);
                                if (tref(T_ACT, j) != 0.0) tref(T_Z, j) += tref(T_W, j) * (ch - s_);
                                tref(T_S, j) = s_ - ch;
                            }
                    }
                    for (int v = 0; v < nv; ++v) {
                        for (int c = 0; c < MHS; ++c)
                            if (rvalid(v, c)) {
                                const double s = rref(R_S, v, c), ch = row_dot(v, c);
                                if (rref(R_ACT, v, c) != 0.0) rref(R_Z, v, c) += wrow(c) * (ch - s);
                                rref(R_S, v, c) = s - ch;
                            }
                        if (wvalid(v)) vref(V_D, v) += rvec[(4 * v + lq) * 16 + li];
                    }
                    for (int i = lane; i < N * 16; i += 64) xdev[i] += wst[i];
                    wave_global_fence();
                }
                bool changed = false;
                for (int v = 0; v < nv; ++v)
                    for (int c = 0; c < MHS; ++c)
                        if (rvalid(v, c)) {
                            const bool act = rref(R_ACT, v, c) != 0.0;
                            if (act && rref(R_Z, v, c) < 0.0) {
                                rref(R_ACT, v, c) = 0.0;
                                rref(R_Z, v, c) = 0.0;
                                changed = true;
                            } else if (!act && rref(R_S, v, c) < -PRES_TOL) {
                                rref(R_ACT, v, c) = 1.0;
                                changed = true;
                            }
                        }
                if constexpr (TS) {
                    for (int j = 0; j < 2; ++j)
                        if (tvalid(j)) {
                            const bool act = tref(T_ACT, j) != 0.0;
                            if (act && tref(T_Z, j) < 0.0) {
                                tref(T_ACT, j) = 0.0;
                                tref(T_Z, j) = 0.0;
                                changed = true;
                            } else if (!act && tref(T_S, j) < -PRES_TOL) {
                                tref(T_ACT, j) = 1.0;
                                changed = true;
                            }
                        }
                }
                verified = __builtin_amdgcn_readfirstlane(!__any(changed));
            }
        }
        if (status != 0 || verified || !(mu_target > C.mu_stop)) break;
        // the early polish did not settle: back to the interior-point iterate, on to mu_stop
        for (int i = lane; i < N * 16; i += 64) xdev[i] = xdev0[i];
        for (int v = 0; v < nv; ++v) {
            vref(V_D, v) = vref(V_D0, v);
            for (int c = 0; c < MHS; ++c) {
                rref(R_S, v, c) = rref(R_DSA, v, c);
                rref(R_Z, v, c) = rref(R_DZA, v, c);
            }
        }
        if constexpr (TS) {
            for (int j = 0; j < 2; ++j) {
                tref(T_S, j) = tref(T_DSA, j);
                tref(T_Z, j) = tref(T_DZA, j);
                tref(T_RP, j) = tref(T_RP0, j);
            }
        }
        wave_global_fence();
        status = 1;
        mu_target = C.mu_stop;
        }
        // ---------------- outputs ----------------
        wave_lds_fence();
        for (int v = 0; v < nv; ++v) {
            const double dd = (status == 2) ? 0.0 : ((status == 0 && !verified) ? vref(V_D0, v) : vref(V_D, v));
            const double tau = tbar_of(v) + dd;
            if (4 * v + lq < NS) rvec[(4 * v + lq) * 16 + li] = wvalid(v) ? tau : 0.0;
            if (wvalid(v) && Q.out_G) Q.out_G[(inst * N + 4 * v + lq) * 6 + li] = tau;
        }
        wave_lds_fence();
        {
            // tau_0 INSIDE the hull for the allocator: pulled towards the centre by the smallest factor that leaves every facet a
            // relative margin of 1e-9 (the polished solution sits ON its active facets)
            double eps = 0.0;
            for (int c = 0; c < MHS; ++c) {
                const int r = 16 * c + li;
                if (lq == 0 && r < MH) {
                    const double b = Q.hullb[inst * MH + r];
                    double s0 = b, st0 = b;
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        s0 -= s_hA[6 * r + g] * s_ctr[g];
                        st0 -= s_hA[6 * r + g] * rvec[g];
                    }
                    if (st0 < 1e-9 * s0 && s0 > st0) eps = fmax(eps, (1e-9 * s0 - st0) / (s0 - st0));
                }
            }
            eps = fmin(wave_red<DMax>(eps), 1.0);
            if (lane < 6) Q.out_tau0[inst * 6 + lane] = s_ctr[lane] + (1.0 - eps) * (rvec[lane] - s_ctr[lane]);
        }
        if (lane == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        wave_lds_fence();
    }
}

template __global__ void ftmpc_solve_ricw64_kernel<6>(const DeviceConsts, const SolveRicwParams);     // N <= 24 (the reference's horizon 15, BASELINE's 20)
template __global__ void ftmpc_solve_ricw64_kernel<10>(const DeviceConsts, const SolveRicwParams);    // N <= 40
template __global__ void ftmpc_solve_ricw64_kernel<6, true>(const DeviceConsts, const SolveRicwParams);      // + the terminal set
template __global__ void ftmpc_solve_ricw64_kernel<10, true>(const DeviceConsts, const SolveRicwParams);

}  // namespace ftmpc
