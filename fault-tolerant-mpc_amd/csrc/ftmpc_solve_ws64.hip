// ftmpc_solve_ws64.hip -- kernel 9: the float64 sibling of kernel 8.  The box-constrained THRUSTER-space QP of the float64
// kernel (ftmpc_solve_f64.hip), with the Newton systems of the interior-point iteration solved through WRENCH space:
//
//     H = DD' H_w DD + 2 rho I,   DD = blockdiag(D_a) (6N x N na),   H_w = L L' (6N x 6N, factorised ONCE per instance)
//     (H + Sigma) x = r   <=>   x = Dg^-1 ( r - DD' L K^-1 L' DD Dg^-1 r ),   Dg = 2 rho + Sigma,
//     K = I + L' S L,   S = DD Dg^-1 DD'  (block diagonal: one 6 x 6 block per stage)
//
// (the algebra, its conditioning and the alternatives that fail are in ftmpc_solve_ws.hip and DESIGN.md; tests/
// test_wrench_schur_form.py checks the identity against the dense form to 1e-12).  For BASELINE config 5 -- N = 40, 16
// thrusters, "fp64 KKT" -- every iteration factorises a 240 x 240 matrix instead of a 560..640-variable one: 1/13 .. 1/19
// of the factorisation flops, and the assembly of K (two block-triangular products, P' = L' S and X = P' L) is tile work
// without a serial chain.  Same QP, same Mehrotra iteration in the thruster variables (bounds, slacks, duals, step rules,
// gradient by recurrence through the Newton system) as the dense kernel: oracle/ftmpc_oracle.c stays the checker.
//
// One 4-wave workgroup per instance, two workgroups per CU.  Everything n x n lives as 16 x 16 float64 tiles in a
// per-workgroup global slot (row-major "operand layout" f64k::t64off: a tile is the A operand of X Y' as it stands):
//     Ks  H_w during the build, then K - I of every iteration, factorised IN PLACE (diagonal slot: inverse of the block)
//     Lt  L' (tile (I,J) = L_IJ'), the only copy of the factor of H_w: L p reads its tiles by columns
// (P' = L' S exists one block row at a time, in LDS).
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354; oracle/qp_oracle.py:ipm_box is the mirror
// (schur_newton_solver restates this form of the Newton step).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace ws64k {
using f64k::f64x4;
using f64k::ld4;
using f64k::lds_barrier;
using f64k::mfma;
using f64k::quad_sum64;
using f64k::t64idx;
using f64k::t64off;
using f64k::v64pos;
constexpr int WG = 256;
constexpr int NWAVE = 4;
constexpr int NBW = 16;                      // block rows of the wrench-space system: 6 N <= 256
constexpr int NPADW = 16 * NBW;
constexpr int NTL = NBW * (NBW + 1) / 2;
constexpr int NMAXST = NPADW / 6;            // stages
// per-workgroup global slot, in doubles
__host__ __device__ constexpr int64_t off_K() { return 0; }
__host__ __device__ constexpr int64_t off_Lt() { return (int64_t)NTL * 256; }
__host__ __device__ constexpr int64_t off_Ld() { return 2 * (int64_t)NTL * 256; }
__host__ __device__ constexpr int64_t off_E() { return off_Ld() + (int64_t)NBW * 256; }
__host__ __device__ constexpr int64_t slot_doubles(int N) { return off_E() + (int64_t)N * 9 * NPADW; }

}  // namespace ws64k

struct SolveWs64Params {
    SolveParams base;     // rec is double; hscratch / work lists unused
    double* slot;         // [gridDim.x][slot_doubles]
    int64_t slot_doubles;
};

#ifndef FTMPC_WS64_WPC
#define FTMPC_WS64_WPC 2      // resident workgroups per CU (register budget 512 / WPC per lane)
#endif
template <int NVT>   // thruster-space variables per thread: N * na <= 256 NVT
__global__ void __launch_bounds__(ws64k::WG, FTMPC_WS64_WPC) ftmpc_solve_ws64_kernel(const DeviceConsts C, const SolveWs64Params Q) {
    using namespace ws64k;
    constexpr int NTP = WG * NVT;
    const SolveParams& P = Q.base;
    __shared__ double recbuf[REC_STRIDE];
    __shared__ double dv[NPADW];        // diagonal added by the factorisation (0: H_w itself, 1: K = I + X) | start-gradient operand
    __shared__ double xv[NPADW];        // rhs / solution of the wrench-space solves (permuted per 16-block)
    __shared__ double tw[NPADW], yv[NPADW], gwv[NPADW];
    __shared__ double rv[NTP], rdg[NTP];
    __shared__ double part[NWAVE * 16];
    __shared__ __attribute__((aligned(32))) double Sbuf[32];     // pivot column | row of E (f64k::potrf_inv16_lds)
    __shared__ double red[NWAVE];
    __shared__ double s_DaT[6 * MAX_NT];          // the healthy columns of D
    __shared__ double s_DD[21 * MAX_NT];          // D_a[g][a] D_a[h][a] for the 21 pairs g >= h
    __shared__ double Sblk[NMAXST * 36];          // stage blocks S_k = D_a diag(1 / Dg) D_a'
    __shared__ double s_R2[6];
    __shared__ unsigned char s_stg[NPADW], s_thr[NPADW];
    __shared__ unsigned char tIJ[2 * NTL];
    __shared__ int s_act[MAX_NT];
    __shared__ int s_flag;
    // finished tiles of the current block row J of the factor (K < J): see ftmpc_solve_f64.hip
    // (also: block row I of P' = L' S while the tiles X_IJ of that row are formed, NBW tiles)
    __shared__ __attribute__((aligned(32))) double Pj[NBW * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int N = C.N, NT = C.NT;
    const double rho = C.rho;
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};

    double* const slot = Q.slot + (int64_t)blockIdx.x * Q.slot_doubles;
    double* const Ks = slot + off_K();
    double* const Fs = Ks;     // the factorisation runs in place: tile (I, J) of the matrix is read once, by the wave that writes tile (I, J) of the factor
    double* const Lt = slot + off_Lt();
    double* const Ld = slot + off_Ld();
    double* const Eall = slot + off_E();

    for (int t = tid; t < NTL; t += WG) {
        int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (t64idx(I + 1, 0) <= t) ++I;
        while (t64idx(I, 0) > t) --I;
        tIJ[2 * t] = (unsigned char)I;
        tIJ[2 * t + 1] = (unsigned char)(t - t64idx(I, 0));
    }
    if (tid < 6) s_R2[tid] = 2.0 * C.R[tid];

    for (int64_t inst = blockIdx.x; inst < P.B; inst += gridDim.x) {
        __syncthreads();
        S64_DECL;
        S64_START();
        // ---------------- prologue ----------------
        if (tid == 0) {
            int na0 = 0;
            for (int i = 0; i < NT; ++i)
                if (P.ub[inst * NT + i] > 0.0) s_act[na0++] = i;
            s_flag = na0;
        }
        __syncthreads();
        const int nat = s_flag;             // healthy thrusters
        const int nt = N * nat;             // thruster-space variables
        const int n = 6 * N;                // wrench-space variables
        const int nb = (n + 15) >> 4;
        const int npad = nb * 16;
        const int ntl = (nb * (nb + 1)) / 2;
        auto write_flat = [&](int status_code, bool from_ubar, const double (&ub0)[NVT], const bool (&tv)[NVT], const int (&tkk)[NVT], const int (&taa)[NVT]) {
            // u = the linearisation point (clip(warm start)) or zero: FTMPC_STATUS_NUMERIC of include/ftmpc.h
            __syncthreads();
            double* ubuf = Pj;
            for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.0;
            __syncthreads();
            if (from_ubar) {
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    if (tv[v]) ubuf[tkk[v] * NT + s_act[taa[v]]] = ub0[v];
            }
            __syncthreads();
            if (tid < NT) P.out_u0[inst * NT + tid] = ubuf[tid];
            if (P.out_U)
                for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = ubuf[i];
            if (tid == 0) {
                if (P.status) P.status[inst] = status_code;
                if (P.iters) P.iters[inst] = 0;
            }
        };
        bool tvalid[NVT];
        int tk[NVT], ta[NVT];
        double ubar[NVT], ubv[NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            const int e = v * WG + tid;
            tvalid[v] = nat > 0 && e < nt;
            tk[v] = tvalid[v] ? e / nat : 0;
            ta[v] = tvalid[v] ? e - tk[v] * nat : 0;
            ubar[v] = 0.0;
            ubv[v] = 1.0;
        }
        if (nat == 0 || nb > NBW || nt > NTP) {
            write_flat(nat == 0 ? 0 : 2, false, ubar, tvalid, tk, ta);
            continue;
        }
#pragma unroll
        for (int v = 0; v < NVT; ++v)
            if (tvalid[v]) {
                const int t = s_act[ta[v]];
                ubv[v] = P.ub[inst * NT + t];
                if (P.warmU) ubar[v] = fmin(fmax(P.warmU[(inst * N + tk[v]) * NT + t], 0.0), ubv[v]);
            }
        if (tid < 6 * MAX_NT) {
            const int g = tid / MAX_NT, a = tid % MAX_NT;
            s_DaT[tid] = (a < nat) ? C.D[g * MAX_NT + s_act[a]] : 0.0;
        }
        if (tid < npad) {
            const int s = tid / 6;
            s_stg[tid] = (unsigned char)(tid < n ? s : 255);
            s_thr[tid] = (unsigned char)(tid < n ? tid - s * 6 : 255);
        }
        __syncthreads();
        for (int e = tid; e < 21 * MAX_NT; e += WG) {   // products of the rows of D_a, pair p = g (g + 1) / 2 + h
            const int p = e / MAX_NT, a = e % MAX_NT;
            int g = 0;
            while ((g + 1) * (g + 2) / 2 <= p) ++g;
            const int hh = p - g * (g + 1) / 2;
            s_DD[e] = (a < nat) ? s_DaT[g * MAX_NT + a] * s_DaT[hh * MAX_NT + a] : 0.0;
        }
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        // wrench-space role of this thread (condensing): column (stage kcol, component acol)
        const int kcol = (tid < npad) ? s_stg[tid] : 255;
        const int acol = (tid < npad) ? s_thr[tid] : 255;
        S64(0);
        // ---------------- phase 1: condense with D_a = I (columns are stage-wrench components), E panels -> global ----------------
        {
            double G[13];
#pragma unroll
            for (int r = 0; r < 13; ++r) G[r] = 0.0;
            double gacc = 0.0;
            for (int k = 0; k < N; ++k) {
                __syncthreads();
                if (tid < REC_STRIDE) recbuf[tid] = recg[k * REC_STRIDE + tid];
                __syncthreads();
                const double* rb = recbuf;
                const bool terminal = (k + 1 == N);
                if (tid >= npad) continue;
                if (kcol < k) {
                    double p[3], vv[3], w[3], q[4];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        p[a] = G[a] + C.dt * G[3 + a];
                        vv[a] = G[3 + a];
                        w[a] = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            p[a] += rb[REC_APW + 3 * a + c] * G[6 + c];
                            vv[a] += rb[REC_AVW + 3 * a + c] * G[6 + c];
                            w[a] += rb[REC_AWW + 3 * a + c] * G[6 + c];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            p[a] += rb[REC_APQ + 4 * a + c] * G[9 + c];
                            vv[a] += rb[REC_AVQ + 4 * a + c] * G[9 + c];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        q[a] = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) q[a] += rb[REC_AQW + 3 * a + c] * G[6 + c];
#pragma unroll
                        for (int c = 0; c < 4; ++c) q[a] += rb[REC_AQQ + 4 * a + c] * G[9 + c];
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        G[a] = p[a];
                        G[3 + a] = vv[a];
                        G[6 + a] = w[a];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) G[9 + a] = q[a];
                } else if (kcol == k) {
                    double F[3], T[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        F[a] = (acol == a) ? 1.0 : 0.0;
                        T[a] = (acol == 3 + a) ? 1.0 : 0.0;
                    }
                    double gr = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) gr += F[a] * rb[REC_RUT + a] + T[a] * rb[REC_RUT + 3 + a];
                    gacc += gr;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double sp = 0.0, sv = 0.0, sw = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            sp += rb[REC_BPF + 3 * a + c] * F[c] + rb[REC_BPT + 3 * a + c] * T[c];
                            sv += rb[REC_BVF + 3 * a + c] * F[c] + rb[REC_BVT + 3 * a + c] * T[c];
                            sw += rb[REC_BWT + 3 * a + c] * T[c];
                        }
                        G[a] = sp;
                        G[3 + a] = sv;
                        G[6 + a] = sw;
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        double sq = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sq += rb[REC_BQT + 3 * a + c] * T[c];
                        G[9 + a] = sq;
                    }
                }
                double gs = 0.0;
#pragma unroll
                for (int r = 0; r < 9; ++r) gs += G[r] * rb[REC_WE + r];
                gacc += gs;
                double* Ek = Eall + (int64_t)k * 9 * npad;
                if (!terminal) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) Ek[r * npad + tid] = C.sq2Q[r] * G[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int c = r; c < 9; ++c) s += C.LPt[9 * r + c] * G[c];
                        Ek[r * npad + tid] = s;
                    }
                }
            }
            if (tid < npad) gwv[tid] = (kcol != 255) ? 2.0 * gacc : 0.0;   // wrench-space gradient at the linearisation point
        }
        __syncthreads();  // E panels visible to the whole workgroup
        S64(1);
        // ---------------- phase 2: H_w tiles on f64 MFMA (tiles round-robin over the waves) ----------------
        for (int t = wave; t < ntl; t += NWAVE) {
            const int I = tIJ[2 * t], J = tIJ[2 * t + 1];
            f64x4 acc = zero4, acc2 = zero4;
            const int kstart = (16 * I) / 6 < N ? (16 * I) / 6 : N;
            int k = kstart;
            for (; k + 1 < N; k += 2) {
                const double* Ek = Eall + (int64_t)k * 9 * npad;
                const double* En = Ek + 9 * npad;
                double a[3], b[3], c[3], d[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int r = 4 * s + lq;
                    a[s] = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.0;
                    b[s] = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.0;
                    c[s] = (r < 9) ? En[r * npad + 16 * I + li] : 0.0;
                    d[s] = (r < 9) ? En[r * npad + 16 * J + li] : 0.0;
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    acc = mfma(a[s], b[s], acc);
                    acc2 = mfma(c[s], d[s], acc2);
                }
            }
            for (; k < N; ++k) {
                const double* Ek = Eall + (int64_t)k * 9 * npad;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int r = 4 * s + lq;
                    const double a = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.0;
                    const double b = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.0;
                    acc = mfma(a, b, acc);
                }
            }
            acc += acc2;
            // + 2 R on the diagonal of every stage block (the rho term stays in thruster space); unit diagonal on the padding
            const int e2 = 16 * J + li;
            const int s2 = s_stg[e2], a2 = s_thr[e2];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = lq + 4 * rr;      // f64 MFMA C/D layout: row = (lane>>4) + 4*reg
                const int e1 = 16 * I + row;
                const int s1 = s_stg[e1], a1 = s_thr[e1];
                double add = 0.0;
                if (s1 != 255 && s1 == s2 && a1 == a2) add = s_R2[a1];
                if (s1 == 255 && e1 == e2) add = 1.0;
                Ks[(int64_t)t * 256 + t64off(row, li)] = acc[rr] + add;
            }
        }
        __syncthreads();
        S64(2);

        // ---- blocked left-looking Cholesky of (Hsrc + diag(dv)) into Fdst: ftmpc_solve_f64.hip's schedule (wave 0 owns the
        // diagonal tile and its potrf + inverse, the off-diagonal tiles of a column run as one prefetched stream per wave).
        // keep_l: the diagonal blocks of the factor itself go to Ld (the first factorisation: H_w = L L').
        auto factor = [&](const double* Hsrc, double* Fdst, auto KEEP) {
            constexpr bool keep_l = decltype(KEEP)::value;
            if (tid == 0) s_flag = 1;
            for (int J = 0; J < nb; ++J) {
                __syncthreads();
                for (int K = wave; K < J; K += NWAVE)
                    *reinterpret_cast<f64x4*>(Pj + K * 256 + 4 * lane) = ld4(Fdst + (int64_t)t64idx(J, K) * 256 + 16 * li + 4 * lq);
                __syncthreads();
                S64(3);
                const double* rowJ = Pj + 4 * lane;
                if (wave == 0) {   // diagonal tile: both operands are row J (LDS), then potrf + inverse
                    f64x4 acc = zero4, acc2 = zero4;
                    int K = 0;
                    for (; K + 1 < J; K += 2) {
                        const f64x4 a0 = ld4(rowJ + K * 256), a1 = ld4(rowJ + (K + 1) * 256);
                        acc = mfma(a0.x, a0.x, acc); acc2 = mfma(a1.x, a1.x, acc2);
                        acc = mfma(a0.y, a0.y, acc); acc2 = mfma(a1.y, a1.y, acc2);
                        acc = mfma(a0.z, a0.z, acc); acc2 = mfma(a1.z, a1.z, acc2);
                        acc = mfma(a0.w, a0.w, acc); acc2 = mfma(a1.w, a1.w, acc2);
                    }
                    for (; K < J; ++K) {
                        const f64x4 a0 = ld4(rowJ + K * 256);
                        acc = mfma(a0.x, a0.x, acc); acc = mfma(a0.y, a0.y, acc);
                        acc = mfma(a0.z, a0.z, acc); acc = mfma(a0.w, a0.w, acc);
                    }
                    acc += acc2;
                    double* tjj = Fdst + (int64_t)t64idx(J, J) * 256;
                    const double* hjj = Hsrc + (int64_t)t64idx(J, J) * 256;
                    const double sg = dv[16 * J + li];
                    double c[4], w[4], l[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        c[rr] = hjj[t64off(lq + 4 * rr, li)] - acc[rr];
                        if (lq + 4 * rr == li) c[rr] += sg;
                    }
                    const bool ok = f64k::potrf_inv16_lds(c, Sbuf, Sbuf + 16, lq, li, w, l);
                    if (!ok && lane == 0) s_flag = 0;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) tjj[t64off(lq + 4 * rr, li)] = w[rr];
                    if constexpr (keep_l) {
                        double* ldj = Ld + J * 256;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) ldj[t64off(lq + 4 * rr, li)] = l[rr];
                    }
                }
                S64(8);
                auto next_tile = [&](int I) {
                    do ++I; while (I < nb && f64k::col_owner(I - J) != wave);
                    return I;
                };
                auto store_c = [&](int I, const f64x4& acc, const double (&h)[4]) {
                    double* tij = Fdst + (int64_t)t64idx(I, J) * 256;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) tij[t64off(lq + 4 * rr, li)] = h[rr] - acc[rr];
                };
                auto load_h = [&](int I, double (&h)[4]) {
                    const double* hij = Hsrc + (int64_t)t64idx(I, J) * 256;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) h[rr] = hij[t64off(lq + 4 * rr, li)];
                };
                {
                    const int nq = (J + 3) >> 2;
                    int I = next_tile(J), q = 0;
                    f64x4 A0[4], A1[4], acc = zero4, acc2 = zero4;
                    double hreg[4] = {0.0, 0.0, 0.0, 0.0};
                    auto fetch = [&](f64x4 (&A)[4], int It, int qq) {
                        const double* rowI = Fdst + (int64_t)t64idx(It, 0) * 256 + 16 * li + 4 * lq;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int K = 4 * qq + i;
                            A[i] = zero4;
                            if (K < J) A[i] = ld4(rowI + K * 256);
                        }
                    };
                    auto batch = [&](const f64x4 (&A)[4], int qq) {
                        f64x4 b[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int K = 4 * qq + i;
                            b[i] = zero4;
                            if (K < J) b[i] = ld4(rowJ + K * 256);
                        }
                        acc = mfma(A[0].x, b[0].x, acc); acc2 = mfma(A[1].x, b[1].x, acc2);
                        acc = mfma(A[0].y, b[0].y, acc); acc2 = mfma(A[1].y, b[1].y, acc2);
                        acc = mfma(A[0].z, b[0].z, acc); acc2 = mfma(A[1].z, b[1].z, acc2);
                        acc = mfma(A[0].w, b[0].w, acc); acc2 = mfma(A[1].w, b[1].w, acc2);
                        acc = mfma(A[2].x, b[2].x, acc); acc2 = mfma(A[3].x, b[3].x, acc2);
                        acc = mfma(A[2].y, b[2].y, acc); acc2 = mfma(A[3].y, b[3].y, acc2);
                        acc = mfma(A[2].z, b[2].z, acc); acc2 = mfma(A[3].z, b[3].z, acc2);
                        acc = mfma(A[2].w, b[2].w, acc); acc2 = mfma(A[3].w, b[3].w, acc2);
                    };
                    auto step = [&](const f64x4 (&Ac)[4], f64x4 (&An)[4]) -> bool {
                        if (q == 0) {
                            acc = zero4;
                            acc2 = zero4;
                            load_h(I, hreg);
                        }
                        int In = I, qn = q + 1;
                        if (qn >= nq) {
                            In = next_tile(I);
                            qn = 0;
                        }
                        if (In < nb) fetch(An, In, qn);
                        batch(Ac, q);
                        if (qn == 0) {
                            acc += acc2;
                            store_c(I, acc, hreg);
                        }
                        I = In;
                        q = qn;
                        return I >= nb;
                    };
                    if (nq == 0) {   // first column: nothing to subtract
                        for (; I < nb; I = next_tile(I)) {
                            load_h(I, hreg);
                            store_c(I, zero4, hreg);
                        }
                    } else if (I < nb) {
                        fetch(A0, I, 0);
                        for (;;) {
                            if (step(A0, A1)) break;
                            if (step(A1, A0)) break;
                        }
                    }
                }
                S64(10);
                __syncthreads();
                S64(11);
                // L_IJ = C_IJ W_J' for the tiles this wave produced, four at a time
                const f64x4 w4 = ld4(Fdst + (int64_t)t64idx(J, J) * 256 + 16 * li + 4 * lq);
                for (int I = next_tile(J); I < nb;) {
                    int Is[4] = {nb, nb, nb, nb};
                    f64x4 a4[4], x[4];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        a4[c4] = zero4;
                        if (I < nb) {
                            Is[c4] = I;
                            a4[c4] = ld4(Fdst + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                            I = next_tile(I);
                        }
                    }
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        x[c4] = zero4;
                        x[c4] = mfma(a4[c4].x, w4.x, x[c4]); x[c4] = mfma(a4[c4].y, w4.y, x[c4]);
                        x[c4] = mfma(a4[c4].z, w4.z, x[c4]); x[c4] = mfma(a4[c4].w, w4.w, x[c4]);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    __builtin_amdgcn_wave_barrier();   // the whole wave has read its C_IJ before they are overwritten
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        if (Is[c4] < nb) {
                            double* tij = Fdst + (int64_t)t64idx(Is[c4], J) * 256;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) tij[t64off(lq + 4 * rr, li)] = x[c4][rr];
                        }
                    }
                }
            }
            __syncthreads();
        };
        // ---- K p = q with the factor in Fs: right-hand side and solution through xv (permuted layout); the right-looking
        // sweeps of ftmpc_solve_f64.hip at four block rows per wave ----
        auto solve = [&]() {
            constexpr int RP = NBW / NWAVE;
            const int myp = v64pos(li);
            double* rb = part + wave * 16;
            f64x4 buf[RP];
            f64x4 wdiag = zero4;
            double psum[RP];
#pragma unroll
            for (int i = 0; i < RP; ++i) {
                psum[i] = 0.0;
                buf[i] = zero4;
            }
            auto fetch_f = [&](int J) {
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int I = wave + NWAVE * i;
                    if (I > J && I < nb) buf[i] = ld4(Fs + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                }
            };
            if (wave == 0) wdiag = ld4(Fs + (int64_t)t64idx(0, 0) * 256 + 16 * li + 4 * lq);
            fetch_f(0);
            for (int J = 0; J < nb; ++J) {
                if (wave == (J & (NWAVE - 1))) {       // owner: r_J = b_J - sum, y_J = W_J r_J
                    const int i = J / NWAVE;
                    double p = 0.0;
#pragma unroll
                    for (int ii = 0; ii < RP; ++ii) p = (ii == i) ? psum[ii] : p;
                    const double r = xv[16 * J + myp] - quad_sum64(p);
                    if (lq == 0) rb[li] = r;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    double y = wdiag.x * rb[lq] + wdiag.y * rb[4 + lq] + wdiag.z * rb[8 + lq] + wdiag.w * rb[12 + lq];
                    y = quad_sum64(y);
                    if (lq == 0) xv[16 * J + myp] = y;
                }
                lds_barrier();
                if (J + 1 < nb) {
                    const double* y4 = xv + 16 * J + 4 * lq;
                    const double y0 = y4[0], y1 = y4[1], y2 = y4[2], y3 = y4[3];
#pragma unroll
                    for (int i = 0; i < RP; ++i) {
                        const int I = wave + NWAVE * i;
                        if (I > J && I < nb) psum[i] += buf[i].x * y0 + buf[i].y * y1 + buf[i].z * y2 + buf[i].w * y3;
                    }
                    fetch_f(J + 1);
                    if (wave == ((J + 1) & (NWAVE - 1))) wdiag = ld4(Fs + (int64_t)t64idx(J + 1, J + 1) * 256 + 16 * li + 4 * lq);
                }
            }
            // ---- backward: L' x = y ----
#pragma unroll
            for (int i = 0; i < RP; ++i) psum[i] = 0.0;
            auto fetch_b = [&](int I) {               // row I of the factor, columns J = wave + 4 i < I (transposed use)
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int J = wave + NWAVE * i;
                    if (J < I) {
                        const double* t = Fs + (int64_t)t64idx(I, J) * 256;
                        buf[i].x = t[t64off(4 * lq + 0, li)];
                        buf[i].y = t[t64off(4 * lq + 1, li)];
                        buf[i].z = t[t64off(4 * lq + 2, li)];
                        buf[i].w = t[t64off(4 * lq + 3, li)];
                    }
                }
            };
            auto fetch_wt = [&](int I) {
                const double* t = Fs + (int64_t)t64idx(I, I) * 256;
                wdiag.x = t[t64off(4 * lq + 0, li)];
                wdiag.y = t[t64off(4 * lq + 1, li)];
                wdiag.z = t[t64off(4 * lq + 2, li)];
                wdiag.w = t[t64off(4 * lq + 3, li)];
            };
            if (wave == ((nb - 1) & (NWAVE - 1))) fetch_wt(nb - 1);
            fetch_b(nb - 1);
            for (int I = nb - 1; I >= 0; --I) {
                if (wave == (I & (NWAVE - 1))) {       // owner: r_I = y_I - sum, x_I = W_I' r_I
                    const int i = I / NWAVE;
                    double p = 0.0;
#pragma unroll
                    for (int ii = 0; ii < RP; ++ii) p = (ii == i) ? psum[ii] : p;
                    const double r = xv[16 * I + myp] - quad_sum64(p);
                    if (lq == 0) rb[li] = r;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    double x = wdiag.x * rb[4 * lq] + wdiag.y * rb[4 * lq + 1] + wdiag.z * rb[4 * lq + 2] + wdiag.w * rb[4 * lq + 3];
                    x = quad_sum64(x);
                    if (lq == 0) xv[16 * I + myp] = x;
                }
                lds_barrier();
                if (I > 0) {
                    const double x0 = xv[16 * I + 0 * 4 + lq], x1 = xv[16 * I + 1 * 4 + lq], x2 = xv[16 * I + 2 * 4 + lq],
                                 x3 = xv[16 * I + 3 * 4 + lq];      // x_I[4 lq + rr]
#pragma unroll
                    for (int i = 0; i < RP; ++i) {
                        const int J = wave + NWAVE * i;
                        if (J < I) psum[i] += buf[i].x * x0 + buf[i].y * x1 + buf[i].z * x2 + buf[i].w * x3;
                    }
                    fetch_b(I - 1);
                    if (wave == ((I - 1) & (NWAVE - 1))) fetch_wt(I - 1);
                }
            }
            __syncthreads();
        };
        // ---- wrench image of a thruster-space vector held NVT per thread: out = DD x, permuted layout ----
        auto to_wrench = [&](const double (&x)[NVT], double* out) {
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) rv[v * WG + tid] = tvalid[v] ? x[v] : 0.0;
            __syncthreads();
            if (tid < npad) {
                double s = 0.0;
                if (kcol != 255)
                    for (int a = 0; a < nat; ++a) s += s_DaT[acol * MAX_NT + a] * rv[kcol * nat + a];
                out[16 * (tid >> 4) + v64pos(tid & 15)] = s;
            }
            __syncthreads();
        };
        // ---- out = L' in (TRANS: tile of Lt times permuted vector, one 32-byte load per lane) or L in (the same tiles read by
        // columns: L_IJ[r][c] = Lt tile[c][r], four 8-byte loads per lane); output block rows dealt over the waves.
        // PERM_OUT: the result keeps the permuted layout (operand of the solve), else natural order ----
        auto tri_mv = [&](auto TRANS, auto PERM_OUT, const double* in, double* out) {
            constexpr bool tr = decltype(TRANS)::value;
            for (int Bo = wave; Bo < nb; Bo += NWAVE) {
                double a0 = 0.0, a1 = 0.0;
                const int X0 = tr ? Bo : 0, X1 = tr ? nb : Bo + 1;
                auto tile = [&](int X) -> f64x4 {
                    if constexpr (tr) {
                        return ld4(Lt + (int64_t)t64idx(X, Bo) * 256 + 16 * li + 4 * lq);      // L'_{Bo,X}[li][lq + 4 s]
                    } else {
                        const double* t = Lt + (int64_t)t64idx(Bo, X) * 256;                   // L_{Bo,X}[li][lq + 4 s] = L'_{X,Bo}[lq + 4 s][li]
                        return f64x4{t[t64off(lq, li)], t[t64off(lq + 4, li)], t[t64off(lq + 8, li)], t[t64off(lq + 12, li)]};
                    }
                };
                int X = X0;
                for (; X + 1 < X1; X += 2) {
                    const f64x4 t0 = tile(X), t1 = tile(X + 1);
                    const double* d0 = in + 16 * X + 4 * lq;
                    const double* d1 = d0 + 16;
                    a0 += t0.x * d0[0] + t0.y * d0[1] + t0.z * d0[2] + t0.w * d0[3];
                    a1 += t1.x * d1[0] + t1.y * d1[1] + t1.z * d1[2] + t1.w * d1[3];
                }
                if (X < X1) {
                    const f64x4 t0 = tile(X);
                    const double* d0 = in + 16 * X + 4 * lq;
                    a0 += t0.x * d0[0] + t0.y * d0[1] + t0.z * d0[2] + t0.w * d0[3];
                }
                const double a = quad_sum64(a0 + a1);
                if (lq == 0) out[16 * Bo + (decltype(PERM_OUT)::value ? v64pos(li) : li)] = a;
            }
            __syncthreads();
        };
        // ---- Newton system in thruster space through wrench space ----
        auto ws_solve = [&](const double (&r)[NVT], double (&x)[NVT]) {
            double t[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) t[v] = r[v] * rdg[v * WG + tid];
            to_wrench(t, tw);
            S64(4);
            tri_mv(std::true_type{}, std::true_type{}, tw, xv);
            S64(6);     // (diagnostic) the two triangular products with L
            solve();
            S64(5);     // (diagnostic) the solve with the factor of K
            tri_mv(std::false_type{}, std::false_type{}, xv, yv);
            S64(6);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 6; ++g) s += s_DaT[g * MAX_NT + ta[v]] * yv[tk[v] * 6 + g];
                x[v] = tvalid[v] ? (r[v] - s) * rdg[v * WG + tid] : 0.0;
            }
        };

        // ---------------- start point and its gradient: DD' (g_w + H_w DD d) + 2 rho (ubar + d) ----------------
        double lo[NVT], hi[NVT], sl[NVT], su[NVT], zl[NVT], zu[NVT], grad[NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            lo[v] = -ubar[v];
            hi[v] = ubv[v] - ubar[v];
            sl[v] = su[v] = 0.5 * ubv[v];
            zl[v] = zu[v] = grad[v] = 0.0;
        }
        {
            double dcur[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) dcur[v] = tvalid[v] ? lo[v] + sl[v] : 0.0;
            to_wrench(dcur, dv);
            for (int I = wave; I < nb; I += NWAVE) {     // H_w (DD d): block rows round-robin over the waves
                double a = 0.0;
                for (int J = 0; J < nb; ++J) {
                    if (J <= I) {
                        const f64x4 t4 = ld4(Ks + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                        const double* d4 = dv + 16 * J + 4 * lq;
                        a += t4.x * d4[0] + t4.y * d4[1] + t4.z * d4[2] + t4.w * d4[3];
                    } else {
                        const double* t = Ks + (int64_t)t64idx(J, I) * 256;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) a += t[t64off(4 * lq + rr, li)] * dv[16 * J + rr * 4 + lq];
                    }
                }
                a = quad_sum64(a);
                if (lq == 0) yv[16 * I + li] = a + gwv[16 * I + li];
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 6; ++g) s += s_DaT[g * MAX_NT + ta[v]] * yv[tk[v] * 6 + g];
                grad[v] = tvalid[v] ? s + 2.0 * rho * (ubar[v] + dcur[v]) : 0.0;
            }
        }
        // ---------------- H_w = L L' once: the factor by rows (Lr) and transposed (Lt) ----------------
        __syncthreads();
        if (tid < npad) dv[tid] = 0.0;
        __syncthreads();
        factor(Ks, Ks, std::true_type{});
        if (s_flag == 0) {     // H_w not positive definite: report the linearisation point, do not iterate
            write_flat(2, true, ubar, tvalid, tk, ta);
            continue;
        }
        for (int t = wave; t < ntl; t += NWAVE) {     // Lt: the transposed tiles (the diagonal slot of the factor holds W, L_JJ came by Ld)
            const int I = tIJ[2 * t], J = tIJ[2 * t + 1];
            const double* src = (I == J) ? Ld + I * 256 : Ks + (int64_t)t * 256;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) Lt[(int64_t)t * 256 + t64off(li, lq + 4 * rr)] = src[t64off(lq + 4 * rr, li)];
        }
        __syncthreads();
        if (tid < npad) dv[tid] = 1.0;       // the identity of K = I + L' S L
        __syncthreads();
        S64(4);

        // ---------------- interior-point iterations (thruster space) ----------------
        int status = 1, nit = 0;
        const double inv2n = 1.0 / (double)(2 * nt);
        {
            double gm = 0.0, wm = 0.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) {
                    gm = fmax(gm, fabs(grad[v]));
                    wm = fmax(wm, ubv[v]);
                }
            gm = f64k::wg_max(gm, red, tid);
            wm = f64k::wg_max(wm, red, tid);
            const double mu0 = fmax(0.02 * gm * wm, 1e-3);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                zl[v] = tvalid[v] ? mu0 / sl[v] : 0.0;
                zu[v] = tvalid[v] ? mu0 / su[v] : 0.0;
            }
        }
        for (int it = 0; it <= C.max_iters; ++it) {
            double csum = 0.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) csum += sl[v] * zl[v] + su[v] * zu[v];
            const double mu = f64k::wg_sum(csum, red, tid) * inv2n;
            if (!(mu >= C.mu_stop)) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            double Sig[NVT];
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                Sig[v] = tvalid[v] ? zl[v] / sl[v] + zu[v] / su[v] : 0.0;
                rdg[v * WG + tid] = tvalid[v] ? 1.0 / (2.0 * rho + Sig[v]) : 0.0;
            }
            __syncthreads();
            for (int idx = tid; idx < N * 21; idx += WG) {          // stage blocks S_k = D_a diag(1 / Dg) D_a'
                const int k = idx / 21, p = idx - 21 * k;
                int g = 0;
                while ((g + 1) * (g + 2) / 2 <= p) ++g;
                const int hh = p - g * (g + 1) / 2;
                double sacc = 0.0;
                for (int a = 0; a < nat; ++a) sacc += s_DD[p * MAX_NT + a] * rdg[k * nat + a];
                Sblk[k * 36 + g * 6 + hh] = sacc;
                Sblk[k * 36 + hh * 6 + g] = sacc;
            }
            __syncthreads();
            S64(4);
            // ---- K - I = X = L' S L, one block row I at a time:
            //   P'_IM = sum_K L'_IK S_KM  (M >= I - 1; K in {M-1, M, M+1}, K >= I) into LDS, each tile laid out as the A operand its
            //           readers want (A operand of the product: the tile of Lt as it stands; B operand: rows of S_MK, read off the
            //           stage blocks -- S is block diagonal, no tiles of it exist);
            //   X_IJ  = sum_{M >= max(I-1, J)} P'_IM (L'_JM)'  (J <= I), A operand from LDS, B operand the tile of Lt, four block
            //           columns per step with the next step's tiles requested before the products of the current one.
            // P' never goes to global memory, and every tile of a row has the same number of products. ----
            for (int I = 0; I < nb; ++I) {
                const int Mlo = (I > 0) ? I - 1 : 0;
                for (int M = Mlo + wave; M < nb; M += NWAVE) {
                    const int e1 = 16 * M + li;
                    const int s1 = s_stg[e1], a1 = s_thr[e1];
                    f64x4 acc = zero4, acc2 = zero4;
                    const int K0 = (M - 1 > I) ? M - 1 : I, K1 = (M + 1 < nb - 1) ? M + 1 : nb - 1;
                    for (int K = K0; K <= K1; ++K) {
                        const f64x4 a4 = ld4(Lt + (int64_t)t64idx(K, I) * 256 + 16 * li + 4 * lq);
                        double b[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int e2 = 16 * K + 4 * s + lq;
                            const int s2 = s_stg[e2];
                            b[s] = (s1 != 255 && s1 == s2) ? Sblk[s1 * 36 + a1 * 6 + s_thr[e2]] : 0.0;
                        }
                        acc = mfma(a4.x, b[0], acc);
                        acc2 = mfma(a4.y, b[1], acc2);
                        acc = mfma(a4.z, b[2], acc);
                        acc2 = mfma(a4.w, b[3], acc2);
                    }
                    acc += acc2;
                    // element (row lq + 4 rr, column li) belongs to the reader lane (li & 3, lq + 4 rr), k-step li >> 2
                    double* pt = Pj + (M - Mlo) * 256 + 4 * (16 * (li & 3) + lq) + (li >> 2);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) pt[16 * rr] = acc[rr];
                }
                __syncthreads();
                for (int J = wave; J <= I; J += NWAVE) {
                    const int M0 = (I - 1 > J) ? I - 1 : J;
                    f64x4 acc = zero4, acc2 = zero4;
                    f64x4 B0[4], B1[4];
                    auto fetchB = [&](f64x4 (&Bq)[4], int Mb) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            Bq[i] = zero4;
                            if (Mb + i < nb) Bq[i] = ld4(Lt + (int64_t)t64idx(Mb + i, J) * 256 + 16 * li + 4 * lq);
                        }
                    };
                    auto mm = [&](const f64x4 (&Bq)[4], int Mb) {
                        f64x4 A[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            A[i] = zero4;
                            if (Mb + i < nb) A[i] = ld4(Pj + (Mb + i - Mlo) * 256 + 4 * lane);
                        }
                        acc = mfma(A[0].x, Bq[0].x, acc); acc2 = mfma(A[1].x, Bq[1].x, acc2);
                        acc = mfma(A[0].y, Bq[0].y, acc); acc2 = mfma(A[1].y, Bq[1].y, acc2);
                        acc = mfma(A[0].z, Bq[0].z, acc); acc2 = mfma(A[1].z, Bq[1].z, acc2);
                        acc = mfma(A[0].w, Bq[0].w, acc); acc2 = mfma(A[1].w, Bq[1].w, acc2);
                        acc = mfma(A[2].x, Bq[2].x, acc); acc2 = mfma(A[3].x, Bq[3].x, acc2);
                        acc = mfma(A[2].y, Bq[2].y, acc); acc2 = mfma(A[3].y, Bq[3].y, acc2);
                        acc = mfma(A[2].z, Bq[2].z, acc); acc2 = mfma(A[3].z, Bq[3].z, acc2);
                        acc = mfma(A[2].w, Bq[2].w, acc); acc2 = mfma(A[3].w, Bq[3].w, acc2);
                    };
                    fetchB(B0, M0);
                    for (int Mb = M0; Mb < nb; Mb += 8) {
                        if (Mb + 4 < nb) fetchB(B1, Mb + 4);
                        mm(B0, Mb);
                        if (Mb + 4 < nb) {
                            if (Mb + 8 < nb) fetchB(B0, Mb + 8);
                            mm(B1, Mb + 4);
                        }
                    }
                    acc += acc2;
                    double* kt = Ks + (int64_t)t64idx(I, J) * 256;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) kt[t64off(lq + 4 * rr, li)] = acc[rr];
                }
                __syncthreads();
            }
            S64(7);
            factor(Ks, Fs, std::false_type{});
            if (s_flag == 0) {
                status = 2;
                break;
            }
            S64(9);
            // predictor: (H + Sig) da = -grad
            double ngrad[NVT], da[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) ngrad[v] = -grad[v];
            ws_solve(ngrad, da);
            double dzl_a[NVT], dzu_a[NVT], ap = 1.0, ad = 1.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                dzl_a[v] = dzu_a[v] = 0.0;
                if (tvalid[v]) {
                    dzl_a[v] = -zl[v] - zl[v] * da[v] / sl[v];
                    dzu_a[v] = -zu[v] + zu[v] * da[v] / su[v];
                    if (da[v] < 0.0) ap = fmin(ap, -sl[v] / da[v]);
                    if (da[v] > 0.0) ap = fmin(ap, su[v] / da[v]);
                    if (dzl_a[v] < 0.0) ad = fmin(ad, -zl[v] / dzl_a[v]);
                    if (dzu_a[v] < 0.0) ad = fmin(ad, -zu[v] / dzu_a[v]);
                }
            }
            ap = f64k::wg_min(ap, red, tid);
            ad = f64k::wg_min(ad, red, tid);
            csum = 0.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) csum += (sl[v] + ap * da[v]) * (zl[v] + ad * dzl_a[v]) + (su[v] - ap * da[v]) * (zu[v] + ad * dzu_a[v]);
            const double mu_aff = f64k::wg_sum(csum, red, tid) * inv2n;
            double sigma = mu_aff / mu;
            sigma = fmin(fmax(sigma * sigma * sigma, 0.0), 1.0);
            // corrector
            double rcl[NVT], rcu[NVT], rhs[NVT], dd[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                rcl[v] = rcu[v] = rhs[v] = 0.0;
                if (tvalid[v]) {
                    rcl[v] = sl[v] * zl[v] + da[v] * dzl_a[v] - sigma * mu;
                    rcu[v] = su[v] * zu[v] - da[v] * dzu_a[v] - sigma * mu;
                    rhs[v] = -(grad[v] - zl[v] + zu[v]) - rcl[v] / sl[v] + rcu[v] / su[v];
                }
            }
            ws_solve(rhs, dd);
            ap = 1e300;
            ad = 1e300;
            double dzl[NVT], dzu[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                dzl[v] = dzu[v] = 0.0;
                if (tvalid[v]) {
                    dzl[v] = (-rcl[v] - zl[v] * dd[v]) / sl[v];
                    dzu[v] = (-rcu[v] + zu[v] * dd[v]) / su[v];
                    if (dd[v] < 0.0) ap = fmin(ap, -sl[v] / dd[v]);
                    if (dd[v] > 0.0) ap = fmin(ap, su[v] / dd[v]);
                    if (dzl[v] < 0.0) ad = fmin(ad, -zl[v] / dzl[v]);
                    if (dzu[v] < 0.0) ad = fmin(ad, -zu[v] / dzu[v]);
                }
            }
            ap = fmin(1.0, 0.9995 * f64k::wg_min(ap, red, tid));
            ad = fmin(1.0, 0.9995 * f64k::wg_min(ad, red, tid));
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) {
                    grad[v] += ap * (rhs[v] - Sig[v] * dd[v]);   // + ap H dd
                    sl[v] += ap * dd[v];
                    su[v] -= ap * dd[v];
                    zl[v] += ad * dzl[v];
                    zu[v] += ad * dzu[v];
                }
            S64(4);     // (diagnostic) the two Newton solves + element-wise work
        }
        // ---------------- outputs ----------------
        __syncthreads();
        double* ubuf = Pj;  // N*NT <= 1024 doubles
        for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NVT; ++v)
            if (tvalid[v]) {
                double u = (sl[v] < su[v]) ? sl[v] : ubv[v] - su[v];
                if (status == 2) u = ubar[v];
                ubuf[tk[v] * NT + s_act[ta[v]]] = u;
            }
        __syncthreads();
        if (tid < NT) P.out_u0[inst * NT + tid] = ubuf[tid];
        if (P.out_U)
            for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = ubuf[i];
        if (tid == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        S64(0);
#ifdef FTMPC_STAMPS
        if (tid == 0 && inst < 512 && P.dbg_H) {
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(P.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = s64_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_ws64_kernel<1>(const DeviceConsts, const SolveWs64Params);   // N * na <= 256 (the reference vehicle, N <= 16)
template __global__ void ftmpc_solve_ws64_kernel<3>(const DeviceConsts, const SolveWs64Params);   // N * na <= 768 (BASELINE config 5: N = 40, 16 thrusters)

}  // namespace ftmpc
