// ftmpc_solve_wg.hip -- kernel 7: condensed-QP build + primal-dual IPM in fp32 for 160 < n <= 240 (the reference's own
// vehicle: 16 thrusters at its shipped horizon N = 15, reactive.yaml:26 -> n = 210..240), ONE 4-WAVE WORKGROUP PER
// INSTANCE with the KKT FACTOR RESIDENT IN LDS.
//
// Same algorithm and the same tile conventions as the one-wave kernels of ftmpc_solve.hip (accumulator layout, tiles
// stored negated / transposed in register order, mm_tn products, in-register potrf + inverse of the diagonal tiles,
// gradient by recurrence with ONE float64 structured reference gradient); what changes is where things live and who works:
//   * the factor (<= 120 tiles) + the inverses of its diagonal blocks (<= 15 tiles) sit in LDS (<= 135 KiB, one workgroup
//     per CU); the Hessian tiles (-H', read once per iteration) and the E panels of the build sit in a per-workgroup
//     global slot (L2-resident);
//   * left-looking blocked Cholesky, block rows dealt round-robin to the four waves, ONE workgroup barrier per block
//     column.  The serial chain per column is  [finish tile (J+1,J)] -> [panel solve] -> [last diagonal term] -> potrf(J+1);
//     everything else (Schur sums of the next column, the diagonal tile two columns ahead) is done by the other waves
//     while the owner factorises, from operands that were published before the barrier;
//   * the two triangular sweeps run on wave 0 alone (tiles and vectors from LDS: no barrier on the serial chain);
//   * one variable per thread for the element-wise interior-point arithmetic.
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354 (NLP + IPOPT) and
// controllers/tools/control_allocator.py:65-94 (see DESIGN.md QP-spec); oracle/qp_oracle.py:ipm_box is the mirror.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace wgk {
constexpr int WG = 256;
constexpr int NWAVE = 4;
__host__ __device__ constexpr int ntiles(int nb) { return nb * (nb + 1) / 2; }
// per-workgroup global slot, in 4-byte words:
//   float64 scratch of the reference gradient: NPAD (gradient) + 8 N (wrenches) + 9 (N + 1) (stage storage) doubles
//   E panels: N x 9 x NPAD floats;   Hessian tiles: ntiles x 256 floats
__host__ __device__ constexpr int64_t slot_f64_words(int npad, int N) { return 2 * (int64_t)(npad + 8 * N + 9 * (N + 1)); }
__host__ __device__ constexpr int64_t slot_e_off(int npad, int N) { return ((slot_f64_words(npad, N) + 255) / 256) * 256; }
__host__ __device__ constexpr int64_t slot_h_off(int npad, int N) { return slot_e_off(npad, N) + (int64_t)N * 9 * npad; }
__host__ __device__ constexpr int64_t slot_words(int nbmax, int N) { return slot_h_off(16 * nbmax, N) + (int64_t)ntiles(nbmax) * 256; }
}  // namespace wgk

namespace {
// X'Y into two accumulators (two independent MFMA chains: a dependent fp32 16x16x4 MFMA costs 40 cycles, an independent one 32)
__device__ __forceinline__ void mm_tn2(const f32x4& X, const f32x4& Y, f32x4& a, f32x4& b) {
    a = mfma4(X.x, Y.x, a);
    b = mfma4(X.y, Y.y, b);
    a = mfma4(X.z, Y.z, a);
    b = mfma4(X.w, Y.w, b);
}
// sum_K T(D, K)' T(D, K) for K < Kend (the diagonal tile D)
__device__ __forceinline__ f32x4 schur_diag(const float* Tl, int lane, int D, int Kend) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (Kend <= 0) return a;
    const float* pd = Tl + tidx(D, 0) * 256 + 4 * lane;
    for (int K = 0; K < Kend; ++K) {
        const f32x4 tk = lds4(pd + K * 256);
        mm_tn2(tk, tk, a, b);
    }
    return a + b;
}
// KKT solve on ONE wave, vectors in registers, tiles streamed from LDS (the serial chain per block is
// reduce -> W mat-vec -> reduce, as in solve_reg of ftmpc_solve.hip; the tiles of a block column / row do not depend on
// the chain and are requested at the top of the step).  xv: right-hand side in, solution out (natural order).
template <int NB>
__device__ __forceinline__ void solve_lds(const float* Tl, const float* Wdl, float* xv, int nb, int lane) {
    const int li = lane & 15, lq = lane >> 4;
    f32x4 Y[NB];
    f32x2 p[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) p[J] = f32x2{0.f, 0.f};
    // forward (right-looking): y_J = W_J (b_J - p_J),  p_I += L_IJ y_J for I > J
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        if (J < nb) {
            f32x4 tc[NB];
#pragma unroll
            for (int I = J + 1; I < NB; ++I)
                if (I < nb) tc[I] = lds4(Tl + tidx(I, J) * 256 + 4 * lane);
            const f32x4 w = lds4(Wdl + J * 256 + 4 * lane);
            float r = xv[16 * J + li];
            if (J > 0) r -= quad_sum(p[J].x + p[J].y);
            float y0 = w.x * r, y1 = w.y * r, y2 = w.z * r, y3 = w.w * r;
            row_sum16x4(y0, y1, y2, y3);
            Y[J] = f32x4{y0, y1, y2, y3};
#pragma unroll
            for (int I = J + 1; I < NB; ++I)
                if (I < nb) {
                    p[I] += f32x2{tc[I].x, tc[I].y} * f32x2{y0, y1};
                    p[I] += f32x2{tc[I].z, tc[I].w} * f32x2{y2, y3};
                }
        }
    }
    // backward: x_J = W_J' (y_J - a_J),  a_K += L_JK' x_J for K < J
    f32x4 a[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) a[J] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
        if (J < nb) {
            f32x4 tr[NB];
#pragma unroll
            for (int K = 0; K < J; ++K) tr[K] = lds4(Tl + tidx(J, K) * 256 + 4 * lane);
            const f32x4 w = lds4(Wdl + J * 256 + 4 * lane);
            f32x4 r = Y[J];
            if (J + 1 < nb) {
                float s0 = a[J].x, s1 = a[J].y, s2 = a[J].z, s3 = a[J].w;
                row_sum16x4(s0, s1, s2, s3);
                r -= f32x4{s0, s1, s2, s3};
            }
            const float xr = quad_sum(w.x * r.x + w.y * r.y + w.z * r.z + w.w * r.w);
            if (lq == 0) xv[16 * J + li] = xr;
#pragma unroll
            for (int K = 0; K < J; ++K) a[K] += tr[K] * xr;
        }
    }
}
}  // namespace

struct SolveWgParams {
    SolveParams base;    // hscratch / tile_words unused
    float* slot;         // [gridDim.x][slot_words]
    int64_t slot_words;
};

template <int NBMAX>
__global__ void __launch_bounds__(wgk::WG, 1) ftmpc_solve_wg32_kernel(const DeviceConsts C, const SolveWgParams Q) {
    using namespace wgk;
    constexpr int NPAD = 16 * NBMAX;
    constexpr int NT_ALL = ntiles(NBMAX);
    static_assert(NPAD <= WG, "one variable per thread");
    const SolveParams& P = Q.base;
    __shared__ __attribute__((aligned(16))) float Tl[(NT_ALL + NBMAX) * 256];   // factor tiles (diagonal slot: W') | W of every diagonal block
    __shared__ __attribute__((aligned(16))) float xv[NPAD], yv[NPAD], sigv[NPAD], dnat[NPAD];
    __shared__ __attribute__((aligned(16))) double recd[REC_STRIDE + 4];
    __shared__ __attribute__((aligned(16))) double sSl[9 * 33];                  // stage storage of the reference gradient (N <= 32)
    __shared__ __attribute__((aligned(16))) float recf[REC_STRIDE];
    __shared__ float S17[16 * 17];
    __shared__ __attribute__((aligned(16))) float s_Da[6 * MAX_NT];
    __shared__ float s_MR[MAX_NT * MAX_NT];
    __shared__ float red[NWAVE];
    __shared__ unsigned char s_stg[NPAD], s_thr[NPAD];
    __shared__ int s_act[MAX_NT];
    __shared__ int s_flag, s_q;
    float* const Wdl = Tl + NT_ALL * 256;
    __shared__ float s_D[6 * MAX_NT];

    const int tid = threadIdx.x;
    // wave-uniform values are forced into SGPRs: branches on them become scalar branches instead of EXEC masking
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int N = C.N, NT = C.NT;
    const float rho = (float)C.rho;
    const float mu_stop = (float)C.mu_stop;
    float* const slot = Q.slot + (int64_t)blockIdx.x * Q.slot_words;
    double* const sbuf = reinterpret_cast<double*>(slot);
    float* const Eall = slot + slot_e_off(NPAD, N);
    float* const Hs = slot + slot_h_off(NPAD, N);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto ldt = [&](int t) { return *reinterpret_cast<const f32x4*>(Tl + t * 256 + 4 * lane); };
    auto stt = [&](int t, f32x4 v) { *reinterpret_cast<f32x4*>(Tl + t * 256 + 4 * lane) = v; };
    auto ldh = [&](int t) { return *reinterpret_cast<const f32x4*>(Hs + (int64_t)t * 256 + 4 * lane); };
    // workgroup reductions (result in every thread): DPP / permlane inside the wave, LDS across the four waves
    auto wg_reduce = [&](float x, auto op) {
        x = wave_reduce<decltype(op)>(x);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        return decltype(op)::f(decltype(op)::f(red[0], red[1]), decltype(op)::f(red[2], red[3]));
    };

    // the allocation matrix goes to LDS through CONSTANT indices: one dynamically indexed access to the by-value
    // argument block would move the whole block (2.5 KiB per lane) to scratch and turn every later C.x into a scratch load
    // (as a chain of selects: `if (tid == i) s_D[i] = ...` became a 60 000-instruction decision tree on tid, run once per launch)
    {
        double dsel = 0.0;
#pragma unroll
        for (int i = 0; i < 6 * MAX_NT; ++i) dsel = (tid == i) ? C.D[i] : dsel;
        if (tid < 6 * MAX_NT) s_D[tid] = (float)dsel;
    }
    const float dtf = (float)C.dt;
    const int qn = *P.qcount;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_q = atomicAdd(P.qhead, 1);
        __syncthreads();
        const int qi = __builtin_amdgcn_readfirstlane(s_q);
        if (qi >= qn) break;
        const int64_t inst = __builtin_amdgcn_readfirstlane(P.qlist[qi]);
        STAMP_DECL;
        STAMP_START();
        // ---------------- prologue ----------------
        if (tid == 0) {
            int na0 = 0;
            for (int i = 0; i < NT; ++i)
                if (P.ub[inst * NT + i] > 0.0) s_act[na0++] = i;
            s_flag = na0;
        }
        __syncthreads();
        const int na = __builtin_amdgcn_readfirstlane(s_flag);
        const int n = N * na;
        const int nb = (n + 15) >> 4;
        const int npad = nb * 16;
        if (na == 0 || nb > NBMAX) {
            for (int i = tid; i < NT; i += WG) P.out_u0[inst * NT + i] = 0.0;
            if (P.out_U)
                for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = 0.0;
            if (tid == 0) {
                if (P.status) P.status[inst] = (na == 0) ? 0 : 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        if (tid < 6 * MAX_NT) {
            const int g = tid / MAX_NT, a = tid % MAX_NT;
            s_Da[tid] = (a < na) ? s_D[g * MAX_NT + s_act[a]] : 0.f;
        }
        if (tid < npad) {
            const int s = tid / na;
            s_stg[tid] = (unsigned char)(tid < n ? s : 255);
            s_thr[tid] = (unsigned char)(tid < n ? tid - s * na : 255);
        }
        __syncthreads();
        if (tid < na * na) {
            const int a = tid / na, b = tid % na;
            float t = 0.f;
            for (int g = 0; g < 6; ++g) t += s_Da[g * MAX_NT + a] * (float)C.R[g] * s_Da[g * MAX_NT + b];
            s_MR[a * MAX_NT + b] = 2.f * (t + (a == b ? rho : 0.f));
        }
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        const int kcol = (tid < npad) ? s_stg[tid] : 255;
        const int acol = (tid < npad) ? s_thr[tid] : 255;
        const bool valid = kcol != 255;
        float ubar = 0.f, ubv = 1.f, gacc = 0.f;
        float Fd[3] = {0.f, 0.f, 0.f}, Td[3] = {0.f, 0.f, 0.f};
        if (valid) {
            const int t = s_act[acol];
            ubv = (float)P.ub[inst * NT + t];
            if (P.warmU) ubar = fminf(fmaxf((float)P.warmU[(inst * N + kcol) * NT + t], 0.f), ubv);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                Fd[a] = s_Da[a * MAX_NT + acol];
                Td[a] = s_Da[(3 + a) * MAX_NT + acol];
            }
        }
        STAMP(0);
        // ---------------- build: the E panels of every stage go to LDS when they fit (the factor area is idle during the
        // build: N x 9 x npad floats), else to the global slot; two instantiations so that each keeps its address space ----------------
        const int ntl = ntiles(nb);
        auto build = [&](auto IN_LDS) {
        float* const Eb = decltype(IN_LDS)::value ? Tl : Eall;
        // ---------------- phase 1: condense (one column per thread) ----------------
        float G[13];
#pragma unroll
        for (int r = 0; r < 13; ++r) G[r] = 0.f;
        for (int k = 0; k < N; ++k) {
            __syncthreads();
            if (tid < REC_STRIDE) recf[tid] = (float)recg[k * REC_STRIDE + tid];
            __syncthreads();
            const float* rb = recf;
            const bool terminal = (k + 1 == N);
            if (tid < npad) {
                if (kcol < k) {
                    float p[3], vv[3], w[3], q[4];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        p[a] = G[a] + dtf * G[3 + a];
                        vv[a] = G[3 + a];
                        w[a] = 0.f;
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
                        q[a] = 0.f;
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
                    float gr = 0.f;
#pragma unroll
                    for (int a = 0; a < 3; ++a) gr += Fd[a] * rb[REC_RUT + a] + Td[a] * rb[REC_RUT + 3 + a];
                    gacc += gr;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float sp = 0.f, sv = 0.f, sw = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            sp += rb[REC_BPF + 3 * a + c] * Fd[c] + rb[REC_BPT + 3 * a + c] * Td[c];
                            sv += rb[REC_BVF + 3 * a + c] * Fd[c] + rb[REC_BVT + 3 * a + c] * Td[c];
                            sw += rb[REC_BWT + 3 * a + c] * Td[c];
                        }
                        G[a] = sp;
                        G[3 + a] = sv;
                        G[6 + a] = sw;
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float sq = 0.f;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sq += rb[REC_BQT + 3 * a + c] * Td[c];
                        G[9 + a] = sq;
                    }
                }
                float gs = 0.f;
#pragma unroll
                for (int r = 0; r < 9; ++r) gs += G[r] * rb[REC_WE + r];
                gacc += gs;
                float* Ek = Eb + (int64_t)k * 9 * npad;
                if (!terminal) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) Ek[r * npad + tid] = (float)C.sq2Q[r] * G[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        float s = 0.f;
#pragma unroll
                        for (int c = r; c < 9; ++c) s += (float)C.LPt[9 * r + c] * G[c];
                        Ek[r * npad + tid] = s;
                    }
                }
            }
        }
        __syncthreads();   // E panels visible to the whole workgroup
        STAMP(1);
        // ---------------- build, phase 2: Hessian tiles on the matrix cores, -H' in register order -> slot ----------------
        for (int t = wave; t < ntl; t += NWAVE) {
            int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (tidx(I + 1, 0) <= t) ++I;
            while (tidx(I, 0) > t) --I;
            const int J = t - tidx(I, 0);
            f32x4 acc = zero4;
            const int kstart = (16 * I) / na < N ? (16 * I) / na : N;
            for (int k = kstart; k < N; ++k) {
                const float* Ek = Eb + (int64_t)k * 9 * npad;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int r = 4 * s3 + lq;
                    const float a = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.f;   // A[m][k] = E[r][16J + m]
                    const float b = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.f;   // B[k][n] = E[r][16I + n]
                    acc = mfma4(a, b, acc);                                       // (E_J' E_I) = (H_IJ)'
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int e1 = (I == J) ? 16 * I + 4 * lq + rr : 16 * I + li;
                const int e2 = (I == J) ? 16 * J + li : 16 * J + 4 * lq + rr;
                const int s1 = s_stg[e1], s2 = s_stg[e2];
                float add = (s1 != 255 && s1 == s2) ? s_MR[s_thr[e1] * MAX_NT + s_thr[e2]] : 0.f;
                if (s1 == 255 && e1 == e2) add = 1.f;
                acc[rr] += add;
            }
            *reinterpret_cast<f32x4*>(Hs + (int64_t)t * 256 + 4 * lane) = -acc;
        }
        __syncthreads();
        STAMP(2);
        };
        if ((int64_t)N * 9 * npad <= (int64_t)(NT_ALL + NBMAX) * 256) build(std::true_type{});
        else build(std::false_type{});
        float gv = valid ? 2.f * (gacc + rho * ubar) : 0.f;
        const float lo = -ubar, hi = ubv - ubar;
        float sl = 0.5f * ubv, su = 0.5f * ubv, zl = 0.f, zu = 0.f, grad = 0.f;
        if (P.dbg_inst == inst) {   // test hook: the QP this workgroup is about to solve
            for (int t = wave; t < ntl; t += NWAVE) {
                int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while (tidx(I + 1, 0) <= t) ++I;
                while (tidx(I, 0) > t) --I;
                const int J = t - tidx(I, 0);
                const f32x4 ht = ldh(t);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int e1 = (I == J) ? 16 * I + 4 * lq + rr : 16 * I + li;
                    const int e2 = (I == J) ? 16 * J + li : 16 * J + 4 * lq + rr;
                    if (I != J || e1 >= e2) {
                        P.dbg_H[(int64_t)e1 * npad + e2] = -ht[rr];
                        P.dbg_H[(int64_t)e2 * npad + e1] = -ht[rr];
                    }
                }
            }
            if (tid < npad) {
                P.dbg_vec[tid] = gv;
                P.dbg_vec[npad + tid] = lo;
                P.dbg_vec[2 * npad + tid] = hi;
            }
            if (tid == 0) {
                P.dbg_vec[3 * NPAD] = (float)n;
                P.dbg_vec[3 * NPAD + 1] = (float)npad;
            }
        }

        // ---------------- KKT factorisation (see the header) ----------------
        // Phase J (W_J and every tile of the columns < J visible) has one wave on the critical chain and three helpers:
        //   owner (J+1) & 3 : tile (J+1, J), the last two terms of diagonal tile J+1, its Cholesky + inverse, W_{J+1}
        //   helpers         : the other tiles (I, J) of column J (panel solve), the Schur sums of column J+1 and of
        //                     diagonal tile J+2 over the columns < J -- dealt tile by tile over the three of them.
        // Partial sums travel through the LDS slot of the tile they belong to (written in one phase, finished in the next),
        // so no tile is tied to a wave and the owner carries nothing but the chain.
        auto factor = [&]() {
            if (tid == 0) s_flag = 1;
            auto potrf_publish = [&](int D, const f32x4& dsum) {
                // diagonal tile D: H + Sigma - sum, Cholesky + inverse in registers, W and W' to LDS
                const float sg = sigv[16 * D + li];
                f32x4 cd;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) cd[rr] = ((4 * lq + rr == li) ? sg : 0.f) - dsum[rr];
                const f32x4 w = potrf_inv16_call(cd, lane, NoWork{});
                if (!(fabsf(w.w) <= 3.0e38f) && lane == 63) s_flag = 0;
                *reinterpret_cast<f32x4*>(Wdl + D * 256 + 4 * lane) = w;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) S17[(4 * lq + rr) * 17 + li] = w[rr];
                wave_lds_fence();
                f32x4 wt;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) wt[rr] = S17[li * 17 + 4 * lq + rr];
                stt(tidx(D, D), wt);
            };
            if (wave == 0) potrf_publish(0, ldh(tidx(0, 0)));
            __syncthreads();   // W_0 published
            for (int J = 0; J + 1 < nb; ++J) {
                const int owner = (J + 1) & 3;
                const int r = (wave - owner - 1) & 3;          // 0..2: helper rank, 3: the owner
                f32x4 wtn = ldt(tidx(J, J));
                wtn = -wtn;
                const f32x4 tjl = (J > 0) ? ldt(tidx(J, J - 1)) : zero4;       // T(J, J-1): the last Schur term of column J
                if (r == 3) {
                    f32x4 part, dsum;
                    if (J > 0) {
                        const f32x4 tl = ldt(tidx(J + 1, J - 1));
                        part = ldt(tidx(J + 1, J));
                        dsum = ldt(tidx(J + 1, J + 1));
                        part = mm_tn(tjl, tl, part);
                        dsum = mm_tn(tl, tl, dsum);
                    } else {
                        part = ldh(tidx(1, 0));
                        dsum = ldh(tidx(1, 1));
                    }
                    const f32x4 tij = mm_tn(wtn, part, zero4);
                    stt(tidx(J + 1, J), tij);
                    dsum = mm_tn(tij, tij, dsum);
                    potrf_publish(J + 1, dsum);
                } else {
                    // the other tiles of column J, two at a time (independent MFMA chains)
                    for (int I = J + 2 + r; I < nb; I += 6) {
                        const int I2 = I + 3;
                        if (I2 < nb) {
                            f32x4 pa, pb;
                            if (J > 0) {
                                const f32x4 ta = ldt(tidx(I, J - 1)), tb = ldt(tidx(I2, J - 1));
                                pa = ldt(tidx(I, J));
                                pb = ldt(tidx(I2, J));
#pragma unroll
                                for (int s4 = 0; s4 < 4; ++s4) {
                                    pa = mfma4(tjl[s4], ta[s4], pa);
                                    pb = mfma4(tjl[s4], tb[s4], pb);
                                }
                            } else {
                                pa = ldh(tidx(I, 0));
                                pb = ldh(tidx(I2, 0));
                            }
                            f32x4 xa = zero4, xb = zero4;
#pragma unroll
                            for (int s4 = 0; s4 < 4; ++s4) {
                                xa = mfma4(wtn[s4], pa[s4], xa);
                                xb = mfma4(wtn[s4], pb[s4], xb);
                            }
                            stt(tidx(I, J), xa);
                            stt(tidx(I2, J), xb);
                        } else {
                            f32x4 pa;
                            if (J > 0) {
                                const f32x4 ta = ldt(tidx(I, J - 1));
                                pa = ldt(tidx(I, J));
                                pa = mm_tn(tjl, ta, pa);
                            } else {
                                pa = ldh(tidx(I, 0));
                            }
                            stt(tidx(I, J), mm_tn(wtn, pa, zero4));
                        }
                    }
                    // Schur sums of column J+1 over the columns < J, Hessian tile included, into the tiles' own slots
                    const float* pj = Tl + tidx(J + 1, 0) * 256 + 4 * lane;
                    for (int I = J + 2 + r; I < nb; I += 6) {
                        const int I2 = I + 3;
                        const float* pa = Tl + tidx(I, 0) * 256 + 4 * lane;
                        if (I2 < nb) {
                            const float* pb = Tl + tidx(I2, 0) * 256 + 4 * lane;
                            const f32x4 ha = ldh(tidx(I, J + 1)), hb = ldh(tidx(I2, J + 1));
                            f32x4 a0 = zero4, a1 = zero4, b0 = zero4, b1 = zero4;
                            for (int K = 0; K < J; ++K) {
                                const f32x4 tj = lds4(pj + K * 256), ta = lds4(pa + K * 256), tb = lds4(pb + K * 256);
                                mm_tn2(tj, ta, a0, a1);
                                mm_tn2(tj, tb, b0, b1);
                            }
                            stt(tidx(I, J + 1), a0 + a1 + ha);
                            stt(tidx(I2, J + 1), b0 + b1 + hb);
                        } else {
                            const f32x4 ha = ldh(tidx(I, J + 1));
                            f32x4 a0 = zero4, a1 = zero4;
                            for (int K = 0; K < J; ++K) {
                                const f32x4 tj = lds4(pj + K * 256), ta = lds4(pa + K * 256);
                                mm_tn2(tj, ta, a0, a1);
                            }
                            stt(tidx(I, J + 1), a0 + a1 + ha);
                        }
                    }
                    // diagonal tile two columns ahead over the columns < J (the helper with the fewest tiles)
                    if (r == 2 && J + 2 < nb) {
                        const f32x4 hd = ldh(tidx(J + 2, J + 2));
                        stt(tidx(J + 2, J + 2), schur_diag(Tl, lane, J + 2, J) + hd);
                    }
                }
                if (r == 3) STAMP(9); else STAMP(10);
                __syncthreads();   // column J and W_{J+1} published
                STAMP(11);
            }
        };
        // ---- KKT solve on wave 0: right-hand side in xv, solution back in xv (natural order) ----
        auto solve = [&]() {
            if (wave == 0) solve_lds<NBMAX>(Tl, Wdl, xv, nb, lane);
            __syncthreads();
        };

        // ---------------- interior-point iterations ----------------
        int status = 1, nit = 0;
        bool first = true;
        bool refined = !(C.mu_refine > 0.0);
        float mu_last = 3.0e38f;
        const float inv2n = 1.0f / (float)(2 * n);
        for (int it = 0; it <= C.max_iters; ++it) {
            __syncthreads();
            const bool do_ref = __builtin_amdgcn_readfirstlane(!refined && mu_last < (float)C.mu_refine);
            const float dcur = valid ? ((sl < su) ? lo + sl : hi - su) : 0.f;
            if (do_ref) {
                // one accurate (float64, structured) gradient at the current iterate: wave 0, the others wait
                if (tid < npad) dnat[tid] = dcur;
                __syncthreads();
                if (wave == 0) {
                    if ((N + 1) * 72 <= (int)sizeof(sSl))
                        struct_grad<lds_f64*, NPAD>(C, (glb_cf64*)recg, (lds_f64*)recd, (lds_cf32*)s_Da, (lds_cf32*)dnat, (lds_f64*)sSl,
                                                   (glb_f64*)sbuf, na, lane);
                    else
                        struct_grad<glb_f64*, NPAD>(C, (glb_cf64*)recg, (lds_f64*)recd, (lds_cf32*)s_Da, (lds_cf32*)dnat,
                                                   (glb_f64*)(sbuf + NPAD + 8 * N), (glb_f64*)sbuf, na, lane);
                }
                __syncthreads();
                grad = (valid && tid < n) ? (float)(sbuf[tid] + 2.0 * C.rho * ((double)ubar + (double)dcur)) : 0.f;
                refined = true;
                STAMP(7);
            } else if (it == 0) {
                // gradient at the start point: g + H d (every tile once: -H' in register order serves both triangles).
                // Each wave sums its tiles into its own copy of the vector (the factor area is idle here) and the four
                // copies are added in a fixed order: the result does not depend on how the waves interleave.
                float* const yw = Tl + wave * NPAD;
                if (tid < npad) {
                    dnat[tid] = dcur;
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) Tl[w * NPAD + tid] = 0.f;
                }
                __syncthreads();
                for (int t = wave; t < ntl; t += NWAVE) {
                    int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                    while (tidx(I + 1, 0) <= t) ++I;
                    while (tidx(I, 0) > t) --I;
                    const int J = t - tidx(I, 0);
                    const f32x4 t4 = ldh(t);      // lane (q, col): -H[16I + col][16J + 4q + r]
                    const f32x4 d4 = lds4(dnat + 16 * J + 4 * lq);
                    const float rowp = quad_sum(t4.x * d4.x + t4.y * d4.y + t4.z * d4.z + t4.w * d4.w);
                    if (lq == 0) yw[16 * I + li] -= rowp;
                    if (I != J) {
                        const float dI = dnat[16 * I + li];
                        float c0 = t4.x * dI, c1 = t4.y * dI, c2 = t4.z * dI, c3 = t4.w * dI;
                        row_sum16x4(c0, c1, c2, c3);
                        wave_lds_fence();
                        if (li == 0) {
                            f32x4 y4 = lds4(yw + 16 * J + 4 * lq);
                            y4.x -= c0;
                            y4.y -= c1;
                            y4.z -= c2;
                            y4.w -= c3;
                            *reinterpret_cast<f32x4*>(yw + 16 * J + 4 * lq) = y4;
                        }
                    }
                    wave_lds_fence();
                }
                __syncthreads();
                if (tid < npad) yv[tid] = (Tl[tid] + Tl[NPAD + tid]) + (Tl[2 * NPAD + tid] + Tl[3 * NPAD + tid]);
                grad = valid ? yv[tid] + gv : 0.f;
                STAMP(3);
            }
            if (first) {
                const float gm = wg_reduce(valid ? fabsf(grad) : 0.f, OpMax{});
                const float wm = wg_reduce(valid ? ubv : 0.f, OpMax{});
                const float mu0 = fmaxf(0.02f * gm * wm, 1e-3f);
                zl = valid ? mu0 / sl : 0.f;
                zu = valid ? mu0 / su : 0.f;
                first = false;
            }
            const float mu = wg_reduce(valid ? sl * zl + su * zu : 0.f, OpAdd{}) * inv2n;
            mu_last = mu;
            if (__builtin_amdgcn_readfirstlane(!(mu >= mu_stop))) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            const float rsl = __builtin_amdgcn_rcpf(sl), rsu = __builtin_amdgcn_rcpf(su);
            const float Sig = valid ? zl * rsl + zu * rsu : 0.f;
            __syncthreads();
            if (tid < npad) sigv[tid] = Sig;
            __syncthreads();
            STAMP(6);
            factor();
            STAMP(4);
            if (__builtin_amdgcn_readfirstlane(s_flag) == 0) {
                status = 2;
                break;
            }
            // predictor: (H + Sig) da = -grad
            if (tid < npad) xv[tid] = -grad;
            __syncthreads();
            STAMP(6);
            solve();
            STAMP(5);
            const float da = (tid < npad && valid) ? xv[tid] : 0.f;
            float dzl_a = 0.f, dzu_a = 0.f, ap = 1.f, ad = 1.f;
            if (valid) {
                dzl_a = -zl - zl * da * rsl;
                dzu_a = -zu + zu * da * rsu;
                const float rda = __builtin_amdgcn_rcpf(da);
                if (da < 0.f) ap = fminf(ap, -sl * rda);
                if (da > 0.f) ap = fminf(ap, su * rda);
                if (dzl_a < 0.f) ad = fminf(ad, -zl * __builtin_amdgcn_rcpf(dzl_a));
                if (dzu_a < 0.f) ad = fminf(ad, -zu * __builtin_amdgcn_rcpf(dzu_a));
            }
            ap = wg_reduce(ap, OpMin{});
            ad = wg_reduce(ad, OpMin{});
            const float mu_aff = wg_reduce(valid ? (sl + ap * da) * (zl + ad * dzl_a) + (su - ap * da) * (zu + ad * dzu_a) : 0.f, OpAdd{}) * inv2n;
            float sigma = mu_aff / mu;
            sigma = fminf(fmaxf(sigma * sigma * sigma, 0.f), 1.f);
            // corrector
            float rcl = 0.f, rcu = 0.f, rhs = 0.f;
            if (valid) {
                rcl = sl * zl + da * dzl_a - sigma * mu;
                rcu = su * zu - da * dzu_a - sigma * mu;
                rhs = -(grad - zl + zu) - rcl * rsl + rcu * rsu;
            }
            __syncthreads();
            if (tid < npad) xv[tid] = rhs;
            __syncthreads();
            STAMP(6);
            solve();
            STAMP(5);
            const float dd = (tid < npad && valid) ? xv[tid] : 0.f;
            float dzl = 0.f, dzu = 0.f;
            ap = 1e30f;
            ad = 1e30f;
            if (valid) {
                dzl = (-rcl - zl * dd) * rsl;
                dzu = (-rcu + zu * dd) * rsu;
                const float rdd = __builtin_amdgcn_rcpf(dd);
                if (dd < 0.f) ap = fminf(ap, -sl * rdd);
                if (dd > 0.f) ap = fminf(ap, su * rdd);
                if (dzl < 0.f) ad = fminf(ad, -zl * __builtin_amdgcn_rcpf(dzl));
                if (dzu < 0.f) ad = fminf(ad, -zu * __builtin_amdgcn_rcpf(dzu));
            }
            ap = fminf(1.f, 0.9995f * wg_reduce(ap, OpMin{}));
            ad = fminf(1.f, 0.9995f * wg_reduce(ad, OpMin{}));
            if (valid) {
                grad += ap * (rhs - Sig * dd);   // + ap H dd
                sl += ap * dd;
                su -= ap * dd;
                zl += ad * dzl;
                zu += ad * dzu;
            }
        }
        STAMP(6);
        // ---------------- outputs ----------------
        __syncthreads();
        float* ubuf = Tl;   // N*NT <= 1024 words, zero = broken thruster
        for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.f;
        __syncthreads();
        if (valid) {
            float u = (sl < su) ? sl : ubv - su;
            if (status == 2) u = ubar;
            ubuf[kcol * NT + s_act[acol]] = u;
        }
        __syncthreads();
        if (tid < NT) P.out_u0[inst * NT + tid] = (double)ubuf[tid];
        if (P.out_U)
            for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = (double)ubuf[i];
        if (tid == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        STAMP(8);
#ifdef FTMPC_STAMPS
        if (tid == 0 && inst < 4096) {
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(P.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = st_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_wg32_kernel<15>(const DeviceConsts, const SolveWgParams);

}  // namespace ftmpc
