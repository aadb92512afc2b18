// ftmpc_solve_ws.hip -- kernel 8: the thruster-space QP of kernel 7 / kernels 2 solved THROUGH WRENCH SPACE, fp32, one 4-wave
// workgroup per instance, for vehicles with more healthy thrusters than wrench components (na > 6).
//
// The condensed Hessian of the box-constrained thruster QP is  H = DD' H_w DD + 2 rho I  with DD = blockdiag(D_a) (6N x N na)
// and H_w the 6N x 6N Hessian over the stage wrenches (oracle/qp_oracle.py: build_qp vs build_qp_wrench agree to 1e-15):
// the thrusters enter the dynamics only through D_a u.  So the Newton system of the interior-point iteration,
//     (H + Sigma) x = r,      Dg := 2 rho + Sigma  (diagonal, > 0),
// is a diagonal plus a rank-6N term.  With H_w = L L' (factorised ONCE per instance) and S = DD Dg^-1 DD' (block diagonal,
// one 6 x 6 block per stage),
//     x = Dg^-1 ( r - DD' L K^-1 L' DD Dg^-1 r ),        K = I + L' S L      (6N x 6N, eigenvalues >= 1),
// so every iteration factorises a 6N x 6N matrix (N = 15: 90 instead of 240 variables, 1/19 of the flops) that is well
// conditioned whatever the barrier weights do -- the forms with S^-1 or H_w^-1 lose the iteration when thrusters sit
// on their bounds (S loses rank) or cancel badly; this one reproduces the dense fp32 iteration count and answer
// (build/schur_experiment.py history in DESIGN.md).  A rank-deficient D_a (degenerate hull) needs no special case.
// Everything else -- the float64 linearisation records, the condensing (run with D_a = I: columns are stage wrenches), tiles in
// accumulator layout, in-register potrf + inverse, the chain/helper factorisation schedule, the gradient by recurrence with
// one float64 structured refinement, the Mehrotra iteration in thruster space -- is kernel 7's (ftmpc_solve_wg.hip).
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354; oracle/qp_oracle.py:ipm_box is the mirror.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace wsk {
constexpr int WG = 256;
constexpr int NWAVE = 4;
// thruster-space variables per thread: one up to N = 16 (six tiles a side: N * NT <= 256), two beyond (eight tiles: N * NT <= 336)
__host__ __device__ constexpr int nvt_of(int nbmax) { return nbmax <= 6 ? 1 : 2; }
using wgk::ntiles;
// per-workgroup global slot (4-byte words): float64 scratch of the reference gradient | E panels N x 9 x NPAD | Hessian tiles
__host__ __device__ constexpr int64_t slot_e_off(int nbmax, int N) { return ((wgk::slot_f64_words(WG * nvt_of(nbmax), N) + 255) / 256) * 256; }
__host__ __device__ constexpr int64_t slot_h_off(int nbmax, int N) { return slot_e_off(nbmax, N) + (int64_t)N * 9 * 16 * nbmax; }
__host__ __device__ constexpr int64_t slot_words(int nbmax, int N) { return slot_h_off(nbmax, N) + (int64_t)ntiles(nbmax) * 256; }
}  // namespace wsk

template <int NB>   // block rows of the wrench-space system: 6 N <= 16 NB
__global__ void __launch_bounds__(wsk::WG) __attribute__((amdgpu_waves_per_eu((NB <= 6) ? 2 : 1, (NB <= 6) ? 2 : 1))) ftmpc_solve_ws32_kernel(const DeviceConsts C, const SolveWgParams Q) {
    using namespace wsk;
    constexpr int NPAD = 16 * NB;
    constexpr int NT_ALL = ntiles(NB);
    constexpr int NVT = nvt_of(NB);        // thruster-space variables per thread: variable e = v * WG + tid
    constexpr int NTP = WG * NVT;
    const SolveParams& P = Q.base;
    __shared__ __attribute__((aligned(16))) float Tl[(NT_ALL + NB) * 256];   // factor of K (diagonal slot: W') | W of every diagonal block
    __shared__ __attribute__((aligned(16))) float Lu[NT_ALL * 256];           // L (H_w = L L'), tiles L_IJ in accumulator layout
    constexpr int NSTG = (16 * NB) / 6;                                        // stages: 6 N <= 16 NB
    __shared__ __attribute__((aligned(16))) float xv[NPAD], yv[NPAD], sigv[NPAD], gwv[NPAD], tw[NPAD];
    __shared__ __attribute__((aligned(16))) float rv[NTP], rdg[NTP];
    __shared__ __attribute__((aligned(16))) float Sblk[NSTG * 36];
    __shared__ __attribute__((aligned(16))) double recd[REC_STRIDE + 4];
    __shared__ __attribute__((aligned(16))) double sSl[9 * (NSTG + 1)];
    __shared__ __attribute__((aligned(16))) float recf[REC_STRIDE];
    __shared__ float S17[16 * 17];
    __shared__ __attribute__((aligned(16))) float s_Da[6 * MAX_NT], s_DaT[6 * MAX_NT];   // identity (condensing) | the healthy columns of D
    __shared__ float s_MR[MAX_NT * MAX_NT];
    __shared__ float red[NWAVE];
    __shared__ unsigned char s_stg[NPAD], s_thr[NPAD];
    __shared__ int s_act[MAX_NT];
    __shared__ int s_flag, s_q;
    __shared__ float s_D[6 * MAX_NT];
    __shared__ unsigned char tIJ[2 * NT_ALL];                                 // block row / column of tile t
    // S = DD Dg^-1 DD' as tiles: diagonal D | sub-diagonal (D+1, D) | super-diagonal (D, D+1), rebuilt every iteration
    __shared__ __attribute__((aligned(16))) float Stl[(3 * NB - 2) * 256];
    __shared__ __attribute__((aligned(16))) float s_DD[21 * MAX_NT];          // D_a[g][a] D_a[h][a] for the 21 pairs g >= h
    float* const Wdl = Tl + NT_ALL * 256;
    float* const dnat = tw;     // start gradient only
    float* const dT = rv;       // reference gradient only
    static_assert((3 * NB - 2) * 256 >= NWAVE * 16 * 17, "transposition scratch lives in the S tiles");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int N = C.N, NT = C.NT;
    const float rho = (float)C.rho;
    const float mu_stop = (float)C.mu_stop;
    float* const slot = Q.slot + (int64_t)blockIdx.x * Q.slot_words;
    double* const sbuf = reinterpret_cast<double*>(slot);
    float* const Eall = slot + slot_e_off(NB, N);
    float* const Hs = slot + slot_h_off(NB, N);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto ldt = [&](int t) { return *reinterpret_cast<const f32x4*>(Tl + t * 256 + 4 * lane); };
    auto stt = [&](int t, f32x4 v) { *reinterpret_cast<f32x4*>(Tl + t * 256 + 4 * lane) = v; };
    auto ldh = [&](int t) { return *reinterpret_cast<const f32x4*>(Hs + (int64_t)t * 256 + 4 * lane); };
    auto ldl = [&](int t) { return *reinterpret_cast<const f32x4*>(Lu + t * 256 + 4 * lane); };
    auto wg_reduce = [&](float x, auto op) {
        x = wave_reduce<decltype(op)>(x);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        return decltype(op)::f(decltype(op)::f(red[0], red[1]), decltype(op)::f(red[2], red[3]));
    };
    auto tile_of = [&](int t, int& I, int& J) {
        I = tIJ[2 * t];
        J = tIJ[2 * t + 1];
    };
    {
        double dsel = 0.0;
#pragma unroll
        for (int i = 0; i < 6 * MAX_NT; ++i) dsel = (tid == i) ? C.D[i] : dsel;
        if (tid < 6 * MAX_NT) s_D[tid] = (float)dsel;
    }
    for (int t = tid; t < NT_ALL; t += WG) {
        int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (tidx(I + 1, 0) <= t) ++I;
        while (tidx(I, 0) > t) --I;
        tIJ[2 * t] = (unsigned char)I;
        tIJ[2 * t + 1] = (unsigned char)(t - tidx(I, 0));
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
        if (wave == 0) {   // the healthy thrusters in order: one load per lane, ranks from the ballot
            const double ubl = (lane < NT) ? P.ub[inst * NT + lane] : 0.0;
            const unsigned long long m = __ballot(ubl > 0.0);
            if (ubl > 0.0) s_act[__popcll(m & ((1ull << lane) - 1ull))] = lane;
            if (lane == 0) s_flag = __popcll(m);
        }
        __syncthreads();
        const int nat = __builtin_amdgcn_readfirstlane(s_flag);   // healthy thrusters
        const int nt = N * nat;                                    // thruster-space variables
        constexpr int na = 6;                                      // wrench components: the columns of the condensing
        const int n = N * na;
        const int nb = (n + 15) >> 4;
        const int npad = nb * 16;
        if (nat == 0 || nb > NB || nt > NTP) {
            for (int i = tid; i < NT; i += WG) P.out_u0[inst * NT + i] = 0.0;
            if (P.out_U)
                for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = 0.0;
            if (tid == 0) {
                if (P.status) P.status[inst] = (nat == 0) ? 0 : 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        if (tid < 6 * MAX_NT) {
            const int g = tid / MAX_NT, a = tid % MAX_NT;
            s_DaT[tid] = (a < nat) ? s_D[g * MAX_NT + s_act[a]] : 0.f;
            s_Da[tid] = (a == g) ? 1.f : 0.f;
        }
        if (tid < npad) {
            const int s = tid / na;
            s_stg[tid] = (unsigned char)(tid < n ? s : 255);
            s_thr[tid] = (unsigned char)(tid < n ? tid - s * na : 255);
        }
        if (tid < MAX_NT * MAX_NT) {   // stage block of H_w: 2 R (the rho term stays in thruster space); constant indices into C
            const int a = tid / MAX_NT, b = tid % MAX_NT;
            float r = 0.f;
#pragma unroll
            for (int g = 0; g < 6; ++g) r = (a == g) ? (float)C.R[g] : r;
            s_MR[tid] = (a == b && a < 6) ? 2.f * r : 0.f;
        }
        __syncthreads();
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;
        // wrench-space role of this thread (condensing, tiles): column (stage kcol, component acol)
        const int kcol = (tid < npad) ? s_stg[tid] : 255;
        const int acol = (tid < npad) ? s_thr[tid] : 255;
        const bool valid = kcol != 255;
        float gacc = 0.f;
        float Fd[3] = {0.f, 0.f, 0.f}, Td[3] = {0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                Fd[a] = s_Da[a * MAX_NT + acol];
                Td[a] = s_Da[(3 + a) * MAX_NT + acol];
            }
        }
        // thruster-space role: variables e = v * WG + tid = (stage tk, healthy thruster ta)
        bool tvalid[NVT];
        int tk[NVT], ta[NVT];
        float ubar[NVT], ubv[NVT];
        float dat[NVT][6];    // column ta of D_a
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            const int e = v * WG + tid;
            tvalid[v] = e < nt;
            tk[v] = tvalid[v] ? e / nat : 0;
            ta[v] = tvalid[v] ? e - tk[v] * nat : 0;
            ubar[v] = 0.f;
            ubv[v] = 1.f;
            if (tvalid[v]) {
                const int t = s_act[ta[v]];
                ubv[v] = (float)P.ub[inst * NT + t];
                if (P.warmU) ubar[v] = fminf(fmaxf((float)P.warmU[(inst * N + tk[v]) * NT + t], 0.f), ubv[v]);
            }
#pragma unroll
            for (int g = 0; g < 6; ++g) dat[v][g] = tvalid[v] ? s_DaT[g * MAX_NT + ta[v]] : 0.f;
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
        double rnext = (tid < REC_STRIDE) ? recg[tid] : 0.0;   // the record of the next stage is requested a stage ahead
        for (int k = 0; k < N; ++k) {
            __syncthreads();
            if (tid < REC_STRIDE) recf[tid] = (float)rnext;
            if (tid < REC_STRIDE && k + 1 < N) rnext = recg[(k + 1) * REC_STRIDE + tid];
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
            f32x4 acc = zero4, acc2 = zero4;
            const int kstart = (16 * I) / na < N ? (16 * I) / na : N;
            int k = kstart;
            for (; k + 1 < N; k += 2) {       // two stages per trip: twelve loads in flight, two accumulator chains
                const float* Ek = Eb + (int64_t)k * 9 * npad;
                const float* En = Ek + 9 * npad;
                float a[3], b[3], c[3], d[3];
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int r = 4 * s3 + lq;
                    a[s3] = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.f;   // A[m][k] = E[r][16J + m]
                    b[s3] = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.f;   // B[k][n] = E[r][16I + n]
                    c[s3] = (r < 9) ? En[r * npad + 16 * J + li] : 0.f;
                    d[s3] = (r < 9) ? En[r * npad + 16 * I + li] : 0.f;
                }
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    acc = mfma4(a[s3], b[s3], acc);                               // (E_J' E_I) = (H_IJ)'
                    acc2 = mfma4(c[s3], d[s3], acc2);
                }
            }
            for (; k < N; ++k) {
                const float* Ek = Eb + (int64_t)k * 9 * npad;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3) {
                    const int r = 4 * s3 + lq;
                    const float a = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.f;
                    const float b = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.f;
                    acc = mfma4(a, b, acc);
                }
            }
            acc += acc2;
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
        if ((int64_t)N * 9 * npad <= (int64_t)(NT_ALL + NB) * 256) build(std::true_type{});
        else build(std::false_type{});
        if (tid < npad) gwv[tid] = valid ? 2.f * gacc : 0.f;     // wrench-space gradient at the linearisation point
        float lo[NVT], hi[NVT], sl[NVT], su[NVT], zl[NVT], zu[NVT], grad[NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            lo[v] = -ubar[v];
            hi[v] = ubv[v] - ubar[v];
            sl[v] = su[v] = 0.5f * ubv[v];
            zl[v] = zu[v] = grad[v] = 0.f;
        }
        bool keep_l = true, k_in_lds = false;

        // the matrix to factorise comes as -M' tiles: H_w from the global slot (once), K = I + L' S L from the factor's own LDS
        // slots, where the assembly left it (read before the slot is overwritten with the partial sum or the tile)
        auto ldk = [&](int t) { return k_in_lds ? ldt(t) : ldh(t); };
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
                // first factorisation (of H_w itself): keep the diagonal block of the factor, L_DD = C W' (C = L L', W = L^-1).
                // (A plain Cholesky of C by rows instead -- no cond(C) eps in L_DD -- was tried: same answers, 2 % slower.)
                if (keep_l) *reinterpret_cast<f32x4*>(Lu + tidx(D, D) * 256 + 4 * lane) = mm_tn(cd, wt, zero4);
            };
            if (wave == 0) potrf_publish(0, ldk(tidx(0, 0)));
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
                        part = ldk(tidx(1, 0));
                        dsum = ldk(tidx(1, 1));
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
                                pa = ldk(tidx(I, 0));
                                pb = ldk(tidx(I2, 0));
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
                                pa = ldk(tidx(I, 0));
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
                            const f32x4 ha = ldk(tidx(I, J + 1)), hb = ldk(tidx(I2, J + 1));
                            f32x4 a0 = zero4, a1 = zero4, b0 = zero4, b1 = zero4;
                            for (int K = 0; K < J; ++K) {
                                const f32x4 tj = lds4(pj + K * 256), ta = lds4(pa + K * 256), tb = lds4(pb + K * 256);
                                mm_tn2(tj, ta, a0, a1);
                                mm_tn2(tj, tb, b0, b1);
                            }
                            stt(tidx(I, J + 1), a0 + a1 + ha);
                            stt(tidx(I2, J + 1), b0 + b1 + hb);
                        } else {
                            const f32x4 ha = ldk(tidx(I, J + 1));
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
                        const f32x4 hd = ldk(tidx(J + 2, J + 2));
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
            if (wave == 0) solve_lds<NB>(Tl, Wdl, xv, nb, lane);
            __syncthreads();
        };
        // ---- wrench image of a thruster-space vector held one per thread: tw = DD v ----
        auto to_wrench = [&](const float (&x)[NVT], float* out) {
#pragma unroll
            for (int v = 0; v < NVT; ++v) rv[v * WG + tid] = tvalid[v] ? x[v] : 0.f;
            __syncthreads();
            if (tid < npad) {
                float s = 0.f;
                if (valid)
                    for (int a = 0; a < nat; ++a) s += s_DaT[acol * MAX_NT + a] * rv[kcol * nat + a];
                out[tid] = s;
            }
            __syncthreads();
        };
        // ---- out = L' in (TRANS) or L in: the output blocks are dealt over the waves, each sums its tiles in registers ----
        auto tri_mv = [&](auto TRANS, const float* in, float* out) {
            for (int Bo = wave; Bo < nb; Bo += NWAVE) {
                if constexpr (decltype(TRANS)::value) {   // (L' in)_J = sum_{I >= J} L_IJ' in_I
                    float acc = 0.f;
                    for (int I = Bo; I < nb; ++I) {
                        const f32x4 l4 = ldl(tidx(I, Bo));            // lane (q, col): L_IJ[4q + r][col]
                        const f32x4 v4 = lds4(in + 16 * I + 4 * lq);
                        acc += l4.x * v4.x + l4.y * v4.y + l4.z * v4.z + l4.w * v4.w;
                    }
                    acc = quad_sum(acc);
                    if (lq == 0) out[16 * Bo + li] = acc;
                } else {                                  // (L in)_I = sum_{J <= I} L_IJ in_J
                    float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
                    for (int J = 0; J <= Bo; ++J) {
                        const f32x4 l4 = ldl(tidx(Bo, J));
                        const float vj = in[16 * J + li];
                        c0 += l4.x * vj;
                        c1 += l4.y * vj;
                        c2 += l4.z * vj;
                        c3 += l4.w * vj;
                    }
                    row_sum16x4(c0, c1, c2, c3);
                    if (li == 0) *reinterpret_cast<f32x4*>(out + 16 * Bo + 4 * lq) = f32x4{c0, c1, c2, c3};
                }
            }
            __syncthreads();
        };
        // ---- Newton system in thruster space through wrench space: x = Dg^-1 (r - DD' L K^-1 L' DD Dg^-1 r) ----
        auto ws_solve = [&](const float (&r)[NVT], float (&x)[NVT]) {
            float t[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) t[v] = r[v] * rdg[v * WG + tid];
            to_wrench(t, tw);
            tri_mv(std::true_type{}, tw, xv);
            solve();
            tri_mv(std::false_type{}, xv, yv);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < 6; ++g) s += dat[v][g] * yv[tk[v] * 6 + g];
                x[v] = tvalid[v] ? (r[v] - s) * rdg[v * WG + tid] : 0.f;
            }
        };

        // ---------------- H_w = L L' once: factor, keep L (transposed back out of the factor's tile order) ----------------
        if (tid < npad) sigv[tid] = 0.f;
        __syncthreads();
        factor();
        keep_l = false;
        k_in_lds = true;
        if (__builtin_amdgcn_readfirstlane(s_flag) == 0) {   // H_w not positive definite in fp32: report the linearisation point
            // (FTMPC_STATUS_NUMERIC, include/ftmpc.h: clip(warm start); zero at broken thrusters), do not iterate
            __syncthreads();
            float* const ub0 = Tl;
            for (int i = tid; i < N * NT; i += WG) ub0[i] = 0.f;
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) ub0[tk[v] * NT + s_act[ta[v]]] = ubar[v];
            __syncthreads();
            for (int i = tid; i < NT; i += WG) P.out_u0[inst * NT + i] = (double)ub0[i];
            if (P.out_U)
                for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = (double)ub0[i];
            if (tid == 0) {
                if (P.status) P.status[inst] = 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        for (int t = wave; t < ntl; t += NWAVE) {
            int I, J;
            tile_of(t, I, J);
            if (I == J) continue;                     // diagonal blocks: written by potrf_publish
            const f32x4 tt = ldt(t);                  // L_IJ'
            float* sc = Stl + wave * 16 * 17;   // (the S tiles are not in use yet)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) sc[(4 * lq + rr) * 17 + li] = tt[rr];
            wave_lds_fence();
            f32x4 lt;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) lt[rr] = sc[li * 17 + 4 * lq + rr];
            wave_lds_fence();
            *reinterpret_cast<f32x4*>(Lu + t * 256 + 4 * lane) = lt;
        }
        __syncthreads();
        if (tid < npad) sigv[tid] = 1.f;                 // the identity of K = I + L' S L
        // S tile (rows of block K, columns of block M, |K - M| <= 1)
        auto s_tile = [&](int K, int M) { return (K == M) ? M : (K > M ? NB + M : 2 * NB - 1 + K); };
        auto lds_S = [&](int K, int M) { return *reinterpret_cast<const f32x4*>(Stl + s_tile(K, M) * 256 + 4 * lane); };
        for (int e = tid; e < 21 * MAX_NT; e += WG) {   // products of the rows of D_a, pair p = g (g + 1) / 2 + h, zero beyond the healthy thrusters
            const int p = e / MAX_NT, a = e % MAX_NT;
            int g = 0;
            while ((g + 1) * (g + 2) / 2 <= p) ++g;
            const int hh = p - g * (g + 1) / 2;
            s_DD[e] = (a < nat) ? s_DaT[g * MAX_NT + a] * s_DaT[hh * MAX_NT + a] : 0.f;
        }
        __syncthreads();

        // ---------------- interior-point iterations (thruster space) ----------------
        int status = 1, nit = 0;
        bool first = true;
        // float64 reference gradient: once, when mu falls below mu_refine, for the six-tile instantiation (N <= 16); at EVERY
        // iterate below mu_refine for the eight-tile one.  The gradient recurrence trusts the Newton identity more than
        // K^-1 in fp32 deserves on long horizons with ill-conditioned allocation matrices (3e-4 .. 3e-3 f_max on synthetic
        // vehicles with cond(H) ~ 1e7); refreshed every iteration the worst instance stays below 1e-5 (kernel 8).
        int refines_left = (C.mu_refine > 0.0) ? ((NB > 6) ? C.max_iters + 1 : 1) : 0;
        float mu_last = 3.0e38f;
        const float inv2n = 1.0f / (float)(2 * nt);
        for (int it = 0; it <= C.max_iters; ++it) {
            __syncthreads();
            const bool do_ref = __builtin_amdgcn_readfirstlane(refines_left > 0 && mu_last < (float)C.mu_refine);
            float dcur[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) dcur[v] = tvalid[v] ? ((sl[v] < su[v]) ? lo[v] + sl[v] : hi[v] - su[v]) : 0.f;
            if (do_ref) {   // float64, structured, at the current iterate: wave 0, the others wait
                __syncthreads();
#pragma unroll
                for (int v = 0; v < NVT; ++v) dT[v * WG + tid] = dcur[v];
                __syncthreads();
                if (wave == 0) {
                    if ((N + 1) * 72 <= (int)sizeof(sSl))
                        struct_grad<lds_f64*, NTP>(C, (glb_cf64*)recg, (lds_f64*)recd, (lds_cf32*)s_DaT, (lds_cf32*)dT, (lds_f64*)sSl,
                                                  (glb_f64*)sbuf, nat, lane);
                    else
                        struct_grad<glb_f64*, NTP>(C, (glb_cf64*)recg, (lds_f64*)recd, (lds_cf32*)s_DaT, (lds_cf32*)dT,
                                                  (glb_f64*)(sbuf + NTP + 8 * N), (glb_f64*)sbuf, nat, lane);
                }
                __syncthreads();
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    grad[v] = tvalid[v] ? (float)(sbuf[v * WG + tid] + 2.0 * C.rho * ((double)ubar[v] + (double)dcur[v])) : 0.f;
                --refines_left;
                STAMP(7);
            } else if (it == 0) {
                // gradient at the start point: DD' (g_w + H_w DD d) + 2 rho (ubar + d); the wrench-space product from the -H_w' tiles
                to_wrench(dcur, dnat);
                float* const yw = Tl + wave * NPAD;   // (the factor area is idle here)
                if (tid < npad) {
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) Tl[w * NPAD + tid] = 0.f;
                }
                __syncthreads();
                for (int t = wave; t < ntl; t += NWAVE) {
                    int I, J;
                    tile_of(t, I, J);
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
                if (tid < npad) yv[tid] = (Tl[tid] + Tl[NPAD + tid]) + (Tl[2 * NPAD + tid] + Tl[3 * NPAD + tid]) + gwv[tid];
                __syncthreads();
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    float s = 0.f;
#pragma unroll
                    for (int g = 0; g < 6; ++g) s += dat[v][g] * yv[tk[v] * 6 + g];
                    grad[v] = tvalid[v] ? s + 2.f * rho * (ubar[v] + dcur[v]) : 0.f;
                }
                STAMP(3);
            }
            if (first) {
                float gm = 0.f, wm = 0.f;
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    gm = fmaxf(gm, tvalid[v] ? fabsf(grad[v]) : 0.f);
                    wm = fmaxf(wm, tvalid[v] ? ubv[v] : 0.f);
                }
                gm = wg_reduce(gm, OpMax{});
                wm = wg_reduce(wm, OpMax{});
                const float mu0 = fmaxf(0.02f * gm * wm, 1e-3f);
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    zl[v] = tvalid[v] ? mu0 / sl[v] : 0.f;
                    zu[v] = tvalid[v] ? mu0 / su[v] : 0.f;
                }
                first = false;
            }
            float csum = 0.f;
#pragma unroll
            for (int v = 0; v < NVT; ++v) csum += tvalid[v] ? sl[v] * zl[v] + su[v] * zu[v] : 0.f;
            const float mu = wg_reduce(csum, OpAdd{}) * inv2n;
            mu_last = mu;
            if (__builtin_amdgcn_readfirstlane(!(mu >= mu_stop))) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            float rsl[NVT], rsu[NVT], Sig[NVT];
            // ---- K = I + L' S L: S blocks, P = S L (into the idle factor area), X' = P' L (as -X' into the Hessian slot) ----
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                rsl[v] = __builtin_amdgcn_rcpf(sl[v]);
                rsu[v] = __builtin_amdgcn_rcpf(su[v]);
                Sig[v] = tvalid[v] ? zl[v] * rsl[v] + zu[v] * rsu[v] : 0.f;
                rdg[v * WG + tid] = tvalid[v] ? 1.0f / (2.f * rho + Sig[v]) : 0.f;
            }
            __syncthreads();
            for (int idx = tid; idx < N * 21; idx += WG) {          // stage blocks S_k = D_a diag(1 / Dg) D_a'
                const int k = idx / 21, p = idx - 21 * k;
                int g = 0;
                while ((g + 1) * (g + 2) / 2 <= p) ++g;
                const int hh = p - g * (g + 1) / 2;
                float sacc = 0.f;
#pragma unroll
                for (int a = 0; a < MAX_NT; ++a) sacc += s_DD[p * MAX_NT + a] * rdg[(k * nat + a) & (NTP - 1)];
                Sblk[k * 36 + g * 6 + hh] = sacc;
                Sblk[k * 36 + hh * 6 + g] = sacc;
            }
            __syncthreads();
#ifdef FTMPC_STAMPS_FINE
            STAMP(3);
#endif
            for (int t = wave; t < 3 * nb - 2; t += NWAVE) {        // the tiles of S
                const int D = t < nb ? t : (t < 2 * nb - 1 ? t - nb : t - (2 * nb - 1));
                const int Kr = t < nb ? D : (t < 2 * nb - 1 ? D + 1 : D);
                const int Mc = t < nb ? D : (t < 2 * nb - 1 ? D : D + 1);
                const int e2 = 16 * Mc + li;
                const int s2 = s_stg[e2], a2 = s_thr[e2];
                f32x4 s4;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int e1 = 16 * Kr + 4 * lq + rr;
                    const int s1 = s_stg[e1];
                    s4[rr] = (s1 != 255 && s1 == s2) ? Sblk[s1 * 36 + s_thr[e1] * 6 + a2] : 0.f;
                }
                *reinterpret_cast<f32x4*>(Stl + s_tile(Kr, Mc) * 256 + 4 * lane) = s4;
            }
            __syncthreads();
            for (int t = wave; t < ntl; t += NWAVE) {     // P_MJ = sum_K S_MK L_KJ, K in {M-1, M, M+1}, K >= J
                int M, J;
                tile_of(t, M, J);
                f32x4 a0 = zero4, a1 = zero4;
                for (int K = (M - 1 > J ? M - 1 : J); K <= M + 1 && K < nb; ++K) mm_tn2(lds_S(K, M), ldl(tidx(K, J)), a0, a1);
                stt(t, a0 + a1);
            }
            __syncthreads();
#ifdef FTMPC_STAMPS_FINE
            STAMP(8);
#endif
            {   // X_IJ' = sum_{M >= I} P_MJ' L_MI, held in registers until every wave is done with P, then -X' into the factor's slots
                constexpr int XPW = (NT_ALL + NWAVE - 1) / NWAVE;
                f32x4 xt[XPW];
#pragma unroll
                for (int i = 0; i < XPW; ++i) {
                    const int t = wave + NWAVE * i;
                    xt[i] = zero4;
                    if (t < ntl) {
                        int I, J;
                        tile_of(t, I, J);
                        f32x4 a0 = zero4, a1 = zero4;
                        for (int M = I; M < nb; ++M) mm_tn2(ldt(tidx(M, J)), ldl(tidx(M, I)), a0, a1);
                        xt[i] = -(a0 + a1);
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < XPW; ++i) {
                    const int t = wave + NWAVE * i;
                    if (t < ntl) stt(t, xt[i]);
                }
            }
            __syncthreads();
#ifdef FTMPC_STAMPS_FINE
            STAMP(2);
#else
            STAMP(6);
#endif
            factor();
            STAMP(4);
            if (__builtin_amdgcn_readfirstlane(s_flag) == 0) {
                status = 2;
                break;
            }
            // predictor: (H + Sig) da = -grad
            STAMP(6);
            float ngrad[NVT], da[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) ngrad[v] = -grad[v];
            ws_solve(ngrad, da);
            STAMP(5);
            float dzl_a[NVT], dzu_a[NVT], ap = 1.f, ad = 1.f;
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                dzl_a[v] = dzu_a[v] = 0.f;
                if (tvalid[v]) {
                    dzl_a[v] = -zl[v] - zl[v] * da[v] * rsl[v];
                    dzu_a[v] = -zu[v] + zu[v] * da[v] * rsu[v];
                    const float rda = __builtin_amdgcn_rcpf(da[v]);
                    if (da[v] < 0.f) ap = fminf(ap, -sl[v] * rda);
                    if (da[v] > 0.f) ap = fminf(ap, su[v] * rda);
                    if (dzl_a[v] < 0.f) ad = fminf(ad, -zl[v] * __builtin_amdgcn_rcpf(dzl_a[v]));
                    if (dzu_a[v] < 0.f) ad = fminf(ad, -zu[v] * __builtin_amdgcn_rcpf(dzu_a[v]));
                }
            }
            ap = wg_reduce(ap, OpMin{});
            ad = wg_reduce(ad, OpMin{});
            csum = 0.f;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                csum += tvalid[v] ? (sl[v] + ap * da[v]) * (zl[v] + ad * dzl_a[v]) + (su[v] - ap * da[v]) * (zu[v] + ad * dzu_a[v]) : 0.f;
            const float mu_aff = wg_reduce(csum, OpAdd{}) * inv2n;
            float sigma = mu_aff / mu;
            sigma = fminf(fmaxf(sigma * sigma * sigma, 0.f), 1.f);
            // corrector
            float rcl[NVT], rcu[NVT], rhs[NVT], dd[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                rcl[v] = rcu[v] = rhs[v] = 0.f;
                if (tvalid[v]) {
                    rcl[v] = sl[v] * zl[v] + da[v] * dzl_a[v] - sigma * mu;
                    rcu[v] = su[v] * zu[v] - da[v] * dzu_a[v] - sigma * mu;
                    rhs[v] = -(grad[v] - zl[v] + zu[v]) - rcl[v] * rsl[v] + rcu[v] * rsu[v];
                }
            }
            STAMP(6);
            ws_solve(rhs, dd);
            STAMP(5);
            float dzl[NVT], dzu[NVT];
            ap = 1e30f;
            ad = 1e30f;
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                dzl[v] = dzu[v] = 0.f;
                if (tvalid[v]) {
                    dzl[v] = (-rcl[v] - zl[v] * dd[v]) * rsl[v];
                    dzu[v] = (-rcu[v] + zu[v] * dd[v]) * rsu[v];
                    const float rdd = __builtin_amdgcn_rcpf(dd[v]);
                    if (dd[v] < 0.f) ap = fminf(ap, -sl[v] * rdd);
                    if (dd[v] > 0.f) ap = fminf(ap, su[v] * rdd);
                    if (dzl[v] < 0.f) ad = fminf(ad, -zl[v] * __builtin_amdgcn_rcpf(dzl[v]));
                    if (dzu[v] < 0.f) ad = fminf(ad, -zu[v] * __builtin_amdgcn_rcpf(dzu[v]));
                }
            }
            ap = fminf(1.f, 0.9995f * wg_reduce(ap, OpMin{}));
            ad = fminf(1.f, 0.9995f * wg_reduce(ad, OpMin{}));
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) {
                    grad[v] += ap * (rhs[v] - Sig[v] * dd[v]);   // + ap H dd
                    sl[v] += ap * dd[v];
                    su[v] -= ap * dd[v];
                    zl[v] += ad * dzl[v];
                    zu[v] += ad * dzu[v];
                }
        }
        STAMP(6);
        // ---------------- outputs ----------------
        __syncthreads();
        float* ubuf = Tl;   // N*NT <= 1024 words, zero = broken thruster
        for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NVT; ++v)
            if (tvalid[v]) {
                float u = (sl[v] < su[v]) ? sl[v] : ubv[v] - su[v];
                if (status == 2) u = ubar[v];
                ubuf[tk[v] * NT + s_act[ta[v]]] = u;
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

template __global__ void ftmpc_solve_ws32_kernel<6>(const DeviceConsts, const SolveWgParams);
template __global__ void ftmpc_solve_ws32_kernel<8>(const DeviceConsts, const SolveWgParams);

}  // namespace ftmpc
