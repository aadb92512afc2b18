// ftmpc_solve_hull.hip -- kernel 11: the reference's OWN formulation on ONE WAVE per instance, fp32.
//
// The reference's controller (spiraling_mpc.py:87-238) optimises the 6-D generalized force of every stage subject to the
// input hull A tau_k <= b (:133-137,175-177; input_bounds.py:43-76) and hands tau_0 to the allocator.  Per instance that is
//     min  1/2 d' H_w d + g_w' d     s.t.  A (ubar_k + d_k) <= b  for every stage k,      tau = ubar + d,  6 N variables
// with m = N * hull_rows inequality rows (26 per stage for every fault set of the 16-thruster vehicle).  The Newton matrix
// of the interior-point iteration,  H_w + blockdiag_k (A' diag(z_k / s_k) A),  is H_w plus one 6 x 6 block per stage --
// the very shape kernel 10 (ftmpc_solve_wsw.hip) assembles its S from -- so this kernel is kernel 10's build (condensing
// with D_a = I on the matrix cores), kernel 2's register-resident Cholesky and sweeps (chol_reg / solve_reg), and a Mehrotra
// iteration over the hull rows (eight or eleven rows per lane) instead of the thruster boxes:
//   * stage blocks  G_k = sum_r w_kr a_r a_r'  in FLOAT64 (products of the fp32 normals are exact there), and the
//     factorisation itself in float64 on the matrix cores (v_mfma_f64_16x16x4, the same register-resident left-looking scheme
//     as chol_reg with f64k::potrf_inv16_lds on the diagonal tiles): near the solution the active rows carry weights
//     z / s ~ 1e7, G is a huge low-rank term that is NOT diagonal, and eliminating through it in fp32 leaves noise of
//     1e-7 |G| ~ 1 on top of H_w's entries (measured: 4 % of the instances diverge, median error 2e-4 f_max, with an fp32
//     factorisation).  The box rows of kernels 2 / 10 do not have this problem -- their barrier term is diagonal.
//     The factor is rounded to fp32 afterwards (its entries are benign) and the sweeps run in fp32 (solve_reg);
//   * seeds of the factorisation = the -H_w' tiles in LDS (fp32, exact in float64) minus the block entries (diagonal and
//     first sub-diagonal tiles only), through per-lane offsets fixed per launch;
//   * C x and C' t as short LDS mat-vecs (6 terms per row, hull_rows terms per wrench component);
//   * gradient by recurrence through the Newton identity, refreshed once (N <= 16) or at every late iterate by the float64
//     structured gradient (struct_grad with D_a = I), as kernels 2 / 8 / 10.
// Same iteration as the float64 kernel's hull mode (ftmpc_solve_f64.hip, MODE 1; oracle/qp_oracle.py:ipm_general is the
// mirror): start at the hull centre D (ub / 2 + stuck), slacks b - A centre, duals mu0 / s.
// The terminal set (72 rows on x_N: a rank-9 dense term) stays on the float64 kernel.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

struct SolveHullParams {
    SolveParams base;          // rec, ub, stuck, status, iters, hscratch / tile_words (float64 scratch of struct_grad), qhead (cursor)
    const double* warmG;       // [B*N*6] previous wrench solution (already shifted) or nullptr: linearise about D stuck
    const double* hullA;       // [n_sets][hull_rows*6] facet normals
    const int32_t* hull_set;   // [B] table number or nullptr (table 0)
    const double* hullb;       // [B*hull_rows] facet offsets
    int32_t hull_rows;
    double* out_tau0;          // [B*6]
    double* out_G;             // [B*N*6] or nullptr
    // terminal set (template TSET): rows term_A (e_N + GN d) <= term_b on the terminal tracking error (spiraling_mpc.py:199-202)
    const double* termA;       // [term_rows*9]
    const double* termb;       // [term_rows]
    const double* eN;          // [B*9] terminal tracking error at the linearisation point (ftmpc_linearize.hip)
    int32_t term_rows;         // <= 80
    // Instances this kernel does not certify -- a weakly active row at termination (z / s within 1e-3 .. 1e3: there an
    // interior-point iterate at mu 1e-10 is up to 7e-5 f_max from the exact solution, and fp32 cannot run the active-set polish
    // that removes it), or hull and terminal rows active together (nearly degenerate problems: rounding their rows to fp32
    // alone moves the solution by 2-3e-5 f_max) -- are appended here and solved again by the float64 kernel with its polish,
    // which the host enqueues behind this one on the same stream.  nullptr: nothing is handed over.
    int32_t* fb_list;          // [B]
    int32_t* fb_count;         // [1], zeroed by the host
};

namespace hullk {
__host__ __device__ constexpr int nvc_of(int nbw) { return (16 * nbw / 6) * 32 / 64; }      // hull rows per lane (32 per stage slot)

__device__ __forceinline__ f64x4 mm_tn64(const f64x4& X, const f64x4& Y, f64x4 acc) {      // X' Y: both operands ARE accumulator-layout registers
    acc = mfma_d(X.x, Y.x, acc);
    acc = mfma_d(X.y, Y.y, acc);
    acc = mfma_d(X.z, Y.z, acc);
    acc = mfma_d(X.w, Y.w, acc);
    return acc;
}

// One block column of the float64 register-resident factorisation (chol_reg_col's scheme: T[tidx(I,J)] = L_IJ' for I > J,
// T[tidx(J,J)] = W_J', Wd[J] = W_J = L_JJ^-1), tiles in the float64 MFMA accumulator layout (lane (q, col): rows q + 4 s of
// column col).  seed(I, J): the tile of -(M_IJ)' in that layout.  S: 16 x 17 doubles (transpose scratch) + 32 (pivot column | row).
template <int NB, int J, class Seed>
__device__ __forceinline__ void chol64_col(const Seed& seed, double* S, int lq, int li, bool& ok, f64x4 (&T)[NB * (NB + 1) / 2], f64x4 (&Wd)[NB]) {
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    f64x4 a0 = seed(J, J), a1 = zero;
    f64x4 bacc[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I) bacc[I] = (I > J) ? seed(I, J) : zero;
#pragma unroll
    for (int K = 0; K < J; ++K) {
        if (K & 1) a1 = mm_tn64(T[tidx(J, K)], T[tidx(J, K)], a1);
        else a0 = mm_tn64(T[tidx(J, K)], T[tidx(J, K)], a0);
#pragma unroll
        for (int I = J + 1; I < NB; ++I) bacc[I] = mm_tn64(T[tidx(J, K)], T[tidx(I, K)], bacc[I]);
    }
    double c[4], w[4], l[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) c[rr] = -(a0[rr] + a1[rr]);      // M_JJ - sum_K L_JK L_JK'
    ok = f64k::potrf_inv16_lds(c, S + 272, S + 288, lq, li, w, l) && ok;
    Wd[J] = f64x4{w[0], w[1], w[2], w[3]};
    wave_lds_fence();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) S[(lq + 4 * rr) * 17 + li] = w[rr];
    wave_lds_fence();
    f64x4 wt, wtn;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        wt[rr] = S[li * 17 + lq + 4 * rr];
        wtn[rr] = -wt[rr];
    }
    wave_lds_fence();
    T[tidx(J, J)] = wt;
#pragma unroll
    for (int I = J + 1; I < NB; ++I) T[tidx(I, J)] = mm_tn64(wtn, bacc[I], zero);      // L_IJ' = W_J (M_IJ' - sum) = -W_J bacc
    if constexpr (J + 1 < NB) chol64_col<NB, J + 1, Seed>(seed, S, lq, li, ok, T, Wd);
}

// an fp32 tile in the registers of a float64 tile (no instruction: a register pair is two registers)
__device__ __forceinline__ f64x4 park32(const f32x4& v) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    return f64x4{__builtin_bit_cast(double, f32x2_{v.x, v.y}), __builtin_bit_cast(double, f32x2_{v.z, v.w}), 0.0, 0.0};
}
__device__ __forceinline__ f32x4 unpark32(const f64x4& v) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ a = __builtin_bit_cast(f32x2_, v.x), b = __builtin_bit_cast(f32x2_, v.y);
    return f32x4{a.x, a.y, b.x, b.y};
}

// sum over the 16 lanes of a DPP row (one row group), float64: the two halves travel separately
__device__ __forceinline__ double row_sum16_d(double x) {
#define FTMPC_DPP_ADD_D(ctrl)                                                                                                        \
    {                                                                                                                                \
        const unsigned long long b_ = __builtin_bit_cast(unsigned long long, x);                                                     \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b_, ctrl, 0xf, 0xf, false);                     \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b_ >> 32), ctrl, 0xf, 0xf, false);             \
        x += __builtin_bit_cast(double, ((unsigned long long)hi_ << 32) | lo_);                                                      \
    }
    FTMPC_DPP_ADD_D(0x128)      // row_ror:8
    FTMPC_DPP_ADD_D(0x124)      // row_ror:4
    FTMPC_DPP_ADD_D(0x122)      // row_ror:2
    FTMPC_DPP_ADD_D(0x121)      // row_ror:1
#undef FTMPC_DPP_ADD_D
    return x;
}

__device__ __forceinline__ double wave_sum_d(double x) { return quad_sum_d(row_sum16_d(x)); }
__device__ __forceinline__ float wave_sum_t(float x) { return wave_sum(x); }
__device__ __forceinline__ double wave_sum_t(double x) { return wave_sum_d(x); }

// The two sweeps on the float64 factor (solve_reg's right-looking scheme): xv is an LDS vector in natural order, right-hand
// side in, solution out.  With active rows the step's components along their normals are small differences of larger
// entries: the slack steps need them to full relative accuracy, which fp32 sweeps do not give.
template <int NB>
__device__ __forceinline__ void solve64(const f64x4 (&T)[NB * (NB + 1) / 2], const f64x4 (&Wd)[NB], double* xv, int lq, int li) {
    f64x4 Y[NB];
    double p[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) p[J] = 0.0;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        double r = xv[16 * J + li];
        if (J > 0) r -= quad_sum_d(p[J]);
#pragma unroll
        for (int s = 0; s < 4; ++s) Y[J][s] = row_sum16_d(Wd[J][s] * r);
#pragma unroll
        for (int I = J + 1; I < NB; ++I) {
            const f64x4& t = T[tidx(I, J)];
            p[I] += (t.x * Y[J].x + t.y * Y[J].y) + (t.z * Y[J].z + t.w * Y[J].w);
        }
    }
    wave_lds_fence();
    f64x4 a[NB];
#pragma unroll
    for (int J = 0; J < NB; ++J) a[J] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int J = NB - 1; J >= 0; --J) {
        f64x4 r = Y[J];
        if (J < NB - 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) r[s] -= row_sum16_d(a[J][s]);
        }
        const f64x4& wb = Wd[J];
        const double xr = quad_sum_d((wb.x * r.x + wb.y * r.y) + (wb.z * r.z + wb.w * r.w));
        if (lq == 0) xv[16 * J + li] = xr;
#pragma unroll
        for (int K = 0; K < J; ++K) a[K] += T[tidx(J, K)] * xr;
    }
    wave_lds_fence();
}
}  // namespace hullk

template <int NBW, bool TSET = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) ftmpc_solve_hull32_kernel(const DeviceConsts C, const SolveHullParams Q) {
    using namespace wswk;
    const SolveParams& P = Q.base;
    constexpr int NPADW = 16 * NBW;
    constexpr int NTW = NBW * (NBW + 1) / 2;
    constexpr int NVW = (NPADW + 63) / 64;          // wrench variables per lane
    constexpr int NTP = 64 * nvt_of(NBW);           // (the struct_grad instantiation kernel 10 already carries)
    constexpr int NSTG = NPADW / 6;
    constexpr int MHP = 32;                         // row slots per stage
    constexpr int NVC = hullk::nvc_of(NBW);         // hull rows per lane
    constexpr int BUILD_WORDS = 2 * DENSE_WORDS + 256;
    static_assert(NTW * 256 >= BUILD_WORDS, "the dense stage-matrix images live in the tile area during the build");
    static_assert(NSTG * MHP <= 64 * NVC, "row slots");
    __shared__ __attribute__((aligned(16))) float Htl[NTW * 256];       // -H_w' tiles
    // float64 scratch: stage record (build: as 2 x REC_STRIDE floats) | stage storage of struct_grad; during a factorisation the
    // 16 x 17 transpose scratch and the pivot column / row of f64k::potrf_inv16_lds
    constexpr int F64SCR = (REC_STRIDE + 4) + 9 * (NSTG + 2);
    static_assert(F64SCR >= 272 + 32, "factorisation scratch");
    __shared__ __attribute__((aligned(32))) double f64scr[F64SCR];
    float* const recbuf = reinterpret_cast<float*>(f64scr);
    double* const sSl = f64scr + REC_STRIDE + 4;
    __shared__ __attribute__((aligned(16))) float xvp[NPADW], dvp[NPADW];
    __shared__ __attribute__((aligned(16))) double xv64[NPADW];          // right-hand side / solution of the float64 sweeps
    __shared__ __attribute__((aligned(16))) float rv[NTP];
    __shared__ __attribute__((aligned(16))) double Sblk[2 + NSTG * 36];  // [0] = 0 | stage blocks G_k at 2 + 36 k + 6 g + h, float64
    __shared__ __attribute__((aligned(16))) float Sblk32[8 + NSTG * 48]; // the same in fp32 (early iterations), kernel 10's padded layout
    __shared__ __attribute__((aligned(16))) float s_DaT[6 * MAX_NT];    // identity: the wrench components are the inputs
    __shared__ __attribute__((aligned(16))) float s_hA[MHP * 6];        // normals, row r at 6 r
    __shared__ __attribute__((aligned(16))) float s_hAT[6 * MHP];       // the same, component g at 32 g
    __shared__ __attribute__((aligned(16))) float cw[NSTG * MHP];       // per-row values, row r of stage k at 32 k + r (zero beyond hull_rows)
    static_assert(NSTG * MHP >= MAX_NT * MAX_NT, "the stage block of H_w (build only) borrows the row-value array");
    float* const mtab = cw;
    // terminal set: raw terminal sensitivity GN (9 x n, row r at NPADW r), rows of term_A (row i at 9 i), offsets, per-row values,
    // the 9 x 9 core A_T' W A_T and a 9-vector (float64: late iterations)
    constexpr int MTP = 80, NTR = 2;              // row slots, rows per lane
    __shared__ __attribute__((aligned(16))) float s_GN[TSET ? 9 * NPADW : 4];
    __shared__ __attribute__((aligned(16))) float s_tA[TSET ? MTP * 9 : 4];
    __shared__ float s_tb[TSET ? MTP : 4], cwt[TSET ? MTP : 4];
    __shared__ __attribute__((aligned(16))) double M9s[TSET ? 81 : 2], y9s[TSET ? 16 : 2];
    __shared__ float s_ctr[12];                                          // hull centre D (ub / 2 + stuck) | D stuck
    __shared__ double s_ctr64[6];                                        // the centre again, unrounded (output stage)
    __shared__ unsigned char s_stg[NPADW], s_thr[NPADW];
    __shared__ unsigned char s_pg[24], s_ph[24];
    float* const dense = Htl;

    const int lane0 = threadIdx.x;
    const int N = C.N, NT = C.NT;
    TileStore<NTW> htiles;
    htiles.p = Htl;
    htiles.bind(P.hscratch);      // (never used: every tile of this store is in LDS)
    const float mu_stop = (float)C.mu_stop;
    double* const sbuf = reinterpret_cast<double*>(P.hscratch + (int64_t)blockIdx.x * P.tile_words);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int MH = Q.hull_rows;
    const int MT = TSET ? Q.term_rows : 0;
    if constexpr (TSET) {
        for (int i = lane0; i < MTP * 9; i += 64) s_tA[i] = (i < MT * 9) ? (float)Q.termA[i] : 0.f;
        for (int i = lane0; i < MTP; i += 64) {
            s_tb[i] = (i < MT) ? (float)Q.termb[i] : 1.f;
            cwt[i] = 0.f;
        }
    }

    auto pull = [&]() {
        int i = 0;
        if (lane0 == 0) i = atomicAdd(P.qhead, 1);
        return i;
    };
    if (lane0 < 21) {
        int g = 0;
        while ((g + 1) * (g + 2) / 2 <= lane0) ++g;
        s_pg[lane0] = (unsigned char)g;
        s_ph[lane0] = (unsigned char)(lane0 - g * (g + 1) / 2);
    }
    // fp32 seeds: word offsets into the padded fp32 blocks of the four entries this lane subtracts from the diagonal tile (I, I)
    // and from the tile (I, I - 1) -- kernel 10's operand offsets for the pairs (M = I, K = I) and (M = I, K = I - 1)
    int g32_d[NBW], g32_s[NBW];
    {
        const int li0 = lane0 & 15, lq0 = lane0 >> 4, n0 = 6 * N;
#pragma unroll
        for (int M = 0; M < NBW; ++M) {
#pragma unroll
            for (int dk = -1; dk <= 0; ++dk) {
                const int K = M + dk;
                int off = 0;
                if (K >= 0) {
                    const int e1 = 16 * K + 4 * lq0, s1 = (e1 * 43) >> 8, j4 = e1 - 6 * s1;       // e / 6 for e < 128
                    const int e2 = 16 * M + li0, s2 = (e2 * 43) >> 8, a2 = e2 - 6 * s2;
                    if (e2 < n0) {
                        if (s1 == s2) off = 8 + 48 * s1 + 8 * a2 + j4;
                        else if (j4 == 4 && s2 == s1 + 1) off = 8 + 48 * s2 + 8 * a2 - 2;
                    }
                }
                if (dk == 0) g32_d[M] = off;
                else g32_s[M] = off;
            }
        }
    }
    for (int i = lane0; i < 8 + NSTG * 48; i += 64) Sblk32[i] = 0.f;
    for (int i = lane0; i < 2 + NSTG * 36; i += 64) Sblk[i] = 0.0;
    for (int i = lane0; i < 6 * MAX_NT; i += 64) s_DaT[i] = ((i / MAX_NT) == (i % MAX_NT)) ? 1.f : 0.f;
    constexpr int na = 6;
    const int n = N * na;
    const int nb = (n + 15) >> 4;
    for (int e = lane0; e < NPADW; e += 64) {
        const int s = e / na;
        s_stg[e] = (unsigned char)(e < n ? s : 255);
        s_thr[e] = (unsigned char)(e < n ? e - s * na : 255);
    }
    const int mhull = N * MH;
    const int64_t qn = P.B;
    int qnext = pull();
    for (;;) {
        const int qi = __builtin_amdgcn_readfirstlane(qnext);
        if (qi >= qn) break;
        const int64_t inst = qi;
        qnext = pull();
        STAMP_DECL;
        STAMP_START();
        wave_lds_fence();
        int lane = lane_now();
        int li = lane & 15, lq = lane >> 4;
        // ---------------- prologue ----------------
        if (nb > NBW || MH > MHP || MH < 1) {       // (the host routes such shapes to the float64 kernel)
            if (lane == 0) {
                if (P.status) P.status[inst] = 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        if (lane < 12) {      // hull centre (the start point: strictly inside every row) and the thrusters-off wrench (cold linearisation point)
            const int g = lane % 6;
            double acc = 0.0;
            for (int i = 0; i < NT; ++i) acc += C.D[g * MAX_NT + i] * ((lane < 6 ? 0.5 * P.ub[inst * NT + i] : 0.0) + P.stuck[inst * NT + i]);
            s_ctr[lane] = (float)acc;
            if (lane < 6) s_ctr64[lane] = acc;
        }
        {
            const int64_t set = Q.hull_set ? Q.hull_set[inst] : 0;
            for (int i = lane; i < MHP * 6; i += 64) {
                const int r = i / 6, g = i - 6 * r;
                const float a = (r < MH) ? (float)Q.hullA[set * MH * 6 + i] : 0.f;
                s_hA[i] = a;
                s_hAT[g * MHP + r] = a;
            }
            for (int t = lane; t < MAX_NT * MAX_NT; t += 64) {     // stage block of H_w: 2 R on the diagonal (build only: in the row-value array)
                const int a1 = t >> 4, a2 = t & (MAX_NT - 1);
                float r = 0.f;
#pragma unroll
                for (int g = 0; g < 6; ++g) r = (a1 == g) ? (float)C.R[g] : r;
                mtab[t] = (a1 == a2 && a1 < 6) ? 2.f * r : 0.f;
            }
        }
        wave_lds_fence();
#define FTMPC_WB_PART 1
#include "ftmpc_wrench_build.inc"
#undef FTMPC_WB_PART

        // this lane's wrench variables e = v * 64 + lane = (stage, component): linearisation point and start point
        bool wvalid[NVW];
        float ubar[NVW], d[NVW], grd[NVW];
#pragma unroll
        for (int v = 0; v < NVW; ++v) {
            const int e = v * 64 + lane;
            wvalid[v] = e < n;
            const int k = (e * 10923) >> 16, g = e - 6 * k;
            ubar[v] = 0.f;
            d[v] = grd[v] = 0.f;
            if (wvalid[v]) {
                ubar[v] = Q.warmG ? (float)Q.warmG[inst * n + e] : s_ctr[6 + g];
                d[v] = s_ctr[g] - ubar[v];
            }
        }
        f32x4 GNt[TSET ? NBW : 1];      // d x_N / d tau as operand tiles: register s of row group q < 3 = state component 3 s + q, the rest zero
#define FTMPC_WB_PART 2
#define FTMPC_WB_TILES htiles
#define FTMPC_WB_TAIL                                                                                            \
    if constexpr (TSET) {                                                                                       \
        _Pragma("unroll") for (int X = 0; X < NBW; ++X) {                                                       \
            GNt[X] = (lq < 3) ? f32x4{G[X].x, G[X].y, G[X].z, 0.f} : zero4;                                     \
            if (lq < 3) {                                                                                       \
                s_GN[(0 + lq) * NPADW + 16 * X + li] = G[X].x;                                                  \
                s_GN[(3 + lq) * NPADW + 16 * X + li] = G[X].y;                                                  \
                s_GN[(6 + lq) * NPADW + 16 * X + li] = G[X].z;                                                  \
            }                                                                                                   \
        }                                                                                                       \
    }
#include "ftmpc_wrench_build.inc"
#undef FTMPC_WB_TAIL
#undef FTMPC_WB_TILES
#undef FTMPC_WB_PART
        wave_lds_fence();   // the dense images in the tile area are dead from here: Htl holds the -H_w' tiles
        for (int i = lane_now(); i < NSTG * MHP; i += 64) cw[i] = 0.f;      // (was the stage block of H_w during the build)

        // y = H_w x for a vector held NVW per lane, from the -H_w' tiles (one read serves both triangles: ftmpc_solve.hip, start
        // gradient).  Also the gradient's update: with row weights z / s ~ 1e7 the Newton identity H dd = rhs - C' (w . C dd)
        // is a difference of huge numbers in fp32, the product itself is not.
        auto h_times = [&](const float (&x)[NVW], float (&y)[NVW]) {
            const int lane = lane_now();
            const int li = lane & 15, lq = lane >> 4;
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                if (e < NPADW) dvp[e] = x[v];
            }
            wave_lds_fence();
            f32x2 arow[NBW];
            f32x4 acol[NBW];
#pragma unroll
            for (int I = 0; I < NBW; ++I) {
                arow[I] = f32x2{0.f, 0.f};
                acol[I] = zero4;
            }
#pragma unroll
            for (int I = 0; I < NBW; ++I) {
                const float dI = dvp[16 * I + li];
#pragma unroll
                for (int J = 0; J <= I; ++J) {
                    const f32x4 t4 = htiles.ld(tidx(I, J), lane);
                    const f32x4 d4 = lds4(dvp + 16 * J + 4 * lq);
                    arow[I] += f32x2{t4.x, t4.y} * f32x2{d4.x, d4.y};
                    arow[I] += f32x2{t4.z, t4.w} * f32x2{d4.z, d4.w};
                    if (J < I) acol[J] += t4 * dI;
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int J = 0; J < NBW; ++J) {
                float c0 = acol[J].x, c1 = acol[J].y, c2 = acol[J].z, c3 = acol[J].w;
                row_sum16x4(c0, c1, c2, c3);
                if (li == 0) *reinterpret_cast<f32x4*>(xvp + 16 * J + 4 * lq) = f32x4{c0, c1, c2, c3};
            }
            wave_lds_fence();
            float yrow[NBW];
#pragma unroll
            for (int I = 0; I < NBW; ++I) yrow[I] = quad_sum(arow[I].x + arow[I].y);
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * v + j < NBW) t = (lq == j) ? yrow[4 * v + j] : t;
                const int e = v * 64 + lane;
                y[v] = (e < n) ? -(t + xvp[e]) : 0.f;      // the tiles hold -H_w
            }
            wave_lds_fence();
        };
        // ---------------- gradient at the start point: g_w + H_w d ----------------
        {
            float y[NVW];
            h_times(d, y);
#pragma unroll
            for (int v = 0; v < NVW; ++v) grd[v] = wvalid[v] ? y[v] + gw[v] : 0.f;
        }
        STAMP(3);

        // ---------------- the hull rows of this lane: row c = v * 64 + lane = (stage hk, facet hr) ----------------
        bool hv[NVC];
        int hcw[NVC];      // slot 32 k + r of the row in the row-value array (k: stage, r: facet)
        float sh[NVC], zh[NVC];
        float smax = 0.f, gmax = 0.f;
        {
            const float rmh = 1.0f / (float)MH;
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                const int c = v * 64 + lane;
                hv[v] = c < mhull;
                int k = (int)(((float)c + 0.5f) * rmh);
                k = (k * MH > c) ? k - 1 : k;
                k = ((k + 1) * MH <= c) ? k + 1 : k;
                const int r = c - k * MH;
                hcw[v] = hv[v] ? k * MHP + r : 0;
                sh[v] = 1.f;
                zh[v] = 0.f;
                if (hv[v]) {      // b - A centre: the same for every stage
                    float a = (float)Q.hullb[inst * MH + r];
#pragma unroll
                    for (int g = 0; g < 6; ++g) a -= s_hA[6 * r + g] * s_ctr[g];
                    sh[v] = a;
                    smax = fmaxf(smax, a);
                }
            }
#pragma unroll
            for (int v = 0; v < NVW; ++v) gmax = fmaxf(gmax, wvalid[v] ? fabsf(grd[v]) : 0.f);
            gmax = wave_max(gmax);
            smax = wave_max(smax);
        }
        // rows of C x for a natural-order LDS vector, float64 (late iterations: the slack steps of active rows are small by
        // cancellation and need the step's components to full relative accuracy) or fp32
        auto rows_Cx = [&](auto xs, auto (&ch)[NVC]) {
            using XT = std::remove_cv_t<std::remove_pointer_t<decltype(xs)>>;
            using OT = std::remove_reference_t<decltype(ch[0])>;
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                const int hra = 6 * (hcw[v] & (MHP - 1)), hxs = 6 * (hcw[v] >> 5);
                const f32x2 a0 = *reinterpret_cast<const f32x2*>(s_hA + hra), a1 = *reinterpret_cast<const f32x2*>(s_hA + hra + 2),
                            a2 = *reinterpret_cast<const f32x2*>(s_hA + hra + 4);
                const XT* x = xs + hxs;
                const XT s2 = ((XT)a0.x * x[0] + (XT)a0.y * x[1]) + ((XT)a1.x * x[2] + (XT)a1.y * x[3]) + ((XT)a2.x * x[4] + (XT)a2.y * x[5]);
                ch[v] = hv[v] ? (OT)s2 : (OT)0;
            }
        };
        // C' t for per-row values t: element e = v * 64 + lane of the result
        auto cols_Ct = [&](const float (&th)[NVC], float (&out)[NVW]) {
            const int lane = lane_now();
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVC; ++v)
                if (hv[v]) cw[hcw[v]] = th[v];
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                float s = 0.f;
                if (e < n) {
                    const int k = (e * 10923) >> 16, g = e - 6 * k;
                    const f32x4* a4 = reinterpret_cast<const f32x4*>(s_hAT + g * MHP);
                    const f32x4* t4 = reinterpret_cast<const f32x4*>(cw + k * MHP);
#pragma unroll
                    for (int r4 = 0; r4 < MHP / 4; ++r4) {
                        const f32x4 a = a4[r4], t = t4[r4];
                        s += (a.x * t.x + a.y * t.y) + (a.z * t.z + a.w * t.w);
                    }
                }
                out[v] = s;
            }
        };
        auto to_lds = [&](const float (&x)[NVW], auto dst) {
            using XT = std::remove_pointer_t<decltype(dst)>;
            const int lane = lane_now();
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                if (e < NPADW) dst[e] = wvalid[v] ? (XT)x[v] : (XT)0;
            }
            wave_lds_fence();
        };
        auto from_lds = [&](auto src, float (&x)[NVW]) {
            const int lane = lane_now();
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                x[v] = (e < n) ? (float)src[e] : 0.f;
            }
        };
        // ---- terminal rows (TSET): row i = lane + 64 j of term_A (e_N + GN d) <= term_b ----
        bool tvr[NTR];
        int tri[NTR];
        float st[NTR], zt[NTR], rpt[NTR];
#pragma unroll
        for (int j = 0; j < NTR; ++j) {
            tri[j] = lane + 64 * j;
            tvr[j] = TSET && tri[j] < MT;
            tri[j] = tvr[j] ? tri[j] : 0;
            st[j] = 1.f;
            zt[j] = rpt[j] = 0.f;
        }
        // A_T (GN x) for a natural-order LDS vector (float64 or fp32)
        auto term_rows = [&](auto xs, auto (&ct)[NTR]) {
            using XT = std::remove_cv_t<std::remove_pointer_t<decltype(xs)>>;
            using OT = std::remove_reference_t<decltype(ct[0])>;
            const int lane = lane_now();
            XT part[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) part[r] = (XT)0;
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                if (e < n) {
                    const XT x = xs[e];
#pragma unroll
                    for (int r = 0; r < 9; ++r) part[r] += (XT)s_GN[r * NPADW + e] * x;
                }
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) part[r] = hullk::wave_sum_t(part[r]);
#pragma unroll
            for (int j = 0; j < NTR; ++j) {
                XT a = (XT)0;
#pragma unroll
                for (int r = 0; r < 9; ++r) a += (XT)s_tA[tri[j] * 9 + r] * part[r];
                ct[j] = tvr[j] ? (OT)a : (OT)0;
            }
        };
        // out += GN' (A_T' t)
        auto term_cols = [&](const float (&tt)[NTR], float (&out)[NVW]) {
            const int lane = lane_now();
            float p9[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < NTR; ++j) a += tvr[j] ? s_tA[tri[j] * 9 + r] * tt[j] : 0.f;
                p9[r] = wave_sum(a);
            }
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                if (e < n) {
                    float a = 0.f;
#pragma unroll
                    for (int r = 0; r < 9; ++r) a += s_GN[r * NPADW + e] * p9[r];
                    out[v] += a;
                }
            }
        };
        if constexpr (TSET) {      // residual at the start point: slack max(residual, 0.1) and the primal residual the steps shrink
            to_lds(d, xvp);
            float ct[NTR];
            term_rows((const float*)xvp, ct);
#pragma unroll
            for (int j = 0; j < NTR; ++j)
                if (tvr[j]) {
                    float res = s_tb[tri[j]] - ct[j];
#pragma unroll
                    for (int r = 0; r < 9; ++r) res -= s_tA[tri[j] * 9 + r] * (float)Q.eN[inst * 9 + r];
                    st[j] = fmaxf(res, 0.1f);
                    rpt[j] = st[j] - res;
                    smax = fmaxf(smax, st[j]);
                }
            smax = wave_max(smax);
        }
        {
            const float mu0 = fmaxf(0.02f * gmax * smax, 1e-3f);
#pragma unroll
            for (int v = 0; v < NVC; ++v) zh[v] = hv[v] ? mu0 / sh[v] : 0.f;
#pragma unroll
            for (int j = 0; j < NTR; ++j) zt[j] = tvr[j] ? mu0 / st[j] : 0.f;
        }
        STAMP(4);

        // ---------------- interior-point iterations over the hull rows ----------------
        f64x4 T64[NTW], W64[NBW];
        int status = 1, nit = 0;
        bool in64 = false, fac64 = false;
#ifndef FTMPC_HULL_REFRESH_ALL
#define FTMPC_HULL_REFRESH_ALL 0
#endif
#ifndef FTMPC_HULL_POLISH
#define FTMPC_HULL_POLISH 1
#endif
#ifndef FTMPC_HULL_PW0
#define FTMPC_HULL_PW0 1e3f      // penalty of the polish relative to max diag(H_w) / |c|^2 (the float64 kernel: 1e6; here W s is formed in fp32)
#endif
#ifndef FTMPC_HULL_NOREFRESH
#define FTMPC_HULL_NOREFRESH 0
#endif
        int refines_left = (C.mu_refine > 0.0 && !FTMPC_HULL_NOREFRESH) ? ((NBW > 6 || FTMPC_HULL_REFRESH_ALL) ? C.max_iters + 1 : 1) : 0;
        float mu_last = 3.0e38f;
        const float inv_m = 1.0f / (float)(mhull + MT);
        // Active-set polish (oracle/qp_oracle.py:polish_general; the float64 kernel runs the same): once the iteration has
        // converged, the rows with z > s are taken as the active set and the equality-constrained problem on it is solved by two
        // steps of the method of multipliers -- penalty W = 1e3 max diag(H_w) / |c|^2 on the active rows, zero on the others:
        // the Newton matrix's own shape, so a polish round is one more pass through this loop (float64 factorisation and
        // sweeps, the float64 reference gradient first) with the multiplier step in place of Mehrotra's.  Then the signs are
        // checked (a negative multiplier leaves the set, a violated row enters) and the round repeated until nothing changes.
        // Without it an iterate at mu 1e-10 is up to 7e-5 f_max from the exact solution where rows are weakly active.
        bool pol = false, verified = false;
        int prd = 0, pin = 0;
        bool pa[NVC], pat[NTR];
        float pwh[NVC], pwt[NTR], d_ipm[NVW];
#pragma unroll
        for (int v = 0; v < NVC; ++v) {
            pa[v] = false;
            pwh[v] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NTR; ++j) {
            pat[j] = false;
            pwt[j] = 0.f;
        }
        for (int it = 0; it <= C.max_iters + 16; ++it) {
            wave_lds_fence();
            lane = lane_now();
            li = lane & 15;
            lq = lane >> 4;
            const bool do_ref = __builtin_amdgcn_readfirstlane(pol ? pin == 0 : (refines_left > 0 && mu_last < (float)C.mu_refine));
            if (do_ref) {   // float64, structured, at the current iterate
#pragma unroll
                for (int v = 0; v < NVW; ++v) rv[v * 64 + lane] = wvalid[v] ? d[v] : 0.f;
                wave_lds_fence();
                struct_grad<lds_f64*, NTP>(C, (glb_cf64*)recg, (lds_f64*)reinterpret_cast<double*>(recbuf), (lds_cf32*)s_DaT, (lds_cf32*)rv,
                                          (lds_f64*)sSl, (glb_f64*)sbuf, 6, lane);
#pragma unroll
                for (int v = 0; v < NVW; ++v) grd[v] = wvalid[v] ? (float)sbuf[v * 64 + lane] : 0.f;
                --refines_left;
                STAMP(8);
            }
            float csum = 0.f, rpn = 0.f;
#pragma unroll
            for (int v = 0; v < NVC; ++v) csum += hv[v] ? sh[v] * zh[v] : 0.f;
#pragma unroll
            for (int j = 0; j < NTR; ++j) {
                csum += tvr[j] ? st[j] * zt[j] : 0.f;
                rpn = fmaxf(rpn, tvr[j] ? fabsf(rpt[j]) : 0.f);
            }
            const float mu = wave_sum(csum) * inv_m;
            if constexpr (TSET) rpn = wave_max(rpn);
            if (!pol) {
                mu_last = mu;
                if (__builtin_amdgcn_readfirstlane(!(mu == mu) || !(rpn == rpn))) {
                    status = 2;
                    break;
                }
                if (__builtin_amdgcn_readfirstlane(mu < mu_stop && rpn < 1e-7f)) {
                    status = 0;
#if FTMPC_HULL_POLISH
                    // ---- enter the polish: active set, penalties, multipliers (in zh / zt), true slacks of the terminal rows ----
                    pol = true;
                    float hdmax = 0.f;
#pragma unroll
                    for (int v = 0; v < NVW; ++v) {
                        const int e = v * 64 + lane;
                        d_ipm[v] = d[v];
                        if (e < n) {      // diagonal entry (r, r) of the stored -(H_II)' tile: lane 16 (r >> 2) + r, register r & 3
                            const int I = e >> 4, r = e & 15;
                            hdmax = fmaxf(hdmax, fabsf(Htl[tidx(I, I) * 256 + 4 * (16 * (r >> 2) + r) + (r & 3)]));
                        }
                    }
                    const float pw = FTMPC_HULL_PW0 * wave_max(hdmax);
#pragma unroll
                    for (int v = 0; v < NVC; ++v) {
                        pa[v] = hv[v] && zh[v] > sh[v];
                        zh[v] = pa[v] ? zh[v] : 0.f;
                        const int hra = 6 * (hcw[v] & (MHP - 1));
                        float a2 = 0.f;
#pragma unroll
                        for (int g = 0; g < 6; ++g) a2 += s_hA[hra + g] * s_hA[hra + g];
                        pwh[v] = hv[v] ? pw / fmaxf(a2, 1e-30f) : 0.f;
                    }
                    if constexpr (TSET) {
                        wave_lds_fence();
                        if (lane < 45) {      // Gram matrix GN GN' (9 x 9), for the norms |A_T,i GN|^2
                            int r1 = 0;
                            while ((r1 + 1) * (r1 + 2) / 2 <= lane) ++r1;
                            const int r2 = lane - r1 * (r1 + 1) / 2;
                            double acc = 0.0;
                            for (int e = 0; e < n; ++e) acc += (double)s_GN[r1 * NPADW + e] * (double)s_GN[r2 * NPADW + e];
                            M9s[r1 * 9 + r2] = acc;
                            M9s[r2 * 9 + r1] = acc;
                        }
                        wave_lds_fence();
#pragma unroll
                        for (int j = 0; j < NTR; ++j) {
                            pat[j] = tvr[j] && zt[j] > st[j];
                            zt[j] = pat[j] ? zt[j] : 0.f;
                            st[j] -= rpt[j];      // the true slack (the residual the steps carried has shrunk below 1e-7)
                            rpt[j] = 0.f;
                            double c2 = 0.0;
                            for (int r1 = 0; r1 < 9; ++r1)
                                for (int r2 = 0; r2 < 9; ++r2) c2 += (double)s_tA[tri[j] * 9 + r1] * M9s[r1 * 9 + r2] * (double)s_tA[tri[j] * 9 + r2];
                            pwt[j] = tvr[j] ? pw / fmaxf((float)c2, 1e-30f) : 0.f;
                        }
                        wave_lds_fence();
                    }
                    fac64 = in64 = true;
                    continue;      // (the next pass takes the float64 reference gradient first)
#else
                    break;
#endif
                }
                if (it >= C.max_iters) break;
            }
            if (pin == 0) ++nit;
            float rsh[NVC], wh[NVC], rst[NTR], wt[NTR];
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                rsh[v] = __builtin_amdgcn_rcpf(sh[v]);
                wh[v] = pol ? (pa[v] ? pwh[v] : 0.f) : (hv[v] ? zh[v] * rsh[v] : 0.f);
            }
#pragma unroll
            for (int j = 0; j < NTR; ++j) {
                rst[j] = __builtin_amdgcn_rcpf(st[j]);
                wt[j] = pol ? (pat[j] ? pwt[j] : 0.f) : (tvr[j] ? zt[j] * rst[j] : 0.f);
            }
            // Early iterations run the factorisation and the sweeps in fp32 (chol_reg / solve_reg); from the iteration in which
            // a row weight z / s passes FTMPC_HULL_W64 on, in float64.  With weights up to that the stage blocks are no larger
            // than H_w's own entries and eliminating through them in fp32 is harmless; beyond, see the header.
#ifndef FTMPC_HULL_W64
#define FTMPC_HULL_W64 8.f      // measured (16 384 instances): 4 and 32 give the same rate and iteration count, 256 costs an iteration
#endif
#ifndef FTMPC_HULL_WF64
#define FTMPC_HULL_WF64 1.0e5f   // ... and the factorisation itself stays fp32 (widened to float64 for the sweeps) up to this weight.
// Measured on 16 384 instances (scripts/hull32_check.py): thresholds 512, 4 096, 65 536 and 1e6 all give the float64 kernel's
// iteration counts and the same errors (7.2e-5 f_max worst) at 1.69 / 1.74 / 1.78 / 1.83 M QP-steps/s; with no float64
// factorisation at all 22 % of the instances break down.  Where the fp32 factorisation does fail below the threshold, the
// iteration is redone in float64.
#endif
            if (!fac64) {
                float wmax = 0.f;
#pragma unroll
                for (int v = 0; v < NVC; ++v) wmax = fmaxf(wmax, wh[v]);
#pragma unroll
                for (int j = 0; j < NTR; ++j) wmax = fmaxf(wmax, wt[j]);
                wmax = wave_max(wmax);
                in64 = in64 || __builtin_amdgcn_readfirstlane(wmax > FTMPC_HULL_W64);
                fac64 = __builtin_amdgcn_readfirstlane(wmax > FTMPC_HULL_WF64);
            }
            // stage blocks G_k = sum_r w_kr a_r a_r'
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVC; ++v)
                if (hv[v]) cw[hcw[v]] = wh[v];
            wave_lds_fence();
            // terminal rows: the 9 x 9 core A_T' W A_T (float64: products of the fp32 rows are exact) and  -P = -(core) GN  as operand tiles
            f32x4 nPt[TSET ? NBW : 1];
            f64x4 nP64[TSET ? NBW : 1];      // (unrounded: with weights ~1e7 the rounding of P to fp32 would be noise of order 1 on H_w again)
            if constexpr (TSET) {
#pragma unroll
                for (int j = 0; j < NTR; ++j)
                    if (tvr[j]) cwt[tri[j]] = wt[j];
                wave_lds_fence();
                if (lane < 45) {
                    int r1 = 0;
                    while ((r1 + 1) * (r1 + 2) / 2 <= lane) ++r1;
                    const int r2 = lane - r1 * (r1 + 1) / 2;
                    double acc = 0.0;
                    for (int i = 0; i < MT; ++i) acc += (double)cwt[i] * ((double)s_tA[i * 9 + r1] * (double)s_tA[i * 9 + r2]);
                    M9s[r1 * 9 + r2] = acc;
                    M9s[r2 * 9 + r1] = acc;
                }
                wave_lds_fence();
#pragma unroll
                for (int X = 0; X < NBW; ++X) {
                    f32x4 t = zero4;
                    f64x4 t64 = {0.0, 0.0, 0.0, 0.0};
                    if (lq < 3) {
                        float gc[9];
#pragma unroll
                        for (int r = 0; r < 9; ++r) gc[r] = s_GN[r * NPADW + 16 * X + li];
#pragma unroll
                        for (int rr = 0; rr < 3; ++rr) {
                            double a = 0.0;
#pragma unroll
                            for (int r = 0; r < 9; ++r) a += M9s[(3 * rr + lq) * 9 + r] * (double)gc[r];
                            t[rr] = -(float)a;
                            t64[rr] = -a;
                        }
                    }
                    nPt[X] = t;
                    nP64[X] = t64;
                }
            }
            bool ok = true;
            for (int attempt = (pin == 0 ? 0 : 2); attempt < 2; ++attempt) {      // (the second multiplier step of a polish round keeps the factor)
                ok = true;
                if (fac64) {
                    for (int idx = lane; idx < N * 21; idx += 64) {      // float64: the products of the fp32 normals are exact, G keeps its rank
                        const int k = (idx * 3121) >> 16, p = idx - 21 * k;      // idx / 21 for idx < 5000
                        const int g = s_pg[p], hh = s_ph[p];
                        const f32x4* ag4 = reinterpret_cast<const f32x4*>(s_hAT + g * MHP);
                        const f32x4* ah4 = reinterpret_cast<const f32x4*>(s_hAT + hh * MHP);
                        const f32x4* w4 = reinterpret_cast<const f32x4*>(cw + k * MHP);
                        double sacc = 0.0, sacb = 0.0;
    #pragma unroll
                        for (int r4 = 0; r4 < MHP / 4; ++r4) {
                            const f32x4 a = ag4[r4], b = ah4[r4], w = w4[r4];
                            sacc += (double)w[0] * ((double)a[0] * (double)b[0]);
                            sacb += (double)w[1] * ((double)a[1] * (double)b[1]);
                            sacc += (double)w[2] * ((double)a[2] * (double)b[2]);
                            sacb += (double)w[3] * ((double)a[3] * (double)b[3]);
                        }
                        sacc += sacb;
                        Sblk[2 + k * 36 + g * 6 + hh] = sacc;
                        Sblk[2 + k * 36 + hh * 6 + g] = sacc;
                    }
                    wave_lds_fence();
                    STAMP(7);
                    // float64 factorisation of H_w + G: seeds -(H_w + G)' read in the float64 accumulator layout (rows q + 4 s);
                    // element [row][col] of the stored -(M_IJ)' tile is -M[16 I + col][16 J + row]
                    const float* hb = Htl + 4 * li + lq;
                    auto gsub = [&](int I, int J, f64x4& r) {
                        const int c1 = 16 * I + li, t1 = (c1 * 43) >> 8;       // e / 6 for e < 128
    #pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int c2 = 16 * J + lq + 4 * rr, t2 = (c2 * 43) >> 8;
                            const int off = (c1 < n && c2 < n && t1 == t2) ? 2 + 36 * t1 + 6 * (c1 - 6 * t1) + (c2 - 6 * t2) : 0;
                            r[rr] -= Sblk[off];
                        }
                    };
                    auto seed = [&](int I, int J) -> f64x4 {
                        const float* t = hb + tidx(I, J) * 256;
                        f64x4 r = {(double)t[0], (double)t[64], (double)t[128], (double)t[192]};
                        if (J == I || J == I - 1) gsub(I, J, r);
                        if constexpr (TSET) {      // - (GN' core GN)' block: the float64 operands are the fp32 ones widened (same row order)
                            const f64x4 gj = {(double)GNt[J].x, (double)GNt[J].y, (double)GNt[J].z, 0.0};
                            r = hullk::mm_tn64(gj, nP64[I], r);
                        }
                        return r;
                    };
                    hullk::chol64_col<NBW, 0>(seed, f64scr, lq, li, ok, T64, W64);
                    STAMP(5);
                    break;
                } else {
                    for (int idx = lane; idx < N * 21; idx += 64) {
                        const int k = (idx * 3121) >> 16, p = idx - 21 * k;
                        const int g = s_pg[p], hh = s_ph[p];
                        const f32x4* ag4 = reinterpret_cast<const f32x4*>(s_hAT + g * MHP);
                        const f32x4* ah4 = reinterpret_cast<const f32x4*>(s_hAT + hh * MHP);
                        const f32x4* w4 = reinterpret_cast<const f32x4*>(cw + k * MHP);
                        float sacc = 0.f;
    #pragma unroll
                        for (int r4 = 0; r4 < MHP / 4; ++r4) {
                            const f32x4 ab = ag4[r4] * ah4[r4], w = w4[r4];
                            sacc += (ab.x * w.x + ab.y * w.y) + (ab.z * w.z + ab.w * w.w);
                        }
                        Sblk32[8 + k * 48 + g * 8 + hh] = sacc;
                        Sblk32[8 + k * 48 + hh * 8 + g] = sacc;
                    }
                    wave_lds_fence();
                    STAMP(7);
                    f32x4 Xt[NTW], Tt[NTW], Wd[NBW];
    #pragma unroll
                    for (int I = 0; I < NBW; ++I) {
    #pragma unroll
                        for (int J = 0; J <= I; ++J) {
                            f32x4 t = htiles.ld(tidx(I, J), lane);
                            if (J == I || J == I - 1) {
                                const int off = (J == I) ? g32_d[I] : g32_s[I];
                                const f32x2 lo2 = *reinterpret_cast<const f32x2*>(Sblk32 + off);
                                const f32x2 hi2 = *reinterpret_cast<const f32x2*>(Sblk32 + off + 2);
                                t -= f32x4{lo2.x, lo2.y, hi2.x, hi2.y};
                            }
                            if constexpr (TSET) t = mm_tn(GNt[J], nPt[I], t);
                            Xt[tidx(I, J)] = t;
                        }
                    }
                    for (int e = lane; e < NPADW; e += 64) dvp[e] = 0.f;      // (no diagonal shift)
                    wave_lds_fence();
                    f32x4 pre0[NBW];
    #pragma unroll
                    for (int I = 0; I < NBW; ++I) pre0[I] = zero4;
                    const RegTiles<NTW> xt{Xt};
                    ok = chol_reg<NBW, RegTiles<NTW>, false, false, false>(xt, dvp, recbuf, n, lane, Tt, Wd, pre0, nullptr);
                    // parked in the float64 arrays' registers (two fp32 tiles' worth per slot would fit; one is enough): one factor
                    // storage for both precisions, so the register allocator sees one live set across the iteration
                    if (in64) {      // float64 sweeps on the fp32 factor (the sweeps take any consistent row order of the tiles)
    #pragma unroll
                        for (int t = 0; t < NTW; ++t) T64[t] = f64x4{(double)Tt[t].x, (double)Tt[t].y, (double)Tt[t].z, (double)Tt[t].w};
    #pragma unroll
                        for (int J = 0; J < NBW; ++J) W64[J] = f64x4{(double)Wd[J].x, (double)Wd[J].y, (double)Wd[J].z, (double)Wd[J].w};
                    } else {
    #pragma unroll
                        for (int t = 0; t < NTW; ++t) T64[t] = hullk::park32(Tt[t]);
    #pragma unroll
                        for (int J = 0; J < NBW; ++J) W64[J] = hullk::park32(Wd[J]);
                    }
                    STAMP(5);
                    if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
                    fac64 = true;      // the fp32 factorisation broke down below the weight it is trusted to: this iteration again, and
                    in64 = true;       // all later ones, in float64
                }
            }
            if (__builtin_amdgcn_readfirstlane(!__all(ok))) {     // (float64 too runs out near mu ~ 1e-12 ... 1e-13)
                --nit;
                if (pol) break;                                     // (not verified: the interior-point iterate is returned)
                status = (mu < 1e-7f && rpn < 1e-7f) ? 0 : 2;      // (as the float64 kernel: the terminal rows must have closed their residual)
                break;
            }
            // (H_w + G) x = rhs; the rows C x of the solution
            auto solve_rows = [&](const float (&rhs)[NVW], float (&ch)[NVC], float (&ctt)[NTR], float* xout) {
                if (in64) {
                    to_lds(rhs, xv64);
                    hullk::solve64<NBW>(T64, W64, xv64, lq, li);
                    rows_Cx((const double*)xv64, ch);
                    if constexpr (TSET) term_rows((const double*)xv64, ctt);
                    if (xout) from_lds((const double*)xv64, *reinterpret_cast<float(*)[NVW]>(xout));
                } else {
                    f32x4 Tt[NTW], Wd[NBW];
#pragma unroll
                    for (int t = 0; t < NTW; ++t) Tt[t] = hullk::unpark32(T64[t]);
#pragma unroll
                    for (int J = 0; J < NBW; ++J) Wd[J] = hullk::unpark32(W64[J]);
                    to_lds(rhs, xvp);
                    solve_reg<NBW>(Tt, Wd, xvp, NBW, lane);
                    rows_Cx((const float*)xvp, ch);
                    if constexpr (TSET) term_rows((const float*)xvp, ctt);
                    if (xout) from_lds((const float*)xvp, *reinterpret_cast<float(*)[NVW]>(xout));
                }
            };
            STAMP(9);
            // predictor: (H_w + G) da = -grd
            float rhs[NVW], dd[NVW];
            float ch[NVC], ctt[NTR] = {0.f, 0.f};
            float dzh_a[NVC], dst_a[NTR], dzt_a[NTR], ap = 1.f, ad = 1.f;
            float rch[NVC], th[NVC], rct[NTR], tt[NTR];
            if (!pol) {
#pragma unroll
            for (int v = 0; v < NVW; ++v) rhs[v] = -grd[v];
            if constexpr (TSET) {      // the terminal rows carry their primal residual: t = -z rp / s
                float t0[NTR];
#pragma unroll
                for (int j = 0; j < NTR; ++j) t0[j] = tvr[j] ? -zt[j] * rpt[j] * rst[j] : 0.f;
                term_cols(t0, rhs);
            }
            solve_rows(rhs, ch, ctt, nullptr);
            // ds = -C da (terminal rows: -rp - C da),  dz = -z - z ds / s
#pragma unroll
            for (int j = 0; j < NTR; ++j) {
                dst_a[j] = dzt_a[j] = 0.f;
                if (tvr[j]) {
                    dst_a[j] = -rpt[j] - ctt[j];
                    dzt_a[j] = -zt[j] - zt[j] * dst_a[j] * rst[j];
                    if (dst_a[j] < 0.f) ap = fminf(ap, -st[j] * __builtin_amdgcn_rcpf(dst_a[j]));
                    if (dzt_a[j] < 0.f) ad = fminf(ad, -zt[j] * __builtin_amdgcn_rcpf(dzt_a[j]));
                }
            }
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                dzh_a[v] = 0.f;
                if (hv[v]) {
                    dzh_a[v] = -zh[v] + zh[v] * ch[v] * rsh[v];
                    if (ch[v] > 0.f) ap = fminf(ap, sh[v] * __builtin_amdgcn_rcpf(ch[v]));
                    if (dzh_a[v] < 0.f) ad = fminf(ad, -zh[v] * __builtin_amdgcn_rcpf(dzh_a[v]));
                }
            }
            ap = wave_min(ap);
            ad = wave_min(ad);
            csum = 0.f;
#pragma unroll
            for (int v = 0; v < NVC; ++v) csum += hv[v] ? (sh[v] - ap * ch[v]) * (zh[v] + ad * dzh_a[v]) : 0.f;
#pragma unroll
            for (int j = 0; j < NTR; ++j) csum += tvr[j] ? (st[j] + ap * dst_a[j]) * (zt[j] + ad * dzt_a[j]) : 0.f;
            const float mu_aff = wave_sum(csum) * inv_m;
            float sigma = mu_aff / mu;
            sigma = fminf(fmaxf(sigma * sigma * sigma, 0.f), 1.f);
            // corrector: rc = s z + ds_a dz_a - sigma mu,  t = -z + rc / s,  rhs = -grd + C' t
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                rch[v] = th[v] = 0.f;
                if (hv[v]) {
                    rch[v] = sh[v] * zh[v] - ch[v] * dzh_a[v] - sigma * mu;
                    th[v] = (-ch[v] * dzh_a[v] - sigma * mu) * rsh[v];      // -z + rc / s without the cancellation
                }
            }
            if constexpr (TSET) {      // rc = s z + ds_a dz_a - sigma mu,  t = -z + (rc - z rp) / s
#pragma unroll
                for (int j = 0; j < NTR; ++j) {
                    rct[j] = tt[j] = 0.f;
                    if (tvr[j]) {
                        rct[j] = st[j] * zt[j] + dst_a[j] * dzt_a[j] - sigma * mu;
                        tt[j] = (dst_a[j] * dzt_a[j] - sigma * mu - zt[j] * rpt[j]) * rst[j];
                    }
                }
            }
            } else {      // polish: the multiplier step  (H_w + C_A' W C_A) dd = -grd + C_A' (W s_A - lam)
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    rch[v] = 0.f;
                    th[v] = pa[v] ? wh[v] * sh[v] - zh[v] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < NTR; ++j) {
                    rct[j] = 0.f;
                    tt[j] = pat[j] ? wt[j] * st[j] - zt[j] : 0.f;
                }
            }
            float ct[NVW];
            cols_Ct(th, ct);
#pragma unroll
            for (int v = 0; v < NVW; ++v) rhs[v] = -grd[v] + ct[v];
            if constexpr (TSET) term_cols(tt, rhs);
            solve_rows(rhs, ch, ctt, dd);
            if (pol) {
                // lam += W (C dd - s),  s -= C dd,  d += dd,  the gradient follows through H_w dd.  C dd - s is a difference of
                // nearly equal numbers times a large weight: taken in float64 from the float64 solution (still in xv64)
                double chd[NVC], cttd[NTR] = {0.0, 0.0};
                rows_Cx((const double*)xv64, chd);
                if constexpr (TSET) term_rows((const double*)xv64, cttd);
#pragma unroll
                for (int v = 0; v < NVC; ++v)
                    if (hv[v]) {
                        if (pa[v]) zh[v] += wh[v] * (float)(chd[v] - (double)sh[v]);
                        sh[v] = (float)((double)sh[v] - chd[v]);
                    }
#pragma unroll
                for (int j = 0; j < NTR; ++j)
                    if (tvr[j]) {
                        if (pat[j]) zt[j] += wt[j] * (float)(cttd[j] - (double)st[j]);
                        st[j] = (float)((double)st[j] - cttd[j]);
                    }
                float hdd[NVW];
                h_times(dd, hdd);
#pragma unroll
                for (int v = 0; v < NVW; ++v)
                    if (wvalid[v]) {
                        grd[v] += hdd[v];
                        d[v] += dd[v];
                    }
                // the multiplier steps have converged when the active rows are met: further steps on the same factor until then
                float ares = 0.f;
#pragma unroll
                for (int v = 0; v < NVC; ++v) ares = fmaxf(ares, pa[v] ? fabsf(sh[v]) : 0.f);
#pragma unroll
                for (int j = 0; j < NTR; ++j) ares = fmaxf(ares, pat[j] ? fabsf(st[j]) : 0.f);
                ares = wave_max(ares);
                ++pin;
                if (__builtin_amdgcn_readfirstlane(pin < 2 || (pin < 4 && ares > 2e-6f))) continue;      // (two steps at least: measured, one leaves 7e-5)
                pin = 0;
                bool changed = ares > 2e-6f;
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    if (pa[v] && zh[v] < 0.f) {
                        pa[v] = false;
                        zh[v] = 0.f;
                        changed = true;
                    } else if (!pa[v] && hv[v] && sh[v] < -1e-6f) {
                        pa[v] = true;
                        changed = true;
                    }
                }
#pragma unroll
                for (int j = 0; j < NTR; ++j) {
                    if (pat[j] && zt[j] < 0.f) {
                        pat[j] = false;
                        zt[j] = 0.f;
                        changed = true;
                    } else if (!pat[j] && tvr[j] && st[j] < -1e-6f) {
                        pat[j] = true;
                        changed = true;
                    }
                }
                if (__builtin_amdgcn_readfirstlane(!__any(changed))) {
                    verified = true;
                    break;
                }
                if (++prd == 3) break;
                continue;
            }
            float dzh[NVC], dst[NTR], dzt[NTR];
            ap = 1e30f;
            ad = 1e30f;
#pragma unroll
            for (int j = 0; j < NTR; ++j) {
                dst[j] = dzt[j] = 0.f;
                if (tvr[j]) {
                    dst[j] = -rpt[j] - ctt[j];
                    dzt[j] = (-rct[j] - zt[j] * dst[j]) * rst[j];
                    if (dst[j] < 0.f) ap = fminf(ap, -st[j] * __builtin_amdgcn_rcpf(dst[j]));
                    if (dzt[j] < 0.f) ad = fminf(ad, -zt[j] * __builtin_amdgcn_rcpf(dzt[j]));
                }
            }
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                dzh[v] = 0.f;
                if (hv[v]) {
                    dzh[v] = (-rch[v] + zh[v] * ch[v]) * rsh[v];
                    if (ch[v] > 0.f) ap = fminf(ap, sh[v] * __builtin_amdgcn_rcpf(ch[v]));
                    if (dzh[v] < 0.f) ad = fminf(ad, -zh[v] * __builtin_amdgcn_rcpf(dzh[v]));
                }
            }
            ap = fminf(1.f, 0.9995f * wave_min(ap));
            ad = fminf(1.f, 0.9995f * wave_min(ad));
            float hdd[NVW];
            h_times(dd, hdd);
#pragma unroll
            for (int v = 0; v < NVW; ++v)
                if (wvalid[v]) {
                    grd[v] += ap * hdd[v];
                    d[v] += ap * dd[v];
                }
#pragma unroll
            for (int v = 0; v < NVC; ++v)
                if (hv[v]) {
                    sh[v] -= ap * ch[v];
                    zh[v] += ad * dzh[v];
                }
#pragma unroll
            for (int j = 0; j < NTR; ++j)
                if (tvr[j]) {
                    st[j] += ap * dst[j];
                    zt[j] += ad * dzt[j];
                    rpt[j] *= (1.f - ap);
                }
            STAMP(10);
        }
        // ---------------- certificate ----------------
        // A polish that was verified IS the certificate (the exact solution on an active set whose signs check).  Handed over to
        // the float64 kernel (the host enqueues it behind this one): a polish that did not settle -- its iterate is dropped for the
        // interior-point one --, and the instances with hull and terminal rows active together (FTMPC_HULL_HANDOVER_BOTH).
#ifndef FTMPC_HULL_HANDOVER_BOTH
#define FTMPC_HULL_HANDOVER_BOTH 1
#endif
        if (status == 0) {
            bool hand_over = false;
#if FTMPC_HULL_POLISH
            if (!verified) {
                hand_over = true;
                if (pol) {
#pragma unroll
                    for (int v = 0; v < NVW; ++v) d[v] = d_ipm[v];
                }
            }
            bool acth = false, actt = false;
#pragma unroll
            for (int v = 0; v < NVC; ++v) acth = acth || pa[v];
#pragma unroll
            for (int j = 0; j < NTR; ++j) actt = actt || pat[j];
            if (FTMPC_HULL_HANDOVER_BOTH && TSET) hand_over = hand_over || (__any(acth) && __any(actt));
#else
            bool weak = false, acth = false, actt = false;
#pragma unroll
            for (int v = 0; v < NVC; ++v)
                if (hv[v]) {
                    weak = weak || (zh[v] < 1e3f * sh[v] && sh[v] < 1e3f * zh[v]);
                    acth = acth || zh[v] > sh[v];
                }
#pragma unroll
            for (int j = 0; j < NTR; ++j)
                if (tvr[j]) {
                    weak = weak || (zt[j] < 1e3f * st[j] && st[j] < 1e3f * zt[j]);
                    actt = actt || zt[j] > st[j];
                }
            hand_over = __any(weak) || (__any(acth) && __any(actt));
#endif
            if (Q.fb_list && hand_over && lane_now() == 0) Q.fb_list[atomicAdd(Q.fb_count, 1)] = (int32_t)inst;
        }
        // ---------------- outputs ----------------
        lane = lane_now();
        wave_lds_fence();
#pragma unroll
        for (int v = 0; v < NVW; ++v) {
            const int e = v * 64 + lane;
            if (e < n) {
                const double tau = (status == 2) ? (double)ubar[v] : (double)ubar[v] + (double)d[v];
                if (e < 6) xv64[e] = tau;
                if (Q.out_G) Q.out_G[inst * n + e] = tau;
            }
        }
        wave_lds_fence();
        {
            // The allocator takes tau_0 next and needs it INSIDE the true hull (float64 normals and offsets): an active facet is
            // met to fp32 accuracy only, a few 1e-7 outside as often as inside.  Pull tau_0 towards the centre by the smallest
            // factor that leaves every facet a relative margin of 1e-8 (of the order of 1e-6: far inside the specification).
            float eps = 0.f;
            if (lane < MH) {
                const int64_t set = Q.hull_set ? Q.hull_set[inst] : 0;
                const double* a = Q.hullA + (set * MH + lane) * 6;
                const double b = Q.hullb[inst * MH + lane];
                double s0 = b, st = b;
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    s0 -= a[g] * s_ctr64[g];
                    st -= a[g] * xv64[g];
                }
                const double need = (1e-8 * s0 - st) / (s0 - st);      // s0 > 0; st < s0 whenever the row matters
                eps = (st < 1e-8 * s0 && s0 > st) ? (float)need * 1.0001f : 0.f;
            }
            eps = wave_max(eps);
            if (lane < 6) Q.out_tau0[inst * 6 + lane] = s_ctr64[lane] + (1.0 - (double)eps) * (xv64[lane] - s_ctr64[lane]);
        }
        if (lane == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        STAMP(11);
#ifdef FTMPC_STAMPS
        if (lane == 0 && inst < 4096) {
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(P.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = st_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_hull32_kernel<6, false>(const DeviceConsts, const SolveHullParams);
template __global__ void ftmpc_solve_hull32_kernel<6, true>(const DeviceConsts, const SolveHullParams);      // + the terminal set

}  // namespace ftmpc
