// ftmpc_solve_wsw.hip -- kernel 10: the thruster-space QP solved THROUGH WRENCH SPACE by ONE WAVE per instance, fp32.
//
// Kernel 8 (ftmpc_solve_ws.hip) gives a 4-wave workgroup to every instance of the reference's 16-thruster vehicle; its
// Newton systems are 6N x 6N -- exactly the size kernel 2 (ftmpc_solve.hip) factorises in the registers of one wave.  This
// kernel is kernel 2's machinery around kernel 8's algebra:
//     H = DD' H_w DD + 2 rho I,   H_w = L L' (once per instance),   K = I + L' S L,   S = DD Dg^-1 DD'  (6 x 6 per stage)
//     (H + Sigma) x = r   <=>   x = Dg^-1 ( r - DD' L K^-1 L' DD Dg^-1 r )
// * condensing on the matrix cores with D_a = I (the columns are stage-wrench components): kernel 2's stage loop, n_a = 6;
// * H_w factorised once by kernel 2's register-resident blocked Cholesky, which also leaves the tiles of L in LDS
//   (in place of the Hessian tiles: L_IJ = C_IJ W_J', one more tile product per tile);
// * per interior-point iteration  P = S L  (tiles in registers; the operand tiles of the block-tridiagonal S are read off
//   the 6 x 6 stage blocks, one per (K, M) pair),  -X' = -(P' L)  (accumulated in registers: they ARE the seeds of the
//   factorisation),  Cholesky of I + X and two solves on registers (chol_reg / solve_reg), the two triangular products with
//   L as VALU mat-vecs over the LDS tiles;
// * the Mehrotra iteration in the N n_a thruster variables (four or six per lane), gradient by recurrence with the float64
//   structured reference gradient (once for N <= 16, at every iterate below mu = 1e-3 beyond: see kernel 8).
// One wave per SIMD (512 registers); LDS 33 KiB (six tiles a side: N <= 16) or 50 KiB (eight: N <= 21) per wave.
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354; oracle/qp_oracle.py:ipm_box is the mirror.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace wswk {
__host__ __device__ constexpr int nvt_of(int nbw) { return nbw <= 6 ? 4 : 6; }     // thruster variables per lane
// per-workgroup global slot (4-byte words): float64 scratch of the reference gradient (gradient | wrenches | stage storage)
// tiles of L kept in LDS: all 21 of the six-tile instantiation; 24 of the 36 of the eight-tile one -- the last twelve (block row 7
// and most of row 6) live in the global slot, read through the Infinity Cache, so that FOUR waves fit a CU's LDS instead of three
__host__ __device__ constexpr int nlds_of(int nbw) { return nbw <= 6 ? nbw * (nbw + 1) / 2 : 24; }
__host__ __device__ constexpr int64_t slot_f64_part(int nbw, int N) { return ((wgk::slot_f64_words(64 * nvt_of(nbw), N) + 255) / 256) * 256; }
__host__ __device__ constexpr int64_t slot_words(int nbw, int N) { return slot_f64_part(nbw, N) + (int64_t)(nbw * (nbw + 1) / 2 - nlds_of(nbw)) * 256; }
// the factorisation reads its seeds (-M' tiles) from registers
template <int NT>
struct RegTiles {
    const f32x4 (&X)[NT];
    static constexpr bool is_global(int) { return false; }
    __device__ __forceinline__ f32x4 ld(int tile, int) const { return X[tile]; }
};
}  // namespace wswk

template <int NBW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) ftmpc_solve_wsw32_kernel(const DeviceConsts C, const SolveParams P) {
    using namespace wswk;
    constexpr int NPADW = 16 * NBW;
    constexpr int NTW = NBW * (NBW + 1) / 2;
    constexpr int NVW = (NPADW + 63) / 64;          // wrench columns per lane
    constexpr int NVT = nvt_of(NBW);                // thruster variables per lane
    constexpr int NTP = 64 * NVT;
    constexpr int NSTG = NPADW / 6;
    constexpr int BUILD_WORDS = 2 * DENSE_WORDS + 256;
    static_assert(NTW * 256 >= BUILD_WORDS, "the dense stage-matrix images live in the tile area during the build");
    constexpr int NLDSW = nlds_of(NBW);
    static_assert(NLDSW * 256 >= BUILD_WORDS, "the dense stage-matrix images live in the LDS tile area during the build");
    __shared__ __attribute__((aligned(16))) float Ltl[NLDSW * 256];     // -H_w' tiles (build) -> tiles of L (H_w = L L'); the rest: global slot
    __shared__ __attribute__((aligned(16))) float recbuf[2 * REC_STRIDE + 8];
    __shared__ __attribute__((aligned(16))) float xvp[NPADW], dvp[NPADW], twv[NPADW], yvv[NPADW];
    __shared__ __attribute__((aligned(16))) float rv[NTP];
    __shared__ __attribute__((aligned(16))) float rdg[NSTG * MAX_NT];   // 1 / Dg of thruster a of stage k at k * 16 + a, zero beyond the healthy ones
    __shared__ __attribute__((aligned(16))) float rv16[NSTG * MAX_NT];  // operand of the wrench images, same layout
    __shared__ __attribute__((aligned(16))) double sSl[9 * (NSTG + 2)];
    // stage blocks S_k, rows padded to eight words behind eight zero words: Sblk[8 + 48 k + 8 g + h], words 6 and 7 of a row stay zero
    __shared__ __attribute__((aligned(16))) float Sblk[8 + NSTG * 48];
    __shared__ __attribute__((aligned(16))) float s_DD[21 * MAX_NT];
    __shared__ __attribute__((aligned(16))) float s_DaT[6 * MAX_NT];
    __shared__ __attribute__((aligned(16))) float mtab[MAX_NT * MAX_NT];
    __shared__ unsigned char s_stg[NPADW], s_thr[NPADW];
    __shared__ unsigned char s_pg[24], s_ph[24];     // pair p = g (g + 1) / 2 + h of the symmetric 6 x 6 stage blocks -> (g, h)
    __shared__ int s_act[MAX_NT];
    float* const dense = Ltl;

    const int lane0 = threadIdx.x;
    const int N = C.N, NT = C.NT;
    TileStore<NLDSW> ltiles;
    ltiles.p = Ltl;
    ltiles.bind(P.hscratch + (int64_t)blockIdx.x * P.tile_words + slot_f64_part(NBW, C.N));      // tiles [NLDSW, NTW) behind the float64 scratch
    const float rho = (float)C.rho;
    const float mu_stop = (float)C.mu_stop;
    double* const sbuf = reinterpret_cast<double*>(P.hscratch + (int64_t)blockIdx.x * P.tile_words);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

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
    // Operand tiles of S for P = S L: lane (q, col) of pair (M, K) holds S[16 K + 4 q + r][16 M + col], r < 4 -- four consecutive
    // words of one padded block row (S_k is symmetric), or of two when the four rows straddle a stage boundary (then the column's
    // stage picks the half, the other half reads the zero padding), or the zero words in front.  Word offsets, fixed per launch.
    constexpr int NPAIR = 3 * NBW - 2;
    constexpr bool SKTAB = NBW <= 6;     // the eight-tile instantiation has no registers to spare: it computes the offsets where it uses them
    auto sk_offset = [&](int M, int K, int lq_, int li_) -> int {
        const int e1 = 16 * K + 4 * lq_, s1 = (e1 * 43) >> 8, j4 = e1 - 6 * s1;       // e / 6 for e < 128
        const int e2 = 16 * M + li_, s2 = (e2 * 43) >> 8, a2 = e2 - 6 * s2;
        int off = 0;
        if (e2 < 6 * N) {
            if (s1 == s2) off = 8 + 48 * s1 + 8 * a2 + j4;
            else if (j4 == 4 && s2 == s1 + 1) off = 8 + 48 * s2 + 8 * a2 - 2;
        }
        return off;
    };
    int skoff[SKTAB ? NPAIR : 1];
    if constexpr (SKTAB) {
        int pi = 0;
#pragma unroll
        for (int M = 0; M < NBW; ++M) {
#pragma unroll
            for (int dk = -1; dk <= 1; ++dk) {
                const int K = M + dk;
                if (K < 0 || K >= NBW) continue;
                skoff[pi++] = sk_offset(M, K, lane0 >> 4, lane0 & 15);
            }
        }
    }
    for (int i = lane0; i < 8 + NSTG * 48; i += 64) Sblk[i] = 0.f;
    const int qn = *P.qcount;
    int qnext = pull();
    for (;;) {
        const int qi = __builtin_amdgcn_readfirstlane(qnext);
        if (qi >= qn) break;
        const int64_t inst = P.qlist[qi];
        qnext = pull();
        STAMP_DECL;
        STAMP_START();
        wave_lds_fence();
        int lane = lane_now();
        int li = lane & 15, lq = lane >> 4;
        // ---------------- prologue ----------------
        double ub_l = 0.0;
        if (lane < NT) ub_l = P.ub[inst * NT + lane];
        const unsigned long long amask = __ballot(lane < NT && ub_l > 0.0);
        const int nat = __popcll(amask);          // healthy thrusters
        const int nt = N * nat;                   // thruster-space variables
        constexpr int na = 6;                     // wrench components: the columns of the condensing
        const int n = N * na;
        const int nb = (n + 15) >> 4;
        if (nat == 0 || nb > NBW || nt > NTP) {
            if (lane < NT) P.out_u0[inst * NT + lane] = 0.0;
            if (P.out_U)
                for (int i = lane; i < N * NT; i += 64) P.out_U[inst * N * NT + i] = 0.0;
            if (lane == 0) {
                if (P.status) P.status[inst] = (nat == 0) ? 0 : 2;
                if (P.iters) P.iters[inst] = 0;
            }
            continue;
        }
        const int myrank = __popcll(amask & ((1ull << lane) - 1ull));
        if (lane < NT && ub_l > 0.0) s_act[myrank] = lane;
        wave_lds_fence();
        for (int i = lane; i < 6 * MAX_NT; i += 64) {
            const int g = i / MAX_NT, a = i % MAX_NT;
            s_DaT[i] = (a < nat) ? (float)C.D[g * MAX_NT + s_act[a]] : 0.f;
        }
        for (int e = lane; e < NPADW; e += 64) {
            const int s = e / na;
            s_stg[e] = (unsigned char)(e < n ? s : 255);
            s_thr[e] = (unsigned char)(e < n ? e - s * na : 255);
        }
        for (int t = lane; t < MAX_NT * MAX_NT; t += 64) {     // stage block of H_w: 2 R on the diagonal (the rho term stays in thruster space)
            const int a1 = t >> 4, a2 = t & (MAX_NT - 1);
            float r = 0.f;
#pragma unroll
            for (int g = 0; g < 6; ++g) r = (a1 == g) ? (float)C.R[g] : r;
            mtab[t] = (a1 == a2 && a1 < 6) ? 2.f * r : 0.f;
        }
        wave_lds_fence();
        for (int e = lane; e < 21 * MAX_NT; e += 64) {   // products of the rows of D_a, pair p = g (g + 1) / 2 + h
            const int p = e / MAX_NT, a = e % MAX_NT;
            int g = 0;
            while ((g + 1) * (g + 2) / 2 <= p) ++g;
            const int hh = p - g * (g + 1) / 2;
            s_DD[e] = (a < nat) ? s_DaT[g * MAX_NT + a] * s_DaT[hh * MAX_NT + a] : 0.f;
        }
#define FTMPC_WB_PART 1
#include "ftmpc_wrench_build.inc"
#undef FTMPC_WB_PART

        // thruster-space role of this lane: variables e = v * 64 + lane = (stage tk, healthy thruster ta)
        bool tvalid[NVT];
        int tk[NVT], ta[NVT];
        float ubar[NVT], ubv[NVT];
        for (int i = lane; i < NSTG * MAX_NT; i += 64) {
            rdg[i] = 0.f;
            rv16[i] = 0.f;
        }
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            const int e = v * 64 + lane;
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
        }
#define FTMPC_WB_PART 2
#define FTMPC_WB_TILES ltiles
#include "ftmpc_wrench_build.inc"
#undef FTMPC_WB_TILES
#undef FTMPC_WB_PART
        wave_lds_fence();   // the dense images in the tile area are dead from here: Ltl holds the -H_w' tiles
        if constexpr (NLDSW < NTW) wave_global_fence();     // ... and the global slot the rest: the stores have landed before the loads below

        // ---- helpers on LDS vectors (natural order) ----
        // wrench image of a thruster-space vector held NVT per lane: out[w] = sum_a D_a[g][a] x[k nat + a]
        auto to_wrench = [&](const float (&x)[NVT], float* out) {
            const int lane = lane_now();
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) rv16[tk[v] * MAX_NT + ta[v]] = x[v];
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVW; ++v) {
                const int e = v * 64 + lane;
                if (e < NPADW) {
                    float s = 0.f;
                    if (e < n) {
                        const int k = (e * 10923) >> 16, g = e - 6 * k;      // e / 6 for e < 4096
                        const f32x4* d4 = reinterpret_cast<const f32x4*>(s_DaT + g * MAX_NT);     // zero beyond the healthy thrusters
                        const f32x4* r4 = reinterpret_cast<const f32x4*>(rv16 + k * MAX_NT);
#pragma unroll
                        for (int a4 = 0; a4 < MAX_NT / 4; ++a4) {
                            const f32x4 d = d4[a4], r = r4[a4];
                            s += (d.x * r.x + d.y * r.y) + (d.z * r.z + d.w * r.w);
                        }
                    }
                    out[e] = s;
                }
            }
            wave_lds_fence();
        };
        // out = L' in (TRANS) or L in, tiles of L from LDS (lane (q, col), register s: L_IJ[4q + s][col]); natural order in and out
        auto tri_mv = [&](auto TRANS, const float* in, float* out) {
            const int lane = lane_now();
            const int li = lane & 15, lq = lane >> 4;
            if constexpr (decltype(TRANS)::value) {      // (L' in)_J[col] = sum_{I >= J} sum_r L_IJ[r][col] in_I[r]
                float acc[NBW];
#pragma unroll
                for (int J = 0; J < NBW; ++J) acc[J] = 0.f;
#pragma unroll
                for (int I = 0; I < NBW; ++I) {
                    const f32x4 v4 = lds4(in + 16 * I + 4 * lq);
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        const f32x4 l4 = ltiles.ld(tidx(I, J), lane);
                        acc[J] += (l4.x * v4.x + l4.y * v4.y) + (l4.z * v4.z + l4.w * v4.w);
                    }
                }
                wave_lds_fence();
#pragma unroll
                for (int J = 0; J < NBW; ++J) {
                    const float r = quad_sum(acc[J]);
                    if (lq == 0) out[16 * J + li] = r;
                }
            } else {                                      // (L in)_I[4q + s] = sum_{J <= I} sum_col L_IJ[4q + s][col] in_J[col]
#pragma unroll
                for (int I = 0; I < NBW; ++I) {
                    f32x4 a = zero4;
#pragma unroll
                    for (int J = 0; J <= I; ++J) {
                        const f32x4 l4 = ltiles.ld(tidx(I, J), lane);
                        a += l4 * in[16 * J + li];
                    }
                    float c0 = a.x, c1 = a.y, c2 = a.z, c3 = a.w;
                    row_sum16x4(c0, c1, c2, c3);
                    if (li == 0) *reinterpret_cast<f32x4*>(out + 16 * I + 4 * lq) = f32x4{c0, c1, c2, c3};
                }
            }
            wave_lds_fence();
        };

        // ---------------- start point and its gradient: DD' (g_w + H_w DD d) + 2 rho (ubar + d) ----------------
        float lo[NVT], hi[NVT], sl[NVT], su[NVT], zl[NVT], zu[NVT], grad[NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            lo[v] = -ubar[v];
            hi[v] = ubv[v] - ubar[v];
            sl[v] = su[v] = 0.5f * ubv[v];
            zl[v] = zu[v] = grad[v] = 0.f;
        }
        {
            float dcur[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) dcur[v] = tvalid[v] ? lo[v] + sl[v] : 0.f;
            to_wrench(dcur, dvp);
            lane = lane_now();
            li = lane & 15;
            lq = lane >> 4;
            // y = H_w d_w from the -H_w' tiles (one read serves both triangles: ftmpc_solve.hip, start gradient)
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
                    const f32x4 t4 = ltiles.ld(tidx(I, J), lane);
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
                if (e < NPADW) yvv[e] = -(t + xvp[e]) + gw[v];      // the tiles hold -H_w
            }
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < 6; ++g) s += s_DaT[g * MAX_NT + ta[v]] * yvv[tk[v] * 6 + g];
                grad[v] = tvalid[v] ? s + 2.f * rho * (ubar[v] + dcur[v]) : 0.f;
            }
        }
        STAMP(3);
        // ---------------- H_w = L L' once: kernel 2's factorisation, the tiles of L written over the Hessian tiles ----------------
        f32x4 Tt[NTW], Wd[NBW];
        {
            for (int e = lane_now(); e < NPADW; e += 64) dvp[e] = 0.f;
            wave_lds_fence();
            f32x4 pre0[NBW];
            chol_prefetch_col0<NBW>(ltiles, lane, pre0);      // (zero for tiles that live in LDS)
            const bool ok = chol_reg<NBW, TileStore<NLDSW>, true, false, true>(ltiles, dvp, recbuf, n, lane, Tt, Wd, pre0, nullptr);
            if constexpr (NLDSW < NTW) wave_global_fence();   // the tiles of L written to the global slot
            if (__builtin_amdgcn_readfirstlane(!ok)) {     // H_w not positive definite in fp32: report the linearisation point
                wave_lds_fence();
                float* ub0 = Ltl;
                for (int i = lane; i < N * NT; i += 64) ub0[i] = 0.f;
                wave_lds_fence();
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    if (tvalid[v]) ub0[tk[v] * NT + s_act[ta[v]]] = ubar[v];
                wave_lds_fence();
                if (lane < NT) P.out_u0[inst * NT + lane] = (double)ub0[lane];
                if (P.out_U)
                    for (int i = lane; i < N * NT; i += 64) P.out_U[inst * (int64_t)N * NT + i] = (double)ub0[i];
                if (lane == 0) {
                    if (P.status) P.status[inst] = 2;
                    if (P.iters) P.iters[inst] = 0;
                }
                continue;
            }
            for (int e = lane_now(); e < NPADW; e += 64) dvp[e] = 1.f;     // the identity of K = I + L' S L, for every later factorisation
            wave_lds_fence();
        }
        STAMP(4);

        // ---- Newton system in thruster space through wrench space ----
        auto ws_solve = [&](const float (&r)[NVT], float (&x)[NVT]) {
            const int lane = lane_now();
            float t[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) t[v] = r[v] * rdg[tk[v] * MAX_NT + ta[v]];
            to_wrench(t, twv);
            tri_mv(std::true_type{}, twv, xvp);
            solve_reg<NBW>(Tt, Wd, xvp, NBW, lane);
            tri_mv(std::false_type{}, xvp, yvv);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < 6; ++g) s += s_DaT[g * MAX_NT + ta[v]] * yvv[tk[v] * 6 + g];
                x[v] = tvalid[v] ? (r[v] - s) * rdg[tk[v] * MAX_NT + ta[v]] : 0.f;
            }
        };

        // ---------------- interior-point iterations (thruster space) ----------------
        int status = 1, nit = 0;
        bool first = true;
        int refines_left = (C.mu_refine > 0.0) ? ((NBW > 6) ? C.max_iters + 1 : 1) : 0;
        float mu_last = 3.0e38f;
        const float inv2n = 1.0f / (float)(2 * nt);
        for (int it = 0; it <= C.max_iters; ++it) {
            wave_lds_fence();
            lane = lane_now();
            li = lane & 15;
            lq = lane >> 4;
            const bool do_ref = __builtin_amdgcn_readfirstlane(refines_left > 0 && mu_last < (float)C.mu_refine);
            if (do_ref) {   // float64, structured, at the current iterate
                float dcur[NVT];
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    dcur[v] = tvalid[v] ? ((sl[v] < su[v]) ? lo[v] + sl[v] : hi[v] - su[v]) : 0.f;
                    rv[v * 64 + lane] = dcur[v];
                }
                wave_lds_fence();
                struct_grad<lds_f64*, NTP>(C, (glb_cf64*)recg, (lds_f64*)reinterpret_cast<double*>(recbuf), (lds_cf32*)s_DaT, (lds_cf32*)rv,
                                          (lds_f64*)sSl, (glb_f64*)sbuf, nat, lane);
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    grad[v] = tvalid[v] ? (float)(sbuf[v * 64 + lane] + 2.0 * C.rho * ((double)ubar[v] + (double)dcur[v])) : 0.f;
                --refines_left;
                STAMP(8);
            }
            if (first) {
                float gm = 0.f, wm = 0.f;
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    gm = fmaxf(gm, tvalid[v] ? fabsf(grad[v]) : 0.f);
                    wm = fmaxf(wm, tvalid[v] ? ubv[v] : 0.f);
                }
                gm = wave_max(gm);
                wm = wave_max(wm);
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
            const float mu = wave_sum(csum) * inv2n;
            mu_last = mu;
            if (__builtin_amdgcn_readfirstlane(!(mu >= mu_stop))) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            float rsl[NVT], rsu[NVT], Sig[NVT];
            wave_lds_fence();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                rsl[v] = __builtin_amdgcn_rcpf(sl[v]);
                rsu[v] = __builtin_amdgcn_rcpf(su[v]);
                Sig[v] = tvalid[v] ? zl[v] * rsl[v] + zu[v] * rsu[v] : 0.f;
                if (tvalid[v]) rdg[tk[v] * MAX_NT + ta[v]] = 1.0f / (2.f * rho + Sig[v]);
            }
            wave_lds_fence();
            for (int idx = lane; idx < N * 21; idx += 64) {          // stage blocks S_k = D_a diag(1 / Dg) D_a'
                const int k = (idx * 3121) >> 16, p = idx - 21 * k;      // idx / 21 for idx < 5000
                const int g = s_pg[p], hh = s_ph[p];
                // all sixteen products at once (s_DD is zero beyond the healthy thrusters, the index stays inside rdg): a loop
                // over the healthy ones alone is a chain of dependent LDS round trips
                const f32x4* dd4 = reinterpret_cast<const f32x4*>(s_DD + p * MAX_NT);
                const f32x4* r4 = reinterpret_cast<const f32x4*>(rdg + k * MAX_NT);
                float sacc = 0.f;
#pragma unroll
                for (int a4 = 0; a4 < MAX_NT / 4; ++a4) {
                    const f32x4 d = dd4[a4], r = r4[a4];
                    sacc += (d.x * r.x + d.y * r.y) + (d.z * r.z + d.w * r.w);
                }
                Sblk[8 + k * 48 + g * 8 + hh] = sacc;
                Sblk[8 + k * 48 + hh * 8 + g] = sacc;
            }
            wave_lds_fence();
            STAMP(7);
            // ---- K - I = X = L' S L.  P = S L (lower tiles, registers): P_MJ += S_KM' L_KJ over K in {M-1, M, M+1}, J <= min(K, M);
            // the operand tile S_KM (rows of block K, columns of block M) is read off the stage blocks once per (K, M) ----
            f32x4 Xt[NTW];
            {
                f32x4 Pt[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) Pt[t] = zero4;
                f32x4 skv[SKTAB ? NPAIR : 1];
                if constexpr (SKTAB) {
#pragma unroll
                    for (int pi = 0; pi < NPAIR; ++pi) {
                        const f32x2 lo2 = *reinterpret_cast<const f32x2*>(Sblk + skoff[pi]);
                        const f32x2 hi2 = *reinterpret_cast<const f32x2*>(Sblk + skoff[pi] + 2);
                        skv[pi] = f32x4{lo2.x, lo2.y, hi2.x, hi2.y};
                    }
                }
                {
                    int pi = 0;
#pragma unroll
                    for (int M = 0; M < NBW; ++M) {
#pragma unroll
                        for (int dk = -1; dk <= 1; ++dk) {
                            const int K = M + dk;
                            if (K < 0 || K >= NBW) continue;
                            f32x4 sk;
                            if constexpr (SKTAB) {
                                sk = skv[pi++];
                            } else {
                                const int off = sk_offset(M, K, lq, li);
                                const f32x2 lo2 = *reinterpret_cast<const f32x2*>(Sblk + off);
                                const f32x2 hi2 = *reinterpret_cast<const f32x2*>(Sblk + off + 2);
                                sk = f32x4{lo2.x, lo2.y, hi2.x, hi2.y};
                            }
#pragma unroll
                            for (int J = 0; J < NBW; ++J)
                                if (J <= K && J <= M) Pt[tidx(M, J)] = mm_tn(sk, ltiles.ld(tidx(K, J), lane), Pt[tidx(M, J)]);
                        }
                    }
                }
                STAMP(5);
                // -X_IJ' = -(sum_{M >= I} P_MJ' L_MI): the seeds of the factorisation, in registers
#pragma unroll
                for (int t = 0; t < NTW; ++t) Xt[t] = zero4;
#pragma unroll
                for (int I = 0; I < NBW; ++I) {
#pragma unroll
                    for (int M = I; M < NBW; ++M) {
                        const f32x4 lmi = ltiles.ld(tidx(M, I), lane);
#pragma unroll
                        for (int J = 0; J <= I; ++J) Xt[tidx(I, J)] = mm_tn(Pt[tidx(M, J)], lmi, Xt[tidx(I, J)]);
                    }
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) Xt[t] = -Xt[t];
            }
            STAMP(6);
            {
                f32x4 pre0[NBW];
#pragma unroll
                for (int I = 0; I < NBW; ++I) pre0[I] = zero4;
                const RegTiles<NTW> xt{Xt};
                const bool ok = chol_reg<NBW, RegTiles<NTW>, false, false, false>(xt, dvp, recbuf, n, lane, Tt, Wd, pre0, nullptr);
                if (__builtin_amdgcn_readfirstlane(!ok)) {
                    status = 2;
                    break;
                }
            }
            STAMP(9);
            // predictor: (H + Sig) da = -grad
            float ngrad[NVT], da[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) ngrad[v] = -grad[v];
            ws_solve(ngrad, da);
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
            ap = wave_min(ap);
            ad = wave_min(ad);
            csum = 0.f;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                csum += tvalid[v] ? (sl[v] + ap * da[v]) * (zl[v] + ad * dzl_a[v]) + (su[v] - ap * da[v]) * (zu[v] + ad * dzu_a[v]) : 0.f;
            const float mu_aff = wave_sum(csum) * inv2n;
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
            ws_solve(rhs, dd);
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
            ap = fminf(1.f, 0.9995f * wave_min(ap));
            ad = fminf(1.f, 0.9995f * wave_min(ad));
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (tvalid[v]) {
                    grad[v] += ap * (rhs[v] - Sig[v] * dd[v]);   // + ap H dd
                    sl[v] += ap * dd[v];
                    su[v] -= ap * dd[v];
                    zl[v] += ad * dzl[v];
                    zu[v] += ad * dzu[v];
                }
            STAMP(10);
        }
        // ---------------- outputs ----------------
        wave_lds_fence();
        lane = lane_now();
        float* ubuf = Ltl;  // N*NT words, zero = broken thruster
        for (int i = lane; i < N * NT; i += 64) ubuf[i] = 0.f;
        wave_lds_fence();
#pragma unroll
        for (int v = 0; v < NVT; ++v)
            if (tvalid[v]) {
                float u = (sl[v] < su[v]) ? sl[v] : ubv[v] - su[v];
                if (status == 2) u = ubar[v];
                ubuf[tk[v] * NT + s_act[ta[v]]] = u;
            }
        wave_lds_fence();
        if (lane < NT) P.out_u0[inst * NT + lane] = (double)ubuf[lane];
        if (P.out_U)
            for (int i = lane; i < N * NT; i += 64) P.out_U[inst * (int64_t)N * NT + i] = (double)ubuf[i];
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

template __global__ void ftmpc_solve_wsw32_kernel<6>(const DeviceConsts, const SolveParams);   // 6 N <= 96  (N <= 16: the reference horizon), N n_a <= 256
template __global__ void ftmpc_solve_wsw32_kernel<8>(const DeviceConsts, const SolveParams);   // 6 N <= 128 (N <= 21), N n_a <= 384

}  // namespace ftmpc
