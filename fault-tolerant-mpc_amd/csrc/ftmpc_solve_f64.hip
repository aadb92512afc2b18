// ftmpc_solve_f64.hip -- kernel 3: condensed-QP build + primal-dual IPM in float64 for ANY
// problem size (n = N * #healthy thrusters up to 1024), e.g. the reference's own 16-thruster
// vehicle at its shipped horizon (N=15: n=240) and BASELINE config 5 (N=40, NT=16, "fp64 KKT").
//
// The KKT matrix no longer fits LDS (n=640 fp64: 1.6 MB), so ONE WORKGROUP (4 wavefronts) owns
// an instance and its 16x16-tiled Hessian / factor live in a per-workgroup global slot (L2 /
// Infinity-Cache resident; MI355X has 288 GB of HBM so slots are simply preallocated per
// resident workgroup).  Same algorithm as ftmpc_solve.hip / oracle/qp_oracle.py:ipm_box:
//   1. condense: one column of G per THREAD, E_k = sqrt(2 W_k) G_k[0:9] panels -> global
//   2. H tiles = sum_k E_k' E_k on v_mfma_f64_16x16x4_f64, tiles distributed over the waves
//   3. Mehrotra IPM: left-looking blocked Cholesky (tile GEMMs on f64 MFMA, diagonal tiles
//      factorised + inverted in registers), blocked triangular solves, float64 throughout.
// Reference path replaced: ft_mpc/controllers/spiraling_mpc.py:87-238,319-354 (NLP + IPOPT) and
// controllers/tools/control_allocator.py:65-94 (see DESIGN.md QP-spec).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "ftmpc_common.h"

namespace ftmpc {

namespace f64k {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int WG = 256;       // threads per workgroup
constexpr int NWAVE = 4;
constexpr int NVT_MAX = 4;    // columns per thread at n = 1024
constexpr int NMAX = WG * NVT_MAX;
constexpr int RPF = 10;                  // block rows (columns) per wave whose tiles are prefetched in the sweeps (n <= 640)
constexpr int RMAXW = NMAX / 16 / NWAVE;  // block rows per wave at n = 1024; rows beyond RPF load their tiles at use

// operand layout of a 16x16 tile in global memory: the four k-steps a lane needs are contiguous
__device__ __forceinline__ int t64off(int r, int c) { return 16 * r + 4 * (c & 3) + (c >> 2); }
__device__ __forceinline__ int v64pos(int c) { return (c & 3) * 4 + (c >> 2); }
__device__ __forceinline__ int t64idx(int I, int J) { return (I * (I + 1)) / 2 + J; }

__device__ __forceinline__ double readlane_d(double x, int l) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// sum over the four lanes that share lane & 15: permlane swaps on both halves (ftmpc_solve.hip quad_sum_d)
// instead of four ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ double quad_sum64(double x) { return quad_sum_d(x); }
// owner wave of tile (J + t, J) of block column J: t = 0 is the diagonal (see the factorisation loop)
__device__ __forceinline__ int col_owner(int t) { return t == 0 ? 0 : (t <= 6 ? 1 + (t - 1) % 3 : (t - 7) & 3); }
// barrier for exchanges that go through LDS only: does not drain the outstanding global loads
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f64x4 mfma(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// workgroup reductions through LDS (red: NWAVE doubles)
__device__ __forceinline__ double wg_sum(double x, double* red, int tid) {
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ double wg_min(double x, double* red, int tid) {
    for (int m = 32; m >= 1; m >>= 1) x = fmin(x, __shfl_xor(x, m, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
}
__device__ __forceinline__ double wg_max(double x, double* red, int tid) {
    for (int m = 32; m >= 1; m >>= 1) x = fmax(x, __shfl_xor(x, m, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = x;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// 16x16 Cholesky + inverse in registers (one row per lane, see ftmpc_solve.hip potrf_inv16)
__device__ __forceinline__ bool potrf_inv16_f64(const double* S, int li, double w[16]) {
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = S[li * 17 + k];
    double invs[16];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double djj = readlane_d(a[j], j);
        ok = ok && (djj > 0.0);
        // 1/sqrt: hardware seed (v_rsq_f64, ~26 bits) + two Newton steps instead of the IEEE sqrt and divide
        // sequences (~70 instructions on the serial pivot chain of the one wave everybody waits for)
        double inv = __builtin_amdgcn_rsq(djj);
        inv = inv * (1.5 - 0.5 * djj * inv * inv);
        inv = inv * (1.5 - 0.5 * djj * inv * inv);
        invs[j] = inv;
        a[j] *= inv;
#pragma unroll
        for (int k = j + 1; k < 16; ++k) a[k] -= a[j] * readlane_d(a[j], k);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double s = (i == li) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= readlane_d(a[k], i) * w[k];
        w[i] = s * invs[i];
    }
    return ok;
}

__device__ __forceinline__ f64x4 ld4(const double* p) { return *reinterpret_cast<const f64x4*>(p); }

// 16x16 Cholesky + inverse of a diagonal tile held in the float64 MFMA C/D layout (lane (lq, li): rows lq + 4 rr, rr < 4, of
// column li) by ONE wave with all 64 lanes at work: right-looking elimination without scaling the pivot column (the float64
// form of ftmpc_solve.hip's potrf_inv16: after step j column j holds sqrt(d_j) L[:, j]), the inverse built alongside from
// E = I.  Per pivot the four lanes that hold column j publish it in LDS (colS, 16 doubles in v64pos order so that a lane
// reads its four rows with one 32-byte load) and the sixteen lanes that hold row j of E publish that (rowE); every lane
// reads the pivot, its rows, its column's entry and E[j][col], then eight FMAs.  The row-per-lane version above spends
// ~900 cycles per pivot on ~32 v_readlane of double words; this one two LDS round trips (~300).
// Out: w = W = L^-1 and l = L in the same layout (zero above the diagonal); returns false on a non-positive pivot.
// np (wave-uniform): rows and columns at or beyond it are identity padding and take no pivot step (W and L are the identity there).
__device__ __forceinline__ bool potrf_inv16_lds(double (&c)[4], double* colS, double* rowE, int lq, int li, double (&w)[4], double (&l)[4], int np = 16) {
    double e[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        e[rr] = (lq + 4 * rr == li) ? 1.0 : 0.0;
        w[rr] = (lq + 4 * rr >= np) ? e[rr] : 0.0;
        l[rr] = w[rr];
    }
    bool ok = true;
    const int pli = v64pos(li);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j >= np) break;
        if (li == j) *reinterpret_cast<f64x4*>(colS + 4 * lq) = f64x4{c[0], c[1], c[2], c[3]};
        if (lq == (j & 3)) rowE[li] = e[j >> 2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double d = colS[v64pos(j)];
        const f64x4 sc = *reinterpret_cast<const f64x4*>(colS + 4 * lq);
        const double sk = colS[pli], ek = rowE[li];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // the next pivot's stores stay behind these loads
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        ok = ok && (d > 0.0);
        double inv = __builtin_amdgcn_rsq(d);
        inv = inv * (1.5 - 0.5 * d * inv * inv);
        inv = inv * (1.5 - 0.5 * d * inv * inv);
        const double invd = inv * inv;
        if (li == j) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) l[rr] = (lq + 4 * rr >= j) ? sc[rr] * inv : 0.0;
        }
        if (lq == (j & 3)) w[j >> 2] = ek * inv;
        const double mk = sk * invd, me = ek * invd;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            c[rr] -= sc[rr] * mk;
            e[rr] -= sc[rr] * me;
        }
    }
    return ok;
}

}  // namespace f64k

struct Solve64Params {
    SolveParams base;     // rec is double here; hscratch unused
    double* Hs;           // [grid][tile_doubles]   Hessian tiles
    double* Ls;           // [grid][tile_doubles]   KKT factor tiles
    double* Eall;         // [grid][N*9*npad_max]   E panels of every stage
    int64_t tile_doubles;
    int64_t e_doubles;
    int32_t npad_max;
    int32_t nb_lo;        // instances with ceil(n/16) <= nb_lo belong to the fp32 kernels
    double* dbg_H;        // [npad*npad] or nullptr
    double* dbg_vec;      // [3*npad_max + 4]
    // general-constraint modes (template MODE: bit 0 = generalized-force variables with the per-stage input-hull
    // rows, bit 1 = terminal-set rows on e_N); reference: spiraling_mpc.py:133-137,175-177 and :199-202
    const double* warmG;      // [B*N*6] linearisation wrenches or nullptr (thrusters off: D stuck)
    const double* hullA;      // [sets][hull_rows*6]
    const int32_t* hull_set;  // [B] or nullptr (set 0)
    const double* hullb;      // [B*hull_rows]
    const double* termA;      // [term_rows*9]
    const double* termb;      // [term_rows]
    const double* eN;         // [B*9] terminal tracking error at the linearisation point (ftmpc_linearize.hip)
    double* out_tau0;         // [B*6]
    double* out_G;            // [B*N*6] or nullptr
    int32_t hull_rows;        // <= 32, N * hull_rows <= 1024
    int32_t term_rows;        // <= 80
};

// (diagnostic build only) per-phase cycle counters of wave 0, see scripts/stamps64.py
#ifdef FTMPC_STAMPS
#define S64_DECL unsigned long long s64_t0 = 0, s64_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define S64_START() s64_t0 = __builtin_amdgcn_s_memtime()
#define S64(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); s64_acc[i] += t_ - s64_t0; s64_t0 = t_; } while (0)
#else
#define S64_DECL
#define S64_START()
#define S64(i)
#endif
// SWEEP_ROWS: block rows per wave whose tiles the triangular sweeps prefetch.  4 serves n <= 256 (the reference
// vehicle) with small register arrays, f64k::RPF everything up to 640; the host picks by the handle's N * NT.
template <int SWEEP_ROWS, int NVT, int MODE = 0>   // NVT: columns per thread (n <= 256 NVT); MODE: see Solve64Params
__global__ void __launch_bounds__(f64k::WG, (SWEEP_ROWS == 4 && MODE == 0) ? 2 : 1) ftmpc_solve_f64_kernel(const DeviceConsts C, const Solve64Params Q) {
    using namespace f64k;
    static_assert(NVT >= 1 && NVT <= NVT_MAX, "columns per thread");
    constexpr bool GEN = (MODE & 1) != 0, TSET = (MODE & 2) != 0;
    static_assert(MODE == 0 || NVT == 1, "the general-constraint modes are built for n <= 256");
    const SolveParams& P = Q.base;
    __shared__ double recbuf[REC_STRIDE];
    __shared__ double dv[NMAX];        // d (permuted per 16-block) for the gradient mat-vec
    __shared__ double xv[NMAX];        // rhs / solution of the KKT solves (permuted)
    __shared__ double part[NWAVE * 16];
    __shared__ __attribute__((aligned(32))) double Sbuf[32];     // pivot column | row of E (potrf_inv16_lds)
    __shared__ double red[NWAVE];
    __shared__ float s_Da[6 * MAX_NT];
    __shared__ double s_MR[MAX_NT * MAX_NT];
    __shared__ unsigned char s_stg[NMAX], s_thr[NMAX];
    __shared__ int s_act[MAX_NT];
    __shared__ int s_flag;
    __shared__ double s_ctr[12];     // GEN: hull centre | D stuck
    // finished tiles of the current block row J (K < J), one conflict-free 32-byte slice per lane: every tile of column J
    // multiplies against them, so they are fetched from L2 once per column instead of once per tile
    __shared__ __attribute__((aligned(32))) double Pj[(16 * NVT - 1) * 256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int N = C.N, NT = C.NT;
    const double rho = C.rho;

    double* Hs = Q.Hs + (int64_t)blockIdx.x * Q.tile_doubles;
    double* Ls = Q.Ls + (int64_t)blockIdx.x * Q.tile_doubles;
    double* Eall = Q.Eall + (int64_t)blockIdx.x * Q.e_doubles;

    // instances: the whole batch, or (P.qlist != nullptr) the *P.qcount entries of a list an earlier kernel on the stream wrote
    // -- kernel 11 hands over the instances whose fp32 answer it does not certify (ftmpc_solve_hull.hip)
    const int64_t n_inst = P.qlist ? (int64_t)*P.qcount : P.B;
    for (int64_t qi = blockIdx.x; qi < n_inst; qi += gridDim.x) {
        const int64_t inst = P.qlist ? (int64_t)P.qlist[qi] : qi;
        __syncthreads();
        S64_DECL;
        S64_START();
        // ---------------- prologue ----------------
        if (tid == 0) {
            int na0 = 0;
            if constexpr (GEN) {
                for (int i = 0; i < 6; ++i) s_act[i] = i;      // the six wrench components are the variables
                na0 = 6;
            } else {
                for (int i = 0; i < NT; ++i)
                    if (P.ub[inst * NT + i] > 0.0) s_act[na0++] = i;
            }
            s_flag = na0;
        }
        if constexpr (GEN) {
            // hull centre D (ub/2 + stuck) (the start point: strictly inside every hull row) and the
            // thrusters-off wrench D stuck (the cold-start linearisation point)
            if (tid < 12) {
                const int g = tid % 6;
                double acc = 0.0;
                for (int i = 0; i < NT; ++i)
                    acc += C.D[g * MAX_NT + i] * ((tid < 6 ? 0.5 * P.ub[inst * NT + i] : 0.0) + P.stuck[inst * NT + i]);
                s_ctr[tid] = acc;
            }
        }
        __syncthreads();
        const int na = s_flag;
        const int n = N * na;
        const int nb = (n + 15) >> 4;
        const int npad = nb * 16;
        if (nb <= Q.nb_lo && na != 0) continue;   // the fp32 LDS kernels own this instance
        if (na == 0 && Q.nb_lo != 0) continue;    // ... including the empty ones
        if (na == 0 || npad > Q.npad_max) {
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
            if constexpr (GEN) s_Da[tid] = (a == g) ? 1.f : 0.f;
            else s_Da[tid] = (a < na) ? (float)C.D[g * MAX_NT + s_act[a]] : 0.f;
        }
        for (int e = tid; e < npad; e += WG) {
            const int s = e / na;
            s_stg[e] = (unsigned char)(e < n ? s : 255);
            s_thr[e] = (unsigned char)(e < n ? e - s * na : 255);
        }
        __syncthreads();
        if (tid < na * na) {
            const int a = tid / na, b = tid % na;
            double t = 0.0;
            if constexpr (GEN) {
                t = (a == b) ? C.R[a] : 0.0;      // the input cost acts on the wrench itself; no allocation regulariser
            } else {
                for (int g = 0; g < 6; ++g) t += C.D[g * MAX_NT + s_act[a]] * C.R[g] * C.D[g * MAX_NT + s_act[b]];
                t += (a == b ? rho : 0.0);
            }
            s_MR[a * MAX_NT + b] = 2.0 * t;
        }
        const double* recg = reinterpret_cast<const double*>(P.rec) + inst * (int64_t)N * REC_STRIDE;

        int kcol[NVT], acol[NVT];
        double ubar[NVT], ubv[NVT], gacc[NVT];
        double G[13][NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            const int e = v * WG + tid;
            kcol[v] = (e < npad) ? s_stg[e] : 255;
            acol[v] = (e < npad) ? s_thr[e] : 255;
            ubar[v] = 0.0;
            ubv[v] = 1.0;
            gacc[v] = 0.0;
            if (kcol[v] != 255) {
                if constexpr (GEN) {
                    ubar[v] = Q.warmG ? Q.warmG[(inst * N + kcol[v]) * 6 + acol[v]] : s_ctr[6 + acol[v]];
                } else {
                    const int t = s_act[acol[v]];
                    ubv[v] = P.ub[inst * NT + t];
                    if (P.warmU) ubar[v] = fmin(fmax(P.warmU[(inst * N + kcol[v]) * NT + t], 0.0), ubv[v]);
                }
            }
#pragma unroll
            for (int r = 0; r < 13; ++r) G[r][v] = 0.0;
        }

        S64(0);
        // ---------------- phase 1: condense, E panels of every stage -> global ----------------
        for (int k = 0; k < N; ++k) {
            __syncthreads();
            if (tid < REC_STRIDE) recbuf[tid] = recg[k * REC_STRIDE + tid];
            __syncthreads();
            const double* rb = recbuf;
            const bool terminal = (k + 1 == N);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                if (e >= npad) continue;
                if (kcol[v] < k) {
                    double p[3], vv[3], w[3], q[4];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        p[a] = G[a][v] + C.dt * G[3 + a][v];
                        vv[a] = G[3 + a][v];
                        w[a] = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            p[a] += rb[REC_APW + 3 * a + c] * G[6 + c][v];
                            vv[a] += rb[REC_AVW + 3 * a + c] * G[6 + c][v];
                            w[a] += rb[REC_AWW + 3 * a + c] * G[6 + c][v];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            p[a] += rb[REC_APQ + 4 * a + c] * G[9 + c][v];
                            vv[a] += rb[REC_AVQ + 4 * a + c] * G[9 + c][v];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        q[a] = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) q[a] += rb[REC_AQW + 3 * a + c] * G[6 + c][v];
#pragma unroll
                        for (int c = 0; c < 4; ++c) q[a] += rb[REC_AQQ + 4 * a + c] * G[9 + c][v];
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        G[a][v] = p[a];
                        G[3 + a][v] = vv[a];
                        G[6 + a][v] = w[a];
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) G[9 + a][v] = q[a];
                } else if (kcol[v] == k) {
                    double F[3], T[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if constexpr (GEN) {
                            F[a] = (acol[v] == a) ? 1.0 : 0.0;
                            T[a] = (acol[v] == 3 + a) ? 1.0 : 0.0;
                        } else {
                            F[a] = C.D[a * MAX_NT + s_act[acol[v]]];
                            T[a] = C.D[(3 + a) * MAX_NT + s_act[acol[v]]];
                        }
                    }
                    double gr = 0.0;
#pragma unroll
                    for (int a = 0; a < 3; ++a) gr += F[a] * rb[REC_RUT + a] + T[a] * rb[REC_RUT + 3 + a];
                    gacc[v] += gr;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        double sp = 0.0, sv = 0.0, sw = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            sp += rb[REC_BPF + 3 * a + c] * F[c] + rb[REC_BPT + 3 * a + c] * T[c];
                            sv += rb[REC_BVF + 3 * a + c] * F[c] + rb[REC_BVT + 3 * a + c] * T[c];
                            sw += rb[REC_BWT + 3 * a + c] * T[c];
                        }
                        G[a][v] = sp;
                        G[3 + a][v] = sv;
                        G[6 + a][v] = sw;
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        double sq = 0.0;
#pragma unroll
                        for (int c = 0; c < 3; ++c) sq += rb[REC_BQT + 3 * a + c] * T[c];
                        G[9 + a][v] = sq;
                    }
                }
                double gs = 0.0;
#pragma unroll
                for (int r = 0; r < 9; ++r) gs += G[r][v] * rb[REC_WE + r];
                gacc[v] += gs;
                double* Ek = Eall + (int64_t)k * 9 * npad;
                if (!terminal) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) Ek[r * npad + e] = C.sq2Q[r] * G[r][v];
                } else {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int c = r; c < 9; ++c) s += C.LPt[9 * r + c] * G[c][v];
                        Ek[r * npad + e] = s;
                    }
                    if constexpr (TSET) {   // raw terminal sensitivity rows for the terminal-set rows A_T (e_N + GN d)
                        double* GNs = Eall + (int64_t)N * 9 * npad;
#pragma unroll
                        for (int r = 0; r < 9; ++r) GNs[r * npad + e] = G[r][v];
                    }
                }
            }
        }
        __syncthreads();  // E panels visible to the whole workgroup

        S64(1);
        // ---------------- phase 2: H tiles on f64 MFMA (tiles round-robin over the waves) ----------------
        const int ntl = (nb * (nb + 1)) / 2;
        for (int t = wave; t < ntl; t += NWAVE) {
            int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (t64idx(I + 1, 0) <= t) ++I;
            while (t64idx(I, 0) > t) --I;
            const int J = t - t64idx(I, 0);
            f64x4 acc = {0.0, 0.0, 0.0, 0.0};
            const int kstart = (16 * I) / na < N ? (16 * I) / na : N;
            for (int k = kstart; k < N; ++k) {
                const double* Ek = Eall + (int64_t)k * 9 * npad;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int r = 4 * s + lq;
                    const double a = (r < 9) ? Ek[r * npad + 16 * I + li] : 0.0;
                    const double b = (r < 9) ? Ek[r * npad + 16 * J + li] : 0.0;
                    acc = mfma(a, b, acc);
                }
            }
            // + 2 (Da' R Da + rho I) on same-stage pairs; unit diagonal on the padding
            const int e2 = 16 * J + li;
            const int s2 = s_stg[e2], a2 = s_thr[e2];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = lq + 4 * rr;      // f64 MFMA C/D layout: row = (lane>>4) + 4*reg
                const int e1 = 16 * I + row;
                const int s1 = s_stg[e1], a1 = s_thr[e1];
                double add = 0.0;
                if (s1 != 255 && s1 == s2) add = s_MR[a1 * MAX_NT + a2];
                if (s1 == 255 && e1 == e2) add = 1.0;
                Hs[(int64_t)t * 256 + t64off(row, li)] = acc[rr] + add;
            }
        }
        __syncthreads();
        if (P.dbg_inst == inst && Q.dbg_H) {
            for (int t = wave; t < ntl; t += NWAVE) {
                int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                while (t64idx(I + 1, 0) <= t) ++I;
                while (t64idx(I, 0) > t) --I;
                const int J = t - t64idx(I, 0);
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = lq + 4 * rr;
                    const int e1 = 16 * I + row, e2 = 16 * J + li;
                    const double h = Hs[(int64_t)t * 256 + t64off(row, li)];
                    if (I != J || e1 >= e2) {
                        Q.dbg_H[(int64_t)e1 * npad + e2] = h;
                        Q.dbg_H[(int64_t)e2 * npad + e1] = h;
                    }
                }
            }
        }

        double gv[NVT], lo[NVT], hi[NVT], sl[NVT], su[NVT], zl[NVT], zu[NVT], grad[NVT];
        bool valid[NVT];
#pragma unroll
        for (int v = 0; v < NVT; ++v) {
            valid[v] = kcol[v] != 255;
            gv[v] = valid[v] ? 2.0 * (gacc[v] + (GEN ? 0.0 : rho * ubar[v])) : 0.0;
            lo[v] = -ubar[v];
            hi[v] = ubv[v] - ubar[v];
            sl[v] = su[v] = 0.5 * ubv[v];
            grad[v] = 0.0;
            zl[v] = zu[v] = 0.0;
            if (P.dbg_inst == inst && Q.dbg_vec) {
                const int e = v * WG + tid;
                if (e < npad) {
                    Q.dbg_vec[e] = gv[v];
                    Q.dbg_vec[Q.npad_max + e] = lo[v];
                    Q.dbg_vec[2 * Q.npad_max + e] = hi[v];
                }
                if (tid == 0 && v == 0) {
                    Q.dbg_vec[3 * Q.npad_max] = (double)n;
                    Q.dbg_vec[3 * Q.npad_max + 1] = (double)npad;
                }
            }
        }

        S64(2);
        __shared__ double Sblk[GEN ? 43 * 36 : 1];     // hull rows: per-stage blocks of A' W A
        // ---- blocked left-looking Cholesky of the KKT matrix; block rows round-robin over the waves ----
        // KKT matrix = H (global tiles) + diag(dv) [box rows] + per-stage 6x6 blocks Sblk [hull rows, MODE & 1]
        //              + ET' ET [terminal rows, MODE & 2: ET = chol(A_T' W A_T)' GN, 9 x npad, global panel]
        auto factor = [&]() {
        // ---- blocked left-looking Cholesky; block rows round-robin over the waves ----
        if (tid == 0) s_flag = 1;
        for (int J = 0; J < nb; ++J) {
            __syncthreads();
            for (int K = wave; K < J; K += NWAVE)
                *reinterpret_cast<f64x4*>(Pj + K * 256 + 4 * lane) = ld4(Ls + (int64_t)t64idx(J, K) * 256 + 16 * li + 4 * lq);
            __syncthreads();
            S64(3);    // (diagnostic) barrier + row staging
            // wave 0 takes the diagonal tile and its potrf + inverse (the serial part of the column); the first six
            // off-diagonal tiles go round-robin over waves 1..3 (about the time of the potrf), the rest over all four
            const double* rowJ = Pj + 4 * lane;
            const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
            auto tile_init = [&](int I) {
                f64x4 acc = zero4;
                if constexpr (TSET) {   // acc = sum L L' - ET_I' ET_J, so that c = H - acc carries + ET' ET
                    const double* ETs = Eall + (int64_t)(N + 1) * 9 * npad;
#pragma unroll
                    for (int s3 = 0; s3 < 3; ++s3) {
                        const int r = 4 * s3 + lq;
                        const double ea = (r < 9) ? -ETs[r * npad + 16 * I + li] : 0.0;
                        const double eb = (r < 9) ? ETs[r * npad + 16 * J + li] : 0.0;
                        acc = mfma(ea, eb, acc);
                    }
                }
                return acc;
            };
            auto hull_blocks = [&](int I, double (&c)[4]) {
                if constexpr (GEN) {   // hull rows: same-stage pairs get their 6x6 block of A' W A (stages straddle tile borders)
                    if (I - J <= 1) {
                        const int e2 = 16 * J + li;
                        const int s2 = s_stg[e2], a2 = s_thr[e2];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int e1 = 16 * I + lq + 4 * rr;
                            const int s1 = s_stg[e1];
                            if (s1 != 255 && s1 == s2) c[rr] += Sblk[s1 * 36 + s_thr[e1] * 6 + a2];
                        }
                    }
                }
            };
            if (wave == 0) {   // diagonal tile: both operands are row J (LDS), then potrf + inverse
                f64x4 acc = tile_init(J), acc2 = zero4;
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
                double* tjj = Ls + (int64_t)t64idx(J, J) * 256;
                const double* hjj = Hs + (int64_t)t64idx(J, J) * 256;
                double c[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) c[rr] = hjj[t64off(lq + 4 * rr, li)] - acc[rr];
                hull_blocks(J, c);
                const double sg = dv[16 * J + li];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (lq + 4 * rr == li) c[rr] += sg;
                double w[4], lunused[4];
                const bool ok = potrf_inv16_lds(c, Sbuf, Sbuf + 16, lq, li, w, lunused);
                if (!ok && lane == 0) s_flag = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tjj[t64off(lq + 4 * rr, li)] = w[rr];
            }
            S64(8);    // (diagnostic) wave 0: diagonal tile + potrf
            // off-diagonal tiles of this wave as ONE stream of (tile, batch of four block columns): the loads of the next
            // step -- the same tile's next batch or the next tile's first -- are requested before the MFMAs of the current
            // one (two register sets, no copies; a batch beyond J is zero-filled), the Hessian tile at the tile's first step.
            auto next_tile = [&](int I) {
                do ++I; while (I < nb && col_owner(I - J) != wave);
                return I;
            };
            auto store_c = [&](int I, const f64x4& acc, const double (&h)[4]) {
                double* tij = Ls + (int64_t)t64idx(I, J) * 256;
                double c[4];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) c[rr] = h[rr] - acc[rr];
                hull_blocks(I, c);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) tij[t64off(lq + 4 * rr, li)] = c[rr];
            };
            auto load_h = [&](int I, double (&h)[4]) {
                const double* hij = Hs + (int64_t)t64idx(I, J) * 256;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) h[rr] = hij[t64off(lq + 4 * rr, li)];
            };
            {
                const int nq = (J + 3) >> 2;
                int I = next_tile(J), q = 0;
                f64x4 A0[4], A1[4], acc = zero4, acc2 = zero4;
                double hreg[4] = {0.0, 0.0, 0.0, 0.0};
                auto fetch = [&](f64x4 (&A)[4], int It, int qq) {
                    const double* rowI = Ls + (int64_t)t64idx(It, 0) * 256 + 16 * li + 4 * lq;
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
                        acc = tile_init(I);
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
                        store_c(I, tile_init(I), hreg);
                    }
                } else if (I < nb) {
                    fetch(A0, I, 0);
                    for (;;) {
                        if (step(A0, A1)) break;
                        if (step(A1, A0)) break;
                    }
                }
            }
            S64(10);   // (diagnostic) wave 0: its off-diagonal stream
            __syncthreads();
            S64(11);   // (diagnostic) wave 0: waiting for the other waves
            // L_IJ = C_IJ W_J' for the tiles this wave produced, four at a time (C_IJ comes back from L2 in operand order)
            const f64x4 w4 = ld4(Ls + (int64_t)t64idx(J, J) * 256 + 16 * li + 4 * lq);
            for (int I = next_tile(J); I < nb;) {
                int Is[4] = {nb, nb, nb, nb};
                f64x4 a4[4], x[4];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    a4[c4] = zero4;
                    if (I < nb) {
                        Is[c4] = I;
                        a4[c4] = ld4(Ls + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
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
                        double* tij = Ls + (int64_t)t64idx(Is[c4], J) * 256;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) tij[t64off(lq + 4 * rr, li)] = x[c4][rr];
                    }
                }
            }
        }
        };
        // ---- KKT solve (the right-hand side comes and the solution goes through xv, permuted layout) ----
        // Right-looking sweeps: wave w keeps the running sums of ITS block rows (forward: rows I = w mod 4;
        // backward: columns J = w mod 4); the owner of step J finishes block J alone, publishes it in LDS and
        // everybody folds it into their sums.  One LDS-only barrier per step (the outstanding tile loads stay
        // in flight across it), every tile requested one step before it is needed.
        auto solve = [&](auto RPC) {
            constexpr int RP = decltype(RPC)::value;
            constexpr int RM = (RP < RPF) ? RP : RMAXW;
            const int myp = v64pos(li);
            double* rb = part + wave * 16;            // per-wave 16-vector scratch
            f64x4 buf[RP];                            // tiles (I, J) of this wave's rows for the current step
            f64x4 wdiag = {0.0, 0.0, 0.0, 0.0};
            double psum[RM];
#pragma unroll
            for (int i = 0; i < RM; ++i) psum[i] = 0.0;
            // ---- forward: L y = b ----
            auto fetch_f = [&](int J) {               // column J of the factor, rows I = wave + 4 i > J
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int I = wave + NWAVE * i;
                    if (I > J && I < nb) buf[i] = ld4(Ls + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                }
            };
            if (wave == 0) wdiag = ld4(Ls + (int64_t)t64idx(0, 0) * 256 + 16 * li + 4 * lq);
            fetch_f(0);
            for (int J = 0; J < nb; ++J) {
                if (wave == (J & (NWAVE - 1))) {       // owner: r_J = b_J - sum, y_J = W_J r_J
                    const int i = J / NWAVE;
                    double p = 0.0;
#pragma unroll
                    for (int ii = 0; ii < RMAXW; ++ii) p = (ii == i) ? psum[ii] : p;
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
                    for (int i = 0; i < RPF; ++i) {
                        const int I = wave + NWAVE * i;
                        if (I > J && I < nb) psum[i] += buf[i].x * y0 + buf[i].y * y1 + buf[i].z * y2 + buf[i].w * y3;
                    }
#pragma unroll
                    for (int i = RPF; i < RMAXW; ++i) {     // n > 640 only
                        const int I = wave + NWAVE * i;
                        if (I > J && I < nb) {
                            const f64x4 t4 = ld4(Ls + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                            psum[i] += t4.x * y0 + t4.y * y1 + t4.z * y2 + t4.w * y3;
                        }
                    }
                    fetch_f(J + 1);
                    if (wave == ((J + 1) & (NWAVE - 1))) wdiag = ld4(Ls + (int64_t)t64idx(J + 1, J + 1) * 256 + 16 * li + 4 * lq);
                }
            }
            // ---- backward: L' x = y ----
#pragma unroll
            for (int i = 0; i < RMAXW; ++i) psum[i] = 0.0;
            auto fetch_b = [&](int I) {               // row I of the factor, columns J = wave + 4 i < I (transposed use)
#pragma unroll
                for (int i = 0; i < RPF; ++i) {
                    const int J = wave + NWAVE * i;
                    if (J < I) {
                        const double* t = Ls + (int64_t)t64idx(I, J) * 256;
                        buf[i].x = t[t64off(4 * lq + 0, li)];
                        buf[i].y = t[t64off(4 * lq + 1, li)];
                        buf[i].z = t[t64off(4 * lq + 2, li)];
                        buf[i].w = t[t64off(4 * lq + 3, li)];
                    }
                }
            };
            auto fetch_wt = [&](int I) {
                const double* t = Ls + (int64_t)t64idx(I, I) * 256;
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
                    for (int ii = 0; ii < RMAXW; ++ii) p = (ii == i) ? psum[ii] : p;
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
                    for (int i = 0; i < RPF; ++i) {
                        const int J = wave + NWAVE * i;
                        if (J < I) psum[i] += buf[i].x * x0 + buf[i].y * x1 + buf[i].z * x2 + buf[i].w * x3;
                    }
#pragma unroll
                    for (int i = RPF; i < RMAXW; ++i) {     // n > 640 only
                        const int J = wave + NWAVE * i;
                        if (J < I) {
                            const double* t = Ls + (int64_t)t64idx(I, J) * 256;
                            psum[i] += t[t64off(4 * lq + 0, li)] * x0 + t[t64off(4 * lq + 1, li)] * x1 + t[t64off(4 * lq + 2, li)] * x2 +
                                       t[t64off(4 * lq + 3, li)] * x3;
                        }
                    }
                    fetch_b(I - 1);
                    if (wave == ((I - 1) & (NWAVE - 1))) fetch_wt(I - 1);
                }
            }
        };
        if constexpr (MODE == 0) {
        // ---------------- interior-point iterations ----------------
        int status = 1, nit = 0;
        bool first = true;
        const double inv2n = 1.0 / (double)(2 * n);
        for (int it = 0; it <= C.max_iters; ++it) {
            __syncthreads();
            // gradient H d + g by a tile mat-vec only at the start point; afterwards it follows the step through
            // the Newton system just solved, H dd = rhs - Sigma dd (see the update at the end of the iteration)
            if (it == 0) {
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                if (e < npad) dv[16 * (e >> 4) + v64pos(e & 15)] = valid[v] ? ((sl[v] < su[v]) ? lo[v] + sl[v] : hi[v] - su[v]) : 0.0;
            }
            __syncthreads();
            // block rows round-robin over the waves, result via xv
            for (int I = wave; I < nb; I += NWAVE) {
                double a = 0.0;
                for (int J = 0; J < nb; ++J) {
                    if (J <= I) {
                        const f64x4 t4 = ld4(Hs + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                        const double* d4 = dv + 16 * J + 4 * lq;
                        a += t4.x * d4[0] + t4.y * d4[1] + t4.z * d4[2] + t4.w * d4[3];
                    } else {
                        const double* t = Hs + (int64_t)t64idx(J, I) * 256;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) a += t[t64off(4 * lq + rr, li)] * dv[16 * J + rr * 4 + lq];
                    }
                }
                a = quad_sum64(a);
                if (lq == 0) xv[16 * I + li] = a;
            }
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                grad[v] = (e < npad && valid[v]) ? xv[e] + gv[v] : 0.0;
            }
            }
            if (first) {
                double gm = 0.0, wm = 0.0;
#pragma unroll
                for (int v = 0; v < NVT; ++v)
                    if (valid[v]) {
                        gm = fmax(gm, fabs(grad[v]));
                        wm = fmax(wm, ubv[v]);
                    }
                gm = wg_max(gm, red, tid);
                wm = wg_max(wm, red, tid);
                const double mu0 = fmax(0.02 * gm * wm, 1e-3);
#pragma unroll
                for (int v = 0; v < NVT; ++v) {
                    zl[v] = valid[v] ? mu0 / sl[v] : 0.0;
                    zu[v] = valid[v] ? mu0 / su[v] : 0.0;
                }
                first = false;
            }
            double t = 0.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (valid[v]) t += sl[v] * zl[v] + su[v] * zu[v];
            const double mu = wg_sum(t, red, tid) * inv2n;
            if (!(mu >= C.mu_stop)) {
                status = (mu == mu) ? 0 : 2;
                break;
            }
            if (it == C.max_iters) break;
            ++nit;
            // KKT matrix H + Sigma: the factorisation reads each Hessian tile where it needs it (no copy of the
            // 2 KiB tiles into the factor slot); Sigma goes through LDS (dv is free between the gradient and here)
            __syncthreads();
            double Sig[NVT];
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                Sig[v] = valid[v] ? zl[v] / sl[v] + zu[v] / su[v] : 0.0;
                const int e = v * WG + tid;
                if (e < npad) dv[e] = Sig[v];
            }
            __syncthreads();
            S64(4);
            factor();
            __syncthreads();
            if (s_flag == 0) {
                status = 2;
                break;
            }

            S64(5);
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                if (e < npad) xv[16 * (e >> 4) + v64pos(e & 15)] = -grad[v];
            }
            __syncthreads();
            S64(7);
            solve(std::integral_constant<int, SWEEP_ROWS>{});
            S64(6);
            double da[NVT], dzl_a[NVT], dzu_a[NVT];
            double ap = 1.0, ad = 1.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                da[v] = (e < npad && valid[v]) ? xv[16 * (e >> 4) + v64pos(e & 15)] : 0.0;
                dzl_a[v] = dzu_a[v] = 0.0;
                if (valid[v]) {
                    dzl_a[v] = -zl[v] - zl[v] * da[v] / sl[v];
                    dzu_a[v] = -zu[v] + zu[v] * da[v] / su[v];
                    if (da[v] < 0.0) ap = fmin(ap, -sl[v] / da[v]);
                    if (da[v] > 0.0) ap = fmin(ap, su[v] / da[v]);
                    if (dzl_a[v] < 0.0) ad = fmin(ad, -zl[v] / dzl_a[v]);
                    if (dzu_a[v] < 0.0) ad = fmin(ad, -zu[v] / dzu_a[v]);
                }
            }
            ap = wg_min(ap, red, tid);
            ad = wg_min(ad, red, tid);
            t = 0.0;
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (valid[v]) t += (sl[v] + ap * da[v]) * (zl[v] + ad * dzl_a[v]) + (su[v] - ap * da[v]) * (zu[v] + ad * dzu_a[v]);
            const double mu_aff = wg_sum(t, red, tid) * inv2n;
            double sigma = mu_aff / mu;
            sigma = fmin(fmax(sigma * sigma * sigma, 0.0), 1.0);
            double rcl[NVT], rcu[NVT], rhsv[NVT];
            __syncthreads();
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                rcl[v] = rcu[v] = 0.0;
                double rhs = 0.0;
                if (valid[v]) {
                    rcl[v] = sl[v] * zl[v] + da[v] * dzl_a[v] - sigma * mu;
                    rcu[v] = su[v] * zu[v] - da[v] * dzu_a[v] - sigma * mu;
                    rhs = -(grad[v] - zl[v] + zu[v]) - rcl[v] / sl[v] + rcu[v] / su[v];
                }
                rhsv[v] = rhs;
                const int e = v * WG + tid;
                if (e < npad) xv[16 * (e >> 4) + v64pos(e & 15)] = rhs;
            }
            __syncthreads();
            S64(7);
            solve(std::integral_constant<int, SWEEP_ROWS>{});
            S64(6);
            double dd[NVT], dzl[NVT], dzu[NVT];
            ap = 1e300;
            ad = 1e300;
#pragma unroll
            for (int v = 0; v < NVT; ++v) {
                const int e = v * WG + tid;
                dd[v] = (e < npad && valid[v]) ? xv[16 * (e >> 4) + v64pos(e & 15)] : 0.0;
                dzl[v] = dzu[v] = 0.0;
                if (valid[v]) {
                    dzl[v] = (-rcl[v] - zl[v] * dd[v]) / sl[v];
                    dzu[v] = (-rcu[v] + zu[v] * dd[v]) / su[v];
                    if (dd[v] < 0.0) ap = fmin(ap, -sl[v] / dd[v]);
                    if (dd[v] > 0.0) ap = fmin(ap, su[v] / dd[v]);
                    if (dzl[v] < 0.0) ad = fmin(ad, -zl[v] / dzl[v]);
                    if (dzu[v] < 0.0) ad = fmin(ad, -zu[v] / dzu[v]);
                }
            }
            ap = fmin(1.0, 0.9995 * wg_min(ap, red, tid));
            ad = fmin(1.0, 0.9995 * wg_min(ad, red, tid));
#pragma unroll
            for (int v = 0; v < NVT; ++v)
                if (valid[v]) {
                    grad[v] += ap * (rhsv[v] - Sig[v] * dd[v]);   // + ap H dd
                    sl[v] += ap * dd[v];
                    su[v] -= ap * dd[v];
                    zl[v] += ad * dzl[v];
                    zu[v] += ad * dzu[v];
                }
        }

        S64(7);
        // ---------------- outputs ----------------
        __syncthreads();
        double* ubuf = dv;  // N*NT <= 1024 doubles
        for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.0;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NVT; ++v)
            if (valid[v]) {
                double u = (sl[v] < su[v]) ? sl[v] : ubv[v] - su[v];
                if (status == 2) u = ubar[v];
                ubuf[kcol[v] * NT + s_act[acol[v]]] = u;
            }
        __syncthreads();
        if (tid < NT) P.out_u0[inst * NT + tid] = ubuf[tid];
        if (P.out_U)
            for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = ubuf[i];
        if (tid == 0) {
            if (P.status) P.status[inst] = status;
            if (P.iters) P.iters[inst] = nit;
        }
        } else {
            // ================= general-constraint interior-point method (MODE != 0) =================
            // rows:  box rows of the thruster variables (MODE == 2 only: -d <= ubar, d <= ub - ubar), hull rows
            //        A_h tau_k <= b_h of every stage (MODE & 1), terminal rows A_T (e_N + GN d) <= b_T (MODE & 2).
            // Same Mehrotra iteration as the box path with general rows C d + s = h (oracle/qp_oracle.py:ipm_general is
            // the mirror): the box / hull rows start strictly feasible and stay so, the terminal rows start at
            // s = max(residual, 0.1) and carry their primal residual r_p, which every step shrinks by (1 - alpha_p).
            constexpr int NVC = 4;                       // hull rows per thread: N * hull_rows <= 1024
            __shared__ double cw[NVC * WG], xs[NMAX / NVT_MAX];
            __shared__ double s_hA[32 * 6];
            __shared__ double s_tA[80 * 9], s_tbv[80], cwt[80], y9[9], M9[81], C9[81], r9[NWAVE * 9];
            const int MH = GEN ? Q.hull_rows : 0, MT = TSET ? Q.term_rows : 0;
            const int mhull = N * MH;
            const double* GNs = Eall + (int64_t)N * 9 * npad;      // raw terminal sensitivity rows [9][npad]
            double* ETs = Eall + (int64_t)(N + 1) * 9 * npad;       // chol(A_T' W A_T)' GN        [9][npad]
            const bool val = valid[0];
            const int ek = kcol[0], ea = acol[0];
            const double ub0 = ubv[0], ubar0 = ubar[0];
            const int xp = 16 * (tid >> 4) + v64pos(tid & 15);      // this thread's slot of the permuted vectors
            if constexpr (GEN) {
                const int set = Q.hull_set ? Q.hull_set[inst] : 0;
                for (int i = tid; i < MH * 6; i += WG) s_hA[i] = Q.hullA[(int64_t)set * MH * 6 + i];
            }
            if constexpr (TSET) {
                for (int i = tid; i < MT * 9; i += WG) s_tA[i] = Q.termA[i];
                if (tid < MT) s_tbv[tid] = Q.termb[tid];
            }
            __syncthreads();
            double d = 0.0, grd = 0.0;
            double sl = 1.0, su = 1.0, zl = 0.0, zu = 0.0;
            double sh[NVC], zh[NVC];
            int hk[NVC], hr[NVC];
            bool hv[NVC];
            double st = 1.0, zt = 0.0, rpt = 0.0;
            const bool tv = tid < MT;
            if (val) d = GEN ? s_ctr[ea] - ubar0 : 0.5 * ub0 - ubar0;
            if constexpr (!GEN) sl = su = 0.5 * ub0;
#pragma unroll
            for (int v = 0; v < NVC; ++v) {
                const int c = v * WG + tid;
                hv[v] = GEN && c < mhull;
                hk[v] = hv[v] ? c / MH : 0;
                hr[v] = hv[v] ? c - hk[v] * MH : 0;
                sh[v] = 1.0;
                zh[v] = 0.0;
                if (hv[v]) {   // b - A centre: the same for every stage
                    double a = Q.hullb[inst * MH + hr[v]];
#pragma unroll
                    for (int g = 0; g < 6; ++g) a -= s_hA[hr[v] * 6 + g] * s_ctr[g];
                    sh[v] = a;
                }
            }
            auto wg_sum9 = [&](double (&p)[9]) {
#pragma unroll
                for (int r = 0; r < 9; ++r)
                    for (int m = 32; m >= 1; m >>= 1) p[r] += __shfl_xor(p[r], m, 64);
                __syncthreads();
                if (lane == 0) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) r9[wave * 9 + r] = p[r];
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 9; ++r) p[r] = r9[r] + r9[9 + r] + r9[18 + r] + r9[27 + r];
            };
            // rows of C x for a vector held one element per thread
            auto rows_Cx = [&](double xval, double& cl, double& cu, double (&ch)[NVC], double& ct) {
                __syncthreads();
                if (tid < npad) xs[tid] = val ? xval : 0.0;
                double y[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (TSET) {
                    if (val) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) y[r] = GNs[r * npad + tid] * xval;
                    }
                    wg_sum9(y);
                } else {
                    __syncthreads();
                }
                cl = -xval;
                cu = xval;
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    ch[v] = 0.0;
                    if (hv[v]) {
#pragma unroll
                        for (int g = 0; g < 6; ++g) ch[v] += s_hA[hr[v] * 6 + g] * xs[hk[v] * 6 + g];
                    }
                }
                ct = 0.0;
                if constexpr (TSET) {
                    if (tv) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) ct += s_tA[tid * 9 + r] * y[r];
                    }
                }
            };
            // element of C' t for per-row values t
            auto cols_Ct = [&](double tl, double tu, const double (&th)[NVC], double tt) -> double {
                double out = GEN ? 0.0 : (val ? tu - tl : 0.0);
                __syncthreads();
                if constexpr (GEN) {
#pragma unroll
                    for (int v = 0; v < NVC; ++v)
                        if (hv[v]) cw[v * WG + tid] = th[v];
                }
                if constexpr (TSET) {
                    if (tv) cwt[tid] = tt;
                }
                __syncthreads();
                if constexpr (TSET) {
                    if (tid < 9) {
                        double a = 0.0;
                        for (int t = 0; t < MT; ++t) a += s_tA[t * 9 + tid] * cwt[t];
                        y9[tid] = a;
                    }
                }
                if constexpr (GEN) {
                    if (val)
                        for (int r = 0; r < MH; ++r) out += s_hA[r * 6 + ea] * cw[ek * MH + r];
                }
                if constexpr (TSET) {
                    __syncthreads();
                    if (val) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) out += GNs[r * npad + tid] * y9[r];
                    }
                }
                return out;
            };
            // the pieces of the KKT matrix beside H: diag (dv), hull blocks (Sblk), terminal panel (ETs)
            auto form_kkt = [&](double wl, double wu, const double (&wh)[NVC], double wt) {
                __syncthreads();
                if (tid < npad) dv[tid] = (!GEN && val) ? wl + wu : 0.0;
                if constexpr (GEN) {
#pragma unroll
                    for (int v = 0; v < NVC; ++v)
                        if (hv[v]) cw[v * WG + tid] = wh[v];
                }
                if constexpr (TSET) {
                    if (tv) cwt[tid] = wt;
                }
                __syncthreads();
                if constexpr (GEN) {
                    for (int idx = tid; idx < N * 36; idx += WG) {
                        const int k = idx / 36, ab = idx - 36 * k, a = ab / 6, b = ab - 6 * a;
                        double acc = 0.0;
                        for (int r = 0; r < MH; ++r) acc += cw[k * MH + r] * s_hA[r * 6 + a] * s_hA[r * 6 + b];
                        Sblk[idx] = acc;
                    }
                }
                if constexpr (TSET) {
                    if (tid < 81) {
                        const int r1 = tid / 9, r2 = tid - 9 * r1;
                        double acc = 0.0;
                        for (int t = 0; t < MT; ++t) acc += cwt[t] * s_tA[t * 9 + r1] * s_tA[t * 9 + r2];
                        M9[tid] = acc;
                    }
                    __syncthreads();
                    if (tid == 0) {   // M = C' C, C upper triangular (a zero pivot gives a zero row)
                        for (int j = 0; j < 9; ++j) {
                            double dj = M9[9 * j + j];
                            for (int k = 0; k < j; ++k) dj -= C9[9 * k + j] * C9[9 * k + j];
                            const double l = dj > 0.0 ? sqrt(dj) : 0.0;
                            for (int i = 0; i < j; ++i) C9[9 * j + i] = 0.0;
                            C9[9 * j + j] = l;
                            for (int i = j + 1; i < 9; ++i) {
                                double sacc = M9[9 * j + i];
                                for (int k = 0; k < j; ++k) sacc -= C9[9 * k + j] * C9[9 * k + i];
                                C9[9 * j + i] = l > 0.0 ? sacc / l : 0.0;
                            }
                        }
                    }
                    __syncthreads();
                    if (tid < npad) {
#pragma unroll
                        for (int r = 0; r < 9; ++r) {
                            double acc = 0.0;
                            for (int r2 = r; r2 < 9; ++r2) acc += C9[9 * r + r2] * GNs[r2 * npad + tid];
                            ETs[r * npad + tid] = acc;
                        }
                    }
                }
                __syncthreads();
            };
            auto pos_step = [](double sv, double dsv) { return (dsv < 0.0) ? -sv / dsv : 1e300; };

            // (H x)_e for a vector held one element per thread: tile mat-vec over the global Hessian tiles.  The gradient
            // follows every step through H dd itself (as the oracle's does): with row weights z / s ~ 1e7 and more the Newton
            // identity H dd = rhs - C' (w . C dd) is a difference of huge numbers and carries the solve's residual.
            auto h_times = [&](double xval) -> double {
                __syncthreads();
                if (tid < npad) dv[xp] = val ? xval : 0.0;
                __syncthreads();
                for (int I = wave; I < nb; I += NWAVE) {
                    double a = 0.0;
                    for (int J = 0; J < nb; ++J) {
                        if (J <= I) {
                            const f64x4 t4 = ld4(Hs + (int64_t)t64idx(I, J) * 256 + 16 * li + 4 * lq);
                            const double* d4 = dv + 16 * J + 4 * lq;
                            a += t4.x * d4[0] + t4.y * d4[1] + t4.z * d4[2] + t4.w * d4[3];
                        } else {
                            const double* t = Hs + (int64_t)t64idx(J, I) * 256;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) a += t[t64off(4 * lq + rr, li)] * dv[16 * J + rr * 4 + lq];
                        }
                    }
                    a = quad_sum64(a);
                    if (lq == 0) xv[16 * I + li] = a;
                }
                __syncthreads();
                return (tid < npad && val) ? xv[tid] : 0.0;
            };
            // gradient at the start point: H d + g
            grd = h_times(d) + ((tid < npad && val) ? gv[0] : 0.0);
            {
                // terminal rows: residual at the start point, slack and primal residual
                double cl, cu, ch[NVC], ct;
                rows_Cx(d, cl, cu, ch, ct);
                if constexpr (TSET) {
                    if (tv) {
                        double res = s_tbv[tid] - ct;
#pragma unroll
                        for (int r = 0; r < 9; ++r) res -= s_tA[tid * 9 + r] * Q.eN[inst * 9 + r];
                        st = fmax(res, 0.1);
                        rpt = st - res;
                    }
                }
                double gm = val ? fabs(grd) : 0.0, sm = 0.0;
                if constexpr (!GEN) sm = val ? sl : 0.0;
#pragma unroll
                for (int v = 0; v < NVC; ++v)
                    if (hv[v]) sm = fmax(sm, sh[v]);
                if (tv) sm = fmax(sm, st);
                gm = wg_max(gm, red, tid);
                sm = wg_max(sm, red, tid);
                const double mu0 = fmax(0.02 * gm * sm, 1e-3);
                if constexpr (!GEN) {
                    zl = val ? mu0 / sl : 0.0;
                    zu = val ? mu0 / su : 0.0;
                }
#pragma unroll
                for (int v = 0; v < NVC; ++v) zh[v] = hv[v] ? mu0 / sh[v] : 0.0;
                zt = tv ? mu0 / st : 0.0;
            }
            int status = 1, nit = 0;
            const double inv_m = 1.0 / (double)((GEN ? mhull : 2 * n) + MT);
            const double zero4[NVC] = {0.0, 0.0, 0.0, 0.0};
            for (int it = 0; it <= C.max_iters; ++it) {
                double t = 0.0;
                if constexpr (!GEN) t += val ? sl * zl + su * zu : 0.0;
#pragma unroll
                for (int v = 0; v < NVC; ++v)
                    if (hv[v]) t += sh[v] * zh[v];
                if (tv) t += st * zt;
                const double mu = wg_sum(t, red, tid) * inv_m;
                const double rpn = wg_max(tv ? fabs(rpt) : 0.0, red, tid);
                if (!(mu == mu) || !(rpn == rpn) || mu > 1e300 || rpn > 1e300) {
                    status = 2;
                    break;
                }
                if (mu < C.mu_stop && rpn < 1e-9) {
                    status = 0;
                    break;
                }
                if (it == C.max_iters) break;
                ++nit;
                const double wl = (!GEN && val) ? zl / sl : 0.0, wu = (!GEN && val) ? zu / su : 0.0, wt = tv ? zt / st : 0.0;
                double wh[NVC];
#pragma unroll
                for (int v = 0; v < NVC; ++v) wh[v] = hv[v] ? zh[v] / sh[v] : 0.0;
                form_kkt(wl, wu, wh, wt);
                factor();
                __syncthreads();
                if (s_flag == 0) {   // the factorisation broke down: C' W C with W ~ 1/mu ruins the conditioning near the end
                    status = (mu < 1e-7 && rpn < 1e-9) ? 0 : 2;
                    --nit;
                    break;
                }
                // predictor: rc = s z  ->  t = -z + (rc - z rp) / s = -z rp / s  (zero except on the terminal rows)
                double rhs = -grd;
                if constexpr (TSET) rhs += cols_Ct(0.0, 0.0, zero4, tv ? -zt * rpt / st : 0.0);
                __syncthreads();
                if (tid < npad) xv[xp] = val ? rhs : 0.0;
                __syncthreads();
                solve(std::integral_constant<int, SWEEP_ROWS>{});
                __syncthreads();
                const double da = (tid < npad && val) ? xv[xp] : 0.0;
                double cl, cu, ch[NVC], ct;
                rows_Cx(da, cl, cu, ch, ct);
                // ds = -rp - C dd,  dz = (-rc - z ds) / s
                double dsl_a = -cl, dsu_a = -cu, dst_a = -rpt - ct, dsh_a[NVC];
                double dzl_a = 0.0, dzu_a = 0.0, dzt_a = 0.0, dzh_a[NVC];
                double ap = 1.0, ad = 1.0;
                if constexpr (!GEN) {
                    if (val) {
                        dzl_a = -zl - zl * dsl_a / sl;
                        dzu_a = -zu - zu * dsu_a / su;
                        ap = fmin(ap, fmin(pos_step(sl, dsl_a), pos_step(su, dsu_a)));
                        ad = fmin(ad, fmin(pos_step(zl, dzl_a), pos_step(zu, dzu_a)));
                    }
                }
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    dsh_a[v] = -ch[v];
                    dzh_a[v] = 0.0;
                    if (hv[v]) {
                        dzh_a[v] = -zh[v] - zh[v] * dsh_a[v] / sh[v];
                        ap = fmin(ap, pos_step(sh[v], dsh_a[v]));
                        ad = fmin(ad, pos_step(zh[v], dzh_a[v]));
                    }
                }
                if (tv) {
                    dzt_a = -zt - zt * dst_a / st;
                    ap = fmin(ap, pos_step(st, dst_a));
                    ad = fmin(ad, pos_step(zt, dzt_a));
                }
                ap = wg_min(ap, red, tid);
                ad = wg_min(ad, red, tid);
                t = 0.0;
                if constexpr (!GEN) t += val ? (sl + ap * dsl_a) * (zl + ad * dzl_a) + (su + ap * dsu_a) * (zu + ad * dzu_a) : 0.0;
#pragma unroll
                for (int v = 0; v < NVC; ++v)
                    if (hv[v]) t += (sh[v] + ap * dsh_a[v]) * (zh[v] + ad * dzh_a[v]);
                if (tv) t += (st + ap * dst_a) * (zt + ad * dzt_a);
                const double mu_aff = wg_sum(t, red, tid) * inv_m;
                double sigma = mu_aff / mu;
                sigma = fmin(fmax(sigma * sigma * sigma, 0.0), 1.0);
                // corrector: rc = s z + ds_a dz_a - sigma mu,  t = -z + (rc - z rp) / s
                double rcl = 0.0, rcu = 0.0, rct = 0.0, rch[NVC], tl = 0.0, tu = 0.0, tt = 0.0, th[NVC];
                if constexpr (!GEN) {
                    if (val) {
                        rcl = sl * zl + dsl_a * dzl_a - sigma * mu;
                        rcu = su * zu + dsu_a * dzu_a - sigma * mu;
                        tl = -zl + rcl / sl;
                        tu = -zu + rcu / su;
                    }
                }
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    rch[v] = th[v] = 0.0;
                    if (hv[v]) {
                        rch[v] = sh[v] * zh[v] + dsh_a[v] * dzh_a[v] - sigma * mu;
                        th[v] = -zh[v] + rch[v] / sh[v];
                    }
                }
                if (tv) {
                    rct = st * zt + dst_a * dzt_a - sigma * mu;
                    tt = -zt + (rct - zt * rpt) / st;
                }
                rhs = -grd + cols_Ct(tl, tu, th, tt);
                __syncthreads();
                if (tid < npad) xv[xp] = val ? rhs : 0.0;
                __syncthreads();
                solve(std::integral_constant<int, SWEEP_ROWS>{});
                __syncthreads();
                const double dd = (tid < npad && val) ? xv[xp] : 0.0;
                rows_Cx(dd, cl, cu, ch, ct);
                const double dsl = -cl, dsu = -cu, dst = -rpt - ct;
                double dsh[NVC], dzl = 0.0, dzu = 0.0, dzt = 0.0, dzh[NVC];
                ap = 1e300;
                ad = 1e300;
                if constexpr (!GEN) {
                    if (val) {
                        dzl = (-rcl - zl * dsl) / sl;
                        dzu = (-rcu - zu * dsu) / su;
                        ap = fmin(ap, fmin(pos_step(sl, dsl), pos_step(su, dsu)));
                        ad = fmin(ad, fmin(pos_step(zl, dzl), pos_step(zu, dzu)));
                    }
                }
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    dsh[v] = -ch[v];
                    dzh[v] = 0.0;
                    if (hv[v]) {
                        dzh[v] = (-rch[v] - zh[v] * dsh[v]) / sh[v];
                        ap = fmin(ap, pos_step(sh[v], dsh[v]));
                        ad = fmin(ad, pos_step(zh[v], dzh[v]));
                    }
                }
                if (tv) {
                    dzt = (-rct - zt * dst) / st;
                    ap = fmin(ap, pos_step(st, dst));
                    ad = fmin(ad, pos_step(zt, dzt));
                }
                ap = fmin(1.0, 0.9995 * wg_min(ap, red, tid));
                ad = fmin(1.0, 0.9995 * wg_min(ad, red, tid));
                const double hdd = h_times(dd);
                if (val) {
                    grd += ap * hdd;
                    d += ap * dd;
                    if constexpr (!GEN) {
                        sl += ap * dsl;
                        su += ap * dsu;
                        zl += ad * dzl;
                        zu += ad * dzu;
                    }
                }
#pragma unroll
                for (int v = 0; v < NVC; ++v)
                    if (hv[v]) {
                        sh[v] += ap * dsh[v];
                        zh[v] += ad * dzh[v];
                    }
                if (tv) {
                    st += ap * dst;
                    zt += ad * dzt;
                    rpt *= (1.0 - ap);
                }
            }
            S64(7);
            // ---------------- active-set polish (oracle/qp_oracle.py:polish_general is the mirror) ----------------
            // An interior-point iterate at mu 1e-10 is up to 7e-5 f_max from the exact solution where rows are weakly active
            // (z ~ s ~ 1e-5).  Active set A = {z > s}; per round the equality-constrained problem on A by two steps of the method
            // of multipliers with penalty W_i = 1e6 max diag(H) / |c_i|^2 (one factorisation of H + C_A' W C_A -- the Newton matrix's
            // own shape -- and one solve per step), then the signs are checked: a row of A with a negative multiplier leaves, a
            // violated row outside enters, and the round is repeated until nothing changes.  Not verified within three rounds
            // (or a factorisation that breaks down): the interior-point iterate is returned as before.
            if (status == 0) {
                constexpr double PW0 = 1e6, PRES_TOL = 1e-10;
                double hd = 0.0;
                if (tid < npad && val) hd = fabs(Hs[(int64_t)t64idx(tid >> 4, tid >> 4) * 256 + t64off(tid & 15, tid & 15)]);
                const double pw = PW0 * wg_max(hd, red, tid);
                double hh[NVC], wph[NVC], wpt = 0.0, ht = 0.0;
                const double hl = ubar0, hu = ub0 - ubar0;
                {
                    double cl, cu, ch[NVC], ct;
                    rows_Cx(ubar0, cl, cu, ch, ct);      // A_h ubar_k: offsets of the hull rows in d
#pragma unroll
                    for (int v = 0; v < NVC; ++v) {
                        hh[v] = 0.0;
                        wph[v] = 0.0;
                        if (hv[v]) {
                            hh[v] = Q.hullb[inst * MH + hr[v]] - ch[v];
                            double a2 = 0.0;
#pragma unroll
                            for (int g = 0; g < 6; ++g) a2 += s_hA[hr[v] * 6 + g] * s_hA[hr[v] * 6 + g];
                            wph[v] = pw / fmax(a2, 1e-300);
                        }
                    }
                    if constexpr (TSET) {
                        if (tv) {
                            ht = s_tbv[tid];
#pragma unroll
                            for (int r = 0; r < 9; ++r) ht -= s_tA[tid * 9 + r] * Q.eN[inst * 9 + r];
                            double c2 = 0.0;
                            for (int e = 0; e < npad; ++e) {
                                double c = 0.0;
#pragma unroll
                                for (int r = 0; r < 9; ++r) c += s_tA[tid * 9 + r] * GNs[r * npad + e];
                                c2 += c * c;
                            }
                            wpt = pw / fmax(c2, 1e-300);
                        }
                    }
                }
                const double d_ipm = d;
                bool al = !GEN && val && zl > sl, au = !GEN && val && zu > su, at = tv && zt > st, ah[NVC];
                zl = al ? zl : 0.0;
                zu = au ? zu : 0.0;
                zt = at ? zt : 0.0;
#pragma unroll
                for (int v = 0; v < NVC; ++v) {
                    ah[v] = hv[v] && zh[v] > sh[v];
                    zh[v] = ah[v] ? zh[v] : 0.0;
                }
                bool verified = false;
                for (int rd = 0; rd < 3 && !verified; ++rd) {
                    const double wl = al ? pw : 0.0, wu = au ? pw : 0.0, wt = at ? wpt : 0.0;
                    double wh[NVC];
#pragma unroll
                    for (int v = 0; v < NVC; ++v) wh[v] = ah[v] ? wph[v] : 0.0;
                    form_kkt(wl, wu, wh, wt);
                    factor();
                    __syncthreads();
                    if (s_flag == 0) break;
                    ++nit;
                    double cl, cu, ch[NVC], ct;
                    for (int in = 0; in < 2; ++in) {
                        const double gd = h_times(d) + ((tid < npad && val) ? gv[0] : 0.0);
                        rows_Cx(d, cl, cu, ch, ct);
                        const double s_l = hl - cl, s_u = hu - cu, s_t = ht - ct;
                        double s_h[NVC], th[NVC];
#pragma unroll
                        for (int v = 0; v < NVC; ++v) {
                            s_h[v] = hh[v] - ch[v];
                            th[v] = ah[v] ? wh[v] * s_h[v] - zh[v] : 0.0;
                        }
                        const double rhs = -gd + cols_Ct(al ? wl * s_l - zl : 0.0, au ? wu * s_u - zu : 0.0, th, at ? wt * s_t - zt : 0.0);
                        __syncthreads();
                        if (tid < npad) xv[xp] = val ? rhs : 0.0;
                        __syncthreads();
                        solve(std::integral_constant<int, SWEEP_ROWS>{});
                        __syncthreads();
                        const double dl = (tid < npad && val) ? xv[xp] : 0.0;
                        rows_Cx(dl, cl, cu, ch, ct);
                        if (al) zl += wl * (cl - s_l);
                        if (au) zu += wu * (cu - s_u);
                        if (at) zt += wt * (ct - s_t);
#pragma unroll
                        for (int v = 0; v < NVC; ++v)
                            if (ah[v]) zh[v] += wh[v] * (ch[v] - s_h[v]);
                        if (val) d += dl;
                    }
                    rows_Cx(d, cl, cu, ch, ct);
                    double changed = 0.0;
                    auto check = [&](bool& a, double& lam, bool row, double res) {
                        if (a && lam < 0.0) {
                            a = false;
                            lam = 0.0;
                            changed = 1.0;
                        } else if (!a && row && res < -PRES_TOL) {
                            a = true;
                            lam = 0.0;
                            changed = 1.0;
                        }
                    };
                    check(al, zl, !GEN && val, hl - cl);
                    check(au, zu, !GEN && val, hu - cu);
                    check(at, zt, tv, ht - ct);
#pragma unroll
                    for (int v = 0; v < NVC; ++v) check(ah[v], zh[v], hv[v], hh[v] - ch[v]);
                    verified = wg_max(changed, red, tid) == 0.0;
                }
                if (!verified) d = d_ipm;
                else if constexpr (!GEN) {      // (the output stage reads the thruster force off the slacks)
                    sl = fmax(hl + d, 0.0);
                    su = fmax(hu - d, 0.0);
                }
            }
            // ---------------- outputs ----------------
            __syncthreads();
            if constexpr (GEN) {
                // The allocator takes tau_0 next and needs it INSIDE the hull: the polished solution sits ON its active facets, a
                // few 1e-13 outside as often as inside, and for a wrench outside the attainable set the allocation's dual is
                // unbounded.  Pull tau_0 towards the hull centre by the smallest factor that leaves every facet a relative
                // margin of 1e-9 (kernel 11 does the same); the whole-horizon output G keeps the solution as it is.
                double eps = 0.0;
                {
                    double cl, cu, ch[NVC], ct;
                    rows_Cx((status == 2) ? ubar0 : ubar0 + d, cl, cu, ch, ct);      // A_h tau_k
#pragma unroll
                    for (int v = 0; v < NVC; ++v)
                        if (hv[v] && hk[v] == 0) {
                            const double bb = Q.hullb[inst * MH + hr[v]];
                            double s0 = bb;
#pragma unroll
                            for (int g = 0; g < 6; ++g) s0 -= s_hA[hr[v] * 6 + g] * s_ctr[g];
                            const double st0 = bb - ch[v];
                            if (st0 < 1e-9 * s0 && s0 > st0) eps = fmax(eps, (1e-9 * s0 - st0) / (s0 - st0));
                        }
                    eps = fmin(wg_max(eps, red, tid), 1.0);
                }
                if (val) {
                    const double tau = (status == 2) ? ubar0 : ubar0 + d;
                    if (ek == 0) Q.out_tau0[inst * 6 + ea] = s_ctr[ea] + (1.0 - eps) * (tau - s_ctr[ea]);
                    if (Q.out_G) Q.out_G[(inst * N + ek) * 6 + ea] = tau;
                }
            } else {
                double* ubuf = dv;
                for (int i = tid; i < N * NT; i += WG) ubuf[i] = 0.0;
                __syncthreads();
                if (val) {
                    double u = (sl < su) ? sl : ub0 - su;
                    if (status == 2) u = ubar0;
                    ubuf[ek * NT + s_act[ea]] = u;
                }
                __syncthreads();
                if (tid < NT) P.out_u0[inst * NT + tid] = ubuf[tid];
                if (P.out_U)
                    for (int i = tid; i < N * NT; i += WG) P.out_U[inst * (int64_t)N * NT + i] = ubuf[i];
            }
            if (tid == 0) {
                if (P.status) P.status[inst] = status;
                if (P.iters) P.iters[inst] = nit;
            }
        }
        S64(9);
#ifdef FTMPC_STAMPS
        if (tid == 0 && inst < 512 && Q.dbg_H) {
            unsigned long long* sb = reinterpret_cast<unsigned long long*>(Q.dbg_H) + inst * 12;
            for (int i = 0; i < 12; ++i) sb[i] = s64_acc[i];
        }
#endif
    }
}

template __global__ void ftmpc_solve_f64_kernel<4, 1>(const DeviceConsts, const Solve64Params);                      // n <= 256
template __global__ void ftmpc_solve_f64_kernel<f64k::RPF, 3>(const DeviceConsts, const Solve64Params);              // n <= 640
template __global__ void ftmpc_solve_f64_kernel<f64k::RPF, f64k::NVT_MAX>(const DeviceConsts, const Solve64Params);  // n <= 1024
// general-constraint modes (n <= 256): 1 = wrench variables + hull rows, 2 = thruster variables + box + terminal set, 3 = wrench + hull + terminal set
template __global__ void ftmpc_solve_f64_kernel<4, 1, 1>(const DeviceConsts, const Solve64Params);
template __global__ void ftmpc_solve_f64_kernel<4, 1, 2>(const DeviceConsts, const Solve64Params);
template __global__ void ftmpc_solve_f64_kernel<4, 1, 3>(const DeviceConsts, const Solve64Params);

}  // namespace ftmpc
