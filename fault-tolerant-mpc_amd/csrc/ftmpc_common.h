// ftmpc_common.h -- shared host/device definitions of the MI355X MPC QP-step path.
//
// Per-stage linearisation record written by the linearise kernel (one per instance and
// stage, REC_STRIDE words) and consumed by the condense+IPM kernel.  The RK4 transition of
// the orbit-centre model (reference: ft_mpc/models/spiral_model.py:44-76 discretised by
// ft_mpc/models/sys_model.py:138-162) has the block structure
//        p        v       w      q            F     tau
//   p [  I      dt*I     Apw    Apq  ]     [ BpF   BpT ]
//   v [  0       I       Avw    Avq  ]     [ BvF   BvT ]
//   w [  0       0       Aww     0   ]     [  0    BwT ]
//   q [  0       0       Aqw    Aqq  ]     [  0    BqT ]
// (p enters nothing, v only p, the attitude chain (w,q) is autonomous), so only the named
// blocks are stored.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ftmpc {

enum : int {
    REC_APW = 0,    // 3x3
    REC_APQ = 9,    // 3x4
    REC_AVW = 21,   // 3x3
    REC_AVQ = 30,   // 3x4
    REC_AWW = 42,   // 3x3
    REC_AQW = 51,   // 4x3
    REC_AQQ = 63,   // 4x4
    REC_BPF = 79,   // 3x3
    REC_BPT = 88,   // 3x3
    REC_BVF = 97,   // 3x3
    REC_BVT = 106,  // 3x3
    REC_BWT = 115,  // 3x3
    REC_BQT = 124,  // 4x3
    REC_WE = 136,   // 9 : W (c_{k+1}[0:9] - xref_{k+1}),  W = diag(Q) or P (terminal)
    REC_RUT = 145,  // 6 : R .* (gen_k - ur_k - [f_virt;0])
    REC_USED = 151,
    REC_STRIDE = 152
};

constexpr int MAX_NT = 16;

// Per-workgroup global slot of the fp32 solve kernels, in 4-byte words:
//   doubles [0, 160)            reference gradient (n <= 160)
//   doubles [160, 160 + 8N)     wrench perturbations of all stages
//   doubles [160 + 8N, ..+9(N+1)) stage storage of the sweeps when it does not fit in LDS
//   (see slot_backup_off_words) then, 256-word aligned, the Hessian tiles (used by the instantiations with NB > 8)
__host__ __device__ constexpr int slot_gens_off() { return 160; }
__host__ __device__ constexpr int slot_stage_off(int N) { return 160 + 8 * N; }
//   words: 5 x 192 (the interior-point iterate kept while the early polish of kernel 2 runs: five arrays of NV x 64 lanes, NV <= 3)
__host__ __device__ constexpr int slot_backup_off_words(int N) { return 2 * (160 + 8 * N + 9 * (N + 1)); }
//   256-word aligned: 55 x 256 words, the factor tiles parked around the float64 gradient of the polish (kernel 2: NB (NB - 1) / 2
//   off-diagonal tiles, and the NB inverse diagonal blocks where they live in registers)
__host__ __device__ constexpr int slot_factor_off_words(int N) { return ((slot_backup_off_words(N) + 5 * 192 + 255) / 256) * 256; }
__host__ __device__ constexpr int slot_tile_off_words(int N) { return slot_factor_off_words(N) + 55 * 256; }

// constants shared by both kernels (passed by value as kernel argument)
struct DeviceConsts {
    int N, NT;
    int max_iters;
    int pad0;
    double dt, inv_mass;
    double J[9], Jinv[9];
    double ArT[9];          // -[r]x Jinv : d a_b / d tau
    double r[3];
    double fvirt[3];
    double D[6 * MAX_NT];   // row-major 6 x NT (row stride MAX_NT)
    double Q[9], R[6], P[81];
    double LPt[81];         // sqrt(2) * chol(P)^T (upper triangular), row-major: E_N = LPt * G9
    double sq2Q[9];         // sqrt(2 Q)
    double rho;
    double mu_stop;
    double mu_refine;       // take the float64 reference gradient once mu falls below this (<= 0: never)
};

// non-quadratic terminal-cost terms (include/ftmpc.h ftmpc_config.tc_*), device copy
constexpr int MAX_TCOST = 24;
struct TermCost {
    int npoly, nroot;
    double poly_coef[MAX_TCOST];
    int poly_exp[MAX_TCOST * 9];
    double root_coef[MAX_TCOST], root_eps[MAX_TCOST], root_pow[MAX_TCOST];
    int root_exp[MAX_TCOST * 9];
    double cconst;
};

// e^k for small non-negative integer k
__host__ __device__ inline double ipow(double e, int k) {
    double r = 1.0;
    for (int i = 0; i < k; ++i) r *= e;
    return r;
}
// V_nq(e) and (when grad != nullptr) its gradient
__host__ __device__ inline double term_cost_nq(const TermCost& T, const double e[9], double* grad) {
    double v = T.cconst;
    if (grad)
        for (int i = 0; i < 9; ++i) grad[i] = 0.0;
    for (int t = 0; t < T.npoly + T.nroot; ++t) {
        const bool root = t >= T.npoly;
        const int* ex = root ? T.root_exp + 9 * (t - T.npoly) : T.poly_exp + 9 * t;
        double m = 1.0;
        for (int j = 0; j < 9; ++j) m *= ipow(e[j], ex[j]);
        double outer = 1.0, coef;                    // d/dm of the term
        if (root) {
            const int r = t - T.npoly;
            const double base = m + T.root_eps[r];
            coef = T.root_coef[r];
            v += coef * pow(base, T.root_pow[r]);
            outer = T.root_pow[r] * pow(base, T.root_pow[r] - 1.0);
        } else {
            coef = T.poly_coef[t];
            v += coef * m;
        }
        if (grad)
            for (int i = 0; i < 9; ++i) {
                if (ex[i] == 0) continue;
                double dm = ex[i] * ipow(e[i], ex[i] - 1);
                for (int j = 0; j < 9; ++j)
                    if (j != i) dm *= ipow(e[j], ex[j]);
                grad[i] += coef * outer * dm;
            }
    }
    return v;
}

struct LinParams {
    int64_t B;
    const double* x0;       // [B*13]
    const double* ub;       // [B*NT]
    const double* stuck;    // [B*NT]
    const double* xref;     // 9 x (N+1) col-major, stride xref_stride per instance (0 shared)
    int64_t xref_stride;
    const double* uref;     // or nullptr
    int64_t uref_stride;
    const double* warmU;    // [B*N*NT] or nullptr
    void* rec;              // [B*N*REC_STRIDE] float or double
    // work lists of the fp32 solve instantiations (nullptr: not built).  Instance b goes to list
    // v = clamp(ceil(N*na/16), 8, 8 + qvmax) - 8 (na = healthy thrusters; na = 0 -> list 0):
    // qlist[v*B + i], i < qcount[v].  The counters are zeroed by the host before the launch.
    int32_t* qlist;
    int32_t* qcount;        // [3]
    int32_t qvmax;          // last instantiation this handle launches (0..2)
    // generalized-force formulation: the linearisation wrench of every stage is given directly
    // ([B*N*6]; nullptr: D (clip(warmU) + stuck) as above)
    const double* warmG;
    double* out_eN;         // nullptr or [B*9]: terminal tracking error c_N[0:9] - xref_N at the linearisation point
    const TermCost* tcost;  // nullptr or the non-quadratic terminal-cost terms: W e_N gets + 1/2 grad V_nq(e_N)
    double* out_cbar;       // nullptr or [B*N*13]: the linearisation trajectory c_1 .. c_N = [p, v, omega, q] (state-bound rows)
};

struct SolveParams {
    int64_t B;
    const void* rec;        // [B*N*REC_STRIDE]
    const double* ub;
    const double* stuck;
    const double* warmU;    // or nullptr
    double* out_u0;         // [B*NT]
    double* out_U;          // [B*N*NT] or nullptr
    int32_t* status;        // or nullptr
    int32_t* iters;         // or nullptr
    float* hscratch;        // [gridDim.x * tile_words] per-workgroup Hessian slot
    int64_t tile_words;     // words per slot
    // work list of this launch, written by the linearise kernel (LinParams::qlist): the waves pull
    // instance numbers from it through the shared cursor *qhead until *qcount are handed out, so a
    // launch whose list is empty returns at once and slow instances do not leave a static tail
    const int32_t* qlist;
    const int32_t* qcount;
    int32_t* qhead;
    // debug dump (test hook): instance dbg_inst writes its QP here; -1 = off
    int64_t dbg_inst;
    float* dbg_H;           // [npad*npad] row-major
    float* dbg_vec;         // [484]: g | lo | hi at stride npad, [480] = n, [481] = npad
};

}  // namespace ftmpc
