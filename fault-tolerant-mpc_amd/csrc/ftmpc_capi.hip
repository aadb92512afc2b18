// ftmpc_capi.hip -- host side of the C-ABI declared in include/ftmpc.h.
//
// Owns the device workspace, converts the double-precision problem constants into the
// kernel argument block and enqueues the two kernels of the path:
//   ftmpc_linearize_kernel  (one lane per instance)      ftmpc_linearize.hip
//   ftmpc_solve_f32_kernel  (one wavefront per instance)  ftmpc_solve.hip
// There is deliberately NO CPU fallback: every entry point fails with FTMPC_ERR_NODEVICE /
// FTMPC_ERR_HIP when the gfx950 device or the code object is unusable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/ftmpc.h"
#include "ftmpc_common.h"

// single translation unit: the kernels are compiled together with their launcher
#include "ftmpc_linearize.hip"
#include "ftmpc_solve.hip"
#include "ftmpc_solve_f64.hip"
#include "ftmpc_solve_wg.hip"
#include "ftmpc_solve_ws.hip"
#include "ftmpc_solve_ws64.hip"
#include "ftmpc_solve_wsw.hip"
#include "ftmpc_solve_hull.hip"
#include "ftmpc_solve_ric.hip"
#include "ftmpc_solve_ricw.hip"
#include "ftmpc_sim.hip"
#include "ftmpc_alloc.hip"

using ftmpc::DeviceConsts;
using ftmpc::LinParams;
using ftmpc::SolveParams;
using ftmpc::Solve64Params;
using ftmpc::TermCost;
using ftmpc::SolveWgParams;
using ftmpc::SolveWs64Params;

static thread_local std::string g_create_error;

struct ftmpc_handle {
    ftmpc_config cfg;
    DeviceConsts dc;
    int device = 0;
    int num_cu = 0;
    int nb_max = 0;  // ceil(N*NT/16)
    std::string err;
    hipStream_t stream = nullptr;  // internal stream of the host-buffer entry point
    // workspace
    int64_t cap_batch = 0;
    void* rec = nullptr;
    // device mirrors of host buffers
    double *d_x0 = nullptr, *d_ub = nullptr, *d_stuck = nullptr, *d_xref = nullptr, *d_uref = nullptr;
    double *d_warm = nullptr, *d_u0 = nullptr, *d_U = nullptr;
    int32_t *d_status = nullptr, *d_iters = nullptr;
    // allocation operator (ftmpc_allocate_batch)
    double *d_atau = nullptr, *d_aub = nullptr, *d_au = nullptr;
    int32_t *d_ast = nullptr, *d_ait = nullptr;
    int64_t cap_alloc = 0;
    int64_t cap_xref = 0, cap_uref = 0;
    // per-instantiation Hessian slots
    float* hs[3] = {nullptr, nullptr, nullptr};   // NB = 8, 9, 10 instantiations
    int grid[3] = {0, 0, 0};
    // work lists of the fp32 instantiations: qlist [3][cap_batch], qctl = {count[3], pad, head[3], pad}
    int32_t* d_qlist = nullptr;
    int32_t* d_qctl = nullptr;
    // Hint from the previous step: the list lengths it ended with, copied to pinned memory behind its kernels.  A list that was
    // empty then gets a SMALL grid now (the kernels are persistent: any grid drains any list, a small one just slower if the hint
    // is wrong: one step, then the hint is right again) -- most batches fill one list, and an idle launch of a full grid costs
    // 0.05 - 0.09 ms.
    int32_t* h_qcnt = nullptr;      // pinned, 4 ints
    hipEvent_t ev_qcnt = nullptr;
    bool qcnt_pending = false, qcnt_valid = false;
    int32_t last_cnt[4] = {0, 0, 0, 0};
    // pinned staging of the host-buffer entry points (hipHostMalloc; mirrors of the device buffers)
    struct Pinned {
        void* p = nullptr;
        size_t bytes = 0;
    };
    Pinned pin_in, pin_out;
    hipStream_t s_in = nullptr, s_out = nullptr;
    static constexpr int MAX_CHUNKS = 8;
    hipEvent_t ev_in[MAX_CHUNKS] = {}, ev_k[MAX_CHUNKS] = {}, ev_out[MAX_CHUNKS] = {};
    int64_t lin_split_max = 8192;   // ftmpc_config.lin_split_max overrides (0 here: never split)
    int stage_chunks = 0;   // 0: whole blocks of 65 536 instances (a persistent launch below that does not fill the device twice)
    // fp32 workgroup-per-instance kernel with the factor in LDS (160 < N*NT <= 240)
    bool use_ws = false;            // kernel 8 (wrench-space Schur form) takes the lists of NB = 9, 10 and of the workgroup kernel
    int ws_nb = 8, grid_ws = 0;
    float* ws_slot = nullptr;
    int64_t ws_slot_words = 0;
    // kernel 10: the wrench-space form on ONE wave per instance (takes kernel 8's list when it applies)
    bool use_wsw = false;
    int grid_wsw = 0;
    float* wsw_slot = nullptr;
    int64_t wsw_slot_words = 0;
    // kernel 11 (the generalized-force formulation with hull rows, one wave per instance, fp32): float64 scratch of its reference gradient
    float* hull_slot = nullptr;
    int64_t hull_slot_words = 0;
    int grid_hull = 0;
    bool use_wg = false;
    float* wg_slot = nullptr;
    int grid_wg = 0;
    int64_t wg_slot_words = 0;
    // float64 through the wrench-space form (kernel 9): 6 N <= 256, N * NT <= 768, ten or more thrusters
    bool tset_thruster = true;         // the thruster form with the terminal set fits the dense float64 kernel's general-constraint mode
    bool sbounds = false;              // state bounds: the thruster-space solve runs on kernel 12's state-bound instantiation
    double* d_cbar = nullptr;          // [B*N*13] linearisation trajectory (state bounds only)
    bool use_ric64 = false;            // kernel 12: float64, Newton systems by the Riccati recursion, one wave per instance
    int ric_nv = 10;
    int grid_ric = 0;
    double* ric_slot = nullptr;
    int64_t ric_slot_doubles = 0;
    bool use_ws64 = false;
    int ws64_nvt = 1, grid_ws64 = 0;
    double* ws64_slot = nullptr;
    int64_t ws64_slot_doubles = 0;
    // float64 general-size path
    bool use_f64 = false;
    int npad_max = 0;
    int grid64 = 0;
    int64_t tile_doubles = 0, e_doubles = 0;
    double *Hs = nullptr, *Ls = nullptr, *Eall = nullptr;
    double *d_dbgH64 = nullptr, *d_dbgv64 = nullptr;
    // general-constraint modes of the float64 kernel (terminal set; generalized-force formulation)
    bool tset = false;
    double* d_term = nullptr;          // [term_rows*9 | term_rows]
    double* d_eN = nullptr;            // [cap_batch*9]
    int grid_gen = 0, npad_gen = 0;
    int64_t tile_doubles_gen = 0, e_doubles_gen = 0;
    double *gHs = nullptr, *gLs = nullptr, *gEall = nullptr;     // slots of the wrench formulation (n = 6 N)
    double *d_hullA = nullptr, *d_hullb = nullptr, *d_warmG = nullptr, *d_tau0 = nullptr, *d_G = nullptr, *d_taud = nullptr;
    int32_t* d_hullset = nullptr;
    int32_t* d_ast2 = nullptr;
    TermCost* d_tcost = nullptr;       // non-quadratic terminal-cost terms (terminal_cost_terms != 0)
    double* d_cost = nullptr;
    int64_t cap_cost = 0;
    // on-device SQP (ftmpc_solve_sqp_batch): iterate, QP solution, trial point | J, Jt, J0, alpha | flags and counters
    double *d_sqU = nullptr, *d_sqQ = nullptr, *d_sqT = nullptr, *d_sqJ = nullptr, *d_sqJall = nullptr;
    int64_t cap_sqJall = 0;
    int32_t* d_sqF = nullptr;
    int64_t cap_sqp = 0;
    // ... its launch sequence (a few hundred small launches per call) as a hipGraph: recorded the second time a call repeats the
    // previous one's shape, replayed from then on; any reallocation or change of the constants starts over
    hipStream_t stream2 = nullptr;     // the two-stage step: allocation beside kernel 13's hand-over pass
    hipEvent_t ev_fork = nullptr, ev_alloc = nullptr;
    uint64_t alloc_epoch = 0;          // bumped by every (re)allocation of a device buffer
    struct SqpKey {
        int64_t B = -1, xs = 0, us = 0;
        const void *xref = nullptr, *uref = nullptr, *warm = nullptr;
        int32_t iters = 0, backtracks = 0;
        double tol = 0;
        uint64_t epoch = 0, consts = 0;
        bool operator==(const SqpKey& o) const {
            return B == o.B && xs == o.xs && us == o.us && xref == o.xref && uref == o.uref && warm == o.warm && iters == o.iters &&
                   backtracks == o.backtracks && tol == o.tol && epoch == o.epoch && consts == o.consts;
        }
    } sqp_key, sqp_seen;
    hipGraphExec_t sqp_exec = nullptr;
    int64_t sqp_graph_launches = 0;    // (diagnostic: ftmpc_sqp_graph_launches)
    int64_t cap_hullA = 0, cap_wrench = 0;
    // kernel 13: the two-stage form in float64 by the Riccati recursion (no terminal set, N <= 40, up to 128 hull rows)
    double* ricw_slot = nullptr;
    int64_t ricw_slot_doubles = 0;
    int grid_ricw = 0;
    bool wrench_handed = false;        // the last two-stage step ran kernel 11 with its hand-over list (d_qctl[0] = its length)
    // debug
    float *d_dbgH = nullptr, *d_dbgv = nullptr;
    // profiling
    bool profiling = false;
    hipEvent_t ev[2 * FTMPC_KERNEL_SLOTS] = {};  // start/stop per kernel slot
    bool ev_valid = false;
    bool ev_used[FTMPC_KERNEL_SLOTS] = {};
};

namespace {

int fail(ftmpc_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                          \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return fail((h), FTMPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

bool inv3(const double* M, double* out) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (!(std::fabs(det) > 1e-300)) return false;
    const double s = 1.0 / det;
    out[0] = (e * i - f * h) * s; out[1] = (c * h - b * i) * s; out[2] = (b * f - c * e) * s;
    out[3] = (f * g - d * i) * s; out[4] = (a * i - c * g) * s; out[5] = (c * d - a * f) * s;
    out[6] = (d * h - e * g) * s; out[7] = (b * g - a * h) * s; out[8] = (a * e - b * d) * s;
    return true;
}

// lower Cholesky of a 9x9 PSD matrix (zero pivots give zero columns)
bool chol9(const double* P, double* L) {
    std::memset(L, 0, 81 * sizeof(double));
    for (int j = 0; j < 9; ++j) {
        double d = P[9 * j + j];
        for (int k = 0; k < j; ++k) d -= L[9 * j + k] * L[9 * j + k];
        if (d < -1e-9 * std::fabs(P[9 * j + j]) - 1e-300) return false;
        const double l = d > 0 ? std::sqrt(d) : 0.0;
        L[9 * j + j] = l;
        for (int i = j + 1; i < 9; ++i) {
            double s = P[9 * i + j];
            for (int k = 0; k < j; ++k) s -= L[9 * i + k] * L[9 * j + k];
            L[9 * i + j] = l > 0 ? s / l : 0.0;
        }
    }
    return true;
}

int build_consts(const ftmpc_config& c, DeviceConsts& d, std::string& why) {
    if (c.N < 1 || c.N > 64) { why = "N out of range 1..64"; return FTMPC_ERR_ARG; }
    if (c.NT < 1 || c.NT > FTMPC_MAX_NT) { why = "NT out of range 1..16"; return FTMPC_ERR_ARG; }
    if (!(c.dt > 0) || !(c.mass > 0)) { why = "dt and mass must be positive"; return FTMPC_ERR_ARG; }
    if (!(c.rho > 0)) { why = "rho must be positive (strict convexity)"; return FTMPC_ERR_ARG; }
    std::memset(&d, 0, sizeof(d));
    d.N = c.N;
    d.NT = c.NT;
    d.max_iters = c.max_iters > 0 ? c.max_iters : 30;
    if ((c.dtype == FTMPC_DTYPE_F64 || c.N * c.NT > 240) && c.max_iters <= 0) d.max_iters = 30;
    d.dt = c.dt;
    d.inv_mass = 1.0 / c.mass;
    std::memcpy(d.J, c.J, sizeof(d.J));
    if (!inv3(c.J, d.Jinv)) { why = "J is singular"; return FTMPC_ERR_ARG; }
    std::memcpy(d.r, c.r, sizeof(d.r));
    std::memcpy(d.fvirt, c.f_virt, sizeof(d.fvirt));
    // ArT = -[r]x Jinv
    const double* r = c.r;
    const double S[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += S[3 * i + k] * d.Jinv[3 * k + j];
            d.ArT[3 * i + j] = -s;
        }
    for (int g = 0; g < 6; ++g)
        for (int t = 0; t < c.NT; ++t) d.D[g * ftmpc::MAX_NT + t] = c.D[g * c.NT + t];
    for (int i = 0; i < 9; ++i) {
        if (!(c.Q[i] >= 0)) { why = "Q must be non-negative"; return FTMPC_ERR_ARG; }
        d.Q[i] = c.Q[i];
        d.sq2Q[i] = std::sqrt(2.0 * c.Q[i]);
    }
    for (int i = 0; i < 6; ++i) {
        if (!(c.R[i] >= 0)) { why = "R must be non-negative"; return FTMPC_ERR_ARG; }
        d.R[i] = c.R[i];
    }
    std::memcpy(d.P, c.P, sizeof(d.P));
    double L[81];
    if (!chol9(c.P, L)) { why = "P must be symmetric positive semi-definite"; return FTMPC_ERR_ARG; }
    const double s2 = std::sqrt(2.0);
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) d.LPt[9 * i + j] = s2 * L[9 * j + i];  // sqrt(2) L'
    d.rho = c.rho;
    d.mu_stop = c.mu_stop > 0 ? c.mu_stop : ((c.dtype == FTMPC_DTYPE_F64 || c.N * c.NT > 240) ? 1e-13 : 1e-11);
    if (c.terminal_set && !(c.mu_stop > 0)) d.mu_stop = 1e-10;   // general rows: C' W C ruins the conditioning below that
    d.mu_refine = 1e-3;
    return FTMPC_OK;
}

template <typename T>
int grow(ftmpc_handle* h, T** p, int64_t count) {
    ++h->alloc_epoch;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    if (count <= 0) return FTMPC_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();   // the runtime keeps the failure as its "last error": left there, the next launch check would report it
        return fail(h, FTMPC_ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e));
    }
    return FTMPC_OK;
}

int tiles_of(int nb) { return nb * (nb + 1) / 2; }
// per-workgroup global slot of the fp32 kernels (layout: ftmpc_common.h): sweep scratch, then the Hessian tiles
// Linearisation: full records per wave for large batches; below `lin_split_max` instances the direction-split grid
// (13 blocks per 64 instances, ftmpc_linearize.hip), which fills the device from a few hundred instances on.
void launch_linearize(ftmpc_handle* h, int64_t B, int blocks, hipStream_t s, const ftmpc::LinParams& lp) {
    // shares of the 13 directions per 64 instances: as many as keep about one wave per SIMD (1024 on the device)
    const int shares = B <= h->lin_split_max ? 13 : (B <= 2 * h->lin_split_max ? 4 : (B <= 5 * h->lin_split_max ? 2 : 1));
    if (shares == 13)
        hipLaunchKernelGGL((ftmpc::ftmpc_linearize_kernel<double, 1>), dim3(blocks, 13), dim3(64), 0, s, h->dc, lp);
    else if (shares > 1)
        hipLaunchKernelGGL((ftmpc::ftmpc_linearize_kernel<double, 2>), dim3(blocks, shares), dim3(64), 0, s, h->dc, lp);
    else
        hipLaunchKernelGGL((ftmpc::ftmpc_linearize_kernel<double, 0>), dim3(blocks), dim3(64), 0, s, h->dc, lp);
}

int64_t slot_words(int nb, int N) { return ftmpc::slot_tile_off_words(N) + (int64_t)tiles_of(nb) * 256; }

int enqueue(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck,
            const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride,
            const double* warmU, double* out_u0, double* out_U, int32_t* status, int32_t* iters,
            hipStream_t s, int64_t dbg_inst) {
    if (h->tset && !h->tset_thruster)
        return fail(h, FTMPC_ERR_ARG, "the thruster form with the terminal set needs N * NT <= 256 (ftmpc_solve_wrench_batch, the reference's two-stage form, has no such limit)");
    if (B <= 0) return FTMPC_OK;
    LinParams lp;
    lp.B = B;
    lp.x0 = x0; lp.ub = ub; lp.stuck = stuck;
    lp.xref = xref; lp.xref_stride = xref_stride;
    lp.uref = uref; lp.uref_stride = uref_stride;
    lp.warmU = warmU;
    lp.rec = h->rec;
    lp.warmG = nullptr;
    lp.out_eN = h->tset ? h->d_eN : nullptr;
    lp.tcost = h->d_tcost;
    lp.out_cbar = h->sbounds ? h->d_cbar : nullptr;
    const int nvar = h->use_f64 ? 0 : (h->nb_max <= 8 ? 1 : (h->nb_max == 9 ? 2 : 3));   // one-wave fp32 instantiations in use
    lp.qlist = h->use_f64 ? nullptr : h->d_qlist;
    lp.qcount = h->d_qctl;
    lp.qvmax = h->use_wg ? 3 : nvar - 1;   // list 3: ceil(n/16) >= 11, the workgroup kernel
    if (!h->use_f64) HIP_TRY(h, hipMemsetAsync(h->d_qctl, 0, 8 * sizeof(int32_t), s));
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (h->qcnt_pending && !capturing) {
        if (hipEventQuery(h->ev_qcnt) == hipSuccess) {
            for (int v = 0; v < 4; ++v) h->last_cnt[v] = h->h_qcnt[v];
            h->qcnt_pending = false;
            h->qcnt_valid = true;
        } else {
            (void)hipGetLastError();      // (not ready yet is not an error)
        }
    }
    auto grid_for = [&](int v, int64_t full) -> int {   // list v: the full persistent grid, or a small one when the list was empty last step
        const int64_t g = std::min<int64_t>(B, full);
        return (int)((h->qcnt_valid && !capturing && h->last_cnt[v] == 0) ? std::min<int64_t>(g, h->num_cu) : g);     // (one workgroup per CU)
    };
    const int lin_blocks = (int)((B + 63) / 64);
    for (bool& u : h->ev_used) u = false;
    if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[0], s));
    launch_linearize(h, B, lin_blocks, s, lp);
    HIP_TRY(h, hipGetLastError());
    if (h->profiling) {
        HIP_TRY(h, hipEventRecord(h->ev[1], s));
        h->ev_used[0] = true;
    }
    SolveParams sp;
    sp.B = B;
    sp.rec = h->rec;
    sp.ub = ub; sp.stuck = stuck; sp.warmU = warmU;
    sp.out_u0 = out_u0; sp.out_U = out_U; sp.status = status; sp.iters = iters;
    sp.dbg_inst = dbg_inst;
    sp.dbg_H = h->d_dbgH;
    sp.dbg_vec = h->d_dbgv;
    if (h->use_f64 && h->use_ric64) {
        ftmpc::SolveRicParams w;
        sp.hscratch = nullptr;
        sp.tile_words = 0;
        sp.qlist = nullptr;
        sp.qcount = nullptr;
        HIP_TRY(h, hipMemsetAsync(h->d_qctl + 4, 0, sizeof(int32_t), s));
        sp.qhead = h->d_qctl + 4;       // shared instance cursor
        sp.dbg_H = reinterpret_cast<float*>(h->d_dbgH64);     // (diagnostic build: phase stamps)
        w.base = sp;
        w.slot = h->ric_slot;
        w.slot_doubles = h->ric_slot_doubles;
        for (int i = 0; i < FTMPC_NX; ++i) {
            w.xlb[i] = h->cfg.xlb[i];
            w.xub[i] = h->cfg.xub[i];
        }
        w.cbar = h->sbounds ? h->d_cbar : nullptr;
        // with state bounds the iteration stops at mu 1e-10 unless the caller asked otherwise, as the other general-constraint
        // modes do: the barrier weight of an active state row enters the Riccati recursion's state weight (see the kernel)
        DeviceConsts dcr = h->dc;
        if (h->sbounds && !(h->cfg.mu_stop > 0)) dcr.mu_stop = 1e-10;
        const int grid = (int)std::min<int64_t>(B, h->grid_ric);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[12], s));
        if (h->sbounds && h->ric_nv == 6) hipLaunchKernelGGL((ftmpc::ftmpc_solve_ric64_kernel<6, true>), dim3(grid), dim3(64), 0, s, dcr, w);
        else if (h->sbounds) hipLaunchKernelGGL((ftmpc::ftmpc_solve_ric64_kernel<10, true>), dim3(grid), dim3(64), 0, s, dcr, w);
        else if (h->ric_nv == 4) hipLaunchKernelGGL(ftmpc::ftmpc_solve_ric64_kernel<4>, dim3(grid), dim3(64), 0, s, h->dc, w);
        else if (h->ric_nv == 6) hipLaunchKernelGGL(ftmpc::ftmpc_solve_ric64_kernel<6>, dim3(grid), dim3(64), 0, s, h->dc, w);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_ric64_kernel<10>, dim3(grid), dim3(64), 0, s, h->dc, w);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[13], s));
            h->ev_used[6] = true;
            h->ev_valid = true;
        }
        return FTMPC_OK;
    }
    if (h->use_f64 && h->use_ws64) {
        SolveWs64Params w;
        sp.hscratch = nullptr;
        sp.tile_words = 0;
        sp.qlist = nullptr;
        sp.qcount = nullptr;
        sp.qhead = nullptr;
        sp.dbg_H = reinterpret_cast<float*>(h->d_dbgH64);     // (diagnostic build: phase stamps)
        w.base = sp;
        w.slot = h->ws64_slot;
        w.slot_doubles = h->ws64_slot_doubles;
        const int grid = (int)std::min<int64_t>(B, h->grid_ws64);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[12], s));
        if (h->ws64_nvt == 1) hipLaunchKernelGGL(ftmpc::ftmpc_solve_ws64_kernel<1>, dim3(grid), dim3(ftmpc::ws64k::WG), 0, s, h->dc, w);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_ws64_kernel<3>, dim3(grid), dim3(ftmpc::ws64k::WG), 0, s, h->dc, w);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[13], s));
            h->ev_used[6] = true;
            h->ev_valid = true;
        }
        return FTMPC_OK;
    }
    if (h->use_f64) {
        Solve64Params q;
        sp.hscratch = nullptr;
        sp.tile_words = 0;
        sp.qlist = nullptr;
        sp.qcount = nullptr;
        sp.qhead = nullptr;
        q.base = sp;
        q.Hs = h->Hs; q.Ls = h->Ls; q.Eall = h->Eall;
        q.tile_doubles = h->tile_doubles;
        q.e_doubles = h->e_doubles;
        q.npad_max = h->npad_max;
        q.nb_lo = 0;
        q.dbg_H = h->d_dbgH64;
        q.dbg_vec = h->d_dbgv64;
        q.warmG = nullptr; q.hullA = nullptr; q.hull_set = nullptr; q.hullb = nullptr; q.out_tau0 = nullptr; q.out_G = nullptr;
        q.hull_rows = 0;
        q.termA = h->tset ? h->d_term : nullptr;
        q.termb = h->tset ? h->d_term + (int64_t)h->cfg.term_rows * 9 : nullptr;
        q.term_rows = h->tset ? h->cfg.term_rows : 0;
        q.eN = h->d_eN;
        const int grid = (int)std::min<int64_t>(B, h->grid64);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[8], s));
        if (h->tset)
            hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<4, 1, 2>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, h->dc, q);
        else if (h->npad_max <= 256)
            hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<4, 1>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, h->dc, q);
        else if (h->npad_max <= 640)
            hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<ftmpc::f64k::RPF, 3>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, h->dc, q);
        else
            hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<ftmpc::f64k::RPF, ftmpc::f64k::NVT_MAX>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, h->dc, q);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[9], s));
            h->ev_used[4] = true;
            h->ev_valid = true;
        }
        return FTMPC_OK;
    }
    // fp32 instantiations NB = 8, 9, 10: each pulls the instances with ceil(n/16) <= NB (the first also the empty
    // ones, the last also shapes beyond every instantiation, which it reports) from the list the linearise kernel
    // wrote for it; a launch whose list is empty returns at once
    auto launch_ws = [&](int v, int ev_slot) -> int {   // kernel 8 on work list v
        SolveWgParams w;
        w.base = sp;
        w.base.hscratch = nullptr;
        w.base.tile_words = 0;
        w.base.qlist = h->d_qlist + (int64_t)v * B;
        w.base.qcount = h->d_qctl + v;
        w.base.qhead = h->d_qctl + 4 + v;
        w.slot = h->ws_slot;
        w.slot_words = h->ws_slot_words;
        const int grid = grid_for(v, h->grid_ws);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[2 * ev_slot], s));
        if (h->ws_nb == 6) hipLaunchKernelGGL(ftmpc::ftmpc_solve_ws32_kernel<6>, dim3(grid), dim3(ftmpc::wsk::WG), 0, s, h->dc, w);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_ws32_kernel<8>, dim3(grid), dim3(ftmpc::wsk::WG), 0, s, h->dc, w);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[2 * ev_slot + 1], s));
            h->ev_used[ev_slot] = true;
        }
        return FTMPC_OK;
    };
    for (int v = 0; v < nvar; ++v) {
        const int NBv = 8 + v;
        sp.hscratch = h->hs[v];
        sp.tile_words = slot_words(NBv, h->dc.N);
        sp.qlist = h->d_qlist + (int64_t)v * B;
        sp.qcount = h->d_qctl + v;
        sp.qhead = h->d_qctl + 4 + v;
        const int grid = grid_for(v, h->grid[v]);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[2 + 2 * v], s));
        if (v == 0) hipLaunchKernelGGL(ftmpc::ftmpc_solve_f32_kernel<8>, dim3(grid), dim3(64), 0, s, h->dc, sp);
        else if (v == 1) hipLaunchKernelGGL(ftmpc::ftmpc_solve_f32_kernel<9>, dim3(grid), dim3(64), 0, s, h->dc, sp);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_f32_kernel<10>, dim3(grid), dim3(64), 0, s, h->dc, sp);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[3 + 2 * v], s));
            h->ev_used[1 + v] = true;
        }
    }
    if (h->use_wg && h->use_wsw) {     // kernel 10 on work list 3
        sp.hscratch = h->wsw_slot;
        sp.tile_words = h->wsw_slot_words;
        sp.qlist = h->d_qlist + (int64_t)3 * B;
        sp.qcount = h->d_qctl + 3;
        sp.qhead = h->d_qctl + 7;
        const int grid = grid_for(3, h->grid_wsw);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[10], s));
        if (h->ws_nb == 6) hipLaunchKernelGGL(ftmpc::ftmpc_solve_wsw32_kernel<6>, dim3(grid), dim3(64), 0, s, h->dc, sp);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_wsw32_kernel<8>, dim3(grid), dim3(64), 0, s, h->dc, sp);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[11], s));
            h->ev_used[5] = true;
        }
    } else if (h->use_wg && h->use_ws) {
        const int rc = launch_ws(3, 5);
        if (rc != FTMPC_OK) return rc;
    } else if (h->use_wg) {
        SolveWgParams w;
        w.base = sp;
        w.base.hscratch = nullptr;
        w.base.tile_words = 0;
        w.base.qlist = h->d_qlist + (int64_t)3 * B;
        w.base.qcount = h->d_qctl + 3;
        w.base.qhead = h->d_qctl + 7;
        w.slot = h->wg_slot;
        w.slot_words = h->wg_slot_words;
        const int grid = grid_for(3, h->grid_wg);
        if (h->profiling) HIP_TRY(h, hipEventRecord(h->ev[10], s));
        hipLaunchKernelGGL(ftmpc::ftmpc_solve_wg32_kernel<15>, dim3(grid), dim3(ftmpc::wgk::WG), 0, s, h->dc, w);
        HIP_TRY(h, hipGetLastError());
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(h->ev[11], s));
            h->ev_used[5] = true;
        }
    }
    if (h->profiling) h->ev_valid = true;
    if (h->h_qcnt && !capturing && !h->qcnt_pending) {     // this step's list lengths, for the next one
        HIP_TRY(h, hipMemcpyAsync(h->h_qcnt, h->d_qctl, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipEventRecord(h->ev_qcnt, s));
        h->qcnt_pending = true;
    }
    return FTMPC_OK;
}

}  // namespace

extern "C" {

int32_t ftmpc_version(void) { return 430; }

#ifndef FTMPC_BUILD_ID
#define FTMPC_BUILD_ID "unknown"
#endif
const char* ftmpc_build_id(void) { return FTMPC_BUILD_ID; }

int ftmpc_default_config(ftmpc_config* cfg, int32_t N, int32_t NT) {
    if (!cfg || N < 1 || N > 64 || NT < 1 || NT > FTMPC_MAX_NT) return FTMPC_ERR_ARG;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (int32_t)sizeof(ftmpc_config);
    cfg->N = N;
    cfg->NT = NT;
    cfg->dtype = FTMPC_DTYPE_F32;
    cfg->max_iters = 30;
    cfg->device_id = 0;
    cfg->dt = 0.1;                                   // reactive.yaml:2
    cfg->mass = 16.8;                                // sys_model.py:52
    cfg->J[0] = 0.2; cfg->J[4] = 0.3; cfg->J[8] = 0.25;  // sys_model.py:53-57
    const double Q[9] = {1, 1, 1, 1, 1, 1, 2, 2, 2};       // reactive.yaml:32
    const double R[6] = {0.1, 0.1, 0.1, 0.01, 0.01, 0.01}; // reactive.yaml:33
    std::memcpy(cfg->Q, Q, sizeof(Q));
    std::memcpy(cfg->R, R, sizeof(R));
    // quadratic part of config/terminal.yaml
    const double pp = 19.574136382485836, pv = 28.1433488118291, vv = 98.382994426126402;
    const double po[3] = {645.23036107820451, 645.43124119008723, 645.70261416462426};
    for (int a = 0; a < 3; ++a) {
        cfg->P[9 * a + a] = pp;
        cfg->P[9 * a + 3 + a] = pv;
        cfg->P[9 * (3 + a) + a] = pv;
        cfg->P[9 * (3 + a) + 3 + a] = vv;
        cfg->P[9 * (6 + a) + 6 + a] = po[a];
    }
    // spiral_parameters.py:33-39: omega_des = [0,0,.6], f_virt = 3.5 y, r = |f_virt|/(m |omega_des|^2) y
    cfg->f_virt[1] = 3.5;
    cfg->r[1] = 3.5 / (cfg->mass * 0.6 * 0.6);
    cfg->rho = 0.05;
    cfg->mu_stop = 0.0;  /* library default by dtype */
    for (int i = 0; i < FTMPC_NX; ++i) {      // no state bounds (the reference's default: params "xub" / "xlb" are None)
        cfg->xlb[i] = -FTMPC_NO_BOUND;
        cfg->xub[i] = FTMPC_NO_BOUND;
    }
    if (NT == 16) {
        // sys_model.py:73-123 restated from the thruster geometry
        const double d1 = 0.12, d2 = 0.09, d3 = 0.05;
        const double fx[8] = {-1, -1, 1, 1, -1, -1, 1, 1};
        const double ty[8] = {-1, 1, 1, -1, -1, 1, 1, -1};
        const double tz[8] = {1, 1, -1, -1, -1, -1, 1, 1};
        for (int i = 0; i < 8; ++i) {
            cfg->D[0 * 16 + i] = fx[i];
            cfg->D[4 * 16 + i] = d3 * ty[i];
            cfg->D[5 * 16 + i] = d1 * tz[i];
        }
        const double fy[4] = {-1, -1, 1, 1}, tzy[4] = {-1, 1, 1, -1};
        const double fz[4] = {-1, 1, -1, 1}, txz[4] = {-1, 1, 1, -1};
        for (int i = 0; i < 4; ++i) {
            cfg->D[1 * 16 + 8 + i] = fy[i];
            cfg->D[5 * 16 + 8 + i] = d2 * tzy[i];
            cfg->D[2 * 16 + 12 + i] = fz[i];
            cfg->D[3 * 16 + 12 + i] = d1 * txz[i];
        }
    }
    return FTMPC_OK;
}

int ftmpc_create(const ftmpc_config* cfg, ftmpc_handle** out) {
    if (!cfg || !out) return fail(nullptr, FTMPC_ERR_ARG, "null argument");
    *out = nullptr;
    // ABI guard first: nothing beyond the first 24 bytes of *cfg is read before the caller's struct is known to be ours
    if (cfg->struct_size != (int32_t)sizeof(ftmpc_config))
        return fail(nullptr, FTMPC_ERR_ARG, "ftmpc_config.struct_size is " + std::to_string(cfg->struct_size) + ", this library expects " +
                                                std::to_string(sizeof(ftmpc_config)) + " (caller built against another include/ftmpc.h? use ftmpc_default_config)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, FTMPC_ERR_NODEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, FTMPC_ERR_ARG, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess)
        return fail(nullptr, FTMPC_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FTMPC_ERR_NODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    if (cfg->dtype != FTMPC_DTYPE_F32 && cfg->dtype != FTMPC_DTYPE_F64)
        return fail(nullptr, FTMPC_ERR_ARG, "dtype must be FTMPC_DTYPE_F32 or FTMPC_DTYPE_F64");
    if (cfg->kernel_select != FTMPC_KERNEL_AUTO && cfg->kernel_select != FTMPC_KERNEL_DENSE && cfg->kernel_select != FTMPC_KERNEL_WORKGROUP)
        return fail(nullptr, FTMPC_ERR_ARG, "kernel_select must be FTMPC_KERNEL_AUTO, FTMPC_KERNEL_DENSE or FTMPC_KERNEL_WORKGROUP");
    if (cfg->stage_chunks < 0 || cfg->stage_chunks > ftmpc_handle::MAX_CHUNKS)
        return fail(nullptr, FTMPC_ERR_ARG, "stage_chunks out of range 0..8");
    ftmpc_handle* h = new (std::nothrow) ftmpc_handle();
    if (!h) return fail(nullptr, FTMPC_ERR_ALLOC, "out of host memory");
    h->cfg = *cfg;
    std::string why;
    int rc = build_consts(*cfg, h->dc, why);
    if (rc != FTMPC_OK) {
        delete h;
        return fail(nullptr, rc, why);
    }
    h->nb_max = (cfg->N * cfg->NT + 15) / 16;
    if (cfg->N * cfg->NT > ftmpc::f64k::NMAX) {
        delete h;
        return fail(nullptr, FTMPC_ERR_ARG, "N*NT > 1024 is not supported");
    }
    // fp32 LDS-resident kernels cover n <= 160; larger problems and dtype F64 use the float64
    // workgroup-per-instance kernel
    // fp32: one-wave register-resident kernels up to n = 160, the workgroup kernel with the factor in LDS up to n = 240;
    // beyond that, and for dtype F64, the float64 workgroup kernel with its tiles in a global slot
    // kernel 8 (the thruster QP through wrench space) takes the workgroup kernel's list -- ceil(n / 16) >= 11 -- when the
    // wrench-space system fits eight tiles a side (N <= 21) and the thruster variables fit its threads (one per thread up
    // to N = 16, two beyond); kernel_select = FTMPC_KERNEL_DENSE: kernel 7 (n <= 240) or the float64 kernel instead.
    // The one-wave kernels keep n <= 160: a workgroup per instance does not compete with a wave per instance there.
    h->ws_nb = (6 * cfg->N <= 96) ? 6 : 8;
    // the terminal set: the thruster-space solve with those rows exists in float64 only, so an fp32 handle that asks for it
    // solves THAT form on the float64 kernel; its two-stage form (ftmpc_solve_wrench_batch) still runs on kernel 11
    const bool solve_f64 = cfg->dtype == FTMPC_DTYPE_F64 || cfg->terminal_set != 0 || cfg->state_bounds != 0;
    h->use_ws = !solve_f64 && cfg->kernel_select != FTMPC_KERNEL_DENSE && h->nb_max > 10 && 6 * cfg->N <= 128 &&
                cfg->N * cfg->NT <= ftmpc::wsk::WG * ftmpc::wsk::nvt_of(h->ws_nb);
    // ... on one wave per instance (kernel 10) when the thruster variables fit four (N <= 16) / six (N <= 21) per lane;
    // kernel_select = FTMPC_KERNEL_WORKGROUP keeps the workgroup-per-instance kernel 8
    h->use_wsw = h->use_ws && cfg->kernel_select != FTMPC_KERNEL_WORKGROUP && cfg->N * cfg->NT <= 64 * ftmpc::wswk::nvt_of(h->ws_nb);
    h->use_f64 = solve_f64 || (h->nb_max > 15 && !h->use_ws);
    h->use_wg = !h->use_f64 && h->nb_max > 10;
    // float64: through the wrench-space form where the thrusters outnumber the wrench components by enough to pay for the
    // assembly of K (ten or more thrusters), the wrench-space system fits sixteen tiles a side and three thruster variables
    // per thread cover N * NT; the terminal-set mode and kernel_select = FTMPC_KERNEL_DENSE keep the dense float64 kernel
    h->use_ws64 = h->use_f64 && cfg->terminal_set == 0 && cfg->kernel_select != FTMPC_KERNEL_DENSE && 6 * cfg->N <= ftmpc::ws64k::NPADW &&
                  cfg->N * cfg->NT <= 3 * ftmpc::ws64k::WG && cfg->NT >= 10;
    h->ws64_nvt = (cfg->N * cfg->NT <= ftmpc::ws64k::WG) ? 1 : 3;
    // float64 box QP (no terminal set): the Riccati recursion on one wave per instance (kernel 12) for every horizon up to 40;
    // kernel_select = FTMPC_KERNEL_WORKGROUP keeps kernel 9 (the wrench-space form), FTMPC_KERNEL_DENSE the dense float64 kernel
    h->sbounds = cfg->state_bounds != 0;
    if (h->sbounds && (cfg->terminal_set != 0 || cfg->N > 40)) {
        delete h;
        return fail(nullptr, FTMPC_ERR_ARG, "state_bounds needs N <= 40 and no terminal_set (the state-bound rows live on the Riccati kernel)");
    }
    if (h->sbounds)
        for (int i = 0; i < FTMPC_NX; ++i)
            if (!(cfg->xlb[i] < cfg->xub[i])) {
                delete h;
                return fail(nullptr, FTMPC_ERR_ARG, "state_bounds: xlb[i] < xub[i] is required for every component (use +-FTMPC_NO_BOUND for none)");
            }
    h->use_ric64 = h->use_f64 && cfg->terminal_set == 0 && (cfg->kernel_select == FTMPC_KERNEL_AUTO || h->sbounds) && cfg->N <= 40;
    h->ric_nv = (cfg->N <= 16 && !h->sbounds) ? 4 : (cfg->N <= 24 ? 6 : 10);
    if (h->use_ric64) h->use_ws64 = false;
    h->tset = cfg->terminal_set != 0;
    if (h->tset) {
        const char* why = nullptr;
        if (cfg->term_rows < 1 || cfg->term_rows > FTMPC_MAX_TERM_ROWS) why = "term_rows out of range 1..80";
        h->tset_thruster = 16 * h->nb_max <= 256;      // (the THRUSTER form with terminal rows lives on the dense float64 kernel's n <= 256 mode; the
                                                       //  two-stage form -- ftmpc_solve_wrench_batch -- has no such limit: kernel 13)
        if (why) {
            delete h;
            return fail(nullptr, FTMPC_ERR_ARG, why);
        }
    }
    h->npad_max = 16 * h->nb_max;
    h->device = cfg->device_id;
    h->num_cu = prop.multiProcessorCount;
    if (hipSetDevice(h->device) != hipSuccess) {
        delete h;
        return fail(nullptr, FTMPC_ERR_HIP, "hipSetDevice failed");
    }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete h;
        return fail(nullptr, FTMPC_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < 2 * FTMPC_KERNEL_SLOTS; ++i) (void)hipEventCreate(&h->ev[i]);
    bool sbad = hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking) != hipSuccess;
    for (int i = 0; i < ftmpc_handle::MAX_CHUNKS; ++i)
        sbad = sbad || hipEventCreateWithFlags(&h->ev_in[i], hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&h->ev_k[i], hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&h->ev_out[i], hipEventDisableTiming) != hipSuccess;
    if (hipHostMalloc(reinterpret_cast<void**>(&h->h_qcnt), 4 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_qcnt, hipEventDisableTiming) != hipSuccess) {
        if (h->h_qcnt) (void)hipHostFree(h->h_qcnt);
        h->h_qcnt = nullptr;      // (no hint: every launch takes its full grid)
        (void)hipGetLastError();
    }
    if (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_alloc, hipEventDisableTiming) != hipSuccess) {
        if (h->stream2) (void)hipStreamDestroy(h->stream2);
        h->stream2 = nullptr;      // (no overlap: the two-stage step allocates after kernel 13, on the one stream)
        (void)hipGetLastError();
    }
    if (sbad || grow(h, &h->d_qctl, 8) != FTMPC_OK) {
        g_create_error = "stream / event / work-list allocation failed";
        ftmpc_destroy(h);
        return FTMPC_ERR_HIP;
    }
    if (cfg->lin_split_max != 0) h->lin_split_max = cfg->lin_split_max < 0 ? 0 : cfg->lin_split_max;
    if (cfg->stage_chunks >= 1 && cfg->stage_chunks <= ftmpc_handle::MAX_CHUNKS) h->stage_chunks = cfg->stage_chunks;
    // persistent grids: resident workgroups per CU from the occupancy query (LDS-bound)
    int per[3] = {0, 0, 0};
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[0], ftmpc::ftmpc_solve_f32_kernel<8>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[1], ftmpc::ftmpc_solve_f32_kernel<9>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per[2], ftmpc::ftmpc_solve_f32_kernel<10>, 64, 0);
    for (int v = 0; v < 3; ++v) h->grid[v] = h->num_cu * (per[v] < 1 ? 1 : per[v]);
    int per64 = 0;
    if (16 * h->nb_max <= 256)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per64, ftmpc::ftmpc_solve_f64_kernel<4, 1>, ftmpc::f64k::WG, 0);
    else if (16 * h->nb_max <= 640)
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per64, ftmpc::ftmpc_solve_f64_kernel<ftmpc::f64k::RPF, 3>, ftmpc::f64k::WG, 0);
    else
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per64, ftmpc::ftmpc_solve_f64_kernel<ftmpc::f64k::RPF, ftmpc::f64k::NVT_MAX>, ftmpc::f64k::WG, 0);
    if (per64 < 1) per64 = 1;
    if (per64 > 2) per64 = 2;
    if (h->tset) per64 = 1;     // the general-constraint instantiations hold ~50 KiB of LDS and one workgroup per CU
    h->grid64 = h->num_cu * per64;
    h->tile_doubles = (int64_t)tiles_of(h->nb_max) * 256;
    h->e_doubles = (int64_t)(cfg->N + 2) * 9 * h->npad_max;     // + raw terminal rows GN and the terminal-set panel
    bool bad = false;
    if (h->use_f64) {
        bad = grow(h, &h->Hs, h->grid64 * h->tile_doubles) != FTMPC_OK || grow(h, &h->Ls, h->grid64 * h->tile_doubles) != FTMPC_OK ||
              grow(h, &h->Eall, h->grid64 * h->e_doubles) != FTMPC_OK ||
              grow(h, &h->d_dbgH64, (int64_t)h->npad_max * h->npad_max) != FTMPC_OK ||
              grow(h, &h->d_dbgv64, 3 * (int64_t)h->npad_max + 4) != FTMPC_OK;
        if (!bad && h->use_ric64) {
            int per = 0;
            if (h->sbounds && h->ric_nv == 6) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ric64_kernel<6, true>, 64, 0);
            else if (h->sbounds) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ric64_kernel<10, true>, 64, 0);
            else if (h->ric_nv == 4) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ric64_kernel<4>, 64, 0);
            else if (h->ric_nv == 6) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ric64_kernel<6>, 64, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ric64_kernel<10>, 64, 0);
            h->grid_ric = h->num_cu * std::max(1, per);
            h->ric_slot_doubles = ftmpc::rick::slot_doubles(cfg->N);
            bad = grow(h, &h->ric_slot, (int64_t)h->grid_ric * h->ric_slot_doubles) != FTMPC_OK;
        }
        if (!bad && h->use_ws64) {
            int per = 0;
            if (h->ws64_nvt == 1) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ws64_kernel<1>, ftmpc::ws64k::WG, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ws64_kernel<3>, ftmpc::ws64k::WG, 0);
            h->grid_ws64 = h->num_cu * std::max(1, std::min(per, FTMPC_WS64_WPC));
            h->ws64_slot_doubles = ftmpc::ws64k::slot_doubles(cfg->N);
            bad = grow(h, &h->ws64_slot, (int64_t)h->grid_ws64 * h->ws64_slot_doubles) != FTMPC_OK;
        }
        if (!bad && h->tset) {
            bad = grow(h, &h->d_term, (int64_t)cfg->term_rows * 10) != FTMPC_OK;
            if (!bad) {
                std::vector<double> t((size_t)cfg->term_rows * 10);
                std::memcpy(t.data(), cfg->term_A, (size_t)cfg->term_rows * 9 * sizeof(double));
                std::memcpy(t.data() + (size_t)cfg->term_rows * 9, cfg->term_b, (size_t)cfg->term_rows * sizeof(double));
                bad = hipMemcpy(h->d_term, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess;
            }
        }
    } else {
        bad = grow(h, &h->hs[0], (int64_t)h->grid[0] * slot_words(8, cfg->N)) != FTMPC_OK ||
              (h->nb_max > 8 && grow(h, &h->hs[1], (int64_t)h->grid[1] * slot_words(9, cfg->N)) != FTMPC_OK) ||
              (h->nb_max > 9 && grow(h, &h->hs[2], (int64_t)h->grid[2] * slot_words(10, cfg->N)) != FTMPC_OK) ||
              grow(h, &h->d_dbgH, 4096 * 24 + 256 * 256) != FTMPC_OK || grow(h, &h->d_dbgv, 3 * 256 + 4) != FTMPC_OK;
        if (!bad && h->use_ws) {
            int per = 0;
            if (h->ws_nb == 6) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ws32_kernel<6>, ftmpc::wsk::WG, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ws32_kernel<8>, ftmpc::wsk::WG, 0);
            h->grid_ws = h->num_cu * (per > 0 ? per : 1);
            h->ws_slot_words = ftmpc::wsk::slot_words(h->ws_nb, cfg->N);
            bad = grow(h, &h->ws_slot, (int64_t)h->grid_ws * h->ws_slot_words) != FTMPC_OK;
        }
        if (!bad && h->use_wsw) {
            int per = 0;
            if (h->ws_nb == 6) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_wsw32_kernel<6>, 64, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_wsw32_kernel<8>, 64, 0);
            h->grid_wsw = h->num_cu * (per > 0 ? per : 1);
            h->wsw_slot_words = ftmpc::wswk::slot_words(h->ws_nb, cfg->N);
            bad = grow(h, &h->wsw_slot, (int64_t)h->grid_wsw * h->wsw_slot_words) != FTMPC_OK;
        }
        if (!bad && h->use_wg) {
            h->grid_wg = h->num_cu;      // ~150 KiB of LDS: one workgroup per CU
            h->wg_slot_words = ftmpc::wgk::slot_words(15, cfg->N);
            bad = grow(h, &h->wg_slot, (int64_t)h->grid_wg * h->wg_slot_words) != FTMPC_OK;
        }
    }
    if (!bad && cfg->terminal_cost_terms) {
        if (cfg->tc_npoly < 0 || cfg->tc_npoly > FTMPC_MAX_TCOST_TERMS || cfg->tc_nroot < 0 || cfg->tc_nroot > FTMPC_MAX_TCOST_TERMS) {
            ftmpc_destroy(h);
            return fail(nullptr, FTMPC_ERR_ARG, "tc_npoly / tc_nroot out of range 0..24");
        }
        TermCost t;
        std::memset(&t, 0, sizeof(t));
        t.npoly = cfg->tc_npoly;
        t.nroot = cfg->tc_nroot;
        std::memcpy(t.poly_coef, cfg->tc_poly_coef, sizeof(t.poly_coef));
        std::memcpy(t.root_coef, cfg->tc_root_coef, sizeof(t.root_coef));
        std::memcpy(t.root_eps, cfg->tc_root_eps, sizeof(t.root_eps));
        std::memcpy(t.root_pow, cfg->tc_root_pow, sizeof(t.root_pow));
        for (int i = 0; i < FTMPC_MAX_TCOST_TERMS * 9; ++i) {
            t.poly_exp[i] = cfg->tc_poly_exp[i];
            t.root_exp[i] = cfg->tc_root_exp[i];
            if (t.poly_exp[i] < 0 || t.poly_exp[i] > 16 || t.root_exp[i] < 0 || t.root_exp[i] > 16) bad = true;
        }
        t.cconst = cfg->tc_const;
        void* p = nullptr;
        bad = bad || hipMalloc(&p, sizeof(TermCost)) != hipSuccess;
        h->d_tcost = static_cast<TermCost*>(p);
        bad = bad || hipMemcpy(h->d_tcost, &t, sizeof(TermCost), hipMemcpyHostToDevice) != hipSuccess;
        if (bad) h->err = "terminal-cost tables: bad exponent or allocation failure";
    }
    if (bad) {
        g_create_error = h->err;
        ftmpc_destroy(h);
        return FTMPC_ERR_ALLOC;
    }
    *out = h;
    return FTMPC_OK;
}

int ftmpc_destroy(ftmpc_handle* h) {
    if (!h) return FTMPC_OK;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->rec, h->d_x0, h->d_ub, h->d_stuck, h->d_xref, h->d_uref, h->d_warm, h->d_u0, h->d_U,
                    h->d_status, h->d_iters, h->hs[0], h->hs[1], h->hs[2], h->d_dbgH, h->d_dbgv, h->Hs, h->Ls, h->Eall, h->d_dbgH64, h->d_dbgv64,
                    h->d_atau, h->d_aub, h->d_au, h->d_ast, h->d_ait, h->d_qlist, h->d_qctl, h->d_term, h->d_eN, h->gHs, h->gLs,
                    h->gEall, h->wg_slot, h->ws_slot, h->ws64_slot, h->ric_slot, h->ricw_slot, h->d_cbar, h->wsw_slot, h->hull_slot, h->d_tcost, h->d_cost, h->d_sqU, h->d_sqQ, h->d_sqT, h->d_sqJ, h->d_sqJall, h->d_sqF, h->d_hullA, h->d_hullb, h->d_warmG, h->d_tau0, h->d_G, h->d_taud, h->d_hullset, h->d_ast2};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (h->sqp_exec) (void)hipGraphExecDestroy(h->sqp_exec);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_alloc) (void)hipEventDestroy(h->ev_alloc);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->h_qcnt) (void)hipHostFree(h->h_qcnt);
    if (h->ev_qcnt) (void)hipEventDestroy(h->ev_qcnt);
    if (h->pin_in.p) (void)hipHostFree(h->pin_in.p);
    if (h->pin_out.p) (void)hipHostFree(h->pin_out.p);
    for (int i = 0; i < 2 * FTMPC_KERNEL_SLOTS; ++i)
        if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < ftmpc_handle::MAX_CHUNKS; ++i) {
        if (h->ev_in[i]) (void)hipEventDestroy(h->ev_in[i]);
        if (h->ev_k[i]) (void)hipEventDestroy(h->ev_k[i]);
        if (h->ev_out[i]) (void)hipEventDestroy(h->ev_out[i]);
    }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->s_in) (void)hipStreamDestroy(h->s_in);
    if (h->s_out) (void)hipStreamDestroy(h->s_out);
    delete h;
    return FTMPC_OK;
}

const char* ftmpc_last_error(const ftmpc_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int ftmpc_reserve(ftmpc_handle* h, int64_t max_batch) {
    if (!h || max_batch < 0) return FTMPC_ERR_ARG;
    if (max_batch <= h->cap_batch) return FTMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int N = h->cfg.N, NT = h->cfg.NT;
    const int64_t B = max_batch;
    float* recf = nullptr;
    // every buffer below is freed before it is re-allocated: until all of them exist again the handle holds NO batch
    // capacity, so a failure part-way (out of memory) makes the next call grow everything again instead of launching on
    // freed pointers
    h->cap_batch = 0;
    if (h->rec) (void)hipFree(h->rec);
    h->rec = nullptr;
    int rc = grow(h, &recf, B * N * ftmpc::REC_STRIDE * 2);   // float64 records
    if (rc != FTMPC_OK) return rc;
    h->rec = recf;
    if ((rc = grow(h, &h->d_x0, B * 13)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_ub, B * NT)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_stuck, B * NT)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_warm, B * N * NT)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_u0, B * NT)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_U, B * N * NT)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_status, B)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_iters, B)) != FTMPC_OK) return rc;
    if (!h->use_f64 && (rc = grow(h, &h->d_qlist, 4 * B)) != FTMPC_OK) return rc;
    if ((rc = grow(h, &h->d_eN, B * 9)) != FTMPC_OK) return rc;
    if (h->sbounds && (rc = grow(h, &h->d_cbar, B * h->cfg.N * 13)) != FTMPC_OK) return rc;
    h->cap_batch = B;
    return FTMPC_OK;
}

static int stage_refs(ftmpc_handle* h, int64_t B, const double* xref, int64_t xref_stride, const double* uref,
                      int64_t uref_stride) {
    const int N = h->cfg.N;
    const int64_t nx = xref_stride == 0 ? 9 * (N + 1) : B * xref_stride;
    if (nx > h->cap_xref) {
        h->cap_xref = 0;   // (a failed growth must not leave a stale capacity)
        int rc = grow(h, &h->d_xref, nx);
        if (rc != FTMPC_OK) return rc;
        h->cap_xref = nx;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_xref, xref, nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (uref) {
        const int64_t nu = uref_stride == 0 ? 6 * (N + 1) : B * uref_stride;
        if (nu > h->cap_uref) {
            h->cap_uref = 0;   // (a failed growth must not leave a stale capacity)
            int rc = grow(h, &h->d_uref, nu);
            if (rc != FTMPC_OK) return rc;
            h->cap_uref = nu;
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_uref, uref, nu * sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    return FTMPC_OK;
}

static int check_strides(ftmpc_handle* h, int64_t xref_stride, int64_t uref_stride, const double* uref) {
    const int N = h->cfg.N;
    if (xref_stride != 0 && xref_stride < 9 * (N + 1)) return fail(h, FTMPC_ERR_ARG, "xref_stride must be 0 or >= 9*(N+1)");
    if (uref && uref_stride != 0 && uref_stride < 6 * (N + 1)) return fail(h, FTMPC_ERR_ARG, "uref_stride must be 0 or >= 6*(N+1)");
    return FTMPC_OK;
}

static int pin_grow(ftmpc_handle* h, ftmpc_handle::Pinned& P, size_t bytes) {
    if (bytes <= P.bytes) return FTMPC_OK;
    if (P.p) (void)hipHostFree(P.p);
    P.p = nullptr;
    P.bytes = 0;
    const size_t want = bytes + bytes / 8;
    hipError_t e = hipHostMalloc(&P.p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, FTMPC_ERR_ALLOC, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    }
    P.bytes = want;
    return FTMPC_OK;
}

// Host-buffer entry: the caller's (pageable) arrays go through PINNED staging buffers in up to `stage_chunks`
// contiguous instance ranges, each on its way independently: host copy -> H2D on the input stream -> kernels on
// the compute stream -> D2H on the output stream -> host copy, so that the copies of one range run under the
// kernels of its neighbours (pinned staging, async H2D -> kernel -> D2H).
int ftmpc_solve_batch(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck,
                      const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride,
                      double* warmU, double* out_u0, double* out_U, int32_t* status, int32_t* iters) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !out_u0) return fail(h, FTMPC_ERR_ARG, "null buffer or negative batch");
    if (B == 0) return FTMPC_OK;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    const int N = h->cfg.N, NT = h->cfg.NT;
    const int64_t nw = (int64_t)N * NT;
    const bool wantU = out_U != nullptr || warmU != nullptr;
    // reference windows on the device (shared: once; per instance: with the ranges below)
    const int64_t nxr = xref_stride == 0 ? 9 * (N + 1) : B * xref_stride;
    const int64_t nur = !uref ? 0 : (uref_stride == 0 ? 6 * (N + 1) : B * uref_stride);
    if (nxr > h->cap_xref) {
        h->cap_xref = 0;   // (a failed growth must not leave a stale capacity)
        if ((rc = grow(h, &h->d_xref, nxr)) != FTMPC_OK) return rc;
        h->cap_xref = nxr;
    }
    if (nur > h->cap_uref) {
        h->cap_uref = 0;   // (a failed growth must not leave a stale capacity)
        if ((rc = grow(h, &h->d_uref, nur)) != FTMPC_OK) return rc;
        h->cap_uref = nur;
    }
    // pinned mirrors: inputs [x0 | ub | stuck | warm | xref | uref], outputs [u0 | U | status | iters]
    const int64_t o_x0 = 0, o_ub = o_x0 + B * 13, o_st = o_ub + B * NT, o_wm = o_st + B * NT, o_xr = o_wm + (warmU ? B * nw : 0),
                  o_ur = o_xr + nxr, in_words = o_ur + nur;
    const int64_t p_u0 = 0, p_U = p_u0 + B * NT, out_words = p_U + (wantU ? B * nw : 0);
    if ((rc = pin_grow(h, h->pin_in, (size_t)in_words * 8)) != FTMPC_OK) return rc;
    if ((rc = pin_grow(h, h->pin_out, (size_t)out_words * 8 + (size_t)B * 8)) != FTMPC_OK) return rc;
    double* pin = static_cast<double*>(h->pin_in.p);
    double* pout = static_cast<double*>(h->pin_out.p);
    int32_t* pst = reinterpret_cast<int32_t*>(pout + out_words);
    int32_t* pit = pst + B;
    // Measured at B = 65 536 (scripts/host_entry_perf.py): one range 3.20 M QP/s, four ranges 2.74 M -- a range of 16 384
    // instances leaves the persistent grid with a ragged tail four times per call; so ranges are whole 65 536-blocks
    // unless ftmpc_config.stage_chunks says otherwise.
    int nch = h->stage_chunks > 0 ? (int)std::min<int64_t>(h->stage_chunks, (B + 16383) / 16384)
                                  : (int)std::min<int64_t>(ftmpc_handle::MAX_CHUNKS, (B + 65535) / 65536);
    if (nch < 1) nch = 1;
    int64_t edge[ftmpc_handle::MAX_CHUNKS + 1];
    for (int c = 0; c <= nch; ++c) edge[c] = (c == nch) ? B : ((B * c / nch) / 64) * 64;
    auto up = [&](double* dst_dev, int64_t poff, const double* src, int64_t off, int64_t cnt) -> hipError_t {
        std::memcpy(pin + poff + off, src + off, (size_t)cnt * 8);
        return hipMemcpyAsync(dst_dev + off, pin + poff + off, (size_t)cnt * 8, hipMemcpyHostToDevice, h->s_in);
    };
    for (int c = 0; c < nch; ++c) {
        const int64_t lo = edge[c], cnt = edge[c + 1] - edge[c];
        if (cnt <= 0) {
            HIP_TRY(h, hipEventRecord(h->ev_out[c], h->s_out));
            continue;
        }
        HIP_TRY(h, up(h->d_x0, o_x0, x0, lo * 13, cnt * 13));
        HIP_TRY(h, up(h->d_ub, o_ub, ub, lo * NT, cnt * NT));
        HIP_TRY(h, up(h->d_stuck, o_st, stuck, lo * NT, cnt * NT));
        if (warmU) HIP_TRY(h, up(h->d_warm, o_wm, warmU, lo * nw, cnt * nw));
        if (xref_stride != 0) HIP_TRY(h, up(h->d_xref, o_xr, xref, lo * xref_stride, cnt * xref_stride));
        else if (c == 0) HIP_TRY(h, up(h->d_xref, o_xr, xref, 0, nxr));
        if (uref && uref_stride != 0) HIP_TRY(h, up(h->d_uref, o_ur, uref, lo * uref_stride, cnt * uref_stride));
        else if (uref && c == 0) HIP_TRY(h, up(h->d_uref, o_ur, uref, 0, nur));
        HIP_TRY(h, hipEventRecord(h->ev_in[c], h->s_in));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_in[c], 0));
        rc = enqueue(h, cnt, h->d_x0 + lo * 13, h->d_ub + lo * NT, h->d_stuck + lo * NT, h->d_xref + lo * xref_stride, xref_stride,
                     uref ? h->d_uref + lo * uref_stride : nullptr, uref_stride, warmU ? h->d_warm + lo * nw : nullptr,
                     h->d_u0 + lo * NT, wantU ? h->d_U + lo * nw : nullptr, h->d_status + lo, h->d_iters + lo, h->stream, -1);
        if (rc != FTMPC_OK) return rc;
        HIP_TRY(h, hipEventRecord(h->ev_k[c], h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->s_out, h->ev_k[c], 0));
        HIP_TRY(h, hipMemcpyAsync(pout + p_u0 + lo * NT, h->d_u0 + lo * NT, (size_t)cnt * NT * 8, hipMemcpyDeviceToHost, h->s_out));
        if (wantU) HIP_TRY(h, hipMemcpyAsync(pout + p_U + lo * nw, h->d_U + lo * nw, (size_t)cnt * nw * 8, hipMemcpyDeviceToHost, h->s_out));
        if (status) HIP_TRY(h, hipMemcpyAsync(pst + lo, h->d_status + lo, (size_t)cnt * 4, hipMemcpyDeviceToHost, h->s_out));
        if (iters) HIP_TRY(h, hipMemcpyAsync(pit + lo, h->d_iters + lo, (size_t)cnt * 4, hipMemcpyDeviceToHost, h->s_out));
        HIP_TRY(h, hipEventRecord(h->ev_out[c], h->s_out));
    }
    for (int c = 0; c < nch; ++c) {
        const int64_t lo = edge[c], cnt = edge[c + 1] - edge[c];
        HIP_TRY(h, hipEventSynchronize(h->ev_out[c]));
        if (cnt <= 0) continue;
        std::memcpy(out_u0 + lo * NT, pout + p_u0 + lo * NT, (size_t)cnt * NT * 8);
        if (out_U) std::memcpy(out_U + lo * nw, pout + p_U + lo * nw, (size_t)cnt * nw * 8);
        if (warmU) std::memcpy(warmU + lo * nw, pout + p_U + lo * nw, (size_t)cnt * nw * 8);
        if (status) std::memcpy(status + lo, pst + lo, (size_t)cnt * 4);
        if (iters) std::memcpy(iters + lo, pit + lo, (size_t)cnt * 4);
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return FTMPC_OK;
}

int ftmpc_eval_cost_batch(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck, const double* xref,
                          int64_t xref_stride, const double* uref, int64_t uref_stride, const double* U, double* out_cost) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !U || !out_cost) return fail(h, FTMPC_ERR_ARG, "null buffer or negative batch");
    if (B == 0) return FTMPC_OK;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    if (B > h->cap_cost) {
        h->cap_cost = 0;   // (a failed growth must not leave a stale capacity)
        if ((rc = grow(h, &h->d_cost, B)) != FTMPC_OK) return rc;
        h->cap_cost = B;
    }
    const int N = h->cfg.N, NT = h->cfg.NT;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_x0, x0, B * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_ub, ub, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_stuck, stuck, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = stage_refs(h, B, xref, xref_stride, uref, uref_stride)) != FTMPC_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_U, U, B * N * NT * sizeof(double), hipMemcpyHostToDevice, s));
    ftmpc::CostParams cp;
    cp.B = B;
    cp.x0 = h->d_x0; cp.ub = h->d_ub; cp.stuck = h->d_stuck;
    cp.xref = h->d_xref; cp.xref_stride = xref_stride;
    cp.uref = uref ? h->d_uref : nullptr; cp.uref_stride = uref_stride;
    cp.U = h->d_U;
    cp.tcost = h->d_tcost;
    cp.out = h->d_cost;
    hipLaunchKernelGGL(ftmpc::ftmpc_cost_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dc, cp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_cost, h->d_cost, B * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return FTMPC_OK;
}

// The line-search SQP over DEVICE buffers (h->d_x0 / d_ub / d_stuck and the given reference windows), enqueued on h->stream:
// on return S describes where the results are (S.U the final sequences, S.J their cost, J0 the cost of the start point).
#ifndef FTMPC_SQP_LINESEARCH_AT_ONCE
#define FTMPC_SQP_LINESEARCH_AT_ONCE 1
#endif
// the launches of one SQP solve on the handle's stream (launch = false: a replay, only the state the caller reads back is formed)
static int sqp_record(ftmpc_handle* h, int64_t B, const double* d_xref, int64_t xref_stride, const double* d_uref, int64_t uref_stride,
                      const double* d_warm, int32_t sqp_iters, int32_t backtracks, double tol, ftmpc::SqpState& S, double** J0_out, bool launch) {
    const int N = h->cfg.N, NT = h->cfg.NT;
    const int64_t nw = (int64_t)N * NT;
    int rc;
    hipStream_t s = h->stream;
    double *J = h->d_sqJ, *Jt = h->d_sqJ + B, *J0 = h->d_sqJ + 2 * B, *alpha = h->d_sqJ + 3 * B;
    S.B = B; S.N = N; S.NT = NT;
    S.ub = h->d_ub;
    S.U = h->d_sqU; S.Uq = h->d_sqQ; S.Ut = h->d_sqT;
    S.J = J; S.Jt = Jt; S.alpha = alpha;
    S.active = h->d_sqF; S.todo = h->d_sqF + B; S.improved = h->d_sqF + 2 * B; S.nmajor = h->d_sqF + 3 * B; S.ipm = h->d_sqF + 4 * B;
    S.status = h->d_sqF + 5 * B;
    S.qstatus = h->d_status; S.qiters = h->d_iters;
    S.tol = tol;
    if (!launch) {      // the iterate's buffer after the swaps of the major iterations
        if (sqp_iters & 1) std::swap(S.U, S.Ut);
        *J0_out = J0;
        return FTMPC_OK;
    }
    const unsigned gE = (unsigned)((B * nw + 255) / 256), gB = (unsigned)((B + 255) / 256);
    ftmpc::CostParams cp;
    cp.B = B;
    cp.x0 = h->d_x0; cp.ub = h->d_ub; cp.stuck = h->d_stuck;
    cp.xref = d_xref; cp.xref_stride = xref_stride;
    cp.uref = d_uref; cp.uref_stride = uref_stride;
    cp.tcost = h->d_tcost;
    auto cost = [&](const double* U, double* out) {
        cp.U = U;
        cp.out = out;
        hipLaunchKernelGGL(ftmpc::ftmpc_cost_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dc, cp);
    };
    hipLaunchKernelGGL(ftmpc::ftmpc_sqp_init_kernel, dim3(gE > gB ? gE : gB), dim3(256), 0, s, S, d_warm);
    cost(S.U, J);
    HIP_TRY(h, hipMemcpyAsync(J0, J, B * sizeof(double), hipMemcpyDeviceToDevice, s));
    for (int it = 0; it < sqp_iters; ++it) {
        // the QP linearised about the current iterate (every instance: a stopped one costs a solve but changes nothing)
        rc = enqueue(h, B, h->d_x0, h->d_ub, h->d_stuck, d_xref, xref_stride, d_uref, uref_stride, S.U, h->d_u0, h->d_sqQ, h->d_status,
                     h->d_iters, s, -1);
        if (rc != FTMPC_OK) return rc;
        hipLaunchKernelGGL(ftmpc::ftmpc_sqp_open_kernel, dim3(gB), dim3(256), 0, s, S);
        if (FTMPC_SQP_LINESEARCH_AT_ONCE) {
            // every trial point of the line search in one launch, the first acceptable one picked in a second (the alpha = 1,
            // 1/2, ... rounds of trial / cost / decide are 3 x backtracks dependent launches for the same result)
            ftmpc::CostParams ct = cp;
            ct.U = S.U;
            ct.Uq = S.Uq;
            ct.todo = S.todo;
            ct.ntrial = backtracks;
            ct.out = h->d_sqJall;
            hipLaunchKernelGGL(ftmpc::ftmpc_cost_kernel, dim3((unsigned)((B * backtracks + 63) / 64)), dim3(64), 0, s, h->dc, ct);
            S.Jall = h->d_sqJall;
            S.ntrial = backtracks;
            hipLaunchKernelGGL(ftmpc::ftmpc_sqp_pick_kernel, dim3(gB), dim3(256), 0, s, S);
        } else {
            for (int bt = 0; bt < backtracks; ++bt) {
                hipLaunchKernelGGL(ftmpc::ftmpc_sqp_trial_kernel, dim3(gE), dim3(256), 0, s, S);
                cost(S.Ut, Jt);
                hipLaunchKernelGGL(ftmpc::ftmpc_sqp_decide_kernel, dim3(gB), dim3(256), 0, s, S);
            }
        }
        hipLaunchKernelGGL(ftmpc::ftmpc_sqp_close_kernel, dim3(gE), dim3(256), 0, s, S);     // new iterate -> Ut
        hipLaunchKernelGGL(ftmpc::ftmpc_sqp_count_kernel, dim3(gB), dim3(256), 0, s, S);
        std::swap(S.U, S.Ut);
        HIP_TRY(h, hipGetLastError());
    }
    *J0_out = J0;
    return FTMPC_OK;
}

// One SQP solve on the handle's stream.  The sequence is sqp_iters x (linearise, QP kernels, open, backtracks x (trial, cost, decide),
// close, count): ~300 launches of mostly tiny kernels for ten major iterations, launch-bound on small batches.  A call that repeats
// the previous call's shape (batch, buffers, strides, counts, constants; no reallocation in between) is recorded into a hipGraph and
// every further one replays it with ONE launch (small batches; see FTMPC_SQP_GRAPH below).  Profiling or a failed capture leave the
// direct launches.  (The closed loop of ftmpc_simulate_batch_ex moves its reference pointer every step: direct launches.)
static int sqp_enqueue(ftmpc_handle* h, int64_t B, const double* d_xref, int64_t xref_stride, const double* d_uref, int64_t uref_stride,
                       const double* d_warm, int32_t sqp_iters, int32_t backtracks, double tol, ftmpc::SqpState& S, double** J0_out) {
    const int64_t nw = (int64_t)h->cfg.N * h->cfg.NT;
    int rc;
    if (B > h->cap_sqp) {
        h->cap_sqp = 0;
        if ((rc = grow(h, &h->d_sqU, B * nw)) != FTMPC_OK || (rc = grow(h, &h->d_sqQ, B * nw)) != FTMPC_OK ||
            (rc = grow(h, &h->d_sqT, B * nw)) != FTMPC_OK || (rc = grow(h, &h->d_sqJ, 4 * B)) != FTMPC_OK ||
            (rc = grow(h, &h->d_sqF, 6 * B)) != FTMPC_OK)
            return rc;
        h->cap_sqp = B;
    }
    if (B * backtracks > h->cap_sqJall) {
        h->cap_sqJall = 0;
        if ((rc = grow(h, &h->d_sqJall, B * backtracks)) != FTMPC_OK) return rc;
        h->cap_sqJall = B * backtracks;
    }
    // FTMPC_SQP_GRAPH: 0 never, 1 always; unset: batches up to 512 (measured, ten major iterations: 7.0 -> 6.2 ms at B = 256; at
    // B = 1 024 the replay is SLOWER than the direct launches, 10.0 against 7.6 ms -- a captured step always launches the full
    // persistent grids, the direct path shrinks those whose work list was empty the step before)
    static const int graphs_mode = [] {
        const char* e = std::getenv("FTMPC_SQP_GRAPH");
        return !e ? 2 : (e[0] == '0' ? 0 : 1);
    }();
    const bool graphs_on = graphs_mode == 1 || (graphs_mode == 2 && B <= 512);
    hipStream_t s = h->stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool outer_capture = hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
    if (!graphs_on || h->profiling || outer_capture || sqp_iters == 0)
        return sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, true);
    ftmpc_handle::SqpKey key;
    key.B = B; key.xs = xref_stride; key.us = uref_stride;
    key.xref = d_xref; key.uref = d_uref; key.warm = d_warm;
    key.iters = sqp_iters; key.backtracks = backtracks; key.tol = tol;
    key.epoch = h->alloc_epoch;
    {   // the constants the kernels take by value, and the switches that pick kernels: FNV-1a over their bytes
        uint64_t f = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) {
            const unsigned char* b = static_cast<const unsigned char*>(p);
            for (size_t i = 0; i < n; ++i) f = (f ^ b[i]) * 1099511628211ull;
        };
        mix(&h->dc, sizeof(h->dc));
        const void* ptrs[2] = {h->d_tcost, h->rec};
        mix(ptrs, sizeof(ptrs));
        const int sw[8] = {h->use_f64, h->use_wg, h->use_ric64, h->nb_max, h->tset, h->sbounds, h->cfg.kernel_select, (int)h->lin_split_max};
        mix(sw, sizeof(sw));
        key.consts = f;
    }
    if (h->sqp_exec && key == h->sqp_key) {
        if ((rc = sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, false)) != FTMPC_OK)
            return rc;
        HIP_TRY(h, hipGraphLaunch(h->sqp_exec, s));
        ++h->sqp_graph_launches;
        return FTMPC_OK;
    }
    if (h->sqp_exec) {
        (void)hipGraphExecDestroy(h->sqp_exec);
        h->sqp_exec = nullptr;
    }
    if (!(key == h->sqp_seen)) {      // first call of this shape: direct launches (it may still be a one-off)
        h->sqp_seen = key;
        return sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, true);
    }
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, true);
    }
    ftmpc::SqpState Sc;
    double* J0c = nullptr;
    rc = sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, Sc, &J0c, true);
    hipGraph_t g = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &g);
    hipGraphExec_t ex = nullptr;
    if (rc == FTMPC_OK && ec == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex) {
        (void)hipGraphDestroy(g);
        h->sqp_exec = ex;
        h->sqp_key = key;
        if ((rc = sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, false)) != FTMPC_OK)
            return rc;
        HIP_TRY(h, hipGraphLaunch(h->sqp_exec, s));
        ++h->sqp_graph_launches;
        return FTMPC_OK;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    h->sqp_seen = ftmpc_handle::SqpKey();      // (do not try again on every call)
    h->sqp_seen.B = -2;
    return sqp_record(h, B, d_xref, xref_stride, d_uref, uref_stride, d_warm, sqp_iters, backtracks, tol, S, J0_out, true);
}

int ftmpc_solve_sqp_batch(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck, const double* xref,
                          int64_t xref_stride, const double* uref, int64_t uref_stride, const double* warmU, int32_t sqp_iters,
                          int32_t backtracks, double tol, double* out_u0, double* out_U, double* out_cost, double* out_cost0,
                          int32_t* out_sqp_iters, int32_t* out_iters, int32_t* status) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !out_u0 || sqp_iters < 0 || backtracks < 1 || !(tol >= 0))
        return fail(h, FTMPC_ERR_ARG, "null buffer, negative batch or bad iteration counts");
    if (B == 0) return FTMPC_OK;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    const int N = h->cfg.N, NT = h->cfg.NT;
    const int64_t nw = (int64_t)N * NT;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_x0, x0, B * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_ub, ub, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_stuck, stuck, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = stage_refs(h, B, xref, xref_stride, uref, uref_stride)) != FTMPC_OK) return rc;
    if (warmU) HIP_TRY(h, hipMemcpyAsync(h->d_warm, warmU, B * nw * sizeof(double), hipMemcpyHostToDevice, s));
    ftmpc::SqpState S;
    double* J0 = nullptr;
    if ((rc = sqp_enqueue(h, B, h->d_xref, xref_stride, uref ? h->d_uref : nullptr, uref_stride, warmU ? h->d_warm : nullptr, sqp_iters,
                          backtracks, tol, S, &J0)) != FTMPC_OK)
        return rc;
    // u0 = stage 0 of the final sequences
    HIP_TRY(h, hipMemcpy2DAsync(out_u0, NT * sizeof(double), S.U, nw * sizeof(double), NT * sizeof(double), (size_t)B, hipMemcpyDeviceToHost, s));
    if (out_U) HIP_TRY(h, hipMemcpyAsync(out_U, S.U, B * nw * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_cost) HIP_TRY(h, hipMemcpyAsync(out_cost, S.J, B * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_cost0) HIP_TRY(h, hipMemcpyAsync(out_cost0, J0, B * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_sqp_iters) HIP_TRY(h, hipMemcpyAsync(out_sqp_iters, S.nmajor, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (out_iters) HIP_TRY(h, hipMemcpyAsync(out_iters, S.ipm, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, S.status, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return FTMPC_OK;
}

int64_t ftmpc_sqp_graph_launches(const ftmpc_handle* h) { return h ? h->sqp_graph_launches : -1; }

int ftmpc_solve_batch_device(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck,
                             const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride,
                             const double* warmU, double* out_u0, double* out_U, int32_t* status, int32_t* iters,
                             void* stream) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !out_u0) return fail(h, FTMPC_ERR_ARG, "null buffer or negative batch");
    if (B == 0) return FTMPC_OK;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (B > h->cap_batch) {
        h->cap_batch = 0;   // (a failed growth must not leave a stale capacity)
        // growing the workspace allocates: callers that time or graph-capture must ftmpc_reserve first
        if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    }
    return enqueue(h, B, x0, ub, stuck, xref, xref_stride, uref, uref_stride, warmU, out_u0, out_U, status, iters,
                   reinterpret_cast<hipStream_t>(stream), -1);
}

// The generalized-force formulation runs on kernel 11 (fp32, one wave per instance; hull rows, and the terminal set when the
// handle has one) when the handle computes in fp32 and the problem fits six tiles a side (N <= 16) and eight row slots per
// lane; the instances kernel 11 does not certify, and everything else (dtype FTMPC_DTYPE_F64, kernel_select =
// FTMPC_KERNEL_DENSE, longer horizons), on the float64 kernel.
static bool hull_fp32(const ftmpc_handle* h, int32_t hull_rows) {
    return h->cfg.dtype != FTMPC_DTYPE_F64 && h->cfg.kernel_select != FTMPC_KERNEL_DENSE && 6 * h->cfg.N <= 96 && hull_rows <= 32 &&
           (int64_t)h->cfg.N * 32 <= 64 * ftmpc::hullk::nvc_of(6) && (!h->cfg.terminal_set || h->cfg.term_rows <= 80);
}

// ... and on kernel 13 (float64, Riccati recursion, one wave per instance: any horizon up to 40, up to 128 hull rows, with or
// without the terminal set) unless kernel_select is FTMPC_KERNEL_DENSE: the whole batch where kernel 11 does not apply (float64
// handles, N > 16, more than 32 rows), and otherwise the instances kernel 11 hands over.  The dense float64 kernel keeps
// kernel_select = FTMPC_KERNEL_DENSE and N > 40.
static bool hull_ricw(const ftmpc_handle* h, int32_t hull_rows) {
    return h->cfg.kernel_select != FTMPC_KERNEL_DENSE && h->cfg.N <= 40 && hull_rows <= ftmpc::rickw::MHMAX &&
           (!h->cfg.terminal_set || h->cfg.term_rows <= 80);
}

// Validation, workspace and hull tables of the generalized-force formulation (shared by the one-step entry and the closed loop).
static int wrench_prepare(ftmpc_handle* h, int64_t B, const double* hull_A, int32_t n_sets, const int32_t* hull_set, const double* hull_b,
                          int32_t hull_rows) {
    const int N = h->cfg.N;
    if (6 * N > 256) return fail(h, FTMPC_ERR_ARG, "the generalized-force formulation needs 6 N <= 256");
    if (hull_rows < 1 || hull_rows > FTMPC_MAX_HULL_ROWS || n_sets < 1) return fail(h, FTMPC_ERR_ARG, "hull_rows out of range (1..128) or no hull table");
    if (!hull_ricw(h, hull_rows) && (hull_rows > 32 || (int64_t)N * hull_rows > 1024))
        return fail(h, FTMPC_ERR_ARG, "with kernel_select = FTMPC_KERNEL_DENSE or N > 40 the generalized-force formulation needs hull_rows <= 32 and N * hull_rows <= 1024");
    if (h->cfg.terminal_set && (h->cfg.term_rows < 1 || h->cfg.term_rows > FTMPC_MAX_TERM_ROWS)) return fail(h, FTMPC_ERR_ARG, "term_rows out of range");
    if (hull_set)   // the kernel indexes hull_A by these: a table number outside [0, n_sets) would be an out-of-bounds device read
        for (int64_t b = 0; b < B; ++b)
            if (hull_set[b] < 0 || hull_set[b] >= n_sets)
                return fail(h, FTMPC_ERR_ARG, "hull_set[" + std::to_string(b) + "] = " + std::to_string(hull_set[b]) + " is not a table number in [0, n_sets)");
    int rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    if (hull_fp32(h, hull_rows)) {     // kernel 11: one wave per instance, H_w tiles in LDS; only the float64 gradient scratch is global
        if (!h->hull_slot) {
            int per = 0;
            if (h->cfg.terminal_set) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_hull32_kernel<6, true>, 64, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_hull32_kernel<6, false>, 64, 0);
            h->grid_hull = h->num_cu * (per > 0 ? per : 1);
            h->hull_slot_words = ftmpc::wswk::slot_words(6, N);
            if ((rc = grow(h, &h->hull_slot, (int64_t)h->grid_hull * h->hull_slot_words)) != FTMPC_OK) return rc;
        }
    }
    if (hull_ricw(h, hull_rows)) {     // kernel 13's per-wave slots (sized by the row count)
        const int64_t need = ftmpc::rickw::slot_doubles(N, hull_rows);
        if (!h->ricw_slot || need > h->ricw_slot_doubles) {
            int per = 0;
            if (h->cfg.terminal_set && N <= 24) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ricw64_kernel<6, true>, 64, 0);
            else if (h->cfg.terminal_set) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ricw64_kernel<10, true>, 64, 0);
            else if (N <= 24) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ricw64_kernel<6>, 64, 0);
            else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, ftmpc::ftmpc_solve_ricw64_kernel<10>, 64, 0);
            h->grid_ricw = h->num_cu * std::max(1, per);
            h->ricw_slot_doubles = 0;
            if ((rc = grow(h, &h->ricw_slot, (int64_t)h->grid_ricw * need)) != FTMPC_OK) return rc;
            h->ricw_slot_doubles = need;
        }
    }
    // the dense float64 kernel's slots: the terminal-set forms, kernel_select = FTMPC_KERNEL_DENSE, N > 40 -- the whole batch, or what
    // kernel 11 hands over (hull and terminal rows active together, a polish that did not settle: SolveHullParams::fb_list)
    if (h->cfg.terminal_set && !h->d_term) {      // the rows of the terminal set on the device: term_A | term_b
        if ((rc = grow(h, &h->d_term, (int64_t)h->cfg.term_rows * 10)) != FTMPC_OK) return rc;
        std::vector<double> t((size_t)h->cfg.term_rows * 10);
        std::memcpy(t.data(), h->cfg.term_A, (size_t)h->cfg.term_rows * 9 * sizeof(double));
        std::memcpy(t.data() + (size_t)h->cfg.term_rows * 9, h->cfg.term_b, (size_t)h->cfg.term_rows * sizeof(double));
        HIP_TRY(h, hipMemcpy(h->d_term, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (!hull_ricw(h, hull_rows) && !h->gHs) {   // per-workgroup slots of the 6N-variable problem (separate from the thruster-space slots of this handle)
        const int nbg = (6 * N + 15) / 16;
        h->npad_gen = 16 * nbg;
        h->grid_gen = h->num_cu;
        h->tile_doubles_gen = (int64_t)tiles_of(nbg) * 256;
        h->e_doubles_gen = (int64_t)(N + 2) * 9 * h->npad_gen;
        if ((rc = grow(h, &h->gHs, h->grid_gen * h->tile_doubles_gen)) != FTMPC_OK || (rc = grow(h, &h->gLs, h->grid_gen * h->tile_doubles_gen)) != FTMPC_OK ||
            (rc = grow(h, &h->gEall, h->grid_gen * h->e_doubles_gen)) != FTMPC_OK)
            return rc;
        if (h->cfg.terminal_set && !h->d_term) {
            if ((rc = grow(h, &h->d_term, (int64_t)h->cfg.term_rows * 10)) != FTMPC_OK) return rc;
            std::vector<double> t((size_t)h->cfg.term_rows * 10);
            std::memcpy(t.data(), h->cfg.term_A, (size_t)h->cfg.term_rows * 9 * sizeof(double));
            std::memcpy(t.data() + (size_t)h->cfg.term_rows * 9, h->cfg.term_b, (size_t)h->cfg.term_rows * sizeof(double));
            HIP_TRY(h, hipMemcpy(h->d_term, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    const int64_t nA = (int64_t)n_sets * hull_rows * 6;
    if (nA > h->cap_hullA) {
        h->cap_hullA = 0;   // (a failed growth must not leave a stale capacity)
        if ((rc = grow(h, &h->d_hullA, nA)) != FTMPC_OK) return rc;
        h->cap_hullA = nA;
    }
    if (B > h->cap_wrench) {
        h->cap_wrench = 0;   // (a failed growth must not leave a stale capacity)
        if ((rc = grow(h, &h->d_hullb, B * FTMPC_MAX_HULL_ROWS)) != FTMPC_OK || (rc = grow(h, &h->d_hullset, B)) != FTMPC_OK ||
            (rc = grow(h, &h->d_warmG, B * N * 6)) != FTMPC_OK || (rc = grow(h, &h->d_tau0, B * 6)) != FTMPC_OK ||
            (rc = grow(h, &h->d_G, B * N * 6)) != FTMPC_OK || (rc = grow(h, &h->d_taud, B * 6)) != FTMPC_OK || (rc = grow(h, &h->d_ast2, 3 * B)) != FTMPC_OK)      // allocation status | iterations | kernel 11's hand-over list
            return rc;
        h->cap_wrench = B;
    }
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_hullA, hull_A, nA * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_hullb, hull_b, B * hull_rows * sizeof(double), hipMemcpyHostToDevice, s));
    if (hull_set) HIP_TRY(h, hipMemcpyAsync(h->d_hullset, hull_set, B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    return FTMPC_OK;
}

// One two-stage step over DEVICE buffers (h->d_x0 / d_ub / d_stuck, the staged hull tables, the given reference windows):
// linearise, the 6N-variable QP with the hull rows, allocation.  Leaves u0 in h->d_u0, tau_0 in h->d_tau0, the wrenches in h->d_G.
static int wrench_enqueue(ftmpc_handle* h, int64_t B, int32_t hull_rows, bool has_set, const double* d_xref, int64_t xref_stride,
                          const double* d_uref, int64_t uref_stride, const double* d_warmG) {
    hipStream_t s = h->stream;
    LinParams lp;
    lp.B = B;
    lp.x0 = h->d_x0; lp.ub = h->d_ub; lp.stuck = h->d_stuck;
    lp.xref = d_xref; lp.xref_stride = xref_stride;
    lp.uref = d_uref; lp.uref_stride = uref_stride;
    lp.warmU = nullptr;
    lp.rec = h->rec;
    lp.qlist = nullptr; lp.qcount = nullptr; lp.qvmax = -1;
    lp.warmG = d_warmG;
    lp.out_eN = h->d_eN;
    lp.tcost = h->d_tcost;
    lp.out_cbar = nullptr;
    launch_linearize(h, B, (int)((B + 63) / 64), s, lp);
    HIP_TRY(h, hipGetLastError());
    // the wrench problem stops at mu 1e-10 unless the caller asked otherwise (general rows: see ftmpc_config.mu_stop)
    DeviceConsts dcg = h->dc;
    if (!(h->cfg.mu_stop > 0)) dcg.mu_stop = 1e-10;
    if (hull_fp32(h, hull_rows)) {
        HIP_TRY(h, hipMemsetAsync(h->d_qctl, 0, 8 * sizeof(int32_t), s));
        ftmpc::SolveHullParams q;
        std::memset(&q, 0, sizeof(q));
        q.base.B = B;
        q.base.rec = h->rec;
        q.base.ub = h->d_ub; q.base.stuck = h->d_stuck;
        q.base.out_u0 = h->d_u0;
        q.base.status = h->d_status; q.base.iters = h->d_iters;
        q.base.hscratch = h->hull_slot;
        q.base.tile_words = h->hull_slot_words;
        q.base.qhead = h->d_qctl + 7;
        q.base.dbg_inst = -1;
        q.base.dbg_H = h->use_f64 ? reinterpret_cast<float*>(h->d_dbgH64) : h->d_dbgH;     // (diagnostic build: phase stamps)
        q.warmG = d_warmG;
        q.hullA = h->d_hullA;
        q.hull_set = has_set ? h->d_hullset : nullptr;
        q.hullb = h->d_hullb;
        q.hull_rows = hull_rows;
        q.out_tau0 = h->d_tau0;
        q.out_G = h->d_G;
        q.termA = h->cfg.terminal_set ? h->d_term : nullptr;
        q.termb = h->cfg.terminal_set ? h->d_term + (int64_t)h->cfg.term_rows * 9 : nullptr;
        q.term_rows = h->cfg.terminal_set ? h->cfg.term_rows : 0;
        q.eN = h->d_eN;
        const int grid = (int)std::min<int64_t>(B, h->grid_hull);
        q.fb_list = h->d_ast2 + 2 * h->cap_wrench;
        q.fb_count = h->d_qctl;      // (the generalized-force path builds no work lists: the counter of list 0 is free)
        // kernel 11 leaves the interior-point iteration at mu 1e-7 for its polish: below that the fp32 slacks and duals of the
        // active rows are noise that spoils the active set they are read for (measured on 16 384 instances: 253 polishes do not
        // settle from mu 1e-10, 66 from 1e-7, same 1.9e-6 f_max worst error; from 1e-6 a wrong set is "verified")
        DeviceConsts dch = dcg;
        if (!(h->cfg.mu_stop > 0)) dch.mu_stop = 1e-7;
        if (h->cfg.terminal_set) hipLaunchKernelGGL((ftmpc::ftmpc_solve_hull32_kernel<6, true>), dim3(grid), dim3(64), 0, s, dch, q);
        else hipLaunchKernelGGL((ftmpc::ftmpc_solve_hull32_kernel<6, false>), dim3(grid), dim3(64), 0, s, dch, q);
        HIP_TRY(h, hipGetLastError());
    }
    const bool handed = hull_fp32(h, hull_rows);
    h->wrench_handed = handed;
    // allocation of the batch on a second stream, beside kernel 13's pass over the hand-over list (not while profiling: the
    // event pairs of ftmpc_last_kernel_ms sit on one stream)
    const bool overlap = handed && hull_ricw(h, hull_rows) && !h->profiling && h->stream2 != nullptr;
    if (overlap) {
        HIP_TRY(h, hipEventRecord(h->ev_fork, s));
        HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        hipLaunchKernelGGL(ftmpc::ftmpc_healthy_wrench_kernel, dim3((unsigned)((B * 6 + 255) / 256)), dim3(256), 0, h->stream2, h->dc, B,
                           (const double*)h->d_tau0, (const double*)h->d_stuck, h->d_taud, (const int32_t*)nullptr, (const int32_t*)nullptr);
        ftmpc::AllocParams a0;
        a0.B = B;
        a0.tau = h->d_taud;
        a0.ub = h->d_ub;
        a0.out_u = h->d_u0;
        a0.status = h->d_ast2;
        a0.iters = h->d_ast2 + B;
        a0.max_iters = 50;
        a0.tol = 1e-8;
        hipLaunchKernelGGL(ftmpc::ftmpc_allocate_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream2, h->dc, a0);
        HIP_TRY(h, hipEventRecord(h->ev_alloc, h->stream2));
    }
    if (hull_ricw(h, hull_rows)) {      // kernel 13: the whole batch, or what kernel 11 handed over
        ftmpc::SolveRicwParams q;
        std::memset(&q, 0, sizeof(q));
        q.base.B = B;
        q.base.rec = h->rec;
        q.base.ub = h->d_ub; q.base.stuck = h->d_stuck;
        q.base.status = h->d_status; q.base.iters = h->d_iters;
        q.base.dbg_inst = -1;
        q.base.qlist = handed ? h->d_ast2 + 2 * h->cap_wrench : nullptr;
        q.base.qcount = handed ? h->d_qctl : nullptr;
        HIP_TRY(h, hipMemsetAsync(h->d_qctl + 4, 0, sizeof(int32_t), s));
        q.base.qhead = h->d_qctl + 4;
        q.slot = h->ricw_slot;
        q.slot_doubles = h->ricw_slot_doubles;
        q.warmG = d_warmG;
        q.hullA = h->d_hullA;
        q.hull_set = has_set ? h->d_hullset : nullptr;
        q.hullb = h->d_hullb;
        q.hull_rows = hull_rows;
        q.out_tau0 = h->d_tau0;
        q.out_G = h->d_G;
        q.termA = h->cfg.terminal_set ? h->d_term : nullptr;
        q.termb = h->cfg.terminal_set ? h->d_term + (int64_t)h->cfg.term_rows * 9 : nullptr;
        q.term_rows = h->cfg.terminal_set ? h->cfg.term_rows : 0;
        q.eN = h->d_eN;
        const int grid = (int)std::min<int64_t>(B, h->grid_ricw);
        if (h->cfg.terminal_set && h->cfg.N <= 24) hipLaunchKernelGGL((ftmpc::ftmpc_solve_ricw64_kernel<6, true>), dim3(grid), dim3(64), 0, s, dcg, q);
        else if (h->cfg.terminal_set) hipLaunchKernelGGL((ftmpc::ftmpc_solve_ricw64_kernel<10, true>), dim3(grid), dim3(64), 0, s, dcg, q);
        else if (h->cfg.N <= 24) hipLaunchKernelGGL(ftmpc::ftmpc_solve_ricw64_kernel<6>, dim3(grid), dim3(64), 0, s, dcg, q);
        else hipLaunchKernelGGL(ftmpc::ftmpc_solve_ricw64_kernel<10>, dim3(grid), dim3(64), 0, s, dcg, q);
        HIP_TRY(h, hipGetLastError());
    } else {
    // the dense float64 kernel: the whole batch, or what kernel 11 handed over
    Solve64Params q;
    std::memset(&q, 0, sizeof(q));
    q.base.B = B;
    q.base.rec = h->rec;
    q.base.ub = h->d_ub; q.base.stuck = h->d_stuck;
    q.base.out_u0 = h->d_u0;
    q.base.status = h->d_status; q.base.iters = h->d_iters;
    q.base.dbg_inst = -1;
    q.base.qlist = handed ? h->d_ast2 + 2 * h->cap_wrench : nullptr;
    q.base.qcount = handed ? h->d_qctl : nullptr;
    q.Hs = h->gHs; q.Ls = h->gLs; q.Eall = h->gEall;
    q.tile_doubles = h->tile_doubles_gen;
    q.e_doubles = h->e_doubles_gen;
    q.npad_max = h->npad_gen;
    q.warmG = d_warmG;
    q.hullA = h->d_hullA;
    q.hull_set = has_set ? h->d_hullset : nullptr;
    q.hullb = h->d_hullb;
    q.hull_rows = hull_rows;
    q.termA = h->cfg.terminal_set ? h->d_term : nullptr;
    q.termb = h->cfg.terminal_set ? h->d_term + (int64_t)h->cfg.term_rows * 9 : nullptr;
    q.term_rows = h->cfg.terminal_set ? h->cfg.term_rows : 0;
    q.eN = h->d_eN;
    q.out_tau0 = h->d_tau0;
    q.out_G = h->d_G;
    const int grid = (int)std::min<int64_t>(B, h->grid_gen);
    if (h->cfg.terminal_set)
        hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<4, 1, 3>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, dcg, q);
    else
        hipLaunchKernelGGL((ftmpc::ftmpc_solve_f64_kernel<4, 1, 1>), dim3(grid), dim3(ftmpc::f64k::WG), 0, s, dcg, q);
    HIP_TRY(h, hipGetLastError());
    }
    // second stage: min-norm allocation of the wrench the healthy thrusters have to produce
    ftmpc::AllocParams ap;
    ap.B = B;
    ap.tau = h->d_taud;
    ap.ub = h->d_ub;
    ap.out_u = h->d_u0;
    ap.status = h->d_ast2;
    ap.iters = h->d_ast2 + B;
    ap.max_iters = 50;
    ap.tol = 1e-8;
    auto allocate = [&](hipStream_t st, const int32_t* list, const int32_t* count) {
        hipLaunchKernelGGL(ftmpc::ftmpc_healthy_wrench_kernel, dim3((unsigned)((B * 6 + 255) / 256)), dim3(256), 0, st, h->dc, B,
                           (const double*)h->d_tau0, (const double*)h->d_stuck, h->d_taud, list, count);
        ftmpc::AllocParams a = ap;
        a.list = list;
        a.count = count;
        hipLaunchKernelGGL(ftmpc::ftmpc_allocate_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, h->dc, a);
    };
    if (overlap) {
        // the handed-over instances (a few dozen of a regular batch, one wave each: ~2 ms of latency, the device nearly idle) run
        // on kernel 13 while the second stream allocates everything kernel 11 certified; their own allocation follows in list mode
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_alloc, 0));
        allocate(s, h->d_ast2 + 2 * h->cap_wrench, h->d_qctl);
    } else {
        allocate(s, nullptr, nullptr);
    }
    HIP_TRY(h, hipGetLastError());
    return FTMPC_OK;
}

int ftmpc_solve_wrench_batch(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck,
                             const double* hull_A, int32_t n_sets, const int32_t* hull_set, const double* hull_b, int32_t hull_rows,
                             const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride, double* warmG,
                             double* out_u0, double* out_tau0, double* out_G, int32_t* status, int32_t* iters, int32_t* alloc_status) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !x0 || !ub || !stuck || !xref || !out_u0 || !hull_A || !hull_b)
        return fail(h, FTMPC_ERR_ARG, "null buffer or negative batch");
    if (B == 0) return FTMPC_OK;
    const int N = h->cfg.N, NT = h->cfg.NT;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    if ((rc = wrench_prepare(h, B, hull_A, n_sets, hull_set, hull_b, hull_rows)) != FTMPC_OK) return rc;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_x0, x0, B * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_ub, ub, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_stuck, stuck, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = stage_refs(h, B, xref, xref_stride, uref, uref_stride)) != FTMPC_OK) return rc;
    if (warmG) HIP_TRY(h, hipMemcpyAsync(h->d_warmG, warmG, B * N * 6 * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = wrench_enqueue(h, B, hull_rows, hull_set != nullptr, h->d_xref, xref_stride, uref ? h->d_uref : nullptr, uref_stride,
                             warmG ? h->d_warmG : nullptr)) != FTMPC_OK)
        return rc;
    HIP_TRY(h, hipMemcpyAsync(out_u0, h->d_u0, B * NT * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_tau0) HIP_TRY(h, hipMemcpyAsync(out_tau0, h->d_tau0, B * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (out_G) HIP_TRY(h, hipMemcpyAsync(out_G, h->d_G, B * N * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (warmG) HIP_TRY(h, hipMemcpyAsync(warmG, h->d_G, B * N * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, h->d_status, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (iters) HIP_TRY(h, hipMemcpyAsync(iters, h->d_iters, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (alloc_status) HIP_TRY(h, hipMemcpyAsync(alloc_status, h->d_ast2, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return FTMPC_OK;
}

int ftmpc_last_handed_over(ftmpc_handle* h, int64_t* count) {
    if (!h || !count) return FTMPC_ERR_ARG;
    *count = 0;
    if (!h->wrench_handed) return FTMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int32_t c = 0;
    HIP_TRY(h, hipMemcpyAsync(&c, h->d_qctl, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    *count = c;
    return FTMPC_OK;
}

int ftmpc_allocate_batch(ftmpc_handle* h, int64_t B, const double* tau, const double* ub, double* out_u, int32_t* status,
                         int32_t* iters) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || !tau || !ub || !out_u) return fail(h, FTMPC_ERR_ARG, "null buffer or negative batch");
    if (B == 0) return FTMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int NT = h->cfg.NT;
    if (B > h->cap_alloc) {
        h->cap_alloc = 0;   // (a failed growth must not leave a stale capacity)
        int rc;
        if ((rc = grow(h, &h->d_atau, B * 6)) != FTMPC_OK || (rc = grow(h, &h->d_aub, B * NT)) != FTMPC_OK ||
            (rc = grow(h, &h->d_au, B * NT)) != FTMPC_OK || (rc = grow(h, &h->d_ast, B)) != FTMPC_OK ||
            (rc = grow(h, &h->d_ait, B)) != FTMPC_OK)
            return rc;
        h->cap_alloc = B;
    }
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_atau, tau, B * 6 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_aub, ub, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    ftmpc::AllocParams ap;
    ap.B = B;
    ap.tau = h->d_atau;
    ap.ub = h->d_aub;
    ap.out_u = h->d_au;
    ap.status = h->d_ast;
    ap.iters = h->d_ait;
    ap.max_iters = 50;
    ap.tol = 1e-8;
    hipLaunchKernelGGL(ftmpc::ftmpc_allocate_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dc, ap);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_u, h->d_au, B * NT * sizeof(double), hipMemcpyDeviceToHost, s));
    if (status) HIP_TRY(h, hipMemcpyAsync(status, h->d_ast, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (iters) HIP_TRY(h, hipMemcpyAsync(iters, h->d_ait, B * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return FTMPC_OK;
}

int ftmpc_shift_warm(int64_t B, int32_t N, int32_t NT, double* warmU) {
    if (B < 0 || N < 1 || NT < 1 || !warmU) return FTMPC_ERR_ARG;
    for (int64_t b = 0; b < B; ++b) {
        double* w = warmU + b * (int64_t)N * NT;
        std::memmove(w, w + NT, (size_t)(N - 1) * NT * sizeof(double));
        std::memset(w + (size_t)(N - 1) * NT, 0, NT * sizeof(double));
    }
    return FTMPC_OK;
}

int ftmpc_set_profiling(ftmpc_handle* h, int32_t enabled) {
    if (!h) return FTMPC_ERR_ARG;
    h->profiling = enabled != 0;
    h->ev_valid = false;
    return FTMPC_OK;
}

int ftmpc_last_kernel_ms(ftmpc_handle* h, float* ms, int32_t n_slots) {
    if (!h || !ms || n_slots < 0) return FTMPC_ERR_ARG;
    if (!h->ev_valid) return fail(h, FTMPC_ERR_ARG, "no profiled solve recorded");
    for (int k = 0; k < std::min<int>(n_slots, FTMPC_KERNEL_SLOTS); ++k) {
        ms[k] = 0.f;
        if (!h->ev_used[k]) continue;
        HIP_TRY(h, hipEventSynchronize(h->ev[2 * k + 1]));
        HIP_TRY(h, hipEventElapsedTime(&ms[k], h->ev[2 * k], h->ev[2 * k + 1]));
    }
    return FTMPC_OK;
}

static const char* const k_kernel_names[FTMPC_KERNEL_SLOTS] = {"ftmpc_linearize_kernel", "ftmpc_solve_f32_kernel<8>", "ftmpc_solve_f32_kernel<9>",
                                                                "ftmpc_solve_f32_kernel<10>", "ftmpc_solve_f64_kernel", "ftmpc_solve_wsw32_kernel | ftmpc_solve_ws32_kernel | ftmpc_solve_wg32_kernel<15>",
                                                                "ftmpc_solve_ws64_kernel"};

const char* ftmpc_kernel_name(int32_t slot) { return (slot >= 0 && slot < FTMPC_KERNEL_SLOTS) ? k_kernel_names[slot] : ""; }

const char* ftmpc_routed_kernel_name(const ftmpc_handle* h, int32_t slot) {
    if (h && slot == 6) return h->use_ric64 ? "ftmpc_solve_ric64_kernel" : "ftmpc_solve_ws64_kernel";
    if (h && slot == 5) return h->use_wsw ? "ftmpc_solve_wsw32_kernel" : (h->use_ws ? "ftmpc_solve_ws32_kernel" : "ftmpc_solve_wg32_kernel<15>");
    return ftmpc_kernel_name(slot);
}

int ftmpc_debug_build_qp(ftmpc_handle* h, int64_t B, const double* x0, const double* ub, const double* stuck,
                         const double* xref, int64_t xref_stride, const double* uref, int64_t uref_stride,
                         const double* warmU, int64_t inst, double* H, int64_t H_cap, double* g, double* lo,
                         double* hi, int32_t* n_out) {
    if (!h || !H || !g || !lo || !hi || !n_out || inst < 0 || inst >= B) return FTMPC_ERR_ARG;
    int rc = check_strides(h, xref_stride, uref_stride, uref);
    if (rc != FTMPC_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    if ((rc = ftmpc_reserve(h, B)) != FTMPC_OK) return rc;
    const int N = h->cfg.N, NT = h->cfg.NT;
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(h->d_x0, x0, B * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_ub, ub, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->d_stuck, stuck, B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = stage_refs(h, B, xref, xref_stride, uref, uref_stride)) != FTMPC_OK) return rc;
    if (warmU) HIP_TRY(h, hipMemcpyAsync(h->d_warm, warmU, B * N * NT * sizeof(double), hipMemcpyHostToDevice, s));
    if (!h->use_f64) HIP_TRY(h, hipMemsetAsync(h->d_dbgv, 0, (3 * 256 + 4) * sizeof(float), s));
    if (h->use_ws && h->nb_max > 15)
        return fail(h, FTMPC_ERR_ARG, "the QP dump needs N * NT <= 240 on the fp32 path (create the handle with dtype FTMPC_DTYPE_F64 for larger shapes)");
    const bool ws64_was = h->use_ws64, ric_was = h->use_ric64;      // (the dump comes from the dense float64 kernel: the only one that forms H)
    h->use_ws64 = false;
    h->use_ric64 = false;
    const bool wsw_was = h->use_wsw;
    h->use_wsw = false;
    const bool ws_was = h->use_ws;
    h->use_ws = false;     // the dump hook (the condensed thruster-space QP) lives in kernel 7; kernel 8 never forms that matrix
    rc = enqueue(h, B, h->d_x0, h->d_ub, h->d_stuck, h->d_xref, xref_stride, uref ? h->d_uref : nullptr, uref_stride,
                 warmU ? h->d_warm : nullptr, h->d_u0, nullptr, h->d_status, h->d_iters, s, inst);
    h->use_ws = ws_was;
    h->use_wsw = wsw_was;
    h->use_ws64 = ws64_was;
    h->use_ric64 = ric_was;
    if (rc != FTMPC_OK) return rc;
    if (h->use_f64) {
        const int64_t pm = h->npad_max;
        std::vector<double> Hd((size_t)(pm * pm)), vd((size_t)(3 * pm + 4));
        HIP_TRY(h, hipMemcpyAsync(Hd.data(), h->d_dbgH64, Hd.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(vd.data(), h->d_dbgv64, vd.size() * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        const int n = (int)vd[3 * pm], npad = (int)vd[3 * pm + 1];
        if (n <= 0) return fail(h, FTMPC_ERR_ARG, "instance has no active thruster or was not dumped");
        if ((int64_t)n * n > H_cap) return fail(h, FTMPC_ERR_ARG, "H buffer too small");
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) H[(int64_t)i * n + j] = Hd[(size_t)i * npad + j];
            g[i] = vd[i];
            lo[i] = vd[pm + i];
            hi[i] = vd[2 * pm + i];
        }
        *n_out = n;
        return FTMPC_OK;
    }
    std::vector<float> Hf(256 * 256), vf(3 * 256 + 4);
    HIP_TRY(h, hipMemcpyAsync(Hf.data(), h->d_dbgH, Hf.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(vf.data(), h->d_dbgv, vf.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    // the one-wave kernels leave (n, npad) at words 480 / 481, the workgroup kernel at 720 / 721
    const int mk = (vf[720] > 0.f) ? 720 : 480;
    const int n = (int)vf[mk], npad = (int)vf[mk + 1];
    if (n == 0) return fail(h, FTMPC_ERR_ARG, "instance has no active thruster or was not dumped");
    if ((int64_t)n * n > H_cap) return fail(h, FTMPC_ERR_ARG, "H buffer too small");
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) H[(int64_t)i * n + j] = Hf[(size_t)i * npad + j];
        g[i] = vf[i];
        lo[i] = vf[npad + i];
        hi[i] = vf[2 * npad + i];
    }
    *n_out = n;
    return FTMPC_OK;
}

int ftmpc_simulate_batch(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                         const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                         double* u_hist, int32_t* not_converged) {
    return ftmpc_simulate_batch_ex(h, B, T, x, ub, stuck, xref_traj, uref_traj, noise, seed, 0, 0, 0.0, u_hist, not_converged);
}

struct WrenchLoop {      // the two-stage structure inside the closed loop: hull tables staged once, constant over the run
    int32_t hull_rows;
    bool has_set;
    int32_t* alloc_failed;   // host [T] or null
};

static int simulate_core(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck, const double* xref_traj,
                         const double* uref_traj, const double noise[4], uint64_t seed, int32_t sqp_iters, int32_t backtracks, double tol,
                         const WrenchLoop* wl, double* u_hist, int32_t* not_converged);

int ftmpc_simulate_batch_ex(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                            const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                            int32_t sqp_iters, int32_t backtracks, double tol, double* u_hist, int32_t* not_converged) {
    if (!h) return FTMPC_ERR_ARG;
    if (sqp_iters < 0 || (sqp_iters > 0 && (backtracks < 1 || !(tol >= 0)))) return fail(h, FTMPC_ERR_ARG, "bad SQP iteration counts");
    if (B < 0 || T < 0 || !x || !ub || !stuck || !xref_traj || !noise) return fail(h, FTMPC_ERR_ARG, "null buffer or negative size");
    if (B == 0 || T == 0) return FTMPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ftmpc_reserve(h, B);
    if (rc != FTMPC_OK) return rc;
    return simulate_core(h, B, T, x, ub, stuck, xref_traj, uref_traj, noise, seed, sqp_iters, backtracks, tol, nullptr, u_hist, not_converged);
}

int ftmpc_simulate_wrench_batch(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                                const double* hull_A, int32_t n_sets, const int32_t* hull_set, const double* hull_b, int32_t hull_rows,
                                const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                                double* u_hist, int32_t* not_converged, int32_t* alloc_failed) {
    if (!h) return FTMPC_ERR_ARG;
    if (B < 0 || T < 0 || !x || !ub || !stuck || !xref_traj || !noise || !hull_A || !hull_b) return fail(h, FTMPC_ERR_ARG, "null buffer or negative size");
    if (B == 0 || T == 0) return FTMPC_OK;
    int rc = wrench_prepare(h, B, hull_A, n_sets, hull_set, hull_b, hull_rows);
    if (rc != FTMPC_OK) return rc;
    WrenchLoop wl{hull_rows, hull_set != nullptr, alloc_failed};
    return simulate_core(h, B, T, x, ub, stuck, xref_traj, uref_traj, noise, seed, 0, 0, 0.0, &wl, u_hist, not_converged);
}

static int simulate_core(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck, const double* xref_traj,
                         const double* uref_traj, const double noise[4], uint64_t seed, int32_t sqp_iters, int32_t backtracks, double tol,
                         const WrenchLoop* wl, double* u_hist, int32_t* not_converged) {
    int rc;
    const int N = h->cfg.N, NT = h->cfg.NT;
    hipStream_t s = h->stream;
    const int64_t ncol = (int64_t)T + N;   // windows t .. t+N for t < T
    double *d_xr = nullptr, *d_ur = nullptr, *d_warmB = nullptr, *d_hist = nullptr;
    int32_t *d_bad = nullptr, *d_abad = nullptr;
    auto cleanup = [&]() {
        if (d_abad) (void)hipFree(d_abad);
        if (d_xr) (void)hipFree(d_xr);
        if (d_ur) (void)hipFree(d_ur);
        if (d_warmB) (void)hipFree(d_warmB);
        if (d_hist) (void)hipFree(d_hist);
        if (d_bad) (void)hipFree(d_bad);
    };
#define SIM_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            cleanup();                                                                     \
            return fail(h, FTMPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
        }                                                                                  \
    } while (0)
    SIM_TRY(hipMalloc(&d_xr, (size_t)ncol * 9 * sizeof(double)));
    if (uref_traj) SIM_TRY(hipMalloc(&d_ur, (size_t)ncol * 6 * sizeof(double)));
    if (!wl) SIM_TRY(hipMalloc(&d_warmB, (size_t)B * N * NT * sizeof(double)));
    if (wl && wl->alloc_failed) {
        SIM_TRY(hipMalloc(&d_abad, (size_t)T * sizeof(int32_t)));
        SIM_TRY(hipMemsetAsync(d_abad, 0, (size_t)T * sizeof(int32_t), s));
    }
    if (u_hist) SIM_TRY(hipMalloc(&d_hist, (size_t)T * B * NT * sizeof(double)));
    SIM_TRY(hipMalloc(&d_bad, (size_t)T * sizeof(int32_t)));
    SIM_TRY(hipMemsetAsync(d_bad, 0, (size_t)T * sizeof(int32_t), s));
    SIM_TRY(hipMemcpyAsync(d_xr, xref_traj, (size_t)ncol * 9 * sizeof(double), hipMemcpyHostToDevice, s));
    if (uref_traj) SIM_TRY(hipMemcpyAsync(d_ur, uref_traj, (size_t)ncol * 6 * sizeof(double), hipMemcpyHostToDevice, s));
    SIM_TRY(hipMemcpyAsync(h->d_x0, x, (size_t)B * 13 * sizeof(double), hipMemcpyHostToDevice, s));
    SIM_TRY(hipMemcpyAsync(h->d_ub, ub, (size_t)B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    SIM_TRY(hipMemcpyAsync(h->d_stuck, stuck, (size_t)B * NT * sizeof(double), hipMemcpyHostToDevice, s));
    ftmpc::SimParams sp;
    sp.B = B;
    sp.x = h->d_x0;
    sp.u0 = h->d_u0;
    sp.ub = h->d_ub;
    sp.stuck = h->d_stuck;
    for (int i = 0; i < 4; ++i) sp.noise[i] = noise[i];
    sp.seed = seed;
    sp.u_hist = d_hist;
    sp.status = h->d_status;
    sp.bad_count = d_bad;
    const int64_t nw = B * (int64_t)N * NT;
    for (int t = 0; t < T; ++t) {
        // window t..t+N of the reference (column-major, so a plain pointer offset); warm start from step 1 on
        const double* Ufin = h->d_U;
        if (sqp_iters > 0) {     // the nonlinear program of this step by the line-search SQP, started from the shifted previous solution
            ftmpc::SqpState S;
            double* J0 = nullptr;
            rc = sqp_enqueue(h, B, d_xr + (int64_t)9 * t, 0, uref_traj ? d_ur + (int64_t)6 * t : nullptr, 0, t > 0 ? d_warmB : nullptr, sqp_iters,
                             backtracks, tol, S, &J0);
            if (rc == FTMPC_OK) {
                Ufin = S.U;
                sp.status = S.status;
                SIM_TRY(hipMemcpy2DAsync(h->d_u0, NT * sizeof(double), S.U, (size_t)N * NT * sizeof(double), NT * sizeof(double), (size_t)B,
                                         hipMemcpyDeviceToDevice, s));
            }
        } else if (wl) {         // the reference's two-stage structure: wrench MPC with the hull rows, then allocation
            rc = wrench_enqueue(h, B, wl->hull_rows, wl->has_set, d_xr + (int64_t)9 * t, 0, uref_traj ? d_ur + (int64_t)6 * t : nullptr, 0,
                                t > 0 ? h->d_warmG : nullptr);
            if (rc == FTMPC_OK && d_abad)
                hipLaunchKernelGGL(ftmpc::ftmpc_count_nonzero_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, (const int32_t*)h->d_ast2,
                                   d_abad + t);
        } else {
            rc = enqueue(h, B, h->d_x0, h->d_ub, h->d_stuck, d_xr + (int64_t)9 * t, 0, uref_traj ? d_ur + (int64_t)6 * t : nullptr, 0,
                         t > 0 ? d_warmB : nullptr, h->d_u0, h->d_U, h->d_status, h->d_iters, s, -1);
        }
        if (rc != FTMPC_OK) {
            cleanup();
            return rc;
        }
        sp.step = t;
        hipLaunchKernelGGL(ftmpc::ftmpc_plant_step_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, h->dc, sp);
        if (wl)     // wrench warm start: shifted by one stage, the last stage repeats
            hipLaunchKernelGGL(ftmpc::ftmpc_shift_warm_kernel, dim3((unsigned)((B * N * 6 + 255) / 256)), dim3(256), 0, s, B, N, 6,
                               (const double*)h->d_G, h->d_warmG, 1);
        else
            hipLaunchKernelGGL(ftmpc::ftmpc_shift_warm_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s, B, N, NT, Ufin, d_warmB, 0);
        SIM_TRY(hipGetLastError());
    }
    SIM_TRY(hipMemcpyAsync(x, h->d_x0, (size_t)B * 13 * sizeof(double), hipMemcpyDeviceToHost, s));
    if (u_hist) SIM_TRY(hipMemcpyAsync(u_hist, d_hist, (size_t)T * B * NT * sizeof(double), hipMemcpyDeviceToHost, s));
    if (not_converged) SIM_TRY(hipMemcpyAsync(not_converged, d_bad, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    if (d_abad) SIM_TRY(hipMemcpyAsync(wl->alloc_failed, d_abad, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    SIM_TRY(hipStreamSynchronize(s));
#undef SIM_TRY
    cleanup();
    return FTMPC_OK;
}

#ifdef FTMPC_STAMPS
/* diagnostic build only: copies the per-instance phase cycle totals (12 u64 per instance, first
 * `count` <= 4096 instances) of the last fp32 solve */
int ftmpc_debug_read_stamps(ftmpc_handle* h, int64_t count, unsigned long long* out) {
    const void* src = h ? (h->use_f64 ? (const void*)h->d_dbgH64 : (const void*)h->d_dbgH) : nullptr;
    if (!h || !out || count < 0 || count > 4096 || !src || (h->use_f64 && count > 512)) return FTMPC_ERR_ARG;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(out, src, (size_t)count * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return FTMPC_OK;
}
#endif

}  // extern "C"

#include "ftmpc_multi.hip"
