/*
 * ftmpc.h -- C-ABI of the MI355X-native batched MPC QP-step path.
 *
 * Drop-in boundary.  The reference (DISCOWER/fault-tolerant-mpc) has
 * no FFI: its seam is the duck-typed Python controller consumed by
 * ft_mpc/simulation/sim_env.py:82.  These entry points are what a ctypes binding behind
 *     SpiralingController.get_control(x0, t)   ft_mpc/controllers/spiraling_mpc.py:288-317
 *     SpiralingController.solve_mpc(c0)        ft_mpc/controllers/spiraling_mpc.py:319-354
 * calls; INTEGRATION.md shows that binding.  Plain C types only, caller-owned buffers, int
 * return codes (0 ok, <0 error) and per-instance status/iters arrays instead of the
 * reference's logged-but-ignored IPOPT status (spiraling_mpc.py:347-352) or the allocator's
 * exit() (controllers/tools/control_allocator.py:88-93).  No global state: one opaque
 * handle per GPU; distinct handles may be used concurrently from distinct host threads.
 *
 * All host-visible numbers are IEEE double (the reference works in numpy float64).
 * State layout x0 = [p(3), v(3), q(4; x,y,z,w), omega(3)]   ft_mpc/models/sys_model.py:35-41.
 */
#ifndef FTMPC_H
#define FTMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTMPC_MAX_NT 16
#define FTMPC_MAX_TERM_ROWS 80   /* rows of the terminal set (config/terminal.yaml: 72) */
#define FTMPC_MAX_TCOST_TERMS 24  /* non-quadratic terms of the terminal cost (config/terminal.yaml: 13 polynomial + 12 root terms) */
#define FTMPC_MAX_HULL_ROWS 128  /* facets of the generalized-force hull (26 for every fault set of the reference vehicle; 112 for a
                                    generic 8-thruster allocation matrix).  More than 32: float64 Riccati kernel only (no terminal set) */
#define FTMPC_NX 13
#define FTMPC_NOPT 9
#define FTMPC_NG 6

/* return codes */
#define FTMPC_OK 0
#define FTMPC_ERR_ARG (-1)       /* bad argument / unsupported shape */
#define FTMPC_ERR_HIP (-2)       /* HIP runtime failure; see ftmpc_last_error */
#define FTMPC_ERR_NODEVICE (-3)  /* no usable gfx950 device */
#define FTMPC_ERR_ALLOC (-4)

/* per-instance status[] values */
#define FTMPC_STATUS_CONVERGED 0   /* complementarity gap fell below mu_stop */
#define FTMPC_STATUS_MAXITER 1     /* stopped at max_iters (last iterate returned, like the reference) */
#define FTMPC_STATUS_NUMERIC 2     /* non-finite value met / factorisation broke down (e.g. an unreachable terminal set);
                                      the linearisation point (clip(warm start)) is returned */

#define FTMPC_STATUS_NO_HULL 3     /* generalized-force formulation only: the healthy thrusters do not span R^6, the input hull
                                      is flat (the reference's Qhull call fails on such fault sets); set by the host front-end */

/* arithmetic of the solve path */
#define FTMPC_DTYPE_F32 0
#define FTMPC_DTYPE_F64 1

/*
 * Problem constants.  Replaces what the reference hard-codes or reads from YAML:
 *   N        tuning.spiraling.horizon          ft_mpc/config/reactive.yaml:26, spiraling_mpc.py:38
 *   dt       time_step                         reactive.yaml:2
 *   mass,J   SystemModel                       ft_mpc/models/sys_model.py:52-57
 *   D        6 x NT allocation matrix, row-major, sys_model.py:73-123 (NT=16); runtime data so
 *            that the NT=8 benchmark can pass its synthetic matrix
 *   Q,R      diag weights                      reactive.yaml:32-33, spiraling_mpc.py:91-93
 *   P        9x9 terminal weight, row-major (quadratic part of ft_mpc/config/terminal.yaml)
 *   r        orbit radius vector               controllers/tools/spiral_parameters.py:39
 *   f_virt   virtual force                     spiral_parameters.py:34-36
 *   rho      min-energy allocation weight      controllers/tools/control_allocator.py:32
 *            (folded into the QP as strict-convexity regulariser; QP-spec)
 */
typedef struct ftmpc_config {
    int32_t N;          /* horizon stages, 1..64 */
    int32_t NT;         /* thrusters, 1..FTMPC_MAX_NT */
    int32_t dtype;      /* FTMPC_DTYPE_F32 | FTMPC_DTYPE_F64 : arithmetic of the IPM/KKT solve */
    int32_t max_iters;  /* IPM iteration cap (fixed upper bound; early exit at mu_stop) */
    int32_t device_id;  /* HIP device ordinal */
    int32_t struct_size; /* sizeof(ftmpc_config) of the header the caller was built against; ftmpc_default_config fills it and
                            ftmpc_create refuses any other value (FTMPC_ERR_ARG), so a caller built against an older, shorter
                            struct is told instead of being read past its end */
    double dt;
    double mass;
    double J[9];
    double D[FTMPC_NG * FTMPC_MAX_NT]; /* row-major 6 x NT, row stride NT */
    double Q[FTMPC_NOPT];
    double R[FTMPC_NG];
    double P[FTMPC_NOPT * FTMPC_NOPT];
    double r[3];
    double f_virt[3];
    double rho;
    double mu_stop;     /* stop when mean complementarity < mu_stop (<=0: library default) */
    /*
     * Terminal set  term_A (c_N[0:9] - xref_N) <= term_b  (the polytope of config/terminal.yaml, term_set; reference
     * spiraling_mpc.py:199-202).  terminal_set != 0 adds these rows to the QP; ftmpc_solve_batch then runs on the dense float64
     * kernel whatever the dtype (needs N * NT <= 256: checked when it is called), ftmpc_solve_wrench_batch as described there
     * (no such limit).  An instance whose terminal set cannot
     * be reached within the horizon ends with FTMPC_STATUS_MAXITER / _NUMERIC (the reference logs IPOPT's failure
     * and carries on, spiraling_mpc.py:347-352).
     */
    int32_t terminal_set;
    int32_t term_rows;  /* <= FTMPC_MAX_TERM_ROWS */
    double term_A[FTMPC_MAX_TERM_ROWS * FTMPC_NOPT];   /* row-major term_rows x 9 */
    double term_b[FTMPC_MAX_TERM_ROWS];
    /*
     * Non-quadratic part of the terminal cost (config/terminal.yaml `cost` beyond e'P e; reference spiraling_mpc.py:196):
     *     V_nq(e) = sum_i tc_poly_coef[i] prod_j e_j^tc_poly_exp[i][j] + sum_r tc_root_coef[r] (prod_j e_j^tc_root_exp[r][j] + tc_root_eps[r])^tc_root_pow[r] + tc_const
     * terminal_cost_terms != 0: every QP carries the exact gradient of V_nq at its linearisation point (the Hessian
     * stays 2P: the root terms are concave away from 0), and ftmpc_eval_cost_batch includes V_nq -- the two pieces of
     * the line-search SQP towards the reference's nonlinear program (ft_mpc_amd.BatchedMPC.solve_sqp).
     */
    int32_t terminal_cost_terms;
    int32_t tc_npoly, tc_nroot, tc_reserved;
    double tc_poly_coef[FTMPC_MAX_TCOST_TERMS];
    int32_t tc_poly_exp[FTMPC_MAX_TCOST_TERMS * FTMPC_NOPT];
    double tc_root_coef[FTMPC_MAX_TCOST_TERMS];
    double tc_root_eps[FTMPC_MAX_TCOST_TERMS];
    double tc_root_pow[FTMPC_MAX_TCOST_TERMS];
    int32_t tc_root_exp[FTMPC_MAX_TCOST_TERMS * FTMPC_NOPT];
    double tc_const;
    /*
     * Implementation switches (diagnostics and A/B runs; 0 = the library's choice everywhere).  They live here, per handle:
     * the library reads no environment variable and keeps no process-global state.
     *   kernel_select   FTMPC_KERNEL_AUTO | FTMPC_KERNEL_DENSE | FTMPC_KERNEL_WORKGROUP: with DENSE the Newton systems are always factorised in the
     *                   thruster variables (kernel 7 / the dense float64 kernel) even where the library would go through
     *                   the 6N-variable wrench-space form (kernel 8 / its float64 sibling)
     *   lin_split_max   batch size up to which the linearisation is split by tangent direction (0: library default 8192;
     *                   < 0: never split)
     *   stage_chunks    ranges the host-buffer entry point stages a batch in (0: whole blocks of 65 536 instances; 1..8)
     */
    int32_t kernel_select;
    int32_t stage_chunks;
    int64_t lin_split_max;
    /*
     * State bounds  xlb <= c_k <= xub  on the orbit-centre state [p, v, omega, q] of the stages k = 1 .. N-1 (the reference's
     * optional params "xub" / "xlb", spiraling_mpc.py:129-130,179-185: `con_ineq.append(x_t)` for t < N; the row of stage 0 does not
     * depend on the decision variables).  state_bounds != 0 adds them to the thruster-space QP of ftmpc_solve_batch; a component
     * with |bound| >= FTMPC_NO_BOUND has no row.  The solve then runs in float64 on the Riccati kernel whatever the dtype (N <= 40,
     * no terminal set; kernel_select is ignored): there the rows are a diagonal barrier term on the state weight of their stage,
     * not dense rows through the sensitivities.  Bounds that cannot be met within the horizon end with FTMPC_STATUS_MAXITER /
     * _NUMERIC (the reference logs IPOPT's failure and carries on, spiraling_mpc.py:347-352).
     */
    int32_t state_bounds;
    int32_t sb_reserved;
    double xlb[FTMPC_NX];
    double xub[FTMPC_NX];
} ftmpc_config;
#define FTMPC_NO_BOUND 1e300

#define FTMPC_KERNEL_AUTO 0
#define FTMPC_KERNEL_DENSE 1
#define FTMPC_KERNEL_WORKGROUP 2   /* the wrench-space form on a 4-wave workgroup per instance (kernel 8) where the library would
                                      give the instance one wave (kernel 10) */

typedef struct ftmpc_handle ftmpc_handle;

/* Fills *cfg with the reference constants for (N, NT) and struct_size.  NT==16 gets the reference D
 * (sys_model.py:73-123); any other NT leaves D zero for the caller to fill. */
int ftmpc_default_config(ftmpc_config* cfg, int32_t N, int32_t NT);

/* Creates a solver bound to cfg->device_id.  Fails (FTMPC_ERR_NODEVICE) when no gfx950
 * device is present: there is no CPU fallback in this library. */
int ftmpc_create(const ftmpc_config* cfg, ftmpc_handle** out);
int ftmpc_destroy(ftmpc_handle* h);
const char* ftmpc_last_error(const ftmpc_handle* h); /* h may be NULL: last create error */

/* Pre-allocates device workspace for batches up to max_batch (optional; solve calls grow
 * the workspace on demand, which allocates and must not happen inside a timed region). */
int ftmpc_reserve(ftmpc_handle* h, int64_t max_batch);

/*
 * One MPC step for B independent instances -- the batched form of get_control()
 * (spiraling_mpc.py:288-317) up to and including the thruster command.  HOST buffers.
 *
 *   x0      [B*13]        robot states
 *   ub      [B*NT]        per-thruster upper bound, 0 for a broken thruster (u_ub_physical,
 *                         sys_model.py:237-240)
 *   stuck   [B*NT]        intensity*f_max for a broken thruster else 0 (faulty_force,
 *                         sys_model.py:236)
 *   xref    [9*(N+1)] if xref_stride==0 (shared) else [B*xref_stride]; column-major 9 x (N+1)
 *                         exactly the reference's x_sp (spiraling_mpc.py:295)
 *   uref    NULL (== hover, u_ref = 0) or as xref with 6 x (N+1)   (spiraling_mpc.py:296)
 *   warmU   NULL (cold start: linearise about thrusters-off) or [B*N*NT] in/out: on entry the
 *           previous solution ALREADY shifted by the caller's policy (the reference shifts by
 *           one stage, spiraling_mpc.py:324-334; ftmpc_shift_warm does that); on exit U*.
 *   out_u0  [B*NT]        thruster forces of stage 0 (0 at broken thrusters)
 *   out_U   NULL or [B*N*NT]
 *   status  NULL or [B]   FTMPC_STATUS_*
 *   iters   NULL or [B]   IPM iterations run
 */
int ftmpc_solve_batch(ftmpc_handle* h, int64_t B,
                      const double* x0, const double* ub, const double* stuck,
                      const double* xref, int64_t xref_stride,
                      const double* uref, int64_t uref_stride,
                      double* warmU,
                      double* out_u0, double* out_U,
                      int32_t* status, int32_t* iters);

/*
 * Value of the NONLINEAR program's cost for given thruster sequences -- the merit function of the line-search SQP:
 * nonlinear RK4 rollout of the orbit-centre model from robot_to_center(x0) under U (spiral_model.py:44-76,
 * sys_model.py:138-162), then
 *     sum_{k=1}^{N-1} e_k'Q e_k + V(e_N) + sum_{k<N} [ut_k'R ut_k + rho |u_k|^2],   V(e) = e'P e [+ V_nq(e) when
 * terminal_cost_terms != 0],  ut_k = D (u_k + stuck) - [Rot(q_k)^T uref_k[0:3]; uref_k[3:6]] - [f_virt; 0]
 * (spiraling_mpc.py:156-171,188,196; q_k is the rollout's own quaternion, as the decision-variable quaternion is there).
 *   U [B*N*NT] thruster forces (entries of broken thrusters are ignored);  out_cost [B].   HOST buffers.
 */
int ftmpc_eval_cost_batch(ftmpc_handle* h, int64_t B,
                          const double* x0, const double* ub, const double* stuck,
                          const double* xref, int64_t xref_stride,
                          const double* uref, int64_t uref_stride,
                          const double* U, double* out_cost);

/*
 * Line-search sequential QP towards the reference's NONLINEAR program (nonlinear RK4 dynamics, full terminal cost when
 * terminal_cost_terms != 0; spiraling_mpc.py:87-238 as IPOPT solves it, :346), entirely on the device: per major iteration
 * one QP step linearised about the current thruster sequences U (ftmpc_solve_batch's QP; Hessian 2 (B'QB + R + rho I) with
 * the quadratic terminal weight, gradient exact), then backtracking alpha = 1, 1/2, ... (`backtracks` trial points) along
 * clip(U_qp) - U on the TRUE cost (ftmpc_eval_cost_batch's kernel) until it decreases by more than tol (1 + |J|); an
 * instance without such a step stops.  Nothing crosses PCIe between the upload of the inputs and the download of the results.
 *   warmU         NULL (start from thrusters off) or [B*N*NT] start sequences (clipped to the bounds)
 *   out_cost / out_cost0   [B] cost of the returned sequences / of the start point;  out_sqp_iters [B] major iterations that
 *   made progress;  out_iters [B] interior-point iterations summed;  status [B] of the last QP.   HOST buffers.
 */
int ftmpc_solve_sqp_batch(ftmpc_handle* h, int64_t B,
                          const double* x0, const double* ub, const double* stuck,
                          const double* xref, int64_t xref_stride,
                          const double* uref, int64_t uref_stride,
                          const double* warmU, int32_t sqp_iters, int32_t backtracks, double tol,
                          double* out_u0, double* out_U, double* out_cost, double* out_cost0,
                          int32_t* out_sqp_iters, int32_t* out_iters, int32_t* status);

/* ftmpc_solve_sqp_batch is a few hundred small launches per call; a call that repeats the previous call's shape (batch size,
 * strides, warm start or not, iteration counts, tolerance; the same handle constants, no workspace growth in between) is recorded
 * into a hipGraph the second time and replayed with one launch from the third on -- for batches up to 512 instances, where it
 * pays (FTMPC_SQP_GRAPH=1 in the environment: for every batch size; =0: never; profiling keeps the direct launches too).  Returns how many calls of this handle were replayed from a graph (diagnostic; -1: NULL). */
int64_t ftmpc_sqp_graph_launches(const ftmpc_handle* h);

/* Same contract with DEVICE pointers (HBM-resident inputs/outputs) enqueued on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  Asynchronous: returns after
 * enqueue.  warmU is read only; pass the same buffer as out_U to update it in place. */
int ftmpc_solve_batch_device(ftmpc_handle* h, int64_t B,
                             const double* x0, const double* ub, const double* stuck,
                             const double* xref, int64_t xref_stride,
                             const double* uref, int64_t uref_stride,
                             const double* warmU,
                             double* out_u0, double* out_U,
                             int32_t* status, int32_t* iters,
                             void* stream);

/*
 * The reference's own two-stage structure: one MPC step in the 6-D GENERALIZED-FORCE space with the
 * input hull as constraint (spiraling_mpc.py:133-137,175-177; controllers/tools/input_bounds.py:43-76), followed by the
 * min-norm thruster allocation (control_allocator.py:65-94).  Per instance
 *     decision  tau_k in R^6, k < N: the TOTAL generalized force on the body ( = the reference's u_t + u_r + u_comp + D f_fault )
 *     cost      as ftmpc_solve_batch with ut_k = tau_k - ur_k - [f_virt;0] ( = the reference's deviation input u_t ), no rho term
 *     s.t.      hull_A tau_k <= hull_b  for every stage  [+ the terminal set when the handle's config has terminal_set != 0]
 * and u0 = argmin |u|^2 s.t. D u = tau_0 - D stuck, 0 <= u <= ub.  Needs N <= 40 and hull_rows <= FTMPC_MAX_HULL_ROWS (with
 * kernel_select = FTMPC_KERNEL_DENSE or N > 40: 6 N <= 256, hull_rows <= 32, N * hull_rows <= 1024).
 * Every converged interior-point iterate is finished by an active-set polish (the exact solution on the active set the iterate
 * identifies, its multiplier and slack signs verified; each polish round that factorises counts in iters[]): the float64 kernels
 * return the oracle's polished solution to 1e-9 f_max.  Routing: dtype FTMPC_DTYPE_F32 with N <= 16 and hull_rows <= 32 runs on one
 * fp32 wave per instance (ftmpc_solve_hull32_kernel; measured on 8 192 boundary vehicles: 1.4e-6 f_max worst; specification 1e-4
 * f_max on wrenches and on the allocated thrust command) and hands what it does not certify to the float64 path inside the same
 * call (ftmpc_last_handed_over); everything else -- float64 handles, N up to 40, up to 128 hull rows, with or without the terminal
 * set -- on ftmpc_solve_ricw64_kernel (float64, Riccati recursion, one wave per instance).
 *   hull_A    [n_sets][hull_rows*6] row-major facet normals, one table per fault INDEX SET (the normals do not depend on
 *             the fault intensities);  hull_set [B] table number of every instance (NULL: table 0 for all)
 *   hull_b    [B*hull_rows] facet offsets (they carry the intensities: b = n . D (ub/2 + stuck) + sum_i |n . D_i| ub_i / 2);
 *             pad unused rows with a zero normal and b = 1
 *   warmG     NULL (linearise about thrusters off: tau = D stuck) or [B*N*6] in/out: previous wrench solution, already shifted
 *   out_u0    [B*NT] allocated thruster forces;  out_tau0 NULL or [B*6];  out_G NULL or [B*N*6] whole-horizon wrenches
 *   status / iters: IPM (as above);  alloc_status NULL or [B]: as ftmpc_allocate_batch
 * HOST buffers.
 */
int ftmpc_solve_wrench_batch(ftmpc_handle* h, int64_t B,
                             const double* x0, const double* ub, const double* stuck,
                             const double* hull_A, int32_t n_sets, const int32_t* hull_set, const double* hull_b, int32_t hull_rows,
                             const double* xref, int64_t xref_stride,
                             const double* uref, int64_t uref_stride,
                             double* warmG,
                             double* out_u0, double* out_tau0, double* out_G,
                             int32_t* status, int32_t* iters, int32_t* alloc_status);

/* Number of instances of the LAST ftmpc_solve_wrench_batch / ftmpc_simulate_wrench_batch step on this handle that the one-wave fp32
 * kernel handed over to the float64 kernel (its active-set polish did not settle, or hull and terminal rows were active together);
 * 0 where the float64 kernel solved the whole batch anyway.  Blocks until that step's kernels have finished. */
int ftmpc_last_handed_over(ftmpc_handle* h, int64_t* count);

/*
 * Batched thruster allocation: the reference's second stage,
 * ControlAllocator.get_physical_input (ft_mpc/controllers/tools/control_allocator.py:27-40,65-94):
 *     min |u|^2   s.t.  D u = tau,  0 <= u <= ub
 * with the handle's D (6 x NT).  The thruster-space MPC step does not call it (its QP allocates
 * inside); it serves callers that hold a generalized force, as the reference's controller does.
 * HOST buffers.
 *   tau     [B*6]    generalized force to realise with the healthy thrusters (the reference passes
 *                    u_des, i.e. the controller output with the faulty wrench already removed)
 *   ub      [B*NT]   upper bounds, 0 for a broken thruster (u_ub_physical, sys_model.py:237-240)
 *   out_u   [B*NT]   thruster forces
 *   status  NULL or [B]: 0 solved (|D u - tau|_inf <= 1e-8 (1 + |tau|_inf)), 1 iteration cap,
 *                    2 tau not attainable (least-residual u returned; the reference prints and
 *                    exit()s here, control_allocator.py:88-93)
 *   iters   NULL or [B]: Newton steps taken
 */
int ftmpc_allocate_batch(ftmpc_handle* h, int64_t B, const double* tau, const double* ub,
                         double* out_u, int32_t* status, int32_t* iters);

/* Shifts a [B*N*NT] host warm-start buffer by one stage in place, zero-filling the last
 * stage (spiraling_mpc.py:327-329). */
int ftmpc_shift_warm(int64_t B, int32_t N, int32_t NT, double* warmU);

/*
 * T closed-loop steps for B independent vehicles entirely on the device -- the batched form of
 * SimulationEnvironment.run_simulation (ft_mpc/simulation/sim_env.py:77-112) around get_control:
 * per step  u = MPC step (warm-started from the shifted previous solution, spiraling_mpc.py:324-334),
 * x <- RK4 plant step with u (sys_model.py:138-226), x += U(0, noise) per component (sim_env.py:88-91,
 * here from a counter-based generator keyed by `seed` instead of the unseeded global RNG), quaternion
 * renormalised (sim_env.py:93).  HOST buffers; nothing crosses PCIe between steps.
 *   x            [B*13]  in: initial states, out: final states
 *   xref_traj    9 x (T+N) column-major, shared by all vehicles: step t tracks columns t..t+N
 *                (the padded trajectory of assign_trajectory, spiraling_mpc.py:255-286)
 *   uref_traj    NULL (hover) or 6 x (T+N) column-major
 *   noise        amplitudes {position, velocity, orientation, angular velocity} (sim_env.py:25-30: 1e-3 each)
 *   u_hist       NULL or [T*B*NT]: applied thruster commands
 *   not_converged NULL or [T]: number of instances whose IPM status was not 0 at each step
 */
int ftmpc_simulate_batch(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                         const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                         double* u_hist, int32_t* not_converged);

/* The same closed loop with the NONLINEAR program solved at every step (sqp_iters > 0: that many major iterations of the
 * line-search SQP of ftmpc_solve_sqp_batch, started from the shifted previous solution; sqp_iters = 0: ftmpc_simulate_batch). */
int ftmpc_simulate_batch_ex(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                            const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                            int32_t sqp_iters, int32_t backtracks, double tol, double* u_hist, int32_t* not_converged);

/* The same closed loop in the reference's TWO-STAGE structure (ftmpc_solve_wrench_batch at every step: generalized-force MPC with
 * the input hull -- and the terminal set when the handle's config has one -- then the min-norm allocation): the wrench warm start
 * is the previous solution shifted by one stage with its last stage repeated.  The hull tables are those of
 * ftmpc_solve_wrench_batch and stay fixed over the run (the fault pattern does not change inside a run).
 *   alloc_failed  NULL or [T]: number of instances whose allocation status was not 0 at each step */
int ftmpc_simulate_wrench_batch(ftmpc_handle* h, int64_t B, int32_t T, double* x, const double* ub, const double* stuck,
                                const double* hull_A, int32_t n_sets, const int32_t* hull_set, const double* hull_b, int32_t hull_rows,
                                const double* xref_traj, const double* uref_traj, const double noise[4], uint64_t seed,
                                double* u_hist, int32_t* not_converged, int32_t* alloc_failed);

/* Per-kernel device timing of the LAST solve call, measured with hipEvents on the launch
 * stream when enabled.  ms[slot] is the duration of kernel slot `slot` (0 when that kernel was
 * not launched), for slot < min(n_slots, FTMPC_KERNEL_SLOTS); ftmpc_kernel_name(slot) is the kernel's name as it appears
 * in rocprofv3 traces:  0 linearise, 1..3 condense+IPM fp32 (one wave per instance) for n <= 128 / 144 / 160,
 * 4 condense+IPM fp64, dense (workgroup per instance, general n), 5 condense+IPM fp32, workgroup per instance, for
 * 160 < N*NT: ftmpc_solve_wsw32_kernel (Newton systems through the 6N-variable wrench-space form, one wave per instance:
 * N <= 16 with N*NT <= 256, N <= 21 with N*NT <= 384), ftmpc_solve_ws32_kernel (the same form on a workgroup per instance:
 * kernel_select = FTMPC_KERNEL_WORKGROUP) or, with kernel_select = FTMPC_KERNEL_DENSE and N*NT <= 240, the dense
 * ftmpc_solve_wg32_kernel<15> (the slot reports all three names), 6 condense+IPM fp64 through the wrench-space form (ftmpc_solve_ws64_kernel, 6 N <= 256). */
#define FTMPC_KERNEL_SLOTS 7
int ftmpc_set_profiling(ftmpc_handle* h, int32_t enabled);
int ftmpc_last_kernel_ms(ftmpc_handle* h, float* ms, int32_t n_slots);
const char* ftmpc_kernel_name(int32_t slot);
/* the ONE kernel slot `slot` launches on this handle (slot 5 names three kernels above; which of them runs is the handle's routing:
 * its shape and kernel_select) */
const char* ftmpc_routed_kernel_name(const ftmpc_handle* h, int32_t slot);

/*
 * Test hook: runs the build for instance `inst` of a host batch and returns the condensed
 * QP the solve kernel sees (active thrusters only):  n = N*na,  H [n*n] row-major,
 * g [n], lo [n], hi [n] (bounds on d = U - Ubar), all as double.  *n_out receives n.
 * H_cap is the capacity of H in elements (>= n*n).
 */
int ftmpc_debug_build_qp(ftmpc_handle* h, int64_t B,
                         const double* x0, const double* ub, const double* stuck,
                         const double* xref, int64_t xref_stride,
                         const double* uref, int64_t uref_stride,
                         const double* warmU, int64_t inst,
                         double* H, int64_t H_cap, double* g, double* lo, double* hi,
                         int32_t* n_out);

/*
 * The batch axis across the GPUs of one node from ONE process.  The reference
 * never couples instances (one controller object per vehicle, spiraling_mpc.py:288-317), so device g owns
 * the contiguous range [B g / G, B (g+1) / G) (ftmpc_multi_shard_bounds); one host thread + one handle + one
 * stream set per device, no collective, nothing on xGMI: every device reads its slice of the caller's arrays
 * and writes its slice of the caller's outputs.
 *   device_ids  NULL (devices 0..n_devices-1) or n_devices ordinals; an ordinal may repeat (several handles
 *               on one GPU);  n_devices <= 0: every visible device.
 * ftmpc_multi_solve_batch has the contract of ftmpc_solve_batch (host buffers, pinned staging per device).
 * ftmpc_multi_upload keeps the shards RESIDENT in each device's HBM; ftmpc_multi_step then runs `steps` MPC
 * steps over them on every device at once (keep_U != 0: the whole-horizon solution is kept too) and returns
 * when all devices are idle; ftmpc_multi_download gathers u0 [B*NT], U [B*N*NT] (needs keep_U), status, iters
 * (each may be NULL) on the host.
 */
typedef struct ftmpc_multi ftmpc_multi;
int ftmpc_multi_create(const ftmpc_config* cfg, const int32_t* device_ids, int32_t n_devices, ftmpc_multi** out);
int ftmpc_multi_destroy(ftmpc_multi* m);
const char* ftmpc_multi_last_error(const ftmpc_multi* m); /* m may be NULL: last create error */
int32_t ftmpc_multi_device_count(const ftmpc_multi* m);
/* host cores the worker thread of device slot `slot` is bound to: the cores nearest its GPU (sysfs local_cpulist of the PCI
 * function) that the process may use; 0 when the two sets do not meet and the affinity was left alone */
int32_t ftmpc_multi_worker_cpus(const ftmpc_multi* m, int32_t slot);
int ftmpc_multi_shard_bounds(const ftmpc_multi* m, int64_t B, int32_t slot, int64_t* lo, int64_t* hi);
int ftmpc_multi_solve_batch(ftmpc_multi* m, int64_t B,
                            const double* x0, const double* ub, const double* stuck,
                            const double* xref, int64_t xref_stride,
                            const double* uref, int64_t uref_stride,
                            double* warmU,
                            double* out_u0, double* out_U,
                            int32_t* status, int32_t* iters);
int ftmpc_multi_upload(ftmpc_multi* m, int64_t B,
                       const double* x0, const double* ub, const double* stuck,
                       const double* xref, int64_t xref_stride,
                       const double* uref, int64_t uref_stride,
                       const double* warmU);
int ftmpc_multi_step(ftmpc_multi* m, int32_t steps, int32_t keep_U);
int ftmpc_multi_download(ftmpc_multi* m, double* out_u0, double* out_U, int32_t* status, int32_t* iters);
/* per-kernel device timing of device slot `slot` (see ftmpc_set_profiling / ftmpc_last_kernel_ms) */
int ftmpc_multi_set_profiling(ftmpc_multi* m, int32_t enabled);
int ftmpc_multi_last_kernel_ms(ftmpc_multi* m, int32_t slot, float* ms, int32_t n_slots);
const char* ftmpc_multi_routed_kernel_name(const ftmpc_multi* m, int32_t slot); /* ftmpc_routed_kernel_name of the device handles */

/* Library/ABI version: major*10000 + minor*100 + patch. */
int32_t ftmpc_version(void);
/* Hash of the kernel and host sources this binary was built from (csrc/Makefile): two builds of the same sources -- the
 * shipped one and the `plain` diagnostic build with the asm-side wait states as written -- report the same string. */
const char* ftmpc_build_id(void);

#ifdef __cplusplus
}
#endif
#endif /* FTMPC_H */
