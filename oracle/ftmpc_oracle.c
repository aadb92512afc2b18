/*
 * oracle/ftmpc_oracle.c -- plain-C float64 restatement of the MPC QP-step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the product library (fault-tolerant-mpc_amd/csrc) never links or calls it.
 * It is the checker and the reported CPU baseline ("port"), never the thing shipped.
 *
 * Parity status: physics pinned by the reference's source text/data (see oracle/refmath.py
 * and tests/test_oracle_pinned.py); the QP solution is pinned against an independent exact
 * solver (scipy BVLS) in tests/test_oracle_qp.py; parity with the reference's IPOPT NLP
 * output is UNPINNED (casadi/cvxpy absent from this image, no golden outputs in the repo).
 *
 * What it restates (paths relative to /root/reference):
 *   centre dynamics + RK4          ft_mpc/models/spiral_model.py:44-76, models/sys_model.py:138-162
 *   quaternion operators           ft_mpc/util/utils.py:4-19, models/sys_model.py:8-29
 *   robot -> centre transform      ft_mpc/models/spiral_model.py:91-109
 *   fault bookkeeping              ft_mpc/models/sys_model.py:198-208,228-243
 *   cost / reference window        ft_mpc/controllers/spiraling_mpc.py:76,156-171,188,196
 *   allocation objective (rho)     ft_mpc/controllers/tools/control_allocator.py:32
 * and the QP-spec of SURVEY.md section 8(a) / oracle/qp_oracle.py (same algorithm, same order).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_NT 16

typedef struct {
    int32_t N, NT, max_iters, flags; /* flags bit 0: no active-set polish (the interior-point iteration to mu_stop alone) */
    double dt, mass;
    double J[9];
    double D[6 * MAX_NT]; /* row-major 6 x NT, row stride NT */
    double Q[9], R[6], P[81];
    double r[3], f_virt[3];
    double rho, mu_stop;
} oracle_config;

/* ------------------------------------------------------------------ small helpers */
static void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
static void mv3(const double* M, const double* v, double* o) {
    for (int i = 0; i < 3; ++i) o[i] = M[3 * i] * v[0] + M[3 * i + 1] * v[1] + M[3 * i + 2] * v[2];
}
static void skew(const double* a, double* S) {
    S[0] = 0; S[1] = -a[2]; S[2] = a[1];
    S[3] = a[2]; S[4] = 0; S[5] = -a[0];
    S[6] = -a[1]; S[7] = a[0]; S[8] = 0;
}
static void mm3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
static int inv3(const double* M, double* o) {
    const double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4], f = M[5], g = M[6], h = M[7], i = M[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0) return -1;
    const double s = 1.0 / det;
    o[0] = (e * i - f * h) * s; o[1] = (c * h - b * i) * s; o[2] = (b * f - c * e) * s;
    o[3] = (f * g - d * i) * s; o[4] = (a * i - c * g) * s; o[5] = (c * d - a * f) * s;
    o[6] = (d * h - e * g) * s; o[7] = (b * g - a * h) * s; o[8] = (a * e - b * d) * s;
    return 0;
}
/* world->body rotation, utils.py:15-19 (no unit-norm assumption) */
static void rot(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = x * x - y * y - z * z + w * w; R[1] = 2 * (x * y + z * w); R[2] = 2 * (x * z - y * w);
    R[3] = 2 * (x * y - z * w); R[4] = -x * x + y * y - z * z + w * w; R[5] = 2 * (y * z + x * w);
    R[6] = 2 * (x * z + y * w); R[7] = 2 * (y * z - x * w); R[8] = -x * x - y * y + z * z + w * w;
}
static void rotT_mul(const double* q, const double* a, double* o) { /* Rot(q)^T a */
    double R[9];
    rot(q, R);
    for (int i = 0; i < 3; ++i) o[i] = R[i] * a[0] + R[3 + i] * a[1] + R[6 + i] * a[2];
}

/* ------------------------------------------------------------------ centre dynamics
 * c = [p(3), v(3), w(3), q(4)], gen = total wrench [F;tau]; spiral_model.py:44-76 */
static void centre_f(const oracle_config* c, const double* Jinv, const double* s, const double* gen, double* ds,
                     double* ab_out) {
    const double *v = s + 3, *w = s + 6, *q = s + 9;
    double Jw[3], wJw[3], t[3], dw[3], wr[3], wwr[3], dwr[3], ab[3];
    mv3(c->J, w, Jw);
    cross3(w, Jw, wJw);
    for (int i = 0; i < 3; ++i) t[i] = gen[3 + i] - wJw[i];
    mv3(Jinv, t, dw);
    cross3(w, c->r, wr);
    cross3(w, wr, wwr);
    cross3(dw, c->r, dwr);
    for (int i = 0; i < 3; ++i) ab[i] = gen[i] / c->mass + dwr[i] + wwr[i];
    ds[0] = v[0]; ds[1] = v[1]; ds[2] = v[2];
    rotT_mul(q, ab, ds + 3);
    ds[6] = dw[0]; ds[7] = dw[1]; ds[8] = dw[2];
    /* qdot = 1/2 Omega(w) q, sys_model.py:18-29 */
    ds[9] = 0.5 * (w[2] * q[1] - w[1] * q[2] + w[0] * q[3]);
    ds[10] = 0.5 * (-w[2] * q[0] + w[0] * q[2] + w[1] * q[3]);
    ds[11] = 0.5 * (w[1] * q[0] - w[0] * q[1] + w[2] * q[3]);
    ds[12] = 0.5 * (-w[0] * q[0] - w[1] * q[1] - w[2] * q[2]);
    if (ab_out) memcpy(ab_out, ab, 3 * sizeof(double));
}

/* continuous Jacobians f_c (13x13) and f_g (13x6); SURVEY.md Appendix A */
static void centre_jac(const oracle_config* c, const double* Jinv, const double* s, const double* gen, double* fc,
                       double* fg) {
    const double *w = s + 6, *q = s + 9;
    double ds[13], ab[3];
    centre_f(c, Jinv, s, gen, ds, ab);
    memset(fc, 0, 169 * sizeof(double));
    memset(fg, 0, 78 * sizeof(double));
    double R[9];
    rot(q, R);
    double Jw[3], Sw[9], SJw[9], M[9], dwdw[9];
    mv3(c->J, w, Jw);
    skew(w, Sw);
    skew(Jw, SJw);
    mm3(Sw, c->J, M);
    for (int i = 0; i < 9; ++i) M[i] -= SJw[i];
    mm3(Jinv, M, dwdw);
    for (int i = 0; i < 9; ++i) dwdw[i] = -dwdw[i];
    double Sr[9], wr[3], Swr[9], T1[9], T2[9], dab[9];
    skew(c->r, Sr);
    cross3(w, c->r, wr);
    skew(wr, Swr);
    mm3(Sr, dwdw, T1);
    mm3(Sw, Sr, T2);
    for (int i = 0; i < 9; ++i) dab[i] = -T1[i] - Swr[i] - T2[i];
    /* p rows */
    fc[0 * 13 + 3] = fc[1 * 13 + 4] = fc[2 * 13 + 5] = 1.0;
    /* v rows: R^T dab (w cols), d(R^T ab)/dq (q cols) */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) fc[(3 + i) * 13 + 6 + j] = R[i] * dab[j] + R[3 + i] * dab[3 + j] + R[6 + i] * dab[6 + j];
    {
        const double x = q[0], y = q[1], z = q[2], ww = q[3];
        const double d0[12] = {x, -y, -z, ww, y, x, ww, z, z, -ww, x, -y};
        const double d1[12] = {y, x, -ww, -z, -x, y, -z, ww, ww, z, y, x};
        const double d2[12] = {z, ww, x, y, -ww, z, y, -x, -x, -y, z, ww};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j)
                fc[(3 + i) * 13 + 9 + j] = 2 * (ab[0] * d0[4 * i + j] + ab[1] * d1[4 * i + j] + ab[2] * d2[4 * i + j]);
    }
    /* w rows */
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) fc[(6 + i) * 13 + 6 + j] = dwdw[3 * i + j];
    /* q rows: 1/2 Xi(q) (w cols), 1/2 Omega(w) (q cols) */
    {
        const double x = q[0], y = q[1], z = q[2], ww = q[3];
        const double xi[12] = {ww, -z, y, z, ww, -x, -y, x, ww, -x, -y, -z};
        const double om[16] = {0, w[2], -w[1], w[0], -w[2], 0, w[0], w[1], w[1], -w[0], 0, w[2], -w[0], -w[1], -w[2], 0};
        for (int i = 0; i < 4; ++i) {
            for (int j = 0; j < 3; ++j) fc[(9 + i) * 13 + 6 + j] = 0.5 * xi[3 * i + j];
            for (int j = 0; j < 4; ++j) fc[(9 + i) * 13 + 9 + j] = 0.5 * om[4 * i + j];
        }
    }
    /* wrench columns */
    double SrJ[9];
    mm3(Sr, Jinv, SrJ);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            fg[(3 + i) * 6 + j] = R[3 * j + i] / c->mass; /* R^T / m */
            fg[(3 + i) * 6 + 3 + j] = -(R[i] * SrJ[j] + R[3 + i] * SrJ[3 + j] + R[6 + i] * SrJ[6 + j]);
            fg[(6 + i) * 6 + 3 + j] = Jinv[3 * i + j];
        }
}

static void mm13(const double* A, const double* B, int bc, double* C) { /* (13x13)(13xbc) */
    for (int i = 0; i < 13; ++i)
        for (int j = 0; j < bc; ++j) {
            double s = 0;
            for (int k = 0; k < 13; ++k) s += A[13 * i + k] * B[bc * k + j];
            C[bc * i + j] = s;
        }
}

/* one RK4 step with Jacobians: sys_model.py:152-158 and the chain rule through it */
static void rk4_jac(const oracle_config* c, const double* Jinv, const double* s, const double* gen, double* nxt,
                    double* A, double* Bg) {
    const double dt = c->dt;
    double k[4][13], sp[4][13], fc[4][169], fg[4][78];
    const double a[4] = {0, dt / 2, dt / 2, dt};
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 13; ++j) sp[i][j] = s[j] + (i ? a[i] * k[i - 1][j] : 0.0);
        centre_f(c, Jinv, sp[i], gen, k[i], 0);
        centre_jac(c, Jinv, sp[i], gen, fc[i], fg[i]);
    }
    for (int j = 0; j < 13; ++j) nxt[j] = s[j] + dt / 6 * (k[0][j] + 2 * k[1][j] + 2 * k[2][j] + k[3][j]);
    double K[4][169], G[4][78], T[169], U[78];
    memcpy(K[0], fc[0], sizeof(K[0]));
    memcpy(G[0], fg[0], sizeof(G[0]));
    for (int i = 1; i < 4; ++i) {
        for (int e = 0; e < 169; ++e) T[e] = a[i] * K[i - 1][e];
        for (int d = 0; d < 13; ++d) T[14 * d] += 1.0;
        mm13(fc[i], T, 13, K[i]);
        for (int e = 0; e < 78; ++e) U[e] = a[i] * G[i - 1][e];
        mm13(fc[i], U, 6, G[i]);
        for (int e = 0; e < 78; ++e) G[i][e] += fg[i][e];
    }
    for (int e = 0; e < 169; ++e) A[e] = dt / 6 * (K[0][e] + 2 * K[1][e] + 2 * K[2][e] + K[3][e]);
    for (int d = 0; d < 13; ++d) A[14 * d] += 1.0;
    for (int e = 0; e < 78; ++e) Bg[e] = dt / 6 * (G[0][e] + 2 * G[1][e] + 2 * G[2][e] + G[3][e]);
}

/* ------------------------------------------------------------------ QP build
 * Returns n (active variables).  H is n x n row-major (caller buffer of (N*NT)^2), g,lo,hi n. */
typedef struct {
    double *H, *g, *lo, *hi, *G, *Gn, *ubar; /* work */
    int act[MAX_NT];
} qp_work;

static int build_qp(const oracle_config* c, const double* x0, const double* ub, const double* stuck,
                    const double* xref, const double* uref, const double* warm, qp_work* W) {
    const int N = c->N, NT = c->NT;
    double Jinv[9];
    inv3(c->J, Jinv);
    int na = 0;
    for (int i = 0; i < NT; ++i)
        if (ub[i] > 0) W->act[na++] = i;
    const int n = N * na;
    if (n == 0) return 0;
    /* robot -> centre, spiral_model.py:103-109 */
    double s[13], wr[3], a1[3], a2[3];
    const double *q0 = x0 + 6, *w0 = x0 + 10;
    rotT_mul(q0, c->r, a1);
    cross3(w0, c->r, wr);
    rotT_mul(q0, wr, a2);
    for (int i = 0; i < 3; ++i) {
        s[i] = x0[i] + a1[i];
        s[3 + i] = x0[3 + i] + a2[i];
        s[6 + i] = w0[i];
    }
    for (int i = 0; i < 4; ++i) s[9 + i] = q0[i];
    double* H = W->H;
    double* g = W->g;
    memset(H, 0, (size_t)n * n * sizeof(double));
    memset(g, 0, (size_t)n * sizeof(double));
    double* G = W->G;   /* 13 x n, d c_k / d U_act */
    double* Gn = W->Gn;
    memset(G, 0, (size_t)13 * n * sizeof(double));
    /* Da' R Da + rho I */
    double MR[MAX_NT * MAX_NT];
    for (int a = 0; a < na; ++a)
        for (int b = 0; b < na; ++b) {
            double t = 0;
            for (int gg = 0; gg < 6; ++gg) t += c->D[gg * NT + W->act[a]] * c->R[gg] * c->D[gg * NT + W->act[b]];
            MR[a * na + b] = t + (a == b ? c->rho : 0.0);
        }
    for (int k = 0; k < N; ++k) {
        double gen[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < NT; ++i) {
            double u = 0;
            if (warm && ub[i] > 0) {
                u = warm[k * NT + i];
                u = u < 0 ? 0 : (u > ub[i] ? ub[i] : u);
            }
            W->ubar[k * NT + i] = u;
            for (int gg = 0; gg < 6; ++gg) gen[gg] += c->D[gg * NT + i] * (u + stuck[i]);
        }
        /* deviation input ut = gen - ur - [f_virt;0]; spiraling_mpc.py:156-171 */
        double ut[6];
        {
            double ur[6] = {0, 0, 0, 0, 0, 0};
            if (uref) {
                rotT_mul(s + 9, uref + 6 * k, ur);
                ur[3] = uref[6 * k + 3]; ur[4] = uref[6 * k + 4]; ur[5] = uref[6 * k + 5];
            }
            for (int gg = 0; gg < 6; ++gg) ut[gg] = gen[gg] - ur[gg] - (gg < 3 ? c->f_virt[gg] : 0.0);
        }
        double nxt[13], A[169], Bg[78];
        rk4_jac(c, Jinv, s, gen, nxt, A, Bg);
        memcpy(s, nxt, sizeof(nxt));
        /* G <- A G ; new block B_k Da */
        const int used = k * na;
        for (int i = 0; i < 13; ++i)
            for (int j = 0; j < used; ++j) {
                double t = 0;
                for (int kk = 0; kk < 13; ++kk) t += A[13 * i + kk] * G[kk * n + j];
                Gn[i * n + j] = t;
            }
        for (int i = 0; i < 13; ++i) {
            memcpy(G + (size_t)i * n, Gn + (size_t)i * n, (size_t)used * sizeof(double));
            for (int a = 0; a < na; ++a) {
                double t = 0;
                for (int gg = 0; gg < 6; ++gg) t += Bg[6 * i + gg] * c->D[gg * NT + W->act[a]];
                G[i * n + used + a] = t;
            }
        }
        const int cols = used + na;
        /* tracking term of stage k+1: 2 E' W E, 2 E' W e   (W = diag Q, or P at the terminal stage) */
        double e[9], We[9];
        for (int i = 0; i < 9; ++i) e[i] = s[i] - xref[9 * (k + 1) + i];
        const int terminal = (k + 1 == N);
        for (int i = 0; i < 9; ++i) {
            if (terminal) {
                double t = 0;
                for (int j = 0; j < 9; ++j) t += c->P[9 * i + j] * e[j];
                We[i] = t;
            } else
                We[i] = c->Q[i] * e[i];
        }
        for (int j = 0; j < cols; ++j) {
            double t = 0;
            for (int i = 0; i < 9; ++i) t += G[i * n + j] * We[i];
            g[j] += 2 * t;
        }
        if (!terminal) {
            for (int i = 0; i < 9; ++i) {
                const double wq = 2 * c->Q[i];
                const double* Gi = G + (size_t)i * n;
                for (int a = 0; a < cols; ++a) {
                    const double t = wq * Gi[a];
                    double* Hr = H + (size_t)a * n;
                    for (int b = 0; b <= a; ++b) Hr[b] += t * Gi[b];
                }
            }
        } else {
            for (int i = 0; i < 9; ++i)
                for (int j = 0; j < 9; ++j) {
                    const double wp = 2 * c->P[9 * i + j];
                    if (wp == 0) continue;
                    const double *Gi = G + (size_t)i * n, *Gj = G + (size_t)j * n;
                    for (int a = 0; a < cols; ++a) {
                        const double t = wp * Gi[a];
                        double* Hr = H + (size_t)a * n;
                        for (int b = 0; b <= a; ++b) Hr[b] += t * Gj[b];
                    }
                }
        }
        /* input term of stage k */
        for (int a = 0; a < na; ++a) {
            double t = 0;
            for (int gg = 0; gg < 6; ++gg) t += c->D[gg * NT + W->act[a]] * c->R[gg] * ut[gg];
            g[used + a] += 2 * (t + c->rho * W->ubar[k * NT + W->act[a]]);
            for (int b = 0; b <= a; ++b) H[(size_t)(used + a) * n + used + b] += 2 * MR[a * na + b];
        }
    }
    for (int a = 0; a < n; ++a)
        for (int b = a + 1; b < n; ++b) H[(size_t)a * n + b] = H[(size_t)b * n + a];
    for (int k = 0; k < N; ++k)
        for (int a = 0; a < na; ++a) {
            const double u = W->ubar[k * NT + W->act[a]];
            W->lo[k * na + a] = -u;
            W->hi[k * na + a] = ub[W->act[a]] - u;
        }
    return n;
}

/* ------------------------------------------------------------------ IPM (oracle/qp_oracle.py ipm_box) */
static int chol(double* M, int n) { /* lower, in place */
    for (int j = 0; j < n; ++j) {
        double d = M[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= M[(size_t)j * n + k] * M[(size_t)j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        M[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = M[(size_t)i * n + j];
            const double *ri = M + (size_t)i * n, *rj = M + (size_t)j * n;
            for (int k = 0; k < j; ++k) t -= ri[k] * rj[k];
            M[(size_t)i * n + j] = t / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, double* x) {
    for (int i = 0; i < n; ++i) {
        double t = x[i];
        const double* r = L + (size_t)i * n;
        for (int k = 0; k < i; ++k) t -= r[k] * x[k];
        x[i] = t / r[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double t = x[i];
        for (int k = i + 1; k < n; ++k) t -= L[(size_t)k * n + i] * x[k];
        x[i] = t / L[(size_t)i * n + i];
    }
}

/* Active-set polish of a box QP from an interior-point iterate (the box-row case of oracle/qp_oracle.py:polish_general; what
 * ftmpc_solve_ric.hip and ftmpc_solve.hip run on the device): the bounds with z > s are taken as active and the problem on that set
 * is solved by two method-of-multipliers steps with the penalty W = 1e6 max diag(H) on the active bounds,
 *     (H + Sigma_A) dd = -(H d + g) + C_A' (W s_A - lam),   lam += W (C_A dd - s_A),
 * then the signs are verified: an active bound with a negative multiplier leaves, an inactive bound that is violated enters; at
 * most three rounds.  Returns the rounds run (> 0) with the exact solution on the verified set in d, or 0 (d untouched). */
static int polish_box(const double* H, const double* g, const double* lo, const double* hi, int n, const double* sl, const double* su,
                      const double* zl, const double* zu, double* d, double* M, double* w6) {
    double *psl = w6, *psu = w6 + n, *pzl = w6 + 2 * n, *pzu = w6 + 3 * n, *r = w6 + 4 * n, *act = w6 + 5 * n;
    double hs = 0;
    for (int i = 0; i < n; ++i) {
        if (H[(size_t)i * n + i] > hs) hs = H[(size_t)i * n + i];
        const int al = zl[i] > sl[i], au = zu[i] > su[i];
        act[i] = (double)(al + 2 * au);
        psl[i] = sl[i];
        psu[i] = su[i];
        pzl[i] = al ? zl[i] : 0.0;
        pzu[i] = au ? zu[i] : 0.0;
    }
    const double pw = 1e6 * hs;
    for (int rd = 1; rd <= 3; ++rd) {
        memcpy(M, H, (size_t)n * n * sizeof(double));
        for (int i = 0; i < n; ++i) {
            const int a = (int)act[i];
            M[(size_t)i * n + i] += ((a & 1) ? pw : 0.0) + ((a & 2) ? pw : 0.0);
        }
        if (chol(M, n) != 0) return 0;
        for (int in = 0; in < 2; ++in) {
            for (int i = 0; i < n; ++i) {
                double t = g[i];      /* the gradient H d + g at the polish's own iterate, d read back from the nearer slack */
                const double* row = H + (size_t)i * n;
                for (int k = 0; k < n; ++k) t += row[k] * ((psl[k] < psu[k]) ? lo[k] + psl[k] : hi[k] - psu[k]);
                const int a = (int)act[i];
                r[i] = -t - ((a & 1) ? pw * psl[i] - pzl[i] : 0.0) + ((a & 2) ? pw * psu[i] - pzu[i] : 0.0);
            }
            chol_solve(M, n, r);
            for (int i = 0; i < n; ++i) {
                const int a = (int)act[i];
                if (a & 1) pzl[i] += pw * (-r[i] - psl[i]);
                if (a & 2) pzu[i] += pw * (r[i] - psu[i]);
                psl[i] += r[i];
                psu[i] -= r[i];
            }
        }
        int changed = 0;
        for (int i = 0; i < n; ++i) {
            int a = (int)act[i];
            if (a & 1) {
                if (pzl[i] < 0) { pzl[i] = 0; a &= ~1; changed = 1; }
            } else if (psl[i] < -1e-10) { a |= 1; changed = 1; }
            if (a & 2) {
                if (pzu[i] < 0) { pzu[i] = 0; a &= ~2; changed = 1; }
            } else if (psu[i] < -1e-10) { a |= 2; changed = 1; }
            act[i] = (double)a;
        }
        if (!changed) {
            for (int i = 0; i < n; ++i) {
                const double di = (psl[i] < psu[i]) ? lo[i] + psl[i] : hi[i] - psu[i];
                d[i] = di < lo[i] ? lo[i] : (di > hi[i] ? hi[i] : di);
            }
            return rd;
        }
    }
    return 0;
}

/* returns iterations (polish rounds included); status: 0 converged, 1 maxiter, 2 numeric.  polish: once mu < 1e-7 the active-set
 * polish above is tried (once); verified -> done with the exact solution, else the iteration runs on to mu_stop. */
static int ipm_box(const double* H, const double* g, const double* lo, const double* hi, int n, int max_iters,
                   double mu_stop, int polish, double* d, double* M, double* wk, int* status) {
    double *sl = wk, *su = wk + n, *zl = wk + 2 * n, *zu = wk + 3 * n, *grad = wk + 4 * n, *Sig = wk + 5 * n,
           *da = wk + 6 * n, *dd = wk + 7 * n, *dzla = wk + 8 * n, *dzua = wk + 9 * n, *rcl = wk + 10 * n,
           *rcu = wk + 11 * n, *rhs = wk + 12 * n;
    double gm = 0, wm = 0;
    for (int i = 0; i < n; ++i) {
        d[i] = 0.5 * (lo[i] + hi[i]);
        sl[i] = d[i] - lo[i];
        su[i] = hi[i] - d[i];
    }
    int nit = 0;
    *status = 1;
    for (int it = 0; it <= max_iters; ++it) {
        for (int i = 0; i < n; ++i) {
            double t = g[i];
            const double* r = H + (size_t)i * n;
            for (int k = 0; k < n; ++k) t += r[k] * d[k];
            grad[i] = t;
        }
        if (it == 0) {
            for (int i = 0; i < n; ++i) {
                if (fabs(grad[i]) > gm) gm = fabs(grad[i]);
                if (hi[i] - lo[i] > wm) wm = hi[i] - lo[i];
            }
            double mu0 = 0.02 * gm * wm;
            if (mu0 < 1e-3) mu0 = 1e-3;
            for (int i = 0; i < n; ++i) {
                zl[i] = mu0 / sl[i];
                zu[i] = mu0 / su[i];
            }
        }
        double mu = 0;
        for (int i = 0; i < n; ++i) mu += sl[i] * zl[i] + su[i] * zu[i];
        mu /= 2.0 * n;
        if (!(mu >= mu_stop)) {
            *status = (mu == mu) ? 0 : 2;
            break;
        }
        if (it == max_iters) break;
        if (polish && mu < 1e-7) {
            polish = 0;
            const int rounds = polish_box(H, g, lo, hi, n, sl, su, zl, zu, d, M, da);      /* (da .. rhs: six vectors free here) */
            if (rounds > 0) {
                nit += rounds;
                *status = 0;
                break;
            }
        }
        ++nit;
        memcpy(M, H, (size_t)n * n * sizeof(double));
        for (int i = 0; i < n; ++i) {
            Sig[i] = zl[i] / sl[i] + zu[i] / su[i];
            M[(size_t)i * n + i] += Sig[i];
        }
        if (chol(M, n) != 0) {
            *status = 2;
            break;
        }
        double ap = 1, ad = 1;
        for (int i = 0; i < n; ++i) da[i] = -grad[i];
        chol_solve(M, n, da);
        for (int i = 0; i < n; ++i) {
            dzla[i] = -zl[i] - zl[i] * da[i] / sl[i];
            dzua[i] = -zu[i] + zu[i] * da[i] / su[i];
            if (da[i] < 0 && -sl[i] / da[i] < ap) ap = -sl[i] / da[i];
            if (da[i] > 0 && su[i] / da[i] < ap) ap = su[i] / da[i];
            if (dzla[i] < 0 && -zl[i] / dzla[i] < ad) ad = -zl[i] / dzla[i];
            if (dzua[i] < 0 && -zu[i] / dzua[i] < ad) ad = -zu[i] / dzua[i];
        }
        double mua = 0;
        for (int i = 0; i < n; ++i) mua += (sl[i] + ap * da[i]) * (zl[i] + ad * dzla[i]) + (su[i] - ap * da[i]) * (zu[i] + ad * dzua[i]);
        mua /= 2.0 * n;
        double sigma = mua / mu;
        sigma = sigma * sigma * sigma;
        sigma = sigma < 0 ? 0 : (sigma > 1 ? 1 : sigma);
        for (int i = 0; i < n; ++i) {
            rcl[i] = sl[i] * zl[i] + da[i] * dzla[i] - sigma * mu;
            rcu[i] = su[i] * zu[i] - da[i] * dzua[i] - sigma * mu;
            rhs[i] = -(grad[i] - zl[i] + zu[i]) - rcl[i] / sl[i] + rcu[i] / su[i];
            dd[i] = rhs[i];
        }
        chol_solve(M, n, dd);
        ap = 1e30;
        ad = 1e30;
        for (int i = 0; i < n; ++i) {
            const double dzl = (-rcl[i] - zl[i] * dd[i]) / sl[i], dzu = (-rcu[i] + zu[i] * dd[i]) / su[i];
            dzla[i] = dzl;
            dzua[i] = dzu;
            if (dd[i] < 0 && -sl[i] / dd[i] < ap) ap = -sl[i] / dd[i];
            if (dd[i] > 0 && su[i] / dd[i] < ap) ap = su[i] / dd[i];
            if (dzl < 0 && -zl[i] / dzl < ad) ad = -zl[i] / dzl;
            if (dzu < 0 && -zu[i] / dzu < ad) ad = -zu[i] / dzu;
        }
        ap = 0.9995 * ap; if (ap > 1) ap = 1;
        ad = 0.9995 * ad; if (ad > 1) ad = 1;
        for (int i = 0; i < n; ++i) {
            sl[i] += ap * dd[i];
            su[i] -= ap * dd[i];
            zl[i] += ad * dzla[i];
            zu[i] += ad * dzua[i];
            d[i] = (sl[i] < su[i]) ? lo[i] + sl[i] : hi[i] - su[i];
        }
    }
    return nit;
}

/* ------------------------------------------------------------------ batch driver */
typedef struct {
    const oracle_config* cfg;
    int64_t b0, b1;
    const double *x0, *ub, *stuck, *xref, *uref, *warm;
    int64_t xs, us;
    double *u0, *U;
    int32_t *status, *iters;
    int rc;
} job;

static void* run_job(void* arg) {
    job* j = (job*)arg;
    const oracle_config* c = j->cfg;
    const int N = c->N, NT = c->NT, nm = N * NT;
    qp_work W;
    W.H = (double*)malloc((size_t)nm * nm * sizeof(double));
    double* M = (double*)malloc((size_t)nm * nm * sizeof(double));
    W.g = (double*)malloc((size_t)nm * 4 * sizeof(double));
    W.lo = W.g + nm; W.hi = W.g + 2 * nm; W.ubar = W.g + 3 * nm;
    W.G = (double*)malloc((size_t)13 * nm * 2 * sizeof(double));
    W.Gn = W.G + (size_t)13 * nm;
    double* wk = (double*)malloc((size_t)nm * 14 * sizeof(double));
    double* d = wk + (size_t)13 * nm;
    if (!W.H || !M || !W.g || !W.G || !wk) { j->rc = -1; return 0; }
    for (int64_t b = j->b0; b < j->b1; ++b) {
        const double* ub = j->ub + b * NT;
        const int n = build_qp(c, j->x0 + b * 13, ub, j->stuck + b * NT, j->xref + b * j->xs,
                               j->uref ? j->uref + b * j->us : 0, j->warm ? j->warm + b * nm : 0, &W);
        int st = 0, nit = 0;
        if (n > 0) nit = ipm_box(W.H, W.g, W.lo, W.hi, n, c->max_iters, c->mu_stop, !(c->flags & 1), d, M, wk, &st);
        const int na = n / N;
        for (int i = 0; i < NT; ++i) j->u0[b * NT + i] = 0;
        if (j->U) memset(j->U + b * nm, 0, (size_t)nm * sizeof(double));
        for (int k = 0; k < N; ++k)
            for (int a = 0; a < na; ++a) {
                const double u = W.ubar[k * NT + W.act[a]] + (st == 2 ? 0.0 : d[k * na + a]);
                if (k == 0) j->u0[b * NT + W.act[a]] = u;
                if (j->U) j->U[b * nm + k * NT + W.act[a]] = u;
            }
        if (j->status) j->status[b] = st;
        if (j->iters) j->iters[b] = nit;
    }
    free(W.H); free(M); free(W.g); free(W.G); free(wk);
    j->rc = 0;
    return 0;
}

int ftmpc_oracle_solve_batch(const oracle_config* cfg, int64_t B, const double* x0, const double* ub,
                             const double* stuck, const double* xref, int64_t xref_stride, const double* uref,
                             int64_t uref_stride, const double* warmU, double* out_u0, double* out_U,
                             int32_t* status, int32_t* iters, int32_t nthreads) {
    if (!cfg || B < 0 || cfg->NT > MAX_NT) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads > B) nthreads = (int32_t)(B > 0 ? B : 1);
    job jobs[256];
    pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) {
        job* j = &jobs[t];
        j->cfg = cfg;
        j->b0 = B * t / nthreads;
        j->b1 = B * (t + 1) / nthreads;
        j->x0 = x0; j->ub = ub; j->stuck = stuck; j->xref = xref; j->uref = uref; j->warm = warmU;
        j->xs = xref_stride; j->us = uref_stride;
        j->u0 = out_u0; j->U = out_U; j->status = status; j->iters = iters;
        j->rc = 0;
        if (t > 0) pthread_create(&th[t], 0, run_job, j);
    }
    run_job(&jobs[0]);
    int rc = jobs[0].rc;
    for (int t = 1; t < nthreads; ++t) {
        pthread_join(th[t], 0);
        if (jobs[t].rc) rc = jobs[t].rc;
    }
    return rc;
}

/* test hook: the dense QP of one instance (H n*n row-major, g, lo, hi); returns n */
int ftmpc_oracle_build_qp(const oracle_config* cfg, const double* x0, const double* ub, const double* stuck,
                          const double* xref, const double* uref, const double* warm, double* H, double* g,
                          double* lo, double* hi) {
    const int nm = cfg->N * cfg->NT;
    qp_work W;
    W.H = H;
    W.g = g;
    W.lo = lo;
    W.hi = hi;
    W.ubar = (double*)malloc((size_t)nm * sizeof(double));
    W.G = (double*)malloc((size_t)13 * nm * 2 * sizeof(double));
    W.Gn = W.G + (size_t)13 * nm;
    const int n = build_qp(cfg, x0, ub, stuck, xref, uref, warm, &W);
    free(W.ubar);
    free(W.G);
    return n;
}

/* one plant step in thruster space (sys_model.py:177-226 through rk4, :152-158): used by the
 * closed-loop plumbing test as an independent restatement of SystemModel.dynamics */
int ftmpc_oracle_plant_step(const oracle_config* cfg, const double* x, const double* u, const double* ub,
                            const double* stuck, double* xn) {
    double Jinv[9], gen[6] = {0, 0, 0, 0, 0, 0};
    if (inv3(cfg->J, Jinv)) return -1;
    for (int i = 0; i < cfg->NT; ++i) {
        const double t = (ub[i] > 0 ? u[i] : 0.0) + stuck[i];
        for (int g = 0; g < 6; ++g) gen[g] += cfg->D[g * cfg->NT + i] * t;
    }
    double k[4][13], s[13];
    const double dt = cfg->dt, a[4] = {0, dt / 2, dt / 2, dt};
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 13; ++j) s[j] = x[j] + (i ? a[i] * k[i - 1][j] : 0.0);
        const double *v = s + 3, *q = s + 6, *w = s + 10;
        double Jw[3], wJw[3], t[3], f[3];
        for (int j = 0; j < 3; ++j) {
            k[i][j] = v[j];
            f[j] = gen[j] / cfg->mass;
        }
        rotT_mul(q, f, k[i] + 3);
        k[i][6] = 0.5 * (w[2] * q[1] - w[1] * q[2] + w[0] * q[3]);
        k[i][7] = 0.5 * (-w[2] * q[0] + w[0] * q[2] + w[1] * q[3]);
        k[i][8] = 0.5 * (w[1] * q[0] - w[0] * q[1] + w[2] * q[3]);
        k[i][9] = 0.5 * (-w[0] * q[0] - w[1] * q[1] - w[2] * q[2]);
        mv3(cfg->J, w, Jw);
        cross3(w, Jw, wJw);
        for (int j = 0; j < 3; ++j) t[j] = gen[3 + j] - wJw[j];
        mv3(Jinv, t, k[i] + 10);
    }
    for (int j = 0; j < 13; ++j) xn[j] = x[j] + dt / 6 * (k[0][j] + 2 * k[1][j] + 2 * k[2][j] + k[3][j]);
    return 0;
}
