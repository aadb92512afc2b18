"""Study (numpy, fp32 emulation): leave the box-QP interior-point iteration early (mu < thr) and finish by an active-set polish with a
diagonal penalty -- iterations saved, rounds needed, distance from the exact solution.  Headline shape by default."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import qp_oracle as qo

T = np.float32


def run(H, g, lo, hi, thr, pw0, mu_stop=1e-11, iters=40):
    n = g.size
    H32 = H.astype(T); H64 = H
    sl = ((hi - lo) * 0.5).astype(T); su = sl.copy()
    lo = lo.astype(T); hi = hi.astype(T)
    zl = zu = None
    nit = 0
    tried = False
    hs = T(np.diag(H).max())
    for it in range(iters + 1):
        d = np.where(sl < su, lo + sl, hi - su)
        grad = (H64 @ d.astype(np.float64) + g).astype(T)
        if zl is None:
            mu0 = max(T(np.abs(grad).max()) * T((hi - lo).max()) * T(0.02), T(1e-3))
            zl = mu0 / sl; zu = mu0 / su
        mu = (sl @ zl + su @ zu) / T(2 * n)
        if not (mu >= T(mu_stop)):
            break
        if not tried and mu < thr:
            tried = True
            al = zl > sl; au = zu > su
            psl, psu, pzl, pzu, pg = sl.copy(), su.copy(), np.where(al, zl, T(0)), np.where(au, zu, T(0)), grad.copy()
            pw = T(pw0) * hs
            ok = False
            rounds = 0
            for rd in range(3):
                Sig = (al * pw + au * pw).astype(T)
                L = qo._chol(H32 + np.diag(Sig))
                rounds += 1
                for inner in range(2):
                    rhs = (-pg - np.where(al, pw * psl - pzl, T(0)) + np.where(au, pw * psu - pzu, T(0))).astype(T)
                    dd = qo._solve(L, rhs)
                    pg = (pg + rhs - Sig * dd).astype(T)
                    pzl = np.where(al, pzl + pw * (-dd - psl), T(0)).astype(T)
                    pzu = np.where(au, pzu + pw * (dd - psu), T(0)).astype(T)
                    psl = psl + dd; psu = psu - dd
                tolz = T(0)
                leave_l = al & (pzl < -tolz); leave_u = au & (pzu < -tolz)
                enter_l = ~al & (psl < -1e-6 * (hi - lo)); enter_u = ~au & (psu < -1e-6 * (hi - lo))
                if not (leave_l.any() or leave_u.any() or enter_l.any() or enter_u.any()):
                    ok = True
                    break
                al = (al & ~leave_l) | enter_l; au = (au & ~leave_u) | enter_u
                pzl = np.where(al, np.maximum(pzl, 0), T(0)); pzu = np.where(au, np.maximum(pzu, 0), T(0))
            nit += rounds
            if ok:
                d = np.clip(np.where(psl < psu, lo + psl, hi - psu), lo, hi)
                return d, nit, rounds, True
        nit += 1
        Sig = zl / sl + zu / su
        L = qo._chol(H32 + np.diag(Sig))
        da = qo._solve(L, -grad)
        dzl_a = -zl - zl * da / sl; dzu_a = -zu + zu * da / su
        ap = min(T(1), qo._max_step(sl, da, su)); ad = min(T(1), qo._max_step_dual(zl, dzl_a, zu, dzu_a))
        mu_aff = ((sl + ap * da) @ (zl + ad * dzl_a) + (su - ap * da) @ (zu + ad * dzu_a)) / T(2 * n)
        sigma = min(max((mu_aff / mu) ** 3, T(0)), T(1))
        rcl = sl * zl + da * dzl_a - sigma * mu; rcu = su * zu - da * dzu_a - sigma * mu
        rhs = -(grad - zl + zu) - rcl / sl + rcu / su
        dd = qo._solve(L, rhs)
        dzl = (-rcl - zl * dd) / sl; dzu = (-rcu + zu * dd) / su
        ap = min(T(1), T(0.9995) * qo._max_step(sl, dd, su)); ad = min(T(1), T(0.9995) * qo._max_step_dual(zl, dzl, zu, dzu))
        sl = sl + ap * dd; su = su - ap * dd; zl = zl + ad * dzl; zu = zu + ad * dzu
    d = np.where(sl < su, lo + sl, hi - su)
    return d, nit, 0, False


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    N, NT, nf = 20, 8, 2
    cfg = qo.QPConfig(N=N, NT=NT)
    x0, ub, stuck, xref = qo.make_batch(B, N, NT, nf, 4242)
    import time
    res = {}
    cases = [("1e-4 w1e3", 1e-4, 1e3), ("3e-5 w1e3", 3e-5, 1e3), ("1e-5 w1e3", 1e-5, 1e3), ("1e-5 w3e2", 1e-5, 3e2), ("3e-6 w1e3", 3e-6, 1e3)]
    for name, thr, pw in cases:
        res[name] = []
    for b in range(B):
        q = qo.build_qp(cfg, x0[b], ub[b], stuck[b], xref)
        lo = -q["Ubar"]; hi = q["ub"] - q["Ubar"]
        dex = qo.solve_exact(q["H"], q["g"], lo, hi)
        for name, thr, pw in cases:
            d, nit, rounds, ok = run(q["H"], q["g"], lo, hi, thr if thr else -1.0, pw)
            res[name].append((nit, rounds, ok, np.abs(d - dex).max() / 1.4, np.abs(d[:q["na"]] - dex[:q["na"]]).max() / 1.4))
    for name, _, _ in cases:
        r = np.array(res[name], float)
        print(f"{name:12s} passes {r[:,0].mean():.2f} (max {r[:,0].max():.0f})  rounds {r[:,1].mean():.2f}  verified {r[:,2].mean():.3f}  "
              f"err U max {r[:,3].max():.2e} med {np.median(r[:,3]):.2e}  err u0 max {r[:,4].max():.2e}")
