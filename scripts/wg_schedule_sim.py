#!/usr/bin/env python3
"""Helper work per phase of kernel 7's factorisation (MFMAs over the three helper waves) under the shipped left-looking
schedule (Q = 0) and under a look-ahead rule that also advances later columns over the block columns already final,
up to Q MFMAs per phase.  Result: look-ahead does not lower the peak (nb = 14: 216 -> 208 at best, mean 156), so the
kernel keeps the simple schedule.  Usage: python scripts/wg_schedule_sim.py"""


def helper_mfmas(nb, Q):
    kd = {c: 0 for c in range(nb)}           # group c = tiles (I, c), I > c, and diagonal tile c+1: terms K < kd[c] done
    rows = []
    for J in range(nb - 1):
        work = max(nb - J - 2, 0) * 8        # panel solves of column J: last term + solve, 8 MFMAs per tile
        for c in range(J + 1, nb):
            nt = (nb - c - 1) + (1 if c + 1 < nb else 0)
            if nt == 0:
                continue
            avail = J - kd[c]
            steps = avail if c == J + 1 else min(avail, max(0, (Q - work) // (nt * 4)))
            work += nt * steps * 4
            kd[c] += steps
        rows.append(work)
    return rows


if __name__ == "__main__":
    for nb in (8, 11, 14, 15):
        for Q in (0, 165, 200):
            r = helper_mfmas(nb, Q)
            print(f"nb={nb:2d} Q={Q:3d} total={sum(r):5d} peak={max(r):4d} mean={sum(r) / max(1, len(r)):6.1f}  {r}")
