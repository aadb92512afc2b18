"""Mirror of ft_mpc/util/utils.py (numpy only): quaternion [x,y,z,w] -> rotation matrices.
Rot is the world->body matrix (utils.py:15-19); RotInv = Rot^T maps body->world."""
import numpy as np


def Rot(q):
    x, y, z, w = np.asarray(q, dtype=float).reshape(-1)[:4]
    xx, yy, zz, ww = x * x, y * y, z * z, w * w
    return np.array([
        [xx - yy - zz + ww, 2 * (x * y + z * w), 2 * (x * z - y * w)],
        [2 * (x * y - z * w), -xx + yy - zz + ww, 2 * (y * z + x * w)],
        [2 * (x * z + y * w), 2 * (y * z - x * w), -xx - yy + zz + ww],
    ])


def RotInv(q):
    return Rot(q).T


def RotFull(q):
    M = np.eye(6)
    M[:3, :3] = Rot(q)
    return M


def RotFullInv(q):
    return RotFull(q).T
