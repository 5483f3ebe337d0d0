import math

import numpy as np


def _rx(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _ry(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rz(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


_AX = {"x": _rx, "y": _ry, "z": _rz}


def euler2mat(ai, aj, ak, axes="sxyz"):
    """static ('s') axes: rotate about axes[1] by ai, then axes[2] by aj, then axes[3] by ak (fixed frame)."""
    if axes[0] != "s":
        raise NotImplementedError("only static-frame axes are restated")
    return _AX[axes[3]](ak) @ _AX[axes[2]](aj) @ _AX[axes[1]](ai)


def mat2euler(M, axes="sxyz"):
    M = np.asarray(M, dtype=np.float64)
    if axes == "szyx":       # R = Rx(ak) Ry(aj) Rz(ai):  R[0,2] = sin(aj), R[0,1] = -cos(aj) sin(ai), R[0,0] = cos(aj) cos(ai)
        cy = math.hypot(M[0, 0], M[0, 1])
        if cy > 4 * np.finfo(float).eps:
            return math.atan2(-M[0, 1], M[0, 0]), math.atan2(M[0, 2], cy), math.atan2(-M[1, 2], M[2, 2])
        return 0.0, math.atan2(M[0, 2], cy), math.atan2(M[2, 1], M[1, 1])
    if axes == "sxyz":       # R = Rz(ak) Ry(aj) Rx(ai)
        cy = math.hypot(M[0, 0], M[1, 0])
        if cy > 4 * np.finfo(float).eps:
            return math.atan2(M[2, 1], M[2, 2]), math.atan2(-M[2, 0], cy), math.atan2(M[1, 0], M[0, 0])
        return math.atan2(-M[1, 2], M[1, 1]), math.atan2(-M[2, 0], cy), 0.0
    raise NotImplementedError(axes)


def quat2mat(q):
    w, x, y, z = [float(v) for v in q]
    n = w * w + x * x + y * y + z * z
    if n < np.finfo(float).eps:
        return np.eye(3)
    s = 2.0 / n
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY],
                     [xY + wZ, 1.0 - (xX + zZ), yZ - wX],
                     [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])
