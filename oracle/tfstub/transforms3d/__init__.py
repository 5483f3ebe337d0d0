"""Minimal `transforms3d` stand-in used ONLY by oracle/gen_goldens.py (TEST INFRASTRUCTURE).

The reference's registration harness imports transforms3d (third party, unpinned in the reference: `import transforms3d.euler
as t3d`, pcrnet-registration/helper.py:6, results_itrPCRNet_no_stop.py) which is not installable here.  The four functions its
call sites use are restated from the library's published conventions (transforms3d 0.3/0.4 docs):
    euler.euler2mat(ai, aj, ak, axes)   'szyx': static frame, rotate about z by ai, then y by aj, then x by ak
                                          ->  R = Rx(ak) @ Ry(aj) @ Rz(ai)
    euler.mat2euler(M, axes)            inverse of the above for 'szyx' (returns ai, aj, ak)
    euler.quat2mat / quaternions.quat2mat(q)   q = (w, x, y, z), normalised inside
    axangles.mat2axangle(M)             (axis, angle) with angle = atan2-form of acos((trace - 1) / 2)
tests/test_registration.py cross-checks every one of them against scipy.spatial.transform.Rotation.
"""
from . import axangles, euler  # noqa: F401
