"""Import-only stand-in for the reference's compiled `pc_distance` package (CUDA ops tf_nndistance / tf_approxmatch, absent
from /root/reference): lets `models/ipcr_model.py` be imported by oracle/gen_goldens.py.  Every entry point raises: the
Chamfer/EMD baselines built on those ops are out of scope and no golden uses them.  TEST INFRASTRUCTURE."""


class _Missing:
    def __init__(self, name):
        self._name = name

    def __getattr__(self, item):
        raise NotImplementedError("%s.%s: the reference's CUDA op is not in the tree" % (self._name, item))


tf_nndistance = _Missing("tf_nndistance")
tf_approxmatch = _Missing("tf_approxmatch")
