"""Build libdpdist_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m dpdist_amd.build        # or: from dpdist_amd.build import build; build()

The library has no torch dependency: plain HIP C++ behind the C ABI of include/dpdist_capi.h.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdpdist_hip.so")
SOURCES = ["gemm_f32.hip", "gemm_x3.hip", "gemm_p8.hip", "mfv3d.hip", "patch_rows.hip", "decoder.hip", "loss_adam.hip", "chamfer.hip", "asloss.hip", "pose.hip", "host_util.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dpdist_capi.h"),
                                                                 os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    # DPD_ABLATIONS=1: also compile the timing-only ablation variants of the plane GEMM (tools/x3_bench.py tile codes 100+)
    flags = FLAGS + (["-DDPD_ABLATIONS"] if os.environ.get("DPD_ABLATIONS") == "1" else []) + os.environ.get("DPD_EXTRA_FLAGS", "").split()
    stamp0 = os.path.join(HERE, "build", "flags.txt")
    flags_changed = os.path.exists(stamp0) and open(stamp0).read() != " ".join(flags)
    if not force and not flags_changed and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # an object is reused when it is newer than its source, every header and this file, and was built with the same flags
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "dpdist_capi.h"),
                                                                                       os.path.abspath(__file__)]
    stamp = os.path.join(HERE, "build", "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and same_flags and os.path.exists(obj) and all(os.path.getmtime(d) < os.path.getmtime(obj)
                                                                    for d in headers + [os.path.join(CSRC, src)]):
            continue
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
