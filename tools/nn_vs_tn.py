import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools")
sys.argv = ["x3_bench.py"]
os.environ["QUICK"] = "1"
import importlib.util
spec = importlib.util.spec_from_file_location("x3b", os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/tools/x3_bench.py")
m = importlib.util.module_from_spec(spec)
# avoid running its __main__ block
src = open(spec.origin).read().split('if __name__ == "__main__":')[0]
exec(compile(src, spec.origin, "exec"), m.__dict__)
for tile in (21, 20, 8, 2):
    for mode in ("NN", "TN"):
        m.run(mode, 8192, 1024, 2528 if tile < 8 or tile >= 20 else 2560, 1, tile, iters=50)
    for mode in ("NN", "TN", "NT"):
        m.run(mode, 8192, 1024, 1024, 1, tile, iters=50)
    for mode in ("NT", "TN"):
        m.run(mode, 4096, 1024, 1024, 1, tile, iters=50)
