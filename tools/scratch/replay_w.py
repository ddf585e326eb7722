"""replay one trial of the soak's file family and keep its files (gpurun_out/replay_w/)"""
import os, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fz
seed, trial, fams = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
keep = os.path.join(ROOT, "gpurun_out", "replay_w")
orig_rmtree = shutil.rmtree
def keep_tree(path, *a, **k):
    if os.path.basename(path).startswith("x3fz"):
        shutil.copytree(path, keep, dirs_exist_ok=True)
    return orig_rmtree(path, *a, **k)
shutil.rmtree = keep_tree
import x3hip
ctx = x3hip.Context(0)
try:
    fz.run(seed, None, 1, fams, trial, context=ctx)
    print("trial passed")
except AssertionError as e:
    print("trial failed:", e)
print("options: file_chunk_frames", ctx.get_option("file_chunk_frames"), "file_workers", ctx.get_option("file_workers"))
