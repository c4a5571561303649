import cProfile, pstats, sys, os, io
sys.argv = ["bench_train.py", "--steps", "6", "--warmup", "3"]
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import runpy
pr = cProfile.Profile()
pr.enable()
runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "bench_train.py"), run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats("train|_ffi|ctypes|detector", 45)
print(s.getvalue()[:9000])
