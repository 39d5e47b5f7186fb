import importlib, os, sys, time, subprocess
for f in ("0", "1.2", "1.5", "2", "3", "4"):
    env = dict(os.environ, OPE_HEAVY_FACTOR=f)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "quick_bench.py")], env=env, capture_output=True, text=True).stdout.splitlines()
    print("factor", f, "|", [l.split("->")[1].split("mse")[0].strip() for l in out if "leaf=16" in l])
