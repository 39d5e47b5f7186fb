"""ope_index_build wall time (host clock around the call, which synchronises) for clouds of several sizes, best of 5.
    python tools/index_probe.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
ctx = ope.Context(0)
for n in (800, 2000, 20_000, 100_000, 500_000, 1_000_000, 4_000_000):
    c = ctx.upload(synth.model_surface(n, 1))
    best = 1e9
    for rep in range(6):
        ctx.sync(); t0 = time.perf_counter()
        ix = ctx.build_index(c, grid=0)
        dt = time.perf_counter() - t0
        ix.free()
        if rep: best = min(best, dt)
    print(f"index over {n:>8} points: {best * 1e3:7.3f} ms")
