"""Developer probe: OPE_HEAVY_FACTOR (environment) on mid-size launches."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
model = synth.model_surface(100_000, 1)
big = synth.scene_cloud(2_000_000)
ctx = ope.Context(0)
ix = ctx.build_index(ctx.upload(model))
out = []
for n in (250_000, 390_000, 500_000, 750_000, 1_500_000):
    cs = ctx.upload(big[:n])
    kw = dict(max_iterations=100, mse_threshold_absolute=-1.0, check_every=0)
    ctx.icp(cs, ix, ope.default_icp_params(**{**kw, "max_iterations": 40}))
    t0 = time.perf_counter(); ctx.icp(cs, ix, ope.default_icp_params(**kw)); dt = time.perf_counter() - t0
    out.append(f"{n//1000}k: {dt/100*1e6:.0f}")
    cs.free()
print(f"factor {os.environ.get('OPE_HEAVY_FACTOR','default')}: " + "  ".join(out), flush=True)
ctx.close()
