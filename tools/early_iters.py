"""Developer probe: cost of the first ICP iterations on C3 (no start leaves yet, no chunk plan yet)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
scene, model = synth.config_clouds("C3")
ctx = ope.Context(0)
cs = ctx.upload(scene); ix = ctx.build_index(ctx.upload(model))
for rep in range(2):
    ctx.icp_begin(cs, ix, ope.default_icp_params(max_iterations=100, mse_threshold_absolute=-1.0, check_every=0), None)
    ts = []
    for it in range(12):
        ctx.sync(); t0 = time.perf_counter(); ctx.icp_iterate(1); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    ctx.icp_end()
    print("ms per iteration: " + " ".join(f"{t:.3f}" for t in ts) + f"   first 12: {sum(ts):.2f} ms", flush=True)
ctx.close()
