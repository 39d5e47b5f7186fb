"""Developer probe: ICP on one 1/W slice of the C3 scene (what one rank of a W-GPU run executes, minus the
all-reduce).  Run under rocprofv3 --kernel-trace --stats to see the per-kernel times at that shard size."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
sharded = importlib.import_module("object-pose-estimation_amd.sharded")
scene, model = synth.config_clouds("C3")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = ope.Context(0)
ix = ctx.build_index(ctx.upload(model))
lo, hi = sharded.shard_range(len(scene), W, 0)
pts = scene[lo:hi]
if len(sys.argv) > 2 and sys.argv[2] == "inliers":      # floor: every query sits on the model surface
    pts = synth.model_surface(hi - lo, 7)
cs = ctx.upload(pts)
kw = dict(max_iterations=200, mse_threshold_absolute=-1.0, check_every=0)
ctx.icp(cs, ix, ope.default_icp_params(**{**kw, "max_iterations": 40}))
t0 = time.perf_counter(); ctx.icp(cs, ix, ope.default_icp_params(**kw)); dt = time.perf_counter() - t0
print(f"shard {hi - lo} pts: {dt / 200 * 1e6:.1f} us/iteration", flush=True)
ctx.close()
