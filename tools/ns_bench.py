"""Developer probe: estimateFinePose's configuration on C3 — normal shooting k = 20 + surface-normal rejector."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
src, tgt = synth.config_clouds(name)
ctx = ope.Context(0)
cs = ctx.upload(src); ct = ctx.upload(tgt)
t0 = time.perf_counter(); ctx.normals(cs, 30, fetch=False); ctx.normals(ct, 30, fetch=False); ctx.sync(); t1 = time.perf_counter()
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else None
ix = ctx.build_index(ct, leaf_size=leaf)
print(f"{name}: normals k=30 on both clouds {1e3*(t1-t0):.1f} ms; index leaf size {leaf or 'default'}", flush=True)
for k in (20, 10):
    kw = dict(max_iterations=50, mse_threshold_absolute=-1.0, check_every=0, corr_mode=ope.CORR_NORMAL_SHOOTING, k_normal_shooting=k,
              use_surface_normal_rej=1, surface_normal_thr=0.7)
    ctx.icp(cs, ix, ope.default_icp_params(**{**kw, "max_iterations": 3}))
    t0 = time.perf_counter(); out = ctx.icp(cs, ix, ope.default_icp_params(**kw)); dt = time.perf_counter() - t0
    print(f"   normal shooting k={k} + surface-normal rejector: {dt/50*1e3:.3f} ms/iteration  n_corr={out.n_corr}", flush=True)
ctx.close()
