import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
src, tgt = synth.config_clouds("C3")
ctx = ope.Context(0)
cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt))
p = ope.default_icp_params(max_iterations=2000, mse_threshold_absolute=-1.0, check_every=0)
ctx.icp_begin(cs, ix, p, None)
ctx.icp_iterate(150); ctx.sync()
for rep in range(3):
    for prof in (0, 1):
        if prof: ctx.icp_profile(100)
        t0 = time.perf_counter(); ctx.icp_iterate(100); ctx.sync(); dt = time.perf_counter() - t0
        if prof: ctx.icp_profile_read(); ctx.icp_profile(0)
        print(f"profile events {'on ' if prof else 'off'}: {dt/100*1e6:.1f} us/iteration", flush=True)
ctx.icp_end(); ctx.close()
