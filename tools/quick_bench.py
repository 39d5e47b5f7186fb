"""Developer timing probe (not the judged bench): ICP iterations/s for C2/C3 with the early exit disabled."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")

def run(name, iters, leaf):
    src, tgt = synth.config_clouds(name)
    ctx = ope.Context(0)
    t0 = time.time(); cs = ctx.upload(src); t1 = time.time(); ct = ctx.upload(tgt); ix = ctx.build_index(ct, leaf_size=leaf); t2 = time.time()
    p = ope.default_icp_params(max_iterations=iters, mse_threshold_absolute=-1.0, check_every=0)
    out = ctx.icp(cs, ix, ope.default_icp_params(max_iterations=3, mse_threshold_absolute=-1.0, check_every=0))
    t3 = time.time(); out = ctx.icp(cs, ix, p); t4 = time.time()
    print(f"{name} leaf={leaf}: upload {t1-t0:.3f}s index {t2-t1:.3f}s  {iters} it in {(t4-t3)*1e3:.2f} ms -> {iters/(t4-t3):.1f} it/s ({(t4-t3)/iters*1e6:.1f} us/it) mse={out.last_mse:.3e} ncorr={out.n_corr}", flush=True)
    ctx.close()

if __name__ == "__main__":
    for leaf in (8, 16, 32):
        run("C2", 50, leaf)
    for leaf in (8, 16, 32):
        run("C3", 100, leaf)
