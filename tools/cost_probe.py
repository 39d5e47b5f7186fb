"""Developer probe: distribution of the per-chunk costs the accumulate kernel measures on the C3 mix (steady state,
ground-truth pose) and what the static snake schedule makes of them."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
tgt = synth.model_surface(100_000, 1)
guess = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
L = ope.lib()
L.ope_debug_chunk_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
TICK_US = 16 / 2400.0   # s_memtime counts shader clocks (~2.4 GHz under load); costs are ticks >> 4
for frac, nq in [(0.10, 1_000_000), (0.0, 1_000_000), (0.10, 125_000)]:
    src = synth.scene_cloud(1_000_000, clutter_frac=frac)[:nq]
    ctx = ope.Context(0)
    cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=0)
    p = ope.default_icp_params(max_iterations=141, mse_threshold_absolute=-1.0, check_every=0)
    ctx.icp_begin(cs, ix, p, guess)
    ctx.icp_iterate(70); ctx.sync()
    ctx.icp_profile(30); ctx.icp_iterate(30); ctx.sync()
    km, kn = ctx.icp_profile_read()
    nch = (nq + 63) // 64
    cost = np.zeros(nch, np.uint32); order = np.zeros(nch, np.uint32); info = np.zeros(4, np.uint32)
    assert L.ope_debug_chunk_costs(ctx.h, cost.ctypes.data, order.ctypes.data, info.ctypes.data, nch) == 0
    us = cost.astype(np.float64) * TICK_US
    n_waves = min(2 * ((nq + 511) // 512), 1024) * 8
    resident = 256 * 4 * 6
    print(f"clutter {frac} n {nq}: kernel {km/kn*1e3:.1f} us; chunks {nch}, waves launched {n_waves} (resident {resident}), plan_info {info.tolist()}")
    print("   chunk duration us: mean %.1f p50 %.1f p90 %.1f p99 %.1f p99.9 %.1f max %.1f; sum/resident waves = %.1f us" % (
        us.mean(), *np.percentile(us, [50, 90, 99, 99.9, 100]), us.sum() / resident))
    so = np.sort(us)[::-1]
    for k in (10, 100, 500, 1500, 3000):
        if k < nch: print(f"   the {k} costliest: mean {so[:k].mean():.1f} us, share of the total {so[:k].sum()/so.sum():.3f}")
    # the snake schedule over the launched waves, in the plan's order
    load = np.zeros(n_waves)
    cs_ = us[order]
    for r in range((nch + n_waves - 1) // n_waves):
        seg = cs_[r * n_waves:(r + 1) * n_waves]
        idx = np.arange(len(seg)) if r % 2 == 0 else n_waves - 1 - np.arange(len(seg))
        load[idx] += seg
    print("   per-wave load of the snake schedule: mean %.1f p50 %.1f p99 %.1f max %.1f us" % (load.mean(), *np.percentile(load, [50, 99, 100])))
    ctx.icp_end(); ctx.close()
