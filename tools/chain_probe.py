"""Developer probe: what the costliest chunks of the C3 frame's tree-kernel launch are made of — which walk served them
(per-lane / packet / 8-lane groups), how long they took, and for packet walks how many node steps, leaf scans and back-ups."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
tgt = synth.model_surface(100_000, 1)
guess = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
L = ope.lib()
L.ope_debug_chunk_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
L.ope_debug_chunk_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
TICK_US = 16 / 2400.0
src = synth.scene_cloud(1_000_000, clutter_frac=0.10)
nq = len(src); nch = (nq + 63) // 64
ctx = ope.Context(0)
cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=0)
p = ope.default_icp_params(max_iterations=200, mse_threshold_absolute=-1.0, check_every=0)
ctx.icp_begin(cs, ix, p, guess)
ctx.icp_iterate(62); ctx.sync()          # launch 63 is a measuring launch (the one before the plan step at 64)
assert L.ope_debug_chunk_stats(ctx.h, nch, None) == 0
ctx.icp_profile(4); ctx.icp_iterate(4); ctx.sync()
km, kn = ctx.icp_profile_read()
cost = np.zeros(nch, np.uint32); order = np.zeros(nch, np.uint32); info = np.zeros(8, np.uint32)
assert L.ope_debug_chunk_costs(ctx.h, cost.ctypes.data, order.ctypes.data, info.ctypes.data, nch) == 0
st = np.zeros((nch, 4), np.uint32)
assert L.ope_debug_chunk_stats(ctx.h, nch, st.ctypes.data) == 0
assert L.ope_debug_chunk_stats(ctx.h, 0, None) == 0
us = cost.astype(np.float64) * TICK_US
print(f"kernel {km/kn*1e3:.1f} us (4 launches, one of them measuring); plan_info {info[:6].tolist()}")
path = st[:, 0]
for name, code in (("per-lane", 0), ("packet", 1), ("groups", 2)):
    m = path == code
    if m.any():
        print(f"  {name:8s}: {m.sum():6d} chunks, duration mean {us[m].mean():6.1f} us, p99 {np.percentile(us[m], 99):6.1f}, max {us[m].max():6.1f}; share of total chunk time {us[m].sum()/us.sum():.2f}")
top = np.argsort(us)[::-1][:300]
print("  the 300 costliest chunks: per-lane %d, packet %d, groups %d" % tuple((path[top] == c).sum() for c in (0, 1, 2)))
pk = top[path[top] == 1]
if len(pk):
    print(f"    packet ones: duration {us[pk].mean():.1f} us; node steps {st[pk,1].mean():.0f}, leaf scans {st[pk,2].mean():.0f}, back-ups {st[pk,3].mean():.0f}"
          f" -> {us[pk].mean()/(st[pk,1].mean()+st[pk,2].mean()):.2f} us per step-or-scan")
gr = np.where(path == 2)[0]
if len(gr):
    slot_us = st[gr, 1].astype(np.float64) * TICK_US
    print(f"    group-walked chunks: {len(gr)}; per-lane duration (measuring launch) mean {us[gr].mean():.1f} us; longest of their eight slots' walks: mean {slot_us.mean():.1f} us, max {slot_us.max():.1f} us"
          f" -> ratio {slot_us.mean()/us[gr].mean():.2f}")
pl = top[path[top] == 0]
if len(pl):
    print(f"    per-lane ones: duration {us[pl].mean():.1f} us")
allpk = path == 1
print(f"  all packet chunks: node steps {st[allpk,1].mean():.1f}, leaf scans {st[allpk,2].mean():.1f}, back-ups {st[allpk,3].mean():.1f}, duration {us[allpk].mean():.1f} us")
ctx.icp_end(); ctx.close()
