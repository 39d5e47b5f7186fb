"""Developer probe (DEVELOPER=1 build): the C3 frame on the tree kernel with whatever OPE_* switches the environment sets.
  steady : from the generator's pose, 40 warm-up + 160 timed launches, HIP-event kernel time (as tools/ab_probe.py)
  window : from a pose 0.13 (Frobenius) off, the driver's window: launches 5-24 timed, wall clock per step and kernel time
  chunks : which walk served the chunks of a measuring launch and how long they took (as tools/chain_probe.py)"""
import ctypes as C, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
if os.environ.get("PROBE_LIB"):   # an A/B build: make -C object-pose-estimation_amd VARIANT=<name> ...
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
synth = importlib.import_module("object-pose-estimation_amd.synth")
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["steady"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
grid = int(os.environ.get("PROBE_GRID", "0"))
label = " ".join([f"lib={os.environ['PROBE_LIB']}"] * bool(os.environ.get("PROBE_LIB")) + [f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OPE_")]) or "defaults"
tgt = synth.model_surface(100_000, 1)
gt = np.linalg.inv(synth.ground_truth_pose()).astype(np.float32)
src = synth.scene_cloud(1_000_000, clutter_frac=0.10)
nch = (len(src) + 63) // 64
fshift = int(os.environ.get('OPE_FAR_SHIFT', '6'))
nch2 = nch + (nch << (6 - fshift)) + 2   # Morton chunks, then the far chunks of the plan
TICK_US = 16 / 2400.0


def off_pose():
    # a start like the SAC-IA pose of bench.py: a few degrees and millimetres off
    a = np.deg2rad(3.0)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = [0.004, -0.003, 0.002]
    return (T @ gt.astype(np.float64)).astype(np.float32)


if "steady" in modes:
    res = []
    for r in range(reps):
        ctx = ope.Context(0)
        cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=grid)
        p = ope.default_icp_params(max_iterations=201, mse_threshold_absolute=-1.0, check_every=0)
        ctx.icp_begin(cs, ix, p, gt)
        ctx.icp_iterate(40); ctx.sync()
        ctx.icp_profile(160)
        t0 = time.perf_counter(); ctx.icp_iterate(160); ctx.sync(); dt = time.perf_counter() - t0
        km, kn = ctx.icp_profile_read()
        out = ctx.icp_end(); ctx.close()
        res.append((km / kn * 1e3, dt / 160 * 1e6))
    print(f"[{label}] steady: kernel us " + " ".join(f"{k:6.1f}" for k, _ in res) + " | step us " + " ".join(f"{s:6.1f}" for _, s in res), flush=True)

if "window" in modes:
    res = []
    for r in range(reps):
        ctx = ope.Context(0)
        cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=grid)
        p = ope.default_icp_params(max_iterations=201, mse_threshold_absolute=-1.0, check_every=0)
        ctx.icp_begin(cs, ix, p, off_pose())
        ctx.icp_iterate(5); ctx.sync()
        ctx.icp_profile(20)
        t0 = time.perf_counter(); ctx.icp_iterate(20); ctx.sync(); dt = time.perf_counter() - t0
        km, kn = ctx.icp_profile_read()
        T = ctx.icp_current_transform()
        ctx.icp_end(); ctx.close()
        res.append((km / kn * 1e3, dt / 20 * 1e6, float(np.linalg.norm(np.asarray(T, np.float64) - gt))))
    print(f"[{label}] window: kernel us " + " ".join(f"{k:6.1f}" for k, _, _ in res) + " | step us " + " ".join(f"{s:6.1f}" for _, s, _ in res)
          + f" | pose err {res[-1][2]:.4f}", flush=True)

if "chunks" in modes:
    L = ope.lib()
    L.ope_debug_chunk_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ope_debug_chunk_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    ctx = ope.Context(0)
    cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt), grid=grid)
    p = ope.default_icp_params(max_iterations=200, mse_threshold_absolute=-1.0, check_every=0)
    ctx.icp_begin(cs, ix, p, gt)
    ctx.icp_iterate(63); ctx.sync()          # launch 63 is the measuring launch before the plan step at 64
    assert L.ope_debug_chunk_stats(ctx.h, nch2, None) == 0
    ctx.icp_profile(1); ctx.icp_iterate(1); ctx.sync()
    km, kn = ctx.icp_profile_read()
    cost = np.zeros(nch2, np.uint32); order = np.zeros(nch2, np.uint32); info = np.zeros(8, np.uint32)
    assert L.ope_debug_chunk_costs(ctx.h, cost.ctypes.data, order.ctypes.data, info.ctypes.data, nch2) == 0
    st = np.zeros((nch2, 4), np.uint32)
    assert L.ope_debug_chunk_stats(ctx.h, nch2, st.ctypes.data) == 0
    assert L.ope_debug_chunk_stats(ctx.h, 0, None) == 0
    us = cost.astype(np.float64) * TICK_US
    print(f"[{label}] chunks: measuring launch {km/kn*1e3:.1f} us; sum of chunk time {us.sum()/1e3:.1f} ms = {us.sum()/6144:.1f} us per wave")
    path = st[:, 0] & 255
    span = st[:, 0] >> 8
    for name, code in (("per-lane", 0), ("packet", 1), ("groups", 2)):
        m = (path == code) & (us > 0)
        if m.any():
            print(f"    {name:8s}: {m.sum():6d} chunks, duration mean {us[m].mean():6.1f} us, p50 {np.percentile(us[m], 50):6.1f}, p99 {np.percentile(us[m], 99):6.1f}, max {us[m].max():6.1f};"
                  f" share of chunk time {us[m].sum()/us.sum():.2f}" + (f"; node steps {st[m,1].mean():.1f} (max {st[m,1].max()}), leaf scans {st[m,2].mean():.1f}, back-ups {st[m,3].mean():.1f}" if code == 1 else ""))
    print(f"    plan_info {info.tolist()}")
    if os.environ.get("OPE_FAR"):
        nfc = (int(info[1]) + (1 << fshift) - 1) >> fshift
        for a, b, nm in ((nch, nch + nfc, f"far chunks ({info[1]} queries)"), (0, nch, "Morton chunks (near lanes only)")):
            m = np.zeros(nch2, bool); m[a:b] = True
            m &= us > 0
            pk = m & (path == 1)
            print(f"    {nm}: {m.sum()} chunks with work, packet {pk.sum()}, duration mean {us[m].mean():.1f} us p50 {np.percentile(us[m],50):.1f} p99 {np.percentile(us[m],99):.1f} max {us[m].max():.1f}, sum {us[m].sum()/1e3:.1f} ms"
                  + (f"; packet node steps {st[pk,1].mean():.1f} (max {st[pk,1].max()}) leaf scans {st[pk,2].mean():.1f} back-ups {st[pk,3].mean():.1f}" if pk.any() else ""))
            if pk.any():
                for lo_, hi_ in ((0, 8), (8, 32), (32, 128), (128, 512), (512, 1 << 24)):
                    q = pk & (span >= lo_) & (span < hi_)
                    if q.any():
                        print(f"        start-leaf span [{lo_},{hi_}): {q.sum()} chunks, {us[q].mean():.1f} us mean, {us[q].max():.1f} max, node steps {st[q,1].mean():.1f}")
    ctx.icp_end(); ctx.close()
