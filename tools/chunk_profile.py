"""Developer probe: per-chunk (one wave = 64 Morton-adjacent queries) traversal cost on C3 after 30 ICP iterations,
with in-kernel s_memtime stamps per phase (diagnostic build of the loop: shares, not absolute times)."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
src, tgt = synth.config_clouds(name)
if len(sys.argv) > 2:  # first 1/W slice of the source (one rank's shard)
    src = src[: len(src) // int(sys.argv[2])]
ctx = ope.Context(0)
cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt))
L = ope.lib()
L.ope_debug_chunk_profile.argtypes = [C.c_void_p] * 3 + [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_longlong)]
out = ctx.icp(cs, ix, ope.default_icp_params(max_iterations=30, mse_threshold_absolute=-1.0, check_every=0))
nch = (len(src) + 63) // 64
for use_hint in (1, 0):
    buf = np.zeros((nch, 10), np.int64)
    t = ope.colmajor(out.T)
    for rep in range(2):
        rc = L.ope_debug_chunk_profile(ctx.h, cs.h, ix.h, t.ctypes.data_as(C.POINTER(C.c_float)), use_hint, buf.ctypes.data_as(C.POINTER(C.c_longlong)))
        assert rc == 0
    cyc, mn, mp, ce, cn, cl, cp, trips = buf[:, :8].T
    lane_trips = buf[:, 8]
    print(f"   lane utilisation of the walk loop (lane-trips with work / 64 x trips): {lane_trips.sum()/(64.0*trips.sum()):.3f}; "
          f"mean lane-trips per query {lane_trips.sum()/len(src):.2f} vs trips per chunk {trips.mean():.1f}")
    print(f"hint={use_hint}: chunks {nch}; sum of chunk cycles {cyc.sum()/1e9:.2f} G (/8192 slots = {cyc.sum()/8192/2400:.0f} us @2.4GHz)")
    print(f"   chunk cycles: mean {cyc.mean():.0f} p50 {np.percentile(cyc,50):.0f} p90 {np.percentile(cyc,90):.0f} p99 {np.percentile(cyc,99):.0f} max {cyc.max()}")
    print(f"   shares: eager-siblings {ce.sum()/cyc.sum():.2f} node-branch {cn.sum()/cyc.sum():.2f} leaf-branch {cl.sum()/cyc.sum():.2f} pop {cp.sum()/cyc.sum():.2f}")
    print(f"   loop trips per chunk: mean {trips.mean():.1f} max {trips.max()}; cycles per trip: node {cn.sum()/trips.sum():.0f} leaf {cl.sum()/trips.sum():.0f} pop {cp.sum()/trips.sum():.0f}")
    h = np.argsort(cyc)[::-1][:200]
    print(f"   heaviest 200 chunks: cycles {cyc[h].mean():.0f} trips {trips[h].mean():.0f} -> per trip node {cn[h].sum()/trips[h].sum():.0f} leaf {cl[h].sum()/trips[h].sum():.0f} pop {cp[h].sum()/trips[h].sum():.0f}; eager {ce[h].mean():.0f}")
