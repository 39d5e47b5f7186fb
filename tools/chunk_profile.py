"""Developer probe: per-chunk (one wave = 64 Morton-adjacent queries) traversal cost on C3 after 30 ICP iterations."""
import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
src, tgt = synth.config_clouds(name)
ctx = ope.Context(0)
cs = ctx.upload(src); ix = ctx.build_index(ctx.upload(tgt))
L = ope.lib()
L.ope_debug_chunk_profile.argtypes = [C.c_void_p] * 3 + [C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_longlong)]
out = ctx.icp(cs, ix, ope.default_icp_params(max_iterations=30, mse_threshold_absolute=-1.0, check_every=0))
nch = (len(src) + 63) // 64
for use_hint in (1, 0):
    buf = np.zeros((nch, 6), np.int64)
    t = ope.colmajor(out.T)
    for rep in range(2):
        rc = L.ope_debug_chunk_profile(ctx.h, cs.h, ix.h, t.ctypes.data_as(C.POINTER(C.c_float)), use_hint, buf.ctypes.data_as(C.POINTER(C.c_longlong)))
        assert rc == 0
    cyc, mn, mp, sn, sp, t0 = buf.T
    span = (t0.max() + cyc[t0.argmax()] - t0.min())
    print(f"hint={use_hint}: chunks {nch}, kernel span {span/1e6:.2f} Mcycles (~{span/2400:.0f} us @2.4GHz), sum of chunk cycles {cyc.sum()/1e9:.2f} G")
    print(f"   chunk cycles: mean {cyc.mean():.0f} p50 {np.percentile(cyc,50):.0f} p90 {np.percentile(cyc,90):.0f} p99 {np.percentile(cyc,99):.0f} max {cyc.max()}")
    print(f"   lane-max node steps: mean {mn.mean():.1f} p99 {np.percentile(mn,99):.0f} max {mn.max()};  lane-max points: mean {mp.mean():.1f} max {mp.max()}")
    print(f"   per-query node steps: mean {sn.sum()/len(src):.1f}; points {sp.sum()/len(src):.1f}")
    steps = mn + mp / 4.0
    print(f"   cycles per (node step + point batch) of the slowest lane: median {np.median(cyc/np.maximum(steps,1)):.0f}")
    order = np.argsort(cyc)[::-1][:5]
    print("   slowest chunks:", [(int(c), int(cyc[c]), int(mn[c]), int(mp[c])) for c in order])
