"""Skip certificates on the C3 frame: kernel time per launch (HIP events on the launch stream) and queries answered from
their certificate, launch by launch, for a run of N iterations from the generator's pose perturbed like a coarse pose.
    python tools/cert_probe.py [auto|off|always] [N]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
if os.environ.get("PROBE_LIB"):
    ope.LIB_PATH = os.path.join(os.path.dirname(ope.LIB_PATH), f"libope_hip_{os.environ['PROBE_LIB']}.so")
mode = sys.argv[1] if len(sys.argv) > 1 else "auto"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = os.environ.get("PROBE_WORKLOAD", "C3")
scene, model = synth.config_clouds(wl)
ctx = ope.Context(0)
cs = ctx.upload(scene); ix = ctx.build_index(ctx.upload(model))
gt = np.linalg.inv(synth.ground_truth_pose())
a = 0.06
R = np.array([[np.cos(a), -np.sin(a), 0, 0.004], [np.sin(a), np.cos(a), 0, -0.003], [0, 0, 1, 0.005], [0, 0, 0, 1]])
guess = (R @ gt).astype(np.float32)
p = ope.default_icp_params(max_iterations=N + 1, transformation_epsilon=0.0, euclidean_fitness_epsilon=0.0, mse_threshold_absolute=-1.0, check_every=0,
                           skip_certificates={"auto": 0, "off": 1, "always": 2}[mode])
for rep in range(2):
    ctx.icp_begin(cs, ix, p, guess)
    ctx.icp_profile(N)
    ctx.sync(); t0 = time.perf_counter()
    cert, reasons = [], []
    if os.environ.get("PROBE_STATS"):
        import ctypes as C
        for k in range(N):
            ctx.icp_iterate(1); cert.append(ctx.icp_certificate_stats())
            if os.environ.get("PROBE_LIB") == "dev":
                r = (C.c_uint32 * 4)(); ope.lib().ope_debug_cert_reasons(ctx.h, r, 1); reasons.append(list(r))
    else:
        ctx.icp_iterate(N)
    ctx.sync(); dt = time.perf_counter() - t0
    ms = ctx.icp_profile_launches(); ctx.icp_profile(0)
    st = ctx.icp_certificate_stats()
    out = ctx.icp_end()
print(f"{wl} certificates={mode} lib={os.environ.get('PROBE_LIB','product')}: {N} iterations {dt*1e3:.2f} ms wall, kernel sum {ms.sum():.2f} ms; stats {st}; kernels {ctx.icp_kernel_launches()}")
print("kernel us per launch, by tens:", [int(1000 * ms[k:k + 10].mean()) for k in range(0, N, 10)])
if cert:
    c = np.diff([0] + [x["certified"] for x in cert])
    print("certified per launch (every 5th):", [int(v) for v in c[::5]])
    print("last_move um (every 5th):", [round(x["last_move"] * 1e6, 1) for x in cert[::5]])
if cert and reasons:
    print("per launch {expired, tie among candidates, no certificate, walks that build} (every 10th):", reasons[::10])
