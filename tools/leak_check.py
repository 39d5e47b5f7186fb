"""Developer probe: device memory before / after many upload + index + ICP + feature cycles (leak check)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ope = importlib.import_module("object-pose-estimation_amd")
synth = importlib.import_module("object-pose-estimation_amd.synth")
bm = importlib.import_module("object-pose-estimation_amd.buildmodel")
src = synth.scene_cloud(60000); tgt = synth.model_surface(40000, 1)
ctx = ope.Context(0)
def cycle():
    cs = ctx.upload(src); ct = ctx.upload(tgt)
    ctx.normals(ct, 12, fetch=False); ctx.normals(cs, 12, fetch=False)
    ix = ctx.build_index(ct)
    ctx.icp(cs, ix, ope.default_icp_params(max_iterations=5))
    ctx.icp(cs, ix, bm.icp_params_with_normals(ope, 0.7, 3))
    keep = ctx.uniform_sampling(ct, 0.01)
    kc = ctx.upload(tgt[keep]); ctx.normals(kc, 30); ctx.fpfh(kc, 0.03); kc.free()
    ctx.voxel_grid(cs, 0.01)
    ix.free(); cs.free(); ct.free()
cycle(); ctx.sync()
free0, total = torch.cuda.mem_get_info()
for _ in range(40): cycle()
ctx.sync()
free1, _ = torch.cuda.mem_get_info()
print(f"free before {free0/2**20:.1f} MiB, after 40 cycles {free1/2**20:.1f} MiB, delta {(free0-free1)/2**20:.2f} MiB")
ctx.close()
