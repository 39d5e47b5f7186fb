"""BuildModel's registration loop on the HIP path (SURVEY.md §8f row "C5").

Host-side mirror of `RegMeshPcd::getIcpNormal` (BuildModel/src/regmeshpcd.cpp:63-206) and
`RegMeshPcd::registerPointClouds` (:210-271): the sequence and parameters are the reference's, every
stage is a C-ABI call into libope_hip.so (normals, index build, ICP loop, transform).  Meshing
(`generateMesh`, :273-343) is out of scope.

Things a maintainer must know about, all recorded in DESIGN.md:
  * the estimator is the reference's: `TransformationEstimationPointToPlane` (Levenberg-Marquardt, :162,:193) =
    `OPE_EST_POINT_TO_PLANE_LM` (csrc/lm.hip: one device reduction per functor evaluation, Eigen's LM logic on the host).
    `estimator="lls"` selects the linearised solve (`IterativeClosestPointWithNormals`' own default) instead: same
    objective, no host synchronisation per iteration, but its increments are NOT within 1e-4 of LM's.
  * `p_maxCorrDist` only reaches the stand-alone `determineCorrespondences` call (:145) whose result
    the reference discards; the ICP object itself keeps PCL's default correspondence distance
    (sqrt(DBL_MAX)).  `use_max_corr_dist_in_icp=False` reproduces that.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


@dataclass
class PairResult:
    T: np.ndarray
    iterations: int
    converged: bool
    fitness: float
    n_source: int
    n_target: int


@dataclass
class RegistrationResult:
    cloud: np.ndarray                      # accumulated, registered cloud (frame of the last input)
    pairs: list = field(default_factory=list)
    rgb: np.ndarray | None = None          # packed colours of `cloud`, when the frames came with colours


def icp_params_with_normals(ope, corr_rej_thresh: float, max_iterations: int, max_corr_dist: float | None = None, estimator: str = "lm"):
    """The parameter block of getIcpNormal (regmeshpcd.cpp:142-194)."""
    kw = dict(max_iterations=int(max_iterations),            # :179
              transformation_epsilon=1e-8,                  # :182
              euclidean_fitness_epsilon=1e-8,               # :184
              corr_mode=ope.CORR_NORMAL_SHOOTING,           # :139-145,:187
              k_normal_shooting=20,                         # :144
              use_surface_normal_rej=1,                     # :148-158,:190
              surface_normal_thr=float(corr_rej_thresh),    # :158
              estimator=ope.EST_POINT_TO_PLANE_LM if estimator == "lm" else ope.EST_POINT_TO_PLANE_LLS)   # :162,:193
    if max_corr_dist is not None:
        kw["max_corr_dist"] = float(max_corr_dist)
    return ope.default_icp_params(**kw)


def get_icp_normal(ope, ctx, source_xyz, target_xyz, corr_rej_thresh: float = 0.7, max_iterations: int = 500,
                   max_corr_dist: float = 0.005, use_max_corr_dist_in_icp: bool = False, k_normals: int = 12, estimator: str = "lm"):
    """One frame pair: normals(k=12) on both clouds -> ICP with normals -> aligned source.

    Returns (aligned_xyz, PairResult).  regmeshpcd.cpp:63-206.
    """
    src = ctx.upload(source_xyz)
    tgt = ctx.upload(target_xyz)
    try:
        ctx.normals(src, k_normals, fetch=False)     # :72-84  (the normals stay on the device, attached to the cloud)
        ctx.normals(tgt, k_normals, fetch=False)     # :86-90
        index = ctx.build_index(tgt)
        try:
            p = icp_params_with_normals(ope, corr_rej_thresh, max_iterations,
                                        max_corr_dist if use_max_corr_dist_in_icp else None, estimator)
            out = ctx.icp(src, index, p)        # :196
            fit, _, _ = ctx.fitness(src, index, out.T)   # :198
            aligned = ctx.transform_cloud(src, out.T)    # :203
        finally:
            index.free()
    finally:
        src.free()
        tgt.free()
    return aligned, PairResult(out.T, out.iterations, out.converged, fit, len(source_xyz), len(target_xyz))


def register_point_clouds(ope, ctx, frames, max_corr_dist: float = 0.005, corr_rej_thresh: float = 0.7,
                          max_iterations: int = 500, colors=None, on_device: bool = True, k_normals: int = 12,
                          estimator: str = "lm", use_max_corr_dist_in_icp: bool = False) -> RegistrationResult:
    """Sequential accumulate-and-register over N frames (regmeshpcd.cpp:210-271).

    cloudTemp = frame 0; for every next frame: align cloudTemp to it, then cloudTemp = aligned + frame.
    `colors` (one packed-rgb uint32 array per frame, the PointXYZRGB payload) ride along untouched, as the rgb
    field does through transformPointCloud and operator+= in the reference.

    on_device (default): the accumulated cloud never leaves the GPU between pairs — every frame is uploaded once,
    `aligned + target` is ope_cloud_concat (transform, append and re-sort on the device) and the result is fetched once
    at the end.  on_device=False goes through get_icp_normal pair by pair (host clouds in and out, as the reference's
    function signature has it); both give the same numbers.
    """
    if len(frames) == 0:
        raise ValueError("register_point_clouds: no frames")
    if colors is not None and (len(colors) != len(frames) or any(len(c) != len(f) for c, f in zip(colors, frames))):
        raise ValueError("register_point_clouds: colors must match the frames point for point")
    res = RegistrationResult(np.ascontiguousarray(frames[0], np.float32))
    if not on_device:
        acc = res.cloud
        for i in range(len(frames) - 1):
            target = np.ascontiguousarray(frames[i + 1], np.float32)
            aligned, pr = get_icp_normal(ope, ctx, acc, target, corr_rej_thresh, max_iterations, max_corr_dist,
                                         use_max_corr_dist_in_icp, k_normals, estimator)
            acc = np.concatenate([aligned, target], axis=0)   # :254-258
            res.pairs.append(pr)
        res.cloud = acc
    else:
        p = icp_params_with_normals(ope, corr_rej_thresh, max_iterations, max_corr_dist if use_max_corr_dist_in_icp else None, estimator)
        acc = ctx.upload(res.cloud)
        try:
            for i in range(len(frames) - 1):
                tgt = ctx.upload(np.ascontiguousarray(frames[i + 1], np.float32))
                index = None
                try:
                    ctx.normals(acc, k_normals, fetch=False)     # :72-84
                    ctx.normals(tgt, k_normals, fetch=False)     # :86-90
                    index = ctx.build_index(tgt)
                    out = ctx.icp(acc, index, p)                 # :196
                    fit, _, _ = ctx.fitness(acc, index, out.T)   # :198
                    res.pairs.append(PairResult(out.T, out.iterations, out.converged, fit, acc.n, tgt.n))
                    nxt = ctx.concat(acc, out.T, tgt)            # :203 transformPointCloud, :254 += target
                finally:
                    if index is not None:
                        index.free()
                acc.free()
                tgt.free()
                acc = nxt
            res.cloud = ctx.download(acc)
        finally:
            acc.free()
    if colors is not None:
        res.rgb = np.concatenate([np.ascontiguousarray(c, np.uint32) for c in colors])
    return res


def build_model_from_directory(ope, ctx, pcd_dir: str, out_path: str | None = None, limits=None, segment=None,
                               max_corr_dist: float = 0.005, corr_rej_thresh: float = 0.7, max_iterations: int = 500,
                               **kw) -> RegistrationResult:
    """BuildModel's file-to-file loop (BuildModel/src/main.cpp:113-153 load, :185-190 crop, :207-225 register + save).

    Loads every `*.pcd` of `pcd_dir`, crops each frame with the pass-through box `limits` = (xmin, xmax, ymin, ymax,
    zmin, zmax) on the device, registers the frames sequentially and writes `<out_path>` as binary PCD
    (`FIELDS x y z rgb` when the inputs carry colour), which is what `savePCDFile(name, cloud, true)` writes.

    The reference walks the directory in `boost::filesystem::directory_iterator` order, which is unspecified; here
    the files are taken in name order.  Between crop and registration the reference cuts the object off its
    supporting plane (`ObjectSegmentationPlane`, main.cpp:186): that step is outside this path (SURVEY.md §8f-4
    ranks it after the file loop) and enters as the optional `segment(xyz, rgb) -> (xyz, rgb)` hook.
    """
    import os
    from . import pcd

    names = sorted(n for n in os.listdir(pcd_dir) if n.lower().endswith(".pcd"))
    if not names:
        raise ValueError(f"build_model_from_directory: no .pcd files in {pcd_dir!r}")
    frames, colors = [], []
    for name in names:
        xyz, rgb = pcd.read_pcd(os.path.join(pcd_dir, name))
        if limits is not None:
            lim = np.asarray(limits, np.float32).reshape(3, 2)
            c = ctx.upload(xyz)
            try:
                keep = ctx.pass_through(c, lim[:, 0], lim[:, 1])      # processingpcd.cpp getPassThrough
            finally:
                c.free()
        else:
            keep = np.flatnonzero(np.isfinite(xyz).all(axis=1)).astype(np.int32)
        xyz = xyz[keep]
        rgb = rgb[keep] if rgb is not None else None
        if segment is not None:
            xyz, rgb = segment(xyz, rgb)
        frames.append(np.ascontiguousarray(xyz, np.float32))
        colors.append(rgb)
    have_rgb = all(c is not None for c in colors)
    res = register_point_clouds(ope, ctx, frames, max_corr_dist, corr_rej_thresh, max_iterations,
                                colors=colors if have_rgb else None, **kw)
    if out_path is not None:
        pcd.write_pcd(out_path, res.cloud, res.rgb)
    return res


# ---------------------------------------------------------------------------------------------------------------
# One process per GPU.  The frames are registered sequentially as in the reference, so the
# parallelism is inside each pairwise registration: every rank uploads the whole accumulated source (its index is
# what the normals of any point need) but computes normals and ICP sums only for its contiguous slice of it; the
# target frame and its index are replicated; the sums are all-reduced once per iteration (sharded.run_sharded_icp).
def get_icp_normal_sharded(ope, ctx, source_xyz, target_xyz, corr_rej_thresh: float = 0.7, max_iterations: int = 500,
                           k_normals: int = 12, engine_cls=None, group=None):
    """get_icp_normal with the source sharded over the ranks of `group`.  Every rank returns the same result."""
    import torch.distributed as dist

    from . import sharded

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    engine_cls = engine_cls or sharded.GpuEngine
    source_xyz = np.ascontiguousarray(source_xyz, np.float32)
    lo, hi = sharded.shard_range(len(source_xyz), world, rank)
    src_all = ctx.upload(source_xyz)
    src = ctx.upload(source_xyz[lo:hi])
    tgt = ctx.upload(target_xyz)
    try:
        all_index = ctx.build_index(src_all)
        try:
            ctx.normals_from(src, all_index, k_normals, fetch=False)   # :72-84, this rank's slice only
        finally:
            all_index.free()
        ctx.normals(tgt, k_normals, fetch=False)                       # :86-90
        index = ctx.build_index(tgt)
        try:
            # the stepwise (torch.distributed) driver exchanges one fixed set of sums per iteration: the linearised
            # estimator; the LM estimator shards through the library's own communicator (ope_icp_iterate)
            p = icp_params_with_normals(ope, corr_rej_thresh, max_iterations, estimator="lls")
            eng = engine_cls(ope, ctx, src, index, p, None, len(source_xyz), len(target_xyz))
            out = sharded.run_sharded_icp(eng, max_iterations, check_every=p.check_every, group=group)   # :196
            aligned = ctx.transform_cloud(src_all, out.T)             # :203
        finally:
            index.free()
    finally:
        src_all.free()
        src.free()
        tgt.free()
    return aligned, PairResult(out.T, out.iterations, bool(out.converged), float("nan"), len(source_xyz), len(target_xyz))


def register_point_clouds_sharded(ope, ctx, frames, corr_rej_thresh: float = 0.7, max_iterations: int = 500, **kw) -> RegistrationResult:
    """register_point_clouds with every pairwise registration sharded over the process group (regmeshpcd.cpp:210-271)."""
    if len(frames) == 0:
        raise ValueError("register_point_clouds_sharded: no frames")
    acc = np.ascontiguousarray(frames[0], np.float32)
    res = RegistrationResult(acc)
    for i in range(len(frames) - 1):
        target = np.ascontiguousarray(frames[i + 1], np.float32)
        aligned, pr = get_icp_normal_sharded(ope, ctx, acc, target, corr_rej_thresh, max_iterations, **kw)
        acc = np.concatenate([aligned, target], axis=0)
        res.pairs.append(pr)
    res.cloud = acc
    return res


def register_point_clouds_sharded_native(ope, ctx, frames, world: int, rank: int, corr_rej_thresh: float = 0.7,
                                         max_iterations: int = 500, k_normals: int = 12, estimator: str = "lm") -> RegistrationResult:
    """The same loop (regmeshpcd.cpp:210-271) as the multi-GPU path proper: device-resident like register_point_clouds, sharded
    like register_point_clouds_sharded, the sums exchanged by the library's own communicator — the caller has connected `ctx`
    (comm_p2p_connect or comm_init) — and the reference's estimator (Levenberg-Marquardt point-to-plane, :162,:193).

    Every rank keeps the whole accumulated cloud on its GPU (any point's normal needs its neighbours from the whole cloud, and
    `cloudTemp = aligned + target` is the same concat everywhere: the ranks' transforms are bit-identical, the sums being added in
    rank order on every rank); what a rank owns is a contiguous slice of it — gathered on the device (ope_cloud_select), its
    normals searched in the whole cloud's index (ope_normals_from), its share of the sums.  Per pair a rank builds the index of
    the accumulated cloud and the concat in full and 1/world of the normals and of the ICP launches; nothing but a new frame
    and the slice's index list crosses PCIe, nothing but 17 + 91 sums per iteration crosses the fabric."""
    from . import sharded

    if len(frames) == 0:
        raise ValueError("register_point_clouds_sharded_native: no frames")
    p = icp_params_with_normals(ope, corr_rej_thresh, max_iterations, None, estimator)
    res = RegistrationResult(np.ascontiguousarray(frames[0], np.float32))
    acc = ctx.upload(res.cloud)
    try:
        for i in range(len(frames) - 1):
            tgt = ctx.upload(np.ascontiguousarray(frames[i + 1], np.float32))
            index = src = None
            try:
                lo, hi = sharded.shard_range(acc.n, world, rank)
                all_index = ctx.build_index(acc)
                try:
                    src = ctx.select(acc, np.arange(lo, hi, dtype=np.int32))
                    ctx.normals_from(src, all_index, k_normals, fetch=False)   # :72-84, this rank's slice
                finally:
                    all_index.free()
                ctx.normals(tgt, k_normals, fetch=False)                       # :86-90
                index = ctx.build_index(tgt)
                ctx.icp_set_global_sizes(acc.n, tgt.n)
                out = ctx.icp(src, index, p)                                   # :196
                res.pairs.append(PairResult(out.T, out.iterations, out.converged, float("nan"), acc.n, tgt.n))
                nxt = ctx.concat(acc, out.T, tgt)                              # :203, :254
            finally:
                if index is not None:
                    index.free()
                if src is not None:
                    src.free()
            acc.free()
            tgt.free()
            acc = nxt
        res.cloud = ctx.download(acc)
    finally:
        acc.free()
    return res
