"""ctypes binding of the CPU ORACLE (oracle/libope_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
PARITY UNPINNED — see oracle/ope_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPE_ORACLE_LIB: load another build of the same sources (tests/test_oracle_sanitizers.py points it at the
# AddressSanitizer/UBSan build)
_LIB_PATH = os.environ.get("OPE_ORACLE_LIB") or os.path.join(_HERE, "libope_oracle.so")
_SRCS = ["kdtree.c", "icp.c", "features.c", "filters.c", "pose.c", "ope_oracle.h", "Makefile"]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    stale = force or not os.path.exists(_LIB_PATH)
    if not stale:
        t = os.path.getmtime(_LIB_PATH)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in _SRCS)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, os.path.basename(_LIB_PATH)], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("euclidean_fitness_epsilon", C.c_double),
        ("max_corr_dist", C.c_double),
        ("use_reciprocal", C.c_int),
        ("min_correspondences", C.c_int),
        ("corr_mode", C.c_int),
        ("k_normal_shooting", C.c_int),
        ("use_surface_normal_rej", C.c_int),
        ("surface_normal_thr", C.c_double),
        ("use_self_occluded_rej", C.c_int),
        ("self_occluded_thr", C.c_double),
        ("mse_threshold_absolute", C.c_double),
        ("failure_after_max_iter", C.c_int),
        ("acc_mode", C.c_int),
        ("estimator", C.c_int),
        ("lm_precision", C.c_int),
        ("transform_mode", C.c_int),
    ]


class IcpResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("state", C.c_int),
        ("last_mse", C.c_double),
        ("n_corr", C.c_int),
        ("fitness", C.c_double),
        ("align_strength", C.c_double),
    ]


class Convergence(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("failure_after_max_iter", C.c_int),
        ("rotation_threshold", C.c_double),
        ("translation_threshold", C.c_double),
        ("mse_threshold_relative", C.c_double),
        ("mse_threshold_absolute", C.c_double),
        ("max_iterations_similar_transforms", C.c_int),
        ("iterations_similar_transforms", C.c_int),
        ("prev_mse", C.c_double),
        ("cur_mse", C.c_double),
        ("state", C.c_int),
    ]


class PoseEstimatorState(C.Structure):
    _fields_ = [
        ("first_time_pose", C.c_int),
        ("fitness_score_fine", C.c_double), ("aligned_strength", C.c_double),
        ("final_pose", C.c_float * 16),
        ("aligned_source", C.POINTER(C.c_float)), ("n_aligned", C.c_int),
        ("cloud_model", C.POINTER(C.c_float)), ("n_model", C.c_int),
        ("sacia_seed", C.c_uint64),
        ("coarse_calls", C.c_int), ("use_self_occluded", C.c_int), ("acc_mode", C.c_int), ("transform_mode", C.c_int),
        ("last_coarse", C.c_float * 16), ("last_fine", C.c_float * 16), ("last_rigid", C.c_float * 16),
        ("last_sacia_error", C.c_double), ("last_sacia_best", C.c_int),
        ("last_n_src_keys", C.c_int), ("last_n_tgt_keys", C.c_int), ("last_n_fine_src", C.c_int), ("last_n_fine_tgt", C.c_int),
        ("last_icp_iterations", C.c_int), ("last_icp_state", C.c_int), ("last_icp_n_corr", C.c_int),
    ]


CONV_NAMES = ["NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE", "NO_CORRESPONDENCES"]

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)


def _declare(L):
    L.orc_kdtree_build.restype = C.c_void_p
    L.orc_kdtree_build.argtypes = [_fp, C.c_int, C.c_int]
    L.orc_kdtree_free.argtypes = [C.c_void_p]
    L.orc_kdtree_knn.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, _ip, _fp, _ip]
    L.orc_kdtree_radius.restype = C.c_int64
    L.orc_kdtree_radius.argtypes = [C.c_void_p, _fp, C.c_int, C.c_float, C.c_int, _lp, _ip, _fp, C.c_int64]
    L.orc_bruteforce_nn.argtypes = [_fp, C.c_int, _fp, C.c_int, _ip, _fp]
    L.orc_umeyama.restype = C.c_int
    L.orc_umeyama.argtypes = [_fp, _fp, C.c_int, C.c_int, _fp]
    L.orc_umeyama_from_sums.restype = C.c_int
    L.orc_umeyama_from_sums.argtypes = [_dp, _dp, _fp]
    L.orc_svd3.argtypes = [_dp, _dp, _dp, _dp]
    L.orc_point_to_plane_lls.restype = C.c_int
    L.orc_point_to_plane_lls.argtypes = [_fp, _fp, _fp, C.c_int, _fp]
    L.orc_point_to_plane_lm.restype = C.c_int
    L.orc_point_to_plane_lm.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_convergence_init.argtypes = [C.POINTER(Convergence)]
    L.orc_convergence_step.restype = C.c_int
    L.orc_convergence_step.argtypes = [C.POINTER(Convergence), C.c_int, _fp, C.c_double]
    L.orc_icp_default_params.argtypes = [C.POINTER(IcpParams)]
    L.orc_icp.restype = C.c_int
    L.orc_icp.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_int, _fp, C.POINTER(IcpParams), _fp,
                          C.POINTER(IcpResult), _fp, _ip, _ip, _fp]
    L.orc_icp_fixed.restype = C.c_int
    L.orc_icp_fixed.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_int, _fp, C.POINTER(IcpParams), _ip, _ip, C.c_int, _fp,
                                C.POINTER(IcpResult), _fp, _ip, _ip, _fp]
    L.orc_icp_set_threads.argtypes = [C.c_int]
    L.orc_fitness.restype = C.c_double
    L.orc_fitness.argtypes = [_fp, C.c_int, _fp, C.c_int, _fp, C.c_double, C.POINTER(C.c_int)]
    L.orc_icp_partial_sums.argtypes = [_fp, C.c_int, C.c_void_p, _fp, _fp, C.c_double, _dp, _dp]
    L.orc_icp_partial_sums_mt.argtypes = [_fp, C.c_int, C.c_void_p, _fp, _fp, C.c_double, _dp, C.c_int, _dp]
    L.orc_transform_points.argtypes = [_fp, C.c_int, _fp, _fp]
    L.orc_transform_normals.argtypes = [_fp, C.c_int, _fp, _fp]
    L.orc_normals_knn.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, _fp]
    L.orc_pair_features.restype = C.c_int
    L.orc_pair_features.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp]
    L.orc_fpfh.argtypes = [_fp, _fp, C.c_int, C.c_float, _fp, _fp, _dp]
    L.orc_remove_nan.restype = C.c_int
    L.orc_remove_nan.argtypes = [_fp, C.c_int, _ip]
    L.orc_pass_through.restype = C.c_int
    L.orc_pass_through.argtypes = [_fp, C.c_int, _fp, _fp, _ip]
    L.orc_voxel_grid.restype = C.c_int
    L.orc_voxel_grid.argtypes = [_fp, C.c_int, _fp, _fp]
    L.orc_voxel_grid_rgb.restype = C.c_int
    L.orc_voxel_grid_rgb.argtypes = [_fp, C.c_void_p, C.c_int, _fp, _fp, C.c_void_p]
    L.orc_statistical_outlier_removal.restype = C.c_int
    L.orc_statistical_outlier_removal.argtypes = [_fp, C.c_int, C.c_int, C.c_double, _ip, _fp]
    L.orc_uniform_sampling.restype = C.c_int
    L.orc_uniform_sampling.argtypes = [_fp, C.c_int, C.c_float, _ip]
    L.orc_sacia_error.restype = C.c_double
    L.orc_sacia_error.argtypes = [_fp, C.c_int, C.c_void_p, _fp, C.c_double]
    L.orc_sacia.restype = C.c_int
    L.orc_sacia.argtypes = [_fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                            C.c_float, C.c_uint64, _ip, _fp, _dp, _ip]
    L.orc_feature_knn.argtypes = [_fp, C.c_int, _fp, C.c_int, C.c_int, _ip, _fp]
    L.orc_pose_estimator_init.argtypes = [C.POINTER(PoseEstimatorState)]
    L.orc_pose_estimator_free.argtypes = [C.POINTER(PoseEstimatorState)]
    L.orc_estimate_coarse_pose.restype = C.c_int
    L.orc_estimate_coarse_pose.argtypes = [C.POINTER(PoseEstimatorState), _fp, C.c_int, _fp, C.c_int, _fp]
    L.orc_estimate_fine_pose.restype = C.c_int
    L.orc_estimate_fine_pose.argtypes = [C.POINTER(PoseEstimatorState), _fp, C.c_int, _fp, C.c_int, _fp]
    L.orc_estimate_final_pose.restype = C.c_int
    L.orc_estimate_final_pose.argtypes = [C.POINTER(PoseEstimatorState), _fp, C.c_int, _fp, C.c_int, _fp, _dp, _dp]


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None:
        assert a.ndim == 2 and a.shape[1] == cols, a.shape
    return a


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class KdTree:
    """Exact k-NN / radius search (stand-in for pcl::search::KdTree)."""

    def __init__(self, xyz, leaf_max: int = 15):
        self.xyz = _f32(xyz, 3)
        self.h = lib().orc_kdtree_build(_p(self.xyz, _fp), len(self.xyz), leaf_max)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None

    def knn(self, q, k: int = 1):
        q = _f32(q, 3)
        idx = np.empty((len(q), k), np.int32)
        d2 = np.empty((len(q), k), np.float32)
        found = np.empty(len(q), np.int32)
        lib().orc_kdtree_knn(self.h, _p(q, _fp), len(q), k, _p(idx, _ip), _p(d2, _fp), _p(found, _ip))
        return idx, d2, found

    def radius(self, q, r: float, sorted_: bool = True):
        q = _f32(q, 3)
        offs = np.empty(len(q) + 1, np.int64)
        total = lib().orc_kdtree_radius(self.h, _p(q, _fp), len(q), r, int(sorted_), _p(offs, _lp), None, None, 0)
        idx = np.empty(max(total, 1), np.int32)
        d2 = np.empty(max(total, 1), np.float32)
        lib().orc_kdtree_radius(self.h, _p(q, _fp), len(q), r, int(sorted_), _p(offs, _lp), _p(idx, _ip),
                                _p(d2, _fp), total)
        return offs, idx[:total], d2[:total]


def bruteforce_nn(tgt, q):
    tgt, q = _f32(tgt, 3), _f32(q, 3)
    idx = np.empty(len(q), np.int32)
    d2 = np.empty(len(q), np.float32)
    lib().orc_bruteforce_nn(_p(tgt, _fp), len(tgt), _p(q, _fp), len(q), _p(idx, _ip), _p(d2, _fp))
    return idx, d2


def umeyama(src, tgt, acc_mode: int = 0) -> np.ndarray:
    """Returns the 4x4 as a numpy (4,4) array in MATH layout (T[r,c])."""
    src, tgt = _f32(src, 3), _f32(tgt, 3)
    T = np.empty(16, np.float32)
    rc = lib().orc_umeyama(_p(src, _fp), _p(tgt, _fp), len(src), acc_mode, _p(T, _fp))
    assert rc == 0
    return T.reshape(4, 4).T.copy()


def point_to_plane_lls(src, tgt, tgt_nrm) -> np.ndarray:
    src, tgt, tn = _f32(src, 3), _f32(tgt, 3), _f32(tgt_nrm, 3)
    T = np.empty(16, np.float32)
    rc = lib().orc_point_to_plane_lls(_p(src, _fp), _p(tgt, _fp), _p(tn, _fp), len(src), _p(T, _fp))
    assert rc == 0
    return T.reshape(4, 4).T.copy()


def point_to_plane_lm(src, tgt, tgt_nrm, precision: int = 0):
    """PCL's TransformationEstimationPointToPlane (LM).  Returns (T (4,4), x (6,), nfev, status)."""
    src, tgt, tn = _f32(src, 3), _f32(tgt, 3), _f32(tgt_nrm, 3)
    T = np.empty(16, np.float32)
    x = np.zeros(6, np.float64)
    nfev = C.c_int(0); status = C.c_int(0)
    rc = lib().orc_point_to_plane_lm(_p(src, _fp), _p(tgt, _fp), _p(tn, _fp), len(src), precision, _p(T, _fp), _p(x, _dp),
                                     C.byref(nfev), C.byref(status))
    if rc != 0:
        raise ValueError("orc_point_to_plane_lm needs at least 4 pairs")
    return T.reshape(4, 4).T.copy(), x, nfev.value, status.value


def umeyama_from_sums(S, pivot) -> np.ndarray:
    S = np.ascontiguousarray(S, np.float64)
    pivot = np.ascontiguousarray(pivot, np.float64)
    T = np.empty(16, np.float32)
    rc = lib().orc_umeyama_from_sums(_p(S, _dp), _p(pivot, _dp), _p(T, _fp))
    assert rc == 0
    return T.reshape(4, 4).T.copy()


def svd3(A):
    A = np.ascontiguousarray(A, np.float64)
    U = np.empty((3, 3)); s = np.empty(3); V = np.empty((3, 3))
    lib().orc_svd3(_p(A, _dp), _p(U, _dp), _p(s, _dp), _p(V, _dp))
    return U, s, V


def colmajor(T) -> np.ndarray:
    """(4,4) math-layout matrix -> column-major float[16]."""
    return np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(16)


def default_icp_params() -> IcpParams:
    p = IcpParams()
    lib().orc_icp_default_params(C.byref(p))
    return p


@dataclass
class IcpOut:
    T: np.ndarray
    iterations: int
    converged: bool
    state: int
    last_mse: float
    n_corr: int
    fitness: float
    align_strength: float
    T_hist: np.ndarray
    corr_q: np.ndarray
    corr_m: np.ndarray
    corr_d2: np.ndarray


def icp(src, tgt, params: IcpParams | None = None, guess=None, src_nrm=None, tgt_nrm=None, fixed=None, n_threads: int = 1) -> IcpOut:
    """fixed: optional (index_query[], index_match[]) — the reference's setFixedCorrespondences (icp_mod.h:268), see icp.c.
    n_threads: threads of the per-query searches only (orc_icp_set_threads); the result does not depend on it."""
    src, tgt = _f32(src, 3), _f32(tgt, 3)
    p = params or default_icp_params()
    sn = _f32(src_nrm, 3) if src_nrm is not None else None
    tn = _f32(tgt_nrm, 3) if tgt_nrm is not None else None
    g = colmajor(guess) if guess is not None else None
    T = np.empty(16, np.float32)
    res = IcpResult()
    hist = np.zeros((max(p.max_iterations, 1), 16), np.float32)
    fq = np.ascontiguousarray(fixed[0], np.int32) if fixed is not None else np.empty(0, np.int32)
    fm = np.ascontiguousarray(fixed[1], np.int32) if fixed is not None else np.empty(0, np.int32)
    assert len(fq) == len(fm)
    cap = len(src) + 2 * len(fq)
    cq = np.empty(cap, np.int32); cm = np.empty(cap, np.int32); cd = np.empty(cap, np.float32)
    lib().orc_icp_set_threads(int(n_threads))
    try:
        rc = lib().orc_icp_fixed(_p(src, _fp), _p(sn, _fp), len(src), _p(tgt, _fp), _p(tn, _fp), len(tgt), _p(g, _fp),
                                 C.byref(p), _p(fq, _ip) if len(fq) else None, _p(fm, _ip) if len(fm) else None, len(fq), _p(T, _fp),
                                 C.byref(res), _p(hist, _fp), _p(cq, _ip), _p(cm, _ip), _p(cd, _fp))
    finally:
        lib().orc_icp_set_threads(1)
    if rc != 0:
        raise ValueError(f"orc_icp rc={rc}")
    n = res.n_corr
    return IcpOut(T.reshape(4, 4).T.copy(), res.iterations, bool(res.converged), res.state, res.last_mse, n,
                  res.fitness, res.align_strength,
                  hist[: res.iterations].reshape(-1, 4, 4).transpose(0, 2, 1).copy(), cq[:n], cm[:n], cd[:n])


def fitness(src, tgt, T, max_range: float = np.finfo(np.float64).max):
    src, tgt = _f32(src, 3), _f32(tgt, 3)
    n = C.c_int(0)
    t = colmajor(T)
    v = lib().orc_fitness(_p(src, _fp), len(src), _p(tgt, _fp), len(tgt), _p(t, _fp), max_range, C.byref(n))
    return v, n.value


def icp_partial_sums(src, tree: KdTree, T, max_corr_dist, pivot, n_threads: int = 1) -> np.ndarray:
    src = _f32(src, 3)
    t = colmajor(T)
    pv = np.ascontiguousarray(pivot, np.float64)
    S = np.empty(17, np.float64)
    if n_threads > 1:
        lib().orc_icp_partial_sums_mt(_p(src, _fp), len(src), tree.h, _p(tree.xyz, _fp), _p(t, _fp), max_corr_dist,
                                      _p(pv, _dp), int(n_threads), _p(S, _dp))
    else:
        lib().orc_icp_partial_sums(_p(src, _fp), len(src), tree.h, _p(tree.xyz, _fp), _p(t, _fp), max_corr_dist,
                                   _p(pv, _dp), _p(S, _dp))
    return S


def transform_points(xyz, T):
    xyz = _f32(xyz, 3)
    out = np.empty_like(xyz)
    t = colmajor(T)
    lib().orc_transform_points(_p(xyz, _fp), len(xyz), _p(t, _fp), _p(out, _fp))
    return out


def normals_knn(xyz, k: int = 30, vp=(0.0, 0.0, 0.0)):
    xyz = _f32(xyz, 3)
    v = np.asarray(vp, np.float32)
    nrm = np.empty_like(xyz)
    curv = np.empty(len(xyz), np.float32)
    lib().orc_normals_knn(_p(xyz, _fp), len(xyz), k, _p(v, _fp), _p(nrm, _fp), _p(curv, _fp))
    return nrm, curv


def pair_features(p1, n1, p2, n2):
    a = [np.asarray(x, np.float32) for x in (p1, n1, p2, n2)]
    f = [C.c_float() for _ in range(4)]
    ok = lib().orc_pair_features(*[_p(x, _fp) for x in a], *[C.byref(x) for x in f])
    return bool(ok), tuple(x.value for x in f)


def fpfh(xyz, nrm, radius: float):
    xyz, nrm = _f32(xyz, 3), _f32(nrm, 3)
    out = np.empty((len(xyz), 33), np.float32)
    spfh = np.empty((len(xyz), 33), np.float32)
    mean_nb = C.c_double(0)
    lib().orc_fpfh(_p(xyz, _fp), _p(nrm, _fp), len(xyz), radius, _p(out, _fp), _p(spfh, _fp), C.byref(mean_nb))
    return out, spfh, mean_nb.value


def uniform_sampling(xyz, leaf: float) -> np.ndarray:
    xyz = _f32(xyz, 3)
    out = np.empty(len(xyz), np.int32)
    n = lib().orc_uniform_sampling(_p(xyz, _fp), len(xyz), leaf, _p(out, _ip))
    return out[:n].copy()


def remove_nan(xyz) -> np.ndarray:
    """pcl::removeNaNFromPointCloud: indices of the finite points, input order."""
    xyz = _f32(xyz, 3)
    out = np.empty(max(len(xyz), 1), np.int32)
    n = lib().orc_remove_nan(_p(xyz, _fp), len(xyz), _p(out, _ip))
    return out[:n].copy()


def pass_through(xyz, lo, hi) -> np.ndarray:
    """pcl::PassThrough on x, y and z with inclusive limits: indices of the survivors, input order."""
    xyz = _f32(xyz, 3)
    lo = np.ascontiguousarray(lo, np.float32); hi = np.ascontiguousarray(hi, np.float32)
    out = np.empty(max(len(xyz), 1), np.int32)
    n = lib().orc_pass_through(_p(xyz, _fp), len(xyz), _p(lo, _fp), _p(hi, _fp), _p(out, _ip))
    return out[:n].copy()


def voxel_grid(xyz, leaf, rgb=None):
    """pcl::VoxelGrid centroids in ascending voxel index; None when PCL would refuse the leaf size.
    rgb (optional, n uint32: the bits of PointXYZRGB::rgb): also the per-voxel colours -> (centroids, colours)."""
    xyz = _f32(xyz, 3)
    lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, np.float32), (3,)))
    out = np.empty((max(len(xyz), 1), 3), np.float32)
    if rgb is None:
        n = lib().orc_voxel_grid(_p(xyz, _fp), len(xyz), _p(lf, _fp), _p(out, _fp))
        return None if n < 0 else out[:n].copy()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint32)
    assert rgb.shape == (len(xyz),)
    oc = np.zeros(max(len(xyz), 1), np.uint32)
    n = lib().orc_voxel_grid_rgb(_p(xyz, _fp), rgb.ctypes.data, len(xyz), _p(lf, _fp), _p(out, _fp), oc.ctypes.data)
    return None if n < 0 else (out[:n].copy(), oc[:n].copy())


def statistical_outlier_removal(xyz, mean_k: int = 30, stddev_mul: float = 1.0, return_distances: bool = False):
    """pcl::StatisticalOutlierRemoval: indices of the inliers, input order (+ the mean-distance vector if asked)."""
    xyz = _f32(xyz, 3)
    out = np.empty(max(len(xyz), 1), np.int32)
    dist = np.zeros(max(len(xyz), 1), np.float32)
    n = lib().orc_statistical_outlier_removal(_p(xyz, _fp), len(xyz), mean_k, stddev_mul, _p(out, _ip), _p(dist, _fp))
    return (out[:n].copy(), dist[: len(xyz)].copy()) if return_distances else out[:n].copy()


def feature_knn(feat, q, k: int):
    feat, q = _f32(feat, 33), _f32(q, 33)
    idx = np.empty((len(q), k), np.int32)
    d2 = np.empty((len(q), k), np.float32)
    lib().orc_feature_knn(_p(feat, _fp), len(feat), _p(q, _fp), len(q), k, _p(idx, _ip), _p(d2, _fp))
    return idx, d2


def sacia_error(src, tree: KdTree, T, thr: float) -> float:
    src = _f32(src, 3)
    t = colmajor(T)
    return lib().orc_sacia_error(_p(src, _fp), len(src), tree.h, _p(t, _fp), thr)


def sacia(src, src_feat, tgt, tgt_feat, n_iter=400, nr_samples=5, k_corr=5, max_corr_dist=0.05,
          min_sample_dist=0.01, seed=1, forced_samples=None):
    src, tgt = _f32(src, 3), _f32(tgt, 3)
    sf, tf = _f32(src_feat, 33), _f32(tgt_feat, 33)
    fs = np.ascontiguousarray(forced_samples, np.int32) if forced_samples is not None else None
    T = np.empty(16, np.float32)
    err = C.c_double(0)
    bi = C.c_int32(-1)
    rc = lib().orc_sacia(_p(src, _fp), _p(sf, _fp), len(src), _p(tgt, _fp), _p(tf, _fp), len(tgt), n_iter,
                         nr_samples, k_corr, max_corr_dist, min_sample_dist, seed, _p(fs, _ip), _p(T, _fp),
                         C.byref(err), C.byref(bi))
    if rc != 0:
        raise ValueError(f"orc_sacia rc={rc}")
    return T.reshape(4, 4).T.copy(), err.value, bi.value


class PoseEstimator:
    """The reference's PoseEstimator (poseestimator.cpp:3-448) on the oracle's primitives: state crosses frames."""

    def __init__(self, sacia_seed: int = 1, use_self_occluded: bool = False, as_device: bool = True):
        self.st = PoseEstimatorState()
        lib().orc_pose_estimator_init(C.byref(self.st))
        self.st.sacia_seed = sacia_seed
        self.st.use_self_occluded = int(use_self_occluded)
        self.st.acc_mode = 1 if as_device else 0
        self.st.transform_mode = 1 if as_device else 0

    def __del__(self):
        try:
            lib().orc_pose_estimator_free(C.byref(self.st))
        except Exception:
            pass

    @staticmethod
    def _m(a16):
        return np.asarray(list(a16), np.float32).reshape(4, 4).T.copy()

    def estimate_final_pose(self, source, target):
        """Returns (finalPose (4,4), fitnessScore, alignStrength, source overwritten with alignedSource, info dict)."""
        src = _f32(source, 3).copy()
        tgt = _f32(target, 3)
        T = np.empty(16, np.float32)
        fit = C.c_double(0); strength = C.c_double(0)
        rc = lib().orc_estimate_final_pose(C.byref(self.st), _p(src, _fp), len(src), _p(tgt, _fp), len(tgt), _p(T, _fp),
                                           C.byref(fit), C.byref(strength))
        if rc != 0:
            raise ValueError(f"orc_estimate_final_pose rc={rc}")
        st = self.st
        info = {"coarse": self._m(st.last_coarse), "fine": self._m(st.last_fine), "rigid": self._m(st.last_rigid),
                "sacia_error": st.last_sacia_error, "sacia_best": st.last_sacia_best,
                "n_src_keys": st.last_n_src_keys, "n_tgt_keys": st.last_n_tgt_keys,
                "n_fine_src": st.last_n_fine_src, "n_fine_tgt": st.last_n_fine_tgt,
                "icp_iterations": st.last_icp_iterations, "icp_state": st.last_icp_state, "icp_n_corr": st.last_icp_n_corr,
                "coarse_calls": st.coarse_calls}
        return T.reshape(4, 4).T.copy(), fit.value, strength.value, src, info


def estimate_coarse_pose(source, target, sacia_seed: int = 1, call_index: int = 0):
    """PoseEstimator::estimateCoarsePose (poseestimator.cpp:16-73) as the `call_index`-th coarse call of an estimator
    seeded with `sacia_seed`: returns (pose (4,4), info)."""
    st = PoseEstimatorState()
    lib().orc_pose_estimator_init(C.byref(st))
    st.sacia_seed = sacia_seed
    st.coarse_calls = call_index
    src, tgt = _f32(source, 3), _f32(target, 3)
    T = np.empty(16, np.float32)
    rc = lib().orc_estimate_coarse_pose(C.byref(st), _p(src, _fp), len(src), _p(tgt, _fp), len(tgt), _p(T, _fp))
    info = {"sacia_error": st.last_sacia_error, "sacia_best": st.last_sacia_best, "n_src_keys": st.last_n_src_keys,
            "n_tgt_keys": st.last_n_tgt_keys}
    lib().orc_pose_estimator_free(C.byref(st))
    if rc != 0:
        raise ValueError(f"orc_estimate_coarse_pose rc={rc}")
    return T.reshape(4, 4).T.copy(), info
