"""object-pose-estimation_amd — MI355X-native registration hot path (ICP + FPFH/SAC-IA).

Thin ctypes binding of the C ABI in include/ope.h (libope_hip.so, hand-written HIP for gfx950).
This package is plumbing for tests, bench.py and the torch.distributed driver; the product is
the shared library and the C++ façade in include/ope/.  There is NO CPU fallback: every entry
point raises if the library or a GPU is missing.

The directory name has a hyphen, so import it with
    importlib.import_module("object-pose-estimation_amd")
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libope_hip.so")

OPE_OK, OPE_EINVAL, OPE_ENODEV, OPE_EHIP, OPE_ENOMEM, OPE_ESTATE, OPE_ECOMM, OPE_EEMPTY, OPE_ERANGE = 0, -1, -2, -3, -4, -5, -6, -7, -8
CONV_NAMES = ["NOT_CONVERGED", "ITERATIONS", "TRANSFORM", "ABS_MSE", "REL_MSE", "NO_CORRESPONDENCES"]
CORR_NEAREST, CORR_NORMAL_SHOOTING = 0, 1
EST_SVD, EST_POINT_TO_PLANE_LLS, EST_POINT_TO_PLANE_LM = 0, 1, 2
COMM_AUTO, COMM_RCCL, COMM_P2P = 0, 1, 2
CERT_AUTO, CERT_OFF, CERT_ALWAYS = 0, 1, 2   # ope_icp_params.skip_certificates
NUM_SUMS, NUM_SUMS_MAX = 17, 44
COMM_ID_BYTES = 128


class OpeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"ope error {code}: {msg}")
        self.code = code


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 build of libope_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs += [os.path.join(_HERE, "..", "include", "ope.h"), os.path.join(_HERE, "Makefile")]
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-j8", "libope_hip.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("euclidean_fitness_epsilon", C.c_double),
        ("max_corr_dist", C.c_double),
        ("min_correspondences", C.c_int),
        ("use_reciprocal", C.c_int),
        ("corr_mode", C.c_int),
        ("k_normal_shooting", C.c_int),
        ("use_surface_normal_rej", C.c_int),
        ("surface_normal_thr", C.c_double),
        ("use_self_occluded_rej", C.c_int),
        ("self_occluded_thr", C.c_double),
        ("mse_threshold_absolute", C.c_double),
        ("failure_after_max_iter", C.c_int),
        ("check_every", C.c_int),
        ("estimator", C.c_int),
        ("deterministic_sums", C.c_int),
        ("tree_walk", C.c_int),
        ("update_launch", C.c_int),
        ("skip_certificates", C.c_int),
    ]


class IcpResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("state", C.c_int),
        ("last_mse", C.c_double),
        ("n_corr", C.c_int64),
        ("align_strength", C.c_double),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_double), ("launches", C.c_int), ("algorithmic_bytes", C.c_double)]


class IndexParams(C.Structure):
    _fields_ = [("leaf_size", C.c_int), ("grid", C.c_int), ("grid_fill", C.c_float), ("grid_max_cells", C.c_int)]


class SaciaParams(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int),
        ("nr_samples", C.c_int),
        ("k_correspondences", C.c_int),
        ("max_corr_dist", C.c_double),
        ("min_sample_dist", C.c_float),
        ("seed", C.c_uint64),
    ]


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/ope.h declares: (name, restype, argtypes)
ABI = [
    ("ope_abi_version", C.c_int, []),
    ("ope_device_count", C.c_int, []),
    ("ope_ctx_create", C.c_int, [C.POINTER(_vp), C.c_int]),
    ("ope_ctx_destroy", None, [_vp]),
    ("ope_ctx_set_stream", C.c_int, [_vp, _vp]),
    ("ope_ctx_sync", C.c_int, [_vp]),
    ("ope_ctx_set_tracing", C.c_int, [_vp, C.c_int]),
    ("ope_ctx_set_wait_limit", C.c_int, [_vp, C.c_double]),
    ("ope_last_error", C.c_char_p, [_vp]),
    ("ope_profile_kernels", C.c_int, [_vp, C.c_int]),
    ("ope_profile_kernels_read", C.c_int, [_vp, C.POINTER(KernelTime), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ope_cloud_upload", C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_ssize_t, C.POINTER(_vp)]),
    ("ope_cloud_set_normals", C.c_int, [_vp, _vp, _fp]),
    ("ope_cloud_concat", C.c_int, [_vp, _vp, _fp, _vp, C.POINTER(_vp)]),
    ("ope_cloud_download", C.c_int, [_vp, _vp, _fp]),
    ("ope_cloud_size", C.c_size_t, [_vp]),
    ("ope_cloud_free", None, [_vp]),
    ("ope_index_default_params", None, [C.POINTER(IndexParams)]),
    ("ope_index_build", C.c_int, [_vp, _vp, C.POINTER(IndexParams), C.POINTER(_vp)]),
    ("ope_index_free", None, [_vp]),
    ("ope_nn_search", C.c_int, [_vp, _vp, _vp, _fp, _ip, _fp]),
    ("ope_knn_search", C.c_int, [_vp, _vp, _vp, _fp, C.c_int, _ip, _fp]),
    ("ope_radius_search", C.c_int, [_vp, _vp, _vp, C.c_float, C.c_int, _ip, _ip, _fp]),
    ("ope_icp_default_params", None, [C.POINTER(IcpParams)]),
    ("ope_icp_run", C.c_int, [_vp, _vp, _vp, _fp, C.POINTER(IcpParams), _fp, C.POINTER(IcpResult)]),
    ("ope_icp_begin", C.c_int, [_vp, _vp, _vp, _fp, C.POINTER(IcpParams)]),
    ("ope_icp_accumulate", C.c_int, [_vp]),
    ("ope_icp_sums_device", _vp, [_vp]),
    ("ope_icp_update", C.c_int, [_vp]),
    ("ope_icp_set_sums_buffer", C.c_int, [_vp, _vp]),
    ("ope_icp_iterate", C.c_int, [_vp, C.c_int]),
    ("ope_icp_profile", C.c_int, [_vp, C.c_int]),
    ("ope_icp_profile_read", C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    ("ope_icp_poll", C.c_int, [_vp, C.POINTER(IcpResult)]),
    ("ope_icp_current_transform", C.c_int, [_vp, _fp]),
    ("ope_icp_end", C.c_int, [_vp, _fp, C.POINTER(IcpResult)]),
    ("ope_icp_set_global_sizes", C.c_int, [_vp, C.c_int64, C.c_int64]),
    ("ope_icp_kernel_launches", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("ope_icp_overlapped_updates", C.c_int64, [_vp]),
    ("ope_icp_certificate_stats", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("ope_icp_set_fixed_correspondences", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t]),
    ("ope_icp_update_fallbacks", C.c_int, [_vp]),
    ("ope_icp_fixed_correspondences", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ope_icp_profile_launches", C.c_int, [_vp, _fp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ope_cloud_select", C.c_int, [_vp, _vp, _ip, C.c_size_t, C.POINTER(_vp)]),
    ("ope_remove_nan_cloud", C.c_int, [_vp, _vp, C.POINTER(_vp), _ip, C.POINTER(C.c_size_t)]),
    ("ope_pass_through_cloud", C.c_int, [_vp, _vp, _fp, _fp, C.POINTER(_vp), _ip, C.POINTER(C.c_size_t)]),
    ("ope_statistical_outlier_removal_cloud", C.c_int, [_vp, _vp, C.c_int, C.c_double, C.POINTER(_vp), _ip, C.POINTER(C.c_size_t)]),
    ("ope_uniform_sampling_cloud", C.c_int, [_vp, _vp, C.c_float, C.POINTER(_vp), _ip, C.POINTER(C.c_size_t)]),
    ("ope_icp_correspondences", C.c_int, [_vp, _ip, _ip, _fp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("ope_icp_last_incremental", C.c_int, [_vp, _fp]),
    ("ope_reject_pairs", C.c_int, [_vp, C.c_int, _fp, _fp, C.c_size_t, C.c_double, C.POINTER(C.c_ubyte)]),
    ("ope_fitness", C.c_int, [_vp, _vp, _vp, _fp, C.c_double, _dp, _dp, C.POINTER(C.c_int64)]),
    ("ope_rigid_transform_svd", C.c_int, [_vp, _fp, _fp, C.c_size_t, _fp]),
    ("ope_transform_cloud", C.c_int, [_vp, _vp, _fp, _fp]),
    ("ope_comm_get_unique_id", C.c_int, [C.c_char_p]),
    ("ope_comm_init_rank", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    ("ope_comm_destroy", C.c_int, [_vp]),
    ("ope_comm_set_transport", C.c_int, [_vp, C.c_int]),
    ("ope_comm_transport", C.c_int, [_vp]),
    ("ope_comm_p2p_open", C.c_int, [_vp, C.c_char_p]),
    ("ope_comm_p2p_connect", C.c_int, [_vp, C.c_char_p, C.c_int, C.c_int]),
    ("ope_normals", C.c_int, [_vp, _vp, C.c_int, _fp, _fp, _fp]),
    ("ope_normals_from", C.c_int, [_vp, _vp, _vp, C.c_int, _fp, _fp, _fp]),
    ("ope_fpfh", C.c_int, [_vp, _vp, C.c_float, _fp]),
    ("ope_uniform_sampling", C.c_int, [_vp, _vp, C.c_float, _ip, C.POINTER(C.c_size_t)]),
    ("ope_remove_nan", C.c_int, [_vp, _vp, _ip, C.POINTER(C.c_size_t)]),
    ("ope_pass_through", C.c_int, [_vp, _vp, _fp, _fp, _ip, C.POINTER(C.c_size_t)]),
    ("ope_voxel_grid", C.c_int, [_vp, _vp, _fp, _fp, C.POINTER(C.c_size_t)]),
    ("ope_voxel_grid_rgb", C.c_int, [_vp, _vp, _fp, _vp, _fp, _vp, C.POINTER(C.c_size_t)]),
    ("ope_statistical_outlier_removal", C.c_int, [_vp, _vp, C.c_int, C.c_double, _ip, C.POINTER(C.c_size_t), _fp]),
    ("ope_sacia_default_params", None, [C.POINTER(SaciaParams)]),
    ("ope_sacia", C.c_int, [_vp, _vp, _fp, _vp, _vp, _fp, C.POINTER(SaciaParams), _ip, _fp, _dp, _ip]),
]

_lib = None


def lib() -> C.CDLL:
    """Load libope_hip.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OpeError(OPE_ENODEV, f"{LIB_PATH} is missing: run __graft_entry__.build() / make -C {_HERE}")
        # torch ships its own libamdhip64.so.7 / librccl.so.1; importing it first makes this library
        # bind to the same HIP runtime (same SONAME), so streams and device pointers are shareable.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for pure C-ABI use
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _f32(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if cols is not None and not (a.ndim == 2 and a.shape[1] == cols):
        raise ValueError(f"expected (n,{cols}) float array, got {a.shape}")
    return a


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def colmajor(T) -> np.ndarray:
    """(4,4) math-layout matrix -> column-major float[16] (Eigen::Matrix4f memory order)."""
    return np.ascontiguousarray(np.asarray(T, np.float32).T).reshape(16)


def from_colmajor(t16) -> np.ndarray:
    return np.asarray(t16, np.float32).reshape(4, 4).T.copy()


def default_icp_params(**kw) -> IcpParams:
    p = IcpParams()
    lib().ope_icp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


@dataclass
class IcpOut:
    T: np.ndarray          # (4,4) math layout, float32 — getFinalTransformation()
    iterations: int
    converged: bool
    state: int
    last_mse: float
    n_corr: int
    align_strength: float


class Context:
    """One ope_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        h = _vp()
        rc = lib().ope_ctx_create(C.byref(h), device)
        if rc != OPE_OK:
            raise OpeError(rc, (lib().ope_last_error(None) or b"").decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            lib().ope_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != OPE_OK:
            raise OpeError(rc, (lib().ope_last_error(self.h) or b"").decode())

    def set_stream(self, stream_ptr: int | None):
        self._chk(lib().ope_ctx_set_stream(self.h, _vp(stream_ptr) if stream_ptr else None))

    def sync(self):
        self._chk(lib().ope_ctx_sync(self.h))

    def profile_kernels(self, on: bool):
        """HIP-event brackets around every coarse-stage / filter kernel launch (ope_profile_kernels)."""
        self._chk(lib().ope_profile_kernels(self.h, int(on)))

    def profile_kernels_read(self) -> dict:
        """{kernel name: {"ms", "launches", "algorithmic_bytes"}} summed since profile_kernels(True)."""
        buf = (KernelTime * 32)()
        n = C.c_size_t(0)
        self._chk(lib().ope_profile_kernels_read(self.h, buf, 32, C.byref(n)))
        return {buf[i].name.decode(): {"ms": buf[i].ms, "launches": buf[i].launches, "algorithmic_bytes": buf[i].algorithmic_bytes}
                for i in range(min(n.value, 32))}

    def set_wait_limit(self, seconds: float):
        """Bound of the device-side waits of the overlapped update launches (ope_ctx_set_wait_limit)."""
        self._chk(lib().ope_ctx_set_wait_limit(self.h, float(seconds)))

    def set_tracing(self, on: bool):
        self._chk(lib().ope_ctx_set_tracing(self.h, int(on)))

    # ---- clouds / index
    def upload(self, xyz, normals=None) -> "Cloud":
        xyz = _f32(xyz, 3)
        h = _vp()
        self._chk(lib().ope_cloud_upload(self.h, xyz.ctypes.data_as(_vp), len(xyz), 12, 0, -1, C.byref(h)))
        c = Cloud(self, h, len(xyz))
        if normals is not None:
            c.set_normals(normals)
        return c

    def upload_struct(self, buf: np.ndarray, stride: int, xyz_off: int, normal_off: int = -1) -> "Cloud":
        """Upload from an array-of-structs byte buffer (e.g. pcl::PointXYZRGBNormal, stride 48)."""
        buf = np.ascontiguousarray(buf)
        n = buf.nbytes // stride
        h = _vp()
        self._chk(lib().ope_cloud_upload(self.h, buf.ctypes.data_as(_vp), n, stride, xyz_off, normal_off, C.byref(h)))
        return Cloud(self, h, n)

    def concat(self, a: "Cloud", T, b: "Cloud") -> "Cloud":
        """[T * a ; b] built on the device (BuildModel's cloudTemp = aligned + target)."""
        t = colmajor(T) if T is not None else None
        h = _vp()
        self._chk(lib().ope_cloud_concat(self.h, a.h, _p(t, _fp), b.h, C.byref(h)))
        return Cloud(self, h, a.n + b.n)

    def select(self, cloud: "Cloud", idx) -> "Cloud":
        """cloud[idx] as a new device-resident cloud (gathered on the device; normals carried)."""
        idx = np.ascontiguousarray(idx, np.int32)
        h = _vp()
        self._chk(lib().ope_cloud_select(self.h, cloud.h, _p(idx, _ip), len(idx), C.byref(h)))
        return Cloud(self, h, len(idx))

    def _filter_cloud(self, fn, cloud, args, want_idx):
        """device-resident form of a filter: (new Cloud, indices or None)"""
        out = np.empty(max(cloud.n, 1), np.int32) if want_idx else None
        n = C.c_size_t(0)
        h = _vp()
        self._chk(fn(self.h, cloud.h, *args, C.byref(h), _p(out, _ip), C.byref(n)))
        return Cloud(self, h, n.value), (out[: n.value].copy() if want_idx else None)

    def remove_nan_cloud(self, cloud: "Cloud", want_idx: bool = False):
        return self._filter_cloud(lib().ope_remove_nan_cloud, cloud, (), want_idx)

    def pass_through_cloud(self, cloud: "Cloud", lo, hi, want_idx: bool = False):
        lo = np.ascontiguousarray(lo, np.float32); hi = np.ascontiguousarray(hi, np.float32)
        if lo.shape != (3,) or hi.shape != (3,):
            raise ValueError("pass_through: lo and hi are 3-vectors")
        return self._filter_cloud(lib().ope_pass_through_cloud, cloud, (_p(lo, _fp), _p(hi, _fp)), want_idx)

    def statistical_outlier_removal_cloud(self, cloud: "Cloud", mean_k: int = 30, stddev_mul: float = 1.0, want_idx: bool = False):
        return self._filter_cloud(lib().ope_statistical_outlier_removal_cloud, cloud, (mean_k, stddev_mul), want_idx)

    def uniform_sampling_cloud(self, cloud: "Cloud", leaf: float, want_idx: bool = False):
        return self._filter_cloud(lib().ope_uniform_sampling_cloud, cloud, (leaf,), want_idx)

    def download(self, cloud: "Cloud") -> np.ndarray:
        out = np.empty((cloud.n, 3), np.float32)
        self._chk(lib().ope_cloud_download(self.h, cloud.h, _p(out, _fp)))
        return out

    def build_index(self, cloud: "Cloud", leaf_size: int | None = None, grid: bool | None = None, grid_fill: float = 0.0,
                    grid_max_cells: int = 0) -> "Index":
        p = IndexParams()
        lib().ope_index_default_params(C.byref(p))
        if leaf_size:
            p.leaf_size = leaf_size
        if grid is not None:
            p.grid = int(grid)
        p.grid_fill = grid_fill
        p.grid_max_cells = grid_max_cells
        h = _vp()
        self._chk(lib().ope_index_build(self.h, cloud.h, C.byref(p), C.byref(h)))
        return Index(self, h, cloud)

    # ---- searches
    def nn(self, queries: "Cloud", index: "Index", T=None):
        n = queries.n
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        t = colmajor(T) if T is not None else None
        self._chk(lib().ope_nn_search(self.h, queries.h, index.h, _p(t, _fp), _p(idx, _ip), _p(d2, _fp)))
        return idx, d2

    def knn(self, queries: "Cloud", index: "Index", k: int, T=None):
        n = queries.n
        idx = np.empty((n, k), np.int32)
        d2 = np.empty((n, k), np.float32)
        t = colmajor(T) if T is not None else None
        self._chk(lib().ope_knn_search(self.h, queries.h, index.h, _p(t, _fp), k, _p(idx, _ip), _p(d2, _fp)))
        return idx, d2

    def radius(self, queries: "Cloud", index: "Index", radius: float, max_nn: int = 0):
        n = queries.n
        counts = np.empty(n, np.int32)
        idx = np.empty((n, max_nn), np.int32) if max_nn else None
        d2 = np.empty((n, max_nn), np.float32) if max_nn else None
        self._chk(lib().ope_radius_search(self.h, queries.h, index.h, radius, max_nn, _p(counts, _ip), _p(idx, _ip),
                                          _p(d2, _fp)))
        return counts, idx, d2

    # ---- ICP
    def icp(self, src: "Cloud", tgt: "Index", params: IcpParams | None = None, guess=None) -> IcpOut:
        p = params or default_icp_params()
        g = colmajor(guess) if guess is not None else None
        T = np.empty(16, np.float32)
        r = IcpResult()
        self._chk(lib().ope_icp_run(self.h, src.h, tgt.h if tgt is not None else None, _p(g, _fp), C.byref(p),
                                    _p(T, _fp), C.byref(r)))
        return IcpOut(from_colmajor(T), r.iterations, bool(r.converged), r.state, r.last_mse, r.n_corr, r.align_strength)

    def icp_begin(self, src: "Cloud", tgt: "Index", params: IcpParams | None = None, guess=None):
        p = params or default_icp_params()
        g = colmajor(guess) if guess is not None else None
        self._chk(lib().ope_icp_begin(self.h, src.h, tgt.h, _p(g, _fp), C.byref(p)))

    def icp_accumulate(self):
        self._chk(lib().ope_icp_accumulate(self.h))

    def icp_sums_ptr(self) -> int:
        return int(lib().ope_icp_sums_device(self.h) or 0)

    def icp_update(self):
        self._chk(lib().ope_icp_update(self.h))

    def icp_set_sums_buffer(self, device_ptr: int | None):
        self._chk(lib().ope_icp_set_sums_buffer(self.h, _vp(device_ptr) if device_ptr else None))

    def icp_iterate(self, n: int = 1):
        self._chk(lib().ope_icp_iterate(self.h, n))

    def icp_profile(self, max_launches: int):
        self._chk(lib().ope_icp_profile(self.h, max_launches))

    def icp_profile_read(self):
        ms = C.c_double(0); n = C.c_int(0)
        self._chk(lib().ope_icp_profile_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def icp_poll(self) -> IcpResult:
        r = IcpResult()
        self._chk(lib().ope_icp_poll(self.h, C.byref(r)))
        return r

    def icp_current_transform(self) -> np.ndarray:
        """(4,4) final transformation after the iterations enqueued so far (synchronises)."""
        T = np.empty(16, np.float32)
        self._chk(lib().ope_icp_current_transform(self.h, _p(T, _fp)))
        return from_colmajor(T)

    def icp_end(self) -> IcpOut:
        T = np.empty(16, np.float32)
        r = IcpResult()
        self._chk(lib().ope_icp_end(self.h, _p(T, _fp), C.byref(r)))
        return IcpOut(from_colmajor(T), r.iterations, bool(r.converged), r.state, r.last_mse, r.n_corr, r.align_strength)

    def icp_profile_launches(self, cap: int = 4096) -> np.ndarray:
        """HIP-event duration (ms) of each accumulate launch timed since icp_profile(n)."""
        ms = np.empty(cap, np.float32)
        n = C.c_size_t(0)
        self._chk(lib().ope_icp_profile_launches(self.h, _p(ms, _fp), cap, C.byref(n)))
        return ms[: n.value].copy()

    def icp_kernel_launches(self) -> dict:
        """Accumulate launches of the current / last run per search kernel: {'grid', 'tree_lane', 'tree_packet', 'knn'}."""
        c = (C.c_int64 * 4)()
        self._chk(lib().ope_icp_kernel_launches(self.h, c))
        return dict(zip(("grid", "tree_lane", "tree_packet", "knn"), (int(v) for v in c)))

    def icp_set_fixed_correspondences(self, src: "Cloud", tgt_cloud: "Cloud", index_query=None, index_match=None):
        """setFixedCorrespondences (icp_mod.h:268): pairs by original indices, for every later run over these clouds; no indices = clear."""
        q = np.ascontiguousarray(index_query if index_query is not None else [], np.int32)
        m = np.ascontiguousarray(index_match if index_match is not None else [], np.int32)
        assert len(q) == len(m)
        self._chk(lib().ope_icp_set_fixed_correspondences(self.h, src.h if src is not None else None, tgt_cloud.h if tgt_cloud is not None else None,
                                                          q.ctypes.data_as(C.POINTER(C.c_int32)), m.ctypes.data_as(C.POINTER(C.c_int32)), len(q)))

    def icp_fixed_correspondences(self):
        """The given pairs as the last iteration saw them: (distance field the reference writes back through the caller's pointer,
        listed in front of the searched pairs, appended behind them) — correspondence_estimation_mod.hpp:150-161, icp_mod.hpp:210-224."""
        n = C.c_size_t(0)
        self._chk(lib().ope_icp_fixed_correspondences(self.h, None, None, None, 0, C.byref(n)))
        d, a, b = np.zeros(n.value, np.float32), np.zeros(n.value, np.int32), np.zeros(n.value, np.int32)
        if n.value:
            self._chk(lib().ope_icp_fixed_correspondences(self.h, d.ctypes.data_as(C.POINTER(C.c_float)), a.ctypes.data_as(C.POINTER(C.c_int32)),
                                                          b.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n)))
        return d, a.astype(bool), b.astype(bool)

    def icp_overlapped_updates(self) -> int:
        """Update steps of the current / last run that were launched overlapped (ope_icp_params.update_launch)."""
        return int(lib().ope_icp_overlapped_updates(self.h))

    def icp_update_fallbacks(self) -> int:
        """Runs of this context that resumed in line after an overlapped update launch gave up its bounded wait (ope.h)."""
        return int(lib().ope_icp_update_fallbacks(self.h))

    def icp_certificate_stats(self) -> dict:
        """Skip certificates of the run in progress (ope_icp_params.skip_certificates): queries answered from their certificate
        (summed over the launches), launches that kept certificates, whether the run keeps them now, the last update's largest
        scene displacement in metres."""
        c = (C.c_int64 * 4)()
        self._chk(lib().ope_icp_certificate_stats(self.h, c))
        return {"certified": int(c[0]), "launches": int(c[1]), "on": bool(c[2]), "last_move": c[3] * 1e-9}

    def icp_set_global_sizes(self, n_src_total: int, n_tgt_total: int):
        self._chk(lib().ope_icp_set_global_sizes(self.h, n_src_total, n_tgt_total))

    def icp_correspondences(self, cap: int):
        q = np.empty(cap, np.int32); m = np.empty(cap, np.int32); d = np.empty(cap, np.float32)
        n = C.c_size_t(0)
        self._chk(lib().ope_icp_correspondences(self.h, _p(q, _ip), _p(m, _ip), _p(d, _fp), cap, C.byref(n)))
        k = min(n.value, cap)
        return q[:k], m[:k], d[:k]

    def icp_last_incremental(self) -> np.ndarray:
        T = np.empty(16, np.float32)
        self._chk(lib().ope_icp_last_incremental(self.h, _p(T, _fp)))
        return from_colmajor(T)

    def reject_pairs(self, kind: int, a, b, threshold: float) -> np.ndarray:
        """CorrespondenceRejector predicates on given pairs (0: surface normal a.b, 1: self-occluded a.(-b/|b|)): bool mask."""
        a, b = _f32(a, 3), _f32(b, 3)
        keep = np.zeros(len(a), np.uint8)
        self._chk(lib().ope_reject_pairs(self.h, kind, _p(a, _fp), _p(b, _fp), len(a), threshold, keep.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return keep.astype(bool)

    def fitness(self, src: "Cloud", tgt: "Index", T, max_range: float = float(np.finfo(np.float64).max)):
        t = colmajor(T)
        score = C.c_double(0); s = C.c_double(0); n = C.c_int64(0)
        self._chk(lib().ope_fitness(self.h, src.h, tgt.h, _p(t, _fp), max_range, C.byref(score), C.byref(s), C.byref(n)))
        return score.value, s.value, n.value

    def rigid_transform_svd(self, src_xyz, tgt_xyz) -> np.ndarray:
        a, b = _f32(src_xyz, 3), _f32(tgt_xyz, 3)
        if len(a) != len(b):
            raise ValueError("paired arrays must have equal length")
        T = np.empty(16, np.float32)
        self._chk(lib().ope_rigid_transform_svd(self.h, _p(a, _fp), _p(b, _fp), len(a), _p(T, _fp)))
        return from_colmajor(T)

    def transform_cloud(self, cloud: "Cloud", T) -> np.ndarray:
        out = np.empty((cloud.n, 3), np.float32)
        t = colmajor(T)
        self._chk(lib().ope_transform_cloud(self.h, cloud.h, _p(t, _fp), _p(out, _fp)))
        return out

    # ---- RCCL
    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        self._chk(lib().ope_comm_init_rank(self.h, unique_id, nranks, rank))

    def comm_destroy(self):
        self._chk(lib().ope_comm_destroy(self.h))

    def comm_p2p_open(self) -> bytes:
        """Allocate this rank's slot buffer; returns its 64-byte hipIpc handle (to be passed to every rank)."""
        buf = C.create_string_buffer(64)
        self._chk(lib().ope_comm_p2p_open(self.h, buf))
        return buf.raw

    def comm_p2p_connect(self, handles, rank: int):
        """Collective: map the peers' buffers (handles in rank order) and run the test exchange."""
        blob = b"".join(handles)
        assert len(blob) == 64 * len(handles)
        self._chk(lib().ope_comm_p2p_connect(self.h, blob, len(handles), rank))

    def comm_set_transport(self, transport: int):
        """COMM_AUTO (peer-to-peer slots if every rank set them up, else RCCL), COMM_RCCL, COMM_P2P (or an error)."""
        self._chk(lib().ope_comm_set_transport(self.h, int(transport)))

    def comm_transport(self) -> int:
        """What iterations of a sharded run will use: COMM_RCCL or COMM_P2P (0 without a communicator)."""
        return int(lib().ope_comm_transport(self.h))

    # ---- features
    def normals(self, cloud: "Cloud", k: int = 30, vp=(0.0, 0.0, 0.0), fetch: bool = True):
        """pcl::NormalEstimation (k-NN).  The normals stay attached to `cloud` on the device; fetch=False skips the
        copy back to the host and returns None."""
        v = np.asarray(vp, np.float32)
        if not fetch:
            self._chk(lib().ope_normals(self.h, cloud.h, k, _p(v, _fp), None, None))
            return None
        nrm = np.empty((cloud.n, 3), np.float32)
        curv = np.empty(cloud.n, np.float32)
        self._chk(lib().ope_normals(self.h, cloud.h, k, _p(v, _fp), _p(nrm, _fp), _p(curv, _fp)))
        return nrm, curv

    def normals_from(self, queries: "Cloud", index: "Index", k: int = 30, vp=(0.0, 0.0, 0.0), fetch: bool = True):
        """Normals of `queries` from their k nearest neighbours in `index` (setSearchSurface); attached to `queries`."""
        v = np.asarray(vp, np.float32)
        if not fetch:
            self._chk(lib().ope_normals_from(self.h, queries.h, index.h, k, _p(v, _fp), None, None))
            return None
        nrm = np.empty((queries.n, 3), np.float32)
        curv = np.empty(queries.n, np.float32)
        self._chk(lib().ope_normals_from(self.h, queries.h, index.h, k, _p(v, _fp), _p(nrm, _fp), _p(curv, _fp)))
        return nrm, curv

    def fpfh(self, cloud: "Cloud", radius: float) -> np.ndarray:
        out = np.empty((cloud.n, 33), np.float32)
        self._chk(lib().ope_fpfh(self.h, cloud.h, radius, _p(out, _fp)))
        return out

    def uniform_sampling(self, cloud: "Cloud", leaf: float) -> np.ndarray:
        out = np.empty(cloud.n, np.int32)
        n = C.c_size_t(0)
        self._chk(lib().ope_uniform_sampling(self.h, cloud.h, leaf, _p(out, _ip), C.byref(n)))
        return out[: n.value].copy()

    # ---- filters either side of the path
    def remove_nan(self, cloud: "Cloud") -> np.ndarray:
        """pcl::removeNaNFromPointCloud: original indices of the finite points, ascending."""
        out = np.empty(max(cloud.n, 1), np.int32)
        n = C.c_size_t(0)
        self._chk(lib().ope_remove_nan(self.h, cloud.h, _p(out, _ip), C.byref(n)))
        return out[: n.value].copy()

    def pass_through(self, cloud: "Cloud", lo, hi) -> np.ndarray:
        """pcl::PassThrough on x, y and z (inclusive limits): original indices of the survivors, ascending."""
        lo = np.ascontiguousarray(lo, np.float32); hi = np.ascontiguousarray(hi, np.float32)
        if lo.shape != (3,) or hi.shape != (3,):
            raise ValueError("pass_through: lo and hi are 3-vectors")
        out = np.empty(max(cloud.n, 1), np.int32)
        n = C.c_size_t(0)
        self._chk(lib().ope_pass_through(self.h, cloud.h, _p(lo, _fp), _p(hi, _fp), _p(out, _ip), C.byref(n)))
        return out[: n.value].copy()

    def voxel_grid(self, cloud: "Cloud", leaf, rgb=None):
        """pcl::VoxelGrid centroids (m,3) in ascending voxel index; OpeError(OPE_ERANGE) where PCL refuses the leaf.
        rgb (optional, n uint32 = the bits of PointXYZRGB::rgb, input order): also the voxels' colours -> (centroids, colours)."""
        lf = np.ascontiguousarray(np.broadcast_to(np.asarray(leaf, np.float32), (3,)))
        out = np.empty((max(cloud.n, 1), 3), np.float32)
        n = C.c_size_t(0)
        if rgb is None:
            self._chk(lib().ope_voxel_grid(self.h, cloud.h, _p(lf, _fp), _p(out, _fp), C.byref(n)))
            return out[: n.value].copy()
        rgb = np.ascontiguousarray(rgb, dtype=np.uint32)
        if rgb.shape != (cloud.n,):
            raise ValueError("rgb: one packed colour per input point")
        oc = np.zeros(max(cloud.n, 1), np.uint32)
        self._chk(lib().ope_voxel_grid_rgb(self.h, cloud.h, _p(lf, _fp), rgb.ctypes.data, _p(out, _fp), oc.ctypes.data, C.byref(n)))
        return out[: n.value].copy(), oc[: n.value].copy()

    def statistical_outlier_removal(self, cloud: "Cloud", mean_k: int = 30, stddev_mul: float = 1.0, return_distances: bool = False):
        """pcl::StatisticalOutlierRemoval (ProcessingPcd::getOutlierRemove): original indices of the inliers, ascending."""
        out = np.empty(max(cloud.n, 1), np.int32)
        dist = np.empty(max(cloud.n, 1), np.float32) if return_distances else None
        n = C.c_size_t(0)
        self._chk(lib().ope_statistical_outlier_removal(self.h, cloud.h, mean_k, stddev_mul, _p(out, _ip), C.byref(n), _p(dist, _fp)))
        return (out[: n.value].copy(), dist[: cloud.n].copy()) if return_distances else out[: n.value].copy()

    def sacia(self, src: "Cloud", src_feat, tgt: "Cloud", tgt_index: "Index", tgt_feat, params: SaciaParams | None = None,
              forced_samples=None):
        p = params or default_sacia_params()
        sf, tf = _f32(src_feat, 33), _f32(tgt_feat, 33)
        fs = np.ascontiguousarray(forced_samples, np.int32) if forced_samples is not None else None
        T = np.empty(16, np.float32)
        err = C.c_double(0); bi = C.c_int32(-1)
        self._chk(lib().ope_sacia(self.h, src.h, _p(sf, _fp), tgt.h, tgt_index.h, _p(tf, _fp), C.byref(p), _p(fs, _ip),
                                  _p(T, _fp), C.byref(err), C.byref(bi)))
        return from_colmajor(T), err.value, bi.value


def default_sacia_params(**kw) -> SaciaParams:
    p = SaciaParams()
    lib().ope_sacia_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib().ope_comm_get_unique_id(buf)
    if rc != OPE_OK:
        raise OpeError(rc, (lib().ope_last_error(None) or b"").decode())
    return buf.raw


class Cloud:
    def __init__(self, ctx: Context, h, n: int):
        self.ctx, self.h, self.n = ctx, h, n

    def set_normals(self, normals):
        nrm = _f32(normals, 3)
        if len(nrm) != self.n:
            raise ValueError("normals length mismatch")
        self.ctx._chk(lib().ope_cloud_set_normals(self.ctx.h, self.h, _p(nrm, _fp)))

    def free(self):
        if self.h:
            lib().ope_cloud_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class Index:
    def __init__(self, ctx: Context, h, cloud: Cloud):
        self.ctx, self.h, self.cloud = ctx, h, cloud

    def free(self):
        if self.h:
            lib().ope_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass
