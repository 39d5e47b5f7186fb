"""Property tests (hypothesis) for the CPU oracle: size-independent facts that hold for any input.

The oracle is pinned by known answers in test_oracle_kat.py; these add randomised structure checks:
exact k-NN equals brute force for any cloud (duplicates, tiny clouds), Umeyama is equivariant under rigid
motions, the voxel grid partitions its input, the filters commute with permutations of the input.
"""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle

clouds = st.integers(0, 2**32 - 1).flatmap(
    lambda seed: st.tuples(st.just(seed), st.integers(1, 300), st.sampled_from(["uniform", "dups", "line", "grid"])))


def make_cloud(seed, n, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    if kind == "dups":
        base = rng.uniform(-1, 1, (max(n // 4, 1), 3)).astype(np.float32)
        return base[rng.integers(0, len(base), n)]
    if kind == "line":
        t = rng.uniform(-1, 1, n).astype(np.float32)
        return np.stack([t, 0.5 * t, -t], axis=1).astype(np.float32)
    g = rng.integers(-3, 4, (n, 3)).astype(np.float32) * np.float32(0.125)
    return g


@settings(max_examples=60, deadline=None)
@given(clouds, st.integers(1, 8))
def test_knn_equals_brute_force(c, k):
    tgt = make_cloud(*c)
    rng = np.random.default_rng(c[0] + 1)
    q = rng.uniform(-1.2, 1.2, (40, 3)).astype(np.float32)
    kk = min(k, len(tgt))
    idx, d2, cnt = oracle.KdTree(tgt).knn(q, kk)
    diff = q[:, None, :] - tgt[None, :, :]
    bf = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]   # same fp32 order
    want = np.sort(bf, axis=1)[:, :kk]
    assert (cnt == kk).all()
    np.testing.assert_array_equal(d2, want)                      # distances bit-equal, ascending
    np.testing.assert_array_equal(np.take_along_axis(bf, idx.astype(np.int64), axis=1), d2)   # and they belong to the indices


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2**32 - 1), st.integers(4, 200))
def test_umeyama_recovers_and_is_equivariant(seed, n):
    rng = np.random.default_rng(seed)
    P = rng.normal(0, 1, (n, 3)).astype(np.float32)
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    if np.linalg.det(A) < 0:
        A[:, 0] = -A[:, 0]
    t = rng.uniform(-2, 2, 3)
    Q = (P.astype(np.float64) @ A.T + t).astype(np.float32)
    T = oracle.umeyama(P, Q, 1)
    np.testing.assert_allclose(T[:3, :3], A, atol=5e-5)
    np.testing.assert_allclose(T[:3, 3], t, atol=5e-5)
    np.testing.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-5)
    # moving both clouds by the same rigid motion conjugates the answer
    B = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    if np.linalg.det(B) < 0:
        B[:, 0] = -B[:, 0]
    P2 = (P.astype(np.float64) @ B.T).astype(np.float32); Q2 = (Q.astype(np.float64) @ B.T).astype(np.float32)
    T2 = oracle.umeyama(P2, Q2, 1)
    np.testing.assert_allclose(T2[:3, :3], B @ A @ B.T, atol=2e-4)


@settings(max_examples=40, deadline=None)
@given(clouds, st.floats(0.05, 0.7))
def test_voxel_grid_partitions_the_cloud(c, leaf):
    x = make_cloud(*c)
    leaf = np.float32(leaf)
    cen = oracle.voxel_grid(x, leaf)
    keys = np.floor(x * (np.float32(1) / leaf)).astype(np.int64)
    uniq, counts = np.unique(keys, axis=0, return_counts=True)
    assert len(cen) == len(uniq)                                  # one centroid per occupied voxel
    # every centroid lies in (the closure of) a distinct occupied voxel and the count-weighted mean is the cloud's mean
    ck = np.floor(cen.astype(np.float64) / float(leaf) + 1e-4).astype(np.int64)
    ck2 = np.floor(cen.astype(np.float64) / float(leaf) - 1e-4).astype(np.int64)
    occupied = {tuple(k) for k in uniq}
    assert all(tuple(a) in occupied or tuple(b) in occupied for a, b in zip(ck, ck2))
    # PCL's order: ascending voxel index with x fastest
    mn = keys.min(0); div = keys.max(0) - mn + 1
    lin = lambda k: (k[:, 0] - mn[0]) + (k[:, 1] - mn[1]) * div[0] + (k[:, 2] - mn[2]) * div[0] * div[1]
    order = np.argsort(lin(uniq), kind="stable")
    sums = np.zeros((len(uniq), 3)); np.add.at(sums, np.searchsorted(lin(uniq)[order], lin(keys)), x.astype(np.float64))
    np.testing.assert_allclose(cen, sums / counts[order][:, None], rtol=1e-5, atol=1e-6)


@settings(max_examples=40, deadline=None)
@given(clouds)
def test_index_filters_commute_with_permutations(c):
    x = make_cloud(*c)
    rng = np.random.default_rng(c[0] + 2)
    x = x.copy(); x[rng.random(len(x)) < 0.1, rng.integers(0, 3)] = np.nan
    perm = rng.permutation(len(x))
    lo, hi = np.float32([-0.5, -0.25, -1.0]), np.float32([0.5, 0.75, 0.0])
    for f in (lambda a: oracle.remove_nan(a), lambda a: oracle.pass_through(a, lo, hi)):
        keep = f(x)
        assert (np.diff(keep) > 0).all()                          # input order kept
        np.testing.assert_array_equal(np.sort(perm[f(x[perm])]), keep)
