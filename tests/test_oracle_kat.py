"""Known-answer tests that pin the CPU oracle (SURVEY.md §8c KAT-1…KAT-8).

The reference has no tests or golden vectors and cannot be built here (PCL absent):
parity is UNPINNED against the reference itself.  These tests pin the oracle
against analytic answers and independent numpy/scipy computations instead.
"""
import numpy as np
import pytest
from scipy.spatial import cKDTree

import oracle
from conftest import load_pkg

synth = __import__("importlib").import_module("object-pose-estimation_amd.synth")


def rigid(rx, ry, rz, t):
    T = np.eye(4)
    T[:3, :3] = synth.rot_xyz(rx, ry, rz)
    T[:3, 3] = t
    return T


def apply(T, p):
    return (p.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


# ---------------------------------------------------------------- KAT-3: NN exactness
@pytest.mark.parametrize("nt,nq,seed", [(1, 5, 0), (7, 50, 1), (1000, 2000, 2), (20000, 5000, 3)])
def test_kdtree_1nn_matches_scipy(nt, nq, seed):
    rng = np.random.default_rng(seed)
    tgt = rng.uniform(-0.1, 0.1, (nt, 3)).astype(np.float32)
    q = rng.uniform(-0.2, 0.2, (nq, 3)).astype(np.float32)
    idx, d2, found = oracle.KdTree(tgt).knn(q, 1)
    dd, ii = cKDTree(tgt.astype(np.float64)).query(q.astype(np.float64))
    assert (found == 1).all()
    # distances agree to fp32 rounding of the squared distance
    np.testing.assert_allclose(d2[:, 0], dd ** 2, rtol=3e-6, atol=1e-12)
    # indices agree wherever the runner-up is not a numerical tie
    bi, bd = oracle.bruteforce_nn(tgt, q)
    np.testing.assert_array_equal(d2[:, 0], bd)  # same float arithmetic -> bit-equal distances
    assert (idx[:, 0] == ii).mean() > 0.999


def test_kdtree_knn_sorted_and_complete():
    rng = np.random.default_rng(5)
    tgt = rng.normal(0, 0.05, (3000, 3)).astype(np.float32)
    q = rng.normal(0, 0.05, (200, 3)).astype(np.float32)
    k = 30
    idx, d2, found = oracle.KdTree(tgt).knn(q, k)
    dd, ii = cKDTree(tgt.astype(np.float64)).query(q.astype(np.float64), k=k)
    assert (found == k).all()
    assert (np.diff(d2, axis=1) >= 0).all()
    np.testing.assert_allclose(d2, dd ** 2, rtol=3e-6, atol=1e-12)
    assert (idx == ii).mean() > 0.999


def test_kdtree_knn_fewer_points_than_k_and_nan_query():
    tgt = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    q = np.array([[0.1, 0, 0], [np.nan, 0, 0]], np.float32)
    idx, d2, found = oracle.KdTree(tgt).knn(q, 5)
    assert found.tolist() == [3, 0]
    assert idx[0, :3].tolist() == [0, 1, 2] and (idx[0, 3:] == -1).all()
    assert (idx[1] == -1).all() and np.isinf(d2[1]).all()


def test_kdtree_radius_matches_scipy():
    rng = np.random.default_rng(6)
    tgt = rng.uniform(-0.1, 0.1, (5000, 3)).astype(np.float32)
    q = tgt[:300]
    r = 0.02
    offs, idx, d2 = oracle.KdTree(tgt).radius(q, r, sorted_=True)
    ref = cKDTree(tgt.astype(np.float64)).query_ball_point(q.astype(np.float64), r)
    for i in range(len(q)):
        mine = set(idx[offs[i]:offs[i + 1]].tolist())
        theirs = set(ref[i])
        # allow only boundary points (|d - r| tiny) to differ
        for j in mine ^ theirs:
            d = np.linalg.norm(tgt[j].astype(np.float64) - q[i].astype(np.float64))
            assert abs(d - r) < 1e-6
        seg = d2[offs[i]:offs[i + 1]]
        assert (np.diff(seg) >= 0).all() and seg[0] == 0.0  # self first


def test_kdtree_skips_nonfinite_targets():
    tgt = np.array([[0, 0, 0], [np.nan, 0, 0], [0.5, 0, 0], [np.inf, 1, 1]], np.float32)
    idx, d2, found = oracle.KdTree(tgt).knn(np.array([[0.4, 0, 0]], np.float32), 4)
    assert found[0] == 2 and idx[0, :2].tolist() == [2, 0]


# ---------------------------------------------------------------- KAT-2: Umeyama vs numpy
def umeyama_np(src, dst):
    src = src.astype(np.float64); dst = dst.astype(np.float64)
    ms, md = src.mean(0), dst.mean(0)
    sigma = (dst - md).T @ (src - ms) / len(src)
    U, d, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(sigma) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = md - R @ ms
    return T


@pytest.mark.parametrize("acc", [0, 1])
def test_umeyama_matches_numpy(acc):
    rng = np.random.default_rng(7)
    src = rng.uniform(-0.1, 0.1, (500, 3)).astype(np.float32)
    Tgt = rigid(10, -20, 30, [0.03, -0.02, 0.05])
    dst = apply(Tgt, src) + rng.normal(0, 1e-4, (500, 3)).astype(np.float32)
    T = oracle.umeyama(src, dst, acc)
    np.testing.assert_allclose(T, umeyama_np(src, dst), atol=2e-6 if acc else 2e-5)
    np.testing.assert_allclose(T, Tgt, atol=2e-4)


def test_umeyama_reflection_case_gives_proper_rotation():
    rng = np.random.default_rng(8)
    src = rng.uniform(-1, 1, (50, 3)).astype(np.float32)
    dst = src.copy(); dst[:, 2] *= -1  # mirror: best proper rotation must have det +1
    T = oracle.umeyama(src, dst, 1)
    assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5
    np.testing.assert_allclose(T, umeyama_np(src, dst), atol=1e-5)


def test_umeyama_planar_rank2_case():
    rng = np.random.default_rng(9)
    src = np.c_[rng.uniform(-1, 1, (100, 2)), np.zeros(100)].astype(np.float32)
    Tgt = rigid(0, 0, 25, [0.1, 0.2, 0.0])
    dst = apply(Tgt, src)
    T = oracle.umeyama(src, dst, 1)
    np.testing.assert_allclose(T, Tgt, atol=1e-5)
    assert abs(np.linalg.det(T[:3, :3].astype(np.float64)) - 1) < 1e-5


def test_umeyama_from_sums_equals_two_pass():
    rng = np.random.default_rng(10)
    src = rng.uniform(-0.1, 0.1, (1000, 3)).astype(np.float32)
    dst = apply(rigid(3, 4, 5, [0.01, 0.02, -0.01]), src)
    pivot = np.array([0.01, -0.02, 0.03])
    s = src.astype(np.float64) - pivot; t = dst.astype(np.float64) - pivot
    S = np.zeros(17)
    S[0] = len(src); S[1:4] = s.sum(0); S[4:7] = t.sum(0); S[7:16] = (t.T @ s).reshape(9); S[16] = 0
    np.testing.assert_allclose(oracle.umeyama_from_sums(S, pivot), oracle.umeyama(src, dst, 1), atol=1e-6)


def test_svd3_reconstructs():
    rng = np.random.default_rng(11)
    for _ in range(20):
        A = rng.normal(size=(3, 3))
        U, s, V = oracle.svd3(A)
        np.testing.assert_allclose(U @ np.diag(s) @ V.T, A, atol=1e-12)
        np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(s, np.linalg.svd(A, compute_uv=False), atol=1e-12)
    A = np.outer([1, 2, 3], [4, 5, 6]).astype(float)  # rank 1
    U, s, V = oracle.svd3(A)
    np.testing.assert_allclose(U @ np.diag(s) @ V.T, A, atol=1e-12)
    np.testing.assert_allclose(U.T @ U, np.eye(3), atol=1e-10)


# ---------------------------------------------------------------- KAT-4: convergence state machine
def conv(**kw):
    c = oracle.Convergence()
    oracle.lib().orc_convergence_init(c)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def step(c, it, T, mse):
    t = oracle.colmajor(T)
    return oracle.lib().orc_convergence_step(c, it, t.ctypes.data_as(oracle._fp), mse), c.state


def test_convergence_defaults():
    c = conv()
    assert c.max_iterations == 100 and c.rotation_threshold == 0.99999
    assert c.translation_threshold == pytest.approx(9e-8) and c.mse_threshold_relative == 1e-5
    assert c.mse_threshold_absolute == 1e-12


def test_convergence_priority_and_states():
    big = rigid(5, 0, 0, [0.01, 0, 0])
    c = conv(max_iterations=3)
    assert step(c, 1, big, 1e-3) == (0, 0)          # first MSE: prev = DBL_MAX -> rel = 1.0 -> continue
    assert step(c, 2, big, 0.5e-3) == (0, 0)
    assert step(c, 3, big, 0.4e-3) == (1, 1)        # ITERATIONS wins first
    c = conv(max_iterations=3, failure_after_max_iter=1)
    assert step(c, 3, big, 1.0) == (0, 0)
    c = conv()
    assert step(c, 1, np.eye(4), 1e-3) == (1, 2)    # TRANSFORM (identity increment)
    c = conv()
    step(c, 1, big, 1e-3)
    assert step(c, 2, big, 1e-3 + 1e-13) == (1, 3)  # ABS_MSE before REL_MSE
    c = conv()
    step(c, 1, big, 1e-3)
    assert step(c, 2, big, 1e-3 * (1 + 5e-6)) == (1, 4)  # REL_MSE
    c = conv(mse_threshold_absolute=-1.0, mse_threshold_relative=0.0)
    step(c, 1, big, 1e-3)
    assert step(c, 2, big, 1e-3) == (0, 0)          # disabled thresholds never fire


def test_rotation_threshold_quirk_q1():
    # icp_mod.hpp:168: rotation threshold = 1 - transformation_epsilon
    src = synth.bumpy_torus(500)
    p = oracle.default_icp_params()
    p.max_iterations = 50; p.transformation_epsilon = 1e-8; p.euclidean_fitness_epsilon = 1e-8
    out = oracle.icp(src, apply(rigid(2, 1, -2, [0.003, 0.001, -0.002]), src), p)
    assert out.converged and out.state in (2, 3, 4)


# ---------------------------------------------------------------- KAT-1: rigid recovery
@pytest.mark.parametrize("pose", [
    (3, 0, 0, [0.005, 0, 0]), (0, -4, 2, [0, 0.01, -0.005]), (5, 5, 5, [0.01, 0.01, 0.01]),
    (-8, 2, 6, [-0.015, 0.005, 0.0]), (0, 0, 10, [0.0, 0.0, 0.02]), (2, -9, -3, [0.01, -0.02, 0.005]),
])
@pytest.mark.parametrize("mode", [0, 1])
def test_icp_recovers_rigid_motion(pose, mode):
    P = synth.bumpy_torus(2000)
    Tgt = rigid(*pose)
    Q = apply(Tgt, P)
    p = oracle.default_icp_params()
    p.max_iterations = 60; p.transformation_epsilon = 1e-12; p.euclidean_fitness_epsilon = 1e-14
    p.acc_mode = 1; p.transform_mode = mode
    out = oracle.icp(P, Q, p)
    assert out.converged
    np.testing.assert_allclose(out.T, Tgt, atol=2e-5)
    # KAT-8: fitness / strength scalars
    assert out.fitness < 1e-9
    assert out.align_strength == pytest.approx(out.n_corr / (len(P) + len(Q)))
    assert out.n_corr == len(P)
    f, n = oracle.fitness(P, Q, out.T)
    assert n == len(P) and f == pytest.approx(out.fitness)


def test_icp_guess_and_history_compose():
    P = synth.bumpy_torus(1500)
    Tgt = rigid(4, -3, 8, [0.01, -0.01, 0.015])
    Q = apply(Tgt, P)
    p = oracle.default_icp_params(); p.max_iterations = 40; p.acc_mode = 1
    guess = rigid(4, -3, 7, [0.01, -0.01, 0.014])
    out = oracle.icp(P, Q, p, guess=guess)
    np.testing.assert_allclose(out.T, Tgt, atol=3e-5)
    assert out.iterations <= 40 and len(out.T_hist) == out.iterations
    np.testing.assert_array_equal(out.T_hist[-1], out.T)


def test_icp_max_corr_dist_and_no_correspondences():
    P = synth.bumpy_torus(500)
    Q = apply(rigid(0, 0, 0, [1.0, 0, 0]), P)      # 1 m away
    p = oracle.default_icp_params(); p.max_corr_dist = 0.01
    out = oracle.icp(P, Q, p)
    assert not out.converged and out.state == 5 and out.iterations == 0
    np.testing.assert_array_equal(out.T, np.eye(4, dtype=np.float32))


def test_icp_reciprocal_subset():
    P = synth.bumpy_torus(800)
    Q = apply(rigid(1, 1, 1, [0.002, 0, 0]), P)[::2]
    p = oracle.default_icp_params(); p.max_iterations = 5; p.use_reciprocal = 1
    out = oracle.icp(P, Q, p)
    assert 0 < out.n_corr <= len(Q)
    assert len(set(out.corr_m.tolist())) == out.n_corr   # reciprocal => one-to-one


def test_icp_nan_points_are_skipped():
    P = synth.bumpy_torus(1000)
    Q = apply(rigid(2, 0, 1, [0.004, 0, 0]), P)
    Pn = P.copy(); Pn[::100] = np.nan
    p = oracle.default_icp_params(); p.max_iterations = 30; p.acc_mode = 1
    a = oracle.icp(Pn, Q, p)
    b = oracle.icp(P[np.isfinite(Pn).all(1)], Q, p)
    np.testing.assert_allclose(a.T, b.T, atol=1e-7)
    assert a.n_corr == b.n_corr == 990


def test_icp_normal_shooting_and_rejectors():
    # sphere: normals are radial, so all three normal-based stages have closed forms
    rng = np.random.default_rng(3)
    u = rng.normal(size=(3000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    c = np.array([0, 0, 1.0])
    P = (0.1 * u + c).astype(np.float32)
    nP = u.astype(np.float32)
    Tgt = rigid(0, 0, 0, [0.002, -0.001, 0.001])
    Q = apply(Tgt, P); nQ = nP.copy()
    p = oracle.default_icp_params()
    p.max_iterations = 20; p.corr_mode = 1; p.k_normal_shooting = 20
    p.use_surface_normal_rej = 1; p.surface_normal_thr = 0.7
    p.use_self_occluded_rej = 1; p.self_occluded_thr = 0.6
    out = oracle.icp(P, Q, p, src_nrm=nP, tgt_nrm=nQ)
    # self-occlusion keeps only points whose outward normal faces the origin: n·(−p/|p|) > 0.6
    assert 0 < out.n_corr < len(P) // 2
    src_final = apply(out.T.astype(np.float64), P)[out.corr_q]
    s = -(src_final / np.linalg.norm(src_final, axis=1, keepdims=True))
    assert ((nP[out.corr_q] * s).sum(1) > 0.55).all()
    assert ((nP[out.corr_q] * nQ[out.corr_m]).sum(1) > 0.7 - 1e-3).all()
    np.testing.assert_allclose(out.T[:3, 3], Tgt[:3, 3], atol=1.5e-3)


# ---------------------------------------------------------------- KAT-7: normals
def test_normals_plane_and_flip():
    rng = np.random.default_rng(12)
    xy = rng.uniform(-0.05, 0.05, (400, 2))
    P = np.c_[xy, np.full(400, 0.5)].astype(np.float32)
    nrm, curv = oracle.normals_knn(P, 30)
    np.testing.assert_allclose(np.abs(nrm[:, 2]), 1.0, atol=1e-3)
    assert (nrm[:, 2] < 0).all()                      # flipped towards the viewpoint at the origin
    assert (curv < 1e-3).all()
    nrm2, _ = oracle.normals_knn(P, 30, vp=(0, 0, 1.0))
    assert (nrm2[:, 2] > 0).all()


def test_normals_sphere_radial():
    rng = np.random.default_rng(13)
    u = rng.normal(size=(4000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    P = (0.2 * u).astype(np.float32)
    nrm, curv = oracle.normals_knn(P, 30, vp=(0, 0, 0))
    cosang = (nrm * -u).sum(1)                        # viewpoint inside: normals point inwards
    assert np.percentile(cosang, 1) > 0.995
    assert (curv > 0).all()


def test_normals_too_few_neighbours_nan():
    P = np.array([[0, 0, 0], [1, 0, 0]], np.float32)
    nrm, curv = oracle.normals_knn(P, 30)
    assert np.isnan(nrm).all() and np.isnan(curv).all()


# ---------------------------------------------------------------- KAT-5: pair features
def test_pair_features_hand_cases():
    z = [0, 0, 1.0]
    ok, (f1, f2, f3, f4) = oracle.pair_features([0, 0, 0], z, [1, 0, 0], z)
    assert ok and f4 == pytest.approx(1.0) and f3 == pytest.approx(0.0)
    assert f2 == pytest.approx(0.0) and f1 == pytest.approx(0.0)
    # n2 tilted about the y axis (v = d x n1 = (0,-1,0)); f1 = atan2(w·n2, n1·n2), w = n1 x v = (1,0,0)
    a = np.deg2rad(30)
    ok, (f1, f2, f3, f4) = oracle.pair_features([0, 0, 0], z, [1, 0, 0], [np.sin(a), 0, np.cos(a)])
    assert ok and f1 == pytest.approx(a, abs=1e-6) and f2 == pytest.approx(0, abs=1e-7)
    # swap branch: acos|n1·d| > acos|n2·d| (n1 more perpendicular to d) => roles exchanged, f3 = -angle2
    ok, (f1, f2, f3, f4) = oracle.pair_features([0, 0, 0], [0, 1, 0], [1, 0, 0], [0.6, 0.8, 0])
    assert ok and f3 == pytest.approx(-0.6) and f4 == pytest.approx(1.0)
    assert f2 == pytest.approx(0.0, abs=1e-7) and f1 == pytest.approx(np.arctan2(0.6, 0.8), abs=1e-6)
    # no swap: |n1·d| >= |n2·d|
    ok, (f1, f2, f3, f4) = oracle.pair_features([0, 0, 0], [0.8, 0.6, 0], [2, 0, 0], [0.6, 0.8, 0])
    assert ok and f3 == pytest.approx(0.8) and f4 == pytest.approx(2.0)
    assert f2 == pytest.approx(0.0, abs=1e-7) and f1 == pytest.approx(np.arctan2(-0.28, 0.96), abs=1e-6)
    # rejects
    ok, f = oracle.pair_features([0, 0, 0], z, [0, 0, 0], z)
    assert not ok and f == (0, 0, 0, 0)
    ok, f = oracle.pair_features([0, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0])   # d parallel to n1 -> |v| = 0
    assert not ok and f == (0, 0, 0, 0)


# ---------------------------------------------------------------- KAT-6: FPFH
def test_fpfh_plane_patch():
    rng = np.random.default_rng(14)
    P = np.c_[rng.uniform(-0.1, 0.1, (1500, 2)), np.zeros(1500)].astype(np.float32)
    N = np.tile(np.array([[0, 0, 1.0]], np.float32), (1500, 1))
    out, spfh, m = oracle.fpfh(P, N, 0.03)
    assert m > 20
    expect = np.zeros(33, np.float32); expect[[5, 16, 27]] = 100.0
    np.testing.assert_allclose(out, np.tile(expect, (1500, 1)), atol=1e-3)
    np.testing.assert_allclose(spfh, np.tile(expect, (1500, 1)), atol=1e-3)


def test_fpfh_subhistograms_sum_to_100_and_isolated_point():
    rng = np.random.default_rng(15)
    u = rng.normal(size=(2000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    P = np.r_[0.1 * u, [[5.0, 5.0, 5.0]]].astype(np.float32)      # last point isolated
    N = np.r_[u, [[0, 0, 1.0]]].astype(np.float32)
    out, spfh, _ = oracle.fpfh(P, N, 0.03)
    for g in range(3):
        np.testing.assert_allclose(out[:-1, 11 * g:11 * (g + 1)].sum(1), 100.0, atol=2e-3)
    # isolated point: radius search returns only itself -> no weighted neighbours -> all-zero row
    assert (out[-1] == 0).all()
    # sphere symmetry: descriptors are (nearly) the same everywhere
    assert np.abs(out[:-1] - out[:-1].mean(0)).max() < 25.0


# ---------------------------------------------------------------- UniformSampling
def test_uniform_sampling_one_per_voxel_deterministic():
    rng = np.random.default_rng(16)
    P = rng.uniform(-0.1, 0.1, (20000, 3)).astype(np.float32)
    leaf = 0.02
    idx = oracle.uniform_sampling(P, leaf)
    vox = np.floor(P * np.float32(1.0 / leaf)).astype(np.int64)
    keys = {tuple(v) for v in vox}
    assert len(idx) == len(keys) == len({tuple(v) for v in vox[idx]})
    # the survivor of each voxel minimises |p - ijk|^2 (PCL quirk: metric minus integer coordinates)
    for i in idx[:50]:
        members = np.where((vox == vox[i]).all(1))[0]
        d = ((P[members].astype(np.float32) - vox[i].astype(np.float32)) ** 2).sum(1)
        assert d[list(members).index(i)] <= d.min() * (1 + 1e-6)
    np.testing.assert_array_equal(idx, oracle.uniform_sampling(P, leaf))


# ---------------------------------------------------------------- SAC-IA
def test_sacia_error_metric_and_forced_samples():
    P = synth.bumpy_torus(600)
    Tgt = rigid(20, 10, 40, [0.05, -0.02, 0.03])
    Q = apply(Tgt, P)
    tree = oracle.KdTree(Q)
    assert oracle.sacia_error(P, tree, Tgt, 0.05) < 1e-3
    assert oracle.sacia_error(P, tree, rigid(0, 0, 0, [10, 0, 0]), 0.05) == pytest.approx(len(P))
    # identical descriptors for matching points -> any 5 well-spread samples give the exact pose
    rng = np.random.default_rng(17)
    feat = rng.uniform(0, 100, (600, 33)).astype(np.float32)
    T, err, it = oracle.sacia(P, feat, Q, feat, n_iter=20, nr_samples=5, k_corr=1, seed=3)
    np.testing.assert_allclose(T, Tgt, atol=1e-4)
    samp = np.tile(np.array([0, 100, 200, 300, 400], np.int32), (3, 1))
    forced = np.r_[samp.ravel(), samp.ravel()]
    T2, err2, it2 = oracle.sacia(P, feat, Q, feat, n_iter=3, nr_samples=5, k_corr=1, forced_samples=forced)
    np.testing.assert_allclose(T2, Tgt, atol=1e-4)
    assert it2 == 0


# ---------------------------------------------------------------- point-to-plane LLS
def test_point_to_plane_lls_small_motion_and_icp():
    P, N = synth.model_surface(3000, 5, return_normals=True)
    Tgt = rigid(1.0, -0.8, 1.5, [0.002, -0.001, 0.0015])
    Q = apply(Tgt, P); NQ = (N.astype(np.float64) @ Tgt[:3, :3].T).astype(np.float32)
    T1 = oracle.point_to_plane_lls(P, Q, NQ)
    assert np.abs(T1 - Tgt).max() < 1e-3                            # one linearised step
    R = T1[:3, :3].astype(np.float64)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)       # rebuilt from angles: a proper rotation
    # independent check of the normal equations with numpy
    s = P.astype(np.float64); d = Q.astype(np.float64); n = NQ.astype(np.float64)
    A = np.c_[np.cross(s, n), n]; b = ((d - s) * n).sum(1)
    x = np.linalg.lstsq(A, b, rcond=None)[0]
    np.testing.assert_allclose(T1[:3, 3], x[3:], atol=2e-6)
    p = oracle.default_icp_params(); p.max_iterations = 30; p.estimator = 1; p.acc_mode = 1
    out = oracle.icp(P, Q, p, tgt_nrm=NQ)
    assert out.converged and np.abs(out.T - Tgt).max() < 1e-6


# ---------------------------------------------------------------- filters
def test_remove_nan_and_pass_through_known_answers():
    x = np.array([[0.1, 0.1, 0.1], [np.nan, 0.0, 0.0], [0.2, 0.5, 1.0], [0.0, np.inf, 0.0], [0.2, 0.5, 1.0000001],
                  [-0.3, 0.5, 0.2], [0.2, 0.5, -1.0]], np.float32)
    np.testing.assert_array_equal(oracle.remove_nan(x), [0, 2, 4, 5, 6])
    # limits are inclusive (passthrough.hpp: removed iff value > max || value < min); z, y, x in sequence = a box
    big = np.float32(np.finfo(np.float32).max)
    np.testing.assert_array_equal(oracle.pass_through(x, [-big, -big, -1.0], [big, big, 1.0]), [0, 2, 5, 6])
    np.testing.assert_array_equal(oracle.pass_through(x, [0.0, 0.1, -1.0], [0.2, 0.5, 1.0]), [0, 2, 6])
    assert len(oracle.remove_nan(np.empty((0, 3), np.float32))) == 0
    assert len(oracle.pass_through(x, [1, 1, 1], [0, 0, 0])) == 0          # empty interval keeps nothing


def test_voxel_grid_known_answers():
    # two voxels of edge 1: {p0, p1, p4} and {p2}; NaN dropped; output in ascending voxel index (x fastest)
    x = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [1.5, 0.1, 0.1], [np.nan, 0, 0], [0.3, 0.9, 0.1]], np.float32)
    c = oracle.voxel_grid(x, 1.0)
    assert c.shape == (2, 3)
    s = (np.float32(0.1) + np.float32(0.2)) + np.float32(0.3)              # float sums in input order ...
    np.testing.assert_array_equal(c[0, 0], s * (np.float32(1) / np.float32(3)))   # ... times the reciprocal (Eigen 3.2 operator/=)
    np.testing.assert_array_equal(c[1], x[2])
    # negative coordinates floor towards -inf; anisotropic leaves; a single point is its own centroid
    y = np.array([[-0.05, 0.0, 0.0], [0.05, 0.0, 0.0], [-0.15, 0.0, 0.0]], np.float32)
    c = oracle.voxel_grid(y, [0.1, 1.0, 1.0])
    np.testing.assert_allclose(c[:, 0], [-0.15, -0.05, 0.05], rtol=1e-6)
    # every input point lies in exactly one voxel: the count-weighted mean of the centroids is the cloud's mean
    rng = np.random.default_rng(3)
    z = rng.uniform(-0.2, 0.2, (5000, 3)).astype(np.float32)
    c = oracle.voxel_grid(z, 0.05)
    keys = np.floor(z / np.float32(0.05)).astype(np.int64)
    assert len(c) == len(np.unique(keys, axis=0))
    # PCL refuses leaves whose voxel index would overflow 32 bits and hands back its input
    assert oracle.voxel_grid(z, 1e-5) is None
    assert len(oracle.voxel_grid(np.full((4, 3), np.nan, np.float32), 0.1)) == 0


def test_libm_f32_restatement_returns_the_c_librarys_bits(tmp_path):
    """oracle/libm_f32.h (and its device twin csrc/libm_f32.hpp) restate atanf / atan2f / acosf so that CPU and GPU agree bit
    for bit; the restatement is the C library's own algorithm: on this box every result equals libm's, so the oracle computes
    what it computed with libm (2e6 atan2f arguments over both quadrant pairs and small operands, every 97th float of [-1, 1]
    for acosf)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <math.h>
#include <stdio.h>
#include "libm_f32.h"
int main(void) {
  unsigned long long s = 88172645463325252ull; long bad = 0;
  for (long i = 0; i < 2000000; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; float a = (float)((double)(s & 0xffffff) / 8388608.0 - 1.0);
    s ^= s << 13; s ^= s >> 7; s ^= s << 17; float b = (float)((double)(s & 0xffffff) / 8388608.0 - 1.0);
    if (i % 3 == 0) a *= 1e-3f; else if (i % 7 == 0) b *= 1e-4f;
    if (lmf_bits(atan2f(a, b)) != lmf_bits(lmf_atan2f(a, b))) ++bad;
    if (lmf_bits(atanf(8.0f * a)) != lmf_bits(lmf_atanf(8.0f * a))) ++bad;
  }
  for (uint32_t u = 0; u <= 0x3f800000u; u += 97) {
    float x = lmf_from(u);
    if (lmf_bits(acosf(x)) != lmf_bits(lmf_acosf(x)) || lmf_bits(acosf(-x)) != lmf_bits(lmf_acosf(-x))) ++bad;
  }
  printf("%ld\n", bad);
  return 0;
}
''')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(root, "oracle"), str(src), "-o", str(exe), "-lm"])
    assert int(subprocess.check_output([str(exe)]).decode().strip()) == 0
