"""CPU tests of the BA oracle: Jacobians against finite differences, Schur == full system, convergence."""
import numpy as np

import ba_synth


def test_se3_exp_log_roundtrip(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        u = np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 2, 3)])
        assert np.allclose(oracle.se3_log(oracle.se3_exp(u)), u, atol=1e-10)
    assert np.allclose(oracle.se3_exp(np.zeros(6)), [0, 0, 0, 1, 0, 0, 0])
    tiny = np.array([1e-7, -2e-7, 3e-7, 0.1, 0.2, 0.3])
    assert np.allclose(oracle.se3_log(oracle.se3_exp(tiny)), tiny, atol=1e-12)


def test_projection_jacobians_match_finite_differences(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        pose = oracle.se3_exp(np.concatenate([rng.normal(0, 0.3, 3), rng.normal(0, 1, 3)]))
        X = rng.uniform(-2, 2, 3) + np.array([0, 0, 6.0]); uv = rng.normal(0, 0.1, 2)
        e, Jp, Jl = oracle.ba_proj_edge(pose, X, uv)
        h = 1e-6
        for a in range(6):                       # oplus: T <- exp(d) * T  (VertexSE3Expmap::oplusImpl)
            d = np.zeros(6); d[a] = h
            ep, _, _ = oracle.ba_proj_edge(oracle.se3_mul(oracle.se3_exp(d), pose), X, uv)
            em, _, _ = oracle.ba_proj_edge(oracle.se3_mul(oracle.se3_exp(-d), pose), X, uv)
            assert np.allclose((ep - em) / (2 * h), Jp[:, a], atol=1e-6)
        for a in range(3):
            d = np.zeros(3); d[a] = h
            ep, _, _ = oracle.ba_proj_edge(pose, X + d, uv); em, _, _ = oracle.ba_proj_edge(pose, X - d, uv)
            assert np.allclose((ep - em) / (2 * h), Jl[:, a], atol=1e-6)


def test_pose_edge_error_zero_at_measurement_and_jacobian_first_order(oracle):
    rng = np.random.default_rng(2)
    Ti = oracle.se3_exp(rng.normal(0, 0.3, 6)); Tj = oracle.se3_exp(rng.normal(0, 0.3, 6))
    # M = Tj * Ti^-1 makes log(Tj^-1 M Ti) vanish
    inv = lambda T: oracle.se3_exp(-oracle.se3_log(T))
    M = oracle.se3_mul(Tj, inv(Ti))
    e, Ji, Jj = oracle.ba_pose_edge(Ti, Tj, M)
    assert np.abs(e).max() < 1e-12
    # g2o's Jacobians are the adjoint approximation: exact to first order at zero error
    h = 1e-6
    for a in range(6):
        d = np.zeros(6); d[a] = h
        ep, _, _ = oracle.ba_pose_edge(oracle.se3_mul(oracle.se3_exp(d), Ti), Tj, M)
        assert np.allclose(ep / h, Ji[:, a], atol=1e-5)
        ep, _, _ = oracle.ba_pose_edge(Ti, oracle.se3_mul(oracle.se3_exp(d), Tj), M)
        assert np.allclose(ep / h, Jj[:, a], atol=1e-5)


def test_schur_equals_full_system(oracle):
    """g2o solves the un-marginalised system (bundle_adjuster.cpp:269); the Schur route must give the same iterates."""
    for seed, kw in [(1, {}), (2, dict(fix_first=True)), (3, dict(outlier_frac=0.15))]:
        p = ba_synth.make_problem(7, 50, 4, seed=seed, **kw)
        a, b = oracle.ba_solve(p, 10, False), oracle.ba_solve(p, 10, True)
        assert a["stats"]["iters"] == b["stats"]["iters"] and a["stats"]["trials"] == b["stats"]["trials"]
        ra, rb = ba_synth.residuals(p, a["pose"], a["point"]), ba_synth.residuals(p, b["pose"], b["point"])
        assert np.abs(ra - rb).max() < 1e-9
        assert abs(a["stats"]["chi2_final"] - b["stats"]["chi2_final"]) < 1e-7 * b["stats"]["chi2_final"]


def test_converges_to_noise_floor_and_flags_outliers(oracle):
    p = ba_synth.make_problem(10, 200, 6, seed=5, outlier_frac=0.05)
    r0 = ba_synth.residuals(p, p["pose"], p["point"])
    out = oracle.ba_solve(p, 10, False)
    r = ba_synth.residuals(p, out["pose"], out["point"])
    assert out["stats"]["chi2_final"] < 0.9 * out["stats"]["chi2_init"]      # outliers keep their (Huber) share
    assert np.median(np.abs(r)) < 0.5 * np.median(np.abs(r0))
    # chi2 per observation = info * |r|^2 (edge->chi2(), bundle_adjuster.cpp:378)
    assert np.allclose(out["chi2"], p["obs_info"] * (r ** 2).sum(1), rtol=1e-9)
    assert 0.02 < (out["chi2"] > 5.991).mean() < 0.15


def test_lm_rejects_a_bad_step_and_recovers(oracle):
    """Start far from the optimum: at least one trial is rejected (lambda grows), yet chi2 never increases."""
    p = ba_synth.make_problem(6, 80, 5, seed=6)
    p["point"] = p["point"] + np.random.default_rng(0).normal(0, 0.6, p["point"].shape)
    out = oracle.ba_solve(p, 15, False)
    assert out["stats"]["trials"] > out["stats"]["iters"]
    assert out["stats"]["chi2_final"] < out["stats"]["chi2_init"]
