"""Synthetic bundle-adjustment problems shared by tests and bench.py.

Cameras on a slightly curved 10 m path (0.2 m spacing) looking at points in an 8 x 4 x 6 m box 4-10 m ahead; each point
is seen by a contiguous run of `run` cameras; observations are normalised image coordinates with N(0, (1/500)^2) noise,
information 500^2 / sigma^2_octave (bundle_adjuster.cpp:51), Huber delta sqrt(5.991); consecutive cameras are chained by
EdgeSE3Expmap odometry edges (bundle_adjuster.cpp:293-311) with Omega = diag(r^2 I3, p^2 I3) * 0.26667 / dt.
Poses are world->camera (q = x,y,z,w ; t).  numpy Philox generator, seed stated by the caller.
"""
import numpy as np

HUBER_DELTA = float(np.sqrt(np.float32(5.991)))     # constexpr float CHI2_THRESHOLD = 5.991 (bundle_adjuster.cpp:28)


def _quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = np.array([(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w])
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0); q = np.zeros(4); q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    if q[3] < 0: q = -q
    return q / np.linalg.norm(q)


def _R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rotvec(v):
    th = np.linalg.norm(v)
    if th < 1e-12: return np.eye(3)
    k = v / th; K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _pose(R, t):
    return np.concatenate([_quat_from_R(R), t])


def _compose(a, b):      # a * b
    Ra, Rb = _R_from_quat(a[:4]), _R_from_quat(b[:4])
    return _pose(Ra @ Rb, Ra @ b[4:] + a[4:])


def _inverse(a):
    R = _R_from_quat(a[:4]).T
    return _pose(R, -R @ a[4:])


def make_problem(n_pose=50, n_point=2000, run=10, seed=42, noise=1.0 / 500, perturb=True, fix_first=False, prior_r=100.0, prior_p=50.0, dt=0.1,
                 outlier_frac=0.0, yaw_total=0.2, z_drift=0.02):
    rng = np.random.Generator(np.random.Philox(seed))
    sf2 = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32) ** 2     # levelSigmaSq
    gt_pose = []
    for i in range(n_pose):
        c = np.array([0.2 * i - 0.1 * (n_pose - 1), 0.05 * np.sin(0.3 * i), z_drift * i])
        R = _rotvec(np.array([0.0, (0.004 if n_pose <= 50 else yaw_total / n_pose) * (i - n_pose / 2), 0.0])) @ _rotvec(np.array([0.01 * np.sin(i), 0, 0]))
        gt_pose.append(_pose(R, -R @ c))                      # world->camera
    gt_pose = np.array(gt_pose)
    half = 0.1 * (n_pose - 1)
    gt_point = np.stack([rng.uniform(-half - 1, half + 1, n_point), rng.uniform(-2, 2, n_point), rng.uniform(4, 10, n_point)], 1)
    run = min(run, n_pose)
    obs_pose, obs_point, uv, info = [], [], [], []
    for l in range(n_point):
        s = int(rng.integers(0, n_pose - run + 1))
        for i in range(s, s + run):
            p = _R_from_quat(gt_pose[i, :4]) @ gt_point[l] + gt_pose[i, 4:]
            assert p[2] > 0.5                                   # cheirality
            z = p[:2] / p[2] + rng.normal(0, noise, 2)
            if outlier_frac > 0 and rng.random() < outlier_frac:
                z = z + rng.normal(0, 60 * noise, 2)
            octv = int(rng.integers(0, 8))
            obs_pose.append(i); obs_point.append(l); uv.append(z); info.append(500.0 ** 2 / float(sf2[octv]))
    edge_i, edge_j, edge_meas, edge_info = [], [], [], []
    W = np.diag([prior_r ** 2] * 3 + [prior_p ** 2] * 3) * (0.26667 / dt)
    for i in range(1, n_pose):                                  # vertex0 = kf, vertex1 = previous kf (bundle_adjuster.cpp:76-77)
        odo_noise = np.concatenate([rng.normal(0, 0.001, 3), rng.normal(0, 0.002, 3)]) if perturb else np.zeros(6)
        M = _compose(gt_pose[i - 1], _inverse(gt_pose[i]))      # poseDifference(prev, kf), mapdb.cpp:224-229
        M = _compose(_pose(_rotvec(odo_noise[:3]), odo_noise[3:]), M)
        edge_i.append(i); edge_j.append(i - 1); edge_meas.append(M); edge_info.append(W)
    pose, point = gt_pose.copy(), gt_point.copy()
    if perturb:
        for i in range(n_pose):
            d = _pose(_rotvec(rng.normal(0, np.radians(0.5) / np.sqrt(3), 3)), rng.normal(0, 0.02 / np.sqrt(3), 3))
            pose[i] = _compose(d, pose[i])
        point = point + rng.normal(0, 0.02 / np.sqrt(3), point.shape)
    pose_fixed = np.zeros(n_pose, np.uint8)
    if fix_first:
        pose_fixed[0] = 1; pose[0] = gt_pose[0]
    return dict(pose=pose, point=point, pose_fixed=pose_fixed, point_fixed=None,
                obs_pose=np.array(obs_pose, np.int32), obs_point=np.array(obs_point, np.int32), obs_uv=np.array(uv), obs_info=np.array(info),
                huber_delta=HUBER_DELTA, edge_i=np.array(edge_i, np.int32), edge_j=np.array(edge_j, np.int32),
                edge_meas=np.array(edge_meas).reshape(-1, 7), edge_info=np.array(edge_info).reshape(-1, 36),
                gt_pose=gt_pose, gt_point=gt_point)


def residuals(prob, pose, point):
    """Per-observation reprojection residual (2-vector) in normalised image units."""
    out = np.zeros((len(prob["obs_pose"]), 2))
    Rs = [_R_from_quat(p[:4]) for p in pose]
    for o, (i, l) in enumerate(zip(prob["obs_pose"], prob["obs_point"])):
        p = Rs[i] @ point[l] + pose[i, 4:]
        out[o] = prob["obs_uv"][o] - p[:2] / p[2]
    return out


def make_problem_fast(n_pose=50, n_point=2000, run=10, seed=42, noise=1.0 / 500, prior_r=100.0, prior_p=50.0, dt=0.1):
    """The same window as make_problem (same trajectory, point box, run-of-`run` visibility, noise, octaves, odometry chain,
    perturbations), generated with vectorised draws: a different random stream, ~50x faster.  bench.py builds its 256 DISTINCT
    C4 windows with it."""
    rng = np.random.Generator(np.random.Philox(seed))
    sf2 = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))])).astype(np.float32) ** 2
    gt_pose = np.empty((n_pose, 7)); Rs = np.empty((n_pose, 3, 3))
    for i in range(n_pose):
        c = np.array([0.2 * i - 0.1 * (n_pose - 1), 0.05 * np.sin(0.3 * i), 0.02 * i])
        R = _rotvec(np.array([0.0, (0.004 if n_pose <= 50 else 0.2 / n_pose) * (i - n_pose / 2), 0.0])) @ _rotvec(np.array([0.01 * np.sin(i), 0, 0]))
        gt_pose[i] = _pose(R, -R @ c); Rs[i] = _R_from_quat(gt_pose[i, :4])
    half = 0.1 * (n_pose - 1)
    gt_point = np.stack([rng.uniform(-half - 1, half + 1, n_point), rng.uniform(-2, 2, n_point), rng.uniform(4, 10, n_point)], 1)
    run = min(run, n_pose)
    start = rng.integers(0, n_pose - run + 1, n_point)
    obs_point = np.repeat(np.arange(n_point, dtype=np.int32), run)
    obs_pose = (start[:, None] + np.arange(run)[None, :]).reshape(-1).astype(np.int32)
    p = np.einsum("oij,oj->oi", Rs[obs_pose], gt_point[obs_point]) + gt_pose[obs_pose, 4:]
    assert (p[:, 2] > 0.5).all()
    uv = p[:, :2] / p[:, 2:3] + rng.normal(0, noise, (len(p), 2))
    info = 500.0 ** 2 / sf2[rng.integers(0, 8, len(p))].astype(np.float64)
    W = np.diag([prior_r ** 2] * 3 + [prior_p ** 2] * 3) * (0.26667 / dt)
    edge_meas = np.empty((n_pose - 1, 7))
    odo = np.concatenate([rng.normal(0, 0.001, (n_pose - 1, 3)), rng.normal(0, 0.002, (n_pose - 1, 3))], 1)
    for i in range(1, n_pose):
        M = _compose(gt_pose[i - 1], _inverse(gt_pose[i]))
        edge_meas[i - 1] = _compose(_pose(_rotvec(odo[i - 1, :3]), odo[i - 1, 3:]), M)
    pose = gt_pose.copy()
    dr, dtv = rng.normal(0, np.radians(0.5) / np.sqrt(3), (n_pose, 3)), rng.normal(0, 0.02 / np.sqrt(3), (n_pose, 3))
    for i in range(n_pose):
        pose[i] = _compose(_pose(_rotvec(dr[i]), dtv[i]), pose[i])
    point = gt_point + rng.normal(0, 0.02 / np.sqrt(3), gt_point.shape)
    return dict(pose=pose, point=point, pose_fixed=np.zeros(n_pose, np.uint8), point_fixed=None, obs_pose=obs_pose, obs_point=obs_point,
                obs_uv=uv, obs_info=info, huber_delta=HUBER_DELTA, edge_i=np.arange(1, n_pose, dtype=np.int32),
                edge_j=np.arange(0, n_pose - 1, dtype=np.int32), edge_meas=edge_meas, edge_info=np.tile(W.reshape(1, 36), (n_pose - 1, 1)),
                gt_pose=gt_pose, gt_point=gt_point)


def residuals_fast(prob, pose, point):
    """Vectorised residuals(): the same numbers."""
    Rs = np.stack([_R_from_quat(p[:4]) for p in pose])
    p = np.einsum("oij,oj->oi", Rs[prob["obs_pose"]], point[prob["obs_point"]]) + pose[prob["obs_pose"], 4:]
    return prob["obs_uv"] - p[:, :2] / p[:, 2:3]


def pose_only_from_window(w, cur, use_gt_points=False):
    """poseBundleAdjust as the reference builds it (bundle_adjuster.cpp:396-491) from a window `w` of make_problem / make_problem_fast: keyframe `cur` free, the
    previous keyframe fixed and tied to it by the chain's odometry edge, every map point `cur` observes fixed; only `cur`'s observations."""
    sel = np.flatnonzero(w["obs_pose"] == cur)
    pts = np.unique(w["obs_point"][sel])
    remap = -np.ones(len(w["point"]), np.int64); remap[pts] = np.arange(len(pts))
    k = int(np.flatnonzero((w["edge_i"] == cur) & (w["edge_j"] == cur - 1))[0])
    point = (w["gt_point"] if use_gt_points and "gt_point" in w else w["point"])[pts] + 0.0
    return dict(pose=np.stack([w["pose"][cur], w["pose"][cur - 1]]), pose_fixed=np.array([0, 1], np.uint8), point=point, point_fixed=np.ones(len(pts), np.uint8),
                obs_pose=np.zeros(len(sel), np.int32), obs_point=remap[w["obs_point"][sel]].astype(np.int32), obs_uv=w["obs_uv"][sel].copy(), obs_info=w["obs_info"][sel].copy(),
                huber_delta=w["huber_delta"], edge_i=np.array([0], np.int32), edge_j=np.array([1], np.int32), edge_meas=w["edge_meas"][k:k + 1].copy(), edge_info=w["edge_info"][k:k + 1].copy())
