"""Workload of bench.py's `greedy_match` leg: the reference's two BoW-guided matchers on what a new keyframe hands them.

`matchForTriangulationDBoW(currentKeyframe, keyframe)` runs once per adjacent keyframe of every new keyframe
(mapper_helpers.cpp:280-293), `matchForLoopClosures` per loop candidate (keyframe_matcher.cpp:50-158).  Inputs here are built with
the product only (extractor + vocabulary descent on the device) plus the neutral frame generator -- never the oracle:

  * 1 + n_adj keyframes of one synthetic 720p sequence, keyframe i shifted by (6 i, 3 i) px, 2000 keypoints each;
  * BoW nodes from `ms_bow_transform` on a synthetic k = 10, L = 6 vocabulary (1 111 111 nodes, the shape of the ORB vocabularies)
    at levelsUp = 4, i.e. <= 100 nodes per keyframe;
  * `usable` masks at 50 % (M1: keypoints WITH a triangulated map point; M2: keypoints WITHOUT one);
  * M2 geometry: pinhole bearings (f = 700 px), cameras translating parallel to the image plane in front of a plane at depth 5
    (so the true correspondences satisfy the epipolar gate), E from the reference's `create_E_21` formula.
"""
import numpy as np

import synth

K_VOC, L_VOC, LEVELS_UP = 10, 6, 4
FOCAL, DEPTH = 700.0, 5.0


def make_vocabulary(seed=7):
    """Flat DBoW2-style arrays of a balanced k = 10, L = 6 tree, ids breadth first; children differ from the parent in ~12 % of the bits."""
    rng = np.random.default_rng(seed)
    r32 = lambda shape: rng.integers(0, 2 ** 32, shape, dtype=np.uint64).astype(np.uint32)
    par, desc_l, first = [np.zeros(1, np.int32)], [np.zeros((1, 8), np.uint32)], 0
    for lv in range(1, L_VOC + 1):
        npar = len(desc_l[-1])
        par.append(np.repeat(first + np.arange(npar, dtype=np.int32), K_VOC))
        if lv == 1:
            desc_l.append(r32((npar * K_VOC, 8)))
        else:
            base = np.repeat(desc_l[-1], K_VOC, axis=0)
            desc_l.append(base ^ (r32(base.shape) & r32(base.shape) & r32(base.shape)))
        first += npar
    par, nd = np.concatenate(par), np.concatenate(desc_l)
    nn, nw = len(par), K_VOC ** L_VOC
    word = np.full(nn, -1, np.int32); word[nn - nw:] = np.arange(nw, dtype=np.int32)
    wt = np.zeros(nn); wt[nn - nw:] = rng.random(nw) + 0.1
    return dict(parent=par, desc=nd, weight=wt, word=word, depth_levels=L_VOC)


def e21(R1, t1, R2, t2):
    """openvslam/essential_solver.cc:149-162: R21 = R2 R1^T, t21 = -R21 t1 + t2, E = [t21]x R21."""
    R21 = R2 @ R1.T
    t21 = -R21 @ t1 + t2
    S = np.array([[0, -t21[2], t21[1]], [t21[2], 0, -t21[0]], [-t21[1], t21[0], 0]])
    return S @ R21


def build(ctx, mi355slam, n_adj=20, w=1280, h=720, seed=3000, usable_frac=0.5):
    """Returns a dict of host arrays per keyframe (index 0 = the new keyframe) + the per-pair essential matrices."""
    rng = np.random.default_rng(seed)
    n = n_adj + 1
    g = synth.SequenceSynth(w, h, seed, 6 * n_adj, 3 * n_adj)
    frames = np.ascontiguousarray(np.stack([g.frame(6 * i, 3 * i) for i in range(n)]))
    ex = mi355slam.OrbExtractor(ctx, w, h, max_batch=n)
    ex.extract(frames)
    kps = [ex.download(f) for f in range(n)]
    sf = mi355slam.scale_factors(8, 1.2)
    ex.close()
    voc = make_vocabulary()
    V = mi355slam.BowVocabulary(ctx, voc["parent"], voc["desc"], voc["weight"], voc["word"], L_VOC)
    kfs = []
    for i, kp in enumerate(kps):
        _, _, node = V.transform(kp["desc"], LEVELS_UP)
        px = np.stack([(kp["x"].astype(np.float64) - w / 2) / FOCAL, (kp["y"].astype(np.float64) - h / 2) / FOCAL, np.ones(len(node))], axis=1)
        kfs.append(dict(desc=kp["desc"], angle=kp["angle"], octave=kp["octave"], node=node.astype(np.int32),
                        bearing=px / np.linalg.norm(px, axis=1, keepdims=True),
                        has_mp=(rng.random(len(node)) < usable_frac).astype(np.uint8),
                        t=np.array([-6.0 * i * DEPTH / FOCAL, -3.0 * i * DEPTH / FOCAL, 0.0])))
    V.close()
    I3 = np.eye(3)
    E = np.stack([e21(I3, kfs[i]["t"], I3, kfs[0]["t"]) for i in range(1, n)])      # create_E_21(kf2.R, kf2.t, kf1.R, kf1.t), keyframe_matcher.cpp:171-175
    return dict(kfs=kfs, E=E, scale_factors=sf, n_adj=n_adj, lowe_ratio=0.75, thr_deg=2.0,
                nodes_per_kf=float(np.mean([len(np.unique(k["node"])) for k in kfs])),
                keypoints_per_kf=float(np.mean([len(k["node"]) for k in kfs])))
