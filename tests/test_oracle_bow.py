"""CPU tests: the oracle's vocabulary-tree descent (oracle/bow.c, N3) against a plain python restatement of DBoW2's transform."""
import numpy as np

import bow_synth


def test_descent_matches_plain_restatement(oracle):
    for seed, kw in enumerate([dict(k=10, depth=3), dict(k=4, depth=5, ragged=0.3), dict(depth=4, max_children=23, ragged=0.2), dict(k=6, depth=3, ties=True)]):
        v = bow_synth.make_vocab(seed, **kw)
        q = bow_synth.make_queries(100 + seed, v, 150)
        for up in (0, 1, 2, 4, 7):
            word, weight, node = oracle.bow_transform(v, q, up)
            want = bow_synth.transform_py(v, q, up)
            assert [(int(a), float(b), int(c)) for a, b, c in zip(word, weight, node)] == want, (seed, up)


def test_first_child_wins_ties_and_levels(oracle):
    # root -> 3 children, the first two identical; each child -> 2 leaves
    parent = np.array([0, 0, 0, 0, 1, 1, 2, 2, 3, 3], np.int32)
    desc = np.zeros((10, 8), np.uint32)
    desc[3] = 0xFFFFFFFF
    desc[5, 0] = 1; desc[7, 0] = 1; desc[9] = 0xFFFFFFFF; desc[8] = 0xFFFFFFFE
    word = np.array([-1, -1, -1, -1, 0, 1, 2, 3, 4, 5], np.int32)
    weight = np.array([0, 0, 0, 0, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5])
    v = dict(parent=parent, desc=desc, weight=weight, word=word, depth_levels=2)
    q = np.zeros((3, 8), np.uint32); q[1, 0] = 1; q[2] = 0xFFFFFFFF
    w, wt, nd = oracle.bow_transform(v, q, levels_up=1)            # node level 1
    assert list(w) == [0, 1, 5] and list(wt) == [1.5, 2.5, 6.5] and list(nd) == [1, 1, 3]     # never child 2: child 1 wins the tie
    w, wt, nd = oracle.bow_transform(v, q, levels_up=0)            # node level 2 = the leaf
    assert list(nd) == [4, 5, 9]
    w, wt, nd = oracle.bow_transform(v, q, levels_up=4)            # level <= 0 -> root
    assert list(nd) == [0, 0, 0]
    v3 = dict(v, depth_levels=6)                                   # leaf above the node level: reported as the leaf itself
    assert list(oracle.bow_transform(v3, q, levels_up=1)[2]) == [4, 5, 9]
    empty = dict(parent=np.zeros(1, np.int32), desc=np.zeros((1, 8), np.uint32), weight=np.zeros(1), word=np.full(1, -1, np.int32), depth_levels=0)
    w, wt, nd = oracle.bow_transform(empty, q)
    assert list(w) == [-1, -1, -1] and list(wt) == [0, 0, 0]


def test_assemble_is_the_ordered_map_accumulation(oracle):
    rng = np.random.default_rng(5)
    n = 400
    word = rng.integers(0, 60, n).astype(np.int32); node = rng.integers(0, 25, n).astype(np.int32)
    weight = rng.random(n) * 3; weight[rng.random(n) < 0.1] = 0.0
    ow, ov, fn, fs, ff = oracle.bow_assemble(word, weight, node)
    v, fv = {}, {}
    for i in range(n):
        if weight[i] > 0:
            v[int(word[i])] = v.get(int(word[i]), 0.0) + float(weight[i])
            fv.setdefault(int(node[i]), []).append(i)
    keys = sorted(v)
    norm = 0.0
    for k in keys: norm += abs(v[k])
    assert list(ow) == keys and [float(x) for x in ov] == [v[k] / norm for k in keys]
    assert abs(sum(ov) - 1.0) < 1e-12
    assert list(fn) == sorted(fv)
    for j, k in enumerate(sorted(fv)): assert list(ff[fs[j]:fs[j + 1]]) == fv[k]
    ow, ov, fn, fs, ff = oracle.bow_assemble(word[:5], np.zeros(5), node[:5])      # only stop words
    assert len(ow) == 0 and len(fn) == 0
