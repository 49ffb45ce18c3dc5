"""Synthetic DBoW2-style vocabulary trees for the N3 tests (no vocabulary file ships with the reference)."""
import numpy as np


def flip_bits(rng, d, p):
    noise = np.packbits(rng.random((len(d), 256)) < p, axis=1, bitorder="little").view(np.uint32)
    return d ^ noise


def make_vocab(seed, k=10, depth=3, ragged=0.0, max_children=None, stop_words=0.05, ties=False):
    """Nodes are created breadth first the way DBoW2's k-means builder and its loaders number them: every parent id is smaller
    than its children's.  ragged = probability that a non-root inner candidate stays a leaf; max_children: children per node drawn
    from 1..max_children instead of the fixed k; ties: some siblings share a descriptor."""
    rng = np.random.default_rng(seed)
    parent, desc, level = [0], [np.zeros(8, np.uint32)], [0]
    frontier = [0]
    for lv in range(1, depth + 1):
        nxt = []
        for p in frontier:
            if p != 0 and rng.random() < ragged: continue
            nc = k if max_children is None else int(rng.integers(1, max_children + 1))
            base = rng.integers(0, 2**32, (1, 8), dtype=np.uint64).astype(np.uint32) if p == 0 else desc[p][None, :]
            kids = flip_bits(rng, np.repeat(base, nc, axis=0), 0.5 if p == 0 else 0.25 / lv)
            if ties and nc > 2: kids[int(rng.integers(1, nc))] = kids[0]
            for c in range(nc):
                parent.append(p); desc.append(kids[c]); level.append(lv); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    parent = np.array(parent, np.int32); desc = np.stack(desc).astype(np.uint32)
    has_child = np.zeros(n, bool); has_child[parent[1:]] = True
    word = np.full(n, -1, np.int32); leaves = np.flatnonzero(~has_child & (np.arange(n) > 0))
    word[leaves] = np.arange(len(leaves), dtype=np.int32)                       # word ids in order of leaf appearance
    weight = np.zeros(n, np.float64)
    weight[leaves] = rng.random(len(leaves)) * 9.0 + 0.01
    weight[leaves[rng.random(len(leaves)) < stop_words]] = 0.0                  # stop words
    return dict(parent=parent, desc=desc, weight=weight, word=word, depth_levels=depth)


def make_queries(seed, vocab, n):
    """descriptors near random nodes of the tree (so descents are decided by small margins) plus pure noise"""
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, len(vocab["parent"]), n)
    q = flip_bits(rng, vocab["desc"][pick].copy(), 0.1)
    q[::7] = rng.integers(0, 2**32, (len(q[::7]), 8), dtype=np.uint64).astype(np.uint32)
    return q


def transform_py(vocab, desc, levels_up):
    """Plain restatement of DBoW2's per-feature transform on python lists (slow; small cases only)."""
    parent = vocab["parent"]; n = len(parent)
    children = [[] for _ in range(n)]
    for i in range(1, n): children[parent[i]].append(i)
    nid_level = vocab["depth_levels"] - levels_up
    out = []
    for f in desc:
        if not children[0]: out.append((-1, 0.0, 0)); continue
        final, level, nid = 0, 0, (0 if nid_level <= 0 else None)
        while True:
            level += 1
            nodes = children[final]
            final = nodes[0]
            best = int(np.unpackbits((f ^ vocab["desc"][final]).view(np.uint8)).sum())
            for c in nodes[1:]:
                d = int(np.unpackbits((f ^ vocab["desc"][c]).view(np.uint8)).sum())
                if d < best: best, final = d, c
            if level == nid_level: nid = final
            if not children[final]: break
        out.append((int(vocab["word"][final]), float(vocab["weight"][final]), final if nid is None else nid))
    return out
