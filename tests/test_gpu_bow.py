"""GPU parity: vocabulary-tree descent (ms_bow_transform, N3) through the C ABI vs the CPU oracle -- word, weight and node per
descriptor are exact (integers and copied doubles); the assembled BowVector / FeatureVector therefore too."""
import numpy as np
import pytest

import bow_synth

pytestmark = pytest.mark.gpu


def check(oracle, ctx, vocab, q, ups=(4,)):
    import mi355slam
    V = mi355slam.BowVocabulary(ctx, vocab["parent"], vocab["desc"], vocab["weight"], vocab["word"], vocab["depth_levels"])
    for up in ups:
        w, wt, nd = V.transform(q, levels_up=up)
        ow, owt, ond = oracle.bow_transform(vocab, q, up)
        assert np.array_equal(w, ow) and np.array_equal(wt, owt) and np.array_equal(nd, ond), up
    V.close()


def test_balanced_vocabulary_k10(oracle, ctx):
    v = bow_synth.make_vocab(1, k=10, depth=4)                          # 11 111 nodes, 10 000 words
    q = bow_synth.make_queries(2, v, 5000)
    check(oracle, ctx, v, q, ups=(0, 1, 2, 3, 4, 5))
    w, wt, nd = oracle.bow_transform(v, q, 2)
    assert len(np.unique(w)) > 1000 and len(np.unique(nd)) > 50          # the descents spread over the tree


def test_ragged_trees_many_children_and_ties(oracle, ctx):
    for seed, kw in enumerate([dict(k=4, depth=6, ragged=0.3), dict(depth=4, max_children=40, ragged=0.2), dict(k=17, depth=3, ties=True),
                               dict(k=16, depth=3), dict(k=1, depth=5), dict(depth=3, max_children=70, ties=True)]):
        v = bow_synth.make_vocab(10 + seed, **kw)
        for n in (1, 15, 16, 17, 1000):
            check(oracle, ctx, v, bow_synth.make_queries(20 + seed, v, n), ups=(0, 2, 4))


def test_empty_inputs_and_bad_trees(oracle, ctx):
    import mi355slam
    v = bow_synth.make_vocab(3, k=5, depth=2)
    V = mi355slam.BowVocabulary(ctx, v["parent"], v["desc"], v["weight"], v["word"], v["depth_levels"])
    w, wt, nd = V.transform(np.zeros((0, 8), np.uint32))
    assert len(w) == 0
    V.close()
    E = mi355slam.BowVocabulary(ctx, np.zeros(1, np.int32), np.zeros((1, 8), np.uint32), np.zeros(1), np.full(1, -1, np.int32), 0)
    w, wt, nd = E.transform(bow_synth.make_queries(4, v, 40))
    assert (w == -1).all() and (wt == 0).all() and (nd == 0).all()
    E.close()
    bad = v["parent"].copy(); bad[3] = 7                                 # a parent that comes after its child
    with pytest.raises(mi355slam.MsError):
        mi355slam.BowVocabulary(ctx, bad, v["desc"], v["weight"], v["word"], 2)


def test_frame_descriptors_through_extractor_and_vocabulary(oracle, ctx):
    """end to end on real descriptor bits: extract a synthetic frame, walk its descriptors down a vocabulary grown from them"""
    import mi355slam
    img = oracle.synth_frame(640, 480, 1000)
    ex = mi355slam.OrbExtractor(ctx, 640, 480, max_batch=1)
    ex.extract(img)
    desc = ex.download(0)["desc"]
    assert len(desc) > 500
    v = bow_synth.make_vocab(7, k=10, depth=3)
    rng = np.random.default_rng(8)
    v["desc"][1:] = desc[rng.integers(0, len(desc), len(v["desc"]) - 1)] ^ bow_synth.flip_bits(rng, np.zeros((len(v["desc"]) - 1, 8), np.uint32), 0.05)
    check(oracle, ctx, v, desc, ups=(1, 4))
    w, wt, nd = oracle.bow_transform(v, desc, 1)
    ow, ov, fn, fs, ff = oracle.bow_assemble(w, wt, nd)
    assert abs(ov.sum() - 1.0) < 1e-12 and fs[-1] == (wt > 0).sum()
