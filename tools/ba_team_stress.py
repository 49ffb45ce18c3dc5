"""Team bundle adjustment beside a saturating front-end batch on another stream: the hand-offs between workgroups must hold
with the caches warm and the CUs contended (uneven arrival at the barriers), and nothing may hang."""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso, ba_synth
W, H, B = 1280, 720, 128
ca, cb = mi355slam.Context(0), mi355slam.Context(0)
frames = np.stack([mso.synth_frame(W, H, 1000 + i // 8, 2 * (i % 8), i % 8) for i in range(B)])
buf = ca.upload(frames)
ex = mi355slam.OrbExtractor(ca, W, H, max_batch=B)
probs = [ba_synth.make_problem(50, 2000, 10, seed=42), ba_synth.make_problem(30, 900, 8, seed=7, outlier_frac=0.03)]
bas = []
for p in probs:
    ba = mi355slam.BundleAdjuster(cb, [p], max_iters=10); ba.set_team(1); ba.solve(); cb.sync()
    ref = ba.download(0)
    bas.append((ba, ref))
bad = 0
t0 = time.perf_counter()
for it in range(40):
    team = (2, 5, 8, 16, 32)[it % 5]
    for _ in range(3): ex.extract(buf, n_frames=B, frame_stride=W * H, row_stride=W)      # ~6 ms of saturating work queued on stream A
    for ba, ref in bas:
        ba.set_team(team); ba.solve()
    cb.sync(); ca.sync()
    for k, (ba, ref) in enumerate(bas):
        out = ba.download(0)
        ok = (out["stats"]["iters"] == ref["stats"]["iters"] and out["stats"]["trials"] == ref["stats"]["trials"]
              and abs(out["stats"]["chi2_final"] - ref["stats"]["chi2_final"]) <= 1e-9 * abs(ref["stats"]["chi2_final"])
              and np.abs(out["pose"] - ref["pose"]).max() < 1e-9 and np.abs(out["point"] - ref["point"]).max() < 1e-9)
        if not ok:
            bad += 1
            print("MISMATCH it %d team %d problem %d: chi2 %r vs %r" % (it, team, k, out["stats"]["chi2_final"], ref["stats"]["chi2_final"]), flush=True)
print("team BA under front-end load: %d launches, %d mismatches, %.1f s" % (40 * len(bas), bad, time.perf_counter() - t0))
sys.exit(1 if bad else 0)
