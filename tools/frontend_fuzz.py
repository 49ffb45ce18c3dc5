"""Extended differential fuzz of the front end against the CPU oracle: N random configurations (sizes biased towards tile edges),
bit-exact keypoints, angles, descriptors and pyramid pixels.  usage: python tools/frontend_fuzz.py [N] [seed]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("slam-module_amd", "oracle"): sys.path.insert(0, os.path.join(R, p))
import numpy as np, mi355slam, mso
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = mi355slam.Context(0)
done = bad = 0
while done < N:
    base_w = int(rng.choice([248, 496, 744, 992, 64, 128, 256, 512])); base_h = int(rng.choice([30, 60, 72, 90, 144, 216]))
    w = base_w + int(rng.integers(-9, 10)) if rng.random() < 0.6 else int(rng.integers(41, 1100))
    h = base_h + int(rng.integers(-5, 6)) + 41 if rng.random() < 0.6 else int(rng.integers(41, 500))
    levels = int(rng.integers(1, 6)); sf = float(rng.choice([1.1, 1.2, 1.25, 1.5, 2.0]))
    if min(w, h) / sf ** (levels - 1) < 41: continue
    thr = int(rng.integers(4, 50)); kp = int(rng.integers(50, 3000)); kind = int(rng.integers(0, 3))
    if kind == 0: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1: img = mso.synth_frame(w, h, int(rng.integers(0, 1000)))
    else: img = (rng.integers(0, 4, (h // 8 + 1, w // 8 + 1), dtype=np.uint8) * 80).repeat(8, 0).repeat(8, 1)[:h, :w].copy()
    img = np.ascontiguousarray(img)
    ocfg = mso.cfg(levels=levels, scale_factor=sf, max_kpts=kp, fast_threshold=thr)
    ex = mi355slam.OrbExtractor(ctx, w, h, levels=levels, scale_factor=sf, max_kpts=kp, fast_threshold=thr, max_batch=1)
    ex.extract(img[None])
    got = ex.download(0); want = mso.orb_extract(ocfg, img)
    ok = len(got["x"]) == len(want["x"]) and all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in ("x", "y", "angle")) \
        and np.array_equal(got["octave"], want["octave"]) and np.array_equal(got["desc"], want["desc"])
    lv, bl = mso.build_pyramid(ocfg, img)
    for l in range(levels):
        ok = ok and np.array_equal(ex.download_level(0, l), lv[l]) and np.array_equal(ex.download_level(0, l, blurred=True), bl[l])
    done += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(w=w, h=h, levels=levels, sf=sf, thr=thr, kp=kp, kind=kind), flush=True)
    ex.close() if hasattr(ex, "close") else None
print("fuzz: %d configurations, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
