"""CPU tests: the oracle against the committed golden vectors and known answers (no GPU)."""
import hashlib
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = json.load(open(os.path.join(GOLD, "geometry.json")))


def test_pattern_table_matches_reference_dump(oracle):
    raw = open(os.path.join(GOLD, "orb_pattern_i8.bin"), "rb").read()
    assert hashlib.sha256(raw).hexdigest() == "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"
    ref = np.frombuffer(raw, np.int8)
    assert np.array_equal(oracle.pattern(), ref)
    assert ref[0] == G["reference_header_probe"]["pattern_first"] and ref[-1] == G["reference_header_probe"]["pattern_last"]
    assert ref.min() == -13 and ref.max() == 12
    # every test point stays inside the 19 px margin the extractor keeps (static_settings.hpp:14)
    assert np.sqrt((ref.reshape(-1, 2).astype(np.float64) ** 2).sum(1)).max() < 19


def test_product_pattern_include_is_the_same_table():
    raw = np.frombuffer(open(os.path.join(GOLD, "orb_pattern_i8.bin"), "rb").read(), np.int8)
    root = os.path.dirname(os.path.dirname(GOLD))
    for rel in ("slam-module_amd/csrc/orb_pattern.inc", "oracle/orb_pattern.inc"):
        txt = "".join(l for l in open(os.path.join(root, rel)) if not l.startswith("//"))
        vals = np.array([int(v) for v in txt.replace("\n", "").split(",") if v.strip()], np.int8)
        assert np.array_equal(vals, raw), rel


def test_geometry_tables(oracle):
    sf = oracle.scale_factors(8, 1.2)
    assert np.allclose(sf.astype(np.float64), G["scale_factors_f32_as_f64"], rtol=0, atol=5e-10)
    assert np.allclose(oracle.level_sigma_sq(8, 1.2), G["level_sigma_sq"], rtol=1e-7)
    assert oracle.level_quotas(8, 1.2, 2000).tolist() == G["quotas_2000"]
    for key, (w, h) in (("sizes_720p", (1280, 720)), ("sizes_vga", (640, 480))):
        ws, hs = oracle.level_sizes(8, 1.2, w, h)
        assert [[int(a), int(b)] for a, b in zip(ws, hs)] == G[key]
    assert oracle.umax().tolist() == G["u_max"]
    assert oracle.level_quotas(8, 1.2, 2000).sum() == 2000
    assert oracle.level_quotas(1, 1.2, 500).tolist() == [500]
    # 749 pixels in the orientation disc
    u = oracle.umax()
    assert (2 * u[0] + 1) + 2 * sum(2 * int(v) + 1 for v in u[1:]) == 749


def test_trig_matches_reference_header_probe(oracle):
    L = oracle.lib()
    assert abs(L.mso_cos(1.0) - G["reference_header_probe"]["cos_1"]) < 5e-9 * 10
    assert abs(L.mso_sin(1.0) - G["reference_header_probe"]["sin_1"]) < 5e-9 * 10
    xs = np.linspace(-20, 20, 4001).astype(np.float32)
    c = np.array([L.mso_cos(float(x)) for x in xs]); s = np.array([L.mso_sin(float(x)) for x in xs])
    assert np.abs(c - np.cos(xs.astype(np.float64))).max() < 1.5e-3      # 3-term minimax polynomial
    assert np.abs(s - np.sin(xs.astype(np.float64))).max() < 1.5e-3


def test_fast_atan2_degrees(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(0)
    pts = rng.integers(-100000, 100000, (2000, 2)).astype(np.float32)
    got = np.array([L.mso_fast_atan2(float(y), float(x)) for x, y in pts])
    want = np.degrees(np.arctan2(pts[:, 1].astype(np.float64), pts[:, 0].astype(np.float64))) % 360
    err = np.abs((got - want + 180) % 360 - 180)
    assert err.max() < 0.02                                    # OpenCV documents ~0.3 deg; this polynomial is better
    assert L.mso_fast_atan2(0.0, 0.0) == 0.0 and L.mso_fast_atan2(1.0, 0.0) == 90.0 and L.mso_fast_atan2(0.0, -1.0) == 180.0
    assert ((got >= 0) & (got <= 360)).all()


def test_gauss7_taps_and_blur_properties(oracle):
    derived, fixed = oracle.gauss7_taps()
    assert derived.tolist() == fixed.tolist() == [18, 34, 48, 56, 48, 34, 18] and fixed.sum() == 256
    const = np.full((40, 50), 137, np.uint8)
    assert (oracle.gauss7(const) == 137).all()                 # taps sum to exactly 1.0 in 8.8 fixed point
    imp = np.zeros((41, 41), np.uint8); imp[20, 20] = 255
    k = fixed.astype(np.int64)
    want = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(oracle.gauss7(imp)[17:24, 17:24], want)
    # BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba -- blurring a padded image equals blurring in place
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (30, 37), dtype=np.uint8)
    pad = np.pad(img, 3, mode="reflect")
    assert np.array_equal(oracle.gauss7(pad)[3:-3, 3:-3], oracle.gauss7(img))


def test_resize_properties(oracle):
    const = np.full((60, 80), 201, np.uint8)
    assert (oracle.resize(const, 67, 50) == 201).all()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    assert np.array_equal(oracle.resize(img, 64, 48), img)      # identity scale: weights (2048, 0)
    # independent float reimplementation of the same fixed-point rule
    sw, sh, dw, dh = 64, 48, 53, 40
    out = oracle.resize(img, dw, dh)
    ref = np.zeros((dh, dw), np.int64)
    for dy in range(dh):
        fy = np.float32((dy + 0.5) * (1.0 / (dh / sh)) - 0.5); sy = int(np.floor(fy)); fy = np.float32(fy - sy)
        b0, b1 = int(np.rint(np.float32((np.float32(1) - fy) * np.float32(2048)))), int(np.rint(np.float32(fy * np.float32(2048))))
        y0, y1 = min(max(sy, 0), sh - 1), min(max(sy + 1, 0), sh - 1)
        for dx in range(dw):
            fx = np.float32((dx + 0.5) * (1.0 / (dw / sw)) - 0.5); sx = int(np.floor(fx)); fx = np.float32(fx - sx)
            if sx < 0: sx, fx = 0, np.float32(0)
            if sx >= sw - 1: sx, fx = sw - 1, np.float32(0)
            a0, a1 = int(np.rint(np.float32((np.float32(1) - fx) * np.float32(2048)))), int(np.rint(np.float32(fx * np.float32(2048))))
            x1 = min(sx + 1, sw - 1)
            r0 = int(img[y0, sx]) * a0 + int(img[y0, x1]) * a1
            r1 = int(img[y1, sx]) * a0 + int(img[y1, x1]) * a1
            ref[dy, dx] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(out, ref.astype(np.uint8))


def _fast_score_py(img, x, y):
    dx = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]; dy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    c = int(img[y, x]); d = [c - int(img[y + dy[i], x + dx[i]]) for i in range(16)]
    best = 0
    for s in range(16):
        arc = [d[(s + i) % 16] for i in range(9)]
        best = max(best, min(arc), -max(arc))
    return best


def test_fast_score_and_detector(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (24, 28), dtype=np.uint8)
    sm = oracle.fast_score_map(img)
    for y in range(3, 21):
        for x in range(3, 25):
            assert sm[y, x] == _fast_score_py(img, x, y)
    # an isolated bright dot on black: score = its value, all 16 arcs agree
    dot = np.zeros((50, 50), np.uint8); dot[25, 25] = 200
    assert oracle.fast_score_map(dot)[25, 25] == 200
    xs, ys, sc = oracle.detect_level(dot, 20, 10)
    assert xs.tolist() == [25] and ys.tolist() == [25] and sc.tolist() == [200]
    # margin 19: a corner at (18, 25) is detected by FAST but dropped by the border filter (feature_detector.cpp:121)
    dot2 = np.zeros((50, 50), np.uint8); dot2[25, 18] = 200
    assert len(oracle.detect_level(dot2, 20, 10)[0]) == 0
    # quota cut keeps the strongest, ties broken by raster index; output in key order
    many = np.zeros((80, 80), np.uint8)
    for i, (x, y) in enumerate([(30, 30), (40, 30), (50, 30), (30, 45), (40, 45)]):
        many[y, x] = [100, 150, 150, 90, 220][i]
    xs, ys, sc = oracle.detect_level(many, 20, 3)
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == [(40, 45, 220), (40, 30, 150), (50, 30, 150)]


def test_ic_angle_and_descriptor_basics(oracle):
    # a horizontal intensity ramp has its centroid on the +x axis -> angle 0; vertical ramp -> 90
    y, x = np.mgrid[0:64, 0:64]
    assert oracle.ic_angle((x * 3).astype(np.uint8), 32, 32) == 0.0
    assert oracle.ic_angle((y * 3).astype(np.uint8), 32, 32) == 90.0
    assert abs(oracle.ic_angle((255 - x * 3).astype(np.uint8), 32, 32) - 180.0) < 1e-3
    # descriptor bit k compares pattern point pairs; at angle 0 on a horizontal ramp bit = (x1 < x2) after cvRound
    ramp = (x * 3).astype(np.uint8)
    d = oracle.orb_descriptor(ramp, 32, 32, 0.0)
    pat = oracle.pattern().reshape(256, 4)
    L = oracle.lib(); ca, sa = L.mso_cos(0.0), L.mso_sin(0.0)
    bits = np.array([(d[k // 32] >> (k % 32)) & 1 for k in range(256)])
    want = np.array([int(np.rint(np.float32(p[0] * ca) - np.float32(p[1] * sa))) < int(np.rint(np.float32(p[2] * ca) - np.float32(p[3] * sa))) for p in pat.astype(np.float32)])
    assert np.array_equal(bits, want.astype(int))


def test_c1_regression_vector(oracle):
    g = np.load(os.path.join(GOLD, "c1_vga_seed1000.npz"))
    img = oracle.synth_frame(640, 480, 1000)
    assert np.array_equal(np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8), g["image_sha256"])
    kp = oracle.orb_extract(oracle.cfg(), img)
    assert len(kp["x"]) == len(g["x"]) == 1592
    assert np.array_equal(kp["desc"], g["desc"]) and np.array_equal(kp["x"], g["x"]) and np.array_equal(kp["y"], g["y"])
    assert np.array_equal(kp["angle"], g["angle"]) and np.array_equal(kp["octave"], g["octave"])


def test_extractor_order_and_empty_output(oracle):
    img = oracle.synth_frame(640, 480, 1003)
    kp = oracle.orb_extract(oracle.cfg(), img, track_xy=np.array([[100.5, 200.25], [3.0, 3.0], [320.0, 240.0]], np.float32), track_id=np.array([7, 8, 9], np.int32))
    assert kp["track_id"][:2].tolist() == [7, 9] and (kp["track_id"][2:] == -1).all()      # tracks first, border one dropped
    assert kp["x"][0] == np.float32(100.5) and kp["y"][0] == np.float32(200.25)             # "correct scale" (:121)
    assert (np.diff(kp["octave"][2:]) >= 0).all()                                            # level-major
    yy, xx = np.mgrid[0:480, 0:640]
    ramp = ((xx * 96) // 639 + (yy * 64) // 479).astype(np.uint8)
    assert len(oracle.orb_extract(oracle.cfg(), ramp)["x"]) == 0


def test_min_distance_rule(oracle):
    """feature_detector.cpp:79-82 scaling + the greedy spacing walk of the detector."""
    assert oracle.level_min_dist(10.0, 1280, 720) == 8 and oracle.level_min_dist(10.0, 357, 201) == 2 and oracle.level_min_dist(0.0, 1280, 720) == 0
    img = np.zeros((120, 160), np.uint8)
    pts = [(40, 40, 220), (46, 40, 200), (40, 47, 180), (90, 60, 150), (93, 62, 140)]     # (x, y, value): isolated dots = corners with that score
    for x, y, v in pts:
        img[y, x] = v
    xs, ys, sc = oracle.detect_level(img, 20, 10, min_dist=0)
    assert sorted(sc.tolist(), reverse=True) == [220, 200, 180, 150, 140]
    xs, ys, sc = oracle.detect_level(img, 20, 10, min_dist=8)
    # 200 is 6 px from 220 -> dropped; 180 is 7 px from 220 -> dropped; 140 is ~3.6 px from 150 -> dropped
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == [(40, 40, 220), (90, 60, 150)]
    xs, ys, sc = oracle.detect_level(img, 20, 10, min_dist=7)      # 180 at distance exactly 7 survives (strict <)
    assert sc.tolist() == [220, 180, 150]


def test_degree_to_radian_constant_is_exact(tmp_path):
    """orb_extractor.cpp:286 converts the keypoint angle with float(angleDeg * M_PI / 180.0) in double.  The device multiplies by the
    double constant M_PI / 180.0 instead of dividing; this checks, for EVERY float in [0, 361], that both give the same float."""
    import subprocess
    src = tmp_path / "deg.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
int main(void) {
    const double c = M_PI / 180.0;
    uint32_t lo, hi; float f0 = 0.0f, f1 = 361.0f;
    unsigned long long bad = 0, n = 0;
    memcpy(&lo, &f0, 4); memcpy(&hi, &f1, 4);
    for (uint32_t u = lo; u <= hi; ++u, ++n) {
        float x; memcpy(&x, &u, 4);
        const float a = (float)(((double)x * M_PI) / 180.0), b = (float)((double)x * c);
        bad += memcmp(&a, &b, 4) != 0;
    }
    printf("%llu %llu\n", n, bad);
    return 0;
}
''')
    exe = tmp_path / "deg"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"])
    n, bad = map(int, subprocess.check_output([str(exe)], text=True).split())
    assert n > 1_100_000_000 and bad == 0
