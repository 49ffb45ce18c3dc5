#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/.

  orb_pattern_i8.bin     the reference's own pattern table: `make -C oracle ref && oracle/_ref/ref_dump > ...`
                         (oracle/ref_dump.cpp includes /root/reference/openvslam/orb_point_pairs.h where it lies)
  geometry.json          S1/S2 tables restated in SURVEY.md section 8 (scale factors, sigma^2, level sizes, quotas, u_max)
                         + the two values the survey's probe of the reference headers printed: cos(1), sin(1)
  c1_vga_seed1000.npz    regression vector: CPU-oracle output for BASELINE config C1 (one 640x480 synthetic frame);
                         NOT a reference output (the reference extractor cannot be built here) -- it pins the oracle and
                         the GPU path against silent drift between rounds.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref):
        raw = subprocess.check_output([ref])
        assert len(raw) == 1024
        open(os.path.join(HERE, "orb_pattern_i8.bin"), "wb").write(raw)
    geometry = {
        "source": "SURVEY.md section 8 'Derived sizes' (restating static_settings.cpp:9-60, image_pyramid.cpp:78, orb_extractor.cpp:174-186)",
        "scale_factors_f32_as_f64": [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196],
        "level_sigma_sq": [1, 1.44000006, 2.07360005, 2.98598456, 4.29981804, 6.19173813, 8.91610336, 12.83919048],
        "sizes_720p": [[1280, 720], [1067, 600], [889, 500], [741, 417], [617, 347], [514, 289], [429, 241], [357, 201]],
        "sizes_vga": [[640, 480], [533, 400], [444, 333], [370, 278], [309, 231], [257, 193], [214, 161], [179, 134]],
        "quotas_2000": [434, 362, 302, 251, 209, 175, 145, 122],
        "u_max": [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3],
        "reference_header_probe": {"note": "printed by the survey's probe TU over the reference's own trigonometric.h / orb_point_pairs.h (SURVEY 8c)",
                                   "cos_1": 0.540614009, "sin_1": 0.841844141, "pattern_first": 8, "pattern_last": -11},
    }
    json.dump(geometry, open(os.path.join(HERE, "geometry.json"), "w"), indent=1)
    import mso
    img = mso.synth_frame(640, 480, 1000)
    kp = mso.orb_extract(mso.cfg(), img)
    np.savez_compressed(os.path.join(HERE, "c1_vga_seed1000.npz"), x=kp["x"], y=kp["y"], angle=kp["angle"],
                        octave=kp["octave"].astype(np.int8), desc=kp["desc"], image_sha256=np.frombuffer(
                            __import__("hashlib").sha256(img.tobytes()).digest(), dtype=np.uint8))
    print("wrote fixtures; C1 keypoints:", len(kp["x"]))


if __name__ == "__main__":
    main()
