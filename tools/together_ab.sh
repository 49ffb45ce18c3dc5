#!/bin/bash
# the pipelined leg (front end beside back end) and C5 under experiment knobs of the bundle adjuster's host side; run from the repository root
for v in "MS_X=1" "MS_BA_EAGER_MAX=65536" "MS_BA_NO_EAGER_VERDICT=1" "MS_BA_EAGER_MAX=65536 MS_BA_NO_EAGER_VERDICT=1" "MS_BA_EAGER_MAX=0 MS_BA_NO_EAGER_VERDICT=1" "MS_X=1"; do
echo "== $v"; env $v timeout -k 10 200 python bench.py --only-c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined_sequence']; t=p['together']
print('alone', p['front_end_alone']['frames_per_s'], p['front_end_alone']['pose_ba_ms_per_frame'], 'together', t['frames_per_s'], t['pose_ba_ms_per_frame'], t['two_stage_new_window_ms_median'], t['keyframes_per_s'], 'c5', d['c5']['frames_per_s'])"
done
