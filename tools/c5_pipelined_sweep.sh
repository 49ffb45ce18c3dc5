#!/bin/bash
# C5's sequences in the deployment shape (front end beside back end per sequence, bench.py pipelined_sequence.sequences_side_by_side) against the windows' team size
for t in 0 16 8 4 0; do
echo "== team $t"; timeout -k 10 300 python bench.py --only-c5 --no-cpu-baseline --c5-team $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined_sequence']; m=p['sequences_side_by_side']
print({k:m[k] for k in ('frames_per_s','pose_ba_ms_per_frame','keyframes_per_s','two_stage_new_window_ms_median','keyframe_wait_ms_median')}, 'one sequence:', p['front_end_alone']['frames_per_s'], p['together']['frames_per_s'], 'c5', d['c5']['frames_per_s'])"
done
