#!/bin/bash
# C5 (8 sequences, one process, one GPU) and the same sequences in the deployment shape against the process's hardware queues (GPU_MAX_HW_QUEUES; ms_prepare_process picks
# it when the environment does not: two per context)
for q in 8 12 16 20 24 32 16 8; do
echo "== GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --only-c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined_sequence']; m=p['sequences_side_by_side']; n=p.get('sequences_side_by_side_native') or {}
print('c5', d['c5']['frames_per_s'], d['c5']['ba_per_s'], 'side by side', m['frames_per_s'], m['keyframes_per_s'], 'native', n.get('frames_per_s'), n.get('keyframes_per_s'), n.get('pose_ba_ms_per_frame'), 'one sequence', p['front_end_alone']['frames_per_s'], p['together']['frames_per_s'])"
done
