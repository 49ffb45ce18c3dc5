#!/bin/bash
# runtime environment switches against the launch-bound legs (one sequence's front end, C5, the sequences side by side): A/B on one box
for v in "MS_X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=18 HSA_ENABLE_INTERRUPT=0" "MS_X=1"; do
echo "== $v"; env $v timeout -k 10 300 python bench.py --only-c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined_sequence']; n=p.get('sequences_side_by_side_native') or {}
fa=p['front_end_alone']
print('front end alone', fa['frames_per_s'], fa['pose_ba_ms_per_frame'], fa['extract_stage_us'], 'together', p['together']['frames_per_s'], 'c5', d['c5']['frames_per_s'], 'native side by side', n.get('frames_per_s'))"
done
