#!/bin/bash
# C5's sequences in the deployment shape (bench.py pipelined_sequence.sequences_side_by_side): against the number of sequences and the hardware queues of the process
for q in 8 16 24; do for n in 8; do
echo "== sequences $n, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q BENCH_N_SEQ=$n timeout -k 10 300 python bench.py --only-c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined_sequence']; m=p['sequences_side_by_side']
print({k:m[k] for k in ('frames_per_s','pose_ba_ms_per_frame','keyframes_per_s','two_stage_new_window_ms_median')}, 'c5', d['c5']['frames_per_s'], d['c5'].get('hw_queues'))"
done; done
