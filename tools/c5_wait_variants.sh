mkdir -p gpurun_out/r4p
run() { echo "== $1"; env $1 timeout -k 10 200 python bench.py --only-c5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('c5',d); print(c['frames_per_s'], c.get('ba_per_s'))"; }
for rep in 1 2; do
run "MS_X=1"
run "MS_WAIT_SPIN_US=0"
run "MS_WAIT_SPIN_US=0 MS_WAIT_COARSE=1 MS_WAIT_SLEEP_US=25"
run "MS_WAIT_COARSE=1 MS_WAIT_SLEEP_US=25"
run "MS_WAIT_SPIN_US=40"
done
