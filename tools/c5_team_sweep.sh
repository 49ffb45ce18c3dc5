for t in 0 32 16 8 4 2 1; do
echo "== team $t"; timeout -k 10 200 python bench.py --only-c5 --no-cpu-baseline --c5-team $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['c5']['frames_per_s'], d['c5']['ba_per_s'], d['c5'].get('ba_team'))"
done
