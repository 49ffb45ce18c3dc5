#!/bin/bash
# A/B builds of the library in ONE GPU session: usage  bash tools/ab_libs.sh <reps> lib1.so lib2.so ...   (paths relative to the repo)
R=$GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq 1 $reps); do for L in "$@"; do
  cp "$R/$L" "$R/slam-module_amd/lib/libmi355slam.so"
  timeout -k 10 200 python3 "$R/bench.py" --no-cpu-baseline --no-ba > "$R/gpurun_out/ab.json" 2>/dev/null || exit 1
  python3 - "$R/gpurun_out/ab.json" "$(basename $L .so)" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s %8.1f frames/s  " % (sys.argv[2], d["value"]) + "  ".join("%s %.4f" % (k, v["ms_per_launch"]) for k, v in d["kernels"].items() if k != "tracks"))
PY
done; done
