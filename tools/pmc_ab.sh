#!/bin/bash
# A/B two builds of the library under the same PMC pass: usage tools/pmc_ab.sh "<counters>" libA.so libB.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "$2" "$3"; do
  cp "$R/$L" "$R/slam-module_amd/lib/libmi355slam.so"
  tag=$(basename "$L" .so)
  rm -rf "$R/gpurun_out/pmc_ab/$tag"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $1 -d "$R/gpurun_out/pmc_ab/$tag" -o p --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-ba --no-cpu-baseline > /dev/null 2>&1 || exit 1
  python3 - "$R/gpurun_out/pmc_ab/$tag" "$tag" <<'PY'
import csv, glob, sys, collections, re, os
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if m: acc[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"]) / 3
for k in ("k_fast", "k_describe"):
    print(sys.argv[2], k, {c: round(v) for c, v in sorted(acc[k].items())})
PY
done
