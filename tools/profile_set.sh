#!/bin/bash
# One profile set of the current build: usage (on the GPU box)  bash tools/profile_set.sh <tag>
# Writes gpurun_out/prof_<tag>/: the default bench line, the bench line + kernel stats under rocprofv3, and one --pmc pass per
# counter group (counters on their own with --kernel-trace only).  Copy what is to be judged into profiles/ afterwards.
set -e
tag=$1
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the snapshot root)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench done"
rocprofv3 --kernel-trace --stats -d "$O/stats" -o p --output-format csv -- python3 "$R/bench.py" --no-cpu-baseline > "$O/bench_under_rocprof.json" 2> "$O/rocprof.err"
echo "stats done"
# the same headline region alone (23 launches of every front-end kernel, all of them 256-frame batches): per-kernel averages that can be held
# against the bench line's HIP-event times (the full command above also launches the kernels once per frame in its C5 leg)
rocprofv3 --kernel-trace --stats -d "$O/stats_headline" -o p --output-format csv -- python3 "$R/bench.py" --only-headline > "$O/bench_headline_under_rocprof.json" 2>> "$O/rocprof.err"
echo "headline stats done"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d "$O/pmc/$name" -o p --output-format csv -- python3 "$R/bench.py" --steps 2 --warmup 1 --only-headline > /dev/null 2>> "$O/rocprof.err"
  echo "pmc $name done"
done
python3 "$R/tools/pmc_summary.py" "$O/pmc" "$O/pmc_traffic.json"
# the local-BA launch (k_ba_lm) has passes of its own; its per-launch averages join the same file under kernels.k_ba_lm
bash "$R/tools/pmc_ba.sh" "$O/pmc_ba.json" > "$O/pmc_ba.log" 2>&1 || true
python3 - "$O/pmc_traffic.json" "$O/pmc_ba.json" <<'PY'
import json, sys
t = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
t["kernels"]["k_ba_lm"] = b
json.dump(t, open(sys.argv[1], "w"), indent=1)
print("merged k_ba_lm into", sys.argv[1])
PY
