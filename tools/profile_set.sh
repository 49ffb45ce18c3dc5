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
# the profiler's preloaded tool initialises the GPU before bench.py runs, so ms_prepare_process comes too late there: give the profiled runs the queue count the
# unprofiled bench line gets from the library (two per context: 16 for the 8 sequences of its C5 leg), and the bench line says which applied (c5.hw_queues)
export GPU_MAX_HW_QUEUES=16
if [ -z "$SKIP_HEAD" ]; then      # SKIP_HEAD=1: the bench line, the kernel stats and the front-end counter passes are already in $O; do the calibration and the BA passes
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
fi
# what FETCH_SIZE / WRITE_SIZE report for loads of known size (tools/fetch_calib.hip): the factors the BA traffic below is corrected with
hipcc --offload-arch=gfx950 -O2 "$R/tools/fetch_calib.hip" -o "$O/fetch_calib" 2>> "$O/rocprof.err" && \
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/pmc_calib" -o p --output-format csv -- "$O/fetch_calib" > "$O/fetch_calib.log" 2>> "$O/rocprof.err" || echo "calibration pass failed"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/pmc_calib_w" -o p --output-format csv -- "$O/fetch_calib" >> "$O/fetch_calib.log" 2>> "$O/rocprof.err" || echo "write calibration pass failed"
echo "calibration done"
python3 - "$O/pmc_calib" "$O/pmc_calibration.json" <<'PY'
import csv, glob, json, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
wacc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1] + "_w", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_calib_write" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            wacc["write16" if ", 4u>" in r["Kernel_Name"] or "uint4" in r["Kernel_Name"] else "write4"].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_calib_write" in r["Kernel_Name"]: continue
        if "k_calib" in r["Kernel_Name"]:
            kn = r["Kernel_Name"]
            name = "strided8" if "strided8" in kn else "stride128" if "k_calib_stride<16>" in kn.replace(" ", "") else "stride256" if "k_calib_stride<32>" in kn.replace(" ", "") else "rows48" if "rows48" in kn else ("stream16" if ", 4u>" in kn or ",4u>" in kn or "uint4" in kn else ("stream8" if ", 2u>" in kn or ",2u>" in kn or "uint2" in kn else "stream4"))
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
GiB = float(1 << 30)
res = {}
# bytes of the 64-byte halves / 128-byte lines each pattern touches (what a counter of fetched bytes should at most see); streams touch everything once
touched = {"stream4": (GiB, GiB), "stream8": (GiB, GiB), "stream16": (GiB, GiB), "strided8": (GiB, GiB), "stride128": (GiB / 2, GiB), "stride256": (GiB / 4, GiB / 2),
           "rows48": (None, None)}
for k, cs in acc.items():
    fetch = sum(cs["FETCH_SIZE"]) / max(len(cs["FETCH_SIZE"]), 1) * 1024
    halves, lines = touched.get(k, (GiB, GiB))
    res[k] = {"FETCH_SIZE_bytes": fetch, "bytes_in_touched_64B_halves": halves, "bytes_in_touched_128B_lines": lines,
              "known_bytes": lines if lines else None, "factor_known_over_counter": round(lines / fetch, 3) if fetch and lines else None,
              "factor_halves_over_counter": round(halves / fetch, 3) if fetch and halves else None, "launches": len(cs["FETCH_SIZE"])}
    if k == "rows48":
        n = GiB / 256 - 1
        res[k].update({"useful_bytes": n * 48, "counter_bytes_per_row_piece": round(fetch / n, 1)})
for k, v in wacc.items():
    wb = sum(v) / len(v) * 1024
    res[k] = {"known_bytes": GiB, "WRITE_SIZE_bytes": wb, "factor_known_over_counter": round(GiB / wb, 3) if wb else None, "launches": len(v)}
json.dump({"note": "tools/fetch_calib.hip under rocprofv3 --pmc FETCH_SIZE (one counter per pass): 1 GiB read once with coalesced 4 / 8 / 16-byte loads per lane, and 128 MiB of doubles read at a "
                   "64-byte stride (every 64-byte half of the same 1 GiB touched).  factor = bytes really fetched / (FETCH_SIZE x 1024).", "kernels": res}, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res))
PY
# the local-BA launch (k_ba_lm) has passes of its own; its per-launch averages join the same file under kernels.k_ba_lm
bash "$R/tools/pmc_ba.sh" "$O/pmc_ba.json" > "$O/pmc_ba.log" 2>&1 || true
[ -f "$O/pmc_traffic.json" ] || { echo "no front-end summary in $O (SKIP_HEAD run): merge pmc_ba.json / pmc_calibration.json into it where it is"; exit 0; }
python3 - "$O/pmc_traffic.json" "$O/pmc_ba.json" <<'PY'
import json, sys
t = json.load(open(sys.argv[1])); b = json.load(open(sys.argv[2]))
try:
    import os
    cal = json.load(open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_calibration.json")))["kernels"]
    b["fetch_calibration"] = {k: v["factor_known_over_counter"] for k, v in cal.items()}
except Exception as e:
    b["fetch_calibration"] = {"error": str(e)}
t["kernels"]["k_ba_lm"] = b
json.dump(t, open(sys.argv[1], "w"), indent=1)
print("merged k_ba_lm into", sys.argv[1])
PY
# the bench line once more, now that the counters of THESE sources exist: the line of the set quotes them (roofline.traffic, valu_issue_frac) instead of withholding
# the numbers of an older tree (bench.py compares src_sha256); the snapshot's own copy of the summary is replaced, nothing outside the scratch tree is touched
cp "$O/pmc_traffic.json" "$R/$(grep -o '"profiles", "r[0-9]*_pmc_traffic.json"' "$R/bench.py" | head -1 | sed 's/"profiles", "/profiles\//; s/"$//')"      # (the file bench.py's PMC_FILE names)
python3 "$R/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench line with this set's counters done"
