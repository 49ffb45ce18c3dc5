#!/bin/bash
# PMC passes over the local-BA leg of bench.py (k_ba_lm, 256 distinct C4 windows per launch): usage  bash tools/pmc_ba.sh <out.json>
# One counter group per run, --kernel-trace only.  Writes the per-launch averages of the 256-workgroup launches.
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=16      # as in tools/profile_set.sh: under the profiler ms_prepare_process comes too late
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the snapshot root)}"
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/pmc_ba.json}; case "$OUT" in /*) ;; *) OUT="$R/$OUT";; esac
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rm -rf "$R/gpurun_out/pmc_ba/$name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d "$R/gpurun_out/pmc_ba/$name" -o p --output-format csv -- python3 "$R/bench.py" --only-ba --ba-steps 2 --no-ba-two-stage > /dev/null 2>&1 || echo "pass $name failed"
  echo "pmc $name done"
done
python3 - "$R/gpurun_out/pmc_ba" "$OUT" <<'PY'
import csv, glob, sys, collections, os, json
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ba_lm" in r["Kernel_Name"] and int(r["Grid_Size"]) == 256 * 512: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c + "_per_launch": round(sum(v) / len(v), 1) for c, v in sorted(acc.items())}
if "FETCH_SIZE_per_launch" in res and "WRITE_SIZE_per_launch" in res:
    res["hbm_bytes_per_launch_uncorrected"] = int((res["FETCH_SIZE_per_launch"] + res["WRITE_SIZE_per_launch"]) * 1024)
    res["hbm_bytes_per_launch"] = int((2.0 * res["FETCH_SIZE_per_launch"] + res["WRITE_SIZE_per_launch"]) * 1024)     # FETCH_SIZE counts half of the fetched bytes of whole-line patterns (tools/fetch_calib.hip)
    res["hbm_bytes_per_launch_range"] = [res["hbm_bytes_per_launch_uncorrected"], res["hbm_bytes_per_launch"]]          # the kernel's gathers use parts of lines: the truth lies between (see pmc_calibration.json: stride128 / stride256 / rows48)
res["launches_seen"] = {c: len(v) for c, v in acc.items()}
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import build_id
res["src_sha256"] = build_id.source_hash(os.environ["GRAFT_REPO_ROOT"])
res["note"] = ("rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) of `bench.py --only-ba --ba-steps 2 --no-ba-two-stage`: the 256-window launches of k_ba_lm "
               "(256 workgroups x 512 threads, 256 distinct C4 windows, 10 LM iterations).  FETCH_SIZE / WRITE_SIZE in KiB; hbm_bytes = 2.0 x FETCH_SIZE + WRITE_SIZE, the factor measured "
               "by tools/fetch_calib.hip for 8-byte gathers and 16-byte streams alike (pmc_calibration.json).")
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res))
PY
