#!/bin/bash
# PMC passes over the local-BA leg of bench.py (k_ba_lm): usage  bash tools/pmc_ba.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  name=$(echo "$grp" | cut -d' ' -f1)
  rm -rf "$R/gpurun_out/pmc_ba/$name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d "$R/gpurun_out/pmc_ba/$name" -o p --output-format csv -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1 || echo "pass $name failed"
done
python3 - "$R/gpurun_out/pmc_ba" <<'PY'
import csv, glob, sys, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_ba_lm" in r["Kernel_Name"]: acc[r["Counter_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
for c, g in sorted(acc.items()):
    print(c, {k: round(sum(v) / len(v)) for k, v in sorted(g.items())})
PY
