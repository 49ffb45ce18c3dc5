#!/bin/bash
# Copy what is to be judged from a gpurun_out/prof_<tag> directory (tools/profile_set.sh) into profiles/ as r<round>_<letter>_*:
# usage  bash tools/install_profiles.sh gpurun_out/prof_r02d r02_d
P=$1; T=$2
cp $P/bench_default.json profiles/${T}_bench_default.json
cp $P/bench_under_rocprof.json profiles/${T}_bench_under_rocprof.json
cp $P/bench_headline_under_rocprof.json profiles/${T}_bench_headline_under_rocprof.json
cp $P/stats/p_kernel_stats.csv profiles/${T}_kernel_stats_full_command.csv
cp $P/stats_headline/p_kernel_stats.csv profiles/${T}_kernel_stats_headline.csv
cp $P/pmc_traffic.json profiles/${T%_*}_pmc_traffic.json
cp $P/pmc_ba.json profiles/${T}_ba_pmc.json
cp $P/pmc_calibration.json profiles/${T}_pmc_calibration.json 2>/dev/null || true
for n in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT; do python3 - $P/pmc/$n/p_counter_collection.csv profiles/${T}_pmc_$n.csv <<'PY'
import csv, sys, collections
# condensed: per kernel and counter, dispatches and totals (the raw file has one row per dispatch and counter)
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"])
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
for (k, c), (n, v) in acc.items(): w.writerow([k, c, n, v, v / n])
PY
done
