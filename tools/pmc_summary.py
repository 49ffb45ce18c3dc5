"""Condense rocprofv3 --pmc counter_collection CSVs into one JSON (profiles/rNN_pmc_traffic.json).

usage: python tools/pmc_summary.py <dir with one sub-directory per --pmc pass> <out.json>
Every pass ran `bench.py --steps 2 --warmup 1 --only-headline` (3 launches of the step); values are per step
(= per launch of the single-launch kernels, per chain of 7 launches for k_resize).  FETCH_SIZE / WRITE_SIZE are KiB on gfx950."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_id
root, out = sys.argv[1], sys.argv[2]
# FETCH_SIZE on gfx950 counts a fetched 128-byte line as 64 bytes: tools/fetch_calib.hip (1 GiB read with 4 / 8 / 16-byte loads per lane and with strided 8-byte
# loads) measures bytes fetched = 2.0 x FETCH_SIZE for every access width, WRITE_SIZE exact.  The factor comes from that run's pmc_calibration.json when it
# lies next to the output, else the measured 2.0.
FETCH_FACTOR = 2.0
try:
    cal = json.load(open(os.path.join(os.path.dirname(os.path.abspath(out)), "pmc_calibration.json")))["kernels"]
    fs = [v["factor_known_over_counter"] for k, v in cal.items() if k.startswith("stream") and v.get("factor_known_over_counter")]      # whole lines consumed: the clean case
    if fs: FETCH_FACTOR = sum(fs) / len(fs)
except Exception:
    pass
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
        if not m: continue
        acc[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[m.group(0)][r["Counter_Name"]] += 1
steps = max(launches["k_fast"].values())
res = {}
for k, cs in sorted(acc.items()):
    e = {c + "_per_step": round(v / steps, 1) for c, v in cs.items()}
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        e["hbm_bytes_per_step"] = int((FETCH_FACTOR * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024 / steps)
        e["hbm_bytes_per_step_uncorrected"] = int((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024 / steps)
        # kernels that use a fraction of the lines they touch (k_describe's 48-byte row pieces): the factor was calibrated on whole-line patterns, so the truth lies
        # between the counter as it is and the corrected figure -- both are given
        e["hbm_bytes_per_step_range"] = [e["hbm_bytes_per_step_uncorrected"], e["hbm_bytes_per_step"]]
    e["launches_per_step"] = round(max(launches[k].values()) / steps, 2)
    res[k] = e
json.dump({"note": "rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) of `bench.py --steps 2 --warmup 1 --only-headline`, "
                   "256 frames per step; summed over a step's launches of each kernel.  FETCH_SIZE / WRITE_SIZE are KiB; hbm_bytes = FETCH_SIZE x %.2f + WRITE_SIZE: "
                   "the calibration run (tools/fetch_calib.hip, pmc_calibration.json) shows FETCH_SIZE counting half of the bytes fetched at every access width "
                   "(4, 8, 16 B per lane, strided 8 B) and WRITE_SIZE exact -- rounds 1 and 2 quoted the uncorrected sum.  SQ_INSTS_* are wave-level instruction counts." % FETCH_FACTOR,
           "fetch_factor": FETCH_FACTOR,
           "steps_seen": steps, "src_sha256": build_id.source_hash(), "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, "steps", steps)
