"""Condense rocprofv3 --pmc counter_collection CSVs into one JSON (profiles/rNN_pmc_traffic.json).

usage: python tools/pmc_summary.py <dir with one sub-directory per --pmc pass> <out.json>
Every pass ran `bench.py --steps 2 --warmup 1 --only-headline` (3 launches of the step); values are per step
(= per launch of the single-launch kernels, per chain of 7 launches for k_resize).  FETCH_SIZE / WRITE_SIZE are KiB on gfx950."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import build_id
root, out = sys.argv[1], sys.argv[2]
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
        e["hbm_bytes_per_step"] = int((cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024 / steps)
    e["launches_per_step"] = round(max(launches[k].values()) / steps, 2)
    res[k] = e
json.dump({"note": "rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) of `bench.py --steps 2 --warmup 1 --only-headline`, "
                   "256 frames per step; summed over a step's launches of each kernel.  FETCH_SIZE / WRITE_SIZE in KiB as reported, no x2 correction: "
                   "these kernels load 4 B per lane and k_blur, whose byte count is known (each level byte once + a 6-row halo per 8 rows through L2), "
                   "reads ~1.08x it, so the half-count artefact of 16 B/lane streams does not apply.  SQ_INSTS_* are wave-level instruction counts.",
           "steps_seen": steps, "src_sha256": build_id.source_hash(), "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, "steps", steps)
