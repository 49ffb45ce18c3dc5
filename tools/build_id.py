"""Identity of the kernel sources a profile was taken from: one SHA-256 over slam-module_amd/csrc (sources, headers, Makefile) and include/.
tools/profile_set.sh / tools/pmc_ba.sh store it in the PMC summaries; bench.py recomputes it and refuses to quote HBM traffic / instruction
counts measured on other sources (the committed counters must not go stale silently)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(root=ROOT):
    h = hashlib.sha256()
    files = []
    for d in (os.path.join(root, "slam-module_amd", "csrc"), os.path.join(root, "include")):
        for fn in sorted(os.listdir(d)):
            if fn.endswith((".hip", ".cpp", ".h", ".inc")) or fn == "Makefile":
                files.append(os.path.join(d, fn))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(source_hash())
