import sys, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for p in ("slam-module_amd","oracle","tests"): sys.path.insert(0, os.path.join(R,p))
import numpy as np, ctypes as C, mi355slam, ba_synth
ctx=mi355slam.Context(0)
ba=mi355slam.BundleAdjuster(ctx,[ba_synth.make_problem()],max_iters=10)
ba.solve(); ctx.sync(); ba.solve(); ctx.sync()
st=ba.download(0)["stats"]; pc=st["phase_cycles"]
print({k: round(v/1e6,2) for k,v in pc.items()})
print("cholesky total %.2f Mcyc: mfma update %.2f, diag factor %.2f; rows below+writeback = stats[7] (not exported)" % (pc["cholesky"]/1e6, pc["schur_prep"]/1e6, pc["schur_prep_rhs"]/1e6))
