import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from zerokit_amd.batch import BatchProver
t=time.time(); p=BatchProver(max_batch=1024, window_bits=7150114); print("bench prover init", round(time.time()-t,2)); p.close()
