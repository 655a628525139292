"""Can two RCCL ranks live on ONE GPU (two threads of one process, each with its own si ctx)?  If so, world-2 collectives of
the in-library communicator can be tested on a one-GPU box.  Prints the outcome; run under `timeout`."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subspaceinference_jl_amd as si  # noqa: E402

uid = si._capi.comm_unique_id()
res = [None, None]


def rank(r):
    try:
        c = si.Context(0)
        c.comm_init_rank(2, r, uid)
        v = c.comm_allreduce_host(np.array([float(r + 1)]), "sum")
        res[r] = ("ok", float(v[0]))
        c.close()
    except Exception as e:  # noqa: BLE001
        res[r] = ("error", repr(e))


ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
for t in ts:
    t.start()
for t in ts:
    t.join(50)
print("RCCL two ranks on one GPU:", res)
