"""Worker of tests/test_cost_gpu.py::test_two_ranks_broadcast_and_disjoint_chunks (one process per GPU).
argv: rank world id_file out_file.  Rank 0 writes the 128-byte RCCL unique id to id_file; only rank 0 passes the ST matrix."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cytospace_amd import _lib  # noqa: E402
from cytospace_amd import cytospace as gcyto  # noqa: E402

rank, world, id_file, out_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
if rank == 0:
    uid = _lib.Communicator.unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
comm = _lib.Communicator(uid, rank, world, device_id=rank)
d = np.load(os.path.join(ROOT, "tests", "golden", "gv11_apply_linear_assignment.npz"))
sc, st = d["ss_counts"], d["ss_st_counts"]
idx_sc = np.split(d["ss_idx_sc"], np.cumsum(d["ss_idx_sc_lens"])[:-1])
res = gcyto.assign_chunks(sc.astype(np.float32), st.astype(np.float32) if rank == 0 else None, d["ss_slots"], idx_sc,
                          subsampled_slots_list=list(d["ss_sub"]), rank=rank, world_size=world, device_id=rank,
                          already_normalized=False, comm=comm)
np.savez(out_file, chunks=np.array(sorted(res)), **{f"m{k}": v for k, v in res.items()})
comm.close()
