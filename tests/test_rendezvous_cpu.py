"""The launcher-side channel (cytospace_amd/rendezvous.py) and bench.py's N > 1 entry, on CPU: two real processes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from cytospace_amd.rendezvous import FileStore
from cytospace_amd.cytospace import schedule_chunks, partition_indices
st = FileStore.from_env(timeout=60)
rank, world = st.rank, st.world
uid = st.bcast(bytes(range(128)) if rank == 0 else None)            # what the RCCL id travels through
assert uid == bytes(range(128))
st.barrier()
assert st.allreduce_max(1.5 + rank) == 1.5 + (world - 1)
assert st.allgather(("r", rank)) == [("r", r) for r in range(world)]
# the chunk fan-out's placement: every chunk on exactly one rank, every rank busy (cytospace.py:430-451 fans chunks out to workers)
idx = partition_indices(np.arange(2300), split_by_interval_int=500, shuffle=False)
owner = schedule_chunks([len(i) for i in idx], world)
mine = [k for k in range(len(idx)) if owner[k] == rank]
got = st.allgather(mine)
assert sorted(sum(got, [])) == list(range(len(idx))) and all(len(g) > 0 for g in got)
for k in range(50):                                                    # many barriers in a row keep their order
    assert st.allreduce_max(k * world + rank) == k * world + world - 1
st.barrier()
if rank == 0:
    print("RDV_OK", got)
"""


def test_file_store_two_processes(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    rdv = tmp_path / "rdv"
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", CYTO_RDV_DIR=str(rdv))) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "RDV_OK" in outs[0][0]


def test_file_store_under_an_external_launcher(tmp_path):
    # as the driver launches bench.py for N > 1: RANK / WORLD_SIZE / MASTER_PORT from a one-process-per-GPU launcher, no
    # CYTO_RDV_DIR -- the directory is derived from the master port and the launcher's pid
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = {k: v for k, v in os.environ.items() if k != "CYTO_RDV_DIR"}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", str(script), ROOT],
                       capture_output=True, text=True, timeout=280, env=dict(env, MASTER_ADDR="127.0.0.1"))
    assert "RDV_OK" in r.stdout, r.stdout + r.stderr


def test_a_missing_rank_times_out_instead_of_hanging(tmp_path):
    from cytospace_amd.rendezvous import FileStore, RendezvousTimeout
    import pytest
    st = FileStore(str(tmp_path / "rdv"), 0, 2, timeout=0.3)
    with pytest.raises(RendezvousTimeout):
        st.barrier()


def test_bench_is_free_of_the_tensor_framework_and_fails_loudly_without_devices():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "torch" not in src
    # `python bench.py --gpus 2` with no launcher starts its own ranks; on a box without GPUs it must say so, not print a line
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0 and r.stdout.strip() == "" and "HIP device" in r.stderr
    # a launcher whose WORLD_SIZE contradicts --gpus is an error too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
