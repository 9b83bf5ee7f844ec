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
    with pytest.raises(RendezvousTimeout):                     # (rank 1 never says hello: rank 0 gives up at start-up)
        FileStore(str(tmp_path / "rdv"), 0, 2, timeout=0.3)
    with pytest.raises(RendezvousTimeout):                     # (rank 0 never makes the directory)
        FileStore(str(tmp_path / "rdv2"), 1, 2, timeout=0.3)
    st = FileStore(str(tmp_path / "rdv3"), 0, 1, timeout=0.3)  # a world of one meets nobody
    st.barrier()
    assert st.allreduce_max(2.5) == 2.5 and st.bcast(b"\x00\x01") == b"\x00\x01"
    st.close()
    assert not os.path.exists(str(tmp_path / "rdv3"))          # rank 0 removes the directory at the end


def _pair(path, timeout=30.0):
    """two ranks of one job in two threads of this process"""
    import threading
    from cytospace_amd.rendezvous import FileStore
    out = [None, None]

    def mk(r):
        try:
            out[r] = FileStore(path, r, 2, timeout=timeout)
        except BaseException as e:     # noqa: BLE001
            out[r] = e
    ts = [threading.Thread(target=mk, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out


def test_a_directory_left_by_an_earlier_job_is_harmless(tmp_path):
    # a restart under the same launcher, a reused CYTO_RDV_DIR or a recycled pid finds the earlier job's files; a stale
    # RCCL id would hang ncclCommInitRank and stale barriers would pass at once.  Keys carry the job's nonce: nobody asks for them.
    import threading
    from cytospace_amd import rendezvous as rdv
    path = str(tmp_path / "rdv")
    a0, a1 = _pair(path)
    assert not isinstance(a0, BaseException) and not isinstance(a1, BaseException), (a0, a1)
    res = {}
    t = threading.Thread(target=lambda: res.setdefault("uid", a1.bcast(None)))
    t.start()
    assert a0.bcast(b"old-id") == b"old-id"
    t.join()
    assert res["uid"] == b"old-id"
    stale = sorted(os.listdir(path))
    assert any(".b1" in f for f in stale)
    a0._closed = a1._closed = True                             # the earlier job died without saying good-bye: its files stay
    keep = {f: open(os.path.join(path, f), "rb").read() for f in stale}
    b0, b1 = _pair(path)                                       # the same directory again
    assert b0.nonce == b1.nonce != a0.nonce
    for f, data in keep.items():                               # ... and even with the old files put back under the new job's feet
        with open(os.path.join(path, f), "wb") as fh:
            fh.write(data)
    t = threading.Thread(target=lambda: res.setdefault("uid2", b1.bcast(None)))
    t.start()
    time_before = __import__("time").monotonic()
    __import__("time").sleep(0.2)
    assert t.is_alive()                                        # rank 1 waits for THIS job's key, the old b1 is not it
    assert b0.bcast(b"new-id") == b"new-id"
    t.join()
    assert res["uid2"] == b"new-id" and __import__("time").monotonic() - time_before < 20
    b1.close(); b0.close()
    assert not os.path.exists(path)
    assert rdv.loads(rdv.dumps((1, b"\xff", [None, 2.5, ("x", True)], {"k": (1,)}))) == (1, b"\xff", [None, 2.5, ("x", True)], {"k": (1,)})


def test_foreign_or_open_directories_and_planted_files_are_refused(tmp_path):
    import pickle
    import pytest
    from cytospace_amd.rendezvous import FileStore, RendezvousError
    # a symlink in the directory's place is not followed
    target = tmp_path / "elsewhere"
    target.mkdir()
    link = tmp_path / "link"
    link.symlink_to(target)
    with pytest.raises(RendezvousError):
        FileStore(str(link), 0, 1)
    # rank 0 closes the mode of a directory it finds and empties it
    d = tmp_path / "open"
    d.mkdir()
    os.chmod(d, 0o777)
    (d / "junk").write_bytes(b"x")
    st = FileStore(str(d), 0, 1)
    assert (os.stat(d).st_mode & 0o777) == 0o700 and os.listdir(d) == []
    # values are JSON: a planted pickle is a decoding error, never code
    class Boom:
        def __reduce__(self):
            return (os.system, ("touch " + str(tmp_path / "pwned"),))
    st._write(f"{st.nonce}.b1", pickle.dumps(Boom()))
    st2 = FileStore.__new__(FileStore)
    st2.__dict__.update(st.__dict__)
    st2.rank, st2._seq = 1, 0
    with pytest.raises(ValueError):
        st2.bcast(None)
    assert not (tmp_path / "pwned").exists()
    # a rank other than 0 refuses a directory that others can write to
    d2 = tmp_path / "open2"
    d2.mkdir()
    os.chmod(d2, 0o777)
    with pytest.raises(RendezvousError):
        FileStore(str(d2), 1, 2, timeout=6.0)
    if os.getuid() == 0:                                       # another owner (only root can make one here)
        d3 = tmp_path / "foreign"
        d3.mkdir()
        os.chown(d3, 12345, 12345)
        with pytest.raises(RendezvousError):
            FileStore(str(d3), 0, 1)


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
