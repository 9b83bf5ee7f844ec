"""The launcher-side channel of a one-process-per-GPU job on ONE node: a directory of small files.

The path itself has a single collective (the broadcast of the transformed ST operand: RCCL over xGMI between GPUs,
csrc/comm.hip); what the ranks need beyond it is host-side and tiny -- hand the 128-byte RCCL id from rank 0 to the others, a
barrier around a timed region, the maximum of the ranks' times.  The reference has no counterpart (its workers are children of
one ProcessPoolExecutor, cytospace.py:430-451).  No torch, no MPI, no sockets (so no hostname resolution): every rank of a node
sees the same /tmp.

    store = FileStore.from_env()      # RANK / WORLD_SIZE (+ CYTO_RDV_DIR, or MASTER_PORT + the launcher's pid) from the environment
    uid   = store.bcast(uid if store.rank == 0 else None)
    store.barrier(); t = time.perf_counter(); ...; store.barrier(); worst = store.allreduce_max(time.perf_counter() - t)
"""
import os
import pickle
import tempfile
import time


class RendezvousTimeout(RuntimeError):
    pass


class FileStore:
    def __init__(self, path, rank, world, timeout=900.0):
        self.path, self.rank, self.world, self.timeout = str(path), int(rank), int(world), float(timeout)
        self._seq = 0
        if self.rank == 0:
            os.makedirs(self.path, exist_ok=True)
        else:
            self._wait(lambda: os.path.isdir(self.path), f"rank 0 never created {self.path}")

    @classmethod
    def from_env(cls, timeout=900.0):
        """RANK / WORLD_SIZE as torch.distributed.run, mpirun wrappers or bench.py's own spawner export them.  The directory:
        CYTO_RDV_DIR if set, else <tmp>/cytohip_rdv_<MASTER_PORT>_<run id>_<pid of the launcher> -- the ranks of one job
        are children of one launcher process, and no two jobs on a node share a master port AND a launcher pid."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        path = os.environ.get("CYTO_RDV_DIR")
        if not path:
            path = os.path.join(tempfile.gettempdir(), "cytohip_rdv_{}_{}_{}".format(
                os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid()))
        return cls(path, rank, world, timeout)

    # ---- primitives ----
    def _wait(self, cond, what):
        t0 = time.monotonic()
        spins = 0
        while not cond():
            spins += 1
            if spins > 200:                      # ~ the first 200 polls spin (a barrier around a timed region should cost microseconds)
                time.sleep(0.0002)
            if time.monotonic() - t0 > self.timeout:
                raise RendezvousTimeout(f"rank {self.rank}/{self.world}: {what} (waited {self.timeout:.0f} s)")

    def _file(self, key):
        return os.path.join(self.path, key)

    def set(self, key, data):
        tmp = self._file(f".{key}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, self._file(key))         # atomic: a reader sees the whole value or nothing

    def get(self, key):
        p = self._file(key)
        self._wait(lambda: os.path.exists(p), f"key {key!r} never appeared")
        with open(p, "rb") as f:
            return f.read()

    # ---- collectives (every rank calls them in the same order) ----
    def allgather(self, obj):
        self._seq += 1
        self.set(f"g{self._seq}.{self.rank}", pickle.dumps(obj))
        return [pickle.loads(self.get(f"g{self._seq}.{r}")) for r in range(self.world)]

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, x):
        return max(self.allgather(x))

    def bcast(self, obj, root=0):
        self._seq += 1
        if self.rank == root:
            self.set(f"b{self._seq}", pickle.dumps(obj))
            return obj
        return pickle.loads(self.get(f"b{self._seq}"))
