"""The launcher-side channel of a one-process-per-GPU job on ONE node: a directory of small files.

The path itself has a single collective (the broadcast of the transformed ST operand: RCCL over xGMI between GPUs,
csrc/comm.hip); what the ranks need beyond it is host-side and tiny -- hand the 128-byte RCCL id from rank 0 to the others, a
barrier around a timed region, the maximum of the ranks' times.  The reference has no counterpart (its workers are children of
one ProcessPoolExecutor, cytospace.py:430-451).  No torch, no MPI, no sockets (so no hostname resolution): every rank of a node
sees the same /tmp.

    store = FileStore.from_env()      # RANK / WORLD_SIZE (+ CYTO_RDV_DIR, or MASTER_PORT + the launcher's pid) from the environment
    uid   = store.bcast(uid if store.rank == 0 else None)
    store.barrier(); t = time.perf_counter(); ...; store.barrier(); worst = store.allreduce_max(time.perf_counter() - t)

What keeps a shared /tmp from hurting:
  * the directory belongs to THIS user and nobody else can write into it: rank 0 makes it with mode 0700, or -- if it exists --
    refuses anything that is a symlink, is not a directory or has another owner, closes its mode to 0700 and empties it; the
    other ranks check owner and mode before they read a byte;
  * values are JSON (bytes and tuples tagged), never pickle: reading a planted file cannot run code;
  * every key carries the job's NONCE, which rank 0 draws at start-up and hands to rank r only in answer to r's own random token
    (hello.r -> ack.r): a directory left by an earlier job under the same name (a restart under one launcher, a reused
    CYTO_RDV_DIR, a recycled pid) holds keys of another nonce, which nobody asks for -- a stale RCCL id cannot be read, a stale
    barrier cannot pass;
  * at exit every rank says good-bye and rank 0, having waited for them, removes the directory.
"""
import atexit
import base64
import json
import os
import secrets
import shutil
import stat
import tempfile
import time


class RendezvousTimeout(RuntimeError):
    pass


class RendezvousError(RuntimeError):
    pass


def _enc(obj):
    if isinstance(obj, (bytes, bytearray)):
        return {"__bytes__": base64.b64encode(bytes(obj)).decode("ascii")}
    if isinstance(obj, tuple):
        return {"__tuple__": [_enc(x) for x in obj]}
    if isinstance(obj, list):
        return [_enc(x) for x in obj]
    if isinstance(obj, dict):
        if not all(isinstance(k, str) for k in obj):
            raise TypeError("FileStore values: dict keys must be strings")
        return {"__dict__": {k: _enc(v) for k, v in obj.items()}}
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if hasattr(obj, "item") and getattr(obj, "shape", None) == ():     # numpy scalars
        return _enc(obj.item())
    raise TypeError(f"FileStore values are None / bool / int / float / str / bytes / tuple / list / dict, not {type(obj).__name__}")


def _dec(x):
    if isinstance(x, list):
        return [_dec(v) for v in x]
    if isinstance(x, dict):
        if "__bytes__" in x:
            return base64.b64decode(x["__bytes__"])
        if "__tuple__" in x:
            return tuple(_dec(v) for v in x["__tuple__"])
        if "__dict__" in x:
            return {k: _dec(v) for k, v in x["__dict__"].items()}
        raise RendezvousError("malformed value in the rendezvous directory")
    return x


def dumps(obj):
    return json.dumps(_enc(obj), allow_nan=True).encode("utf-8")


def loads(data):
    return _dec(json.loads(data.decode("utf-8")))


class FileStore:
    def __init__(self, path, rank, world, timeout=900.0):
        self.path, self.rank, self.world, self.timeout = str(path), int(rank), int(world), float(timeout)
        self._seq = 0
        self._closed = False
        self._made_dir = False
        if self.rank == 0:
            self._claim_directory()
            self.nonce = secrets.token_hex(8)
            for r in range(1, self.world):                   # the other ranks' tokens -> the nonce, to each under its own token
                tok = loads(self._read_when_there(f"hello.{r}", f"rank {r} never said hello"))
                self._write(f"ack.{r}", dumps({"token": tok, "nonce": self.nonce}))
        else:
            self._wait(lambda: os.path.isdir(self.path), f"rank 0 never created {self.path}")
            self._check_directory()
            token = secrets.token_hex(8)
            hello, ack = self._file(f"hello.{self.rank}"), self._file(f"ack.{self.rank}")
            box = {}

            def answered():
                if not os.path.exists(hello):                # (rank 0 empties a directory it finds: say it again)
                    self._write(f"hello.{self.rank}", dumps(token))
                try:
                    with open(ack, "rb") as f:
                        a = loads(f.read())
                except (OSError, ValueError):
                    return False
                if isinstance(a, dict) and a.get("token") == token:      # (an ack for another token: an earlier job's)
                    box["nonce"] = a.get("nonce")
                    return True
                return False
            self._wait(answered, "rank 0 never answered")
            self.nonce = str(box["nonce"])
        atexit.register(self.close)

    # ---- the directory ----
    def _claim_directory(self):
        """rank 0: a directory only this user can touch, empty."""
        parent = os.path.dirname(os.path.abspath(self.path))
        os.makedirs(parent, exist_ok=True)
        try:
            os.mkdir(self.path, 0o700)
            self._made_dir = True
        except FileExistsError:
            self._check_directory(fix_mode=True)
            for name in os.listdir(self.path):               # whatever an earlier job left (our own files are flat)
                p = os.path.join(self.path, name)
                try:
                    if os.path.isdir(p) and not os.path.islink(p):
                        shutil.rmtree(p, ignore_errors=True)
                    else:
                        os.unlink(p)
                except OSError:
                    pass
        os.chmod(self.path, 0o700)

    def _check_directory(self, fix_mode=False):
        st = os.lstat(self.path)
        if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode):
            raise RendezvousError(f"{self.path} is not a directory (refusing to follow it)")
        if st.st_uid != os.getuid():
            raise RendezvousError(f"{self.path} belongs to uid {st.st_uid}, not to this user ({os.getuid()}): refusing to use it")
        if st.st_mode & 0o077:
            if fix_mode:
                os.chmod(self.path, 0o700)
            else:
                # (rank 0 closes the mode right after it claims the directory: give it a moment, then insist)
                t0 = time.monotonic()
                while os.lstat(self.path).st_mode & 0o077:
                    if time.monotonic() - t0 > min(self.timeout, 5.0):
                        raise RendezvousError(f"{self.path} is writable by others (mode {oct(st.st_mode & 0o777)}): refusing to use it")
                    time.sleep(0.01)

    @classmethod
    def from_env(cls, timeout=900.0):
        """RANK / WORLD_SIZE as torch.distributed.run, mpirun wrappers or bench.py's own spawner export them.  The directory:
        CYTO_RDV_DIR if set, else <tmp>/cytohip_rdv_<uid>_<MASTER_PORT>_<run id>_<pid of the launcher> -- the ranks of one job
        are children of one launcher process, and no two jobs on a node share a master port AND a launcher pid.  (A name an
        earlier job used is harmless: see the head of this file.)"""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        path = os.environ.get("CYTO_RDV_DIR")
        if not path:
            path = os.path.join(tempfile.gettempdir(), "cytohip_rdv_{}_{}_{}_{}".format(
                os.getuid(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid()))
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

    def _write(self, key, data):
        tmp = self._file(f".{key}.{self.rank}.tmp")
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(data)
        os.replace(tmp, self._file(key))         # atomic: a reader sees the whole value or nothing

    def _read_when_there(self, key, what):
        p = self._file(key)
        self._wait(lambda: os.path.exists(p), what)
        with open(p, "rb") as f:
            return f.read()

    def set(self, key, data):
        self._write(f"{self.nonce}.{key}", data)

    def get(self, key):
        return self._read_when_there(f"{self.nonce}.{key}", f"key {key!r} never appeared")

    # ---- collectives (every rank calls them in the same order) ----
    def allgather(self, obj):
        self._seq += 1
        self.set(f"g{self._seq}.{self.rank}", dumps(obj))
        return [loads(self.get(f"g{self._seq}.{r}")) for r in range(self.world)]

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, x):
        return max(self.allgather(x))

    def bcast(self, obj, root=0):
        self._seq += 1
        if self.rank == root:
            self.set(f"b{self._seq}", dumps(obj))
            return obj
        return loads(self.get(f"b{self._seq}"))

    # ---- the end: everybody says good-bye, rank 0 removes the directory ----
    def close(self, wait=5.0):
        if self._closed:
            return
        self._closed = True
        try:
            if self.rank != 0:
                self._write(f"{self.nonce}.bye.{self.rank}", b"1")
                return
            t0 = time.monotonic()
            while time.monotonic() - t0 < wait and not all(
                    os.path.exists(self._file(f"{self.nonce}.bye.{r}")) for r in range(1, self.world)):
                time.sleep(0.005)
            shutil.rmtree(self.path, ignore_errors=True)
        except OSError:
            pass
