"""configs[3] as configured on one GPU (bench.py's c4_strong leg: 20 sub-spot chunks of 10 000 cells against 50 000 spots, 2 000 genes) and the
50-chunk single-cell leg, each alone, with the chunk call's own wall-clock stamps (CYTO_TRACE_CHUNKS=1) -- where the time of a chunked call
goes beyond its kernels (developer tool)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cytospace_amd import _lib  # noqa: E402

comm = _lib.Communicator(_lib.Communicator.unique_id(), 0, 1, device_id=0)
for rep in range(2):
    t = time.perf_counter()
    r = bench.extra_c4_sharded(0, 0, 1, None, comm, 0, G=2000, total_chunks=20)
    print(f"c4_strong rep {rep}: {r['seconds']} s (leg wall {time.perf_counter() - t:.2f} s incl. the generator) longest LAP batch {r['rank0_longest_lap_kernel_ms']} ms", flush=True)
r = bench.extra_c5_chunks(0)
print(f"c5_chunks: {r['wall_s']} s, cost builds {r['cost_build_ms_total']} ms, chunk0 {r['chunk0']}", flush=True)
comm.close()
