"""Rehearsal of bench.py's N-rank flow on a box with ONE GPU: the same code path the driver starts on an 8-GPU node (the
self-launcher, one process per rank, barrier + max-over-ranks timing, exactly one JSON line from rank 0), with the ranks
SHARING the visible GPU(s) over gloo because RCCL refuses two ranks on one device.  The line carries a "rehearsal" tag and
its timings are not a result.  Same flags as bench.py: python tools/bench_rehearsal.py --gpus 2 --steps 3 --warmup 1

With --world1-rccl as the FIRST argument it does something else: one rank, one GPU, but through the N-rank code path (gloo
control plane + the in-library RCCL communicator of world 1) -- how the sharded modes' lines of
profiles/rNN_bench_modes_n1.jsonl are made on a 1-GPU box; those timings ARE measurements."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "--world1-rccl":
    del sys.argv[1]
    bench.FORCE_DIST = True
else:
    bench.REHEARSAL = True
    bench.ENTRY = os.path.abspath(__file__)   # every rank must come up through this file, not bench.py

if __name__ == "__main__":
    bench.main()
