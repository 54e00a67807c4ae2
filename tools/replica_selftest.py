"""CPU rehearsal of bench.py's N>1 path (gloo): each rank 'measures' a different time; rank 0 prints the
aggregate exactly as bench.py does.  Launched by tests/test_replicas_cpu.py through torch.distributed.run."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mila_amd.replicas import Ranks  # noqa: E402

r = Ranks(backend="gloo")
r.barrier()
ms = 10.0 + 5.0 * r.rank           # rank 1 is the slow one
tput, worst = r.aggregate_throughput(units_per_rank=64, ms_this_rank=ms)
r.barrier()
if r.rank == 0:
    print(json.dumps({"world": r.world, "value": tput, "ms": worst}))
r.close()
