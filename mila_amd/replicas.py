"""Replica plumbing for bench.py --gpus N: the decode path does not shard, so N GPUs
run N independent replicas.  The only cross-rank traffic is the barrier around the timed region and the
MAX-over-ranks reduction of the measured time -- never a data-path collective."""
import os


def local_device():
    """HIP device ordinal of this replica: one process per GPU, the launcher's LOCAL_RANK (0 when run alone)."""
    return int(os.environ.get("LOCAL_RANK", "0"))


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch.distributed as dist
            # gloo: the replicas exchange two host scalars per run (barrier, MAX of the measured time); the north star rules RCCL out
            # of this path and nothing here needs it
            self.backend = backend or "gloo"
            if self.backend != "gloo":
                raise ValueError("replicas: only the gloo backend is used (no data-path collective, no RCCL)")
            dist.init_process_group("gloo")
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """MAX reduction of a python float (timing only)."""
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def aggregate_throughput(self, units_per_rank, ms_this_rank):
        """whole-job throughput = units all ranks processed / max-over-ranks time (weak scaling)."""
        ms = self.max_over_ranks(ms_this_rank)
        return units_per_rank * self.world / (ms * 1e-3), ms

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
