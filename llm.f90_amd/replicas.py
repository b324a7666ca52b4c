"""N>1 for a model that does not shard: N independent replicas, one process per GPU.

The decode path of a 1.1B (or 7B) model has no exchange step worth a collective -- a 0.7-1 ms token
cannot absorb 44+ all-reduces -- so `bench.py --gpus N` runs N replicas of the single-GPU path
(DESIGN.md section 5).  The only inter-process traffic is the driver's timing protocol: a barrier on both
sides of the timed region and a MAX over ranks of the elapsed time (torch.distributed; backend "nccl"
== RCCL on the GPU box, "gloo" in the CPU tests).  No collective touches the data path.
"""
from __future__ import annotations

import os


class Replicas:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                backend = os.environ.get("LLMK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                torch.cuda.set_device(self.local)
                self.device = torch.device("cuda", self.local)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                self.device = torch.device("cpu")
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self) -> None:
        if self.dist is not None:
            import torch
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            self.dist.barrier()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def aggregate_rate(self, units_per_rank: int, seconds: float) -> float:
        """whole-job throughput: units of ALL replicas / max-over-ranks time (weak scaling)."""
        return self.world * units_per_rank / self.max_over_ranks(seconds)

    def close(self) -> None:
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
