"""Data-parallel sharding of prompts + the one collective of the multi-GPU path.

Sequences are independent, so N GPUs decode N shards of the prompt set with a full
draft+target replica each and no exchange inside the loop. At the end every rank
contributes one fixed struct of int64 counters (tokens, proposals, acceptances, wall time, steps, its NUMA node and CPU count) and a single all-gather (RCCL over xGMI on
the GPUs, gloo in the CPU tests) lets rank 0 form the whole-job numbers:
throughput = sum(tokens) / max(wall time). The reference has no distributed code to mirror
("Parallelism strategies: absent")."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Sequence

import torch

FIELDS = ("tokens", "proposed", "accepted", "accepted_strict", "wall_ns", "steps", "numa_node", "cpus")


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin: item i goes to rank i % world (BASELINE config 4: 32 prompts -> 4 per GPU)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


@dataclass
class JobStats:
    per_rank: torch.Tensor  # int64 [world][len(FIELDS)]

    def total(self, name: str) -> int:
        return int(self.per_rank[:, FIELDS.index(name)].sum())

    def max_wall_s(self) -> float:
        return float(self.per_rank[:, FIELDS.index("wall_ns")].max()) / 1e9

    def tokens_per_s(self) -> float:
        w = self.max_wall_s()
        return self.total("tokens") / w if w > 0 else 0.0

    def column(self, name: str) -> List[int]:
        return [int(x) for x in self.per_rank[:, FIELDS.index(name)]]

    def acceptance(self, strict: bool = False) -> float:
        return self.total("accepted_strict" if strict else "accepted") / max(self.total("proposed"), 1)


def gather_stats(local: Dict[str, int], device: torch.device, group=None) -> JobStats:
    """all_gather of the 64-byte counter struct (missing fields count as -1); with no process group it is the identity."""
    import torch.distributed as dist

    row = torch.tensor([int(local.get(f, -1)) for f in FIELDS], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return JobStats(row.cpu().unsqueeze(0))
    world = dist.get_world_size(group)
    out = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(out, row, group=group)
    return JobStats(torch.stack(out).cpu())
