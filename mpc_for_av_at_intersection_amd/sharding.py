"""Multi-GPU layout: one process per GPU, scenario instances sharded contiguously over ranks.

Instances are independent (the reference has no cross-instance state), and all agents of an instance live on the same
rank, so the interaction coupling (prediction of the other agents + conflict search) is rank-local and the data path
needs NO collective: throughput scales by replication (bench.py reports "scaling": "weak").  `gather_agent_states` is
the one exchange the layout can need -- collecting every rank's per-agent 6-tuples (x, y, v, yaw, a, steer), e.g. for
logging on rank 0 or for an agent-sharded variant -- as a single all-gather (RCCL on GPUs, gloo on CPU in tests)."""
from typing import Tuple

import torch
import torch.distributed as dist


def shard_instances(n_instances: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of the instance range owned by `rank`; sizes differ by at most one"""
    base, extra = divmod(n_instances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_agent_states(local: torch.Tensor, n_instances: int, rank: int, world: int) -> torch.Tensor:
    """local: (n_local, A, 6) -> (n_instances, A, 6) on every rank, in instance order (one all-gather)"""
    if world == 1:
        return local
    sizes = [shard_instances(n_instances, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * width:r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)
