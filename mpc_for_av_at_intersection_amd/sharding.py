"""Multi-GPU layouts: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

INSTANCE-SHARDED (the throughput layout, what bench.py times).  Scenario instances are independent (the reference has no
cross-instance state) and all agents of an instance live on one rank, so the interaction coupling (prediction of the other
agents + conflict search) is rank-local and the data path needs NO collective: rank r owns instances
shard_instances(B, r, world) of the one workload.  `gather_agent_states` collects every rank's agent 6-tuples for logging.

AGENT-SHARDED (the layout with a real exchange step).  Rank r drives agents r*A/world .. of EVERY
instance and sees the other ranks' agents only as moving obstacles: per step ONE all-gather of 6-double agent states
(x, y, v, yaw, accel, steer -- what MovingObstacle*.get() returns, mpc_intersection.py:119-122), 48 B per agent, after which
every rank predicts the others itself (the prediction is a deterministic rollout of those six numbers).  On GPUs the exchange
is mpcx_allgather_states (RCCL over xGMI, csrc/mpcx_comm.hip) inside mpcx_closed_loop_run; `torch_exchange` is the same
exchange through torch.distributed (any backend: gloo rehearsals on CPU tensors or on a shared GPU).  Results are bit-identical
to the single-rank run of the same workload (tests/test_gpu_multirank.py)."""
from typing import Tuple

import torch
import torch.distributed as dist


def shard_instances(n_instances: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous [lo, hi) of the instance range owned by `rank`; sizes differ by at most one"""
    base, extra = divmod(n_instances, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_agents(n_agents: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the agents of every instance driven by `rank` (agent-sharded layout; world must divide n_agents)"""
    if n_agents % world:
        raise ValueError('%d agents do not divide over %d ranks' % (n_agents, world))
    a_loc = n_agents // world
    return rank * a_loc, (rank + 1) * a_loc


def gather_agent_states(local: torch.Tensor, n_instances: int, rank: int, world: int) -> torch.Tensor:
    """instance-sharded: local (n_local, A, 6) -> (n_instances, A, 6) on every rank, in instance order (one all-gather)"""
    if world == 1:
        return local
    sizes = [shard_instances(n_instances, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * width:r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def interleave_agent_blocks(blocks: torch.Tensor) -> torch.Tensor:
    """(world, B, A_loc, 6) rank blocks -> (B, world * A_loc, 6) pool in (instance, global agent) order -- what
    interleave_kernel (csrc/mpcx_comm.hip) does on the device after the RCCL all-gather"""
    world, B, a_loc, k = blocks.shape
    return blocks.permute(1, 0, 2, 3).reshape(B, world * a_loc, k).contiguous()


def torch_exchange(world: int, staging_device=None):
    """agent-sharded exchange through torch.distributed: returns f(local (B, A_loc, 6)) -> pool (B, world*A_loc, 6).
    staging_device='cpu' moves the rows through host memory (gloo with ranks sharing one GPU)."""
    def exchange(local: torch.Tensor) -> torch.Tensor:
        if world == 1:
            return local
        src = local.contiguous() if staging_device is None else local.to(staging_device).contiguous()
        out = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src)         # rank blocks one after the other along dim 0
        return interleave_agent_blocks(out.view((world,) + tuple(src.shape))).to(local.device)
    return exchange


def init_comm(ctx, rank: int, world: int):
    """create the context's RCCL communicator: rank 0 makes the unique id, torch.distributed broadcasts its bytes"""
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(world, rank, box[0])
