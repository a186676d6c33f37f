"""N > 1 host logic without a GPU: world-size-2 gloo processes on CPU tensors (the kernels are covered by test_gpu_multirank)."""
import os

import numpy as np
import torch

from tests import mp_workers


def _spawn(target, args_of_rank, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 23000 + os.getpid() % 4000
    procs = [ctx.Process(target=target, args=(r, world, port, q) + tuple(args_of_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_agent_sharded_exchange_world_2_gloo():
    res = _spawn(mp_workers.exchange_worker, (), 2)
    assert [r[1:3] for r in res] == [(0, 4), (4, 8)]
    assert all(r[3] for r in res)


def test_interleave_matches_the_device_kernel_formula():
    """interleave_agent_blocks == the index map of interleave_kernel (csrc/mpcx_comm.hip): all[b][r*A_loc + a] = xchg[r][b][a]"""
    from mpc_for_av_at_intersection_amd.sharding import interleave_agent_blocks, shard_agents
    world, B, a_loc = 4, 3, 2
    x = torch.arange(world * B * a_loc * 6, dtype=torch.float64).reshape(world, B, a_loc, 6)
    got = interleave_agent_blocks(x)
    want = torch.empty(B, world * a_loc, 6, dtype=torch.float64)
    for r in range(world):
        for b in range(B):
            for a in range(a_loc):
                want[b, r * a_loc + a] = x[r, b, a]
    assert torch.equal(got, want)
    assert [shard_agents(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    try:
        shard_agents(8, 0, 3)
        assert False
    except ValueError:
        pass
