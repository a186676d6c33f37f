"""Multi-rank layouts on the GPU box: 2 processes share cuda:0 (rows travel over gloo), results must be BIT-IDENTICAL to the
single-rank run of the same workload; plus the RCCL entry points on a one-rank communicator."""
import os

import numpy as np
import pytest
import torch

from tests import mp_workers

pytestmark = pytest.mark.gpu
KEYS = ('state', 'applied', 'u', 'x', 'status', 'iters', 'traj_idx', 'target_ind', 'cut_len', 'hit_idx')


def _reference(B, steps, seed):
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context
    ctx = Context(0)
    sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=seed)
    sim.run(steps)
    sim.check()
    return sim.snapshot()


def _run_ranks(layout, world, B, steps, seed, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    port = 24000 + os.getpid() % 4000
    out = str(tmp_path / ('%s_%%d.npz' % layout))
    procs = [ctx.Process(target=mp_workers.closed_loop_worker, args=(r, world, port, layout, B, steps, seed, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return [np.load(out % r) for r in range(world)]


def test_agent_sharded_two_ranks_bit_identical(tmp_path):
    """agent a of every instance on rank a // 4; the pool is all-gathered before the conflict search every step"""
    B, steps, seed = 24, 12, 7
    ref = _reference(B, steps, seed)
    parts = _run_ranks('agents', 2, B, steps, seed, tmp_path)
    for k in KEYS:
        full = ref[k].reshape((B, 8) + ref[k].shape[1:])
        for r, part in enumerate(parts):
            got = part[k].reshape((B, 4) + part[k].shape[1:])
            assert np.array_equal(got, full[:, 4 * r:4 * r + 4]), (k, r)
    # the exchange mattered: within these steps some agent found a conflict with another agent and had its path cut
    assert (ref['hit_idx'] >= 0).any(), 'no conflict in the reference run: the coupling was not exercised'


def test_instance_sharded_two_ranks_bit_identical(tmp_path):
    """rank r owns instances shard_instances(B, r, 2) of the ONE workload; no exchange on the data path"""
    from mpc_for_av_at_intersection_amd.sharding import shard_instances
    B, steps, seed = 25, 10, 11
    ref = _reference(B, steps, seed)
    parts = _run_ranks('instances', 2, B, steps, seed, tmp_path)
    for k in KEYS:
        full = ref[k].reshape((B, 8) + ref[k].shape[1:])
        for r, part in enumerate(parts):
            lo, hi = shard_instances(B, r, 2)
            assert np.array_equal(part[k].reshape((hi - lo, 8) + part[k].shape[1:]), full[lo:hi]), (k, r)


def test_rccl_entry_points_one_rank():
    """mpcx_comm_unique_id / mpcx_comm_init / mpcx_allgather_states / mpcx_closed_loop_run(exchange = agents) on a communicator of
    ONE rank (all this box has): the RCCL calls really run; the run must equal the plain single-rank run bit for bit"""
    from mpc_for_av_at_intersection_amd import _lib
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context
    B, steps, seed = 16, 6, 3
    ref = _reference(B, steps, seed)
    ctx = Context(0)
    ctx.comm_init(1, 0, ctx.comm_unique_id())
    loc = torch.arange(B * 8 * 6, dtype=torch.float64, device=ctx.device).reshape(B, 8, 6)
    for layout in (_lib.SHARD_INSTANCES, _lib.SHARD_AGENTS):
        out = torch.zeros(B * 8 * 6, dtype=torch.float64, device=ctx.device)
        ctx.allgather_states(layout, loc, out)
        ctx.synchronize()
        assert torch.equal(out, loc.reshape(-1))
    sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=seed, agent_shard=(0, 1), exchange='rccl')
    sim.run(steps)
    sim.check()
    snap = sim.snapshot()
    for k in KEYS:
        assert np.array_equal(snap[k], ref[k]), k
    ctx.comm_destroy()


def test_stale_context_horizon_is_reclaimed():
    """another user of the same Context changes T between two runs of a batch"""
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
    ctx = Context(0)
    a = synthetic_batch(ctx, B=4, A=8, T=20, seed=5)
    b = synthetic_batch(ctx, B=4, A=8, T=20, seed=5)
    a.run(3)
    ctx.set_mpc_params(MpcParams(T=13))        # e.g. a lib.MPC(T=13).step on the session context
    a.run(3)
    b.run(6)
    sa, sb = a.snapshot(), b.snapshot()
    for k in KEYS:
        assert np.array_equal(sa[k], sb[k]), k
