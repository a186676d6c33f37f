"""dev aid (GPU box): the benchmark workload as S instance shards on S contexts / HIP streams in flight at once (instances are
independent; a launch is bound by its slowest problems, so a second shard fills the SIMDs the first one's stragglers leave idle)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpc_for_av_at_intersection_amd.runtime import Context
from mpc_for_av_at_intersection_amd.batch import stock_routes, synthetic_batch
from mpc_for_av_at_intersection_amd.sharding import shard_instances

B, K, W = 4096, 20, 3
for S in [int(v) for v in os.environ.get('SHARDS', '1,2,4').split(',')]:
    streams = [torch.cuda.Stream() for _ in range(S)]
    ctxs = [Context(0, stream=s) for s in streams]
    with torch.cuda.stream(streams[0]):
        routes, dl, cd = stock_routes(ctxs[0])
    torch.cuda.synchronize()
    sims = []
    for r in range(S):
        with torch.cuda.stream(streams[r]):
            sims.append(synthetic_batch(ctxs[r], B=B, A=8, T=20, seed=0, routes=routes, dl=dl, cd=cd, instance_slice=shard_instances(B, r, S)))
    torch.cuda.synchronize()
    for graph in (False, True):
        for r, s in enumerate(sims):
            with torch.cuda.stream(streams[r]):
                s.run(W, graph=graph)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for r, s in enumerate(sims):
                with torch.cuda.stream(streams[r]):
                    s.run(K, graph=graph)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print('S=%d graph=%d: %.3f ms/step  %.2f M ts/s' % (S, graph, best / K * 1e3, B * 8 * K / best / 1e6), flush=True)
    for c in ctxs:
        c.close()
