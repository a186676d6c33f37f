"""Soak: closed loop of the bench workload for many steps; counts QP failures, max iterations, and (second run with
MPCX_QP_KERNEL=wave in a child process) compares the two solvers' trajectories."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from mpc_for_av_at_intersection_amd.batch import synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context
    ctx = Context(0)
    steps = int(sys.argv[2]); B = int(sys.argv[3]); seed = int(sys.argv[4])
    sim = synthetic_batch(ctx, B=B, A=8, T=20, seed=seed)
    fails = 0; itmax = 0; itsum = 0.0
    for k in range(steps):
        sim.step()
        st = sim.sol['status']; it = sim.sol['iters']
        fails += int((st != 0).sum()); itmax = max(itmax, int(it.max())); itsum += float(it.float().mean())
    snap = sim.snapshot()
    np.save(sys.argv[5], snap['state'])
    print('kernel=%s steps=%d B=%d seed=%d: failures %d, max iters %d, mean iters %.3f, mean speed %.2f' % (
        os.environ.get('MPCX_QP_KERNEL', 'stage'), steps, B, seed, fails, itmax, itsum / steps, snap['state'][:, 2].mean()), flush=True)
else:
    steps, B = 120, 2048
    for seed in (1, 2):
        outs = []
        for kern in ('stage', 'wave'):
            env = dict(os.environ); env['MPCX_QP_KERNEL'] = kern
            out = 'gpurun_out/soak_%s_%d.npy' % (kern, seed)
            subprocess.run([sys.executable, __file__, 'child', str(steps), str(B), str(seed), out], env=env, check=True)
            outs.append(np.load(out))
        d = np.abs(outs[0] - outs[1])
        print('seed %d: state deviation stage vs wave after %d steps: max %.3e, 99.9%% quantile %.3e' % (seed, steps, d.max(), np.quantile(d, 0.999)), flush=True)
