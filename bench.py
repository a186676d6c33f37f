#!/usr/bin/env python3
"""bench.py -- headline benchmark of BASELINE.json: MPC timesteps/s, N=20 horizon, 8-agent intersection, batch=4096.

One "step" = one pass of the hot path over the batch: every (instance, agent) pair goes through the body of the reference's
scenario loop (conflict search + path cut, reference window, rollout, QP, plant) -- mpc_for_av_at_intersection_amd/batch.py.
One *timestep* of the metric = one INSTANCE (all 8 agents) advanced by DT = 0.2 s.

The workload is ONE batch of `--batch` (4096) instances whatever --gpus is: rank r owns instances
shard_instances(4096, r, N) -- strong scaling, no data-path collective (all agents of an instance are rank-local).
value = 4096 * steps / max-over-ranks time.  Extra keys of the same JSON line (each measured after the headline region):
  steady_state   the same measurement after >= 100 closed-loop steps (traffic has built up; different iteration mix)
  expand         motion-primitive expansion of a 2^20-node Prius frontier, HIP-event timed
  agent_sharded  (N > 1) the layout with a real exchange step: agents sharded over ranks, one RCCL all-gather per step
  weak           (N > 1) every rank runs the full 4096-instance batch
  config2        (N = 1) SURVEY 8(d) config 2 / BASELINE configs[1]: 256 independent perturbed-state instances (batch.config2_batch)
  free_flow      (N = 1) the same generator at 32 768 instances: a launch of 32 768 QPs none of which stands in a queue at the junction
  shard_proxy    (N = 1) the per-rank shares of the headline batch at N = 2 / 4 / 8 (2048 / 1024 / 512 instances) timed on this GPU
  cpu_baseline   the oracle on this host: one thread, and all cores (threads over agents in C)

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix rate (datasheet; 256 CU x 4 SIMD x 16 FMA lanes x 2 x 2.4 GHz)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md chip table
PMC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_final.csv")


def qp_flops_condensed(T: int, iters: float, extra_passes: float = 0.0) -> float:
    """algorithmic (structure-exploiting, condensed) FP64 flops of one agent-QP,
    F_qp = 16/3 T^3 + 16 T^2 + K (n^3/3 + 4 n^2 + 6 T^2 + 12 m) with `iters` = K interior-point iterations (0 for a QP the trial pass
    solves).  extra_passes = the passes that formula does not know, each ONE factorisation, ONE solve and one product with G (against
    two solves and four products in an iteration): the trial pass every QP runs since round 2 and the active-set polish passes of the
    constrained QPs since round 3."""
    n, m = 2 * T, 8 * T
    return ((16.0 / 3.0) * T ** 3 + 16.0 * T ** 2 + iters * (n ** 3 / 3.0 + 4.0 * n ** 2 + 6.0 * T ** 2 + 12.0 * m)
            + extra_passes * (n ** 3 / 3.0 + 2.0 * n ** 2 + 3.0 * T ** 2 + 4.0 * m))


def qp_flops_stage(T: int, iters: float) -> float:
    """FP64 flops the stage-structured solver (csrc/mpcx_qp_stage.h) needs per agent-QP, counted on its source (FMA = 2, one count
    per stage and iteration): Riccati step 300 (G = P F 25 FMA, Phi = F'G 14, elimination + gains 40, cost-to-go update 54, +
    adds), two forward sweeps 2 x 52, corrector vector sweep 60, costate sweep 16, four row passes over 8 rows 512 (each row:
    gap, reciprocal, products, ratio tests), two cost-gradient evaluations 50; set-up per stage 200 (two sincos, weights, two
    rollouts).  Recomputation (rows are recomputed where needed instead of being carried) and padding are NOT counted."""
    per_iter = 300 + 2 * 52 + 60 + 16 + 512 + 50
    per_trial = 300 + 52 + 16 + 256 + 25      # the trial pass: Riccati step, ONE forward sweep, costate sweep, two row passes, one gradient
    setup = 200
    return T * (setup + iters * per_iter + per_trial)


def pmc_executed_flops(kernel='qp_quad'):
    """FP64 flops the dominant kernel EXECUTES per launch according to the committed PMC passes: SQ_INSTS_VALU_FLOPS_FP64 (flops
    per wavefront-instruction, FMA = 2) x mean active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU)."""
    if not os.path.exists(PMC_FILE):
        return None
    v = {}
    for line in open(PMC_FILE):
        parts = line.strip().rsplit(',', 2)
        if len(parts) == 3 and parts[0].startswith('mpcx::' + kernel):
            v[parts[1]] = float(parts[2])
    try:
        lanes = v['SQ_THREAD_CYCLES_VALU'] / v['SQ_INSTS_VALU']
        return dict(flops_per_launch=v['SQ_INSTS_VALU_FLOPS_FP64'] * lanes, mean_active_lanes=lanes,
                    f64_share_of_valu=(v['SQ_INSTS_VALU_ADD_F64'] + v['SQ_INSTS_VALU_MUL_F64'] + v['SQ_INSTS_VALU_FMA_F64'] + v['SQ_INSTS_VALU_TRANS_F64']) / v['SQ_INSTS_VALU'])
    except KeyError:
        return None


def agent_step_bytes(T: int, A: int) -> float:
    """algorithmic HBM bytes per agent-step."""
    return 32 + 16 * T + 48 * (A - 1) + 24 * (T + 1) + 32 * (T + 1) + 16 * T + 8


def qp_launch_bytes(T: int, P: int) -> float:
    """algorithmic HBM bytes of one QP launch: x0, xref, xbar, reaches_end, warm start in; x, u, status/iters/kkt out"""
    return P * (32 + 2 * 32 * (T + 1) + (T + 1) + 16 * T + 32 * (T + 1) + 16 * T + 40)


def pmc_traffic_bytes(kernel='mpcx::qp_quad_kernel'):
    """HBM bytes per launch of the dominant kernel from the COMMITTED rocprofv3 PMC passes (bench.py cannot run the profiler on
    itself): 2 x FETCH_SIZE (gfx950 under-reports reads by 2x, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE, KiB -> bytes."""
    if not os.path.exists(PMC_FILE):
        return None
    vals = {}
    for line in open(PMC_FILE):
        parts = line.strip().rsplit(',', 2)          # kernel names contain commas (template arguments)
        if len(parts) == 3 and parts[0].startswith(kernel) and parts[1] in ('FETCH_SIZE', 'WRITE_SIZE'):
            vals[parts[1]] = float(parts[2])
    if len(vals) != 2:
        return None
    return (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0


def host_cores():
    """(threads this process may really use, CPU model string)"""
    n = len(os.sched_getaffinity(0))
    try:                                            # cgroup v2 CPU quota of the box
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(math.ceil(float(quota) / float(period)))))
    except Exception:
        pass
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except Exception:
        pass
    return n, model


def cpu_baseline(sim, snap, n_agents: int):
    """Oracle (CPU port of the same per-agent step, oracle/) timed on this host on a bounded sample of the SAME workload: the
    first n (instance, agent) pairs of rank 0's batch from the captured state -- once on one thread, once on all cores, both
    through orc_agent_steps_mt (C, pthreads over agents; no Python per agent-step)."""
    from oracle import oracle_py as orc
    A = sim.A
    po = orc.MpcParams(T=sim.params.T, L=sim.params.L)
    tab = sim.path.cpu().numpy(); off = sim.path_off.cpu().numpy(); ln = sim.path_len.cpu().numpy()
    centers = np.asarray(sim.ip.circle_centers).reshape(2, 2)
    cores, model = host_cores()

    P = len(snap['state'])

    def run(n, threads, seconds):
        """passes over the first n agents until `seconds` have gone by; returns (agent-steps done, seconds, threads that ran)"""
        n = max(A, (min(n, P) // A) * A)
        done, ran, t0 = 0, threads, time.perf_counter()
        while True:
            r = orc.agent_steps_batch(po, threads, A, tab, off[:n], ln[:n], sim.dl, snap['state'][:n], snap['applied'][:n], snap['u'][:n],
                                      snap['traj_idx'][:n], snap['prev_cut'][:n], snap['target_ind'][:n], centers, sim.ip.radius,
                                      sim.ip.cutoff_margin)
            done += n; ran = r['threads']
            if time.perf_counter() - t0 >= seconds:
                return done, time.perf_counter() - t0, ran

    n1, t1, _ = run(min(n_agents, 4096), 1, 0.0)                 # one short pass: calibrates the sample size
    n1, t1, _ = run(min(n_agents, int(n1 / t1 * 2.5)), 1, 10.0)  # ~10 s on one thread, passes of ~2.5 s
    nm, tm, ran = run(min(n_agents, P), cores, 10.0)             # ~10 s on all cores, passes over the whole captured batch
    return dict(value=(nm / A) / tm, unit='MPC timesteps/s', cores=ran, kind='port', cpu_model=model,
                sample='%d agent-steps (= %d instance timesteps; passes over the %d captured agents of rank 0, state after warm-up), '
                       'oracle/liboracle.so orc_agent_steps_mt (C, %d pthreads over agents, gcc -O3), %.1f s' % (nm, nm // A, min(n_agents, P), ran, tm),
                agent_qp_per_s=nm / tm,
                value_1thread=(n1 / A) / t1, agent_qp_per_s_1thread=n1 / t1,
                sample_1thread='%d agent-steps on one thread, %.1f s' % (n1, t1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4096, help='scenario instances of the WHOLE job (sharded over the GPUs)')
    ap.add_argument('--agents', type=int, default=8)
    ap.add_argument('--horizon', type=int, default=20)
    ap.add_argument('--cpu-agents', type=int, default=1 << 20, help='cap of the CPU baseline sample (it is sized for ~10 s per leg)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='headline measurement only')
    ap.add_argument('--steady-steps', type=int, default=100, help='closed-loop steps before the steady-state measurement')
    ap.add_argument('--burn-in', type=int, default=3, help='closed-loop steps taken while the workload is built (SURVEY 8d config 2: "run 3 burn-in steps"), '
                                                            'so that every timed step starts from a previous solution whatever --warmup is')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path is HIP-only')
    # rehearsal knob (single-GPU box): several ranks may share cuda:0 over gloo; the driver's runs use RCCL, one GPU per rank
    backend = os.environ.get('MPCX_DIST_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mpc_for_av_at_intersection_amd.batch import ALL_STOCK_PAIRS, config2_batch, prius_frontier, stock_routes, synthetic_batch
    from mpc_for_av_at_intersection_amd.runtime import Context, MpcParams
    from mpc_for_av_at_intersection_amd import sharding
    ctx = Context(local)
    routes, dl, cd = stock_routes(ctx)
    lo, hi = sharding.shard_instances(args.batch, rank, world)
    A, T = args.agents, args.horizon

    def make(**kw):
        return synthetic_batch(ctx, B=args.batch, A=A, T=T, seed=1000, routes=routes, dl=dl, cd=cd, **kw)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if backend == 'nccl':
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=ctx.device if backend == 'nccl' else 'cpu')
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device=ctx.device if backend == 'nccl' else 'cpu')
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    iters_sum = torch.zeros((), dtype=torch.float64, device=ctx.device)
    fail_sum = torch.zeros((), dtype=torch.float64, device=ctx.device)

    def timed(sim, steps):
        """EXACTLY `steps` steps between two barriers; returns (max-over-ranks seconds, IPM iterations summed over ranks, failures,
        qp ms per launch, launches, max iterations).  Iterations / failures come from the library's device-side run statistics
        (mpcx_closed_loop_stats) or, for the staged rehearsal exchange, from torch reductions."""
        staged = callable(sim.exchange)
        ctx.profile_qp(not os.environ.get('MPCX_BENCH_NO_QP_EVENTS')); ctx.profile_qp_read()      # (dev aid: what the event pairs around every QP launch cost the timed region)
        ctx.closed_loop_stats(reset=True)
        iters_sum.zero_(); fail_sum.zero_()
        barrier()
        t0 = time.perf_counter()
        if staged:
            for _ in range(steps):
                sim.step()
                iters_sum.add_(sim.sol['iters'].sum())
                fail_sum.add_((sim.sol['status'] != 0).sum())
        else:
            sim.run(steps)          # ONE mpcx_closed_loop_run call: `steps` steps enqueued back to back
        barrier()
        el = time.perf_counter() - t0
        qp_ms, qp_n = ctx.profile_qp_read()
        ctx.profile_qp(False)
        st = ctx.closed_loop_stats(reset=True)
        it_loc = float(iters_sum.item()) if staged else float(st['iterations'])
        fl_loc = float(fail_sum.item()) if staged else float(st['failures'])
        it = sum_over_ranks(it_loc); fl = sum_over_ranks(fl_loc)
        timed.max_iterations = int(max_over_ranks(float(st['max_iterations'])))      # of the region just timed
        return max_over_ranks(el), it, fl, qp_ms / max(qp_n, 1), qp_n

    # ------------------------------------------------------------------ headline: strong scaling over instances
    sim = make(instance_slice=(lo, hi))
    if args.burn_in > 0:
        sim.run(args.burn_in)
    iters_sum += sim.sol['iters'].sum()              # loads torch's lazily compiled reduction kernels outside any timed region,
    fail_sum += (sim.sol['status'] != 0).sum()       # even with --warmup 0
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        sim.step()
        iters_sum += sim.sol['iters'].sum()
        fail_sum += (sim.sol['status'] != 0).sum()
    torch.cuda.synchronize()
    snap = sim.snapshot() if (rank == 0 and not args.no_cpu) else None

    elapsed, it_total, failures, qp_ms, qp_launches = timed(sim, args.steps)
    max_iters = timed.max_iterations
    P_total = args.batch * A
    P_rank = sim.P
    mean_iters = it_total / (P_total * args.steps)
    value = args.batch * args.steps / elapsed
    stage = ((T > 20 or P_rank >= 11264   # MPCX_STAGE_MIN_BATCH of csrc/mpcx_qp.hip
              ) and os.environ.get('MPCX_QP_KERNEL') != 'wave') \
        or os.environ.get('MPCX_QP_KERNEL') == 'stage'
    # SURVEY 8(d)'s algorithmic count F_qp is the one the roofline is priced with (rounds stay comparable); the stage solver's own
    # need, counted on its source, is reported beside it
    trial_frac = float(((sim.sol['iters'] == 0) & (sim.sol['status'] == 0)).double().mean().item())
    # passes beside the iterations: the trial pass of every QP + the polish passes of the constrained ones (1.07 per constrained QP on the
    # benchmark workload: 7 % need a second pass)
    extra_passes = 1.0 + 1.07 * (1.0 - trial_frac)
    flops_survey = qp_flops_condensed(T, mean_iters)                      # SURVEY 8(d)'s formula as written
    flops_qp = qp_flops_condensed(T, mean_iters, extra_passes)           # + the trial / polish passes
    qp_ms = qp_ms if qp_ms > 0 else float('nan')        # (dev aid MPCX_BENCH_NO_QP_EVENTS: no kernel time)
    achieved_tf = flops_qp * P_rank / (qp_ms * 1e-3) / 1e12
    survey_tf = flops_survey * P_rank / (qp_ms * 1e-3) / 1e12
    own_tf = (qp_flops_stage(T, mean_iters) if stage else flops_qp) * P_rank / (qp_ms * 1e-3) / 1e12
    line = {
        'metric': 'MPC timesteps/sec (whole node), N=20 horizon, 8-agent intersection, batch=4096',
        'value': value, 'unit': 'MPC timesteps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'configs[2]/[3]: %d-agent coupled stock intersection, batch=%d instances in all (sharded over the GPUs), N=%d, '
                               'interaction (prediction + conflict search + path cut) on-device, seeded staggered starts, %d burn-in steps'
                               % (A, args.batch, T, args.burn_in),
                   'instances_total': args.batch, 'instances_per_gpu': hi - lo, 'agents': A, 'horizon': T,
                   'parallelism': 'instances sharded over %d GPU(s), no data-path collective' % world},
        'agent_qp_per_s': value * A, 'mean_ipm_iters': mean_iters, 'max_ipm_iters': max_iters, 'qp_failures': int(failures),
        'qp_solved_by_trial_pass': trial_frac,
        'roofline': {'bound': 'fp64_valu', 'achieved': achieved_tf, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved_tf / FP64_PEAK_TFLOPS,
                     'achieved_survey_formula': survey_tf, 'frac_survey_formula': survey_tf / FP64_PEAK_TFLOPS,
                     'flops_per_qp_survey_formula': flops_survey, 'extra_passes_per_qp': extra_passes,
                     'traffic': pmc_traffic_bytes() if (T == 20 and P_rank == 32768) else None,
                     'algorithmic_bytes_per_launch': qp_launch_bytes(T, P_rank),
                     'traffic_note': 'HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE from the committed rocprofv3 PMC passes (%s), same workload' % os.path.relpath(PMC_FILE, ROOT),
                     'kernel': ('qp_quad_kernel<8,%d> (stage-structured IPM, 8 lanes per QP)' % (2 if T <= 16 else 3 if T <= 24 else 4)) if stage
                               else 'qp_kernel<%d> (condensed IPM, one wavefront per QP)' % T,
                     'kernel_ms': qp_ms, 'kernel_launches_timed': qp_launches, 'flops_per_qp': flops_qp,
                     'executed_pmc': None,
                     'achieved_stage_count': own_tf, 'frac_stage_count': own_tf / FP64_PEAK_TFLOPS,
                     'note': 'mean_ipm_iters counts interior-point iterations; since round 2 every QP first runs a trial pass (unconstrained minimiser, accepted '
                             'when it violates no row: qp_solved_by_trial_pass of the QPs end there with 0 iterations), and since round 3 every constrained QP '
                             'ends with an active-set polish pass (exact minimiser instead of a sqrt(mu)-accurate iterate; fewer iterations): both are counted '
                             'as one factorisation + one solve each in `achieved` / `frac`; `*_survey_formula` is SURVEY 8(d)\'s F_qp(T, K) as written, '
                             'without those passes.  The launch is bound by its slowest wavefront (queue imbalance + max iterations x time per round), not by the mean.  '
                             'The kernel is pure FP64 VALU (SQ_INSTS_MFMA = 0): roof = FP64 vector rate 78.6 TFLOP/s.  achieved = SURVEY 8(d)\'s '
                             'algorithmic count F_qp(T, measured mean iterations) + the trial pass, x QPs per launch / HIP-event kernel time on the '
                             'launch stream (the count rounds 1 and 2 are compared by).  *_stage_count = the same with the flops the stage '
                             'solver\'s own algorithm needs (O(T) Riccati sweeps, counted on its source: about half of F_qp)'},
        'roofline_hbm': {'bound': 'hbm', 'achieved': agent_step_bytes(T, A) * P_total / (elapsed / args.steps) / 1e9, 'peak': HBM_PEAK_GBS * world,
                         'unit': 'GB/s', 'note': 'algorithmic bytes per whole step / step time; not the binding roof'},
    }
    line['roofline_hbm']['frac'] = line['roofline_hbm']['achieved'] / (HBM_PEAK_GBS * world)
    ex = pmc_executed_flops() if (stage and T == 20 and P_rank == 32768) else None
    if ex is not None:
        tf = ex['flops_per_launch'] / (qp_ms * 1e-3) / 1e12
        line['roofline']['executed_pmc'] = {'tflops': tf, 'frac': tf / FP64_PEAK_TFLOPS, 'mean_active_lanes_per_valu_instruction': ex['mean_active_lanes'],
                                            'f64_share_of_valu_instructions': ex['f64_share_of_valu'],
                                            'note': 'lane-flops the kernel executes (committed PMC passes: SQ_INSTS_VALU_FLOPS_FP64 x mean active lanes) / live kernel time; includes '
                                                    'recomputation, padded slots and masked work'}

    if not args.no_extras:
        # -------------------------------------------------------------- steady state: the same batch after >= 100 closed-loop steps
        try:
            more = args.steady_steps - sim.steps_done
            if more > 0:
                sim.run(more)
            el, it, fl, ms, _ = timed(sim, args.steps)
            sim.check()
            line['steady_state'] = {'value': args.batch * args.steps / el, 'unit': 'MPC timesteps/s', 'after_steps': int(sim.steps_done - args.steps),
                                    'ms_per_step': 1e3 * el / args.steps, 'mean_ipm_iters': it / (P_total * args.steps), 'qp_failures': int(fl),
                                    'kernel_ms': ms, 'note': 'same %d steps measurement, taken after the closed loop has run long enough '
                                                            'for traffic to build up at the junction (paths cut, agents queueing)' % args.steps}
            line['steady_state_value'] = line['steady_state']['value']
        except Exception as e:                       # an extra must never take the headline line down
            line['steady_state'] = {'error': repr(e)}
        # -------------------------------------------------------------- SURVEY 8(d) config 2 (BASELINE configs[1]) and the same generator at 32 768 QPs
        try:
            if world == 1:
                routes12, dl12, cd12 = stock_routes(ctx, ALL_STOCK_PAIRS)

                def perturbed(Bc):
                    s2 = config2_batch(ctx, B=Bc, T=T, seed=0, routes=routes12, dl=dl12, cd=cd12, burn_in=args.burn_in)
                    s2.run(args.warmup)
                    el2, it2, fl2, ms2, _ = timed(s2, args.steps)
                    s2.check()
                    k2 = it2 / (Bc * args.steps)
                    tf2 = float(((s2.sol['iters'] == 0) & (s2.sol['status'] == 0)).double().mean().item())
                    f2 = qp_flops_condensed(T, k2)
                    return {'instances': Bc, 'agents': 1, 'horizon': T, 'value': Bc * args.steps / el2, 'unit': 'MPC timesteps/s (= agent-QPs/s: one agent per instance)',
                            'ms_per_step': 1e3 * el2 / args.steps, 'kernel_ms': ms2, 'mean_ipm_iters': k2, 'max_ipm_iters': timed.max_iterations,
                            'qp_solved_by_trial_pass': tf2, 'qp_failures': int(fl2),
                            'frac_survey_formula': f2 * Bc / (ms2 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                            'mean_speed_at_end': float(s2.state[:, 2].mean().item())}
                line['config2'] = perturbed(256)
                line['config2']['note'] = ('SURVEY 8(d) config 2 = BASELINE configs[1]: 256 INDEPENDENT single-ego instances, route uniform over the 12 stock A* paths, arc '
                                           'position uniform, lateral offset N(0, 0.3 m), heading error N(0, 0.05 rad), v ~ U[0, 8.33 m/s] (seed 0), %d burn-in + %d warm-up steps, '
                                           'then %d timed closed-loop steps; condensed solver qp_kernel<%d> (one wavefront per QP)' % (args.burn_in, args.warmup, args.steps, T))
                line['free_flow'] = perturbed(32768)
                line['free_flow']['note'] = ('NOT the metric\'s configuration: config 2\'s generator at 32 768 instances -- a QP launch of the headline\'s size in which nobody '
                                             'queues at the junction (no other agents): most QPs are constrained (acceleration bound when slow, steering-rate bound in turns). '
                                             'frac_survey_formula = SURVEY 8(d) F_qp(T, measured mean iterations) x QPs / kernel time / 78.6 TFLOP/s')
        except Exception as e:
            line.setdefault('config2', {'error': repr(e)}); line.setdefault('free_flow', {'error': repr(e)})
        # -------------------------------------------------------------- the 8-GPU regime on one GPU: per-rank shares of the headline batch
        try:
            if world == 1:
                px = {}
                for n_rank in (2, 4, 8):
                    share = args.batch // n_rank
                    sp = make(instance_slice=(0, share))
                    sp.run(args.burn_in + args.warmup)
                    elp, itp, flp, msp, _ = timed(sp, args.steps)
                    px['n%d' % n_rank] = {'instances_per_gpu': share, 'ms_per_step': 1e3 * elp / args.steps, 'kernel_ms': msp,
                                          'timesteps_per_s_this_gpu': share * args.steps / elp,
                                          'implied_whole_job_value': args.batch * args.steps / elp,
                                          'implied_strong_scaling_factor': elapsed / elp, 'qp_failures': int(flp)}
                    del sp
                px['note'] = ('NOT a multi-GPU measurement: rank 0\'s share of the %d-instance batch at N = 2 / 4 / 8 (instances [0, %d/N)) timed alone on this GPU; '
                              'implied_* assume every rank takes as long as rank 0 and ignore the barrier.  Shares below 1408 instances (11 264 QPs) run the condensed '
                              'solver qp_kernel<%d>, one wavefront per QP' % (args.batch, args.batch, T))
                line['shard_proxy'] = px
        except Exception as e:
            line['shard_proxy'] = {'error': repr(e)}
        # -------------------------------------------------------------- the same workload at four times the batch: bound by work, not by the tail
        try:
            if world == 1:
                Bbig = 4 * args.batch
                big = synthetic_batch(ctx, B=Bbig, A=A, T=T, seed=1000, routes=routes, dl=dl, cd=cd)
                big.run(args.burn_in + args.warmup)
                el, it, fl, ms, _ = timed(big, args.steps)
                fq = qp_flops_condensed(T, it / (Bbig * A * args.steps), extra_passes)
                line['work_bound'] = {'instances': Bbig, 'value': Bbig * args.steps / el, 'unit': 'MPC timesteps/s', 'ms_per_step': 1e3 * el / args.steps,
                                      'kernel_ms': ms, 'mean_ipm_iters': it / (Bbig * A * args.steps), 'qp_failures': int(fl),
                                      'frac': fq * Bbig * A / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                                      'note': 'NOT the metric\'s configuration: the same seeded workload with 4x the instances on this GPU.  At the '
                                              'metric\'s batch a QP launch lasts as long as its slowest problem (max iterations x time per round); '
                                              'with 16 problems per lane group instead of 4 it is bound by the work, which is what the trial pass cut'}
                del big
        except Exception as e:
            line['work_bound'] = {'error': repr(e)}
        # -------------------------------------------------------------- the five-state controller of lib/mpc_jerk.py on the same workload
        try:
            if world == 1:
                jp = MpcParams.jerk()
                js = synthetic_batch(ctx, B=args.batch, A=A, seed=1000, routes=routes, dl=dl, cd=cd, mpc=jp)
                js.run(args.burn_in + args.warmup)
                el, it, fl, ms, _ = timed(js, args.steps)
                line['mpc_jerk'] = {'value': args.batch * args.steps / el, 'unit': 'MPC timesteps/s', 'horizon': jp.T, 'ms_per_step': 1e3 * el / args.steps,
                                    'kernel_ms': ms, 'mean_ipm_iters': it / (P_total * args.steps), 'max_ipm_iters': timed.max_iterations,
                                    'qp_failures': int(fl),
                                    'note': 'NOT the metric: the same instances driven by the controller of main/lib/mpc_jerk.py with its own '
                                            'constants (N = 13, jerk penalty, free initial acceleration state): qp_quad_kernel<8,2,*,true>, seven-state sweep'}
                del js
        except Exception as e:
            line['mpc_jerk'] = {'error': repr(e)}
        # -------------------------------------------------------------- A* expansion, config 5: 2^20-node Prius frontier, nodes sharded
        try:
            n_all = 1 << 20
            nlo, nhi = sharding.shard_instances(n_all, rank, world)
            gold_logs = None
            try:                                   # "plus all nodes of the golden expansion logs" (SURVEY 8(d) config 5): the Prius searches of the fixture
                ar = np.load(os.path.join(ROOT, 'tests', 'golden', 'astar_runs.npz'))
                gold_logs = np.concatenate([ar[k] for k in ar.files if k.startswith('mod_pri_') and k.endswith('/dbg_node')])
            except Exception:
                pass
            st = torch.cuda.current_stream(ctx.device)
            reps = 10

            def time_frontier(free_space):
                model, nodes = prius_frontier(ctx, n_all, seed=0, free_space=free_space, embed=gold_logs if free_space else None)
                mine = nodes[nlo:nhi].contiguous()
                out = ctx.expand(model, mine)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                barrier()
                e0.record(st)
                for _ in range(reps):
                    ctx.expand(model, mine, out=out)
                e1.record(st)
                torch.cuda.synchronize()
                ms_ = max_over_ranks(e0.elapsed_time(e1) / reps)
                ff = sum_over_ranks(float((out['collide'] == 0).sum().item())) / (n_all * model.n_prim)
                return model, ms_, ff
            _, ms_uniform, ff_uniform = time_frontier(False)
            model, ms, free_frac = time_frontier(True)
            bytes_node = 24 + model.n_prim * (24 + 8 + 1)
            fma_node = float(sum(model.n_pts_of) * 4 + sum(model.n_pts_of) * 3 * model.n_rows)
            nps = n_all / (ms * 1e-3)
            line['expand'] = {'nodes_per_s': nps, 'ms': ms, 'nodes': n_all, 'primitives': model.n_prim, 'half_plane_rows': model.n_rows,
                              'obstacles': model.n_obst, 'free_fraction': free_frac,
                              'algorithmic_bytes_per_node': bytes_node, 'hbm_GBps': nps * bytes_node / 1e9,
                              'traffic': pmc_traffic_bytes('frontier:mpcx::expand_coop_kernel') if world == 1 else None,
                              'algorithmic_bytes_per_launch': bytes_node * n_all,
                              'hbm_frac': nps * bytes_node / 1e9 / (HBM_PEAK_GBS * world),
                              'fp64_fma_per_node_no_early_out': fma_node, 'fp64_equiv_frac_no_early_out': nps * fma_node * 2 / 1e12 / (FP64_PEAK_TFLOPS * world),
                              'golden_log_nodes_embedded': 0 if gold_logs is None else int(len(gold_logs)),
                              'uniform_frontier': {'ms': ms_uniform, 'nodes_per_s': n_all / (ms_uniform * 1e-3), 'free_fraction': ff_uniform,
                                                   'hbm_frac': n_all / (ms_uniform * 1e-3) * bytes_node / 1e9 / (HBM_PEAK_GBS * world),
                                                   'note': 'round 2\'s frontier (x, y uniform over the junction area whatever stands there): kept for comparison'},
                              'note': 'expand_kernel on the frontier SURVEY 8(d) config 5 defines: 2^20 nodes (seed 0) sampled from FREE SPACE (poses at which the car '
                                      'itself collides are rejected) with theta ~ U[-pi, pi), plus the nodes of the golden Prius expansion logs; Prius primitives + '
                                      'PriusDimensions, stock intersection obstacles; HIP events on the launch stream, mean of %d launches, nodes sharded evenly over the '
                                      'GPUs (no exchange); bytes = 24 B read + P*(24+8+1) B written per node (SURVEY 8d); the FP64 figure is what the node rate would '
                                      'cost WITHOUT early-outs (every point x row test); it can exceed 1 because exact box culling skips most tests' % reps}
        except Exception as e:
            line['expand'] = {'error': repr(e)}
        # -------------------------------------------------------------- device-resident A* (SURVEY 8(f)-2): 1024 searches in one launch
        try:
            if world == 1:
                from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
                from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
                from mpc_for_av_at_intersection_amd.lib.motion_primitive_search import MotionPrimitiveSearch, plan_many, plan_many_device
                from mpc_for_av_at_intersection_amd.lib.scenario import intersection
                cdb, mpsb = BicycleModelDimensions(), load_motion_primitives('bicycle_model')
                pairs = [(sp, ti) for sp in (1, 2, 3, 4) for ti in (1, 2, 3)]
                mk = lambda n: [MotionPrimitiveSearch(intersection(turn_indicator=pairs[i % 12][1], start_pos=pairs[i % 12][0]), cdb, mpsb, margin=cdb.radius,
                                                      variant='modified', ctx=ctx) for i in range(n)]
                plan_many_device(mk(1024))             # warm-up at full size: module load, heading table, the allocator's first big blocks
                ss = mk(1024)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res, inf = plan_many_device(ss)
                torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
                hs = mk(12)
                t0 = time.perf_counter()
                href = plan_many(hs)
                t_host = time.perf_counter() - t0
                same = all(res[i][0] == href[i][0] and res[i][1] == href[i][1] for i in range(12))
                line['device_search'] = {'searches': 1024, 'ms_total': 1e3 * t_dev, 'ms_device': 1e3 * inf['t_device'], 'ms_host_check': 1e3 * inf['t_check'], 'ms_host_results': 1e3 * inf['t_results'],
                                         'launches': inf['launches'],
                                         'heuristic_overrides': inf['overrides'], 'expansions_total': int(sum(inf['expansions'])),
                                         'host_queue_ms_for_12': 1e3 * t_host, 'host_queue_ms_scaled_to_1024': 1e3 * t_host * 1024 / 12,
                                         'same_cost_and_path_as_host_search': bool(same),
                                         'note': 'mpcx_astar_batch: the 12 stock routes (`modified` heuristic) replicated to 1024 independent searches, open list + closed set + '
                                                 'successor generation resident on the device, one wavefront per search, the reference\'s pop order (golden runs replayed node '
                                                 'for node in tests/test_gpu_astar.py); ms_total includes the host side (cos/sin table, check of the heuristic values against '
                                                 'Python floats, result copies and 1024 trajectory assemblies), second call of the process (the first also pays the allocator\'s first big blocks); host_queue_* = plan_many (exact host queues + batched expansion)'}
                # the three variants round 4 moved onto the device (+ base), on the non-stock worlds: edge values computed in the kernel
                from mpc_for_av_at_intersection_amd.lib.scenario import world as load_world
                vcases = [('base', 't_intersection/1_1', {}), ('multi_lane', 'intersection_multi_lanes/1_1_2_1_2', {}),
                          ('multi_lane', 'intersection_multi_lanes/3_2_1_2_3', dict(wh_obstacle=0.2, wc_center=0.02)), ('roundabout', 'roundabout/1_3', {}),
                          ('roundabout', 'roundabout_big/1_1', {}), ('roundabout', 'roundabout_big/2_2', {}), ('single_lane', 'intersection/2_3', {}),
                          ('single_lane', 'intersection/4_1', {})]
                mkv = lambda n: [MotionPrimitiveSearch(load_world(vcases[i % len(vcases)][1]), cdb, mpsb, margin=cdb.radius, variant=vcases[i % len(vcases)][0], ctx=ctx,
                                                       **vcases[i % len(vcases)][2]) for i in range(n)]
                plan_many_device(mkv(256))
                sv = mkv(256)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                resv, infv = plan_many_device(sv)
                torch.cuda.synchronize(); t_v = time.perf_counter() - t0
                hv = mkv(len(vcases))
                same_v = all(resv[i][0] == h.run()[0] for i, h in enumerate(hv))
                line['device_search_variants'] = {'searches': 256, 'variants': sorted({c[0] for c in vcases}), 'ms_total': 1e3 * t_v, 'ms_device': 1e3 * infv['t_device'],
                                                  'launches': infv['launches'], 'overrides': infv['overrides'], 'expansions_total': int(sum(infv['expansions'])),
                                                  'same_cost_as_host_search': bool(same_v),
                                                  'note': '8 golden cases of tests/golden/astar_worlds.npz (roundabouts, T- and multi-lane intersections; multi_lane with default and '
                                                          'non-default weights, roundabout, single_lane, base) replicated to 256 searches: heuristic AND edge values evaluated in the '
                                                          'kernel, every free successor logged and checked on the host, re-runs with per-search overrides included'}
        except Exception as e:
            line.setdefault('device_search', {'error': repr(e)})
            line.setdefault('device_search_variants', {'error': repr(e)})
        # the two multi-rank extras create a second communicator and a second batch: a rank that fails inside one of them would
        # leave the others waiting in a collective, so a watchdog on every rank gives them a deadline, after which rank 0 prints
        # the line it has (headline + the extras already measured) and every rank leaves with exit code 3
        watchdog = None
        if world > 1:
            import threading

            def bail():
                if rank == 0:
                    line.setdefault('agent_sharded', {'error': 'deadline of 180 s exceeded (a rank failed or the exchange hung)'})
                    print(json.dumps(line), flush=True)
                os._exit(3)             # the line says what was measured; the exit code says the run did not end properly
            watchdog = threading.Timer(180.0, bail)
            watchdog.daemon = True
            watchdog.start()
        # -------------------------------------------------------------- agent-sharded layout: one RCCL all-gather per step
        if world > 1 and A % world == 0:
            try:
                if backend == 'nccl':
                    sharding.init_comm(ctx, rank, world)
                    sim_a = make(agent_shard=(rank, world), exchange='rccl')
                else:
                    sim_a = make(agent_shard=(rank, world), exchange=sharding.torch_exchange(world, 'cpu'))
                sim_a.run(args.burn_in + args.warmup)
                # the pool every rank assembled must be the one torch.distributed assembles
                ref_pool = sharding.torch_exchange(world, None if backend == 'nccl' else 'cpu')(sim_a.obs_local.view(sim_a.B, sim_a.A, 6))
                same = bool(torch.equal(ref_pool.reshape(-1, 6), sim_a.obs6))
                el, it, fl, ms, _ = timed(sim_a, args.steps)
                sim_a.check()
                line['agent_sharded'] = {'value': args.batch * args.steps / el, 'unit': 'MPC timesteps/s', 'ms_per_step': 1e3 * el / args.steps,
                                         'agents_per_rank': sim_a.A, 'allgather_bytes_per_rank_per_step': sim_a.P * 48,
                                         'exchange': 'mpcx_allgather_states (RCCL over xGMI) inside mpcx_closed_loop_run' if backend == 'nccl' else 'torch.distributed %s (rehearsal)' % backend,
                                         'pool_matches_torch_all_gather': same, 'mean_ipm_iters': it / (P_total * args.steps), 'qp_failures': int(fl)}
                del sim_a
            except Exception as e:
                line['agent_sharded'] = {'error': repr(e)}
        # -------------------------------------------------------------- weak scaling: every rank runs the whole 4096-instance batch
        if world > 1:
            try:
                sim_w = make()
                sim_w.run(args.burn_in + args.warmup)
                el, it, fl, ms, _ = timed(sim_w, args.steps)
                line['weak'] = {'value': world * args.batch * args.steps / el, 'unit': 'MPC timesteps/s', 'instances_per_gpu': args.batch,
                                'ms_per_step': 1e3 * el / args.steps, 'note': 'every rank owns a full %d-instance batch' % args.batch}
                del sim_w
            except Exception as e:
                line['weak'] = {'error': repr(e)}
        if watchdog is not None:
            watchdog.cancel()

    if rank == 0:
        if snap is not None:
            try:
                line['cpu_baseline'] = cpu_baseline(sim, snap, args.cpu_agents)
            except Exception as e:
                line['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(line))
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
