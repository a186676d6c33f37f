#!/usr/bin/env python3
"""The reference's stock scenario (main/scenarios/mpc_intersection.py) end to end on this package, without plotting:
plan the route with the GPU-backed motion-primitive A*, drive it with the LTV-MPC while two scripted cars cross the
intersection, cut the reference path in front of predicted conflicts -- every call below has the reference's name and
signature (lib.* shadows main/lib/*), every computation runs in libmpcx.so on the MI355X.

    python examples/stock_intersection.py [--horizon 13] [--start-pos 4] [--turn 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--horizon', type=int, default=13)
    ap.add_argument('--start-pos', type=int, default=4)
    ap.add_argument('--turn', type=int, default=1)
    ap.add_argument('--max-steps', type=int, default=400)
    args = ap.parse_args()

    import mpc_for_av_at_intersection_amd.lib.mpc as mpc_mod
    from mpc_for_av_at_intersection_amd.lib.car_dimensions import BicycleModelDimensions
    from mpc_for_av_at_intersection_amd.lib.collision_avoidance import check_collision_moving_cars, get_cutoff_curve_by_position_idx
    from mpc_for_av_at_intersection_amd.lib.motion_primitive import load_motion_primitives
    from mpc_for_av_at_intersection_amd.lib.motion_primitive_search_modified import MotionPrimitiveSearch
    from mpc_for_av_at_intersection_amd.lib.moving_obstacles import MovingObstacleTIntersection
    from mpc_for_av_at_intersection_amd.lib.moving_obstacles_prediction import MovingObstaclesPrediction
    from mpc_for_av_at_intersection_amd.lib.scenario import intersection
    from mpc_for_av_at_intersection_amd.lib.simulation import HistorySimulation, Simulation, State
    from mpc_for_av_at_intersection_amd.lib.trajectories import calc_nearest_index_in_direction, resample_curve

    mpc_mod.T = args.horizon
    mpc_mod.Qf = np.diag([1.0, 1.0, 0.0, 0.5]) * args.horizon
    DT, TIME_HORIZON, FRAME_WINDOW = 0.2, 7., 20
    cd = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    mps = load_motion_primitives(version='bicycle_model')
    scenario = intersection(start_pos=args.start_pos, turn_indicator=args.turn)
    traffic = [MovingObstacleTIntersection(cd, direction=1, offset=2., turning=False, speed=25 / 3.6, dt=DT),
               MovingObstacleTIntersection(cd, direction=-1, offset=4., turning=True, speed=25 / 3.6, dt=DT)]

    t0 = time.perf_counter()
    search = MotionPrimitiveSearch(scenario, cd, mps, margin=cd.radius)
    cost, path, full = search.run(debug=True)
    print('planned: cost %.2f, %d primitives, %d expansions, %d launches, %.1f ms' % (
        cost, len(path) - 1, len(search.debug_data), search.kernel_launches, 1e3 * (time.perf_counter() - t0)))

    dl = np.linalg.norm(full[0, :2] - full[1, :2])
    mpc = mpc_mod.MPC(cx=full[:, 0], cy=full[:, 1], cyaw=full[:, 2], dl=dl, dt=DT, car_dimensions=cd)
    state = State(x=full[0, 0], y=full[0, 1], yaw=full[0, 2], v=0.0)
    sim = HistorySimulation(car_dimensions=cd, sample_time=DT, initial_state=state)
    margin = 4 * int(np.ceil(cd.radius / dl))
    tidx, tmp, cuts = 0, None, 0
    t0 = time.perf_counter()
    for step in range(args.max_steps):
        if mpc.is_goal(state):
            break
        for o in traffic:
            o.step()
        if tmp is None or np.any(tmp[tidx, :] != tmp[-1, :]):
            tidx = calc_nearest_index_in_direction(state, full[:, 0], full[:, 1], start_index=tidx, forward=True)
        traj = full[tidx:]
        if state.v < Simulation.MAX_SPEED:
            rdl = DT * np.minimum(np.cumsum(np.zeros(traj.shape[0]) + mpc_mod.MAX_ACCEL) + state.v, Simulation.MAX_SPEED)
            ego = resample_curve(traj, dl=rdl)
        else:
            ego = resample_curve(traj, dl=DT * Simulation.MAX_SPEED)
        others = [np.vstack(MovingObstaclesPrediction(*o.get(), sample_time=DT, car_dimensions=cd).state_prediction(TIME_HORIZON)).T
                  for o in traffic]
        hit = check_collision_moving_cars(cd, ego, traj, others, frame_window=FRAME_WINDOW)
        if hit is not None:
            cut = max(tidx + 1, get_cutoff_curve_by_position_idx(full, hit[0], hit[1]) - margin)
            tmp = full[:cut]
            cuts += 1
        else:
            tmp = full
        mpc.set_trajectory_fromarray(tmp)
        delta, acc = mpc.step(state)
        state = sim.step(a=acc, delta=delta, xref_deviation=mpc.get_current_xref_deviation())
    wall = time.perf_counter() - t0
    h = sim.history
    print('%s after %d steps (%.1f s simulated, %.0f ms wall, %.2f ms per step): %d steps with a cut path, top speed %.2f m/s, '
          'final speed %.3f m/s, max cross-track deviation %.3f m' % (
              'goal reached' if mpc.is_goal(state) else 'NOT at goal', step, step * DT, 1e3 * wall, 1e3 * wall / max(step, 1), cuts,
              max(h.v), state.v, np.nanmax(h.xref_deviation)))


if __name__ == '__main__':
    main()
