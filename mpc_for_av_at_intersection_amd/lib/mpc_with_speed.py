"""Speed-reference variant of the LTV-MPC (reference: main/lib/mpc_with_speed.py): the path carries a speed profile `cv`
that enters the window as xref[2,:] (:103-104) and is tracked with weight 20 (Q_v_yaw = diag(20, 0.5), :23); cross-/along-
track weights are the literals 10 / 1 of :161,165; MAX_DECEL = -5 (:35).  `set_trajectory_fromarray(traj, cutoff_idx)`
rebuilds cv = MAX_SPEED with zeros from cutoff_idx on (:276-282).  Same device path as lib/mpc.py: the window kernel takes
the profile through `path_v`, the QP kernel already handles a speed weight and a speed reference."""
import sys
from typing import Optional, Tuple

import numpy as np

from ..runtime import MpcParams
from ._session import context
from .mpc import smooth_yaw, MPCSolutionNotFoundException  # noqa: F401
from .simulation import Simulation, State

NX = 4
NU = 2
T = 13
R = np.diag([0.01, 0.01])
Rd = np.diag([0.01, 1.0])
Q_v_yaw = np.diag([20, 0.5])
Qf = np.diag([1.0, 1.0, 0., 0.5]) * T
GOAL_DIS = 1.5
STOP_SPEED = 0.5 / 3.6
MAX_TIME = 13.0
MAX_ITER = 1
DU_TH = 0.1
MAX_DSTEER = np.deg2rad(30.0)
MAX_ACCEL = 2.0
MAX_DECEL = -5
MAX_SPEED = 25 / 3.6
W_PERP, W_PARA = 10., 1.0          # literals inside _linear_mpc_control (:161,165)


def _params(car_dimensions, dt) -> MpcParams:
    g = globals()
    Tn = int(g['T'])
    return MpcParams(T=Tn, dt=float(dt), L=float(car_dimensions.distance_back_to_front_wheel), w_perp=W_PERP, w_para=W_PARA,
                     R=tuple(np.diag(g['R'])), Rd=tuple(np.diag(g['Rd'])), Q_v_yaw=tuple(np.diag(g['Q_v_yaw'])),
                     Qf_base=tuple(np.diag(np.asarray(g['Qf'], float)) / Tn), max_speed=float(Simulation.MAX_SPEED),
                     min_speed=float(Simulation.MIN_SPEED), max_accel=float(g['MAX_ACCEL']), max_decel=float(g['MAX_DECEL']),
                     max_steer=float(Simulation.MAX_STEER), max_dsteer=float(g['MAX_DSTEER']))


class MPC:
    def __init__(self, cx, cy, cv, cyaw, dl: float, car_dimensions, dt: float = 0.2, ctx=None):
        self.cx, self.cy, self.cv = cx, cy, cv
        self.cyaw = smooth_yaw(cyaw)
        self.dl, self.dt, self.car_dimensions = dl, dt, car_dimensions
        self.goal: Tuple[float, float] = cx[-1], cy[-1]
        self.target_ind = 0
        self.odelta = self.oa = self.ox = self.oy = self.oyaw = self.ov = self.xref = None
        self.di = 0.0
        self.ai = 0.0
        self.status = 0
        self._ctx = ctx if ctx is not None else context()

    def set_trajectory_fromarray(self, trajectory: np.ndarray, cutoff_idx: int = 999):
        self.cx, self.cy, self.cyaw = trajectory[:, 0], trajectory[:, 1], trajectory[:, 2]
        self.cv = np.full_like(self.cyaw, MAX_SPEED)
        if cutoff_idx != 999:
            self.cv[cutoff_idx:] = 0

    def step(self, state: State) -> Tuple[float, float]:
        ctx = self._ctx
        p = _params(self.car_dimensions, self.dt)
        if ctx.params != p:
            ctx.set_mpc_params(p)
        x0 = ctx.f64([[state.x, state.y, state.v, state.yaw]])
        warm = None
        if self.oa is not None and self.odelta is not None:
            warm = ctx.f64(np.stack([np.asarray(self.oa, float), np.asarray(self.odelta, float)])[None])
        tind = ctx.i32([self.target_ind])
        path = ctx.f64(np.column_stack([self.cx, self.cy, self.cyaw]))
        pv = ctx.f64(self.cv)
        sol = None
        for it in range(max(1, int(globals()['MAX_ITER']))):      # mpc_with_speed.py: the same MAX_ITER loop as lib/mpc.py:226-237
            if it > 0 and int(sol['status'].cpu()[0]) != 0:
                raise TypeError("zip argument #1 must support iteration")      # what the reference's next pass does with oa = None
            pre = ctx.prepare(x0, warm if it == 0 else sol['u'], path, ctx.i32([0]), ctx.i32([len(self.cx)]), float(self.dl), tind, path_v=pv,
                              x_prev=None if it == 0 else sol['x'])
            sol = ctx.qp_solve(x0, pre['xref'], pre['xbar'], pre['reaches_end'], warm if it == 0 else sol['u'].clone())
            ctx.synchronize()
            ti = int(tind.cpu()[0])
            if ti < 0:
                raise Exception("something wrong")
        self.target_ind = ti
        self.xref = pre['xref'].cpu().numpy()[0]
        self.status = int(sol['status'].cpu()[0])
        if self.status == 0:
            x = sol['x'].cpu().numpy()[0]; u = sol['u'].cpu().numpy()[0]
            self.ox, self.oy, self.ov, self.oyaw = x[0].copy(), x[1].copy(), x[2].copy(), x[3].copy()
            self.oa, self.odelta = u[0].copy(), u[1].copy()
            self.di, self.ai = self.odelta[0], self.oa[0]
        else:
            print("Error: Cannot solve mpc...", file=sys.stderr)
            self.oa = self.odelta = self.ox = self.oy = self.oyaw = self.ov = None
            self.ai = MAX_DECEL
        return self.di, self.ai

    def is_goal(self, state: State) -> bool:
        import math
        d = math.hypot(state.x - self.goal[0], state.y - self.goal[1])
        near = d <= GOAL_DIS
        if abs(self.target_ind - len(self.cx)) >= 5:
            near = False
        return bool(near and abs(state.v) <= STOP_SPEED)
