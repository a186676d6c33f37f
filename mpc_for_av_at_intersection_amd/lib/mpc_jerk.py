"""Jerk-penalising variant of the LTV-MPC, GPU-backed (reference: main/lib/mpc_jerk.py).

Drop-in for lib/mpc.py (`from lib.mpc_jerk import MPC, MAX_ACCEL`, scenarios/mpc_intersection.py:20): same class, same
`step(state) -> (di, ai)` contract and attributes.  What the reference changes, and where it lands here:

  * a fifth state that integrates the acceleration input and feeds the speed (A[4,4] = 1, A[2,4] = dt, B[4,0] = dt;
    mpc_jerk.py:67, 73, 78), whose initial value is NOT pinned (`x[:4, 0] == x0`, line 193) -- one more unknown of the QP;
  * `jerk_penalty_weight * square(x[4, t+1] - x[4, t])` for t < T-1 (line 190);
  * constants hard-coded in the module instead of read from mpc_config.json (lines 16-39): T = 13, cross-track / along-track
    weights 10 / 1 (lines 167, 171), Rd = diag(0.3, 1), MAX_DECEL = -5.

Reference window, linearisation point (the four-state plant rollout, lines 112-126), bounds and rate constraints are those of
lib/mpc.py, so mpcx_mpc_prepare_batch is shared; the QP is solved by the stage-structured solver's seven-state sweep
(csrc/mpcx_qp_stage.h, Cx::JERK), selected by mpcx_mpc_params.model = MPCX_MODEL_JERK5.  The reference returns rows 0..3 of
x (lines 201-206); so does this class.
"""
import math

import numpy as np

from .. import _lib
from ..runtime import MpcParams
from . import mpc as _base
from .mpc import MPCSolutionNotFoundException, smooth_yaw  # noqa: F401
from .simulation import Simulation

NX = 5
NU = 2
T = 13

R = np.diag([0.01, 0.01])
Rd = np.diag([.3, 1.0])
Q_v_yaw = np.diag([0., 0.5])
Qf = np.diag([1.0, 1.0, 0., 0.5, 0]) * T
GOAL_DIS = 1.5
STOP_SPEED = 0.5 / 3.6
MAX_TIME = 13.0

jerk_penalty_weight = 1

MAX_ITER = 1
DU_TH = 0.1

MAX_DSTEER = np.deg2rad(30.0)
MAX_ACCEL = 2.0
MAX_DECEL = -5

# cross-track / along-track weights: literals inside the reference's cost loop (mpc_jerk.py:167, 171)
W_PERP = 10.0
W_PARA = 1.0


class MPC(_base.MPC):
    def _make_params(self) -> MpcParams:
        g = globals()
        Tn = int(g['T'])
        qf = np.diag(np.asarray(g['Qf'], dtype=float))
        if qf[4] != 0.0:
            raise NotImplementedError('Qf[4] = %r: a terminal weight on the acceleration state is not implemented '
                                      '(the reference has 0, mpc_jerk.py:24)' % (qf[4],))
        return MpcParams(T=Tn, dt=float(self.dt), L=float(self.car_dimensions.distance_back_to_front_wheel),
                         w_perp=float(g['W_PERP']), w_para=float(g['W_PARA']), R=tuple(np.diag(g['R'])), Rd=tuple(np.diag(g['Rd'])),
                         Q_v_yaw=tuple(np.diag(g['Q_v_yaw'])), Qf_base=tuple(qf[:4] / Tn),
                         max_speed=float(Simulation.MAX_SPEED), min_speed=float(Simulation.MIN_SPEED),
                         max_accel=float(g['MAX_ACCEL']), max_decel=float(g['MAX_DECEL']), max_steer=float(Simulation.MAX_STEER),
                         max_dsteer=float(g['MAX_DSTEER']), model=_lib.MODEL_JERK5, jerk_weight=float(g['jerk_penalty_weight']))

    def _fail_decel(self, p: MpcParams) -> float:
        return globals()['MAX_DECEL']           # mpc_jerk.py:296

    def _max_iter(self) -> int:
        return int(globals()['MAX_ITER'])

    def is_goal(self, state) -> bool:
        d = math.hypot(state.x - self.goal[0], state.y - self.goal[1])
        near = d <= GOAL_DIS
        if abs(self.target_ind - len(self.cx)) >= 5:
            near = False
        return bool(near and abs(state.v) <= STOP_SPEED)
