"""Live-tunable variant of the LTV-MPC (reference: main/lib/mpc_sensitivity.py).

Same controller as lib/mpc.py, except that the cost weights and the acceleration / steering-rate limits are re-read
from `config/mpc_config_sensitivity.json` before EVERY solve (mpc_sensitivity.py:150-163), so a sweep script can edit
the file between (or during) closed-loop runs.  Horizon, goal and iteration constants are read once at import
(:17-41).  The reloaded values reach the device through mpcx_set_mpc_params; to run a whole sweep as ONE batch use
Context.set_instance_tuning / IntersectionBatch(tuning=...) instead (one parameter row per instance).
"""
import json
import os

import numpy as np

from ..runtime import MpcParams
from . import mpc as _base
from .mpc import MPCSolutionNotFoundException, smooth_yaw  # noqa: F401
from .simulation import Simulation

_DEFAULT = dict(_base._CFG)


def _config_path():
    for cand in (os.environ.get('MPCX_MPC_SENSITIVITY_CONFIG'), os.path.join('..', 'config', 'mpc_config_sensitivity.json')):
        if cand and os.path.exists(cand):
            return cand
    return None


def _load():
    cfg = dict(_DEFAULT)
    path = _config_path()
    if path:
        with open(path, 'r') as f:
            cfg.update(json.load(f))
    return cfg


_CFG = _load()
NX = _CFG['NX']
NU = _CFG['NU']
T = _CFG['T']
GOAL_DIS = _CFG['GOAL_DIS']
STOP_SPEED = _CFG['STOP_SPEED']
MAX_TIME = _CFG['MAX_TIME']
MAX_ITER = _CFG['MAX_ITER']
DU_TH = _CFG['DU_TH']
MAX_DSTEER = np.deg2rad(_CFG['MAX_DSTEER'])
MAX_ACCEL = _CFG['MAX_ACCEL']
MAX_DECEL = _CFG['MAX_DECEL']


def params_from_config(cfg, car_dimensions, dt, horizon=None) -> MpcParams:
    """MpcParams for one configuration dict with the keys of mpc_config_sensitivity.json"""
    return MpcParams(T=int(horizon if horizon is not None else T), dt=float(dt), L=float(car_dimensions.distance_back_to_front_wheel),
                     w_perp=float(cfg['w_perp']), w_para=float(cfg['w_para']), R=tuple(map(float, cfg['R'])),
                     Rd=tuple(map(float, cfg['Rd'])), Q_v_yaw=tuple(map(float, cfg['Q_v_yaw'])), Qf_base=tuple(map(float, cfg['Qf'])),
                     max_speed=float(Simulation.MAX_SPEED), min_speed=float(Simulation.MIN_SPEED),
                     max_accel=float(cfg['MAX_ACCEL']), max_decel=float(cfg['MAX_DECEL']), max_steer=float(Simulation.MAX_STEER),
                     max_dsteer=float(np.deg2rad(cfg['MAX_DSTEER'])))


class MPC(_base.MPC):
    def _make_params(self) -> MpcParams:
        return params_from_config(_load(), self.car_dimensions, self.dt, horizon=globals()['T'])

    def _fail_decel(self, p: MpcParams) -> float:
        return globals()['MAX_DECEL']           # the import-time constant, not the reloaded one (mpc_sensitivity.py:313)

    def _max_iter(self) -> int:
        return int(globals()['MAX_ITER'])

    def is_goal(self, state) -> bool:
        d = float(np.hypot(state.x - self.goal[0], state.y - self.goal[1]))
        near = d <= GOAL_DIS and abs(self.target_ind - len(self.cx)) < 5
        return bool(near and abs(state.v) <= STOP_SPEED)
