"""Shared test helpers: golden fixtures (tests/golden, generated from the reference by make_golden.py) and
model builders. Nothing here reads /root/reference."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
_cache = {}


def gold(name):
    if name not in _cache:
        path = os.path.join(GOLD, name)
        if name.endswith('.json'):
            with open(path) as f:
                _cache[name] = json.load(f)
        else:
            _cache[name] = np.load(path)
    return _cache[name]


def prim_meta(version='bicycle_model'):
    return gold('primitives_meta.json')[version]


def car(version='bicycle_model'):
    return gold('primitives_meta.json')['cars'][version]


def search_tables(version='bicycle_model', scenario='int_4_1'):
    """(templates, last_pose, edge_cost, hp, hp_off) with primitive ids = sorted names"""
    meta = prim_meta(version)
    prim, tm, sc = gold('primitives.npz'), gold('templates.npz'), gold('scenarios.npz')
    names = meta['names']
    templates = [tm['%s/%s' % (version, n)] for n in names]
    last = np.array([prim['%s/%s' % (version, n)][-1] for n in names])
    tag = 'bic' if version == 'bicycle_model' else 'pri'
    return templates, last, np.array(meta['total_length']), sc[scenario + '/hp_' + tag], sc[scenario + '/hp_off']


def smoothed_path(sp, ti):
    """A* trajectory of intersection(sp, ti) with the yaw column unwrapped as MPC.__init__ does (mpc.py:257)."""
    from oracle import oracle_py as orc
    full = gold('mpc_pre.npz')['path_%d_%d' % (sp, ti)].copy()
    full[:, 2] = orc.smooth_yaw(full[:, 2])
    return full


def ego_resample_dl(n, v0, dt=0.2, max_accel=2.0, max_speed=30.0 / 3.6):
    """scenarios/mpc_intersection.py:110-116 (caller code): the dl passed to resample_curve"""
    if v0 < max_speed:
        return dt * np.minimum(np.cumsum(np.zeros(n) + max_accel) + v0, max_speed)
    return dt * max_speed
