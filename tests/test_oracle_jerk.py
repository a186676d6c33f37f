"""The five-state MPC of main/lib/mpc_jerk.py: oracle (oracle/oracle_jerk.c) against the exact minimiser of the literal problem,
and the host build of the stage-structured solver's seven-state sweep (csrc/mpcx_qp_stage.h, Cx::JERK) against the oracle.

Parity status: cvxpy/ECOS cannot run here and the reference holds no output of mpc_jerk.py, so the pin is the same as for
lib/mpc.py -- tests/qp_literal.py restates the cvxpy problem line by line (objective terms and constraint list of
mpc_jerk.py:143-199) and its exact active-set solution is the point ECOS converges to."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as orc
from tests import helpers as H
from tests import qp_literal as QL

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'stage_ref', 'stage_ref.cpp')
INC = ['-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'mpc_for_av_at_intersection_amd', 'csrc')]


def jerk_cases(T=13, stride=1):
    """closed-loop QP inputs of the golden T = 13 run (state, window, linearisation point, clipped-tail flags, warm start):
    recorded with lib/mpc.py in the loop, used here as inputs of the five-state problem -- same shapes, same value ranges"""
    cl = H.gold('closedloop.npz')
    k = 'T%d/' % T
    out = []
    for i in range(0, len(cl[k + 'x0']), stride):
        warm = np.stack([cl[k + 'oa'][i - 1], cl[k + 'od'][i - 1]]) if (i > 0 and cl[k + 'status'][i - 1] == 0) else None
        out.append((cl[k + 'x0'][i], cl[k + 'xref'][i], cl[k + 'xbar'][i], cl[k + 're'][i], warm))
    return out


def test_params_of_the_jerk_module():
    p = orc.MpcParams.jerk()
    assert (p.T, p.w_perp, p.w_para, p.Rd, p.max_decel, p.model, p.jerk_weight) == (13, 10.0, 1.0, (0.3, 1.0), -5.0, 1, 1.0)
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    import dataclasses
    q = MpcParams.jerk()
    assert all(getattr(q, f.name) == getattr(p, f.name) for f in dataclasses.fields(orc.MpcParams))


def test_literal_jerk_problem_shape():
    """5 (T+1) + 2T variables, 5T + 4 equalities (only x[:4, 0] pinned), the inequality rows of lib/mpc.py"""
    p = orc.MpcParams.jerk()
    x0, xref, xbar, re, _ = jerk_cases()[5]
    P, q, c0, Aeq, beq, G, h, ix, iu = QL.build(p, x0, xref, xbar, re)
    T = p.T
    assert P.shape == (5 * (T + 1) + 2 * T,) * 2 and Aeq.shape[0] == 5 * T + 4 and G.shape[0] == 2 * (T - 1) + 2 * (T + 1) + 4 * T
    assert not Aeq[:, ix(4, 0)][-4:].any()                   # no row pins x[4, 0]
    # the jerk term couples consecutive fifth states with weight jerk_penalty_weight, t = 0 .. T-2
    assert P[ix(4, 1), ix(4, 0)] == -1.0 and P[ix(4, T), ix(4, T - 1)] == 0.0


def test_oracle_jerk_vs_exact_active_set_solution():
    p = orc.MpcParams.jerk()
    dist, x4 = [], []
    for k, (x0, xref, xbar, re, warm) in enumerate(jerk_cases()):
        sol = orc.qp_solve(p, x0, xref, xbar, re, warm)
        assert sol.status == 0 and sol.x.shape == (5, p.T + 1)
        zo = QL.pack(p, sol.x, sol.u)
        ex = QL.exact_solution(p, x0, xref, xbar, re, zo)
        assert ex['eq'] < 1e-9 and ex['stat'] < 1e-7 * max(1.0, np.abs(ex['lam']).max() if len(ex['lam']) else 1.0), (k, ex['eq'], ex['stat'])
        assert (ex['lam'] >= -1e-7).all() and ex['slack'].min() > -1e-9, k
        dist.append(np.abs(zo - ex['z']).max())
        x4.append(sol.x[4, 0])
    dist = np.array(dist)
    print('jerk oracle vs exact: max %.2e median %.2e, x4_0 in [%.3f, %.3f]' % (dist.max(), np.median(dist), min(x4), max(x4)))
    assert dist.max() < 5e-6 and np.median(dist) < 1e-7      # (1e-4 is the stated tolerance against the reference's optimum)
    assert max(np.abs(x4)) > 0.1                             # the free initial acceleration state is really used


def test_jerk_solution_differs_from_four_state_solution():
    """guards against a silent fall-through to the lib/mpc.py problem: same inputs, same weights, the two models disagree"""
    pj = orc.MpcParams.jerk()
    p4 = orc.MpcParams.jerk(model=0)
    x0, xref, xbar, re, warm = jerk_cases()[10]
    a, b = orc.qp_solve(pj, x0, xref, xbar, re, warm), orc.qp_solve(p4, x0, xref, xbar, re, warm)
    assert a.status == b.status == 0 and np.abs(a.u - b.u).max() > 1e-2


def _stage_lib(tmp_path):
    so = str(tmp_path / 'libstage_ref.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unknown-pragmas'] + INC + ['-o', so, SRC], check=True)
    return C.CDLL(so)


def test_stage_solver_jerk_host_build_matches_oracle(tmp_path):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    lib = _stage_lib(tmp_path)
    vp = C.c_void_p
    worst, n_end = 0.0, 0
    for T in (13, 20):                                       # 13: the module's horizon; 20: the benchmark's
        p = orc.MpcParams.jerk(T=T)
        cp = MpcParams.jerk(T=T).to_c()
        for x0, xref, xbar, re, warm in jerk_cases(T, stride=2):
            a = [np.ascontiguousarray(v, np.float64) for v in (x0, xref, xbar)]
            re8 = np.ascontiguousarray(re, np.uint8)
            uw = None if warm is None else np.ascontiguousarray(warm, np.float64)
            x = np.zeros((4, T + 1)); u = np.zeros((2, T)); kkt = np.zeros(4); st = C.c_int32(-1); it = C.c_int32(-1)
            lib.stage_ref_solve(C.byref(cp), *(v.ctypes.data_as(vp) for v in a), re8.ctypes.data_as(vp),
                                None if uw is None else uw.ctypes.data_as(vp), x.ctypes.data_as(vp), u.ctypes.data_as(vp),
                                C.byref(st), C.byref(it), kkt.ctypes.data_as(vp))
            r = orc.qp_solve(p, x0, xref, xbar, re, warm)
            assert st.value == r.status == 0
            assert abs(it.value - r.iters) <= 1
            worst = max(worst, np.abs(u - r.u).max(), np.abs(x - r.x[:4]).max())
            n_end += int(re8.any())
    assert worst < 1e-8, worst
    assert n_end >= 5


def test_stage_solver_jerk_under_sanitizers(tmp_path):
    from mpc_for_av_at_intersection_amd.runtime import MpcParams
    exe = str(tmp_path / 'stage_ref_asan')
    subprocess.run(['g++', '-O1', '-g', '-std=c++17', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-DSTAGE_REF_MAIN',
                    '-Wno-unknown-pragmas'] + INC + ['-o', exe, SRC], check=True)
    T = 13
    sel = jerk_cases(T, stride=9)
    cp = MpcParams.jerk().to_c()
    inp, outp = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(inp, 'wb') as f:
        f.write(bytes(cp)); f.write(struct.pack('i', len(sel)))
        for x0, xref, xbar, re, warm in sel:
            f.write(np.ascontiguousarray(x0, np.float64).tobytes()); f.write(np.ascontiguousarray(xref, np.float64).tobytes())
            f.write(np.ascontiguousarray(xbar, np.float64).tobytes()); f.write(np.ascontiguousarray(re, np.uint8).tobytes())
            f.write(struct.pack('i', 0 if warm is None else 1))
            f.write(np.ascontiguousarray(np.zeros((2, T)) if warm is None else warm, np.float64).tobytes())
    res = subprocess.run([exe, inp, outp], env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'), capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert 'runtime error' not in res.stderr and 'AddressSanitizer' not in res.stderr
    raw = open(outp, 'rb').read()
    rec = 8 + 8 * (2 * T + 4 * (T + 1) + 4)
    assert len(raw) == rec * len(sel)
    p = orc.MpcParams.jerk()
    for i, (x0, xref, xbar, re, warm) in enumerate(sel):
        st, it = struct.unpack_from('ii', raw, i * rec)
        u = np.frombuffer(raw, np.float64, 2 * T, i * rec + 8).reshape(2, T)
        r = orc.qp_solve(p, x0, xref, xbar, re, warm)
        assert st == r.status == 0 and np.abs(u - r.u).max() < 1e-8
