"""ctypes binding of libmpcx.so (the HIP/gfx950 hot path). There is NO CPU fallback: if the library is
missing or no GPU is usable the loaders below raise."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPCX_LIB: developer override (instrumented builds of the same sources, scripts/*_phase_profile.py); the product is libmpcx.so
LIB_PATH = os.environ.get('MPCX_LIB') or os.path.join(_HERE, 'libmpcx.so')
_lib = None

c_dp = C.c_void_p  # device pointers travel as raw addresses


MODEL_BICYCLE4, MODEL_JERK5 = 0, 1      # mpcx_mpc_params.model


class MpcParamsC(C.Structure):
    """mirror of mpcx_mpc_params (include/mpcx.h)"""
    _fields_ = [('T', C.c_int32), ('max_iter', C.c_int32), ('dt', C.c_double), ('L', C.c_double),
                ('w_perp', C.c_double), ('w_para', C.c_double), ('R', C.c_double * 2), ('Rd', C.c_double * 2),
                ('Q_v_yaw', C.c_double * 2), ('Qf', C.c_double * 4), ('R_end', C.c_double * 2),
                ('max_speed', C.c_double), ('min_speed', C.c_double), ('max_accel', C.c_double),
                ('max_decel', C.c_double), ('max_steer', C.c_double), ('max_dsteer', C.c_double), ('tol', C.c_double),
                ('model', C.c_int32), ('reserved', C.c_int32), ('jerk_weight', C.c_double)]


class InteractionParamsC(C.Structure):
    """mirror of mpcx_interaction_params (include/mpcx.h)"""
    _fields_ = [('pred_steps', C.c_int32), ('frame_window', C.c_int32),
                ('cutoff_margin', C.c_int32), ('max_path_len', C.c_int32), ('dt', C.c_double), ('L', C.c_double), ('radius', C.c_double),
                ('circle_centers', C.c_double * 4), ('max_accel', C.c_double), ('max_speed', C.c_double),
                ('path_cum', C.c_void_p), ('path_cum_err', C.c_double), ('path_first_within', C.c_void_p),
                ('plan_cnt', C.c_void_p), ('plan_disc', C.c_void_p), ('plan_box', C.c_void_p), ('path_disc', C.c_void_p),
                ('plan_cap', C.c_int32), ('plan_steps', C.c_int32), ('plan_dl', C.c_double), ('plan_radius', C.c_double)]


class ClosedLoopC(C.Structure):
    """mirror of mpcx_closed_loop (include/mpcx.h); every pointer is a device address"""
    _PTRS = ['state', 'applied', 'obs6', 'path_xyyaw', 'path_cs', 'path_v', 'path_off', 'path_len', 'obs_off', 'obs_cnt',
             'obs_skip', 'traj_idx', 'target_ind', 'hit_idx', 'cut_len', 'hit_xy', 'xref', 'xbar', 'reaches_end',
             'x_sol', 'u_sol', 'status', 'iters', 'kkt']
    _fields_ = ([('P', C.c_int32), ('exchange', C.c_int32), ('dl', C.c_double)] + [(n, C.c_void_p) for n in _PTRS] +
                [('n_inst', C.c_int32), ('agents_local', C.c_int32), ('obs_local', C.c_void_p)])


class AstarSearchC(C.Structure):
    """mirror of mpcx_astar_search (include/mpcx.h)"""
    _fields_ = [('start', C.c_double * 3), ('goal_box', C.c_double * 4), ('goal_point', C.c_double * 3), ('allowed_dtheta', C.c_double),
                ('wh', C.c_double * 5), ('wc', C.c_double * 4), ('hp_norm', C.c_void_p),
                ('variant', C.c_int32), ('max_expansions', C.c_int32), ('ov_off', C.c_int32), ('ov_cnt', C.c_int32)]


class AstarBuffersC(C.Structure):
    """mirror of mpcx_astar_buffers (include/mpcx.h); every pointer is a device address"""
    _fields_ = ([(n, C.c_int32) for n in ('heap_cap', 'table_cap', 'log_cap', 'push_cap', 'path_cap')] +
                [(n, C.c_void_p) for n in ('heap', 'table', 'log', 'push_log', 'path', 'cost', 'miss', 'status', 'n_exp', 'n_push', 'path_len', 'path_prim')])


# the same struct as a numpy record (plan_many_device fills thousands of rows without a Python loop per field)
import numpy as _np
ASTAR_SEARCH_DTYPE = _np.dtype([('start', '<f8', 3), ('goal_box', '<f8', 4), ('goal_point', '<f8', 3), ('allowed_dtheta', '<f8'),
                                ('wh', '<f8', 5), ('wc', '<f8', 4), ('hp_norm', '<u8'), ('variant', '<i4'), ('max_expansions', '<i4'),
                                ('ov_off', '<i4'), ('ov_cnt', '<i4')])
assert ASTAR_SEARCH_DTYPE.itemsize == C.sizeof(AstarSearchC)
ASTAR_BASE, ASTAR_MODIFIED, ASTAR_MULTI_LANE, ASTAR_ROUNDABOUT, ASTAR_SINGLE_LANE = 0, 1, 2, 3, 4
ASTAR_VARIANTS = {'base': ASTAR_BASE, 'modified': ASTAR_MODIFIED, 'multi_lane': ASTAR_MULTI_LANE, 'roundabout': ASTAR_ROUNDABOUT,
                  'single_lane': ASTAR_SINGLE_LANE}
ASTAR_FOUND, ASTAR_EXHAUSTED, ASTAR_CAPACITY, ASTAR_MISS, ASTAR_PATH_CAPACITY = 0, 1, 2, 3, 4
COMM_ID_BYTES = 128
SHARD_INSTANCES, SHARD_AGENTS = 1, 2


EXPORTS = ['mpcx_create', 'mpcx_destroy', 'mpcx_last_error', 'mpcx_version', 'mpcx_set_mpc_params',
           'mpcx_qp_solve_batch', 'mpcx_mpc_prepare_batch', 'mpcx_search_model_create', 'mpcx_search_model_destroy',
           'mpcx_expand_batch', 'mpcx_interaction_batch', 'mpcx_moving_collision_batch', 'mpcx_plant_step_batch',
           'mpcx_transform_batch', 'mpcx_cutoff_index_batch', 'mpcx_predict_obstacles_batch', 'mpcx_selftest_wave_ops', 'mpcx_selftest_mfma',
           'mpcx_closed_loop_run', 'mpcx_profile_qp', 'mpcx_profile_qp_read', 'mpcx_set_instance_tuning', 'mpcx_set_qp_solver', 'mpcx_qp_set_order_hint', 'mpcx_expand_multi_batch',
           'mpcx_comm_unique_id', 'mpcx_comm_init', 'mpcx_comm_destroy', 'mpcx_allgather_states', 'mpcx_closed_loop_stats',
           'mpcx_mpc_prepare_batch_ov', 'mpcx_set_linearisation_passes', 'mpcx_astar_batch']


def load():
    """Load libmpcx.so and declare the prototypes of include/mpcx.h. Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libmpcx.so not built (%s): run `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C mpc_for_av_at_intersection_amd/csrc`; there is no CPU fallback' % LIB_PATH)
    # torch first: libmpcx.so must bind to the HIP runtime torch ships (same SONAME), not open a second copy
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    lib.mpcx_create.restype = vp; lib.mpcx_create.argtypes = [i32, vp]
    lib.mpcx_destroy.restype = None; lib.mpcx_destroy.argtypes = [vp]
    lib.mpcx_last_error.restype = C.c_char_p; lib.mpcx_last_error.argtypes = [vp]
    lib.mpcx_version.restype = C.c_char_p; lib.mpcx_version.argtypes = []
    lib.mpcx_set_mpc_params.restype = i32; lib.mpcx_set_mpc_params.argtypes = [vp, C.POINTER(MpcParamsC)]
    lib.mpcx_qp_solve_batch.restype = i32; lib.mpcx_qp_solve_batch.argtypes = [vp, i32] + [vp] * 10
    lib.mpcx_mpc_prepare_batch.restype = i32
    lib.mpcx_mpc_prepare_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp, vp, vp]
    lib.mpcx_mpc_prepare_batch_ov.restype = i32
    lib.mpcx_mpc_prepare_batch_ov.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, C.c_double, vp, vp, C.c_int64, vp, vp, vp]
    lib.mpcx_set_linearisation_passes.restype = i32; lib.mpcx_set_linearisation_passes.argtypes = [vp, i32]
    lib.mpcx_astar_batch.restype = i32
    lib.mpcx_astar_batch.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp]
    lib.mpcx_search_model_create.restype = vp
    lib.mpcx_search_model_create.argtypes = [vp, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.mpcx_search_model_destroy.restype = None; lib.mpcx_search_model_destroy.argtypes = [vp]
    lib.mpcx_expand_batch.restype = i32; lib.mpcx_expand_batch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    lib.mpcx_interaction_batch.restype = i32
    lib.mpcx_interaction_batch.argtypes = [vp, C.POINTER(InteractionParamsC), i32] + [vp] * 6 + [i32] + [vp] * 8
    lib.mpcx_moving_collision_batch.restype = i32
    lib.mpcx_moving_collision_batch.argtypes = [vp, C.POINTER(InteractionParamsC), i32] + [vp] * 8 + [i32] + [vp] * 6
    lib.mpcx_plant_step_batch.restype = i32; lib.mpcx_plant_step_batch.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.mpcx_transform_batch.restype = i32; lib.mpcx_transform_batch.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.mpcx_cutoff_index_batch.restype = i32
    lib.mpcx_cutoff_index_batch.argtypes = [vp, i32, vp, vp, vp, vp, C.c_double, vp]
    lib.mpcx_predict_obstacles_batch.restype = i32
    lib.mpcx_predict_obstacles_batch.argtypes = [vp, i32, i32, C.c_double, C.c_double, vp, vp]
    lib.mpcx_selftest_wave_ops.restype = i32; lib.mpcx_selftest_wave_ops.argtypes = [vp, vp, vp]
    lib.mpcx_selftest_mfma.restype = i32; lib.mpcx_selftest_mfma.argtypes = [vp, vp, vp, vp]
    lib.mpcx_closed_loop_run.restype = i32
    lib.mpcx_closed_loop_run.argtypes = [vp, C.POINTER(InteractionParamsC), C.POINTER(ClosedLoopC), i32, i32]
    lib.mpcx_profile_qp.restype = i32; lib.mpcx_profile_qp.argtypes = [vp, i32]
    lib.mpcx_profile_qp_read.restype = i32
    lib.mpcx_profile_qp_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.mpcx_set_instance_tuning.restype = i32; lib.mpcx_set_instance_tuning.argtypes = [vp, vp, i32]
    lib.mpcx_set_qp_solver.restype = i32; lib.mpcx_set_qp_solver.argtypes = [vp, i32]
    lib.mpcx_qp_set_order_hint.restype = i32; lib.mpcx_qp_set_order_hint.argtypes = [vp, vp, vp, vp]
    lib.mpcx_expand_multi_batch.restype = i32; lib.mpcx_expand_multi_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mpcx_comm_unique_id.restype = i32; lib.mpcx_comm_unique_id.argtypes = [vp]
    lib.mpcx_comm_init.restype = i32; lib.mpcx_comm_init.argtypes = [vp, i32, i32, vp]
    lib.mpcx_comm_destroy.restype = i32; lib.mpcx_comm_destroy.argtypes = [vp]
    lib.mpcx_allgather_states.restype = i32; lib.mpcx_allgather_states.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.mpcx_closed_loop_stats.restype = i32; lib.mpcx_closed_loop_stats.argtypes = [vp, C.POINTER(C.c_int64), i32]
    _lib = lib
    return lib
