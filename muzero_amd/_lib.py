"""ctypes binding of libmzhip.so (include/mzhip.h).  There is NO CPU fallback: a missing library or a
missing GPU raises, loudly."""
import ctypes
import os

from .exceptions import MuZeroError, MuZeroHipError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MZ_LIB') or os.path.join(_HERE, 'libmzhip.so')     # MZ_LIB: experiment builds only

c_void_p, c_int32, c_int64, c_double = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double


class PoolConfig(ctypes.Structure):
    """mz_pool_config_t"""
    _fields_ = [('num_trees', c_int32), ('num_actions', c_int32), ('max_simulations', c_int32),
                ('num_players', c_int32), ('has_known_bounds', c_int32), ('lanes_per_tree', c_int32),
                ('discount', c_double), ('default_value', c_double), ('known_min', c_double),
                ('known_max', c_double), ('log_table', c_void_p)]


class MoveIO(ctypes.Structure):
    """mz_move_io_t"""
    _fields_ = [('num_trees', c_int32), ('num_simulations', c_int32), ('hidden_size', c_int32), ('reserved', c_int32),
                ('stream', c_void_p), ('observation', c_void_p), ('legal', c_void_p), ('n_legal', c_void_p),
                ('noise', c_void_p), ('noise_frac', c_double), ('u', c_void_p), ('temperature', c_void_p),
                ('hidden', c_void_p), ('logits', c_void_p), ('reward', c_void_p), ('value', c_void_p),
                ('parent_block', c_void_p), ('action', c_void_p), ('out_visits', c_void_p),
                ('out_root_value', c_void_p), ('out_minmax', c_void_p), ('out_policy', c_void_p),
                ('out_action', c_void_p), ('events', c_void_p),
                ('trace_reward', c_void_p), ('trace_value', c_void_p), ('trace_logits', c_void_p),
                ('trace_parent', c_void_p), ('trace_action', c_void_p)]


class History(ctypes.Structure):
    """mz_history_t"""
    _fields_ = [('num_games', c_int32), ('max_moves', c_int32), ('observation_size', c_int32), ('num_actions', c_int32),
                ('observations', c_void_p), ('to_plays', c_void_p), ('actions', c_void_p), ('rewards', c_void_p),
                ('root_values', c_void_p), ('policies', c_void_p), ('lengths', c_void_p)]


class Env(ctypes.Structure):
    """mz_env_t"""
    _fields_ = [('kind', c_int32), ('num_games', c_int32), ('state', c_void_p), ('to_play', c_void_p),
                ('steps', c_void_p), ('ended', c_void_p), ('restart_state', c_void_p)]


MZ_ENV_TICTACTOE, MZ_ENV_CARTPOLE = 1, 2


class ReplayConfig(ctypes.Structure):
    """mz_replay_config_t"""
    _fields_ = [('num_games', c_int32), ('max_moves', c_int32), ('num_actions', c_int32), ('num_unroll_steps', c_int32),
                ('td_steps', c_int32), ('value_support', c_int32), ('reward_support', c_int32), ('reserved', c_int32),
                ('value_min', c_double), ('value_max', c_double), ('reward_min', c_double), ('reward_max', c_double)]


# name -> (restype, argtypes); every symbol include/mzhip.h declares
PROTOTYPES = {
    'mz_last_error': (ctypes.c_char_p, []),
    'mz_version': (ctypes.c_int, []),
    'mz_device_count': (ctypes.c_int, []),
    'mz_pool_create': (ctypes.c_int, [ctypes.POINTER(PoolConfig), c_void_p, ctypes.POINTER(c_void_p)]),
    'mz_pool_destroy': (ctypes.c_int, [c_void_p]),
    'mz_pool_set_stream': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_pool_reset': (ctypes.c_int, [c_void_p]),
    'mz_expand_roots': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_expand_roots_logits': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_add_root_noise': (ctypes.c_int, [c_void_p, c_void_p, c_double]),
    'mz_select': (ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    'mz_pool_set_selection': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    'mz_expand_backprop': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_expand_backprop_logits': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_step': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_step_logits': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_root_stats': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_select_action': (ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_pool_set_temperature_table': (ctypes.c_int, [c_void_p, c_double, c_void_p]),
    'mz_pool_counters': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_pool_read_block': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    'mz_pool_read_root_priors': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    'mz_pool_num_blocks': (ctypes.c_int, [c_void_p]),
    'mz_pool_bytes': (c_int64, [c_void_p]),
    'mz_net_num_weights': (c_int64, [c_int32]),
    'mz_net_hidden_size': (c_int32, [c_int32]),
    'mz_net_num_actions': (c_int32, [c_int32]),
    'mz_net_observation_size': (c_int32, [c_int32]),
    'mz_net_create': (ctypes.c_int, [c_int32, c_void_p, c_int64, c_int64, c_void_p, ctypes.POINTER(c_void_p)]),
    'mz_net_destroy': (ctypes.c_int, [c_void_p]),
    'mz_net_set_stream': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_net_training_steps': (c_int64, [c_void_p]),
    'mz_net_status': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_net_set_mid_event': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_debug_set_run_mode': (ctypes.c_int, [c_void_p, c_int32]),
    'mz_debug_run_info': (ctypes.c_int, [c_void_p, c_void_p]),
    'mz_debug_set_job_rows': (ctypes.c_int, [c_void_p, c_int32]),
    'mz_infer_initial': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_infer_recurrent_step': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_infer_recurrent_run': (ctypes.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'mz_search_move': (ctypes.c_int, [c_void_p, c_void_p, ctypes.POINTER(MoveIO)]),
    'mz_search_move_injected': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
    'mz_infer_recurrent': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_infer_recurrent_noalloc': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_env_ttt_reset': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_env_ttt_observe': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    'mz_env_ttt_step': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    'mz_env_cartpole_reset': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_env_cartpole_observe': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_env_ttt_reset_where': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_env_cartpole_reset_where': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_history_record_observation': (ctypes.c_int, [c_void_p, ctypes.POINTER(History), c_void_p, c_void_p, c_void_p]),
    'mz_history_record_move': (ctypes.c_int, [c_void_p, ctypes.POINTER(History)] + [c_void_p] * 6),
    'mz_dirichlet_from_gamma': (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'mz_history_flush': (ctypes.c_int, [c_void_p, ctypes.POINTER(History), ctypes.POINTER(History)] + [c_void_p] * 6),
    'mz_selfplay_turnover': (ctypes.c_int, [c_void_p, ctypes.POINTER(Env), ctypes.POINTER(History), ctypes.POINTER(History),
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_double,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mz_selfplay_step_record': (ctypes.c_int, [c_void_p, ctypes.POINTER(Env), ctypes.POINTER(History)] + [c_void_p] * 6),
    'mz_replay_targets': (ctypes.c_int, [c_void_p, ctypes.POINTER(ReplayConfig)] + [c_void_p] * 12),
    'mz_env_cartpole_step': (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load libmzhip.so and bind every prototype; raises MuZeroError if the build is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MuZeroError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'(hipcc --offload-arch=gfx950). muzero_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status: int) -> None:
    if status < 0:
        raise MuZeroHipError(status, load().mz_last_error().decode('utf-8', 'replace'))
