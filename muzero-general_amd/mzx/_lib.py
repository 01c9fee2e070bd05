"""
ctypes binding of the C ABI declared in ``include/mzx.h`` (libmzx.so).

This is the only place the package touches native code.  The product backend
is the gfx950 HIP library next to this file; it is the ONLY backend the package
ever constructs: if ``libmzx.so`` is missing, or no GPU is visible when a compute
entry point is reached, an exception is raised -- there is no CPU fallback.
(The CPU test-suite builds a serial test double of the same ABI under
``tests/hostcheck`` and injects it explicitly; nothing in here knows about it.)
"""
import ctypes
import os

import torch

MZX_MAX_LAYERS = 8
ABI_VERSION = 1
MOVE_NO_SYNC = 1

c_i32, c_i64, c_f64, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p


class NetConfig(ctypes.Structure):
    _L = c_i32 * MZX_MAX_LAYERS
    _fields_ = [
        ("network", c_i32), ("observation_shape", c_i32 * 3), ("stacked_observations", c_i32),
        ("action_space_size", c_i32), ("support_size", c_i32), ("encoding_size", c_i32),
        ("n_fc_representation_layers", c_i32), ("fc_representation_layers", _L),
        ("n_fc_dynamics_layers", c_i32), ("fc_dynamics_layers", _L),
        ("n_fc_reward_layers", c_i32), ("fc_reward_layers", _L),
        ("n_fc_value_layers", c_i32), ("fc_value_layers", _L),
        ("n_fc_policy_layers", c_i32), ("fc_policy_layers", _L),
        ("downsample", c_i32), ("blocks", c_i32), ("channels", c_i32),
        ("reduced_channels_reward", c_i32), ("reduced_channels_value", c_i32), ("reduced_channels_policy", c_i32),
        ("n_resnet_fc_reward_layers", c_i32), ("resnet_fc_reward_layers", _L),
        ("n_resnet_fc_value_layers", c_i32), ("resnet_fc_value_layers", _L),
        ("n_resnet_fc_policy_layers", c_i32), ("resnet_fc_policy_layers", _L),
    ]


class SearchConfig(ctypes.Structure):
    _fields_ = [
        ("num_trees", c_i32), ("num_simulations", c_i32), ("action_space_size", c_i32), ("num_players", c_i32),
        ("support_size", c_i32), ("tape_words", c_i32), ("discount", c_f64), ("root_exploration_fraction", c_f64),
        ("h_pb_c_table", ctypes.POINTER(c_f64)), ("h_sqrt_table", ctypes.POINTER(c_f64)),
    ]


class SearchIO(ctypes.Structure):
    _fields_ = [
        ("d_observation", c_vp), ("d_legal_actions", c_vp), ("d_to_play", c_vp), ("d_noise", c_vp), ("d_tape", c_vp),
        ("d_visit_counts", c_vp), ("d_root_value", c_vp), ("d_root_predicted_value", c_vp), ("d_info", c_vp),
    ]


class Move(ctypes.Structure):
    """``mzx_move`` (include/mzx.h): one self-play move of a shard behind mzx_selfplay_search / mzx_selfplay_select."""
    _fields_ = [
        ("num_games", c_i32), ("action_space_size", c_i32), ("tape_words", c_i32), ("num_threads", c_i32),
        ("streams", c_vp), ("legal_actions", c_vp), ("to_play", c_vp), ("observation", c_vp),
        ("observation_floats", c_i64), ("dirichlet_alpha", c_f64), ("add_exploration_noise", c_i32), ("flags", c_i32),
        ("h_in", c_vp), ("d_in", c_vp), ("in_bytes", c_i64), ("h_out", c_vp), ("d_out", c_vp), ("out_bytes", c_i64),
        ("io", SearchIO),
    ]


class ActorConfig(ctypes.Structure):
    """``mzx_actor_config``: one slot group of a shard of natively stepped games (mzx_selfplay_rounds)."""
    _fields_ = [
        ("game", c_vp), ("search", c_vp), ("bank", c_vp), ("streams", c_vp), ("first_slot", c_i32), ("max_moves", c_i32),
        ("temperature", c_f64), ("d_arena", c_vp), ("arena_bytes", c_i64), ("move", Move),
    ]


RETRY_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_vp, c_i32, c_i32, ctypes.POINTER(c_i32))


class Rounds(ctypes.Structure):
    """``mzx_rounds``: arguments and results of one mzx_selfplay_rounds call."""
    _fields_ = [
        ("temperature", c_f64), ("temperature_threshold", c_i32), ("table_stride", c_i32), ("pow_table", c_vp),
        ("table_temperatures", c_vp), ("num_temperatures", c_i32), ("reserved", c_i32), ("min_games", c_i64),
        ("max_rounds", c_i64), ("sequence", c_i64), ("retry", RETRY_FN), ("retry_ctx", c_vp), ("rounds", c_i64),
        ("games", c_i64), ("searches", c_i64), ("search_seconds", c_f64), ("phase_seconds", c_f64 * 6),
    ]


class TreeDump(ctypes.Structure):
    _fields_ = [
        ("d_visit", c_vp), ("d_value_sum", c_vp), ("d_reward", c_vp), ("d_to_play", c_vp), ("d_parent", c_vp),
        ("d_child", c_vp), ("d_prior", c_vp), ("d_minmax", c_vp), ("d_n_nodes", c_vp),
    ]


class ObsLayout(ctypes.Structure):
    _fields_ = [
        ("channels", c_i32), ("height", c_i32), ("width", c_i32), ("stacked_observations", c_i32),
        ("action_space_size", c_i32), ("num_games", c_i32), ("ring", c_i32),
    ]


# name -> (restype, argtypes); every symbol include/mzx.h declares
PROTOTYPES = {
    "mzx_abi_version": (ctypes.c_int, []),
    "mzx_last_error": (ctypes.c_char_p, []),
    "mzx_is_device_build": (ctypes.c_int, []),
    "mzx_net_create": (ctypes.c_int, [ctypes.POINTER(NetConfig), ctypes.POINTER(c_vp)]),
    "mzx_net_destroy": (None, [c_vp]),
    "mzx_net_num_tensors": (c_i32, [c_vp]),
    "mzx_net_num_params": (c_i64, [c_vp]),
    "mzx_net_tensor_info": (ctypes.c_int, [c_vp, c_i32, ctypes.c_char_p, c_i32, ctypes.POINTER(c_i64),
                                           ctypes.POINTER(c_i64), ctypes.POINTER(c_i32 * 4)]),
    "mzx_net_hidden_size": (c_i64, [c_vp]),
    "mzx_net_input_size": (c_i64, [c_vp]),
    "mzx_net_derived_floats": (c_i64, [c_vp]),
    "mzx_net_workspace_floats": (c_i64, [c_vp, c_i32]),
    "mzx_net_flops": (c_i64, [c_vp, c_i32]),
    "mzx_net_set_weights": (ctypes.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "mzx_net_fused_supported": (ctypes.c_int, [c_vp]),
    "mzx_net_set_mode": (ctypes.c_int, [c_vp, c_i32]),
    "mzx_net_streamed_supported": (ctypes.c_int, [c_vp]),
    "mzx_net_streamed_plan": (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(c_i32 * 24)]),
    "mzx_net_streamed_shape": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32 * 16)]),
    "mzx_net_streamed_tower": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32 * 16)]),
    "mzx_net_streamed_heads": (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(c_i32 * 16)]),
    "mzx_net_streamed_split": (ctypes.c_int, [c_vp, c_i32, ctypes.POINTER(c_i32 * 2)]),
    "mzx_net_operator_out_floats": (c_i64, [c_vp, c_i32, c_i32]),
    "mzx_net_num_operators": (ctypes.c_int, [c_vp, c_i32]),
    "mzx_net_fused_schedule": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32]),
    "mzx_net_debug_prefix": (ctypes.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_i64,
                                            c_vp, c_i64, c_vp]),
    "mzx_net_initial_inference": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mzx_net_recurrent_inference": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mzx_search_create": (ctypes.c_int, [ctypes.POINTER(SearchConfig), c_vp, ctypes.POINTER(c_vp)]),
    "mzx_search_destroy": (None, [c_vp]),
    "mzx_search_arena_bytes": (c_i64, [c_vp]),
    "mzx_search_run": (ctypes.c_int, [c_vp, ctypes.POINTER(SearchIO), c_vp, c_i64, c_vp]),
    "mzx_search_run_from_roots": (ctypes.c_int, [c_vp, ctypes.POINTER(SearchIO), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mzx_search_fused_supported": (ctypes.c_int, [c_vp]),
    "mzx_search_kernel_name": (ctypes.c_char_p, [c_vp]),
    "mzx_search_set_mode": (ctypes.c_int, [c_vp, c_i32]),
    "mzx_search_route": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32 * 8)]),
    "mzx_net_search_route": (ctypes.c_int, [c_vp, c_i32, c_i32, ctypes.POINTER(c_i32 * 8)]),
    "mzx_tuning_set": (ctypes.c_int, [ctypes.c_char_p, c_i32]),
    "mzx_tuning_get": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "mzx_tuning_name": (ctypes.c_char_p, [c_i32]),
    "mzx_tuning_help": (ctypes.c_char_p, [c_i32]),
    "mzx_search_arena_offsets": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i64 * 8)]),
    "mzx_search_lockstep_begin": (ctypes.c_int, [c_vp, ctypes.POINTER(SearchIO), c_vp, c_vp, c_vp, c_i64, c_vp]),
    "mzx_search_lockstep_select": (ctypes.c_int, [c_vp, ctypes.POINTER(SearchIO), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mzx_search_lockstep_apply": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mzx_search_finish": (ctypes.c_int, [c_vp, ctypes.POINTER(SearchIO), c_vp, c_vp]),
    "mzx_search_dump": (ctypes.c_int, [c_vp, ctypes.POINTER(TreeDump), c_vp, c_vp]),
    "mzx_obs_stacked_floats": (c_i64, [ctypes.POINTER(ObsLayout)]),
    "mzx_obs_stack": (ctypes.c_int, [ctypes.POINTER(ObsLayout), c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mzx_support_to_scalar": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mzx_rng_create": (ctypes.c_int, [c_i32, ctypes.POINTER(c_vp)]),
    "mzx_rng_destroy": (None, [c_vp]),
    "mzx_rng_seed": (ctypes.c_int, [c_vp, c_i32, c_i32, c_vp]),
    "mzx_rng_get_state": (ctypes.c_int, [c_vp, c_i32, c_vp, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                         ctypes.POINTER(c_f64)]),
    "mzx_rng_set_state": (ctypes.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_f64]),
    "mzx_rng_root_draws": (ctypes.c_int, [c_vp, c_vp, c_i32, c_f64, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32]),
    "mzx_rng_advance": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp]),
    "mzx_rng_random_sample": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp]),
    "mzx_rng_randint": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_vp]),
    "mzx_rng_choice_weighted": (ctypes.c_int, [c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    "mzx_selfplay_search": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(Move), c_vp, c_vp, c_i64, c_vp]),
    "mzx_replay_priorities": (ctypes.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_f64, c_vp, c_vp, c_vp, c_vp]),
    "mzx_game_create": (ctypes.c_int, [ctypes.c_char_p, c_i32, c_vp, c_vp, c_i32, c_i32, ctypes.POINTER(c_vp)]),
    "mzx_game_destroy": (None, [c_vp]),
    "mzx_game_info": (ctypes.c_int, [c_vp, ctypes.POINTER(c_i32 * 8)]),
    "mzx_game_reset": (ctypes.c_int, [c_vp, c_vp, c_i32]),
    "mzx_game_observe": (ctypes.c_int, [c_vp, c_vp]),
    "mzx_game_legal_actions": (ctypes.c_int, [c_vp, c_vp]),
    "mzx_game_to_play": (ctypes.c_int, [c_vp, c_vp]),
    "mzx_game_step": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mzx_actor_create": (ctypes.c_int, [ctypes.POINTER(ActorConfig), ctypes.POINTER(c_vp)]),
    "mzx_actor_destroy": (None, [c_vp]),
    "mzx_selfplay_rounds": (ctypes.c_int, [ctypes.POINTER(c_vp), c_i32, ctypes.POINTER(Rounds), c_vp]),
    "mzx_actor_finished": (ctypes.c_int, [c_vp, c_i64, ctypes.POINTER(c_i64 * 3)]),
    "mzx_actor_take": (ctypes.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(c_i32)]),
    "mzx_selfplay_select": (ctypes.c_int, [c_vp, ctypes.POINTER(Move), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32,
                                           c_vp]),
}


class MzxError(RuntimeError):
    pass


class Library:
    """A loaded shared object exporting the mzx C ABI."""

    def __init__(self, path):
        if not os.path.isfile(path):
            raise MzxError(
                f"native library not found: {path}. Build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950); the engine has no non-native path."
            )
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)
        if self.mzx_abi_version() != ABI_VERSION:
            raise MzxError(f"{path}: ABI version {self.mzx_abi_version()} != {ABI_VERSION}")

    def check(self, rc):
        if rc != 0:
            raise MzxError(f"mzx error {rc}: {self.mzx_last_error().decode()}")

    # ---- the tuning table (include/mzx.h "Tuning"): process-wide named integers, no environment variables
    def tuning_get(self, name):
        v = c_i32()
        self.check(self.mzx_tuning_get(name.encode(), ctypes.byref(v), None))
        return int(v.value)

    def tuning_set(self, name, value):
        self.check(self.mzx_tuning_set(name.encode(), int(value)))

    def tuning(self, **values):
        """Context manager: ``with lib.tuning(rt_search=0, row_split_min=0): ...`` sets entries and restores them."""
        return _TuningScope(self, values)

    def tuning_table(self):
        out, i = {}, 0
        while True:
            name = self.mzx_tuning_name(i)
            if name is None:
                return out
            out[name.decode()] = self.tuning_get(name.decode())
            i += 1


class _TuningScope:
    def __init__(self, lib, values):
        self.lib, self.values, self.saved = lib, values, {}

    def __enter__(self):
        for k, v in self.values.items():
            self.saved[k] = self.lib.tuning_get(k)
            self.lib.tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            self.lib.tuning_set(k, v)
        return False


class Backend:
    """Library + the torch device its buffers live on + the stream calls are enqueued on."""

    def __init__(self, lib, device):
        self.lib = lib
        self.device = torch.device(device)

    def stream(self):
        if self.device.type == "cuda":
            return c_vp(torch.cuda.current_stream(self.device).cuda_stream)
        return c_vp(0)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    @staticmethod
    def ptr(t):
        return c_vp(0) if t is None else c_vp(t.data_ptr())


LIB_PATH = os.environ.get("MZX_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmzx.so")   # MZX_LIB: A/B builds
_default = None


def default_backend():
    """The product backend: libmzx.so on the current CUDA(HIP) device.  Raises without a GPU."""
    global _default
    if _default is None:
        lib = Library(LIB_PATH)
        if not lib.mzx_is_device_build():
            raise MzxError(f"{LIB_PATH} is not a device build")
        if not torch.cuda.is_available():
            raise MzxError(
                "mzx needs an AMD GPU (torch.cuda.is_available() is False): the self-play engine "
                "has no CPU execution path."
            )
        _default = Backend(lib, torch.device("cuda", torch.cuda.current_device()))
    return _default
