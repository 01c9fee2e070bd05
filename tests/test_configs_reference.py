"""
mzx.configs restates, for the bench and the GPU box (where the reference tree is absent), the attributes
the self-play path reads from the reference's game files.  This test (build container only) checks every
restated attribute against the real ``games/<name>.py`` MuZeroConfig, including the temperature schedule.
"""
import pytest

from mzx import configs
from oracle import ref_shim

ATTRS = ["observation_shape", "action_space", "players", "stacked_observations", "num_simulations", "discount",
         "root_dirichlet_alpha", "root_exploration_fraction", "pb_c_base", "pb_c_init", "network", "support_size",
         "downsample", "blocks", "channels", "reduced_channels_reward", "reduced_channels_value",
         "reduced_channels_policy", "resnet_fc_reward_layers", "resnet_fc_value_layers", "resnet_fc_policy_layers",
         "encoding_size", "fc_representation_layers", "fc_dynamics_layers", "fc_reward_layers", "fc_value_layers",
         "fc_policy_layers", "max_moves", "temperature_threshold", "muzero_player", "opponent", "use_last_model_value",
         "checkpoint_interval"]


@pytest.mark.reference
@pytest.mark.parametrize("name", ["cartpole", "tictactoe", "connect4", "breakout", "lunarlander", "gomoku", "atari"])
def test_restated_configs_equal_the_reference_game_files(name):
    ref = ref_shim.muzero_config(name)
    ours = configs.BY_NAME[name]()
    for a in ATTRS:
        want, got = getattr(ref, a), getattr(ours, a)
        if isinstance(want, (list, tuple)):
            assert list(got) == list(want), (name, a)
        else:
            assert got == want, (name, a, got, want)
    for steps in (0, 1, 4999, 5000, 7499, 7500, 10 ** 5, 499999, 500000, 749999, 750000, 10 ** 6):
        ours.training_steps = ref.training_steps
        assert ours.visit_softmax_temperature_fn(trained_steps=steps) == ref.visit_softmax_temperature_fn(trained_steps=steps), (name, steps)


@pytest.mark.reference
@pytest.mark.parametrize("name", ["gridworld", "simple_grid", "twentyone"])
def test_restated_shapes_of_the_other_shipped_games(name):
    """tests/shipped_shapes.py (used by the GPU routing test) against the live game files."""
    import shipped_shapes
    ref = ref_shim.muzero_config(name)
    ours = shipped_shapes.shapes()[name]
    for a in shipped_shapes.NETWORK_ATTRS:
        want, got = getattr(ref, a), getattr(ours, a)
        if isinstance(want, (list, tuple)):
            assert list(got) == list(want), (name, a)
        else:
            assert got == want, (name, a, got, want)
