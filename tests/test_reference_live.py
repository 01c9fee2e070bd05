"""
Build-container-only checks against the LIVE reference (/root/reference): the
restated hot-path configs and the plugin-game test doubles really are what the
reference ships.  Skipped on the GPU box (the reference does not travel).
"""
import numpy
import pytest

from mzx import configs
from oracle import ref_shim

import games_fixture

pytestmark = pytest.mark.reference

FIELDS = ["observation_shape", "action_space", "players", "stacked_observations", "max_moves", "num_simulations",
          "discount", "temperature_threshold", "root_dirichlet_alpha", "root_exploration_fraction", "pb_c_base",
          "pb_c_init", "network", "support_size", "downsample", "blocks", "channels", "reduced_channels_reward",
          "reduced_channels_value", "reduced_channels_policy", "resnet_fc_reward_layers", "resnet_fc_value_layers",
          "resnet_fc_policy_layers", "encoding_size", "fc_representation_layers", "fc_dynamics_layers",
          "fc_reward_layers", "fc_value_layers", "fc_policy_layers", "training_steps", "muzero_player", "opponent"]


@pytest.mark.parametrize("game", ["cartpole", "tictactoe", "connect4", "breakout"])
def test_hot_path_configs_equal_reference(game):
    ref = ref_shim.game_module(game).MuZeroConfig()
    mine = configs.BY_NAME[game]()
    for f in FIELDS:
        assert getattr(mine, f) == getattr(ref, f), (game, f)
    for steps in (0, 0.6 * ref.training_steps, 0.9 * ref.training_steps):
        assert mine.visit_softmax_temperature_fn(steps) == ref.visit_softmax_temperature_fn(steps)


@pytest.mark.parametrize("game", ["tictactoe", "connect4"])
def test_game_doubles_equal_reference(game):
    Ref = ref_shim.game_module(game).Game
    Mine = games_fixture.GAMES[game]
    rs = numpy.random.RandomState(0)
    for episode in range(40):
        a, b = Ref(episode), Mine(episode)
        oa, ob = a.reset(), b.reset()
        done = False
        while not done:
            assert numpy.array(oa).dtype == numpy.array(ob).dtype
            assert numpy.array_equal(oa, ob)
            assert a.legal_actions() == b.legal_actions() and a.to_play() == b.to_play()
            act = int(rs.choice(a.legal_actions()))
            (oa, ra, done), (ob, rb, db) = a.step(act), b.step(act)
            assert ra == rb and done == db
        assert numpy.array_equal(oa, ob)
