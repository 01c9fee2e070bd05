"""
The network / search shapes of every game file the reference ships (games/*.py), for tests that must run where the
reference tree is absent (the GPU box): mzx.configs carries seven of them; the other four are restated here from
games/gridworld.py:22-63, games/simple_grid.py:22-62, games/twentyone.py:22-70 and games/spiel.py:33-82 (with its default
open_spiel game, tic_tac_toe).  tests/test_configs_reference.py checks the restated attributes against the live files.
"""
from mzx import configs


def shapes():
    return {
        "cartpole": configs.cartpole(), "lunarlander": configs.lunarlander(), "tictactoe": configs.tictactoe(),
        "connect4": configs.connect4(), "breakout": configs.breakout(), "gomoku": configs.gomoku(), "atari": configs.atari(),
        "gridworld": configs.cartpole(observation_shape=(7, 7, 3), action_space=list(range(3)), max_moves=15, discount=0.997,
                                      num_simulations=20),
        "simple_grid": configs.cartpole(observation_shape=(1, 1, 9), action_space=list(range(2)), encoding_size=5,
                                        fc_representation_layers=[16], discount=0.978, max_moves=6, num_simulations=10),
        "twentyone": configs.tictactoe(action_space=list(range(2)), players=list(range(1)), blocks=2, channels=32,
                                       reduced_channels_reward=32, reduced_channels_value=32, reduced_channels_policy=32,
                                       resnet_fc_reward_layers=[16], resnet_fc_value_layers=[16], resnet_fc_policy_layers=[16],
                                       root_dirichlet_alpha=0.25, max_moves=21, num_simulations=21,
                                       fc_representation_layers=[16], fc_dynamics_layers=[16], fc_reward_layers=[16],
                                       fc_value_layers=[16], fc_policy_layers=[16]),
        "spiel": configs.tictactoe(blocks=2, discount=0.1, fc_dynamics_layers=[16], fc_reward_layers=[16]),
    }


NETWORK_ATTRS = ["observation_shape", "action_space", "players", "stacked_observations", "num_simulations", "discount",
                 "root_dirichlet_alpha", "network", "support_size", "downsample", "blocks", "channels",
                 "reduced_channels_reward", "reduced_channels_value", "reduced_channels_policy", "resnet_fc_reward_layers",
                 "resnet_fc_value_layers", "resnet_fc_policy_layers", "encoding_size", "fc_representation_layers",
                 "fc_dynamics_layers", "fc_reward_layers", "fc_value_layers", "fc_policy_layers", "max_moves"]
