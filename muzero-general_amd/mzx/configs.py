"""
Hot-path hyper-parameters of the BASELINE.json configurations.

The engine itself takes ANY object with the reference's ``MuZeroConfig``
attributes (games/*.py) -- an unmodified reference game file drops in.  The
reference tree does not travel to the GPU box, so the bench and the GPU parity
tests need the named configurations without it: this module restates only the
attributes the self-play path reads (SURVEY.md section 5 "Config / flags"), with the
values of the reference game files.  ``tests/test_configs_reference.py`` checks
them against the real files when /root/reference is present.
"""


class HotPathConfig:
    """Attribute bag with the fields self_play.py / models.py read."""

    def __init__(self, **kw):
        # defaults shared by every reference game file
        self.seed = 0
        self.stacked_observations = 0
        self.muzero_player = 0
        self.opponent = None
        self.num_workers = 1
        self.selfplay_on_gpu = True
        self.temperature_threshold = None
        self.root_exploration_fraction = 0.25
        self.pb_c_base = 19652
        self.pb_c_init = 1.25
        self.support_size = 10
        self.downsample = False
        self.blocks = 1
        self.channels = 2
        self.reduced_channels_reward = 2
        self.reduced_channels_value = 2
        self.reduced_channels_policy = 2
        self.resnet_fc_reward_layers = []
        self.resnet_fc_value_layers = []
        self.resnet_fc_policy_layers = []
        self.encoding_size = 8
        self.fc_representation_layers = []
        self.fc_dynamics_layers = [16]
        self.fc_reward_layers = [16]
        self.fc_value_layers = [16]
        self.fc_policy_layers = [16]
        self.training_steps = 10000
        self.checkpoint_interval = 10      # trainer.py:87 publishes weights every checkpoint_interval steps (games/*.py: 10 ... 1000)
        self.self_play_delay = 0
        self.ratio = None
        self.use_last_model_value = True   # every BASELINE game file but breakout (games/breakout.py:109)
        self.__dict__.update(kw)

    temperature_schedule = "fractions"

    def visit_softmax_temperature_fn(self, trained_steps):
        if self.temperature_schedule == "constant":  # games/tictactoe.py:112-122, games/connect4.py:112-122
            return 1
        if self.temperature_schedule == "lunarlander":  # games/lunarlander.py:117-125
            return 0.35
        if self.temperature_schedule == "breakout":  # games/breakout.py:117-133
            if trained_steps < 500e3:
                return 1.0
            elif trained_steps < 750e3:
                return 0.5
            return 0.25
        # games/cartpole.py:115-128, games/gomoku.py:115-128
        if trained_steps < 0.5 * self.training_steps:
            return 1.0
        elif trained_steps < 0.75 * self.training_steps:
            return 0.5
        return 0.25


def cartpole(**kw):
    """games/cartpole.py:11-113 -- BASELINE configs C1 / C2 (FullyConnectedNetwork)."""
    base = dict(
        observation_shape=(1, 1, 4), action_space=list(range(2)), players=list(range(1)),
        max_moves=500, num_simulations=50, discount=0.997, root_dirichlet_alpha=0.25,
        network="fullyconnected", encoding_size=8, fc_representation_layers=[],
        fc_dynamics_layers=[16], fc_reward_layers=[16], fc_value_layers=[16], fc_policy_layers=[16],
        training_steps=10000, ratio=1.5,
    )
    base.update(kw)
    return HotPathConfig(**base)


def tictactoe(**kw):
    """games/tictactoe.py:11-110 -- BASELINE config C3 (MuZeroResidualNetwork, 16 ch, 1 block)."""
    base = dict(
        observation_shape=(3, 3, 3), action_space=list(range(9)), players=list(range(2)),
        opponent="expert", max_moves=9, num_simulations=25, discount=1, root_dirichlet_alpha=0.1,
        network="resnet", downsample=False, blocks=1, channels=16,
        reduced_channels_reward=16, reduced_channels_value=16, reduced_channels_policy=16,
        resnet_fc_reward_layers=[8], resnet_fc_value_layers=[8], resnet_fc_policy_layers=[8],
        encoding_size=32, fc_value_layers=[], fc_policy_layers=[], training_steps=1000000,
        temperature_schedule="constant",
    )
    base.update(kw)
    return HotPathConfig(**base)


def connect4(**kw):
    """games/connect4.py:11-110 -- BASELINE config C4 (MuZeroResidualNetwork, 64 ch, 3 blocks)."""
    base = dict(
        observation_shape=(3, 6, 7), action_space=list(range(7)), players=list(range(2)),
        opponent="expert", max_moves=42, num_simulations=200, discount=1, root_dirichlet_alpha=0.3,
        network="resnet", downsample=False, blocks=3, channels=64,
        reduced_channels_reward=2, reduced_channels_value=2, reduced_channels_policy=4,
        resnet_fc_reward_layers=[64], resnet_fc_value_layers=[64], resnet_fc_policy_layers=[64],
        encoding_size=32, fc_dynamics_layers=[64], fc_reward_layers=[64],
        fc_value_layers=[], fc_policy_layers=[], training_steps=100000,
        temperature_schedule="constant",
    )
    base.update(kw)
    return HotPathConfig(**base)


def breakout(**kw):
    """games/breakout.py:17-115 -- BASELINE config C5 (ResNet with the "resnet" down-sampling stem)."""
    base = dict(
        observation_shape=(3, 96, 96), action_space=list(range(4)), players=list(range(1)),
        max_moves=2500, num_simulations=30, discount=0.997, root_dirichlet_alpha=0.25,
        network="resnet", downsample="resnet", blocks=2, channels=16,
        reduced_channels_reward=4, reduced_channels_value=4, reduced_channels_policy=4,
        resnet_fc_reward_layers=[16], resnet_fc_value_layers=[16], resnet_fc_policy_layers=[16],
        encoding_size=10, fc_value_layers=[], fc_policy_layers=[], training_steps=int(1000e3),
        temperature_schedule="breakout", use_last_model_value=False, checkpoint_interval=500,
    )
    base.update(kw)
    return HotPathConfig(**base)


def lunarlander(**kw):
    """
    games/lunarlander.py:17-115 -- the architecture of the shipped results/lunarlander/model.checkpoint
    (SURVEY.md section 8c): fully connected, encoding 10, 64-wide hidden layers, 4 actions.
    """
    base = dict(
        observation_shape=(1, 1, 8), action_space=list(range(4)), players=list(range(1)),
        max_moves=700, num_simulations=50, discount=0.999, root_dirichlet_alpha=0.25,
        network="fullyconnected", blocks=2, channels=16,
        reduced_channels_reward=16, reduced_channels_value=16, reduced_channels_policy=16,
        encoding_size=10, fc_representation_layers=[],
        fc_dynamics_layers=[64], fc_reward_layers=[64], fc_value_layers=[64], fc_policy_layers=[64],
        training_steps=200000, temperature_schedule="lunarlander",
    )
    base.update(kw)
    return HotPathConfig(**base)


def gomoku(**kw):
    """games/gomoku.py:11-112 AS SHIPPED -- 128 channels x 6 blocks on an 11 x 11 board, 121 actions, 400 simulations:
    too wide for the LDS-resident engine, runs on the streamed MFMA engine (csrc/mzx_batched.hip)."""
    base = dict(
        observation_shape=(3, 11, 11), action_space=list(range(11 * 11)), players=list(range(2)),
        opponent="random", max_moves=121, num_simulations=400, discount=1, root_dirichlet_alpha=0.3,
        network="resnet", downsample=False, blocks=6, channels=128,
        reduced_channels_reward=2, reduced_channels_value=2, reduced_channels_policy=4,
        resnet_fc_reward_layers=[64], resnet_fc_value_layers=[64], resnet_fc_policy_layers=[64],
        encoding_size=32, fc_dynamics_layers=[64], fc_reward_layers=[64],
        fc_value_layers=[], fc_policy_layers=[], training_steps=10000,
        use_last_model_value=False, ratio=1, checkpoint_interval=50,   # temperature: the fractions schedule (games/gomoku.py:115-128)
    )
    base.update(kw)
    return HotPathConfig(**base)


def atari(**kw):
    """games/atari.py:17-114 AS SHIPPED -- the paper-scale network: 256 channels x 16 blocks behind the "resnet"
    down-sampling stem, 32 stacked 96 x 96 x 3 frames, 256-channel head convolutions, [256, 256] head MLPs, support
    300 (73.5 M parameters).  Streamed MFMA engine (csrc/mzx_batched.hip)."""
    base = dict(
        observation_shape=(3, 96, 96), action_space=list(range(4)), players=list(range(1)),
        stacked_observations=32, max_moves=27000, num_simulations=50, discount=0.997, root_dirichlet_alpha=0.25,
        network="resnet", support_size=300, downsample="resnet", blocks=16, channels=256,
        reduced_channels_reward=256, reduced_channels_value=256, reduced_channels_policy=256,
        resnet_fc_reward_layers=[256, 256], resnet_fc_value_layers=[256, 256], resnet_fc_policy_layers=[256, 256],
        encoding_size=10, fc_dynamics_layers=[16], fc_reward_layers=[16], fc_value_layers=[], fc_policy_layers=[],
        training_steps=int(1000e3), temperature_schedule="breakout", checkpoint_interval=1000,
    )
    base.update(kw)
    return HotPathConfig(**base)


BY_NAME = {"gomoku": gomoku, "atari": atari, "cartpole": cartpole, "tictactoe": tictactoe, "connect4": connect4, "breakout": breakout,
           "lunarlander": lunarlander}
