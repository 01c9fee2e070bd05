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


@pytest.mark.parametrize("game", ["cartpole", "tictactoe", "connect4", "breakout", "gomoku", "atari"])
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


# ---- the "next" rows against the live reference classes, through the serial build of the C ABI

def _random_history(ref_self_play, cfg, moves, seed):
    rs = numpy.random.RandomState(seed)
    gh = ref_self_play.GameHistory()
    for t in range(moves + 1):
        gh.observation_history.append(rs.standard_normal(cfg.observation_shape).astype("float32"))
        gh.action_history.append(0 if t == 0 else int(rs.randint(0, len(cfg.action_space))))
        gh.reward_history.append(0.0 if t == 0 else float(rs.randint(0, 2)))
        gh.to_play_history.append(t % len(cfg.players))
    gh.root_values = [float(v) for v in rs.standard_normal(moves)]
    return gh


@pytest.mark.parametrize("game,overrides,moves", [("tictactoe", dict(stacked_observations=3), 7),
                                                  ("cartpole", dict(stacked_observations=2), 23),
                                                  ("connect4", dict(), 11)])
def test_reanalyse_equals_live_reference_worker(game, overrides, moves):
    """mzx.replay.Reanalyse.reanalyse_game against replay_buffer.Reanalyse (unmodified) on a fresh random game."""
    import torch
    import hostcheck
    from mzx import models, replay, synthetic
    from oracle.make_golden import reference_reanalyse
    ref_models, ref_self_play = ref_shim.load()
    cfg = ref_shim.game_module(game).MuZeroConfig()
    for k, v in overrides.items():
        setattr(cfg, k, v)
    gh = _random_history(ref_self_play, cfg, moves, seed=moves)
    torch.manual_seed(0)
    weights = synthetic.fill_state_dict(ref_models.MuZeroNetwork(cfg).state_dict(), 77)
    want = reference_reanalyse(cfg, weights, gh)
    worker = replay.Reanalyse({"weights": weights, "num_reanalysed_games": 0}, cfg, _backend=hostcheck.backend())
    got = worker.reanalyse_game(gh)                       # the reference's own GameHistory object drops in
    assert got.shape == want.shape and got.dtype == want.dtype
    assert numpy.allclose(got, want, atol=3e-4, rtol=3e-4), numpy.abs(got - want).max()


def test_frame_store_equals_live_reference_history():
    """FrameStore (ring of k + 1 device slots) against the reference GameHistory of every game of a shard."""
    import torch
    import hostcheck
    from mzx import observations
    _, ref_self_play = ref_shim.load()
    cfg = configs.HotPathConfig(observation_shape=(2, 5, 4), stacked_observations=3, action_space=list(range(6)),
                                players=[0])
    G, moves = 5, 9
    games = [_random_history(ref_self_play, cfg, moves, seed=100 + g) for g in range(G)]
    store = observations.FrameStore(cfg, G, hostcheck.backend())
    for t in range(moves + 1):
        store.push(numpy.stack([g.observation_history[t] for g in games]),
                   None if t == 0 else [g.action_history[t] for g in games])
        got = store.stacked().cpu().numpy()
        for i, g in enumerate(games):
            part = ref_self_play.GameHistory()
            part.observation_history = g.observation_history[: t + 1]
            part.action_history = g.action_history[: t + 1]
            want = torch.tensor(numpy.array(part.get_stacked_observations(-1, 3, 6))).float().numpy()
            assert numpy.array_equal(got[i], want), (t, i)


def test_cnn_stem_equals_live_reference_module():
    """downsample="CNN" (models.py:278-297) on a fresh geometry: heads of both inferences within 1e-4."""
    import torch
    import hostcheck
    from mzx import models, synthetic
    ref_models, _ = ref_shim.load()
    cfg = ref_shim.game_module("breakout").MuZeroConfig()
    cfg.downsample, cfg.observation_shape, cfg.channels, cfg.blocks = "CNN", (3, 64, 48), 8, 1
    torch.manual_seed(0)
    ref = ref_models.MuZeroNetwork(cfg)
    weights = synthetic.fill_state_dict(ref.state_dict(), 5)
    ref.set_weights(weights)
    ref.eval()
    net = models.MuZeroNetwork(cfg, _backend=hostcheck.backend())
    net.set_weights(weights)
    obs = torch.tensor(synthetic.observations(2, net.input_shape, seed=4))
    act = torch.tensor([[1], [3]])
    with torch.no_grad():
        want_i = ref.initial_inference(obs)
        want_r = ref.recurrent_inference(want_i[3], act)
    got_i = net.initial_inference(obs)
    got_r = net.recurrent_inference(want_i[3], act)
    for want, got in ((want_i, got_i), (want_r, got_r)):
        for k in (0, 2, 3):
            assert numpy.abs(got[k].cpu().numpy() - want[k].numpy()).max() < 1e-4


@pytest.mark.parametrize("game,opponent,muzero_player,sims", [("tictactoe", "random", 0, 12), ("tictactoe", "expert", 1, 12),
                                                              ("connect4", "expert", 0, 8), ("connect4", "random", 1, 8)])
def test_play_game_against_opponents_equals_live_reference(game, opponent, muzero_player, sims):
    """
    test_mode games (self_play.py:139-162, :188-220): MuZero searches its own moves, the opponent's come from the
    plugin's expert_agent / numpy.random.choice -- all draws from the process-global numpy stream, in the
    reference's order.  Same seed => the same GameHistory (moves, rewards, None root values on opponent turns).
    """
    import torch
    import hostcheck
    from mzx import self_play, synthetic
    ref_models, ref_self_play = ref_shim.load()
    Game = ref_shim.game_module(game).Game
    cfg = ref_shim.game_module(game).MuZeroConfig()
    cfg.num_simulations = sims
    torch.manual_seed(0)
    weights = synthetic.fill_state_dict(ref_models.MuZeroNetwork(cfg).state_dict(), 91)
    want = ref_self_play.SelfPlay({"weights": weights}, Game, cfg, 13).play_game(0, None, False, opponent, muzero_player)
    got = self_play.SelfPlay({"weights": weights}, Game, cfg, 13, _backend=hostcheck.backend()).play_game(
        0, None, False, opponent, muzero_player)
    assert [int(a) for a in got.action_history] == [int(a) for a in want.action_history]
    assert got.reward_history == want.reward_history and got.to_play_history == want.to_play_history
    assert len(got.root_values) == len(want.root_values)
    for a, b in zip(got.root_values, want.root_values):
        assert (a is None) == (b is None)
        if a is not None:
            assert abs(a - b) < 3e-4 * max(1.0, abs(b))     # decoded values: ~1e-4 relative (inverse value transform)
    assert any(v is None for v in want.root_values) and any(v is not None for v in want.root_values)
    assert got.child_visits == want.child_visits


class _Remote:
    def __init__(self, fn):
        self.remote = fn


class _Storage:
    """shared_storage stand-in (SharedStorage.get_info / set_info): the loop runs `passes` times."""

    def __init__(self, weights, training_steps, passes):
        self.info = {"terminate": False, "weights": weights, "num_played_steps": 0}
        self.passes, self.training_steps, self.log = passes, training_steps, []
        self.get_info = _Remote(self._get)
        self.set_info = _Remote(self._set)

    def _get(self, key):
        if key == "training_step":
            # the loop condition is the first read of every pass (self_play.py:32-34); later reads see the same step
            return self.step
        return self.info[key]

    def _set(self, keys, values=None):
        self.log.append((keys, values))

    step = 0


class _Buffer:
    def __init__(self, storage, passes, training_steps):
        self.games = []
        self.save_game = _Remote(self._save)
        self.storage, self.passes, self.training_steps = storage, passes, training_steps

    def _save(self, game_history, shared_storage=None):
        self.games.append(game_history)
        if len(self.games) >= self.passes:
            self.storage.step = self.training_steps      # ends the loop at the next condition check


@pytest.mark.parametrize("test_mode", [False, True])
def test_continuous_self_play_equals_live_reference(test_mode):
    """The actor loop (self_play.py:31-108) with stand-in storage / buffer objects: same games, same reports."""
    import torch
    import hostcheck
    from mzx import self_play, synthetic
    ref_models, ref_self_play = ref_shim.load()
    Game = ref_shim.game_module("tictactoe").Game
    results = []
    for module, extra in ((ref_self_play, {}), (self_play, {"_backend": hostcheck.backend()})):
        cfg = ref_shim.game_module("tictactoe").MuZeroConfig()
        cfg.num_simulations, cfg.training_steps, cfg.ratio, cfg.self_play_delay = 10, 1000, None, 0
        torch.manual_seed(0)
        weights = synthetic.fill_state_dict(ref_models.MuZeroNetwork(cfg).state_dict(), 17)
        passes = 2
        storage = _Storage(weights, cfg.training_steps, passes)
        buffer = _Buffer(storage, passes, cfg.training_steps)
        actor = module.SelfPlay({"weights": weights}, Game, cfg, 21, **extra)
        if test_mode:
            # test mode reports through set_info and saves nothing: end after `passes` reports
            orig = storage._set

            def counting(keys, values=None, orig=orig, storage=storage):
                orig(keys, values)
                if isinstance(keys, dict) and "episode_length" in keys:
                    storage.reports = getattr(storage, "reports", 0) + 1
                if getattr(storage, "reports", 0) >= passes and isinstance(keys, dict) and "opponent_reward" in keys:
                    storage.step = cfg.training_steps
            storage.set_info = _Remote(counting)
        actor.continuous_self_play(storage, buffer, test_mode)
        results.append((buffer.games, storage.log))
    (ref_games, ref_log), (my_games, my_log) = results
    assert len(ref_games) == len(my_games) == (0 if test_mode else 2)
    for a, b in zip(ref_games, my_games):
        assert [int(x) for x in a.action_history] == [int(x) for x in b.action_history]
        assert a.reward_history == b.reward_history and a.child_visits == b.child_visits
        assert numpy.allclose(a.root_values, b.root_values, atol=3e-4, rtol=3e-4)
    assert len(ref_log) == len(my_log) == (4 if test_mode else 0)
    for (ka, _), (kb, _) in zip(ref_log, my_log):
        assert set(ka) == set(kb)
        for key in ka:
            assert numpy.allclose(ka[key], kb[key], atol=3e-4, rtol=3e-4), key


@pytest.mark.parametrize("game", ["cartpole", "tictactoe"])
def test_mcts_node_graph_equals_live_reference(game):
    """MCTS(config).run: the returned Node graph (every node, not only the root) against the reference's."""
    import torch
    import hostcheck
    from mzx import models, self_play, synthetic
    ref_models, ref_self_play = ref_shim.load()
    cfg = configs.BY_NAME[game](num_simulations=30)
    torch.manual_seed(0)
    ref = ref_models.MuZeroNetwork(cfg)
    weights = synthetic.fill_state_dict(ref.state_dict(), 23)
    ref.set_weights(weights)
    ref.eval()
    net = models.MuZeroNetwork(cfg, _backend=hostcheck.backend())
    net.set_weights(weights)
    obs = synthetic.observations(1, net.input_shape, seed=9)[0]
    legal = list(cfg.action_space)[1:] if game == "tictactoe" else list(cfg.action_space)
    numpy.random.seed(5)
    with torch.no_grad():
        want, want_info = ref_self_play.MCTS(cfg).run(ref, obs, legal, 0, True)
    numpy.random.seed(5)
    got, got_info = self_play.MCTS(cfg).run(net, obs, legal, 0, True)
    assert got_info["max_tree_depth"] == want_info["max_tree_depth"]
    assert abs(got_info["root_predicted_value"] - want_info["root_predicted_value"]) < 3e-4
    nodes = 0
    stack = [(want, got)]
    while stack:
        a, b = stack.pop()
        nodes += 1
        assert a.visit_count == b.visit_count and a.to_play == b.to_play and list(a.children) == list(b.children)
        assert abs(a.value_sum - b.value_sum) < 3e-4 * max(1.0, abs(a.value_sum)) and abs(a.reward - b.reward) < 3e-4
        if a.hidden_state is not None:
            assert numpy.abs(a.hidden_state.numpy() - b.hidden_state.cpu().numpy()).max() < 1e-4
        for action in a.children:
            ca, cb = a.children[action], b.children[action]
            assert abs(ca.prior - cb.prior) < 1e-5
            stack.append((ca, cb))
    assert nodes > cfg.num_simulations      # every expanded node and the unexpanded leaves under them
