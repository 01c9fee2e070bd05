"""
Reference-compat facade: ``MCTS.run`` returning the searched tree as ``Node``
objects and ``override_root_with`` -- the loop of diagnose_model.py:31-78, replayed with the mzx classes
against a trace of the unmodified reference (tests/golden/virtual_*.npz, oracle/make_golden.py).
Visit counts / depths identical, values and priors within tolerance.  GPU twin: tests/test_gpu_parity.py.
"""
import json
import os

import numpy
import pytest
import torch

import hostcheck
from conftest import GOLDEN
from mzx import configs, models, self_play, synthetic

TOL = 1e-4


@pytest.fixture(scope="module")
def backend():
    return hostcheck.backend()


def check_virtual_trajectory(backend, name):
    z = numpy.load(os.path.join(GOLDEN, f"virtual_{name}.npz"))
    meta = json.loads(str(z["meta"]))
    cfg = configs.BY_NAME[meta["game"]](**meta["overrides"])
    net = models.MuZeroNetwork(cfg, _backend=backend)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), meta["weight_seed"]))
    numpy.random.seed(meta["seed"])
    close = lambda a, b: numpy.allclose(a, b, atol=3 * TOL, rtol=3 * TOL)

    def check(step, root, info, action, reward, value):
        visits = [root.children[a].visit_count if a in root.children else 0 for a in cfg.action_space]
        assert visits == z["visits"][step].tolist(), (step, visits)
        assert root.visit_count == int(z["root_visits"][step]) and info["max_tree_depth"] == int(z["max_tree_depth"][step])
        assert close([root.children[a].prior for a in cfg.action_space], z["priors"][step])
        assert close([root.children[a].value() for a in cfg.action_space], z["child_values"][step])
        assert close([root.children[a].reward for a in cfg.action_space], z["child_rewards"][step])
        assert close(root.value(), z["root_value"][step])
        if step == 0:
            assert close(info["root_predicted_value"], z["predicted"][0])
        else:
            assert info["root_predicted_value"] is None          # self_play.py:277
            assert action == int(z["action"][step])
            assert close(reward, z["reward"][step]) and close(value, z["prior_value"][step])

    to_play = 0
    root, info = self_play.MCTS(cfg).run(net, z["observation"], cfg.action_space, to_play, True)
    check(0, root, info, None, None, None)
    # the returned graph goes below the root: grandchildren of the most visited child exist and are consistent
    best = max(root.children.values(), key=lambda c: c.visit_count)
    assert best.expanded() and sum(c.visit_count for c in best.children.values()) == best.visit_count - 1
    assert tuple(best.hidden_state.shape) == (1,) + tuple(net.hidden_shape)
    virtual_to_play = to_play
    for step in range(1, meta["horizon"] + 1):
        action = self_play.SelfPlay.select_action(root, 0)
        virtual_to_play = cfg.players[virtual_to_play + 1] if virtual_to_play + 1 < len(cfg.players) else cfg.players[0]
        value, reward, policy_logits, hidden_state = net.recurrent_inference(root.hidden_state, torch.tensor([[action]]))
        value = models.support_to_scalar(value, cfg.support_size, _backend=backend).item()
        reward = models.support_to_scalar(reward, cfg.support_size, _backend=backend).item()
        given = self_play.Node(0)
        given.expand(cfg.action_space, virtual_to_play, reward, policy_logits, hidden_state)
        root, info = self_play.MCTS(cfg).run(net, None, cfg.action_space, virtual_to_play, True, given)
        assert root is given                                     # searched in place, like the reference
        check(step, root, info, action, reward, value)


@pytest.mark.parametrize("name", ["cartpole", "tictactoe"])
def test_virtual_trajectory_matches_reference(backend, name):
    check_virtual_trajectory(backend, name)


def test_override_root_rejections(backend):
    cfg = configs.cartpole(num_simulations=5)
    net = models.MuZeroNetwork(cfg, _backend=backend)
    obs = synthetic.observations(1, net.input_shape, seed=1)[0]
    root, _ = self_play.MCTS(cfg).run(net, obs, cfg.action_space, 0, False)
    with pytest.raises(NotImplementedError):          # a root that already carries visits
        self_play.MCTS(cfg).run(net, None, cfg.action_space, 0, False, root)
    fresh = self_play.Node(0)
    with pytest.raises(ValueError):                   # not expanded
        self_play.BatchedMCTS(cfg, net, 1, mode=0).run_from_roots([fresh], [0], False, [numpy.random.RandomState(0)])
    v, r, p, h = net.recurrent_inference(root.hidden_state, torch.tensor([[1]]))
    fresh.expand(cfg.action_space, 1, 0.0, p, h)
    with pytest.raises(NotImplementedError):          # root.to_play differs from the argument
        self_play.MCTS(cfg).run(net, None, cfg.action_space, 0, False, fresh)
