"""
Pins the CPU oracle (oracle/*.py) against traces of the UNMODIFIED reference
(tests/golden/*, produced by oracle/make_golden.py in the build container).
Bit-exact for every integer and float64 tree statistic; fp32 network outputs
are compared exactly too (same ATen CPU kernels, same op order, batch 1 and N).
"""
import json
import os

import numpy
import pytest
import torch

from mzx import configs, synthetic
import lockstep
from oracle import mcts_oracle, net_oracle

from conftest import GOLDEN


def load(name):
    z = numpy.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return z, json.loads(str(z["meta"]))


def config_for(meta):
    return configs.BY_NAME[meta["game"]](**meta.get("overrides", {}))


def state_dict_for(meta_net_name, weight_seed, zero_keys=()):
    z, meta = load(meta_net_name)
    template = {k: torch.zeros(shape, dtype=getattr(torch, dt.split(".")[1])) for k, shape, dt in meta["keys"]}
    sd = synthetic.fill_state_dict(template, weight_seed)
    for k in zero_keys:
        sd[k] = torch.zeros_like(sd[k])
    return sd


NET_FOR_GAME = {"cartpole": "net_fc_cartpole.npz", "tictactoe": "net_resnet_tictactoe.npz",
                "connect4": "net_resnet_connect4.npz", "lunarlander": "net_fc_lunarlander_pretrained.npz"}


def compare_tree(tree, z, c, cfg):
    n = len(tree.visit)
    g = lambda k: z[f"c{c}_{k}"]
    assert n == g("visit").shape[0]
    assert numpy.array_equal(numpy.array(tree.visit, numpy.int32), g("visit"))
    # float64 statistics: compare BIT PATTERNS
    assert numpy.array_equal(numpy.array(tree.value_sum, numpy.float64).view(numpy.int64),
                             g("value_sum").view(numpy.int64))
    assert numpy.array_equal(numpy.array(tree.reward, numpy.float64).view(numpy.int64),
                             g("reward").view(numpy.int64))
    assert numpy.array_equal(numpy.array(tree.to_play, numpy.int32), g("to_play"))
    assert numpy.array_equal(numpy.array(tree.parent, numpy.int32), g("parent"))
    for i in range(n):
        k = len(tree.actions[i])
        assert k == g("n_children")[i]
        assert list(g("child_action")[i, :k]) == list(tree.actions[i])
        assert list(g("child")[i, :k]) == list(tree.child[i])
        assert numpy.array_equal(numpy.array(tree.prior[i]).view(numpy.int64),
                                 g("prior")[i, :k].view(numpy.int64))
    # per-simulation trace = (parent, action, depth) of node k+1
    for k, (parent, action, depth) in enumerate(tree.trace):
        assert parent == g("parent")[k + 1]
        assert action == g("parent_action")[k + 1]
        assert depth == g("depth")[k + 1]
    assert numpy.array_equal(numpy.array([tree.minimum, tree.maximum]).view(numpy.int64),
                             g("minmax").view(numpy.int64))
    assert tree.max_depth == int(g("max_tree_depth"))
    assert tree.root_predicted_value == float(g("root_predicted_value"))


@pytest.mark.parametrize("name", ["cartpole", "tictactoe", "connect4", "cartpole_ties", "tictactoe_custom",
                                  "cartpole_custom", "lunarlander_pretrained"])
def test_tree_lockstep_bit_exact(name):
    """Tree arithmetic alone: network outputs replayed from the reference run."""
    z, meta = load(f"tree_{name}.npz")
    cfg = config_for(meta)
    total_ties = 0
    for c, case in enumerate(meta["cases"]):
        ev = mcts_oracle.ReplayEvaluator(z[f"c{c}_net_value"], z[f"c{c}_net_reward"], z[f"c{c}_net_priors"])
        rng = numpy.random.RandomState(case["rng_seed"])
        tree = mcts_oracle.run_search(cfg, ev, z[f"c{c}_obs"], case["legal"], case["to_play"], True, rng)
        compare_tree(tree, z, c, cfg)
        total_ties += tree.tie_draws
    if name == "cartpole_ties":
        assert total_ties > 4 * len(meta["cases"])  # the fixture really exercises repeated ties
    else:
        assert total_ties >= sum(len(c["legal"]) > 1 for c in meta["cases"])


@pytest.mark.parametrize("name", ["cartpole", "tictactoe", "connect4", "cartpole_ties", "tictactoe_custom",
                                  "cartpole_custom", "lunarlander_pretrained"])
def test_search_end_to_end_bit_exact(name):
    """Oracle network + oracle tree vs the reference's models.py + self_play.py."""
    z, meta = load(f"tree_{name}.npz")
    cfg = config_for(meta)
    sd = state_dict_for(NET_FOR_GAME[meta["game"]], meta["weight_seed"] or 0, meta.get("zero_keys", ()))
    sd = lockstep.fixture_weights(z, sd) or sd      # trained weights shipped with the fixture
    net = net_oracle.make_oracle_network(cfg, sd)
    for c, case in enumerate(meta["cases"]):
        ev = net_oracle.NetworkEvaluator(net, cfg.support_size, record=True)
        rng = numpy.random.RandomState(case["rng_seed"])
        tree = mcts_oracle.run_search(cfg, ev, z[f"c{c}_obs"], case["legal"], case["to_play"], True, rng)
        compare_tree(tree, z, c, cfg)
        got = numpy.stack([e["policy_logits"] for e in ev.log])
        assert numpy.array_equal(got, z[f"c{c}_policy_logits"])
        assert numpy.array_equal(numpy.stack([e["value_logits"] for e in ev.log]), z[f"c{c}_value_logits"])


@pytest.mark.parametrize("name", ["fc_cartpole", "fc_cartpole_pretrained", "fc_lunarlander_pretrained", "fc_cartpole_stacked",
                                  "resnet_tictactoe", "resnet_connect4", "resnet_breakout", "resnet_breakout_cnn",
                                  "resnet_cnn_small"])
def test_network_outputs(name):
    z, meta = load(f"net_{name}.npz")
    cfg = config_for(meta)
    template = {k: torch.zeros(shape, dtype=getattr(torch, dt.split(".")[1])) for k, shape, dt in meta["keys"]}
    if "flat_weights" in z.files:
        sd, off = {}, 0
        for k, t in template.items():
            if t.dtype.is_floating_point:
                sd[k] = torch.from_numpy(z["flat_weights"][off:off + t.numel()].reshape(t.shape).copy())
                off += t.numel()
    else:
        sd = synthetic.fill_state_dict(template, meta["weight_seed"])
    net = net_oracle.make_oracle_network(cfg, sd)
    with torch.no_grad():
        o = net.initial_inference(torch.tensor(z["obs"]))
        r1 = net.recurrent_inference(o[3], torch.tensor(z["act1"]).long())
        r2 = net.recurrent_inference(r1[3], torch.tensor(z["act2"]).long())
        o1 = net.initial_inference(torch.tensor(z["obs"][:1]))
    for tag, res in (("init", o), ("rec1", r1), ("rec2", r2), ("init_b1", o1)):
        for key, t in zip(("value", "reward", "policy", "hidden"), res):
            ref = z[f"{tag}_{key}"]
            assert t.shape == ref.shape
            assert numpy.array_equal(t.numpy(), ref), (name, tag, key, numpy.abs(t.numpy() - ref).max())
        vs = net_oracle.support_to_scalar(res[0], cfg.support_size).numpy()
        assert numpy.array_equal(vs, z[f"{tag}_value_scalar"])
        rs = net_oracle.support_to_scalar(res[1], cfg.support_size).numpy()
        assert numpy.array_equal(rs, z[f"{tag}_reward_scalar"])
    # known-answer property of the reference: the root reward decodes to exactly 0
    assert numpy.all(z["init_reward_scalar"] == 0.0)


def test_rng_tape_matches_numpy_choice():
    """
    The engine replaces numpy.random.choice(ties) by masked rejection over a tape
    of raw MT19937 words (DESIGN.md "tie tape"); pin that equivalence here.
    """
    for seed in range(50):
        for n in (2, 3, 4, 5, 7, 9):
            r = numpy.random.RandomState(seed)
            r.dirichlet([0.25] * n)
            state = r.get_state()
            want = r.choice(list(range(n)))
            after = r.get_state()
            t = numpy.random.RandomState()
            t.set_state(state)
            tape = t.randint(0, 2 ** 32, size=64, dtype=numpy.uint32)
            rng, mask, used = n - 1, n - 1, 0
            for s in (1, 2, 4, 8, 16):
                mask |= mask >> s
            while True:
                w = int(tape[used]) & mask
                used += 1
                if w <= rng:
                    break
            assert w == want
            t.set_state(state)
            if used:
                t.randint(0, 2 ** 32, size=used, dtype=numpy.uint32)
            assert t.get_state()[2] == after[2] and numpy.array_equal(t.get_state()[1], after[1])


def test_select_action_and_policy_row():
    rng = numpy.random.RandomState(3)
    assert mcts_oracle.select_action([1, 9, 3], [4, 5, 6], 0, rng) == 5
    a = mcts_oracle.select_action([1, 9, 3], [4, 5, 6], 1.0, numpy.random.RandomState(3))
    ref = numpy.random.RandomState(3).choice([4, 5, 6], p=numpy.array([1, 9, 3]) / 13)
    assert a == ref
