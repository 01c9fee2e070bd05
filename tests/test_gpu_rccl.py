"""
The RCCL side of the weight refresh on the 1-GPU box.

The N > 1 path is covered by the gloo tests on CPU (tests/test_dist_gloo.py: same host code, other transport).  What a
single MI355X can still check is that the tensors this package hands to ``torch.distributed`` are something RCCL takes
ON THE DEVICE: a process group of ONE rank with backend "nccl" (= RCCL on ROCm), the collectives issued for real
(``ShardedStorage(collectives_with_one_rank=True)``, ``broadcast_weights(with_one_rank=True)``; a lone rank skips them
by default) -- the six-word binary64 control all-reduce from a device tensor, the staged broadcast of the flat fp32
weight buffer, ``refresh_derived()`` after it, searches before and after.  Replaces, for the self-play side, what the
reference does through Ray's object store (shared_storage.py:7-40, self_play.py:33-37, muzero.py:177-196).

Runs in a child process (its own process group, a timeout of its own): a hung communicator must not take the suite along.
"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import os, sys, json
for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy, torch
import torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(PORT)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mzx import _lib, configs, games, models, self_play, shared_storage, synthetic
from test_dist_gloo import _ListBuffer


class _Trainer:
    # The real storage (duck type of shared_storage.py:7-40) playing the trainer: new weights at a later training step
    # once the first games are in, ``terminate`` once a second shard's worth of games is.

    def __init__(self, weights_a, weights_b, shard):
        self.inner = shared_storage.LocalStorage(training_step=0, terminate=False, weights=weights_a,
                                                 num_played_games=0, num_played_steps=0)
        self.weights_b, self.shard, self.published = weights_b, shard, False

    def get_info(self, keys):
        return self.inner.get_info(keys)

    def set_info(self, keys, values=None):
        self.inner.set_info(keys, values)
        played = self.inner.get_info("num_played_games")
        if played >= 1 and not self.published:
            self.published = True
            self.inner.set_info({"weights": self.weights_b, "training_step": 7})
        if played >= 2 * self.shard:
            self.inner.set_info("terminate", True)

be = _lib.default_backend()
out = {"backend": dist.get_backend()}

# ---- the collective alone: the flat buffer of a residual network (BatchNorm-derived terms to refresh), on the device
cfg = configs.tictactoe()
net = models.MuZeroNetwork(cfg, _backend=be)
net.set_weights(synthetic.fill_state_dict(net.state_dict(), 41))
flat = net.flat_weights()
out["flat_device"] = str(flat.device)
before = flat.clone()
B = 8
obs = synthetic.observations(B, cfg.observation_shape, seed=77)
def search():
    engine = self_play.BatchedMCTS(cfg, net, B)
    r = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True, [numpy.random.RandomState(5 + i) for i in range(B)])
    return numpy.asarray(r.visit_counts).tolist()
v0 = search()
shared_storage.broadcast_weights(net, src=0, with_one_rank=True)
torch.cuda.synchronize()
out["flat_unchanged"] = bool(torch.equal(before, net.flat_weights()))
out["same_search"] = v0 == search()
word = torch.tensor([3.0, 0.0, 7.0, 2.0, 18.0, 1.0], dtype=torch.float64, device="cuda")
work = dist.all_reduce(word, op=dist.ReduceOp.SUM, async_op=True)
work.wait()
out["control_word"] = word.cpu().tolist()

# ---- the actor loop over a ShardedStorage whose lone rank issues its collectives: initial weights, a mid-run update, stop
cfg = configs.tictactoe()
cfg.num_simulations = 8
cfg.training_steps = 100
cfg.ratio = None
cfg.self_play_delay = 0
cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 9
G = 3
template = models.MuZeroNetwork(cfg, _backend=be).state_dict()
wa, wb = synthetic.fill_state_dict(template, 1), synthetic.fill_state_dict(template, 2)
start = synthetic.fill_state_dict(template, 50)
actor = self_play.SelfPlay({"weights": start}, games.TicTacToeBatched, cfg, 0, num_games=G, _backend=be)
storage = shared_storage.ShardedStorage(_Trainer(wa, wb, G), src=0, collectives_with_one_rank=True)
buffer = _ListBuffer()
seen = []
play_rounds = actor.play_rounds
def recording(*a, **kw):
    seen.append(actor.model.flat_weights().clone())
    return play_rounds(*a, **kw)
actor.play_rounds = recording
actor.continuous_self_play(storage, buffer)
ref = models.MuZeroNetwork(cfg, _backend=be)
versions = []
for w in (wa, wb):
    ref.set_weights(w)
    versions.append(ref.flat_weights().clone())
distinct = []
for w in seen:
    if not distinct or not torch.equal(distinct[-1], w):
        distinct.append(w)
out.update(rounds=len(seen), distinct=len(distinct),
           first_is_a=bool(torch.equal(distinct[0], versions[0])), last_is_b=bool(torch.equal(distinct[-1], versions[1])),
           refreshes=storage.refreshes, broadcasts=storage.weight_broadcasts, issued=storage.lone_collectives_issued,
           terminate=storage.control["terminate"], games=len(buffer.games))
dist.barrier()
dist.destroy_process_group()
print("RCCL1 " + json.dumps(out))
"""


def _run_child(tmp_path, on_device):
    from test_dist_gloo import _free_port

    child = CHILD
    if not on_device:      # the same script over gloo and the serial test double of the ABI (the flag's host logic, CI)
        for a, b in (("torch.cuda.set_device(0)\n", ""), ("torch.cuda.synchronize()\n", ""), ('device="cuda"', 'device="cpu"'),
                     ('dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))',
                      'dist.init_process_group("gloo", rank=0, world_size=1)'),
                     ("be = _lib.default_backend()", "import hostcheck; be = hostcheck.backend()")):
            assert a in child
            child = child.replace(a, b)
    script = tmp_path / "rccl_child.py"
    script.write_text(f"ROOT = {ROOT!r}\nPORT = {_free_port()}\n" + child)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RCCL1 ")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    import json

    return json.loads(lines[-1][6:])


def _check(r):
    # the broadcast of a lone rank leaves its buffer alone; derived terms rebuilt from the same weights: the same search
    assert r["flat_unchanged"] and r["same_search"]
    assert r["control_word"] == [3.0, 0.0, 7.0, 2.0, 18.0, 1.0]
    # the actor loop: every round on the trainer's weights (never the junk it started from), first version then the
    # update, one staged broadcast per published version, each exchange a real collective
    assert r["distinct"] == 2 and r["first_is_a"] and r["last_is_b"] and r["rounds"] >= 2
    assert r["broadcasts"] == 2 and r["refreshes"] >= 3 and r["terminate"] is True and r["games"] > 0
    assert r["issued"] == r["refreshes"] + r["broadcasts"]


@pytest.mark.gpu
def test_rccl_one_rank_weight_refresh_on_device(tmp_path):
    r = _run_child(tmp_path, True)
    print("RCCL, one rank on the device:", r)
    assert r["backend"] == "nccl" and r["flat_device"].startswith("cuda")
    _check(r)


def test_one_rank_collectives_over_gloo(tmp_path):
    """The host logic of ``collectives_with_one_rank`` / ``with_one_rank`` on CPU (gloo, the serial test double)."""
    r = _run_child(tmp_path, False)
    assert r["backend"] == "gloo"
    _check(r)
