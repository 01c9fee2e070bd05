"""
N>1 path on CPU: world_size-2 gloo.  The hot path shards by games (no data-path
collective); the ONE exchange step is the flat-weight broadcast
(mzx.shared_storage.broadcast_weights; RCCL on the GPU box, gloo here).  The
model in these processes is bound to the serial test double of the ABI
(tests/hostcheck), which exercises the same host code: flat buffer layout,
refresh of derived BatchNorm terms, shard seeding.
"""
import os
import socket
import sys

import numpy
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hostcheck
    from mzx import configs, models, self_play, shared_storage, synthetic

    be = hostcheck.backend()
    cfg = configs.tictactoe()  # resnet: has BatchNorm-derived terms that must be refreshed
    net = models.MuZeroNetwork(cfg, _backend=be)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 40 + rank))  # ranks start different
    before = net.flat_weights().clone()
    shared_storage.broadcast_weights(net, src=0)
    after = net.flat_weights().clone()
    # every rank searches the SAME roots with the SAME streams: results must now coincide
    B = 4
    obs = synthetic.observations(B, cfg.observation_shape, seed=77)
    engine = self_play.BatchedMCTS(cfg, net, B)
    res = engine.run(list(obs), [list(cfg.action_space)] * B, [0] * B, True,
                     [numpy.random.RandomState(5 + i) for i in range(B)])
    seeds = shared_storage.shard_seeds(cfg.seed, 3)
    torch.save(dict(before=before, after=after, visits=res.visit_counts, root_values=res.root_values, seeds=seeds),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert not torch.equal(r0["before"], r1["before"])
    assert torch.equal(r0["after"], r0["before"])            # the source keeps its weights
    assert torch.equal(r1["after"], r0["after"])             # the other rank received them
    assert numpy.array_equal(r0["visits"], r1["visits"])     # ... and its derived BN terms were refreshed
    assert numpy.array_equal(r0["root_values"], r1["root_values"])
    assert r0["seeds"] == [0, 1, 2] and r1["seeds"] == [3, 4, 5]  # disjoint game shards (muzero.py:185)


# ----------------------------------------------------------------------------- the actor loop, sharded
class _ScriptedStorage:
    """
    Rank 0's real storage (duck type of shared_storage.py:7-40) playing the trainer: after the actors have
    reported their first round of games it publishes NEW weights at a later training step, after the second
    round it sets ``terminate``.
    """

    def __init__(self, weights_a, weights_b):
        from mzx import shared_storage
        self.inner = shared_storage.LocalStorage(training_step=0, terminate=False, weights=weights_a,
                                                 num_played_games=0, num_played_steps=0)
        self.weights_b = weights_b
        self.reports = 0

    def get_info(self, keys):
        return self.inner.get_info(keys)

    def set_info(self, keys, values=None):
        self.inner.set_info(keys, values)
        if isinstance(keys, dict) and keys.get("num_played_steps", 0) > 0:   # job-wide counts after a round
            self.reports += 1
            if self.reports == 1:
                self.inner.set_info({"weights": self.weights_b, "training_step": 7})
            elif self.reports == 2:
                self.inner.set_info("terminate", True)


class _ListBuffer:
    """replay_buffer.py:33-65 reduced to what the actor loop touches: keep the game, report the counts."""

    def __init__(self):
        self.games, self.steps = [], 0

    def save_game(self, game_history, shared_storage=None):
        assert game_history.priorities is not None       # filled by the actor (vectorised initial PER priorities)
        self.games.append(game_history)
        self.steps += len(game_history.root_values)
        if shared_storage:
            shared_storage.set_info("num_played_games", len(self.games))
            shared_storage.set_info("num_played_steps", self.steps)


def _actor_worker(rank, world, port, out_dir, pipeline=False):
    for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import games_fixture
    import hostcheck
    from mzx import configs, models, self_play, shared_storage, synthetic

    be = hostcheck.backend()
    cfg = configs.tictactoe()
    cfg.num_simulations = 8
    cfg.training_steps = 100
    cfg.ratio = None
    cfg.self_play_delay = 0
    cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 9      # games/tictactoe.py:87-96
    cfg.self_play_pipeline = bool(pipeline)   # True: two slot groups take turns, searches on a worker thread (SelfPlay._slot_groups)
    G = 3
    template = models.MuZeroNetwork(cfg, _backend=be).state_dict()
    wa, wb = synthetic.fill_state_dict(template, 1), synthetic.fill_state_dict(template, 2)
    seeds = shared_storage.shard_seeds(cfg.seed, G)
    made = []

    class Game(games_fixture.GAMES["tictactoe"]):
        def __init__(self, seed=None):
            made.append(seed)
            super().__init__(seed)

    if pipeline == "batched":      # the batched game protocol, two slot groups that take turns (SelfPlay._rounds_batched)
        from mzx import games as board_games

        class Game(board_games.TicTacToeBatched):      # noqa: F811
            def __init__(self, seeds):
                made.extend(s for s in seeds if s not in made)      # (each slot group builds its own object over its seeds)
                super().__init__(seeds)

    # every rank starts from its OWN junk weights: what it plays with must come through the broadcast
    start = synthetic.fill_state_dict(template, 50 + rank)
    actor = self_play.SelfPlay({"weights": start}, Game, cfg, seeds[0], num_games=G, _backend=be)
    storage = shared_storage.ShardedStorage(_ScriptedStorage(wa, wb) if rank == 0 else None, src=0)
    buffer = _ListBuffer()
    seen = []
    play_rounds = actor.play_rounds      # (continuous_self_play plays rounds: finished slots are refilled at once)

    def recording_play_rounds(*a, **kw):
        seen.append(actor.model.flat_weights().clone())      # the weights each call is played with
        return play_rounds(*a, **kw)

    actor.play_rounds = recording_play_rounds
    actor.continuous_self_play(storage, buffer)
    job_counts = storage.storage.inner.get_info(["num_played_games", "num_played_steps"]) if rank == 0 else None
    torch.save(dict(seeds=made, rounds=len(seen), seen=seen, games=len(buffer.games), steps=buffer.steps,
                    refreshes=storage.refreshes, broadcasts=storage.weight_broadcasts, job_counts=job_counts,
                    control=storage.control,
                    first_actions=[[int(a) for a in g.action_history] for g in buffer.games]),
               os.path.join(out_dir, f"actor{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("pipeline", [False, True, "batched"])
def test_continuous_self_play_two_ranks_with_midrun_weight_update(tmp_path, pipeline):
    world, port = 2, _free_port()
    mp.spawn(_actor_worker, args=(world, port, str(tmp_path), pipeline), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"actor{k}.pt", weights_only=False) for k in range(world)]
    # disjoint game shards, seeded like the reference's workers (muzero.py:185)
    assert r[0]["seeds"] == [0, 1, 2] and r[1]["seeds"] == [3, 4, 5]
    # The exchange is asynchronous: how many rounds a rank plays before it sees the update / the stop flag depends on
    # timing.  What does not: both ranks consumed the SAME sequence of control exchanges and stopped on the same
    # one, with the same control values ...
    assert r[0]["refreshes"] == r[1]["refreshes"] >= 3
    assert r[0]["control"] == r[1]["control"] and r[0]["control"]["terminate"] is True
    # ... one broadcast per published version (the initial weights, the mid-run update), none in between ...
    assert r[0]["broadcasts"] == r[1]["broadcasts"] == 2
    # ... every round was played with the trainer's weights -- first version, then the update, on BOTH ranks,
    # although only rank 0 can see the storage (each rank started from its own junk weights)
    versions = []
    for k in range(world):
        distinct = []
        for w in r[k]["seen"]:
            if not distinct or not torch.equal(distinct[-1], w):
                distinct.append(w)
        assert len(distinct) == 2 and r[k]["rounds"] >= 2, (k, len(distinct), r[k]["rounds"])
        versions.append(distinct)
    assert torch.equal(versions[0][0], versions[1][0]) and torch.equal(versions[0][1], versions[1][1])
    # the job-wide played counts reached the real storage (what the trainer's ratio throttle reads); they are as
    # fresh as the last exchange
    total_games = r[0]["games"] + r[1]["games"]
    assert 0 < r[0]["job_counts"]["num_played_games"] <= total_games
    assert 0 < r[0]["job_counts"]["num_played_steps"] <= r[0]["steps"] + r[1]["steps"]
    # different seeds -> different games
    assert r[0]["first_actions"] != r[1]["first_actions"]


class _CountingStorage:
    """Rank 0's real storage: terminates the job once the job-wide number of played games reaches a target."""

    def __init__(self, weights, target_games):
        from mzx import shared_storage
        self.inner = shared_storage.LocalStorage(training_step=0, terminate=False, weights=weights,
                                                 num_played_games=0, num_played_steps=0)
        self.target = target_games

    def get_info(self, keys):
        return self.inner.get_info(keys)

    def set_info(self, keys, values=None):
        self.inner.set_info(keys, values)
        if isinstance(keys, dict) and keys.get("num_played_games", 0) >= self.target:
            self.inner.set_info("terminate", True)


def _uneven_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    import games_fixture
    import hostcheck
    from mzx import configs, models, self_play, shared_storage, synthetic

    be = hostcheck.backend()
    cfg = configs.tictactoe()
    cfg.num_simulations = 4
    cfg.training_steps = 100
    cfg.ratio = None
    cfg.self_play_delay = 0
    cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 9
    G = 2
    template = models.MuZeroNetwork(cfg, _backend=be).state_dict()
    weights = synthetic.fill_state_dict(template, 1)

    class Game(games_fixture.GAMES["tictactoe"]):
        def step(self, action):
            if rank == 1:
                time.sleep(0.03)        # rank 1's environment is slow: its games take several times longer
            return super().step(action)

    seeds = shared_storage.shard_seeds(cfg.seed, G)
    actor = self_play.SelfPlay({"weights": synthetic.fill_state_dict(template, 60 + rank)}, Game, cfg, seeds[0],
                               num_games=G, _backend=be)
    storage = shared_storage.ShardedStorage(_CountingStorage(weights, 40) if rank == 0 else None, src=0)
    buffer = _ListBuffer()
    rounds = []
    play_rounds = actor.play_rounds

    def counting_play_rounds(*a, **kw):
        rounds.append(time.perf_counter())
        return play_rounds(*a, **kw)

    actor.play_rounds = counting_play_rounds
    t0 = time.perf_counter()
    actor.continuous_self_play(storage, buffer)
    torch.save(dict(rounds=len(rounds), games=len(buffer.games), wall=time.perf_counter() - t0,
                    refreshes=storage.refreshes, stalls=storage.polls_without_progress, control=storage.control),
               os.path.join(out_dir, f"uneven{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_game_lengths_do_not_lock_step_the_ranks(tmp_path):
    """
    Rank 1 plays through a slow environment.  With a blocking exchange per loop iteration both ranks would play the
    same number of rounds; with the asynchronous exchange the fast rank keeps playing while the slow one is inside
    a game, and both still stop on the same control exchange.
    """
    world, port = 2, _free_port()
    mp.spawn(_uneven_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"uneven{k}.pt", weights_only=False) for k in range(world)]
    assert r[0]["refreshes"] == r[1]["refreshes"]                       # same sequence of exchanges ...
    assert r[0]["control"] == r[1]["control"] and r[0]["control"]["terminate"] is True
    assert r[0]["rounds"] >= 2 * r[1]["rounds"] >= 2, (r[0]["rounds"], r[1]["rounds"])   # ... at each rank's own pace
    assert r[0]["stalls"] > 0                                           # polls that found the exchange incomplete and moved on
    assert r[0]["games"] + r[1]["games"] >= 40


class _VersionedStorage:
    """Rank 0's real storage playing the trainer for the world-4 case: publishes NEW weights once the job has reported
    ``update_at`` games, terminates at ``target`` games."""

    def __init__(self, weights_a, weights_b, update_at, target):
        from mzx import shared_storage
        self.inner = shared_storage.LocalStorage(training_step=0, terminate=False, weights=weights_a,
                                                 num_played_games=0, num_played_steps=0)
        self.weights_b, self.update_at, self.target, self.updated = weights_b, update_at, target, False

    def get_info(self, keys):
        return self.inner.get_info(keys)

    def set_info(self, keys, values=None):
        self.inner.set_info(keys, values)
        if isinstance(keys, dict):
            played = keys.get("num_played_games", 0)
            if played >= self.update_at and not self.updated:
                self.updated = True
                self.inner.set_info({"weights": self.weights_b, "training_step": 11})
            if played >= self.target:
                self.inner.set_info("terminate", True)


def _world4_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "muzero-general_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    import games_fixture
    import hostcheck
    from mzx import configs, games as board_games, models, self_play, shared_storage, synthetic

    torch.set_num_threads(1)
    be = hostcheck.backend()
    cfg = configs.tictactoe()
    cfg.num_simulations = 4
    cfg.training_steps = 100
    cfg.ratio = None
    cfg.self_play_delay = 0
    cfg.PER, cfg.PER_alpha, cfg.td_steps = True, 0.5, 9
    G = 4
    template = models.MuZeroNetwork(cfg, _backend=be).state_dict()
    wa, wb = synthetic.fill_state_dict(template, 1), synthetic.fill_state_dict(template, 2)
    delay = (0.0, 0.004, 0.0, 0.02)[rank]

    class Slow(games_fixture.GAMES["tictactoe"]):
        def step(self, action):
            if delay:
                time.sleep(delay)
            return super().step(action)

    if rank == 2:       # a natively played shard: rounds inside the library on a worker thread, the hand-off overlapped
        board_games.NativeBatchedGame.backend = be
        Game = board_games.TicTacToeNative
    else:               # per-object plugin games as two slot groups: a search of one group is queued across calls
        Game = Slow
        cfg.self_play_pipeline = True
    seeds = shared_storage.shard_seeds(cfg.seed, G)
    actor = self_play.SelfPlay({"weights": synthetic.fill_state_dict(template, 70 + rank)}, Game, cfg, seeds[0],
                               num_games=G, _backend=be)
    storage = shared_storage.ShardedStorage(_VersionedStorage(wa, wb, 12, 60) if rank == 0 else None, src=0)
    buffer = _ListBuffer()
    seen, queued_at_refresh = [], []
    refresh = storage.refresh

    def recording_refresh(model, *a, **kw):
        live = actor._live or {}
        queued_at_refresh.append(any(g.get("pending") is not None for g in live.get("groups", ())))
        before = model.flat_weights().clone()
        out = refresh(model, *a, **kw)
        if not torch.equal(before, model.flat_weights()):
            seen.append((len(queued_at_refresh), queued_at_refresh[-1]))     # a refresh that changed the weights
        return out

    storage.refresh = recording_refresh
    t0 = time.perf_counter()
    actor.continuous_self_play(storage, buffer)
    final = actor.model.flat_weights().clone()
    torch.save(dict(games=len(buffer.games), steps=buffer.steps, wall=time.perf_counter() - t0, refreshes=storage.refreshes,
                    broadcasts=storage.weight_broadcasts, control=storage.control, changed=seen,
                    queued_at_refresh=queued_at_refresh, final=final, searches=actor.stats["searches"],
                    native=bool(actor._live and actor._live.get("native"))),
               os.path.join(out_dir, f"w4_{rank}.pt"))
    actor.close_game()
    dist.barrier()
    dist.destroy_process_group()


def test_world4_unequal_ranks_midrun_update_with_searches_queued_across_calls(tmp_path):
    """
    Four ranks at four speeds (rank 3's environment is 5 x slower than rank 1's, ranks 0 and 2 do not wait at all; rank 2
    plays a NATIVE shard whose rounds run on a worker thread with the hand-off overlapped), a weight update published in
    the middle of the run, per-object shards pipelined as two slot groups -- so that a refresh can land while a rank holds a
    search it queued before ("weight staleness of a pipelined shard": that search runs on the old
    weights, everything after it on the new ones).  What must hold whatever the timing: every rank consumed the same
    sequence of control exchanges and stopped on the same one; exactly two broadcasts (the initial weights, the update);
    every rank ends on the trainer's SECOND weights; the job reached its target; the fast ranks were not held back.
    """
    world, port = 4, _free_port()
    mp.spawn(_world4_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"w4_{k}.pt", weights_only=False) for k in range(world)]
    assert len({x["refreshes"] for x in r}) == 1 and r[0]["refreshes"] >= 3
    assert all(x["control"] == r[0]["control"] for x in r) and r[0]["control"]["terminate"] is True
    assert all(x["broadcasts"] == 2 for x in r)
    assert all(torch.equal(x["final"], r[0]["final"]) for x in r)
    # the weights changed exactly twice on every rank (junk -> first version -> update)
    assert all(len(x["changed"]) == 2 for x in r), [x["changed"] for x in r]
    assert sum(x["games"] for x in r) >= 60
    assert r[2]["native"] and not r[0]["native"]
    assert r[0]["searches"] > 2 * r[3]["searches"] > 0, [x["searches"] for x in r]     # nobody waits for the slow rank
    # the staleness path was exercised: some refresh of a pipelined rank ran with a search queued from the call before
    assert any(any(x["queued_at_refresh"]) for k, x in enumerate(r) if k != 2), [sum(x["queued_at_refresh"]) for x in r]
