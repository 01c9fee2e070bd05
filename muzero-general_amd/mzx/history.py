"""
Records and tree facades of the self-play surface (split out of mzx/self_play.py in round 6; ``mzx.self_play`` re-exports
every name).

Same names / fields as /root/reference/self_play.py: ``Node`` (:433-476), ``MinMaxStats`` (:553-570), ``GameHistory``
(:479-550) -- plus what a shard played through the batched protocol hands out instead of building B Python records move by
move: ``ShardGameHistory`` (a GameHistory whose list fields materialise on first touch from the shard's arrays),
``_ShardRecord`` (those arrays), ``ShardGames`` (the list a call returns, with its grouping by record).
"""
import contextlib
import gc

import numpy
import torch


class Node:
    """
    self_play.py:433-476.  The search itself runs on the device; ``Node`` objects are how callers see a
    searched tree (``BatchedMCTS.node_graph``) and how tooling hands a root in (``override_root_with``).
    """

    def __init__(self, prior):
        self.visit_count = 0
        self.to_play = -1
        self.prior = prior
        self.value_sum = 0
        self.children = {}
        self.hidden_state = None
        self.reward = 0

    def expanded(self):
        return len(self.children) > 0

    def value(self):
        if self.visit_count == 0:
            return 0
        return self.value_sum / self.visit_count

    def expand(self, actions, to_play, reward, policy_logits, hidden_state):
        """self_play.py:451-465 (host side: tooling such as diagnose_model.py:57-70 expands a root itself)."""
        self.to_play = to_play
        self.reward = reward
        self.hidden_state = hidden_state
        logits = policy_logits.detach().to("cpu", torch.float32)
        policy_values = torch.softmax(torch.tensor([logits[0][a] for a in actions]), dim=0).tolist()
        for action, p in zip(actions, policy_values):
            self.children[action] = Node(p)

    def add_exploration_noise(self, dirichlet_alpha, exploration_fraction):
        """self_play.py:467-476"""
        actions = list(self.children.keys())
        noise = numpy.random.dirichlet([dirichlet_alpha] * len(actions))
        frac = exploration_fraction
        for a, n in zip(actions, noise):
            self.children[a].prior = self.children[a].prior * (1 - frac) + n * frac


class MinMaxStats:
    """self_play.py:553-570 (the device keeps one per tree; this mirrors the class for tooling)."""

    def __init__(self):
        self.maximum = -float("inf")
        self.minimum = float("inf")

    def update(self, value):
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def normalize(self, value):
        if self.maximum > self.minimum:
            return (value - self.minimum) / (self.maximum - self.minimum)
        return value


class GameHistory:
    """self_play.py:479-550 -- field-identical record consumed by replay_buffer.py:33-65."""

    def __init__(self):
        self.observation_history = []
        self.action_history = []
        self.reward_history = []
        self.to_play_history = []
        self.child_visits = []
        self.root_values = []
        self.reanalysed_predicted_root_values = None
        # For PER
        self.priorities = None
        self.game_priority = None

    def store_search_statistics(self, root, action_space):
        # self_play.py:496-511
        if root is not None:
            total = sum(child.visit_count for child in root.children.values())
            self.child_visits.append(
                [root.children[a].visit_count / total if a in root.children else 0 for a in action_space]
            )
            self.root_values.append(root.value())
        else:
            self.root_values.append(None)

    def get_stacked_observations(self, index, num_stacked_observations, action_space_size):
        # self_play.py:513-550
        index = index % len(self.observation_history)
        current = self.observation_history[index]
        pieces = [current.copy() if hasattr(current, "copy") else numpy.array(current)]
        first_plane = pieces[0][0]
        for past in range(index - 1, index - 1 - num_stacked_observations, -1):
            if past >= 0:
                pieces.append(self.observation_history[past])
                pieces.append([numpy.ones_like(first_plane) * self.action_history[past + 1] / action_space_size])
            else:
                pieces.append(numpy.zeros_like(current))
                pieces.append([numpy.zeros_like(first_plane)])
        if len(pieces) == 1:
            return pieces[0]
        return numpy.concatenate(pieces)


class ShardGameHistory(GameHistory):
    """
    The GameHistory of one game of a shard played through the batched protocol.  The shard records a move as a
    handful of arrays over all games; this object is a VIEW of game ``i`` in them whose list-typed fields
    (``observation_history`` ... ``root_values``, self_play.py:482-489) are created on first access and are
    ordinary lists from then on -- field-identical to the eager record (tests/test_selfplay_shard.py).  A
    consumer pays only for the fields it touches (the replay buffer: ``root_values`` on save, the rest for
    sampled games); pickling (Ray object store) materialises everything.
    """
    _LAZY = ("observation_history", "action_history", "reward_history", "to_play_history", "child_visits", "root_values")
    # The PER fields (self_play.py:488-489) are lazy as well: ``mzx.replay.fill_initial_priorities_many`` stores the
    # priorities of a whole record ON the record ([games][moves] float32 + [games]); a view resolves its row on first
    # access (None while the record has none) -- handing 4096 games to the buffer sets nothing per game.  Assigning the
    # attribute (the stock ``save_game`` copies it, the trainer updates it) makes it an ordinary instance attribute.
    _PER = ("priorities", "game_priority")
    _WITH_LEADING_ENTRY = frozenset(("action_history", "reward_history", "to_play_history"))     # n + 1 entries (self_play.py:118-120)

    def __init__(self, source, i, n):
        self.__dict__["_view"] = (source, i, n)
        self.reanalysed_predicted_root_values = None

    @classmethod
    def make_many(cls, source, k, n):
        """k views of one record (games 0 .. k - 1, n moves each): what ``__init__`` sets, without attribute stores per
        object through the interpreter (a shard hands out thousands of games per call).  A record that already carries its
        PER priorities (``SelfPlay.continuous_self_play`` computes them while it collects a call's games) hands every view
        its row at once: the buffer reads them for every game, and an instance attribute costs nothing to read."""
        new = cls.__new__
        out = [new(cls) for _ in range(k)]
        if source.priorities is not None:
            for j, (h, p, t) in enumerate(zip(out, source.priorities, source.game_priority)):
                h.__dict__ = {"_view": (source, j, n), "reanalysed_predicted_root_values": None, "priorities": p, "game_priority": t}
            return out
        for j, h in enumerate(out):
            h.__dict__ = {"_view": (source, j, n), "reanalysed_predicted_root_values": None}
        return out

    def __getattr__(self, name):          # reached only while the field has not been materialised
        if name in ShardGameHistory._LAZY:
            source, i, n = self.__dict__["_view"]
            rows = source._lists.get(name)
            if rows is not None:          # the record's rows are lists already (_ShardRecord._row): hand this game's over
                plus = name in ShardGameHistory._WITH_LEADING_ENTRY
                if plus or source.simple[i]:
                    value = rows[i]
                    if value is not None and len(value) == n + plus:
                        rows[i] = None
                        self.__dict__[name] = value
                        return value
            value = source.field(name, i, n)
            self.__dict__[name] = value
            return value
        if name in ShardGameHistory._PER:
            source, i, _ = self.__dict__["_view"]
            rows = source.priorities if name == "priorities" else source.game_priority
            if rows is None:
                return None               # (not cached: the record may get its priorities later)
            value = rows[i]
            self.__dict__[name] = value
            return value
        raise AttributeError(name)

    def materialize(self):
        for name in ShardGameHistory._LAZY + ShardGameHistory._PER:
            value = getattr(self, name)
            if name in ShardGameHistory._PER:
                self.__dict__[name] = value
        return self

    def __getstate__(self):
        state = dict(self.materialize().__dict__)
        state.pop("_view", None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)


class _ShardRecord:
    """Game-major arrays of one finished shard (what ShardGameHistory objects view)."""

    def __init__(self, A, obs, acts, rews, tps, vis=None, vals=None, totals=None, ratios=None, simple=None, legal_mask=None,
                 time_major=False):
        if time_major:     # [move][game] arrays (the rows of a slot group's ring, copied out as they lie): views, no transposition
            sw = lambda a: None if a is None else numpy.swapaxes(a, 0, 1)
            obs, acts, rews, tps, vis, vals, totals, ratios, legal_mask = (sw(a) for a in (obs, acts, rews, tps, vis, vals, totals,
                                                                                             ratios, legal_mask))
        self.A, self.obs, self.acts, self.rews, self.tps = A, obs, acts, rews, tps
        self.vis, self.vals, self.totals, self.ratios, self.simple, self.legal_mask = vis, vals, totals, ratios, simple, legal_mask
        self._lists = {}        # field -> the whole record as nested Python lists, one row per game (first touch)
        self.priorities = self.game_priority = None      # [games][moves] float32, [games]: set for the whole record at once

    _ARRAY_OF = {"action_history": "acts", "reward_history": "rews", "to_play_history": "tps", "child_visits": "ratios",
                 "root_values": "vals"}

    def _row(self, name, i, length):
        """
        Game i's list of a list-typed field.  The FIRST touch of a field converts the whole record -- one contiguous
        game-major copy + one ``tolist()`` for all its games -- because the consumers touch every game (the actor's own
        ``fill_initial_priorities``, pickling for the Ray object store): per game that is a list hand-over instead of a
        strided gather of its moves.  Each row is handed out once (the history keeps it), so nothing is shared.
        """
        rows = self._lists.get(name)
        if rows is None:
            with gc_paused():      # (a million small objects at once)
                rows = self._lists[name] = numpy.ascontiguousarray(getattr(self, self._ARRAY_OF[name])).tolist()
        row = rows[i]
        if row is None:         # (handed out before: a second request goes to the arrays)
            return getattr(self, self._ARRAY_OF[name])[i, :length].tolist()
        rows[i] = None
        return row if len(row) == length else row[:length]

    def field(self, name, i, n):
        if name == "observation_history":
            if not self.obs.flags.c_contiguous:      # move-major rows of a slot group's ring: game-major once, for all games
                self.obs = numpy.ascontiguousarray(self.obs)
            return list(self.obs[i, : n + 1])
        if name in ("action_history", "reward_history", "to_play_history"):
            return self._row(name, i, n + 1)
        if n == 0:
            return []
        if self.simple[i]:      # every row "all actions legal, root visited": the record's own rows
            return self._row(name, i, n)
        out = []                 # illegal actions get 0, an unvisited root reports value 0 (self_play.py:496-511)
        for t in range(n):
            total = int(self.totals[i, t])
            if name == "child_visits":
                ok = (lambda a: True) if self.legal_mask is None else (lambda a: self.legal_mask[i, t, a])
                out.append([int(self.vis[i, t, a]) / total if ok(a) else 0 for a in range(self.A)])
            else:
                out.append(float(self.vals[i, t]) if total else 0)
        return out


class ShardGames(list):
    """The finished games a shard hands out in one call (a plain list of GameHistory objects) + ``records``: for the games
    that are fresh views of shard records, [(record, moves, [views in record order])] -- consumers that treat a record's
    games together (``mzx.replay.fill_initial_priorities_many``) need not rediscover the grouping game by game."""
    records = ()



@contextlib.contextmanager
def gc_paused():
    """
    Bulk allocation of small acyclic objects (thousands of history views, a record's nested lists, per-game priority
    arrays): with the cyclic collector on, every 700 allocations start a young-generation pass and the older
    generations -- everything the replay buffer keeps alive -- are re-scanned again and again for nothing.  Collection is
    only deferred to the end of the block.
    """
    collecting = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if collecting:
            gc.enable()
