"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the MCTS hot path.

This file is a from-scratch, array-based CPU restatement of the reference's
per-move Monte-Carlo tree search.  It is the *checker* the HIP path is compared
with; nothing under ``muzero-general_amd/`` may import it (only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks this file bit-for-bit
against traces of the unmodified reference (``/root/reference/self_play.py``
executed in the build container by ``oracle/make_golden.py``; fixtures under
``tests/golden/``).

What is restated (reference file:line, relative to /root/reference):
  * MCTS.run                self_play.py:260-361
  * MCTS.select_child       self_play.py:363-378
  * MCTS.ucb_score          self_play.py:380-404
  * MCTS.backpropagate      self_play.py:406-430
  * Node.expand / value / add_exploration_noise   self_play.py:433-476
  * MinMaxStats             self_play.py:553-570
  * SelfPlay.select_action  self_play.py:222-245

Layout: the reference keeps a Python object graph (Node.children dict).  Here a
tree is a struct of flat lists indexed by the CANONICAL NODE INDEX (SURVEY.md
section 8c'): root = 0, the leaf expanded by simulation k (0-based) = k + 1.  A
child *slot* is (parent index, position of the action in the parent's action
list).  All statistics are Python floats (IEEE binary64), evaluated in exactly
the reference's operation order, so value sums / min-max bounds / UCB scores
are bit-identical to the reference's.

The network is abstracted behind an ``evaluator`` with two methods (the
reference calls models.py here, self_play.py:286-295 and :339-344):
    initial(observation, actions)          -> (value, reward, priors, hidden)
    recurrent(hidden, action, actions)     -> (value, reward, priors, hidden)
``value``/``reward`` are Python floats (the reference's support_to_scalar(...).item()),
``priors`` the float64 widening of the fp32 softmax over the logits of ``actions``
(self_play.py:460-462).
"""
import math

import numpy


class Tree:
    """Flat, canonical-index tree (see module docstring)."""

    __slots__ = (
        "actions", "visit", "value_sum", "reward", "to_play", "hidden",
        "prior", "child", "parent", "parent_slot", "minimum", "maximum",
        "trace", "max_depth", "root_predicted_value", "tie_draws", "margins", "_sim_margin", "value_margins", "_sim_value_margin",
    )

    def __init__(self):
        self.actions = []      # per node: list of actions (slot -> action)
        self.visit = []        # per node: int
        self.value_sum = []    # per node: float (binary64)
        self.reward = []       # per node: float (fp32-exact)
        self.to_play = []      # per node: int
        self.hidden = []       # per node: opaque
        self.prior = []        # per node: list[float] per slot
        self.child = []        # per node: list[int] per slot (node index or -1)
        self.parent = []       # per node: parent node index (-1 for root)
        self.parent_slot = []  # per node: slot within the parent
        self.minimum = float("inf")     # MinMaxStats, self_play.py:558-560
        self.maximum = -float("inf")
        self.trace = []        # per simulation: (parent index, action, depth)
        self.max_depth = 0
        self.root_predicted_value = None
        self.tie_draws = 0     # number of select_child calls with more than one maximiser
        # diagnostics (SURVEY.md section 8c'): per simulation, the smallest gap between the best and the
        # second-best UCB score over the levels of its selection walk, as (gap, depth of that level) -- how
        # close the simulation came to taking another branch under a perturbation of the network outputs
        self.margins = []
        self._sim_margin = (float("inf"), 0)
        # the same gap in units of the backed-up values r + gamma v, relative to their magnitude: a UCB score contains
        # normalize(q) = (q - min) / (max - min) (MinMaxStats, self_play.py:562-570), so a perturbation eps of a decoded value
        # moves a score by eps / (max - min) -- early in a search, when the tree-wide range is a few hundredths, a score gap
        # of 2e-3 is a value gap of 1e-4.  value_margins[k] = margins[k][0] * (max - min) / max(1, |min|, |max|) (range 1 while
        # max <= min: nothing is normalised then)
        self.value_margins = []
        self._sim_value_margin = float("inf")

    def root_visit_counts(self, action_space):
        """Visit count per action of ``action_space`` (0 for non-children)."""
        out = []
        for a in action_space:
            if a in self.actions[0]:
                c = self.child[0][self.actions[0].index(a)]
                out.append(self.visit[c] if c >= 0 else 0)
            else:
                out.append(0)
        return out

    def node_value(self, n):
        # Node.value, self_play.py:446-449
        if self.visit[n] == 0:
            return 0
        return self.value_sum[n] / self.visit[n]


def _new_node(tree, parent, parent_slot):
    tree.actions.append([])
    tree.visit.append(0)
    tree.value_sum.append(0)
    tree.reward.append(0)
    tree.to_play.append(-1)
    tree.hidden.append(None)
    tree.prior.append([])
    tree.child.append([])
    tree.parent.append(parent)
    tree.parent_slot.append(parent_slot)
    return len(tree.visit) - 1


def _expand(tree, n, actions, to_play, reward, priors, hidden):
    # Node.expand, self_play.py:451-465 (children are created lazily here: a
    # child Node in the reference holds only its prior until it is expanded).
    tree.to_play[n] = to_play
    tree.reward[n] = reward
    tree.hidden[n] = hidden
    tree.actions[n] = list(actions)
    tree.prior[n] = [float(p) for p in priors]
    tree.child[n] = [-1] * len(actions)


def _normalize(tree, value):
    # MinMaxStats.normalize, self_play.py:566-570
    if tree.maximum > tree.minimum:
        return (value - tree.minimum) / (tree.maximum - tree.minimum)
    return value


def _minmax_update(tree, value):
    # MinMaxStats.update, self_play.py:562-564
    tree.maximum = max(tree.maximum, value)
    tree.minimum = min(tree.minimum, value)


def ucb_score(tree, cfg, parent, slot):
    """self_play.py:380-404, one child slot of ``parent``."""
    n_parent = tree.visit[parent]
    c = tree.child[parent][slot]
    n_child = tree.visit[c] if c >= 0 else 0
    pb_c = math.log((n_parent + cfg.pb_c_base + 1) / cfg.pb_c_base) + cfg.pb_c_init
    pb_c *= math.sqrt(n_parent) / (n_child + 1)
    prior_score = pb_c * tree.prior[parent][slot]
    if n_child > 0:
        q = tree.node_value(c)
        value_score = _normalize(
            tree,
            tree.reward[c] + cfg.discount * (q if len(cfg.players) == 1 else -q),
        )
    else:
        value_score = 0
    return prior_score + value_score


def _select_slot(tree, cfg, parent, rng):
    # MCTS.select_child, self_play.py:363-378.  The reference evaluates every
    # score twice; evaluating once is bit-identical (pure function of the tree).
    scores = [ucb_score(tree, cfg, parent, s) for s in range(len(tree.actions[parent]))]
    best = max(scores)
    if len(scores) > 1:
        second = sorted(scores)[-2]
        if best - second < tree._sim_margin[0]:
            tree._sim_margin = (best - second, tree._sim_margin[1])
            spread = tree.maximum - tree.minimum if tree.maximum > tree.minimum else 1.0
            scale = max(1.0, abs(tree.minimum), abs(tree.maximum)) if tree.maximum > tree.minimum else 1.0
            tree._sim_value_margin = (best - second) * spread / scale
    ties = [s for s, v in enumerate(scores) if v == best]
    if len(ties) > 1:
        tree.tie_draws += 1
    # numpy.random.choice(list): no state consumed for one candidate, otherwise
    # list[randint(0, n)] on the legacy MT19937 stream.
    return ties[0] if len(ties) == 1 else int(rng.choice(ties))


def _backpropagate(tree, cfg, path, value, to_play):
    # MCTS.backpropagate, self_play.py:406-430
    n_players = len(cfg.players)
    if n_players == 1:
        for n in reversed(path):
            tree.value_sum[n] += value
            tree.visit[n] += 1
            _minmax_update(tree, tree.reward[n] + cfg.discount * tree.node_value(n))
            value = tree.reward[n] + cfg.discount * value
    elif n_players == 2:
        for n in reversed(path):
            tree.value_sum[n] += value if tree.to_play[n] == to_play else -value
            tree.visit[n] += 1
            _minmax_update(tree, tree.reward[n] + cfg.discount * -tree.node_value(n))
            value = (
                -tree.reward[n] if tree.to_play[n] == to_play else tree.reward[n]
            ) + cfg.discount * value
    else:
        raise NotImplementedError("More than two player mode not implemented.")


def run_search(cfg, evaluator, observation, legal_actions, to_play,
               add_exploration_noise, rng, num_simulations=None):
    """
    One MCTS.run (self_play.py:260-361) on the canonical-index tree.

    ``rng`` is a numpy legacy ``RandomState`` (or the ``numpy.random`` module):
    draw order = Dirichlet (:473) then one ``choice`` per tied select_child.
    Returns the finished ``Tree``.
    """
    tree = Tree()
    assert legal_actions, f"Legal actions should not be an empty array. Got {legal_actions}."
    assert set(legal_actions).issubset(set(cfg.action_space)), \
        "Legal actions should be a subset of the action space."
    root = _new_node(tree, -1, -1)
    value, reward, priors, hidden = evaluator.initial(observation, list(legal_actions))
    tree.root_predicted_value = value
    _expand(tree, root, legal_actions, to_play, reward, priors, hidden)

    if add_exploration_noise:
        # Node.add_exploration_noise, self_play.py:467-476
        noise = rng.dirichlet([cfg.root_dirichlet_alpha] * len(legal_actions))
        frac = cfg.root_exploration_fraction
        tree.prior[root] = [
            p * (1 - frac) + n * frac for p, n in zip(tree.prior[root], noise)
        ]

    sims = cfg.num_simulations if num_simulations is None else num_simulations
    for _ in range(sims):
        virtual_to_play = to_play
        node = root
        path = [node]
        depth = 0
        tree._sim_margin = (float("inf"), 0)
        tree._sim_value_margin = float("inf")
        while True:
            depth += 1
            before = tree._sim_margin[0]
            slot = _select_slot(tree, cfg, node, rng)
            if tree._sim_margin[0] < before:
                tree._sim_margin = (tree._sim_margin[0], depth)
            # players play turn by turn, self_play.py:331-334
            if virtual_to_play + 1 < len(cfg.players):
                virtual_to_play = cfg.players[virtual_to_play + 1]
            else:
                virtual_to_play = cfg.players[0]
            nxt = tree.child[node][slot]
            if nxt < 0:
                break
            node = nxt
            path.append(node)
        parent = node
        action = tree.actions[parent][slot]
        leaf = _new_node(tree, parent, slot)
        tree.child[parent][slot] = leaf
        path.append(leaf)
        value, reward, priors, hidden = evaluator.recurrent(
            tree.hidden[parent], action, list(cfg.action_space)
        )
        _expand(tree, leaf, cfg.action_space, virtual_to_play, reward, priors, hidden)
        _backpropagate(tree, cfg, path, value, virtual_to_play)
        tree.max_depth = max(tree.max_depth, depth)
        tree.trace.append((parent, int(action), depth))
        tree.margins.append(tree._sim_margin)
        tree.value_margins.append(tree._sim_value_margin)
    return tree


def select_action(visit_counts, actions, temperature, rng):
    """SelfPlay.select_action, self_play.py:222-245 (visit counts in child order)."""
    visit_counts = numpy.array(visit_counts, dtype="int32")
    if temperature == 0:
        return actions[int(numpy.argmax(visit_counts))]
    if temperature == float("inf"):
        return rng.choice(actions)
    dist = visit_counts ** (1 / temperature)
    dist = dist / sum(dist)
    return rng.choice(actions, p=dist)


def child_visit_policy(tree, action_space):
    """GameHistory.store_search_statistics, self_play.py:496-509 (the policy row)."""
    counts = tree.root_visit_counts(action_space)
    total = sum(counts)
    root_actions = tree.actions[0]
    return [counts[i] / total if a in root_actions else 0 for i, a in enumerate(action_space)]


def stacked_observations(observation_history, action_history, index,
                         num_stacked_observations, action_space_size):
    """GameHistory.get_stacked_observations, self_play.py:513-550."""
    index = index % len(observation_history)
    current = numpy.array(observation_history[index])
    planes = [current.copy()]
    for past in range(index - 1, index - num_stacked_observations - 1, -1):
        if past >= 0:
            planes.append(numpy.array(observation_history[past]))
            planes.append(
                (numpy.ones_like(current[0]) * action_history[past + 1] / action_space_size)[None]
            )
        else:
            planes.append(numpy.zeros_like(current))
            planes.append(numpy.zeros_like(current[0])[None])
    return numpy.concatenate(planes)


class ReplayEvaluator:
    """
    Lock-step evaluator: replays network outputs recorded from the reference
    (value, reward, priors per expansion, in expansion order).  Used to pin the
    tree arithmetic independently of any network arithmetic.
    """

    def __init__(self, values, rewards, priors):
        self.values, self.rewards, self.priors = values, rewards, priors
        self.k = 0

    def _next(self, actions):
        k = self.k
        self.k += 1
        p = [float(x) for x in self.priors[k][: len(actions)]]
        return float(self.values[k]), float(self.rewards[k]), p, k

    def initial(self, observation, actions):
        return self._next(actions)

    def recurrent(self, hidden, action, actions):
        return self._next(actions)
