"""
The batched search behind the C ABI (split out of mzx/self_play.py in round 6; ``mzx.self_play`` re-exports every name).

``BatchedMCTS`` searches B roots per call -- MCTS.run (/root/reference/self_play.py:260-361) for a shard of games: staging
of a move's inputs and outputs, root noise / tie-break tape from the per-game streams, the library call, tape retries,
the searched trees as reference ``Node`` graphs (``node_graph``), roots the caller expanded itself (``run_from_roots``:
override_root_with, :275-277).  ``MCTS`` is the reference's single-root class on top of it (same signature and return value).
"""
import ctypes
import math

import numpy
import torch

from . import _lib
from .history import Node

# raw MT19937 words per tree handed to a search for its tie draws: a search normally consumes ONE (the all-zero
# scores of its first walk); a tree that needs more is searched again with a longer tape (BatchedMCTS.run)
TAPE_WORDS = 16


class SearchResult:
    """Per-root outputs of one batched search (host numpy arrays)."""

    def __init__(self, visit_counts, root_values, root_predicted_values, info, legal_actions):
        self.visit_counts = visit_counts                  # [B][A] int32, by action
        self.root_values = root_values                    # [B] float64, root.value()
        self.root_predicted_values = root_predicted_values  # [B] float64
        self.max_tree_depth = info[:, 0]
        self.flags = info[:, 1]
        self.tape_used = info[:, 2]
        self.sum_depth = info[:, 3]
        self.legal_actions = legal_actions
        self.shared_legal = None     # set when every root has the same legal-action list (list protocol)
        self.legal_array = self.n_legal = self.streams = None    # bank searches: [B][A] padded lists, their lengths, the streams
        self.pending_words = None    # tie-break words the streams have still to consume (run(..., _defer_advance=True))

    def root(self, i):
        """A ``Node`` whose children carry the visit counts of root i (self_play.py:222-245, :496-511)."""
        node = Node(0)
        total = int(self.visit_counts[i].sum())
        node.visit_count = total
        node.value_sum = float(self.root_values[i]) * total
        for a in self.legal_actions[i]:
            child = Node(0)
            child.visit_count = int(self.visit_counts[i][a])
            node.children[a] = child
        node._root_value = float(self.root_values[i])
        node.value = lambda: node._root_value if total else 0
        return node



class PendingSearch:
    """A search that may still be running on the device (``BatchedMCTS.run(..., _asynchronous=True)``)."""

    def __init__(self, complete):
        self._complete, self._result = complete, None

    def result(self):
        if self._complete is not None:
            self._result, self._complete = self._complete(), None
        return self._result


def _validate(config):
    A = len(config.action_space)
    if list(config.action_space) != list(range(A)):
        raise NotImplementedError("config.action_space must be list(range(n)) (the game files only edit its length)")
    P = len(config.players)
    if list(config.players) != list(range(P)):
        raise NotImplementedError("config.players must be list(range(n))")
    if P > 2:
        raise NotImplementedError("More than two player mode not implemented.")  # self_play.py:429-430


class BatchedMCTS:
    """
    MCTS.run (self_play.py:260-361) for B roots at once.  One instance owns the
    device arena for up to ``max_trees`` roots of a given network.
    """

    fused_move = True     # A/B switch of the tests: False = the separate calls (root_draws, _launch, advance, numpy select)

    def __init__(self, config, model, max_trees, num_simulations=None, mode=None):
        _validate(config)
        self.config = config
        self.model = model
        self.backend = model.backend
        self.max_trees = int(max_trees)
        self.num_simulations = int(config.num_simulations if num_simulations is None else num_simulations)
        self.A = len(config.action_space)
        n = self.num_simulations + 1
        # host tables with Python's own math (the reference's exact values, self_play.py:384-391)
        self._pbc = (ctypes.c_double * n)(
            *[math.log((k + config.pb_c_base + 1) / config.pb_c_base) + config.pb_c_init for k in range(n)]
        )
        self._sqrt = (ctypes.c_double * n)(*[math.sqrt(k) for k in range(n)])
        self._handles = {}
        self._arena = None
        self._mode = mode
        self._buffers = {}

    def __del__(self):
        try:
            for h in self._handles.values():
                self.backend.lib.mzx_search_destroy(h)
            self._handles = {}
        except Exception:
            pass

    def handle(self, num_trees, tape_words=TAPE_WORDS):
        key = num_trees if tape_words == TAPE_WORDS else (num_trees, tape_words)
        if key not in self._handles:
            lib = self.backend.lib
            c = _lib.SearchConfig()
            c.num_trees = num_trees
            c.num_simulations = self.num_simulations
            c.action_space_size = self.A
            c.num_players = len(self.config.players)
            c.support_size = self.config.support_size
            c.tape_words = tape_words
            c.discount = float(self.config.discount)
            c.root_exploration_fraction = float(self.config.root_exploration_fraction)
            c.h_pb_c_table = ctypes.cast(self._pbc, ctypes.POINTER(ctypes.c_double))
            c.h_sqrt_table = ctypes.cast(self._sqrt, ctypes.POINTER(ctypes.c_double))
            h = ctypes.c_void_p()
            net = self.model.handle if self.model is not None else None
            lib.check(lib.mzx_search_create(ctypes.byref(c), net, ctypes.byref(h)))
            if self._mode is not None:
                lib.check(lib.mzx_search_set_mode(h, int(self._mode)))
            self._handles[key] = h
        return self._handles[key]

    def kernel_name(self, num_trees):
        """The search kernel the last ``run`` of this shard size launched (``mzx_search_kernel_name``)."""
        name = self.backend.lib.mzx_search_kernel_name(self.handle(num_trees))
        return name.decode() if name else ""

    def arena(self, num_trees):
        need = self.backend.lib.mzx_search_arena_bytes(self.handle(num_trees))
        if self._arena is None or self._arena.numel() < need:
            big = self.backend.lib.mzx_search_arena_bytes(self.handle(self.max_trees)) if num_trees <= self.max_trees else need
            self._arena = self.backend.zeros((max(need, big),), torch.uint8)
        return self._arena

    def export_trees(self, num_trees):
        """
        Canonical-order copy of the trees of the last ``run`` (diagnose tooling / parity tests):
        dict of numpy arrays visit [B][N], value_sum, reward, to_play, parent, child [B][N][A],
        prior, minmax [B][2], n_nodes [B].  With the fused kernel this needs mode flag 2.
        """
        be, lib, B, N, A = self.backend, self.backend.lib, num_trees, self.num_simulations + 1, self.A
        t = dict(
            visit=be.zeros((B, N), torch.int32), value_sum=be.zeros((B, N), torch.float64),
            reward=be.zeros((B, N), torch.float64), to_play=be.zeros((B, N), torch.int32),
            parent=be.zeros((B, N), torch.int32), child=be.zeros((B, N, A), torch.int32),
            prior=be.zeros((B, N, A), torch.float64), minmax=be.zeros((B, 2), torch.float64),
            n_nodes=be.zeros((B,), torch.int32),
        )
        d = _lib.TreeDump(*[be.ptr(t[k]) for k in ("visit", "value_sum", "reward", "to_play", "parent", "child",
                                                   "prior", "minmax", "n_nodes")])
        lib.check(lib.mzx_search_dump(self.handle(B), ctypes.byref(d), be.ptr(self.arena(B)), be.stream()))
        return {k: v.cpu().numpy() for k, v in t.items()}

    def arena_offsets(self, num_trees):
        out = (ctypes.c_int64 * 8)()
        self.backend.lib.check(self.backend.lib.mzx_search_arena_offsets(self.handle(num_trees), ctypes.byref(out)))
        return dict(zip(("tables", "trees", "hidden", "workspace", "tree_bytes", "workspace_bytes", "total"), out))

    def _buf(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        if key not in self._buffers:
            self._buffers[key] = self.backend.empty(shape, dtype)
        return self._buffers[key]

    def make_io(self, B, observations, legal, to_play, noise, tape):
        """Upload one move's inputs; returns (io struct, output tensors)."""
        be = self.backend
        dev = lambda a, dt: torch.as_tensor(a).to(dt).contiguous().to(be.device, non_blocking=True)
        t_obs = dev(observations, torch.float32)
        t_legal = dev(legal, torch.int32)
        t_tp = dev(to_play, torch.int32)
        t_noise = None if noise is None else dev(noise, torch.float64)
        t_tape = dev(tape.view(numpy.int32) if isinstance(tape, numpy.ndarray) else tape, torch.int32)
        out = dict(
            visits=self._buf("visits", (B, self.A), torch.int32), root_value=self._buf("rv", (B,), torch.float64),
            predicted=self._buf("rpv", (B,), torch.float64), info=self._buf("info", (B, 4), torch.int32),
        )
        io = _lib.SearchIO(be.ptr(t_obs), be.ptr(t_legal), be.ptr(t_tp), be.ptr(t_noise), be.ptr(t_tape),
                           be.ptr(out["visits"]), be.ptr(out["root_value"]), be.ptr(out["predicted"]),
                           be.ptr(out["info"]))
        keep = (t_obs, t_legal, t_tp, t_noise, t_tape)
        return io, out, keep

    def run_from_roots(self, roots, to_play, add_exploration_noise, rngs):
        """
        MCTS.run(..., override_root_with=root) (self_play.py:275-277) for B roots the caller expanded itself
        (``Node.expand`` after a ``recurrent_inference``, diagnose_model.py:57-74): the roots' hidden states,
        child priors and rewards replace initial_inference.  Roots that already carry visits are not supported.
        """
        legal, priors, rewards, hidden = [], numpy.zeros((len(roots), self.A), numpy.float64), [], []
        for i, root in enumerate(roots):
            if not root.expanded() or root.hidden_state is None:
                raise ValueError("override_root_with must be an expanded Node with a hidden_state")
            if root.visit_count or any(c.visit_count or c.expanded() for c in root.children.values()):
                raise NotImplementedError("override_root_with: only freshly expanded roots (no visits yet) are supported")
            if root.to_play != to_play[i]:
                raise NotImplementedError("override_root_with: root.to_play must equal the to_play argument")
            acts = list(root.children.keys())
            legal.append(acts)
            priors[i, : len(acts)] = [root.children[a].prior for a in acts]
            rewards.append(float(root.reward))
            hidden.append(root.hidden_state.detach().reshape(1, -1))
        override = dict(
            hidden=torch.cat(hidden).to(self.backend.device, torch.float32).contiguous(),
            priors=torch.as_tensor(priors).to(self.backend.device),
            reward=torch.as_tensor(numpy.asarray(rewards, numpy.float64)).to(self.backend.device),
        )
        if override["hidden"].shape[1] != self.model.hidden_size:
            raise ValueError("override_root_with: hidden_state does not match the network's encoded state")
        return self.run(None, legal, to_play, add_exploration_noise, rngs, _override=override)

    def node_graph(self, num_trees, i, root_actions):
        """
        Tree i of the last ``run`` as reference ``Node`` objects (self_play.py:433-476): children dicts keyed
        by action, visit_count / value_sum / prior / reward / to_play / hidden_state ([1, *hidden_shape]
        device tensor) per node -- what diagnose_model.py:145-192 walks.  Needs the per-operator engine
        (mode 0), whose arena holds every node's hidden state in canonical order.
        """
        t = self.export_trees(num_trees)
        off = self.arena_offsets(num_trees)
        N, A, Hf = self.num_simulations + 1, self.A, self.model.hidden_size
        hid = self.arena(num_trees)[off["hidden"]: off["hidden"] + num_trees * N * Hf * 4].view(torch.float32)
        hid = hid.view(num_trees, N, Hf)[i].clone()
        n_nodes = int(t["n_nodes"][i])
        nodes = [None] * n_nodes

        def build(n, prior):
            node = Node(prior)
            node.visit_count = int(t["visit"][i, n])
            node.value_sum = float(t["value_sum"][i, n])
            node.to_play = int(t["to_play"][i, n])
            node.reward = float(t["reward"][i, n])
            node.hidden_state = hid[n].view((1,) + tuple(self.model.hidden_shape))
            nodes[n] = node
            return node

        root = build(0, 0)
        order = [0]
        while order:
            n = order.pop()
            actions = list(root_actions) if n == 0 else list(self.config.action_space)
            for slot, a in enumerate(actions):
                c = int(t["child"][i, n, slot])
                prior = float(t["prior"][i, n, slot])
                if c >= 0:
                    nodes[n].children[a] = build(c, prior)
                    order.append(c)
                else:
                    nodes[n].children[a] = Node(prior)
        return root

    def _staging(self, B, tape_words, obs_floats, with_noise):
        """
        Persistent I/O of one (B, tape) geometry: ONE pinned host block + ONE device block for a move's inputs
        (observations | noise | legal | to_play | tape; 8-byte fields first) and one of each for its outputs
        (root value | predicted root value | visit counts | info) -- a move costs one upload, one launch, one
        download and one stream synchronisation instead of five uploads and four blocking downloads.
        """
        key = ("staging", B, tape_words, obs_floats, bool(with_noise))
        st = self._buffers.get(key)
        if st is not None:
            return st
        A, be = self.A, self.backend
        on_gpu = be.device.type == "cuda"

        def carve(fields):
            off, table = 0, {}
            for name, dtype, count in fields:
                nbytes = numpy.dtype(dtype).itemsize * count
                table[name] = (off, dtype, count)
                off += (nbytes + 15) & ~15
            return table, max(off, 16)

        fin, n_in = carve([("noise", numpy.float64, B * A if with_noise else 0), ("obs", numpy.float32, B * obs_floats),
                           ("legal", numpy.int32, B * A), ("to_play", numpy.int32, B), ("tape", numpy.uint32, B * tape_words)])
        fout, n_out = carve([("root_value", numpy.float64, B), ("predicted", numpy.float64, B),
                             ("visits", numpy.int32, B * A), ("info", numpy.int32, B * 4)])
        h_in = torch.empty(n_in, dtype=torch.uint8, pin_memory=on_gpu)
        h_out = torch.empty(n_out, dtype=torch.uint8, pin_memory=on_gpu)
        d_in = be.empty((n_in,), torch.uint8) if on_gpu else h_in
        d_out = be.zeros((n_out,), torch.uint8) if on_gpu else h_out
        views = lambda block, table: {k: block.numpy()[o:o + numpy.dtype(dt).itemsize * c].view(dt)
                                      for k, (o, dt, c) in table.items()}
        st = dict(h_in=h_in, h_out=h_out, d_in=d_in, d_out=d_out, vin=views(h_in, fin), vout=views(h_out, fout),
                  pin={k: d_in.data_ptr() + o for k, (o, _, _) in fin.items()},
                  pout={k: d_out.data_ptr() + o for k, (o, _, _) in fout.items()}, on_gpu=on_gpu)
        self._buffers[key] = st
        return st

    def _launch(self, B, obs, legal, to_play, noise, tape, tape_words, override):
        """One mzx_search_run / mzx_search_run_from_roots over B roots; host copies of the outputs."""
        lib, A = self.backend.lib, self.A
        obs_dev = obs if isinstance(obs, torch.Tensor) else None      # already stacked on the device (FrameStore)
        obs_floats = 0 if obs_dev is not None else int(obs.size // B)
        st = self._staging(B, tape_words, obs_floats, noise is not None)
        vin = st["vin"]
        if obs_dev is None:
            vin["obs"][:] = obs.reshape(-1)
        if noise is not None:
            vin["noise"][:] = noise.reshape(-1)
        vin["legal"][:] = legal.reshape(-1)
        vin["to_play"][:] = to_play
        vin["tape"][:] = tape.reshape(-1).view(numpy.uint32)
        if st["on_gpu"]:
            st["d_in"].copy_(st["h_in"], non_blocking=True)
        c_vp, pin, pout = ctypes.c_void_p, st["pin"], st["pout"]
        io = _lib.SearchIO(c_vp(obs_dev.data_ptr()) if obs_dev is not None else c_vp(pin["obs"]), c_vp(pin["legal"]),
                           c_vp(pin["to_play"]), c_vp(pin["noise"]) if noise is not None else c_vp(0), c_vp(pin["tape"]),
                           c_vp(pout["visits"]), c_vp(pout["root_value"]), c_vp(pout["predicted"]), c_vp(pout["info"]))
        arena = self.arena(B)
        handle = self.handle(B, tape_words)
        if override is not None:
            ptr = self.backend.ptr
            lib.check(lib.mzx_search_run_from_roots(handle, ctypes.byref(io), ptr(override["hidden"]),
                                                    ptr(override["priors"]), ptr(override["reward"]), ptr(arena),
                                                    arena.numel(), self.backend.stream()))
        else:
            lib.check(lib.mzx_search_run(handle, ctypes.byref(io), self.backend.ptr(arena), arena.numel(),
                                         self.backend.stream()))
        if st["on_gpu"]:
            st["h_out"].copy_(st["d_out"], non_blocking=True)
            torch.cuda.current_stream(self.backend.device).synchronize()
        vout = st["vout"]
        return (vout["visits"].reshape(B, A).copy(), vout["root_value"].copy(), vout["predicted"].copy(),
                vout["info"].reshape(B, 4).copy())

    def _move_search(self, B, obs, legal, to_play, bank, bank_idx, with_noise, asynchronous=False):
        """
        ``mzx_selfplay_search``: root draws of the bank straight into the pinned staging block, ONE upload, the search,
        ONE download, the stream synchronisation -- the statements of ``StreamBank.root_draws`` + ``_launch`` behind one
        call.  Returns a function that hands out host copies of the outputs, n_legal [B] and the noise the roots got
        ([B][A] view, or None).  ``asynchronous``: the call returns once the download is queued (MZX_MOVE_NO_SYNC) and
        the returned function first waits for an event recorded behind it -- the host is free in between (the staging
        block of this (B, tape) geometry is in use until then: one search in flight per engine).
        """
        lib, A = self.backend.lib, self.A
        obs_dev = obs if isinstance(obs, torch.Tensor) else None
        obs_floats = 0 if obs_dev is not None else int(obs.size // B)
        st = self._staging(B, TAPE_WORDS, obs_floats, with_noise)
        pin, pout = st["pin"], st["pout"]
        mv = st.get("move")
        if mv is None:
            c_vp = ctypes.c_void_p
            mv = st["move"] = _lib.Move()
            mv.num_games, mv.action_space_size, mv.tape_words = B, A, TAPE_WORDS
            mv.h_in, mv.d_in, mv.in_bytes = st["h_in"].data_ptr(), st["d_in"].data_ptr(), st["h_in"].numel()
            mv.h_out, mv.d_out, mv.out_bytes = st["h_out"].data_ptr(), st["d_out"].data_ptr(), st["h_out"].numel()
            mv.io = _lib.SearchIO(c_vp(pin["obs"]), c_vp(pin["legal"]), c_vp(pin["to_play"]),
                                  c_vp(pin["noise"]) if with_noise else c_vp(0), c_vp(pin["tape"]), c_vp(pout["visits"]),
                                  c_vp(pout["root_value"]), c_vp(pout["predicted"]), c_vp(pout["info"]))
        asynchronous = bool(asynchronous) and st["on_gpu"]
        mv.num_threads = bank.threads
        mv.flags = _lib.MOVE_NO_SYNC if asynchronous else 0
        mv.streams, mv.legal_actions, mv.to_play = bank_idx.ctypes.data, legal.ctypes.data, to_play.ctypes.data
        mv.dirichlet_alpha, mv.add_exploration_noise = float(self.config.root_dirichlet_alpha), int(with_noise)
        if obs_dev is not None:
            mv.observation, mv.observation_floats = None, 0
            mv.io.d_observation = obs_dev.data_ptr()
        else:
            mv.observation, mv.observation_floats = obs.ctypes.data, obs_floats
        n_legal = numpy.empty(B, numpy.int32)
        arena = self.arena(B)
        rc = lib.mzx_selfplay_search(self.handle(B, TAPE_WORDS), bank.handle, ctypes.byref(mv), n_legal.ctypes.data,
                                     self.backend.ptr(arena), arena.numel(), self.backend.stream())
        if rc != 0:
            message = lib.mzx_last_error().decode()
            if message.startswith("Legal actions"):       # self_play.py:296-301 raise AssertionError
                raise AssertionError(message)
            lib.check(rc)
        done = None
        if asynchronous:
            done = st.get("event")
            if done is None:
                done = st["event"] = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.backend.device))
        keep = (obs, legal, to_play, bank_idx)      # the library read them during the call; the caller may reuse them after

        def outputs():
            if done is not None:
                done.synchronize()
            vout = st["vout"]
            return (vout["visits"].reshape(B, A).copy(), vout["root_value"].copy(), vout["predicted"].copy(),
                    vout["info"].reshape(B, 4).copy(), n_legal, st["vin"]["noise"].reshape(B, A) if with_noise else None)
        outputs.keep = keep
        return outputs

    def run(self, observations, legal_actions, to_play, add_exploration_noise, rngs, _override=None, _defer_advance=False,
            _asynchronous=False):
        """
        observations: B stacked observations; legal_actions: B lists; to_play: B ints;
        rngs: B numpy RandomState-like objects (dirichlet / randint / get_state / set_state), or a pair
        ``(StreamBank, stream indices)`` -- the native bank serves all B games, and the whole host side of the
        move (draws, staging, upload, search, download) is ONE call into the library (``mzx_selfplay_search``).
        ``_defer_advance``: leave the consumption of the tie-break words to ``SelfPlay._select_actions_bank``
        (``mzx_selfplay_select`` does it in the same pass as the action draw); ``result.pending_words`` holds them.
        ``_asynchronous`` (bank searches): returns a ``PendingSearch`` as soon as the search is queued on the device;
        its ``result()`` waits and completes the call -- the host steps another group of games in between.
        """
        cfg, A = self.config, self.A
        B = len(legal_actions)
        bank = None
        if isinstance(rngs, tuple):
            bank, bank_idx = rngs
            bank_idx = numpy.ascontiguousarray(bank_idx, dtype=numpy.int32)
            assert bank_idx.size == B
        else:
            assert len(rngs) == B
        assert len(to_play) == B
        legal = numpy.full((B, A), -1, numpy.int32)
        n_legal = numpy.empty(B, numpy.int32)
        action_set = set(cfg.action_space)
        fused = self.fused_move and bank is not None and _override is None     # the move's host side behind one library call
        if isinstance(legal_actions, numpy.ndarray):   # batched protocol: [B][A] int32, lists padded with -1
            legal = numpy.ascontiguousarray(legal_actions, dtype=numpy.int32)
            assert legal.shape == (B, A), "legal_actions array must be [num_trees][len(action_space)]"
            if not fused:                              # (mzx_selfplay_search validates the rows itself)
                n_legal = (legal >= 0).sum(1).astype(numpy.int32)
                assert (n_legal > 0).all(), "Legal actions should not be an empty array."
                assert (legal < A).all() and ((legal >= 0) == (numpy.arange(A)[None, :] < n_legal[:, None])).all(), \
                    "Legal actions should be a subset of the action space (padded with -1 at the end)."
            legal_actions = legal
        if not isinstance(legal_actions, numpy.ndarray) and B > 1 and legal_actions.count(legal_actions[0]) == B:
            # every game offers the same list (the common case): validate it once (self_play.py:296-301)
            acts = legal_actions[0]
            assert acts, f"Legal actions should not be an empty array. Got {acts}."
            assert set(acts).issubset(action_set), "Legal actions should be a subset of the action space."
            assert len(set(acts)) == len(acts), "Legal actions must not repeat."
            legal[:, : len(acts)] = acts
            n_legal[:] = len(acts)
            shared = [list(acts)] * B
        else:
            shared = None
        for i, acts in enumerate(legal_actions if not isinstance(legal_actions, numpy.ndarray) and shared is None else ()):
            # self_play.py:296-301
            assert acts, f"Legal actions should not be an empty array. Got {acts}."
            assert set(acts).issubset(action_set), "Legal actions should be a subset of the action space."
            assert len(set(acts)) == len(acts), "Legal actions must not repeat."
            legal[i, : len(acts)] = acts
            n_legal[i] = len(acts)
        if fused:
            noise = tape = None
        elif bank is not None:
            noise, tape = bank.root_draws(bank_idx, cfg.root_dirichlet_alpha, n_legal, A, TAPE_WORDS,
                                          with_noise=bool(add_exploration_noise))
        else:
            noise = numpy.zeros((B, A), numpy.float64) if add_exploration_noise else None
            tape = numpy.zeros((B, TAPE_WORDS), numpy.uint32)
            states = []
            for i in range(B):
                if add_exploration_noise:
                    noise[i, : n_legal[i]] = rngs[i].dirichlet([cfg.root_dirichlet_alpha] * int(n_legal[i]))
                states.append(rngs[i].get_state())
                tape[i] = rngs[i].randint(0, 2 ** 32, size=TAPE_WORDS, dtype=numpy.uint32)
        if _override is not None:
            obs = numpy.zeros((B, 1), numpy.float32)    # not read: the roots are given
        elif isinstance(observations, torch.Tensor):   # already stacked on the device (mzx.observations.FrameStore)
            assert observations.shape[0] == B
            obs = observations.reshape(B, -1)
        else:
            obs = numpy.ascontiguousarray(numpy.asarray(observations, dtype=numpy.float32).reshape(B, -1))
        to_play = numpy.ascontiguousarray(to_play, dtype=numpy.int32)
        outputs = None
        if fused:
            outputs = self._move_search(B, obs, legal, to_play, bank, bank_idx, bool(add_exploration_noise), _asynchronous)
        pending = PendingSearch(lambda: self._complete_run(
            outputs, B, obs, legal, legal_actions, shared, to_play, n_legal, noise, tape, bank, bank_idx if bank is not None else None,
            rngs, states if bank is None else None, _override, _defer_advance))
        return pending if _asynchronous else pending.result()

    def _complete_run(self, outputs, B, obs, legal, legal_actions, shared, to_play, n_legal, noise, tape, bank, bank_idx, rngs,
                      states, _override, _defer_advance):
        """The second half of ``run``: outputs of the (possibly still running) search, tape retries, the result record."""
        cfg, A = self.config, self.A
        if outputs is not None:
            visits, root_values, predicted, info, n_legal, noise = outputs()
        else:
            visits, root_values, predicted, info = self._launch(B, obs, legal, to_play, noise, tape, TAPE_WORDS, _override)
        # A tree that exhausted its tie-break tape (a network with equal priors ties at every level) is searched
        # again with a longer tape: same roots, same noise, the stream peeked further -- the reference never
        # fails here (numpy.random.choice at self_play.py:371 simply keeps drawing).
        words = TAPE_WORDS
        while (info[:, 1] & 1).any():
            words *= 8
            if words > (1 << 22):
                raise _lib.MzxError("tie-break tape: a search consumed more than 4M random words")
            redo = numpy.nonzero(info[:, 1] & 1)[0]
            if bank is not None:
                _, long_tape = bank.root_draws(bank_idx[redo], cfg.root_dirichlet_alpha, n_legal[redo], A, words,
                                               with_noise=False)
            else:
                long_tape = numpy.zeros((len(redo), words), numpy.uint32)
                for k, i in enumerate(redo):
                    rngs[i].set_state(states[i])
                    long_tape[k] = rngs[i].randint(0, 2 ** 32, size=words, dtype=numpy.uint32)
            ov = None if _override is None else {k: v[torch.as_tensor(redo, device=v.device)] for k, v in _override.items()}
            sub_obs = obs[torch.as_tensor(redo, device=obs.device)] if isinstance(obs, torch.Tensor) else obs[redo]
            v2, r2, p2, i2 = self._launch(len(redo), sub_obs, legal[redo], to_play[redo],
                                          None if noise is None else noise[redo], long_tape, words, ov)
            visits[redo], root_values[redo], predicted[redo], info[redo] = v2, r2, p2, i2
        result = SearchResult(visits, root_values, predicted, info,
                              legal_actions if isinstance(legal_actions, numpy.ndarray) else
                              (shared if shared is not None else [list(a) for a in legal_actions]))
        if shared is not None:
            result.shared_legal = shared[0]
        if (info[:, 1] != 0).any():
            raise _lib.MzxError(f"search flagged trees {numpy.nonzero(info[:, 1])[0][:8]} (flags {set(info[:, 1])}): "
                                "node arena exhausted")
        if bank is not None:
            result.legal_array, result.n_legal, result.streams = legal, n_legal, bank_idx
            if _defer_advance and self.fused_move:     # (the numpy action draw of the A/B switch does not advance)
                result.pending_words = numpy.ascontiguousarray(info[:, 2], dtype=numpy.int32)
            else:
                bank.advance(bank_idx, info[:, 2])   # consume exactly what the device consumed
        else:
            for i in range(B):  # rewind, then consume exactly what the device consumed
                rngs[i].set_state(states[i])
                if info[i, 2]:
                    rngs[i].randint(0, 2 ** 32, size=int(info[i, 2]), dtype=numpy.uint32)
        return result


class MCTS:
    """self_play.py:249-361 -- single-root form, same signature and return value."""

    def __init__(self, config):
        self.config = config

    def run(self, model, observation, legal_actions, to_play, add_exploration_noise, override_root_with=None):
        # the single-root facade returns the whole searched tree as Node objects, like the reference; it uses
        # the per-operator engine, whose arena keeps every node's hidden state
        engine = getattr(model, "_mcts_engine", None)
        if engine is None or engine.config is not self.config:
            engine = BatchedMCTS(self.config, model, 1, mode=0)
            model._mcts_engine = engine
        rng = [numpy.random.mtrand._rand]
        if override_root_with:
            res = engine.run_from_roots([override_root_with], [to_play], add_exploration_noise, rng)
            searched = engine.node_graph(1, 0, res.legal_actions[0])
            searched.hidden_state = override_root_with.hidden_state
            override_root_with.__dict__.update(searched.__dict__)   # the reference searches the given object in place
            root, predicted = override_root_with, None
        else:
            res = engine.run([observation], [list(legal_actions)], [to_play], add_exploration_noise, rng)
            root, predicted = engine.node_graph(1, 0, res.legal_actions[0]), float(res.root_predicted_values[0])
        extra_info = {
            "max_tree_depth": int(res.max_tree_depth[0]),
            "root_predicted_value": predicted,
        }
        return root, extra_info
