"""
Shared driver for the lock-step parity harness (SURVEY.md section 8c'): feeds the
network outputs RECORDED from the reference run (tests/golden/tree_*.npz) into
the C ABI's lock-step entry points and returns the finished trees, so the tree
arithmetic can be compared bit for bit.  Works with any backend (HIP library on
the GPU, tests/hostcheck serial build on the CPU).
"""
import ctypes
import json
import math
import os

import numpy
import torch

from mzx import _lib, configs, self_play

from conftest import GOLDEN


def load_fixture(name):
    z = numpy.load(os.path.join(GOLDEN, f"tree_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    cfg = configs.BY_NAME[meta["game"]](**meta.get("overrides", {}))
    return z, meta, cfg


def fixture_weights(z, template):
    """Trained weights shipped inside a fixture (``flat_weights``: float tensors in state_dict key order), or None."""
    if "flat_weights" not in z.files:
        return None
    sd, off = {}, 0
    for k, t in template.items():
        if t.dtype.is_floating_point:
            sd[k] = torch.from_numpy(z["flat_weights"][off:off + t.numel()].reshape(tuple(t.shape)).copy())
            off += t.numel()
    assert off == z["flat_weights"].size
    return sd


LOCKSTEP_TAPE_WORDS = 64   # the lock-step entry points have no overflow re-run: give every case a long tape


def rng_inputs(cfg, cases, A, tape_words=LOCKSTEP_TAPE_WORDS):
    """Per-case Dirichlet noise + raw-word tape exactly as the engine derives them from a RandomState."""
    B = len(cases)
    legal = numpy.full((B, A), -1, numpy.int32)
    noise = numpy.zeros((B, A), numpy.float64)
    tape = numpy.zeros((B, tape_words), numpy.uint32)
    to_play = numpy.zeros(B, numpy.int32)
    for i, case in enumerate(cases):
        rs = numpy.random.RandomState(case["rng_seed"])
        acts = case["legal"]
        legal[i, : len(acts)] = acts
        noise[i, : len(acts)] = rs.dirichlet([cfg.root_dirichlet_alpha] * len(acts))
        tape[i] = rs.randint(0, 2 ** 32, size=tape_words, dtype=numpy.uint32)
        to_play[i] = case["to_play"]
    return legal, noise, tape, to_play


class Lockstep:
    def __init__(self, backend, cfg, B, S, hidden_net=None):
        self.be, self.lib, self.cfg, self.B, self.S = backend, backend.lib, cfg, B, S
        self.A = len(cfg.action_space)
        n = S + 1
        self._pbc = (ctypes.c_double * n)(*[math.log((k + cfg.pb_c_base + 1) / cfg.pb_c_base) + cfg.pb_c_init for k in range(n)])
        self._sqrt = (ctypes.c_double * n)(*[math.sqrt(k) for k in range(n)])
        c = _lib.SearchConfig()
        c.num_trees, c.num_simulations, c.action_space_size = B, S, self.A
        c.num_players, c.support_size, c.tape_words = len(cfg.players), cfg.support_size, LOCKSTEP_TAPE_WORDS
        c.discount, c.root_exploration_fraction = float(cfg.discount), float(cfg.root_exploration_fraction)
        c.h_pb_c_table = ctypes.cast(self._pbc, ctypes.POINTER(ctypes.c_double))
        c.h_sqrt_table = ctypes.cast(self._sqrt, ctypes.POINTER(ctypes.c_double))
        self.handle = ctypes.c_void_p()
        self.lib.check(self.lib.mzx_search_create(ctypes.byref(c), hidden_net, ctypes.byref(self.handle)))
        self.arena = backend.zeros((self.lib.mzx_search_arena_bytes(self.handle),), torch.uint8)

    def close(self):
        self.lib.mzx_search_destroy(self.handle)

    def dev(self, a, dtype):
        return torch.as_tensor(numpy.ascontiguousarray(a)).to(dtype).to(self.be.device)

    def run(self, legal, noise, tape, to_play, values, rewards, priors, check_select=None):
        """
        values/rewards: [B][S+1] float64, priors: [B][S+1][A] float64 (node order = expansion order).
        check_select(k, parent, action, leaf) is called per simulation with host arrays.
        """
        be, lib, B, S, A = self.be, self.lib, self.B, self.S, self.A
        t_legal, t_noise = self.dev(legal, torch.int32), self.dev(noise, torch.float64)
        t_tape, t_tp = self.dev(tape.view(numpy.int32), torch.int32), self.dev(to_play, torch.int32)
        out = dict(visits=be.zeros((B, A), torch.int32), root_value=be.zeros((B,), torch.float64),
                   info=be.zeros((B, 4), torch.int32))
        io = _lib.SearchIO(None, be.ptr(t_legal), be.ptr(t_tp), be.ptr(t_noise), be.ptr(t_tape),
                           be.ptr(out["visits"]), be.ptr(out["root_value"]), None, be.ptr(out["info"]))
        st = be.stream()
        arena = be.ptr(self.arena)
        t_pri = self.dev(priors, torch.float64)
        t_val, t_rew = self.dev(values, torch.float64), self.dev(rewards, torch.float64)
        root_pri = t_pri[:, 0].contiguous()
        root_rew = t_rew[:, 0].contiguous()
        lib.check(lib.mzx_search_lockstep_begin(self.handle, ctypes.byref(io), be.ptr(root_pri), be.ptr(root_rew),
                                                arena, self.arena.numel(), st))
        sel = [be.zeros((B,), torch.int32) for _ in range(3)]
        for k in range(S):
            lib.check(lib.mzx_search_lockstep_select(self.handle, ctypes.byref(io), be.ptr(sel[0]), be.ptr(sel[1]),
                                                     be.ptr(sel[2]), arena, st))
            if check_select is not None:
                check_select(k, *[t.cpu().numpy() for t in sel])
            v, r, p = t_val[:, k + 1].contiguous(), t_rew[:, k + 1].contiguous(), t_pri[:, k + 1].contiguous()
            lib.check(lib.mzx_search_lockstep_apply(self.handle, be.ptr(v), be.ptr(r), be.ptr(p), arena, st))
        lib.check(lib.mzx_search_finish(self.handle, ctypes.byref(io), arena, st))
        dump = self.dump()
        dump.update({k: v.cpu().numpy() for k, v in out.items()})
        return dump

    def dump(self):
        be, lib, B, N, A = self.be, self.lib, self.B, self.S + 1, self.A
        t = dict(
            visit=be.zeros((B, N), torch.int32), value_sum=be.zeros((B, N), torch.float64),
            reward=be.zeros((B, N), torch.float64), to_play=be.zeros((B, N), torch.int32),
            parent=be.zeros((B, N), torch.int32), child=be.zeros((B, N, A), torch.int32),
            prior=be.zeros((B, N, A), torch.float64), minmax=be.zeros((B, 2), torch.float64),
            n_nodes=be.zeros((B,), torch.int32),
        )
        d = _lib.TreeDump(*[be.ptr(t[k]) for k in ("visit", "value_sum", "reward", "to_play", "parent", "child",
                                                   "prior", "minmax", "n_nodes")])
        lib.check(lib.mzx_search_dump(self.handle, ctypes.byref(d), be.ptr(self.arena), be.stream()))
        return {k: v.cpu().numpy() for k, v in t.items()}


def run_fixture(backend, name):
    """Replay every case of a tree fixture through the lock-step ABI; assert bit-exact parity."""
    z, meta, cfg = load_fixture(name)
    cases = meta["cases"]
    B, S, A = len(cases), meta["num_simulations"], len(cfg.action_space)
    legal, noise, tape, to_play = rng_inputs(cfg, cases, A)
    values = numpy.stack([z[f"c{c}_net_value"] for c in range(B)])
    rewards = numpy.stack([z[f"c{c}_net_reward"] for c in range(B)])
    priors = numpy.stack([z[f"c{c}_net_priors"] for c in range(B)])
    ls = Lockstep(backend, cfg, B, S)

    def check_select(k, parent, action, leaf):
        for c in range(B):
            assert parent[c] == z[f"c{c}_parent"][k + 1], (name, c, k)
            assert action[c] == z[f"c{c}_parent_action"][k + 1], (name, c, k)
            assert leaf[c] == k + 1

    got = ls.run(legal, noise, tape, to_play, values, rewards, priors, check_select)
    ls.close()
    bits = lambda a: numpy.ascontiguousarray(a, dtype=numpy.float64).view(numpy.int64)
    for c, case in enumerate(cases):
        g = lambda k: z[f"c{c}_{k}"]
        n = g("visit").shape[0]
        assert got["n_nodes"][c] == n
        assert numpy.array_equal(got["visit"][c, :n], g("visit"))
        assert numpy.array_equal(bits(got["value_sum"][c, :n]), bits(g("value_sum")))
        assert numpy.array_equal(bits(got["reward"][c, :n]), bits(g("reward")))
        assert numpy.array_equal(got["to_play"][c, :n], g("to_play"))
        assert numpy.array_equal(got["parent"][c, :n], g("parent"))
        assert numpy.array_equal(bits(got["minmax"][c]), bits(g("minmax")))
        for i in range(n):
            k = int(g("n_children")[i])
            assert numpy.array_equal(got["child"][c, i, :k], g("child")[i, :k])
            assert numpy.array_equal(bits(got["prior"][c, i, :k]), bits(g("prior")[i, :k]))
        # root visit counts by ACTION, root value, max depth
        want = numpy.zeros(A, numpy.int32)
        for s, a in enumerate(case["legal"]):
            ch = g("child")[0, s]
            want[a] = g("visit")[ch] if ch >= 0 else 0
        assert numpy.array_equal(got["visits"][c], want)
        assert got["root_value"][c] == g("value_sum")[0] / g("visit")[0]
        assert got["info"][c, 0] == int(g("max_tree_depth"))
        assert got["info"][c, 1] == 0
    return got
