"""
Host-side mirror of the reference's network surface for the self-play path.

Same names, arguments and return conventions as /root/reference/models.py:
  MuZeroNetwork(config)                factory, models.py:7-41
  net.initial_inference(observation)   models.py:172-190 / :601-618
  net.recurrent_inference(state, act)  models.py:192-195 / :620-623
  net.get_weights() / set_weights()    models.py:69-73 (reference state_dict keys,
                                       including DataParallel's ".module." infix)
  dict_to_cpu, support_to_scalar       models.py:44-53, :645-666
All arithmetic runs in the gfx950 kernels behind include/mzx.h; this module only
owns device buffers (torch tensors) and the flat weight buffer that RCCL
broadcasts.  Training-only pieces (scalar_to_support, autograd) are out of scope.
"""
import collections
import ctypes
import math

import torch

from . import _lib


def dict_to_cpu(dictionary):
    """models.py:44-53"""
    cpu_dict = {}
    for key, value in dictionary.items():
        if isinstance(value, torch.Tensor):
            cpu_dict[key] = value.cpu()
        elif isinstance(value, dict):
            cpu_dict[key] = dict_to_cpu(value)
        else:
            cpu_dict[key] = value
    return cpu_dict


def support_to_scalar(logits, support_size, _backend=None):
    """
    models.py:645-666.  Device tensors are decoded by the library (``mzx_support_to_scalar``, the
    search kernels' own decode); host tensors take the torch expression below (diagnostic use).
    """
    if logits.is_cuda or _backend is not None:
        be = _backend if _backend is not None else _lib.default_backend()
        x = logits.to(be.device, torch.float32).contiguous()
        if x.dim() != 2 or x.shape[1] != 2 * support_size + 1:
            raise ValueError(f"expected [rows, {2 * support_size + 1}] logits, got {tuple(x.shape)}")
        out = be.empty((x.shape[0], 1), torch.float32)
        be.lib.check(be.lib.mzx_support_to_scalar(be.ptr(x), x.shape[0], int(support_size), be.ptr(out), be.stream()))
        return out
    probabilities = torch.softmax(logits, dim=1)
    support = torch.arange(-support_size, support_size + 1, device=logits.device).float().expand(probabilities.shape)
    x = torch.sum(support * probabilities, dim=1, keepdim=True)
    return torch.sign(x) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(x) + 1 + 0.001)) - 1) / (2 * 0.001)) ** 2 - 1)


def _layers(values):
    values = list(values)
    if len(values) > _lib.MZX_MAX_LAYERS:
        raise NotImplementedError(f"at most {_lib.MZX_MAX_LAYERS} hidden layers per MLP are supported")
    arr = (ctypes.c_int32 * _lib.MZX_MAX_LAYERS)(*values)
    return len(values), arr


def net_config_from(config):
    """MuZeroConfig attributes read by the factory (models.py:7-41) -> mzx_net_config."""
    c = _lib.NetConfig()
    if config.network == "fullyconnected":
        c.network = 0
    elif config.network == "resnet":
        c.network = 1
    else:
        raise NotImplementedError('The network parameter should be "fullyconnected" or "resnet".')
    c.observation_shape = (ctypes.c_int32 * 3)(*[int(v) for v in config.observation_shape])
    c.stacked_observations = int(config.stacked_observations)
    c.action_space_size = len(config.action_space)
    c.support_size = int(config.support_size)
    c.encoding_size = int(config.encoding_size)
    for name in ("fc_representation_layers", "fc_dynamics_layers", "fc_reward_layers", "fc_value_layers",
                 "fc_policy_layers", "resnet_fc_reward_layers", "resnet_fc_value_layers", "resnet_fc_policy_layers"):
        n, arr = _layers(getattr(config, name))
        setattr(c, "n_" + name, n)
        setattr(c, name, arr)
    if config.downsample in (False, None, 0):
        c.downsample = 0
    elif config.downsample == "resnet":
        c.downsample = 1
    elif config.downsample == "CNN":
        c.downsample = 2
    else:
        raise NotImplementedError('downsample should be "resnet" or "CNN".')  # models.py:327
    c.blocks = int(config.blocks)
    c.channels = int(config.channels)
    c.reduced_channels_reward = int(config.reduced_channels_reward)
    c.reduced_channels_value = int(config.reduced_channels_value)
    c.reduced_channels_policy = int(config.reduced_channels_policy)
    return c


STREAMED_SHAPE_FIELDS = ("T", "rows", "mtiles", "lds", "ntiles_wg", "nsplit", "NT", "WN", "WM", "MT", "groups", "cpg",
                         "phases", "Cs", "ntiles", "cchunks")


def streamed_k_loop(MT, NT, tower=False):
    """Which K loop ``rb_gemm_kernel<MT, NT>`` / ``rb_tower_kernel<MT, NT>`` compiles (csrc/mzx_batched.hip): one set of
    position fragments refilled in place, the four-chunk ring of the small tilings (layer kernel only), or two alternating
    fragment sets."""
    if (MT - 1) * NT >= 5:
        return "in-place"
    return "ring" if (MT * NT <= 4 and not tower) else "two-sets"


def streamed_launches(lib, handle, recurrent, batch):
    """
    Host-side (no GPU): the launch of every GEMM operator of a residual program on the streamed engine at ``batch``
    samples -- [{op, MT, NT, phases, k_loop, WM, WN, T, nsplit, cpg, taps, stride, in_layout, res}] -- i.e. WHICH
    ``rb_gemm_kernel<MT, NT>`` instantiation runs, with how many channel phases and which K loop.  The at-size parity
    tests assert these, bench.py prints them, and tests/test_streamed_coverage.py checks that every launch a bench
    workload makes is one a -m gpu parity test makes too.
    """
    out = []
    plan = (ctypes.c_int32 * 24)()
    shape = (ctypes.c_int32 * 16)()
    # towers first: their operators run inside ONE rb_tower_kernel<MT, NT> launch, not one by one
    in_tower = set()
    tw = (ctypes.c_int32 * 16)()
    index = 0
    while lib.mzx_net_streamed_tower(handle, int(bool(recurrent)), index, int(batch), ctypes.byref(tw)) == 0:
        t = dict(zip(("first", "count", "C", "H", "W", "T", "MT", "NT", "WM", "WN", "lds", "groups", "n_tail"), list(tw)))
        index += 1
        if t["groups"] == 0:     # at this batch the layers of the tower launch one by one
            continue
        in_tower.update(range(t["first"], t["first"] + t["count"] + t["n_tail"]))    # (the tail runs inside the launch)
        out.append(dict(op=t["first"], MT=t["MT"], NT=t["NT"], phases=1, k_loop="tower " + streamed_k_loop(t["MT"], t["NT"], tower=True),
                        WM=t["WM"], WN=t["WN"], T=t["T"], nsplit=1 + t["n_tail"], cpg=(t["C"] + 15) // 16, taps=9, stride=1,
                        in_layout=-t["count"], cin=t["C"], cout=t["C"]))
    # head chains whose levels run as grouped launches (tuning "rb_heads" = 2, the default)
    hd = (ctypes.c_int32 * 16)()
    lib.check(lib.mzx_net_streamed_heads(handle, int(bool(recurrent)), int(batch), ctypes.byref(hd)))
    level_of = {int(hd[2 + k]): (int(hd[14]) >> (2 * k)) & 3 for k in range(hd[0])}    # Linear operator -> its level in its chain
    first_gemm = len(out)
    for op in range(lib.mzx_net_num_operators(handle, int(bool(recurrent)))):
        if op in in_tower:
            continue
        lib.check(lib.mzx_net_streamed_plan(handle, int(bool(recurrent)), op, ctypes.byref(plan)))
        p = dict(zip(HipNetwork.STREAMED_PLAN_FIELDS, list(plan)))
        if p["kind"] != 0:
            continue
        lib.check(lib.mzx_net_streamed_shape(handle, int(bool(recurrent)), op, int(batch), ctypes.byref(shape)))
        s = dict(zip(STREAMED_SHAPE_FIELDS, list(shape)))
        out.append(dict(op=op, MT=s["MT"], NT=s["NT"], phases=s["phases"], k_loop=streamed_k_loop(s["MT"], s["NT"]),
                        WM=s["WM"], WN=s["WN"], T=s["T"], nsplit=s["nsplit"], cpg=s["cpg"], taps=p["taps"],
                        stride=p["stride"], in_layout=p["in_layout"], cin=p["cin"], cout=p["cout"]))
    # head chains: the operators of one level that share an instantiation run as slices of ONE rb_gemm_multi_kernel launch
    # (csrc/mzx_batched.hip rb_run_program; instantiated for NT = 1, MT <= 4) -- same shapes, another entry point
    for level in sorted(set(level_of.values())):
        members = [l for l in out[first_gemm:] if level_of.get(l["op"]) == level]
        for l in members:
            same = [m for m in members if (m["MT"], m["NT"]) == (l["MT"], l["NT"])]
            if len(same) > 1 and l["NT"] == 1 and l["MT"] <= 4:
                l["k_loop"] += " grouped"
    return out


def streamed_split(lib, handle, batch):
    """(first half, second half) the row-per-tree search runs a shard of ``batch`` trees as; (batch, 0) = undivided."""
    out = (ctypes.c_int32 * 2)()
    lib.check(lib.mzx_net_streamed_split(handle, int(batch), ctypes.byref(out)))
    return int(out[0]), int(out[1])


def instantiation_key(launch):
    """What identifies the compiled code path of a launch: (MT, NT, channel phases > 1, K loop)."""
    return (launch["MT"], launch["NT"], launch["phases"], launch["k_loop"])


def launch_key(launch):
    """The instantiation plus everything else that selects a branch inside it (wave grid, samples per workgroup, column
    split, kernel size / stride, gather layout)."""
    return instantiation_key(launch) + (launch["WM"], launch["WN"], launch["T"], launch["nsplit"], launch["cpg"],
                                        launch["taps"], launch["stride"], launch["in_layout"])


def summarize_launches(launches):
    """Compact, sorted labels '<MT,NT> phases K-loop' of a launch list (bench.py reports these per workload)."""
    return [(f"heads x{nt}" if kl == "heads" else f"<{mt},{nt}> {ph}ph {kl}")
            for mt, nt, ph, kl in sorted({instantiation_key(l) for l in launches})]


class MuZeroNetwork:
    """models.py:7-41: ``MuZeroNetwork(config)`` returns the network object."""

    def __new__(cls, config, _backend=None):
        return HipNetwork(config, _backend)


class HipNetwork:
    """The reference's AbstractNetwork surface (models.py:56-73) over the HIP kernels."""

    def __init__(self, config, backend=None):
        self.backend = backend if backend is not None else _lib.default_backend()
        lib = self.backend.lib
        self.action_space_size = len(config.action_space)
        self.full_support_size = 2 * config.support_size + 1
        self.support_size = config.support_size
        self._cfg = net_config_from(config)
        handle = ctypes.c_void_p()
        lib.check(lib.mzx_net_create(ctypes.byref(self._cfg), ctypes.byref(handle)))
        self.handle = handle
        self.hidden_size = lib.mzx_net_hidden_size(handle)
        self.input_size = lib.mzx_net_input_size(handle)
        self.num_params = lib.mzx_net_num_params(handle)
        self._tensors = []  # (key, offset, numel, shape)
        name = ctypes.create_string_buffer(256)
        off, numel, dims = ctypes.c_int64(), ctypes.c_int64(), (ctypes.c_int32 * 4)()
        for i in range(lib.mzx_net_num_tensors(handle)):
            lib.check(lib.mzx_net_tensor_info(handle, i, name, 256, ctypes.byref(off), ctypes.byref(numel),
                                              ctypes.byref(dims)))
            shape = tuple(d for d in dims if d > 0)
            self._tensors.append((name.value.decode(), off.value, numel.value, shape))
        if config.network == "resnet":
            c_in = self._cfg.observation_shape[0] * (config.stacked_observations + 1) + config.stacked_observations
            self.input_shape = (c_in, self._cfg.observation_shape[1], self._cfg.observation_shape[2])
            hw = self.hidden_size // config.channels
            h = (math.ceil(config.observation_shape[1] / 16) if config.downsample else config.observation_shape[1])
            self.hidden_shape = (config.channels, h, hw // h)
        else:
            self.input_shape = (self.input_size,)
            self.hidden_shape = (self.hidden_size,)
        self._flat = self.backend.zeros((self.num_params,), torch.float32)
        self._derived = self.backend.zeros((lib.mzx_net_derived_floats(handle),), torch.float32)
        self._workspace = None
        self._ws_batch = 0
        self._num_batches_tracked = {}
        self.set_weights(self._initial_weights())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.backend.lib.mzx_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ---- torch.nn.Module-flavoured plumbing the reference callers use ----
    def to(self, device):
        """self_play.py:28 -- the engine only runs on the GPU it was created on."""
        device = torch.device(device)
        if device.type != self.backend.device.type:
            raise _lib.MzxError(
                f"mzx networks live on {self.backend.device}; .to({device}) is not supported "
                "(set config.selfplay_on_gpu = True)"
            )
        return self

    def eval(self):
        return self

    def parameters(self):
        yield self._flat  # self_play.py:284 only reads next(model.parameters()).device

    def state_dict(self):
        out = collections.OrderedDict()
        flat = self._flat
        for key, off, numel, shape in self._tensors:
            out[key] = flat[off:off + numel].view(shape)
            if key.endswith(".running_var"):
                nbt = key[: -len("running_var")] + "num_batches_tracked"
                out[nbt] = self._num_batches_tracked.get(nbt, torch.zeros((), dtype=torch.int64))
        return out

    def get_weights(self):
        """models.py:69-70"""
        return dict_to_cpu(self.state_dict())

    def flat_weights(self):
        """The contiguous fp32 buffer holding every float tensor of the state_dict (RCCL payload)."""
        return self._flat

    def set_weights(self, weights):
        """models.py:72-73 -- strict load of a reference-format state_dict."""
        expected = [k for k, *_ in self._tensors]
        missing = [k for k in expected if k not in weights]
        extra = [k for k in weights if k not in set(expected) and not k.endswith("num_batches_tracked")]
        if missing or extra:
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:4]}..., unexpected keys {extra[:4]}...")
        host = torch.empty(self.num_params, dtype=torch.float32)
        for key, off, numel, shape in self._tensors:
            t = weights[key]
            if tuple(t.shape) != shape:
                raise RuntimeError(f"size mismatch for {key}: {tuple(t.shape)} vs {shape}")
            host[off:off + numel] = t.detach().to("cpu", torch.float32).reshape(-1)
        for k, v in weights.items():
            if k.endswith("num_batches_tracked"):
                self._num_batches_tracked[k] = torch.as_tensor(v).clone().cpu()
        self._flat.copy_(host)
        self.refresh_derived()

    def refresh_derived(self):
        """Re-derive folded BatchNorm terms after the flat buffer changed in place (e.g. RCCL broadcast)."""
        lib = self.backend.lib
        lib.check(lib.mzx_net_set_weights(self.handle, self.backend.ptr(self._flat), self.num_params,
                                          self.backend.ptr(self._derived), self._derived.numel(),
                                          self.backend.stream()))

    def _initial_weights(self):
        """torch.nn default initialisation of the same layers (Linear/Conv2d kaiming-uniform, BatchNorm identity)."""
        sd = collections.OrderedDict()
        bounds = {}
        for key, off, numel, shape in self._tensors:
            leaf = key.rsplit(".", 1)[-1]
            if leaf == "weight" and len(shape) >= 2:
                fan_in = int(math.prod(shape[1:]))
                bound = 1.0 / math.sqrt(fan_in)
                bounds[key[: -len("weight")]] = bound
                sd[key] = torch.empty(shape).uniform_(-bound, bound)
            elif leaf == "weight":
                sd[key] = torch.ones(shape)
            elif leaf == "bias":
                prefix = key[: -len("bias")]
                if prefix in bounds:
                    sd[key] = torch.empty(shape).uniform_(-bounds[prefix], bounds[prefix])
                else:
                    sd[key] = torch.zeros(shape)
            elif leaf == "running_mean":
                sd[key] = torch.zeros(shape)
            elif leaf == "running_var":
                sd[key] = torch.ones(shape)
            else:
                raise AssertionError(key)
        return sd

    # ---- engine selection / diagnostics (no reference counterpart) ----
    def fused_supported(self):
        """bit 0: initial_inference, bit 1: recurrent_inference run on the fused MFMA engine."""
        return int(self.backend.lib.mzx_net_fused_supported(self.handle))

    def streamed_supported(self):
        """bit 0: initial_inference, bit 1: recurrent_inference run on the streamed MFMA engine (large residual nets)."""
        return int(self.backend.lib.mzx_net_streamed_supported(self.handle))

    STREAMED_PLAN_FIELDS = ("kind", "in_layout", "out_layout", "res_layout", "taps", "stride", "cin", "cout", "hin",
                            "win", "hout", "wout", "T", "th", "tw", "tiles_x", "tiles_y", "PH", "PW", "cpg", "phases",
                            "rows", "mtiles", "lds_bytes")

    def streamed_plan(self, recurrent, op):
        """Workgroup tiling of operator `op` on the streamed engine (dict, see include/mzx.h)."""
        out = (ctypes.c_int32 * 24)()
        self.backend.lib.check(self.backend.lib.mzx_net_streamed_plan(self.handle, int(bool(recurrent)), int(op),
                                                                      ctypes.byref(out)))
        return dict(zip(self.STREAMED_PLAN_FIELDS, list(out)))

    def streamed_launches(self, recurrent, batch):
        """``streamed_launches`` of this network (module function above)."""
        return streamed_launches(self.backend.lib, self.handle, recurrent, batch)

    def streamed_split(self, batch):
        return streamed_split(self.backend.lib, self.handle, batch)

    def operator_out_floats(self, recurrent, op):
        return int(self.backend.lib.mzx_net_operator_out_floats(self.handle, int(bool(recurrent)), int(op)))

    def set_mode(self, mode):
        """0 = one kernel per operator, 1 = fused engine where available, streamed engine otherwise (default), 2 = fused with
        4-wave workgroups, 3 = streamed engine for everything, 4 / 5 = as 3 / 1 with the streamed engine layer by layer
        (no tower launches: the A/B of csrc/mzx_batched.hip's rb_tower_kernel)."""
        self.backend.lib.check(self.backend.lib.mzx_net_set_mode(self.handle, int(mode)))

    def num_operators(self, recurrent):
        return int(self.backend.lib.mzx_net_num_operators(self.handle, int(bool(recurrent))))

    def debug_prefix(self, recurrent, fused, n_ops, x, action=None):
        """Output tensor [batch, -1] of operator n_ops-1 of the chosen program on the chosen engine."""
        b, lib = self.backend, self.backend.lib
        x = self._prepare(x, self.hidden_size if recurrent else self.input_size)
        n = x.shape[0]
        act = None if action is None else action.to(b.device).reshape(-1).to(torch.int32).contiguous()
        cap = self.operator_out_floats(recurrent, n_ops - 1) if fused != 2 else 0   # (fused == 2: cycle stamps, not a tensor)
        if cap <= 0:
            cap = max(self.hidden_size, self.input_size, 4096) * 64
        out = b.zeros((n, cap), torch.float32)
        scratch = b.empty(((self.hidden_size + 2 * self.full_support_size + self.action_space_size) * n,), torch.float32)
        ws = self._ws(n)
        lib.check(lib.mzx_net_debug_prefix(self.handle, int(bool(recurrent)), int(fused), int(n_ops), b.ptr(x),
                                           b.ptr(act), n, b.ptr(out), out.numel(), b.ptr(scratch), scratch.numel(),
                                           b.ptr(ws), ws.numel(), b.stream()))
        return out

    # ---- inference ----
    def _ws(self, batch):
        if self._workspace is None or self._ws_batch < batch:
            n = self.backend.lib.mzx_net_workspace_floats(self.handle, batch)
            self._workspace = self.backend.empty((max(n, 1),), torch.float32)
            self._ws_batch = batch
        return self._workspace

    def _prepare(self, x, per_sample):
        x = x.to(self.backend.device, torch.float32)
        if x.dim() < 2 or x[0].numel() != per_sample:
            raise ValueError(f"expected [batch, ...] with {per_sample} values per sample, got {tuple(x.shape)}")
        return x.contiguous()

    def initial_inference(self, observation):
        b, lib = self.backend, self.backend.lib
        obs = self._prepare(observation, self.input_size)
        n = obs.shape[0]
        value = b.empty((n, self.full_support_size), torch.float32)
        reward = b.empty((n, self.full_support_size), torch.float32)
        policy = b.empty((n, self.action_space_size), torch.float32)
        hidden = b.empty((n,) + self.hidden_shape, torch.float32)
        ws = self._ws(n)
        lib.check(lib.mzx_net_initial_inference(self.handle, b.ptr(obs), n, b.ptr(value), b.ptr(reward), b.ptr(policy),
                                                b.ptr(hidden), b.ptr(ws), ws.numel(), b.stream()))
        return value, reward, policy, hidden

    def recurrent_inference(self, encoded_state, action):
        b, lib = self.backend, self.backend.lib
        state = self._prepare(encoded_state, self.hidden_size)
        n = state.shape[0]
        act = action.to(b.device).reshape(-1).to(torch.int32).contiguous()
        if act.numel() != n:
            raise ValueError("action must hold one action per sample")
        value = b.empty((n, self.full_support_size), torch.float32)
        reward = b.empty((n, self.full_support_size), torch.float32)
        policy = b.empty((n, self.action_space_size), torch.float32)
        hidden = b.empty((n,) + self.hidden_shape, torch.float32)
        ws = self._ws(n)
        lib.check(lib.mzx_net_recurrent_inference(self.handle, b.ptr(state), b.ptr(act), n, b.ptr(value), b.ptr(reward),
                                                  b.ptr(policy), b.ptr(hidden), b.ptr(ws), ws.numel(), b.stream()))
        return value, reward, policy, hidden
