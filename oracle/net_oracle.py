"""
TEST INFRASTRUCTURE ONLY -- CPU oracle for the network half of the hot path.

A functional (no ``torch.nn.Module``) fp32 restatement of the reference's two
inference networks, driven directly by a reference-format ``state_dict`` (keys
carry the ``.module.`` infix that ``torch.nn.DataParallel`` adds,
models.py:98-126 / :486-520).  Evaluation-mode semantics only (BatchNorm uses
running statistics, eps 1e-5; self_play.py:29 calls ``model.eval()``).

Parity status: PINNED by ``tests/test_oracle_golden.py`` against outputs of the
unmodified reference ``models.py`` executed in the build container
(``oracle/make_golden.py`` -> ``tests/golden/net_*.npz``).

Reference (relative to /root/reference):
  * mlp                                   models.py:630-642
  * MuZeroFullyConnectedNetwork           models.py:80-195
  * conv3x3 / ResidualBlock / DownSample  models.py:206-275
  * Representation/Dynamics/Prediction    models.py:300-433
  * MuZeroResidualNetwork                 models.py:436-623
  * support_to_scalar                     models.py:645-666
Only this module and the product's tests know the reference's floating-point
op order; the third-party arithmetic (ATen linear/conv/softmax) is the same
library the reference itself calls (torch CPU kernels).
"""
import math

import torch
import torch.nn.functional as F


def _mlp(sd, prefix, x):
    """models.py:630-642: Linear layers at even Sequential indices, ELU between."""
    idx = sorted(
        {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight")}
    )
    for j, i in enumerate(idx):
        x = F.linear(x, sd[f"{prefix}{i}.weight"], sd[f"{prefix}{i}.bias"])
        if j < len(idx) - 1:
            x = F.elu(x)
    return x


def _bn(sd, prefix, x):
    return F.batch_norm(
        x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.1, 1e-5,
    )


def _resblock(sd, prefix, x):
    """models.py:213-229"""
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], None, 1, 1)
    out = F.relu(_bn(sd, prefix + ".bn1", out))
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, prefix + ".bn2", out)
    return F.relu(out + x)


def _resblocks(sd, prefix, x):
    i = 0
    while f"{prefix}.{i}.conv1.weight" in sd:
        x = _resblock(sd, f"{prefix}.{i}", x)
        i += 1
    return x


def _scale_rows(x):
    """models.py:136-145 / :156-169: min-max scale over dim 1, +1e-5 on tiny scales."""
    lo = x.min(1, keepdim=True)[0]
    hi = x.max(1, keepdim=True)[0]
    scale = hi - lo
    scale = torch.where(scale < 1e-5, scale + 1e-5, scale)
    return (x - lo) / scale


def _scale_planes(x):
    """models.py:527-553 / :573-599: per-(sample, channel) spatial min-max scale."""
    flat = x.view(x.shape[0], x.shape[1], -1)
    lo = flat.min(2, keepdim=True)[0].unsqueeze(-1)
    hi = flat.max(2, keepdim=True)[0].unsqueeze(-1)
    scale = hi - lo
    scale = torch.where(scale < 1e-5, scale + 1e-5, scale)
    return (x - lo) / scale


def _zero_reward_logits(n, full_support):
    """models.py:176-183: log of a one-hot at the support centre (-inf elsewhere)."""
    onehot = torch.zeros(n, full_support)
    onehot[:, full_support // 2] = 1.0
    return torch.log(onehot)


class FullyConnectedOracle:
    """models.py:80-195 as pure functions of a state_dict."""

    def __init__(self, state_dict, action_space_size, support_size, dtype=torch.float32):
        self.dtype = dtype
        self.sd = {k: v.to(dtype) for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.A = action_space_size
        self.F = 2 * support_size + 1

    def _prediction(self, s):
        return (
            _mlp(self.sd, "prediction_policy_network.module.", s),
            _mlp(self.sd, "prediction_value_network.module.", s),
        )

    def initial_inference(self, observation):
        obs = observation.view(observation.shape[0], -1)
        s = _scale_rows(_mlp(self.sd, "representation_network.module.", obs))
        policy, value = self._prediction(s)
        return value, _zero_reward_logits(len(observation), self.F), policy, s

    def recurrent_inference(self, encoded_state, action):
        onehot = torch.zeros(action.shape[0], self.A, dtype=self.dtype)
        onehot.scatter_(1, action.long(), 1.0)
        x = torch.cat((encoded_state, onehot), dim=1)
        nxt = _mlp(self.sd, "dynamics_encoded_state_network.module.", x)
        reward = _mlp(self.sd, "dynamics_reward_network.module.", nxt)  # on the UNscaled state, :157-159
        s = _scale_rows(nxt)
        policy, value = self._prediction(s)
        return value, reward, policy, s


class ResidualOracle:
    """models.py:436-623 as pure functions of a state_dict."""

    def __init__(self, state_dict, action_space_size, support_size, downsample, dtype=torch.float32):
        self.dtype = dtype   # float32 = the reference's arithmetic; float64 = "exact" yardstick for conditioning checks
        self.sd = {k: v.to(dtype) for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.A = action_space_size
        self.F = 2 * support_size + 1
        self.downsample = downsample
        if downsample not in (False, None, "resnet", "CNN"):
            raise NotImplementedError('downsample should be "resnet" or "CNN".')  # models.py:327

    def _downsample(self, x):
        sd, p = self.sd, "representation_network.module.downsample_net"
        if self.downsample == "CNN":
            # DownsampleCNN, models.py:278-297: Conv2d(K = 2 * ceil(H / 16), stride 4, padding 2) + ReLU +
            # MaxPool2d(3, 2) + Conv2d(5, padding 2) + ReLU + MaxPool2d(3, 2), AdaptiveAvgPool2d(ceil(H/16), ceil(W/16))
            h_w = (math.ceil(x.shape[2] / 16), math.ceil(x.shape[3] / 16))
            x = F.relu(F.conv2d(x, sd[p + ".features.0.weight"], sd[p + ".features.0.bias"], 4, 2))
            x = F.max_pool2d(x, 3, 2)
            x = F.relu(F.conv2d(x, sd[p + ".features.3.weight"], sd[p + ".features.3.bias"], 1, 2))
            x = F.max_pool2d(x, 3, 2)
            return F.adaptive_avg_pool2d(x, h_w)
        # DownSample.forward, models.py:264-275
        x = F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 1)
        x = _resblocks(sd, p + ".resblocks1", x)
        x = F.conv2d(x, sd[p + ".conv2.weight"], None, 2, 1)
        x = _resblocks(sd, p + ".resblocks2", x)
        x = F.avg_pool2d(x, 3, 2, 1)
        x = _resblocks(sd, p + ".resblocks3", x)
        return F.avg_pool2d(x, 3, 2, 1)

    def representation_unscaled(self, obs):
        """The encoded state BEFORE the per-plane min-max scaling (diagnostics: which planes are near-flat)."""
        sd, p = self.sd, "representation_network.module"
        if self.downsample:
            x = self._downsample(obs)
        else:
            x = F.relu(_bn(sd, p + ".bn", F.conv2d(obs, sd[p + ".conv.weight"], None, 1, 1)))
        return _resblocks(sd, p + ".resblocks", x)

    def _representation(self, obs):
        return _scale_planes(self.representation_unscaled(obs))

    def _prediction(self, s):
        sd, p = self.sd, "prediction_network.module"
        x = _resblocks(sd, p + ".resblocks", s)
        v = F.conv2d(x, sd[p + ".conv1x1_value.weight"], sd[p + ".conv1x1_value.bias"])
        q = F.conv2d(x, sd[p + ".conv1x1_policy.weight"], sd[p + ".conv1x1_policy.bias"])
        value = _mlp(sd, p + ".fc_value.", v.reshape(v.shape[0], -1))
        policy = _mlp(sd, p + ".fc_policy.", q.reshape(q.shape[0], -1))
        return policy, value

    def initial_inference(self, observation):
        s = self._representation(observation)
        policy, value = self._prediction(s)
        return value, _zero_reward_logits(len(observation), self.F), policy, s

    def dynamics_unscaled(self, encoded_state, action):
        """The next state BEFORE the per-plane min-max scaling (what the reward head reads)."""
        sd, p = self.sd, "dynamics_network.module"
        b, _, h, w = encoded_state.shape
        # scalar action plane action/|A| (NOT one-hot), models.py:557-571
        plane = action[:, :, None, None] * torch.ones((b, 1, h, w)).float() / self.A
        x = torch.cat((encoded_state, plane.to(self.dtype)), dim=1)
        x = F.relu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, 1, 1)))
        return _resblocks(sd, p + ".resblocks", x)

    def recurrent_inference(self, encoded_state, action):
        sd, p = self.sd, "dynamics_network.module"
        b = encoded_state.shape[0]
        state = self.dynamics_unscaled(encoded_state, action)
        r = F.conv2d(state, sd[p + ".conv1x1_reward.weight"], sd[p + ".conv1x1_reward.bias"])
        reward = _mlp(sd, p + ".fc.", r.reshape(b, -1))
        s = _scale_planes(state)
        policy, value = self._prediction(s)
        return value, reward, policy, s


def make_oracle_network(cfg, state_dict, dtype=torch.float32):
    """Counterpart of the factory models.py:7-41 for the two architectures."""
    if cfg.network == "fullyconnected":
        return FullyConnectedOracle(state_dict, len(cfg.action_space), cfg.support_size, dtype)
    if cfg.network == "resnet":
        return ResidualOracle(state_dict, len(cfg.action_space), cfg.support_size, cfg.downsample, dtype)
    raise NotImplementedError('The network parameter should be "fullyconnected" or "resnet".')


def support_to_scalar(logits, support_size):
    """models.py:645-666"""
    p = torch.softmax(logits, dim=1)
    support = torch.arange(-support_size, support_size + 1).float().expand(p.shape)
    x = torch.sum(support * p, dim=1, keepdim=True)
    eps = 0.001
    return torch.sign(x) * (
        ((torch.sqrt(1 + 4 * eps * (torch.abs(x) + 1 + eps)) - 1) / (2 * eps)) ** 2 - 1
    )


class NetworkEvaluator:
    """
    Adapts a network object exposing ``initial_inference`` / ``recurrent_inference``
    (this oracle, or the reference's own model in make_golden.py) to the
    evaluator protocol of ``mcts_oracle.run_search``; mirrors the glue in
    self_play.py:280-295, :339-344 and :460-462.  Can record every output.
    """

    def __init__(self, net, support_size, record=False):
        self.net, self.support_size = net, support_size
        self.log = [] if record else None

    def _finish(self, value, reward, policy_logits, hidden, actions):
        v = support_to_scalar(value, self.support_size).item()
        r = support_to_scalar(reward, self.support_size).item()
        pri = torch.softmax(
            torch.tensor([policy_logits[0][a] for a in actions]), dim=0
        ).tolist()
        if self.log is not None:
            self.log.append(
                dict(value=v, reward=r, priors=list(pri),
                     value_logits=value[0].numpy().copy(), reward_logits=reward[0].numpy().copy(),
                     policy_logits=policy_logits[0].numpy().copy(),
                     hidden=hidden[0].numpy().copy())
            )
        return v, r, pri, hidden

    def initial(self, observation, actions):
        with torch.no_grad():
            # (the reference feeds float32, self_play.py:280-285; a binary64 yardstick network takes the observation in
            # its own precision)
            obs = torch.tensor(observation).float().to(getattr(self.net, "dtype", torch.float32)).unsqueeze(0)
            return self._finish(*self.net.initial_inference(obs), actions)

    def recurrent(self, hidden, action, actions):
        with torch.no_grad():
            out = self.net.recurrent_inference(hidden, torch.tensor([[action]]))
            return self._finish(*out, actions)
