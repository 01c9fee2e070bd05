import sys, os
sys.path.insert(0, "muzero-general_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy, torch
from mzx import configs, models, synthetic, self_play
from oracle import net_oracle, build_ref, mcts_oracle
cfg = configs.breakout(num_simulations=50)
net = models.MuZeroNetwork(cfg)
ref_models, _ = build_ref.load()
torch.manual_seed(0)
sd = {k: v.clone() for k, v in ref_models.MuZeroNetwork(cfg).get_weights().items()}
net.set_weights(sd)
B = 64
obs = synthetic.observations(B, net.input_shape, seed=77)
o32 = net_oracle.make_oracle_network(cfg, sd)
o64 = net_oracle.make_oracle_network(cfg, sd, dtype=torch.float64)
x = torch.tensor(obs)
dev = net.initial_inference(x)
with torch.no_grad():
    w32 = o32.initial_inference(x); w64 = o64.initial_inference(x.double())
names = ("value", "reward", "policy", "hidden")
for k, nm in enumerate(names):
    if k == 1: continue
    g = dev[k].cpu().numpy().reshape(B, -1).astype(numpy.float64)
    a = w32[k].numpy().reshape(B, -1).astype(numpy.float64); b = w64[k].numpy().reshape(B, -1)
    e32 = numpy.abs(g - a).max(1); e64 = numpy.abs(g - b).max(1); own = numpy.abs(a - b).max(1)
    worst = numpy.argsort(-e32)[:4]
    print("initial", nm, "max dev-vs-f32 %.3e dev-vs-f64 %.3e f32-vs-f64 %.3e" % (e32.max(), e64.max(), own.max()), "worst rows", [(int(r), float(e32[r]), float(e64[r]), float(own[r])) for r in worst])
print("row 40: dev policy", dev[2][40].cpu().numpy(), "f32", w32[2][40].numpy(), "f64", w64[2][40].numpy())
# recurrent from the oracle's f64 hidden (as f32) for every action
for act in range(len(cfg.action_space)):
    h = w64[3].float()
    a_t = torch.full((B,), act, dtype=torch.int32)
    devr = net.recurrent_inference(h, a_t)
    with torch.no_grad():
        r32 = o32.recurrent_inference(h, a_t.long().reshape(-1, 1)); r64 = o64.recurrent_inference(h.double(), a_t.long().reshape(-1, 1))
    for k, nm in enumerate(names):
        g = devr[k].cpu().numpy().reshape(B, -1).astype(numpy.float64)
        a = r32[k].numpy().reshape(B, -1).astype(numpy.float64); b = r64[k].numpy().reshape(B, -1)
        e32 = numpy.abs(g - a).max(1); e64 = numpy.abs(g - b).max(1); own = numpy.abs(a - b).max(1)
        print("recurrent a=%d" % act, nm, "max dev-vs-f32 %.3e (row %d) dev-vs-f64 %.3e f32-vs-f64 %.3e; row 40: %.3e %.3e %.3e" % (e32.max(), int(e32.argmax()), e64.max(), own.max(), e32[40], e64[40], own[40]))
# the oracle searches of tree 40 in f32 / f64 and the device's
rs = numpy.random.RandomState(8)
A = len(cfg.action_space)
legal = [sorted(rs.choice(A, size=rs.randint(2, A + 1), replace=False).tolist()) for _ in range(B)]
for dt in (torch.float32, torch.float64):
    onet = net_oracle.make_oracle_network(cfg, sd, dtype=dt)
    ev = net_oracle.NetworkEvaluator(onet, cfg.support_size)
    tree = mcts_oracle.run_search(cfg, ev, obs[40], legal[40], 0, True, numpy.random.RandomState(4040))
    print(dt, "trace", [(p, a) for p, a, _ in tree.trace][:8], "margins", [("%.2e" % m[0], m[1]) for m in tree.margins[:8]], "root counts", tree.root_visit_counts(cfg.action_space))
engine = self_play.BatchedMCTS(cfg, net, B)
res = engine.run(list(obs), legal, [0] * B, True, [numpy.random.RandomState(4000 + i) for i in range(B)])
t = engine.export_trees(B)
tr = []
for n in range(1, 9):
    par = int(t["parent"][40, n]); slot = int(numpy.nonzero(t["child"][40, par] == n)[0][0]); tr.append((par, legal[40][slot] if par == 0 else slot))
print("device trace", tr, "root counts", list(res.visit_counts[40]), "legal", legal[40])
print("device root priors", t["prior"][40, 0], "reward node1..3", t["reward"][40, :4], "value_sum", t["value_sum"][40, :4])
