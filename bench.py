"""
bench.py -- BASELINE.json metric: MCTS simulations/s (whole job) and self-play steps/s, CartPole & Connect4.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL: either the driver launches this file under
``python -m torch.distributed.run --nproc-per-node N ...`` or, when started as plain ``python bench.py --gpus N``,
the script re-executes itself that way (the reference spawns its N self-play workers itself, muzero.py:156-203).

One "step" = one pass of the hot path over one batch of synthetic input: B independent roots, each doing
initial_inference + num_simulations x {select, recurrent_inference, expand, backpropagate} (one self-play move
per tree, self_play.py:144-150).  Default workload = BASELINE config C2: CartPole FullyConnectedNetwork, 4096
trees x 50 simulations per GPU; the same invocation also measures the Connect4 half of the metric (C4: Connect4
ResNet, 1024 trees x 200 simulations per GPU, the configuration BASELINE.json shards over 8 GPUs) and reports it
under ``workloads``.  Inputs (stacked observations, legal actions, Dirichlet noise, tie tape, weights) are
resident in HBM before the timed region; every rank owns an independent shard of trees (weak scaling, no
data-path collective; RCCL only broadcasts the flat weight buffer, outside the timed region, as the reference's
weight pull self_play.py:37).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline"      algorithmic tree bytes (SURVEY.md section 8d formula, with the measured mean leaf depth) /
                  HIP-event time of the search launch, vs the HBM peak (residual networks: network FLOPs vs the
                  dense FP32-input MFMA peak)
  "workloads"     the other BASELINE configurations (C3, C4 = the Connect4 half of the metric, C5), C2 on the reference's
                  shipped checkpoint, and the reference's games/gomoku.py / games/atari.py as shipped, measured in this
                  invocation -- compact entries: w workload, v sims/s (median block; min / max beside it), ms per step, B
                  trees, S simulations, L mean leaf depth, k the search kernel (+ its shape), roofline {bound, achieved
                  (TFLOP/s | GB/s), frac, traffic (PMC HBM bytes per step | null), launch_ms}
  "per_rank" / "single_gpu_reference"   N > 1: every rank's own rate, and rank 0 timed alone just before
  "selfplay_end_to_end*"  self-play steps/s through the plugin surface (N = 1)
  "selfplay_actor_loop"   continuous_self_play itself, batched protocol: play_rounds + initial PER priorities (td_steps 50) +
                          save_game of every finished game, in-process storage (N = 1)
  "observation_stacker"   the path's HBM-bound kernel (mzx_obs_stack, atari geometry) against the HBM peak
  "cpu_baseline"  kind "reference": the UNMODIFIED reference MCTS(config).run + models.py (oracle/_ref, compiled
                  from /root/reference by oracle/build_ref.py) on this box's host cores, bounded sample (kind "port": the
                  CPU oracle oracle/*.py, when the bytecode did not travel).
Weights: the reference constructor's under torch.manual_seed(0) (--weights synthetic = rounds 1-4's).
N > 1 validates itself: after the RCCL broadcast every rank's flat buffer has rank 0's checksum, and one search on
rank-independent inputs builds the same trees on every rank -- or the run fails ("collective" in the line).
``--dry-run`` (CI only): gloo + the serial test double of the ABI on the CPU -- exercises the launch / sharding /
broadcast / JSON plumbing without a GPU; its numbers are NOT measurements and the line says so.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "muzero-general_amd"))
sys.path.insert(0, ROOT)

# dmabuf IPC is the only mode the host driver supports: RCCL between the ranks of a node needs it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

if os.environ.get("MZX_BENCH_CPU_WORKER"):
    # a spawned CPU-baseline worker re-imports this module: keep its import-time chatter (libdrm's "amdgpu.ids: No such
    # file" line, once per process) out of the parent's output -- the driver reads the JSON line from the tail of it
    os.dup2(os.open(os.devnull, os.O_WRONLY), 2)

import numpy  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3  # dense FP32-input MFMA peak (= FP32 vector peak), same guide

# workloads that choose their network engine themselves (unless --net-mode says otherwise)
WORKLOAD_NET_MODE = {}
# workloads that run with entries of the library's tuning table moved (include/mzx.h "Tuning": the A/B legs)
WORKLOAD_TUNING = {"c4-ws": {"wide_towers": 0}, "c4-rows": {"rt_search": 0}}
DEFAULT_WORKLOAD = "c2"
# measured in the same invocation and reported under "workloads" when the main workload is the default one.  Every
# launch the streamed ones make (c4-large, gomoku, atari) is parity-tested at size: tests/test_streamed_coverage.py
DEFAULT_ALSO = "c2-ckpt,c3,c4,c4-ws,c4-large,c5,c5-512,gomoku,atari"
# workloads whose weights are not the reference constructor's: the checkpoint the reference ships (results/cartpole/
# model.checkpoint, muzero.py:426-464), carried by the committed fixture tests/golden/net_fc_cartpole_pretrained.npz
WORKLOAD_WEIGHTS = {"c2-ckpt": "checkpoint:net_fc_cartpole_pretrained.npz"}

WORKLOADS = {
    # name: (config factory name, overrides, trees per GPU, description)
    "c2": ("cartpole", {}, 4096, "C2 CartPole FullyConnectedNetwork, 4096 trees x 50 sims per GPU"),
    "c2-ckpt": ("cartpole", {}, 4096, "C2 on the reference's shipped results/cartpole/model.checkpoint"),
    "c3": ("tictactoe", {}, 1024, "C3 Tic-tac-toe MuZeroResidualNetwork, 1024 trees x 25 sims per GPU"),
    # (round 5: the library runs every simulation of this search in ONE launch, mzx::rt_search_kernel -- the trunks as towers
    # inside; "c4-rows" is round 4's route, the same towers launch by launch, "c4-ws" the LDS-resident whole-search kernel)
    "c4": ("connect4", {}, 1024, "C4 Connect4 ResNet, 1024 trees x 200 sims per GPU"),
    "c4-ws": ("connect4", {}, 1024, "C4 on mzx::rz_search_kernel (LDS-resident engine, A/B)"),
    "c4-rows": ("connect4", {}, 1024, "C4 on per-simulation launches (A/B)"),
    # the same network and search at a large shard (nine rounds of the shape rt_search_kernel runs best on: three boards per
    # 256-thread workgroup, <8,1>)
    "c4-large": ("connect4", {}, 9216, "C4 Connect4 ResNet at a large shard, 9216 trees x 200 sims per GPU"),
    "c5": ("breakout", {"num_simulations": 50}, 64, "C5 Breakout ResNet (resnet stem), 64 trees x 50 sims per GPU"),
    # (BASELINE's C5 is 512 trees over 8 GPUs = 64 workgroups on 256 CUs; the same kernel with the 512 trees on ONE GPU)
    "c5-512": ("breakout", {"num_simulations": 50}, 512, "C5 Breakout ResNet, 512 trees x 50 sims on one GPU"),
    # the reference's large residual configurations as shipped (streamed MFMA engine, csrc/mzx_batched.hip)
    "gomoku": ("gomoku", {}, 1024, "games/gomoku.py as shipped: 128 ch x 6 blocks, 11 x 11, 1024 trees x 400 sims per GPU"),
    # (1024 trees: two half-shards of 512 on two streams, 0.75 of the MFMA peak for the whole step; --trees 256: 0.61)
    "atari": ("atari", {}, 1024, "games/atari.py as shipped: 256 ch x 16 blocks, 96 x 96 x 131 input, 1024 trees x 50 sims per GPU"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--also", default=None,
                    help="comma list of further workloads measured in the same invocation and reported under "
                         "'workloads' (default: c4 when the main workload is c2; 'none' = skip)")
    ap.add_argument("--also-steps", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=None,
                    help="timed blocks of exactly --steps steps each (default: 25 for blocks shorter than 0.1 s, fewer for "
                         "long ones); the line's value is the MEDIAN block, min / max are reported beside it")
    ap.add_argument("--trees", type=int, default=None, help="trees per GPU (default: the workload's)")
    ap.add_argument("--mode", default="auto", choices=["auto", "generic", "fused", "fused-v1"],
                    help="fused-v1: first-generation fully connected whole-search kernel (A/B; instrumented builds only)")
    ap.add_argument("--net-mode", default="fused", choices=["fused", "fused-4wave", "per-operator", "streamed"],
                    help="residual networks: fused MFMA engine (default) or one kernel per operator")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall budget of each CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=None)
    ap.add_argument("--selfplay-moves", type=int, default=32,
                    help="moves per game of the end-to-end self-play legs (SelfPlay(num_games=B) on the synthetic game; "
                         "0 = skip); the games of the path's configurations run for hundreds of moves (cartpole: 500)")
    ap.add_argument("--weights", default="reference", choices=["reference", "synthetic"],
                    help="reference: torch.manual_seed(0); models.MuZeroNetwork(config) of the UNMODIFIED reference (oracle/_ref "
                         "bytecode; SURVEY.md section 8d); synthetic: mzx.synthetic.fill_state_dict(seed 0), rounds 1-4's weights")
    ap.add_argument("--tuning", default="",
                    help="name=value,... entries of the library's tuning table (include/mzx.h) for this run (A/B measurements)")
    ap.add_argument("--dry-run", action="store_true",
                    help="CI plumbing check on the CPU (gloo + tests/hostcheck); not a measurement")
    return ap.parse_args()


# ----------------------------------------------------------------------------- launch (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` started by hand: become N ranks (one per GPU) and forward the exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MZX_BENCH_SELF_LAUNCHED="1")
    raise SystemExit(subprocess.call(cmd, env=env))


class Env:
    """Rank / device plumbing; ``dry`` swaps RCCL + the GPU for gloo + the serial test double (CI only)."""

    def __init__(self, args):
        self.dry = bool(args.dry_run)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = None
        if self.dry:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import hostcheck  # test double of the ABI: plumbing check only
            self.backend = hostcheck.backend()
        else:
            torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.dry:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))

    def sync(self):
        if not self.dry:
            torch.cuda.synchronize()

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()

    def fence(self):
        """barrier + device synchronize, as the contract brackets the timed region."""
        self.sync()
        self.barrier()
        self.sync()

    def all_max(self, x):
        if self.world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.dry else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def all_gather(self, x):
        if self.world == 1:
            return [x]
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if self.dry else "cuda")
        out = [torch.zeros_like(t) for _ in range(self.world)]
        torch.distributed.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self.world > 1:
            torch.distributed.destroy_process_group()


class Stopwatch:
    """HIP events on the launch stream (torch's current stream = the stream the ABI calls are given)."""

    def __init__(self, env, n):
        self.env = env
        self.ev = None if env.dry else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                        for _ in range(n)]
        self.t = [[0.0, 0.0] for _ in range(n)]

    def start(self, k):
        if self.ev:
            self.ev[k][0].record()
        else:
            self.t[k][0] = time.perf_counter()

    def stop(self, k):
        if self.ev:
            self.ev[k][1].record()
        else:
            self.t[k][1] = time.perf_counter()

    def mean_ms(self):
        if self.ev:
            return float(numpy.mean([a.elapsed_time(b) for a, b in self.ev]))
        return float(numpy.mean([(b - a) * 1e3 for a, b in self.t]))


# ----------------------------------------------------------------------------- CPU baselines
def _cpu_worker_port(args):
    """One host core: the oracle's per-node MCTS with its batch-1 torch network (restatement of the reference)."""
    workload, worker, seconds, _ = args
    torch.set_num_threads(1)
    from mzx import configs, synthetic
    from oracle import mcts_oracle, net_oracle

    name, overrides, _, _ = WORKLOADS[workload]
    cfg = configs.BY_NAME[name](**overrides)
    sd = synthetic.fill_state_dict(_state_dict_template(cfg), 0)
    net = net_oracle.make_oracle_network(cfg, sd)
    c_in = cfg.observation_shape[0] * (cfg.stacked_observations + 1) + cfg.stacked_observations
    obs = synthetic.observations(64, (c_in,) + tuple(cfg.observation_shape[1:]), seed=123 + worker)
    legal = list(cfg.action_space)
    sims, searches, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ev = net_oracle.NetworkEvaluator(net, cfg.support_size)
        rng = numpy.random.RandomState(1000 + worker * 1000 + searches)
        mcts_oracle.run_search(cfg, ev, obs[searches % 64], legal, 0, True, rng)
        sims += cfg.num_simulations
        searches += 1
    return sims, searches, time.perf_counter() - t0


def _cpu_worker_reference(args):
    """One host core: the UNMODIFIED reference -- self_play.MCTS(config).run on models.MuZeroNetwork (oracle/_ref)."""
    workload, worker, seconds, weights_kind = args
    torch.set_num_threads(1)
    from mzx import configs, synthetic
    from oracle import build_ref

    ref_models, ref_self_play = build_ref.load()
    name, overrides, _, _ = WORKLOADS[workload]
    cfg = configs.BY_NAME[name](**overrides)
    torch.manual_seed(0)
    model = ref_models.MuZeroNetwork(cfg)      # (the weights the GPU leg uses: this very constructor under this very seed)
    if weights_kind == "synthetic":
        model.set_weights(synthetic.fill_state_dict(model.get_weights(), 0))
    model.eval()
    c_in = cfg.observation_shape[0] * (cfg.stacked_observations + 1) + cfg.stacked_observations
    obs = synthetic.observations(64, (c_in,) + tuple(cfg.observation_shape[1:]), seed=123 + worker)
    legal = list(cfg.action_space)
    with torch.no_grad():   # one warm-up search, then the timed sample
        ref_self_play.MCTS(cfg).run(model, obs[0], legal, 0, True)
    sims, searches, t0 = 0, 0, time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t0 < seconds:
            numpy.random.seed(1000 + worker * 1000 + searches)
            ref_self_play.MCTS(cfg).run(model, obs[searches % 64], legal, 0, True)
            sims += cfg.num_simulations
            searches += 1
    return sims, searches, time.perf_counter() - t0


def _cpu_worker_reference_play_game(args):
    """One host core: the UNMODIFIED reference's SelfPlay.play_game (self_play.py:110-183: search, select_action, Game.step,
    GameHistory per move) -- self-play STEPS per second.  Games: the synthetic fixed-shape game of the
    GPU legs with its max_moves (C2), or the real connect4 rules (mzx.games.Connect4: the per-object plugin class, identical
    to games/connect4.py observation for observation) on the C4 network, whole games.  Counts the moves of FINISHED games and
    the time at which the last of them finished."""
    what, worker, seconds, moves = args
    torch.set_num_threads(1)
    import copy

    from mzx import configs, games as board_games, synthetic
    from oracle import build_ref

    ref_models, ref_self_play = build_ref.load()
    if what == "c2":
        cfg = copy.copy(configs.cartpole())
        cfg.max_moves = moves
        Game = synthetic.make_synthetic_game(cfg.observation_shape, len(cfg.action_space), len(cfg.players))
    else:
        cfg = configs.connect4()
        Game = board_games.PER_OBJECT["connect4"]
    cfg.selfplay_on_gpu = False         # the CPU baseline: the reference actor's own switch (self_play.py:28)
    torch.manual_seed(0)
    weights = ref_models.MuZeroNetwork(cfg).get_weights()
    actor = ref_self_play.SelfPlay({"weights": weights}, Game, cfg, 1000 + worker)     # (seeds numpy + torch, builds the model)
    steps, games, t0, t_last = 0, 0, time.perf_counter(), None
    with torch.no_grad():
        while t_last is None or time.perf_counter() - t0 < seconds:
            h = actor.play_game(1.0, cfg.temperature_threshold, False, "self", 0)
            steps += len(h.action_history) - 1
            games += 1
            t_last = time.perf_counter() - t0
    return steps, games, t_last


def _state_dict_template(cfg):
    """{key: zero tensor} in reference state_dict order, from the library's host-side weight table (no GPU)."""
    from mzx import _lib, models

    lib = _lib.Library(_lib.LIB_PATH)
    c = models.net_config_from(cfg)
    h = ctypes.c_void_p()
    lib.check(lib.mzx_net_create(ctypes.byref(c), ctypes.byref(h)))
    out = {}
    name = ctypes.create_string_buffer(256)
    off, numel, dims = ctypes.c_int64(), ctypes.c_int64(), (ctypes.c_int32 * 4)()
    for i in range(lib.mzx_net_num_tensors(h)):
        lib.check(lib.mzx_net_tensor_info(h, i, name, 256, ctypes.byref(off), ctypes.byref(numel), ctypes.byref(dims)))
        out[name.value.decode()] = torch.zeros(tuple(d for d in dims if d > 0))
    lib.mzx_net_destroy(h)
    return out


def _cpu_pool_map(pool_cores, jobs):
    """Runs (worker function, argument list) jobs one after the other in ONE pool of spawned single-thread processes that do
    not see the GPU (the reference wraps its networks in torch.nn.DataParallel, models.py:98-126, which would route batch-1
    inferences through cuda:0).  Returns the per-job result lists."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    hidden = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    hidden["MZX_BENCH_CPU_WORKER"] = os.environ.get("MZX_BENCH_CPU_WORKER")
    os.environ.update({"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": "", "MZX_BENCH_CPU_WORKER": "1"})
    try:
        with ctx.Pool(pool_cores) as pool:
            return [pool.map(fn, argl) for fn, argl in jobs]
    finally:
        for k, v in hidden.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _search_baseline_entry(res, cores, seconds, kind):
    sims = sum(r[0] for r in res)
    wall = max(r[2] for r in res)
    rates = [r[0] / r[2] for r in res]
    what = ("reference MCTS.run (oracle/_ref)" if kind == "reference" else "oracle/mcts_oracle.py + net_oracle.py")
    return {
        "value": sims / wall, "unit": "sims/s", "cores": cores, "kind": kind,
        "sample": f"{sum(r[1] for r in res)} searches x {sims // max(1, sum(r[1] for r in res))} sims, "
                  f"{cores} processes x {seconds:.0f} s, 1 thread each; {what}",
        "per_core": res[0][0] / res[0][2],
        # spread over the worker processes (each one's own sims / wall): one aggregate number hides a slow socket / a
        # busy host; the aggregate above is sum(sims) / max(wall)
        "per_core_min_median_max": [float(numpy.min(rates)), float(numpy.median(rates)), float(numpy.max(rates))],
    }


def cpu_baselines(workload, seconds, cores, kind, weights_kind="reference", c4=False, steps_moves=0):
    """
    The CPU legs BASELINE.md section 3 names, timed on this box's host cores in one pool: `workload`'s MCTS.run (the line's
    cpu_baseline), and -- when the unmodified reference travelled (kind "reference") -- the Connect4 half of the metric
    (MCTS(config).run on games/connect4.py's network, S = 200) and self-play STEPS per second (SelfPlay.play_game on the
    synthetic C2 game and on whole connect4 games).  Baselines only (a large GPU / CPU ratio says nothing about kernel
    quality).  Returns {"cpu_baseline": ..., "cpu_baseline_c4": ..., "cpu_baseline_steps": ...}.
    """
    cores = cores or min(os.cpu_count() or 1, 64)
    worker = _cpu_worker_reference if kind == "reference" else _cpu_worker_port
    jobs = [(worker, [(workload, w, seconds, weights_kind) for w in range(cores)])]
    extra = kind == "reference"
    if extra and c4:
        jobs.append((worker, [("c4", w, seconds, weights_kind) for w in range(cores)]))
    if extra and steps_moves > 0:
        jobs.append((_cpu_worker_reference_play_game, [("c2", w, seconds, steps_moves) for w in range(cores)]))
        if c4:
            jobs.append((_cpu_worker_reference_play_game, [("connect4", w, seconds, 0) for w in range(cores)]))
    res = _cpu_pool_map(cores, jobs)
    out = {"cpu_baseline": _search_baseline_entry(res[0], cores, seconds, kind)}
    k = 1
    if extra and c4:
        e = _search_baseline_entry(res[k], cores, seconds, kind)
        e["workload"] = "C4 Connect4 ResNet, MCTS.run of one tree at a time, 200 sims"
        out["cpu_baseline_c4"] = e
        k += 1
    if extra and steps_moves > 0:
        legs = {}
        for name in (["c2_synthetic_game"] + (["connect4_whole_games"] if c4 else [])):
            r = res[k]
            k += 1
            rates = [x[0] / x[2] for x in r]
            legs[name] = {"value": sum(x[0] for x in r) / max(x[2] for x in r), "unit": "steps/s", "cores": cores,
                          "games_finished": sum(x[1] for x in r), "moves_per_game": sum(x[0] for x in r) / max(1, sum(x[1] for x in r)),
                          "per_core_min_median_max": [float(numpy.min(rates)), float(numpy.median(rates)), float(numpy.max(rates))]}
        legs["kind"] = "reference"
        legs["sample"] = (f"reference SelfPlay.play_game (oracle/_ref), {cores} processes x 1 thread, whole games for {seconds:.0f} s; "
                          f"C2: synthetic game, {steps_moves} moves; connect4: real rules, C4 network")
        out["cpu_baseline_steps"] = legs
    return out


# ----------------------------------------------------------------------------- side legs (N = 1)
def observation_stacker_leg(backend, games=64, stacked=32, iters=20):
    """
    The HBM-bound kernel beside the search: mzx_obs_stack at the games/atari.py
    geometry (3x96x96 frames, 32 stacked observations -> 131 planes per sample).  Algorithmic bytes per
    launch = every frame plane read once + every output plane written once; HIP events on the launch stream.
    """
    from mzx import configs, observations
    cfg = configs.HotPathConfig(observation_shape=(3, 96, 96), stacked_observations=stacked, action_space=list(range(4)))
    store = observations.FrameStore(cfg, games, backend)
    rs = numpy.random.RandomState(0)
    for t in range(stacked + 2):
        store.push(rs.rand(games, 3, 96, 96).astype(numpy.float32), None if t == 0 else rs.randint(0, 4, size=games))
    store.stacked()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        store.stacked()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / iters
    planes_in, planes_out = 3 * (stacked + 1), 3 * (stacked + 1) + stacked
    nbytes = games * (planes_in + planes_out) * 96 * 96 * 4
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "mzx_obs_stack", "bound": "hbm",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": nbytes, "launch_ms": ms, "games": games, "stacked_observations": stacked}


def selfplay_leg(cfg, net, B, moves, batched=False, game="synthetic", lockstep=False, pipeline=None, fused=True, native=False):
    """
    What a user of the drop-in engine sees per process -- game stepping, per-game numpy-compatible streams (native
    bank), temperature sampling and GameHistory records around one batched search per move.  Default:
    SelfPlay(num_games=B).play_rounds -- every slot of the shard is one reference actor whose next game starts the
    moment one ends (self_play.py:31-52), so every search runs at full width; `rounds` rounds = B x rounds self-play
    steps.  lockstep=True: play_games, one whole shard of games searched as a thinning batch until its longest game ends
    (rounds 1-3's number, kept as the A/B).  game = "synthetic": the fixed-shape synthetic game, `moves` moves per game;
    game = "connect4" / "tictactoe": the real rules (mzx.games: per-object classes with the reference plugin surface, or
    the batched protocol), whole games to their natural end.  pipeline: config.self_play_pipeline (None = the engine's
    default: per-object shards from 1024 games on run as two slot groups that take turns on the GPU, one searched while the
    host steps the other's Game objects; `search_share` is then search time / wall with the two overlapping).
    native=True (round 6): the game steps inside the library (mzx.games.NativeBatchedGame) and play_rounds is ONE call of
    mzx_selfplay_rounds per shard's worth of games -- no interpreter statement per move.
    """
    import copy

    from mzx import games as board_games
    from mzx import self_play, synthetic

    c = copy.copy(cfg)
    c.self_play_pipeline = pipeline
    if game == "synthetic":
        c.max_moves = moves
        make = (board_games.make_native_synthetic_game if native else
                synthetic.make_synthetic_batched_game if batched else synthetic.make_synthetic_game)
        Game = make(c.observation_shape, len(c.action_space), len(c.players))
        rounds = moves
    else:
        Game = (board_games.NATIVE if native else board_games.BATCHED if batched else board_games.PER_OBJECT)[game]
        rounds = c.max_moves          # as many rounds as the longest possible game (connect4: 42)
    sp = self_play.SelfPlay({"weights": net.get_weights()}, Game, c, 0, num_games=B, _backend=net.backend)
    sp.engine.fused_move = fused     # (False: the A/B of mzx_selfplay_search / mzx_selfplay_select, one slot group only)
    if lockstep:
        sp.play_games(1.0, None, False, "self", 0)          # warm-up (allocations, kernel attributes)
    else:
        sp.play_rounds(1.0, None, min_games=1 << 60, max_rounds=2)
    sp.stats = {"searches": 0, "simulations": 0, "search_seconds": 0.0}
    t0 = time.perf_counter()
    if lockstep:
        histories = sp.play_games(1.0, None, False, "self", 0)
    else:
        histories = sp.play_rounds(1.0, None, min_games=1 << 60, max_rounds=rounds)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    t1 = time.perf_counter()
    steps = sp.stats["searches"]     # one search = one self-play step of one game (lock-step: finished games are not searched)
    for h in histories:       # batched protocol: per-game records are views until touched; time touching ALL of them
        if hasattr(h, "materialize"):
            h.materialize()
    materialize = time.perf_counter() - t1
    # (compact: the whole line must stay under 8 kB; sims/s = steps/s x num_simulations, the mode is play_rounds unless the key
    # says lock-step)
    return {
        "steps_per_sec": round(steps / wall, 1), "games": B, "games_finished": len(histories),
        "moves_per_finished_game": round(sum(len(h.action_history) - 1 for h in histories) / max(1, len(histories)), 2),
        "game": game, "wall_s": round(wall, 5), "search_share": round(sp.stats["search_seconds"] / wall, 4),
        **({"steps_per_sec_with_all_histories_as_lists": round(steps / (wall + materialize), 1)} if batched else {}),
        "game_protocol": "native rounds (mzx_selfplay_rounds)" if native else "batched" if batched else "B Game objects",
        "slot_groups": len((sp._live or {}).get("groups", ())) or 1,
        **({"native_phase_ms": [round(x * 1e3, 2) for x in sp.stats["native_phase_seconds"]]} if "native_phase_seconds" in sp.stats else {}),
    }


def actor_loop_leg(cfg, net, B, moves, shards=3, native=True):
    """
    ``SelfPlay.continuous_self_play`` itself (self_play.py:31-108) through the batched game protocol: rounds of searches,
    finished games refilled, and the HAND-OFF the self-play legs above do not time -- initial PER priorities of every
    finished game (replay_buffer.py:39-51; games/cartpole.py:99-101: td_steps 50, PER_alpha 0.5), ``save_game`` of a
    buffer that does what the stock one does on that branch (replay_buffer.py:33-65: keep the game, count games /
    steps, report them to the storage) -- until ``shards`` x B games are saved.  In-process storage, no Ray.
    """
    import copy

    from mzx import self_play, shared_storage, synthetic

    c = copy.copy(cfg)
    c.max_moves = moves
    c.PER, c.PER_alpha, c.td_steps = True, 0.5, 50
    c.training_steps, c.ratio, c.self_play_delay = 1 << 60, None, 0
    from mzx import games as board_games
    make = board_games.make_native_synthetic_game if native else synthetic.make_synthetic_batched_game
    Game = make(c.observation_shape, len(c.action_space), len(c.players))
    sp = self_play.SelfPlay({"weights": net.get_weights()}, Game, c, 0, num_games=B, _backend=net.backend)
    sp.play_rounds(1.0, None, min_games=1 << 60, max_rounds=2)          # warm-up (allocations, kernel attributes)
    storage = shared_storage.LocalStorage(training_step=0, terminate=False, weights=net.get_weights(),
                                          num_played_games=0, num_played_steps=0)

    class Buffer:
        def __init__(self):
            self.buffer, self.games, self.steps, self.with_priorities = {}, 0, 0, 0

        def save_game(self, game_history, shared_storage=None):
            self.with_priorities += game_history.priorities is not None
            self.buffer[self.games] = game_history
            self.games += 1
            self.steps += len(game_history.root_values)
            if shared_storage:
                shared_storage.set_info("num_played_games", self.games)
                shared_storage.set_info("num_played_steps", self.steps)
                if self.games >= shards * B:
                    shared_storage.set_info("terminate", True)

    buffer = Buffer()
    sp.stats = {"searches": 0, "simulations": 0, "search_seconds": 0.0}
    t0 = time.perf_counter()
    sp.continuous_self_play(storage, buffer)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return {"steps_per_sec": sp.stats["searches"] / wall, "games_saved": buffer.games, "with_priorities": buffer.with_priorities,
            "steps_saved": buffer.steps, "search_share": sp.stats["search_seconds"] / wall,
            "game_protocol": "native rounds (mzx_selfplay_rounds) + device priorities" if native else "batched (Python game) + device priorities",
            **({"main_thread_ms": [round(x * 1e3, 1) for x in sp.stats["handoff_seconds"]],
                "library_call_ms": round(sum(sp.stats.get("native_phase_seconds", [0.0])[:6]) * 1e3, 1)} if "handoff_seconds" in sp.stats else {})}


# ----------------------------------------------------------------------------- weights
def bench_weights(cfg, net, workload, kind):
    """(state_dict, label).  SURVEY.md section 8(d): the weights of the reference's own constructor under
    torch.manual_seed(0) -- models.MuZeroNetwork(config).get_weights(), models.py:7-41, from the oracle/_ref bytecode of the
    unmodified file (the GPU leg only takes the TENSORS; nothing of the reference runs in the timed region)."""
    from mzx import synthetic

    special = WORKLOAD_WEIGHTS.get(workload)
    if special and special.startswith("checkpoint:"):
        z = numpy.load(os.path.join(ROOT, "tests", "golden", special.split(":", 1)[1]), allow_pickle=True)
        sd, off = {}, 0
        for k, t in net.state_dict().items():
            if t.dtype.is_floating_point:
                sd[k] = torch.from_numpy(z["flat_weights"][off:off + t.numel()].reshape(tuple(t.shape)).copy())
                off += t.numel()
        assert off == z["flat_weights"].size
        return sd, "results/cartpole/model.checkpoint of the reference (tests/golden fixture)"
    if kind == "reference":
        try:
            from oracle import build_ref
            if build_ref.available():
                ref_models, _ = build_ref.load()
                torch.manual_seed(0)
                model = ref_models.MuZeroNetwork(cfg)
                return {k: v.clone() for k, v in model.get_weights().items()}, "reference constructor, torch.manual_seed(0)"
        except Exception as e:      # (oracle/_ref missing or built for another interpreter: say so in the line)
            print(f"bench: reference constructor unavailable ({e}); synthetic weights", file=sys.stderr)
    return synthetic.fill_state_dict(net.state_dict(), 0), "synthetic seed 0 (mzx.synthetic.fill_state_dict)"


def _checksum(t):
    """Order-independent 64-bit checksum of a tensor's bits (sum of the int32 words as int64 + a weighted sum)."""
    w = t.detach().contiguous().view(torch.int32).to(torch.int64).flatten()
    idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
    return int(w.sum().item()), int((w * (idx % 65521)).sum().item())


# ----------------------------------------------------------------------------- the search workload
def run_search_workload(env, args, workload, steps, warmup, trees=None, solo_reference=False, compact=False):
    """
    Times `steps` passes of the hot path of `workload` on every rank (contract bracket: barrier +
    synchronize on both sides, MAX over ranks).  Returns (rank-0 result dict, cfg, net).
    """
    from mzx import configs, models, self_play, shared_storage, synthetic

    name, overrides, default_trees, description = WORKLOADS[workload]
    cfg = configs.BY_NAME[name](**overrides)
    B = trees or default_trees
    if B != default_trees:       # --trees: the label follows what ran
        description = description.replace(f"{default_trees} trees", f"{B} trees")
    S, A = cfg.num_simulations, len(cfg.action_space)
    rank, world = env.rank, env.world

    net = models.MuZeroNetwork(cfg, _backend=env.backend)
    # rank 0 holds the "trainer's" weights; every other rank starts from DIFFERENT ones and receives rank 0's through the
    # RCCL broadcast of the flat buffer (the reference's per-game weight pull, self_play.py:37)
    weights, weights_label = bench_weights(cfg, net, workload, args.weights)
    if rank != 0:
        weights = synthetic.fill_state_dict(net.state_dict(), 100 + rank)
    net.set_weights(weights)
    env.fence()
    t_b0 = time.perf_counter()
    shared_storage.broadcast_weights(net, src=0)
    env.sync()
    broadcast_ms = (time.perf_counter() - t_b0) * 1e3
    # self-validation (N > 1): every rank must now hold rank 0's buffer -- checksums of the flat device buffer, gathered
    # and compared on every rank; a wrong broadcast fails the run instead of producing a plausible number
    flat_sum = _checksum(net.flat_weights()) if hasattr(net, "flat_weights") else (0, 0)
    if world > 1:
        sums = [None] * world
        torch.distributed.all_gather_object(sums, flat_sum)
        if any(x != sums[0] for x in sums):
            raise SystemExit(f"rank {rank}: weight buffers differ after the broadcast: {sums}")

    net_mode = WORKLOAD_NET_MODE.get(workload, "fused") if args.net_mode == "fused" else args.net_mode
    if net_mode == "per-operator":
        net.set_mode(0)
    elif net_mode == "fused-4wave" and net.fused_supported():
        net.set_mode(2)
    elif net_mode == "streamed" and cfg.network == "resnet":
        net.set_mode(3)          # every layer on the streamed MFMA engine (A/B against the LDS-resident engine)
    streamed = cfg.network == "resnet" and net_mode != "per-operator" and (
        net_mode == "streamed" or bool(net.streamed_supported()))
    net_fused = bool(net.fused_supported()) and net_mode not in ("per-operator", "streamed")
    mode = {"auto": None, "generic": 0, "fused": 1, "fused-v1": 17}[args.mode]
    engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
    handle = engine.handle(B)
    fused_kind = int(engine.backend.lib.mzx_search_fused_supported(handle)) if mode != 0 else 0
    if fused_kind == 2 and not net_fused:
        fused_kind = 0
    fused = fused_kind == 1
    be, lib = engine.backend, engine.backend.lib
    tuning = dict(WORKLOAD_TUNING.get(workload, {}))
    tuning.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tuning.split(",") if kv})
    tuning_saved = {k: lib.tuning_get(k) for k in tuning}
    for k, v in tuning.items():      # (restored below)
        lib.tuning_set(k, v)

    # synthetic inputs, resident in HBM: a few distinct input sets rotated over the steps
    c_in = cfg.observation_shape[0] * (cfg.stacked_observations + 1) + cfg.stacked_observations
    obs_shape = (c_in,) + tuple(cfg.observation_shape[1:])
    obs_bytes = 4 * B * int(numpy.prod(obs_shape))
    # large inputs (games/atari.py: 131 x 96 x 96 floats per tree, 4.9 GB at 1024 trees): uniform [0, 1) observations drawn
    # ON the device from a seeded generator instead of 10 GB of host doubles per set, and as many sets as steps are run
    big = obs_bytes > (1 << 30) and not env.dry
    n_sets = min(4, max(1, steps, warmup)) if big else 4
    sets = []
    for k in range(n_sets):
        if big:
            gen = torch.Generator(device="cuda")
            gen.manual_seed(123 + 7919 * k + rank)
            obs = torch.rand((B, int(numpy.prod(obs_shape))), device="cuda", generator=gen)
        else:
            obs = synthetic.observations(B, obs_shape, seed=123 + 7919 * k + rank)
        legal = numpy.tile(numpy.arange(A, dtype=numpy.int32), (B, 1))
        noise = numpy.zeros((B, A))
        tape = numpy.zeros((B, self_play.TAPE_WORDS), numpy.uint32)
        for i in range(B):
            rs = numpy.random.RandomState(1000 + (rank * B + i) + 104729 * k)
            noise[i] = rs.dirichlet([cfg.root_dirichlet_alpha] * A)
            tape[i] = rs.randint(0, 2 ** 32, size=self_play.TAPE_WORDS, dtype=numpy.uint32)
        sets.append(engine.make_io(B, obs.reshape(B, -1), legal, numpy.zeros(B, numpy.int32), noise, tape))
        # make_io reuses its output tensors: give every set its own
        sets[-1][1].update({k2: v.clone() for k2, v in sets[-1][1].items()})
        io = sets[-1][0]
        io.d_visit_counts, io.d_root_value = be.ptr(sets[-1][1]["visits"]), be.ptr(sets[-1][1]["root_value"])
        io.d_root_predicted_value, io.d_info = be.ptr(sets[-1][1]["predicted"]), be.ptr(sets[-1][1]["info"])
    arena = engine.arena(B)
    env.sync()

    def step(k):
        io = sets[k % n_sets][0]
        lib.check(lib.mzx_search_run(handle, ctypes.byref(io), be.ptr(arena), arena.numel(), be.stream()))

    def timed(n):
        watch = Stopwatch(env, n)
        t0 = time.perf_counter()
        for k in range(n):
            watch.start(k)
            step(k)
            watch.stop(k)
        env.sync()
        return time.perf_counter() - t0, watch

    for k in range(warmup):
        step(k)
    search_check = None
    if world > 1:
        # one search on inputs that do NOT depend on the rank (rank 0's first set: seeds without the rank term): with equal
        # weights every rank must build the same trees -- visit counts and root values compared as checksums on every rank
        common = sets[0] if rank == 0 else None
        obs_c = synthetic.observations(B, obs_shape, seed=123) if not big else None
        if obs_c is None:
            gen = torch.Generator(device="cuda")
            gen.manual_seed(123)
            obs_c = torch.rand((B, int(numpy.prod(obs_shape))), device="cuda", generator=gen)
        noise_c = numpy.zeros((B, A))
        tape_c = numpy.zeros((B, self_play.TAPE_WORDS), numpy.uint32)
        for i in range(B):
            rs = numpy.random.RandomState(1000 + i)
            noise_c[i] = rs.dirichlet([cfg.root_dirichlet_alpha] * A)
            tape_c[i] = rs.randint(0, 2 ** 32, size=self_play.TAPE_WORDS, dtype=numpy.uint32)
        io_c, out_c, _keep = engine.make_io(B, obs_c.reshape(B, -1), numpy.tile(numpy.arange(A, dtype=numpy.int32), (B, 1)),
                                            numpy.zeros(B, numpy.int32), noise_c, tape_c)
        lib.check(lib.mzx_search_run(handle, ctypes.byref(io_c), be.ptr(arena), arena.numel(), be.stream()))
        env.sync()
        mine = (_checksum(out_c["visits"]), _checksum(out_c["root_value"]))
        alls = [None] * world
        torch.distributed.all_gather_object(alls, mine)
        if any(x != alls[0] for x in alls):
            raise SystemExit(f"rank {rank}: the same search on the same inputs gave different trees across ranks: {alls}")
        search_check = {"visit_count_checksum": mine[0][0], "ranks_equal": True}
        del common
    devs = None
    if world > 1:
        devs = [None] * world
        torch.distributed.all_gather_object(devs, "cpu" if env.dry else torch.cuda.get_device_name(torch.cuda.current_device()))
    solo = None
    if solo_reference and world > 1:
        # rank 0 alone, every other GPU idle: the N = 1 rate on this very box, for the weak-scaling ratio
        env.fence()
        if rank == 0:
            solo_elapsed, _ = timed(steps)
            solo = B * S * steps / solo_elapsed
    # R timed blocks of exactly `steps` steps, each bracketed by barrier + synchronize on both sides and reduced with
    # MAX over the ranks; the line reports the MEDIAN block (one 3 ms sample says little), min / max beside it
    blocks = []
    repeats = args.repeats
    while True:
        env.fence()
        own_elapsed, watch = timed(steps)
        env.barrier()
        env.sync()
        elapsed = env.all_max(own_elapsed)
        blocks.append((elapsed, own_elapsed, watch.mean_ms(), env.all_gather(own_elapsed)))
        if repeats is None:      # decided from the first block, identically on every rank (elapsed is the all-rank max)
            repeats = 25 if elapsed < 0.1 else (9 if elapsed < 0.5 else (3 if elapsed < 3.0 else 1))
        if len(blocks) >= repeats:
            break
    order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
    elapsed, own_elapsed, launch_ms, gathered = blocks[order[len(order) // 2]]
    per_rank = [B * S * steps / t for t in gathered]
    block_rates = [world * B * S * steps / b[0] for b in blocks]

    # sanity: the timed work is real (every tree ran S simulations, no flags)
    for _, out, _ in sets[: 0 if tuning.get("rt_dbg") else min(n_sets, max(steps, warmup, 1))]:
        visits, info = out["visits"].cpu().numpy(), out["info"].cpu().numpy()
        assert (visits.sum(1) == S).all(), "a tree did not complete its simulations"
        assert (info[:, 1] == 0).all(), "search flagged an overflow"
    mean_leaf_depth = float(sets[0][1]["info"].cpu().numpy()[:, 3].mean() / S)
    if rank != 0:
        for k, v in tuning_saved.items():
            lib.tuning_set(k, v)
        return None, cfg, net

    value = world * B * S * steps / elapsed
    Hf, L = net.hidden_size, mean_leaf_depth
    bytes_per_sim = 28 * A * L + 29 * (L + 1) + (8 * A + 16) + 8 * Hf
    achieved = bytes_per_sim * B * S / (launch_ms * 1e-3) / 1e9
    ran = lib.mzx_search_kernel_name(handle)      # the search kernel the last step launched
    ran = ran.decode() if ran else ""
    route = (ctypes.c_int32 * 8)()
    lib.check(lib.mzx_search_route(handle, route))      # (under the workload's tuning: restored right below)
    for k, v in tuning_saved.items():
        lib.tuning_set(k, v)
    # short tag of what ran: the search kernel (+ its shape where the library plans one per shard)
    if route[0] == 3:
        tag = f"mzx::rt_search_kernel<{route[2]},1> {route[1]} trees x {route[6]} threads, {route[4]} per CU"
    elif route[0] == 2 or "row_select_kernel" in ran:
        tag = "launches: row_select / rb_tower x2 / rb_gemm_multi x2 / row_expand" + (", two streams" if "two half-shards" in ran else "")
    elif ran.startswith("mzx::"):
        tag = ran
    else:
        tag = "one kernel per operator"
    traffic, traffic_source = None, None
    try:  # PMC-measured HBM bytes per STEP of this workload (muzero-general_amd/tools/pmc_traffic.py from separate rocprofv3 --pmc passes)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            table = json.load(f)
        entry = table.get(workload)
        if entry and B == default_trees and entry.get("kernel_tag", tag).split(" ")[0] == tag.split(" ")[0]:
            traffic = entry["bytes_per_step"]
            # NOT an observation of this run: PMC counters need their own rocprofv3 --pmc passes, so the figure is the one
            # the builder collected for this workload and kernel on the commit named here
            traffic_source = {"file": "profiles/pmc_traffic.json", "commit": table.get("_collected_at_commit", "unknown"),
                              "how": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (gpu_job.sh pmc), not this run"}
    except (OSError, ValueError):
        pass
    roofline = {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": bytes_per_sim * B * S,
        "kernel": tag, "launch_ms": launch_ms, "algorithmic_bytes_per_sim": bytes_per_sim,
    }
    if cfg.network == "resnet":  # dense contractions: FP32 MFMA roofline
        f_init = int(lib.mzx_net_flops(net.handle, 0))
        f_rec = int(lib.mzx_net_flops(net.handle, 1))
        flops = B * (f_init + S * f_rec)
        tf = flops / (launch_ms * 1e-3) / 1e12
        roofline = {
            "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": "whole step: " + tag,
            "launch_ms": launch_ms, "flops_per_simulation": f_rec, "flops_initial_inference": f_init, "flops_per_step": flops,
        }
    instantiations = None
    if route[0] == 2:     # WHICH rb_tower_kernel / rb_gemm_kernel<MT, NT> instantiations this workload launched (host-side planner)
        first, second = int(route[6]), int(route[7])
        launches = net.streamed_launches(0, B)
        for b in sorted({first, second} - {0}):
            launches += net.streamed_launches(1, b)
        instantiations = {"half_shards": [first, second], "kernels": models.summarize_launches(launches)}
    result = {
        "metric": "mcts_simulations_per_sec", "value": value, "unit": "sims/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": description, "trees_per_gpu": B, "num_simulations": S, "action_space": A,
            "network": cfg.network, "search_kernel": tag, "tree_statistics_dtype": "f64",
            "weights": weights_label + " (RCCL-broadcast flat buffer)", "mean_leaf_depth": L,
        },
        "repeats": {"n": len(blocks), "steps_per_block": steps, "median": sorted(block_rates)[len(block_rates) // 2],
                    "min": min(block_rates), "max": max(block_rates), "unit": "sims/s"},
        "search_steps_per_sec": world * B * steps / elapsed,
        "weight_broadcast_ms": broadcast_ms,
        "collective_world_size": (torch.distributed.get_world_size() if world > 1 else 1),
        "collective_backend": (torch.distributed.get_backend() if world > 1 else None),
        "roofline": roofline,
    }
    if world > 1:       # what the collective ran on, and that it did what it claims (checked above, on every rank)
        result["collective"] = {
            "library_version": None if env.dry else ".".join(str(v) for v in torch.cuda.nccl.version()),
            "devices": devs, "weights_equal_after_broadcast": True, "weights_checksum": flat_sum[0],
            "same_search_same_trees_on_every_rank": search_check,
        }
    if compact:      # an entry of "workloads": the same measurement, short keys (the whole line must fit the driver's stdout tail)
        result = {
            "w": workload, "v": round(value, 1), "ms": round(elapsed / steps * 1e3, 4), "B": B, "S": S, "L": round(L, 2), "k": tag,
            "min": round(min(block_rates), 1), "max": round(max(block_rates), 1), "n": len(blocks),
            "roofline": {"bound": roofline["bound"], "achieved": round(roofline["achieved"], 3), "frac": round(roofline["frac"], 4),
                         "traffic": traffic, "launch_ms": round(launch_ms, 4)},
        }
        if world > 1:        # every rank's own time per step of the median block (the first multi-GPU run yields the scaling of every workload)
            result["per_rank_ms"] = [round(t / steps * 1e3, 4) for t in gathered]
        if WORKLOAD_WEIGHTS.get(workload):
            result["weights"] = weights_label
        if instantiations:
            result["inst"] = "; ".join(instantiations["kernels"]) if isinstance(instantiations["kernels"], (list, tuple)) else instantiations["kernels"]
            result["halves"] = instantiations["half_shards"]
    elif instantiations:
        result["config"]["instantiations"] = instantiations
    if world > 1:
        result["per_rank"] = {"sims_per_sec": per_rank, "min": min(per_rank), "max": max(per_rank),
                              "ms_per_step": [t / steps * 1e3 for t in gathered]}
        if solo is not None:
            result["single_gpu_reference"] = {
                "sims_per_sec": solo, "weak_scaling_efficiency": value / (world * solo),   # rank 0 timed alone on the same box
            }
    return result, cfg, net


def _compact_floats(obj, top=True):
    """Floats below the top level to six significant digits (the line must fit the driver's 8 kB tail; `value`, `ms_per_step`
    and the other top-level scalars keep every digit)."""
    if isinstance(obj, dict):
        return {k: (v if top and isinstance(v, float) else _compact_floats(v, False)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_compact_floats(v, False) for v in obj]
    if isinstance(obj, float) and obj == obj and abs(obj) != float("inf"):
        return float(f"{obj:.6g}")
    return obj


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus)
    env = Env(args)
    if args.gpus != env.world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env.world}")
    also = args.also
    if also is None:
        also = DEFAULT_ALSO if (args.workload == DEFAULT_WORKLOAD and not args.dry_run and args.trees is None) else "none"
    also = [w for w in also.split(",") if w and w != "none"]

    line, cfg, net = run_search_workload(env, args, args.workload, args.steps, args.warmup, args.trees,
                                         solo_reference=True)
    others = []
    c4_net = None
    for w in also:
        if w not in WORKLOADS:
            raise SystemExit(f"unknown workload {w}")
        # one step of games/gomoku.py as shipped is 1024 x 400 simulations of a 128-channel network (~3.5 s)
        r, c2_, n2 = run_search_workload(env, args, w, 1 if w in ("gomoku", "atari") else args.also_steps, 1, compact=True)
        if w == "c4":
            c4_net = (c2_, n2)
        else:
            del n2
        if r is not None:
            others.append(r)
    if args.workload == "c4":
        c4_net = (cfg, net)
    # the search workloads leave cyclic garbage that holds gigabytes of pinned host and device buffers (games/atari.py: 4.9 GB
    # of observations per input set); collected HERE, not by a generation-2 pass in the middle of a timed self-play leg
    # (round 6: the actor loop measured 4.1 M steps/s behind the default workloads and 9.4 M alone -- 73 ms of its hand-off
    # were such a pass freeing those buffers)
    import gc
    gc.collect()
    if env.rank == 0:
        if others:
            line["workloads"] = others
            # (every workloads[*].roofline.traffic comes from the same file as the main roofline's: see roofline.traffic_source)
            line["workloads_traffic_source"] = "profiles/pmc_traffic.json, as roofline.traffic_source"
        if args.dry_run:
            line["dry_run"] = True
            line["data"] = "DRY RUN on the CPU test double (gloo): plumbing check, NOT a measurement"
        B = args.trees or WORKLOADS[args.workload][2]
        if env.world == 1 and args.selfplay_moves > 0:
            line["selfplay_end_to_end"] = selfplay_leg(cfg, net, B, args.selfplay_moves)
            line["selfplay_end_to_end_batched_game"] = selfplay_leg(cfg, net, B, args.selfplay_moves, batched=True)
            try:       # the round loop inside the library (natively stepped game), and the actor loop with the replay hand-off
                line["selfplay_end_to_end_native_rounds"] = selfplay_leg(cfg, net, B, args.selfplay_moves, batched=True, native=True)
                line["selfplay_actor_loop"] = actor_loop_leg(cfg, net, B, args.selfplay_moves)
                line["selfplay_actor_loop"]["python_game_steps_per_sec"] = actor_loop_leg(cfg, net, B, args.selfplay_moves,
                                                                                           native=False)["steps_per_sec"]
            except Exception as e:      # noqa: BLE001  (a failure here must not cost the line)
                line["selfplay_actor_loop"] = {"error": repr(e)[:200]}
            if not args.dry_run:
                # A/B of round 5's host path: ONE slot group (no overlap of the host with the search), and one group on the
                # separate calls of rounds 1-4 (root_draws / upload / mzx_search_run / download / advance / numpy action draw)
                line["selfplay_end_to_end_batched_game"]["one_group_steps_per_sec"] = selfplay_leg(
                    cfg, net, B, args.selfplay_moves, batched=True, pipeline=False)["steps_per_sec"]
                line["selfplay_end_to_end_batched_game"]["separate_calls_steps_per_sec"] = selfplay_leg(
                    cfg, net, B, args.selfplay_moves, batched=True, pipeline=False, fused=False)["steps_per_sec"]
            if c4_net is not None and not args.dry_run:
                # the Connect4 half of BASELINE.json's "self-play steps/sec": whole games with the real rules
                # (games/connect4.py:125-346 semantics, mzx.games), C4 network, 1024 games per process
                c4_cfg, c4_model = c4_net
                line["selfplay_end_to_end_connect4"] = selfplay_leg(c4_cfg, c4_model, WORKLOADS["c4"][2], 0, game="connect4")
                # (the same with ONE slot group: search, then step every Game object, in turn -- the A/B of the pipelining)
                line["selfplay_end_to_end_connect4"]["one_group_steps_per_sec"] = selfplay_leg(
                    c4_cfg, c4_model, WORKLOADS["c4"][2], 0, game="connect4", pipeline=False)["steps_per_sec"]
                line["selfplay_end_to_end_connect4_batched_game"] = selfplay_leg(c4_cfg, c4_model, WORKLOADS["c4"][2], 0,
                                                                                 batched=True, game="connect4")
                # rounds 1-3 played whole shards in lock-step (the batch thins out while the longest game ends): the A/B
                line["selfplay_end_to_end_connect4_batched_game"]["lockstep_steps_per_sec"] = selfplay_leg(
                    c4_cfg, c4_model, WORKLOADS["c4"][2], 0, batched=True, game="connect4", lockstep=True)["steps_per_sec"]
                line["selfplay_end_to_end_connect4_batched_game"]["native_rounds_steps_per_sec"] = selfplay_leg(
                    c4_cfg, c4_model, WORKLOADS["c4"][2], 0, batched=True, game="connect4", native=True)["steps_per_sec"]
                # (the shard size is the actor's choice: 1536 games = six boards per workgroup of rt_search_kernel, whole rounds of
                # 256 workgroups -- the shape the tower kernel runs best on)
                line["selfplay_end_to_end_connect4_batched_game"]["native_rounds_1536_games_steps_per_sec"] = selfplay_leg(
                    c4_cfg, c4_model, 1536, 0, batched=True, game="connect4", native=True)["steps_per_sec"]
            if not args.dry_run:
                line["observation_stacker"] = observation_stacker_leg(net.backend)
        line["cpu_baseline"] = None
        if env.world == 1 and args.cpu_seconds > 0 and not args.dry_run:
            from oracle import build_ref
            # the UNMODIFIED reference when its bytecode travelled (oracle/_ref), else the oracle's restatement stands in;
            # with it also the Connect4 half of the metric and self-play steps/s
            line.update(cpu_baselines(args.workload, args.cpu_seconds, args.cpu_cores,
                                      "reference" if build_ref.available() else "port", args.weights,
                                      c4=c4_net is not None, steps_moves=args.selfplay_moves))
        print(json.dumps(_compact_floats(line)), flush=True)
    env.barrier()
    env.close()


if __name__ == "__main__":
    main()
