"""
bench.py -- BASELINE.json metric: MCTS simulations/s (whole job) and self-play steps/s.

    python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic input: B
independent roots, each doing initial_inference + num_simulations x {select,
recurrent_inference, expand, backpropagate} (one self-play move per tree,
self_play.py:144-150).  Default workload = BASELINE config C2: CartPole
FullyConnectedNetwork, 4096 trees x 50 simulations per GPU.  Inputs (stacked
observations, legal actions, Dirichlet noise, tie tape, weights) are resident in
HBM before the timed region; every rank owns an independent shard of trees (weak
scaling, no data-path collective; RCCL only broadcasts the flat weight buffer
once, outside the timed region, as the reference's weight pull self_play.py:37).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     algorithmic tree bytes (SURVEY.md section 8d formula, with the measured mean
                  leaf depth) / HIP-event time of the search launch, vs the HBM peak
  "observation_stacker": the path's HBM-bound kernel (mzx_obs_stack, atari geometry) against the HBM peak
  "cpu_baseline": the CPU oracle (oracle/*.py: the reference's per-node algorithm and its
                  batch-1 torch network, kind "port") timed on this box's host cores on a
                  bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "muzero-general_amd"))
sys.path.insert(0, ROOT)

# dmabuf IPC is the only mode the host driver supports: RCCL between the ranks of a node needs it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3  # dense FP32-input MFMA peak (= FP32 vector peak), same guide

WORKLOADS = {
    # name: (config factory name, overrides, trees per GPU, description)
    "c2": ("cartpole", {}, 4096, "C2 CartPole FullyConnectedNetwork, 4096 trees x 50 sims per GPU"),
    "c3": ("tictactoe", {}, 1024, "C3 Tic-tac-toe MuZeroResidualNetwork, 1024 trees x 25 sims per GPU"),
    "c4": ("connect4", {}, 1024, "C4 Connect4 ResNet, 1024 trees x 200 sims per GPU"),
    "c5": ("breakout", {"num_simulations": 50}, 64, "C5 Breakout ResNet (resnet stem), 64 trees x 50 sims per GPU"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--trees", type=int, default=None, help="trees per GPU (default: the workload's)")
    ap.add_argument("--mode", default="auto", choices=["auto", "generic", "fused"])
    ap.add_argument("--net-mode", default="fused", choices=["fused", "fused-4wave", "per-operator"],
                    help="residual networks: fused MFMA engine (default) or one kernel per operator")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=None)
    ap.add_argument("--selfplay-moves", type=int, default=4,
                    help="moves of the end-to-end self-play leg (SelfPlay(num_games=B) on the synthetic game; 0 = skip)")
    return ap.parse_args()


# ----------------------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    """One host core: the oracle's per-node MCTS with its batch-1 torch network (the reference's CPU algorithm)."""
    workload, worker, seconds = args
    torch.set_num_threads(1)
    from mzx import configs, synthetic
    from oracle import mcts_oracle, net_oracle

    name, overrides, _, _ = WORKLOADS[workload]
    cfg = configs.BY_NAME[name](**overrides)
    template = _state_dict_template(cfg)
    sd = synthetic.fill_state_dict(template, 0)
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


_TEMPLATES = {}


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
    return {"kernel": "mzx_obs_stack (GameHistory.get_stacked_observations on the device)", "bound": "hbm",
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": nbytes, "launch_ms": ms, "games": games, "stacked_observations": stacked}


def cpu_baseline(workload, seconds, cores):
    import multiprocessing as mp

    cores = cores or min(os.cpu_count() or 1, 64)
    ctx = mp.get_context("spawn")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(workload, w, seconds) for w in range(cores)])
    sims = sum(r[0] for r in res)
    wall = max(r[2] for r in res)
    one = res[0][0] / res[0][2]
    return {
        "value": sims / wall, "unit": "sims/s", "cores": cores, "kind": "port",
        "sample": f"{sum(r[1] for r in res)} searches x {sims // max(1, sum(r[1] for r in res))} sims of the same "
                  f"workload, {cores} processes x {seconds:.0f} s, torch.set_num_threads(1) each",
        "per_core": one,
    }


# ----------------------------------------------------------------------------- end-to-end self-play leg
def selfplay_leg(cfg, net, B, moves, batched=False):
    """
    SelfPlay(num_games=B).play_games on the synthetic fixed-shape game (reference plugin surface): what a
    user of the drop-in engine sees per process -- B Python Game.step calls, per-game numpy-compatible
    streams (native bank), temperature sampling and GameHistory records around one batched search per move.
    """
    import copy

    from mzx import self_play, synthetic

    c = copy.copy(cfg)
    c.max_moves = moves
    make = synthetic.make_synthetic_batched_game if batched else synthetic.make_synthetic_game
    Game = make(c.observation_shape, len(c.action_space), len(c.players))
    sp = self_play.SelfPlay({"weights": net.get_weights()}, Game, c, 0, num_games=B)
    sp.play_games(1.0, None, False, "self", 0)          # warm-up (allocations, kernel attributes)
    sp.stats = {"searches": 0, "simulations": 0, "search_seconds": 0.0}
    t0 = time.perf_counter()
    histories = sp.play_games(1.0, None, False, "self", 0)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    steps = sum(len(h.action_history) - 1 for h in histories)
    return {
        "steps_per_sec": steps / wall, "sims_per_sec": steps * c.num_simulations / wall, "games": B, "moves_per_game": moves,
        "wall_s": wall, "search_share": sp.stats["search_seconds"] / wall,
        "game_protocol": "batched (one object steps the shard)" if batched else "reference plugin surface (B Game objects)",
        "note": "one host process: plugin game stepping + host bookkeeping + batched search "
                "(search_share = fraction of the wall spent inside BatchedMCTS.run incl. uploads/downloads)",
    }


# ----------------------------------------------------------------------------- GPU leg
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mzx import _lib, configs, models, self_play, shared_storage, synthetic

    name, overrides, default_trees, description = WORKLOADS[args.workload]
    cfg = configs.BY_NAME[name](**overrides)
    B = args.trees or default_trees
    S, A = cfg.num_simulations, len(cfg.action_space)

    net = models.MuZeroNetwork(cfg)
    # rank 0 holds the "trainer's" weights; every other rank starts from different ones and receives
    # them through the RCCL broadcast of the flat buffer (the reference's per-game weight pull)
    net.set_weights(synthetic.fill_state_dict(net.state_dict(), 0 if rank == 0 else 100 + rank))
    t_b0 = time.perf_counter()
    shared_storage.broadcast_weights(net, src=0)
    torch.cuda.synchronize()
    broadcast_ms = (time.perf_counter() - t_b0) * 1e3

    if args.net_mode == "per-operator":
        net.set_mode(0)
    elif args.net_mode == "fused-4wave" and net.fused_supported():
        net.set_mode(2)
    net_fused = bool(net.fused_supported()) and args.net_mode != "per-operator"
    mode = {"auto": None, "generic": 0, "fused": 1}[args.mode]
    engine = self_play.BatchedMCTS(cfg, net, B, mode=mode)
    handle = engine.handle(B)
    fused_kind = int(engine.backend.lib.mzx_search_fused_supported(handle)) if mode != 0 else 0
    if fused_kind == 2 and not net_fused:
        fused_kind = 0
    fused = fused_kind == 1
    be, lib = engine.backend, engine.backend.lib

    # synthetic inputs, resident in HBM: a few distinct input sets rotated over the steps
    c_in = cfg.observation_shape[0] * (cfg.stacked_observations + 1) + cfg.stacked_observations
    n_sets = 4
    sets = []
    for k in range(n_sets):
        obs = synthetic.observations(B, (c_in,) + tuple(cfg.observation_shape[1:]), seed=123 + 7919 * k + rank)
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
    torch.cuda.synchronize()

    def step(k):
        io = sets[k % n_sets][0]
        lib.check(lib.mzx_search_run(handle, ctypes.byref(io), be.ptr(arena), arena.numel(), be.stream()))

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for k in range(args.warmup):
        step(k)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step(k)
        ev[k][1].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_ms = float(numpy.mean([a.elapsed_time(b) for a, b in ev]))

    # sanity: the timed work is real (every tree ran S simulations, no flags)
    for _, out, _ in sets[: min(n_sets, max(args.steps, args.warmup, 1))]:
        visits, info = out["visits"].cpu().numpy(), out["info"].cpu().numpy()
        assert (visits.sum(1) == S).all(), "a tree did not complete its simulations"
        assert (info[:, 1] == 0).all(), "search flagged an overflow"
    mean_leaf_depth = float(sets[0][1]["info"].cpu().numpy()[:, 3].mean() / S)

    if rank == 0:
        sims_total = world * B * S * args.steps
        value = sims_total / elapsed
        Hf = net.hidden_size
        L = mean_leaf_depth
        bytes_per_sim = 28 * A * L + 29 * (L + 1) + (8 * A + 16) + 8 * Hf
        achieved = bytes_per_sim * B * S / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        try:  # PMC-measured HBM bytes per launch of this kernel (collected by a separate rocprofv3 --pmc run)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                key = "fused-lds" if fused else ("residual-whole-search" if fused_kind == 2 else "generic-per-op")
                entry = json.load(f).get(f"{args.workload}:{key}")
            if not fused and fused_kind != 2 and net_fused:
                entry = None
            if entry and B == default_trees:
                traffic, traffic_src = entry["bytes"], entry["source"]
        except (OSError, ValueError):
            pass
        kernel_name = ("fused-lds" if fused else
                       "residual whole-search kernel (arena trees, fused MFMA network)" if fused_kind == 2 else
                       "per-sim launches + fused-mfma network" if net_fused else "generic-per-op")
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": bytes_per_sim * B * S,
            "kernel": "mzx::fused_fc_search" if fused else "whole step (one kernel per operator)",
            "launch_ms": launch_ms, "algorithmic_bytes_per_sim": bytes_per_sim,
            "note": "tree bytes per simulation x B x S / HIP-event time of one search launch; trees live in "
                    "LDS in the fused kernel, so this is algorithmic traffic, not HBM traffic (DESIGN.md)",
        }
        if cfg.network == "resnet":  # dense contractions: FP32 MFMA roofline
            f_init = int(lib.mzx_net_flops(net.handle, 0))
            f_rec = int(lib.mzx_net_flops(net.handle, 1))
            flops = B * (f_init + S * f_rec)
            tf = flops / (launch_ms * 1e-3) / 1e12
            roofline = {
                "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "whole step: root kernels + mzx::rz_search_kernel (all simulations, one launch)" if fused_kind == 2
                          else "whole step: select / rz_network_kernel (fused MFMA network) / expand+backprop per simulation"
                          if net_fused else "whole step (one kernel per operator)",
                "launch_ms": launch_ms, "flops_per_simulation": f_rec, "flops_initial_inference": f_init,
                "flops_per_step": flops, "tree_bytes_per_sim": bytes_per_sim,
                "note": "network FLOPs (2 x MAC of every conv / linear layer, as the reference's modules count) of one "
                        "step / HIP-event time of the step, vs the dense FP32-input MFMA peak",
            }
        line = {
            "metric": "mcts_simulations_per_sec", "value": value, "unit": "sims/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": description, "trees_per_gpu": B, "num_simulations": S, "action_space": A,
                "network": cfg.network, "search_kernel": kernel_name,
                "tree_statistics_dtype": "f64", "weights": "synthetic seed 0 (RCCL-broadcast flat buffer)",
                "mean_leaf_depth": L,
            },
            "search_steps_per_sec": world * B * args.steps / elapsed,
            "weight_broadcast_ms": broadcast_ms,
            "roofline": roofline,
        }
        if world == 1 and args.selfplay_moves > 0:
            line["selfplay_end_to_end"] = selfplay_leg(cfg, net, B, args.selfplay_moves)
            line["selfplay_end_to_end_batched_game"] = selfplay_leg(cfg, net, B, args.selfplay_moves, batched=True)
        if world == 1 and args.selfplay_moves > 0:
            line["observation_stacker"] = observation_stacker_leg(net.backend)
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(args.workload, args.cpu_seconds, args.cpu_cores)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
