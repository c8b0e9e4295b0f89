#!/usr/bin/env python3
"""bench.py -- conformers/sec of the AGDIFF diffusion-sampling hot path on MI355X.

A "step" is one Langevin denoising step (score-network forward + update, dualenc.py:478-545) over one packed batch of
synthetic GEOM-Drugs-shaped conformers.  `value` = conformers generated per second by a 5000-step sampling job =
G_total / (ms_per_step * 5000 / 1000), whole job over all ranks, inputs resident in HBM when the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      N > 1 without a launcher: this process never touches a GPU; it starts N fresh rank processes (one per GPU, RCCL over
      127.0.0.1) and relays rank 0's JSON line.  Under `python -m torch.distributed.run --nproc-per-node N` (the driver's
      form) RANK / LOCAL_RANK / WORLD_SIZE come from the environment.

Workloads:
  drugs200 (default)  BASELINE.json configs[2] as SURVEY §8(d) restates it (scripts/test.py:40-61,130-141): 200 Drugs-shaped
                   molecules, 2 x U{50..500} conformers each, packed by the driver's plan_batches (--max-atoms per batch);
                   every batch runs --warmup + --steps steps; ms_per_step = one step of EVERY batch, value = all conformers
                   / (that x 5000).  "saturated" schedule (beta_end = 2e-5: sigma < 0.5 on every step, so the global SchNet
                   branch runs on every step and the radius graph stays at the 32-neighbour cap -- the heaviest per-step
                   work the path can see) is the headline; `extra` carries the reference's default schedule with and without
                   skipping the discarded global branch, and the filter-polynomial fallback (--radius-poly off), each on
                   every 4th batch.
  drugs            `--mols` distinct molecules x `--copies` conformers each in ONE batch per GPU (round 1-2 headline: 8 x 128)
  qm9 | large | alanine   QM9-shaped, 200-atom molecules (configs[4] shape), alanine dipeptide (configs[0]).
Scaling: weak (default; every rank gets its own batches of the same shape) or `--scaling strong` (ONE global job is
cut into contiguous graph ranges by agdiff_amd.dist.shard_graphs, one per rank).  With more than one rank
(or --force-dist) the shards' positions are all-gathered over RCCL after every step.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

JOB_STEPS = 5000
FLOP_PER_EDGE_CFCONV = 2 * (128 * 192 + 128 * 128 + 64 * 64)     # filter MLP of conv1+conv2, one block
PEAK = {"f32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0}      # dense MFMA TFLOP/s (f32-input MFMA; bf16 MFMA), MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0
MFMA_PASSES = {"f32": 1, "bf16x3": 3, "f16x3": 3}        # MFMA FLOPs issued per algorithmic FLOP


def mfma_per_channel_tile(kt, plan, passes):
    """MFMAs k_cfconv_quad issues per 16-channel tile of a 16-row tile (agdiff_params_t.poly_kt k-tiles, poly_plan): plan 0 every
    k-tile all passes; plan 1 two instructions at one k-tile; else plan p all passes for k-tiles 0 .. p - 1
    and one for the others."""
    if not plan:
        return kt * passes
    if plan == 1 and kt == 1:
        return 2
    return plan * passes + (kt - plan)

PROFILE_ROUND = "r06"


def build_batch(kind, mols, copies, seed):
    from agdiff_amd import synth
    if kind == "alanine":          # BASELINE.json configs[0]: one molecule, `copies` conformers (250 in the example)
        return synth.alanine_dipeptide(mols * copies)
    return synth.make_packed_batch(kind, mols, copies, seed=seed)


def make_cfg(kind, schedule):
    from agdiff_amd import drugs_model_config, qm9_model_config
    base = qm9_model_config if kind in ("qm9", "alanine") else drugs_model_config
    return base(beta_end=2e-5) if schedule == "saturated" else base()


# ------------------------------------------------------------------------------------------ launcher (--gpus N)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, selftest=False):
    """`python bench.py --gpus N` without torchrun: N fresh child interpreters, one per GPU, started BEFORE anything in
    this process initialises HIP (torch.cuda.device_count() does not, on this image).  Rank 0's stdout is relayed; any
    child failing makes the whole run fail.  Returns the exit code."""
    if not selftest:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py --gpus %d: only %d GPU(s) visible on this node (torch.cuda.device_count()); "
                  "nothing was measured" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AGDIFF_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others waiting in a collective for ever: watch all of them, and when one exits non-zero stop
    # the rest (torchrun does the same for the driver's launches)
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
        if failed is not None:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    reader.join(timeout=30)
    codes = [p.returncode for p in procs]
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()
    if failed is not None:
        print("bench.py --gpus %d: rank %d exited with code %s; the other ranks were stopped" % (n, failed[0], failed[1]), file=sys.stderr)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py --gpus %d: rank(s) failed: %s" % (n, bad), file=sys.stderr)
        return 1
    return 0


def launcher_selftest(world, rank):
    """Wiring check of spawn_ranks on CPU (tests/test_dist_cpu.py): the ranks meet over gloo, rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    if os.environ.get("AGDIFF_SELFTEST_FAIL_RANK") == str(rank):       # (test hook: a rank that dies before its first collective)
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(x)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": True, "rccl_ranks": world, "sum_of_ranks": int(x.item()),
                          "local_rank_env": os.environ.get("LOCAL_RANK")}), flush=True)


def collective_selfcheck(dist, torch, dev, rank, world, gloo=False):
    """What the process group itself reports, from real collectives (`rccl_ranks` used to be the WORLD_SIZE
    environment value): an all-reduce of ones (= ranks that took part), an all-gather of every rank's device index, the
    backend's name.  Ranks of an RCCL group must sit on distinct devices (RCCL refuses two ranks on one GPU); the one-GPU
    rehearsal over gloo is exempt."""
    where = torch.device("cpu") if gloo else dev
    ones = torch.ones(1, dtype=torch.int64, device=where)
    dist.all_reduce(ones)
    mine = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=where)
    devs = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(devs, mine)
    devices = [int(d.item()) for d in devs]
    rep = {"ranks": int(ones.item()), "world_size": dist.get_world_size(), "backend": dist.get_backend(),
           "devices": devices, "distinct_devices": len(set(devices))}
    if not gloo and rep["distinct_devices"] != world:
        raise RuntimeError("the %d ranks sit on %d distinct devices %s" % (world, rep["distinct_devices"], devices))
    return rep


# ------------------------------------------------------------------------------------------ CPU baseline (oracle)
def cpu_worker(kind, schedule, seed, mols, copies, threads, budget_s):
    """One CPU process of the baseline: the oracle (CPU port of the reference path) on `mols` x `copies`
    conformers with `threads` torch threads; prints one JSON line."""
    import torch
    from oracle import agdiff_oracle as O
    torch.set_num_threads(threads)
    cfg = make_cfg(kind, schedule)
    sd = O.synth_state_dict_for(cfg)
    b = build_batch(kind, mols, copies, seed)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(at.shape[0], 3, generator=g)
    kw = dict(extend_order=False, w_global=1.0, global_start_sigma=0.5, clip=1000.0)
    O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, b["num_graphs"], n_steps=1, **kw)     # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.langevin_dynamics_sample_diffusion(sd, cfg, at, pos, bi, bt, ba, b["num_graphs"], n_steps=1, **kw)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 40:
            break
    print(json.dumps({"G": int(b["num_graphs"]), "atoms": int(at.shape[0]), "steps": n, "s_per_step": el / n,
                      "threads": threads}), flush=True)


def _spawn_cpu_workers(kind, schedule, seed, mols, copies, threads, budget_s, procs):
    """`procs` independent interpreter processes (disjoint molecules: seed + p), started BEFORE this process touches
    the GPU.  Returns their JSON records."""
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads),
               MKL_NUM_THREADS=str(threads))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                            json.dumps([kind, schedule, seed + 7919 * p, mols, copies, threads, budget_s])],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, cwd=ROOT) for p in range(procs)]
    recs = []
    for p in ps:
        out, _ = p.communicate()
        lines = [l for l in out.decode().splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            recs.append(json.loads(lines[-1]))
    return recs


def cpu_baseline(kind, schedule, seed):
    """Two figures from the oracle (kind "port") on the host cores of this box, same synthetic checkpoint and
    schedule as the GPU run:
      single   ONE process x 8 torch threads on a >= 4,400-atom sample (SURVEY §6's size: one Drugs-shaped
               molecule x 100 conformers) -- how the reference's own driver would run on this host;
      value    the WHOLE HOST: P = logical CPUs / 16 processes x 8 threads over disjoint molecules (the path is
               embarrassingly parallel over molecules; one process per 16 logical CPUs = 8 physical cores keeps the
               torch intra-op pools off each other's hyper-threads), aggregate conformers/s.  `cores` = P x 8."""
    ncpu = os.cpu_count() or 1
    thr = min(8, ncpu)
    mols, copies = (1, 250) if kind == "alanine" else (1, 100)
    single = _spawn_cpu_workers(kind, schedule, seed, mols, copies, thr, 12.0, 1)
    procs = max(1, min(32, ncpu // 16))
    many = _spawn_cpu_workers(kind, schedule, seed + 1, mols, copies, thr, 12.0, procs)
    if not single or not many:
        return None
    s = single[0]
    val1 = s["G"] / (s["s_per_step"] * JOB_STEPS)
    val = sum(r["G"] / (r["s_per_step"] * JOB_STEPS) for r in many)
    return {"value": val, "unit": "conformers/s", "cores": len(many) * thr, "kind": "port",
            "sample": "whole host: %d oracle processes x %d torch threads (host has %d logical CPUs), each one "
                      "%s-shaped molecule x %d conformers (%d..%d atoms), %d..%d steps timed after one warm-up step, "
                      "%.2f..%.2f s/step, same synthetic checkpoint and %s schedule, extrapolated to %d steps"
                      % (len(many), thr, ncpu, kind, copies, min(r["atoms"] for r in many), max(r["atoms"] for r in many),
                         min(r["steps"] for r in many), max(r["steps"] for r in many),
                         min(r["s_per_step"] for r in many), max(r["s_per_step"] for r in many), schedule, JOB_STEPS),
            "single_process": {"value": val1, "cores": thr, "atoms": s["atoms"], "s_per_step": s["s_per_step"],
                               "steps": s["steps"]}}


# ------------------------------------------------------------------------------------------ timed GPU runs
def timed_run(model, dev, b, cfg, W, K, schedule, skip, save_traj, seed, rank, use_dist, profile=False, nan_every=64):
    """W untimed + K timed denoising steps of one packed batch; returns (seconds, run, global-branch share, gather,
    (cfconv ms summed over the timed region's bracketed launches, launches) or None)."""
    import torch
    import torch.distributed as dist
    from agdiff_amd import _lib
    lib = _lib.load()
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    at, bi, bt, ba = T(b["atom_type"]), T(b["bond_index"]), T(b["bond_type"]), T(b["batch"])
    g = torch.Generator(device="cpu").manual_seed(seed + rank)
    pos_init = torch.randn(at.shape[0], 3, generator=g).to(dev)
    Tn = cfg.num_diffusion_timesteps
    if schedule == "default" and W + K < Tn:
        # visit the whole schedule so that the share of global-active steps (sigma < global_start_sigma = 0.5: 2012 of the
        # reference's 5000 steps) is the JOB's among the K timed steps: the two ranges are sampled evenly, each with its share
        sig = ((1.0 - model.alphas).sqrt() / model.alphas.sqrt()).detach().cpu().numpy()
        act = np.nonzero(sig < 0.5)[0]
        ina = np.nonzero(sig >= 0.5)[0]
        pick = lambda pool, m: pool[np.linspace(0, pool.size - 1, m).round().astype(int)] if (m > 0 and pool.size) else pool[:0]
        ka = int(round(K * act.size / float(Tn)))
        ka = min(max(ka, 1 if act.size else 0), K - (1 if ina.size else 0))
        wa = int(round(W * act.size / float(Tn)))
        timed = np.sort(np.concatenate([pick(act, ka), pick(ina, K - ka)]))[::-1]
        warm = np.sort(np.concatenate([pick(act, wa), pick(ina, W - wa)]))[::-1]
        idx = warm.tolist() + timed.tolist()
    else:
        idx = list(reversed(range(Tn - (W + K), Tn)))
    gather = None
    if use_dist:
        from agdiff_amd.dist import StepAllGather
        gather = StepAllGather(at.shape[0], dev)
    run = model.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], False, n_steps=W + K, step_lr=1e-6,
                               clip=1000.0, global_start_sigma=0.5, w_global=1.0, step_indices=idx,
                               save_traj=save_traj, skip_discarded_global=skip, nan_check_every=nan_every)
    if gather is not None:
        run.on_step = lambda k, i, pos: gather(k, i, pos, run.ws.nan_flag)
    run.advance(W)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    g0 = run.global_steps
    if profile:
        lib.agdiff_profile_cfconv(1)
    t0 = time.perf_counter()
    run.advance(K)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = None
    if profile:
        lib.agdiff_profile_cfconv(0)
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        _lib.check(lib.agdiff_profile_cfconv_read(ctypes.byref(ms), ctypes.byref(n)), "agdiff_profile_cfconv_read")
        prof = (ms.value, n.value)
    return el, run, (run.global_steps - g0) / max(K, 1), gather, prof


def run_job_steps(mdl, mcfg, batches, confs_of, schedule, sk, W_, K_, prof, *, dev, rank, world, seed, save_traj, use_dist, strong,
                  every=1, gcalls=None, tile_acc=None, timed=None, edges_of=None, tiles_of=None):
    """W_ + K_ denoising steps of every `every`-th packed batch of the job on this rank: (ms of one step summed over the batches,
    conformers, records, global-branch share, last run, CFConv profile sums).  With `use_dist` the ranks' positions (+ NaN
    flag) are all-gathered after EVERY step of every batch -- north_star's collective, weak and strong scaling alike (strong:
    every batch is first cut into per-rank graph ranges); the number of collectives is asserted: one per step.
    `timed` / `edges_of` / `tiles_of`: stand-ins for timed_run / live_edges / tile_stats (tests/test_dist_cpu.py runs this
    function with a CPU sampler stub over gloo)."""
    import torch
    from agdiff_amd import driver
    from agdiff_amd.dist import shard_of
    timed = timed or timed_run
    edges_of = edges_of or live_edges
    tiles_of = tiles_of or tile_stats
    gcalls = gcalls if gcalls is not None else [0]
    tile_acc = tile_acc if tile_acc is not None else {}
    tot_ms, G_local, recs, gl, last = 0.0, 0, [], 0.0, None
    pm = pn = pf = pe = 0.0
    for bidx, bm in enumerate(batches):
        if bidx % every:
            continue
        b = driver.pack_batch(bm, confs_of)
        if strong:
            b, _, _ = shard_of(b, rank, world)
            if b is None:       # more ranks than graphs in this batch: the rank only takes part in the step collectives
                if use_dist:
                    from agdiff_amd.dist import StepAllGather
                    idle = StepAllGather(0, dev)
                    for k in range(W_ + K_):
                        idle(k, k, torch.zeros(0, 3, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))
                    idle.result()
                    gcalls[0] += idle.calls
                recs.append({"molecules": len(bm), "conformers": 0, "atoms": 0, "edges": 0, "ms_per_step": 0.0})
                continue
        last = None
        el, run, gfrac, gather, pr = timed(mdl, dev, b, mcfg, W_, K_, schedule, sk, save_traj, seed + bidx, rank, use_dist,
                                           profile=prof)
        run.check_nan()
        if use_dist:
            assert gather is not None and gather.calls == W_ + K_, "one all-gather per denoising step"
            gcalls[0] += gather.calls
            parts, any_nan = gather.result()
            assert len(parts) == world and torch.equal(parts[rank], run.pos) and not any_nan
        ms = el / K_ * 1e3
        tot_ms += ms
        G_local += b["num_graphs"]
        E_b = edges_of(run)
        gl += gfrac
        if pr is not None and pr[1] > 0:
            pm, pn = pm + pr[0], pn + pr[1]
            pf += float(E_b) * FLOP_PER_EDGE_CFCONV * pr[1]
            pe += float(E_b) * pr[1]
            ts = tiles_of(run)
            if ts:
                for kk, vv in ts.items():
                    tile_acc[kk] = tile_acc.get(kk, 0.0) + vv * pr[1]
                tile_acc["launches"] = tile_acc.get("launches", 0.0) + pr[1]
        recs.append({"molecules": len(bm), "conformers": int(b["num_graphs"]), "atoms": run.topo.N,
                     "edges": E_b, "ms_per_step": ms})
        last = run
    return tot_ms, G_local, recs, gl / max(len(recs), 1), last, (pm, pn, pf, pe)


def drugs200_job(seed):
    """configs[2]: 200 molecules, G = 2 x U{50..500} conformers each (utils/datasets.py:720-721,763; scripts/test.py:135-141)."""
    from agdiff_amd import driver, synth
    rng = np.random.default_rng(seed)
    mols200 = []
    for i in range(200):
        at_, r_, c_, t_ = synth.random_molecule(rng, synth.sample_n_atoms(rng, "drugs"))
        mols200.append(dict(atom_type=at_, edge_index=np.stack([r_, c_]), edge_type=t_,
                            num_refs=int(rng.integers(50, 501)), name="m%d" % i, index=i))
    return mols200, driver.num_confs("2x")


def live_edges(run):
    """Directed edges of the run's current graph.  The fused sampler front keeps no full edge list (and no `num_edges`): the
    graph is the radius rows (`rad_cnt` per target) plus the local edges, which extend_graph_order_radius always keeps
    (common.py:222-233)."""
    if run._fused_front():
        return int(run.ws.rad_cnt.sum().item()) + run.topo.L
    return int(run.ws.num_edges.item())


def tile_stats(run):
    """16-row tiles one agdiff_cfconv_node launch walks on the run's current graph, live rows and executed rows.  Radius tiles:
    on quads (topo.group_targets == 4, tune_cfconv_quad_tiles >= 0: k_cfconv_quad) max over a quad's targets of ceil(rows / 4) --
    quarter k of a tile holds four rows of the quad's k-th target --, else every target's rows padded to whole tiles; local quad
    tiles as the topology built them."""
    if not run._fused_front():
        return None
    cnt = run.ws.rad_cnt.to("cpu").numpy().astype(np.int64)
    quad = run.topo.group_targets == 4 and int(run.pk.struct.tune_cfconv_quad_tiles) >= 0
    if quad:
        qt = run.topo.quad_tgt.to("cpu").numpy().astype(np.int64).reshape(-1, 4)
        rt = int(((np.where(qt >= 0, cnt[np.maximum(qt, 0)], 0) + 3) // 4).max(axis=1).sum())
    else:
        rt = int(((cnt + 15) // 16).sum())
    R, L, T = int(cnt.sum()), int(run.topo.L), int(run.topo.T)
    return {"radius_tiles": rt, "local_tiles": T, "radius_rows_live": R, "local_rows_live": L, "radius_rows_in_quad_tiles": int(quad)}


def load_pmc(precision, edges, kernels):
    """HBM traffic / SQ counters of the dominant kernel from the committed rocprofv3 --pmc summaries of this round
    (tools/pmc_bench.sh; separate counter passes, never inside a timed run): used when they were taken
    on this workload (edge count of a launch within 2 %), else null."""
    tf = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (PROFILE_ROUND, precision))
    if not os.path.exists(tf):
        return None
    tj = json.load(open(tf))
    if abs(tj.get("edges_per_launch", 0) - edges) > 0.02 * max(edges, 1) or not all(k in tj.get("kernels", {}) for k in kernels):
        return None
    return tj


def main():
    global JOB_STEPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10 per batch for drugs200, else 1000)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 2 per batch for drugs200, else 20)")
    ap.add_argument("--workload", default="drugs200", choices=["drugs", "drugs200", "qm9", "large", "alanine"])
    ap.add_argument("--mols", type=int, default=8)
    ap.add_argument("--copies", type=int, default=128)
    ap.add_argument("--max-atoms", type=int, default=196608, help="drugs200: atoms per packed batch (driver.plan_batches; measured on the default job in round 3: 50 k / 100 k / 200 k / 400 k / 800 k atoms -> 125 / 134 / 139 / 137 / 131 conformers/s).  196,608 = 3 x 256 CUs x 16 waves x 16 nodes: the node kernels (one 16-wave workgroup per CU and round) then run exactly three full rounds; 200,000 left a fourth round of 7 workgroups")
    ap.add_argument("--schedule", default="saturated", choices=["saturated", "default"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="default: strong with more than one rank -- ONE job, every packed batch (--max-atoms x ranks atoms) cut into "
                         "per-rank graph ranges, as agdiff_amd.driver shards it --, so that a scaling run measures the sharding and "
                         "the per-step all-gather on FIXED total work; weak: every rank samples its own copy of the job")
    ap.add_argument("--no-skip", action="store_true", help="run the global encoder even where its result is discarded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the runs reported under `extra`")
    ap.add_argument("--no-traj", action="store_true")
    ap.add_argument("--no-attr-far", action="store_true", help="local edges beyond the cutoff through the encoder MLP instead of their "
                                                               "far polynomials (A/B runs)")
    ap.add_argument("--no-full-job", action="store_true", help="skip extra.full_job (one complete 5000-step job, ~40 s)")
    ap.add_argument("--step-graphs", choices=["auto", "on", "off"], default="off",
                    help="the loop's steps as replayed HIP graphs (model.step_graphs; measured: no faster, profiles/r06_step_graphs.txt)")
    ap.add_argument("--no-gather-extra", action="store_true", help="skip extra.all_gather_world1")
    ap.add_argument("--end-to-end-batches", type=int, default=3,
                    help="extra.end_to_end: driver.run_job over this many consecutive batches x all steps (0: skip; then three full jobs instead of two)")
    ap.add_argument("--no-qm9-extra", action="store_true", help="skip extra.configs1_qm9")
    ap.add_argument("--no-profile", action="store_true", help="no event pairs around the CFConv launches of the timed region")
    ap.add_argument("--breakdown", default=None, help="write per-op timings (ms) to this JSON file")
    ap.add_argument("--seed", type=int, default=2021)
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16x3", "f16x3"])
    ap.add_argument("--radius-poly", default="auto", choices=["auto", "radius", "kt2", "kt3", "kt4", "off"],
                    help="filter polynomials (agdiff_amd/packing.py): off = every edge through the encoder + filter MLPs")
    ap.add_argument("--poly-passes", default="auto", choices=["auto", "full", "from64", "from96"],
                    help="MFMA passes over the filter polynomials' high terms: auto = one when the host's bound allows it "
                         "(agdiff_params_t.poly_plan), full = three for every term (A/B runs)")
    ap.add_argument("--group-targets", type=int, default=None, choices=[1, 2, 4],
                    help="targets per wave of agdiff_cfconv_node (BatchTopology.group_targets; default by batch size; A/B runs)")
    ap.add_argument("--tune", action="append", default=[], metavar="FIELD=VALUE",
                    help="kernel-variant threshold (agdiff_params_t.tune_*; PackedParams.TUNING), e.g. cfconv_four_min_quads=-1 (A/B runs)")
    ap.add_argument("--front", default="fused", choices=["fused", "split", "unfused"],
                    help="serial front of a step: one launch (update + local edges + radius graph), the same with the graph "
                         "phase launched after the local branch's fork, or the unfused kernels (A/B runs)")
    ap.add_argument("--serial", action="store_true", help="local and global branch on ONE stream (tune_serial_branches): per-kernel "
                    "stand-alone times under rocprofv3")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL all-gather path even with one rank")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the multi-rank path on a one-GPU box: every rank computes on cuda:0 and the ranks meet over "
                         "gloo (RCCL refuses two ranks on one device); the line says so and its value is not a scaling figure")
    ap.add_argument("--job-steps", type=int, default=JOB_STEPS, help="denoising steps of one sampling job (5000; the "
                    "alanine dipeptide example runs 100)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(*json.loads(args.cpu_worker))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # no launcher around us: become one.  Nothing above this line has touched a GPU.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], selftest=args.launcher_selftest or args.rehearse_on_one_gpu))    # (a rehearsal puts every rank on GPU 0)
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.launcher_selftest:
        launcher_selftest(world, rank)
        return

    # From here on this process measures.  Its stdout carries ONE line, the JSON record: whatever libraries print there (RCCL's
    # version banner comes through C stdio when a process group forms) goes to stderr instead -- file descriptor 1 is pointed at
    # stderr and the record is written to a duplicate of the original taken first.
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    JOB_STEPS = args.job_steps
    d200 = args.workload == "drugs200"
    K = args.steps if args.steps is not None else (10 if d200 else 1000)
    W = args.warmup if args.warmup is not None else (2 if d200 else 20)
    kind = "drugs" if d200 else args.workload

    # The CPU leg runs first, in separate interpreter processes, before this process initialises the GPU.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(kind, args.schedule, args.seed)

    import torch
    import torch.distributed as dist
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            if args.rehearse_on_one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            group_report = collective_selfcheck(dist, torch, dev, rank, world, gloo=args.rehearse_on_one_gpu)
        except Exception as e:                      # RCCL could not form the group / a collective failed: say so, measure nothing
            print("bench.py rank %d: the %d-rank %s group did not come up (%s: %s) -- MASTER_ADDR=%s MASTER_PORT=%s LOCAL_RANK=%d "
                  "HSA_ENABLE_IPC_MODE_LEGACY=%s, %d GPU(s) visible; nothing was measured"
                  % (rank, world, "gloo" if args.rehearse_on_one_gpu else "RCCL", type(e).__name__, e, os.environ.get("MASTER_ADDR"),
                     os.environ.get("MASTER_PORT"), local_rank, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), torch.cuda.device_count()),
                  file=sys.stderr, flush=True)
            sys.exit(3)
        if group_report["ranks"] != args.gpus:
            print("bench.py rank %d: --gpus %d but the group's all-reduce of ones gave %d" % (rank, args.gpus, group_report["ranks"]),
                  file=sys.stderr, flush=True)
            sys.exit(3)
    else:
        group_report = {"ranks": 1, "backend": None, "devices": [local_rank], "distinct_devices": 1}

    from agdiff_amd import _lib, driver, get_model, synth
    from agdiff_amd.dist import shard_of
    lib = _lib.load()

    # The measured path never touches oracle/: the synthetic checkpoint comes from the product-side closed-form
    # filler (agdiff_amd/synth.py; the oracle fills its own copy with the same function in the cpu_baseline leg).
    def make_model(schedule, radius_poly=None, weights="filler", poly_passes=None):
        cfg = make_cfg(kind, schedule)
        m = get_model(cfg)
        m.precision = args.precision
        m.radius_poly = radius_poly or args.radius_poly
        m.poly_passes = poly_passes or args.poly_passes
        m.attr_far_rows = not args.no_attr_far
        m.fused_front, m.front_split_graph = args.front != "unfused", args.front == "split"
        m.step_graphs = {"auto": "auto", "on": True, "off": False}[args.step_graphs]
        if args.serial:
            m.tuning["serial_branches"] = 1
        for kv in args.tune:
            m.tuning[kv.split("=")[0]] = int(kv.split("=")[1])
        if args.group_targets:
            m.group_targets = args.group_targets
        fill = synth.restoring_state_dict if weights == "restoring" else synth.synth_state_dict
        m.load_state_dict(fill(m.state_dict()))
        return m.to(dev).eval(), cfg
    model, cfg = make_model(args.schedule)

    save_traj, skip = not args.no_traj, not args.no_skip
    profile = not args.no_profile
    prof_ms = prof_n = 0.0
    prof_flop = prof_edges = 0.0
    per_batch = None
    strong = (args.scaling or ("strong" if world > 1 else "weak")) == "strong"

    gcalls = [0]          # all-gathers this rank issued in run_job (one per step per batch)
    tile_acc = {}         # tile statistics of the profiled CFConv launches (weighted by launches)

    def run_job(mdl, mcfg, batches, confs_of, schedule, sk, W_, K_, prof, every=1, dist_here=True):
        return run_job_steps(mdl, mcfg, batches, confs_of, schedule, sk, W_, K_, prof, every=every, dev=dev, rank=rank, world=world,
                             seed=args.seed, save_traj=save_traj, use_dist=use_dist and dist_here, strong=strong, gcalls=gcalls,
                             tile_acc=tile_acc)

    def reduce_job(tot_ms, G_local, G_all_if_strong):
        tt = torch.tensor([tot_ms], dtype=torch.float64, device=dev)
        gt = torch.tensor([G_local], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if not strong:
                dist.all_reduce(gt)
        return float(tt.item()), (G_all_if_strong if strong else int(gt.item()))

    if d200:
        # weak scaling (default): every rank samples its OWN copy of the 200-molecule job (per-GPU work fixed); strong: ONE
        # job, every packed batch cut into per-rank graph ranges.  Either way the ranks' positions are all-gathered over RCCL
        # after every denoising step (north_star), on a side stream
        # (weak: every rank takes the SAME 200 molecules -- the number of batches, hence of collectives, must agree over the
        # ranks -- with its own initial positions and noise: seed + rank in timed_run)
        mols200, confs_of = drugs200_job(args.seed)
        batches = driver.plan_batches(mols200, confs_of, driver.sharded_capacity(args.max_atoms, world) if strong else args.max_atoms)
        G_job = sum(confs_of(m["num_refs"]) for m in mols200)
        tot_ms, G_local, per_batch, global_frac, run, (prof_ms, prof_n, prof_flop, prof_edges) = run_job(
            model, cfg, batches, confs_of, args.schedule, skip, W, K, profile)
        # whole job = every rank works through its batches one after the other: time = max over ranks of its sum
        ms_per_step, G_total = reduce_job(tot_ms, G_local, G_job)      # ms for ONE step of EVERY batch of the slowest rank
        value = G_total / (ms_per_step * JOB_STEPS / 1e3)
        wl = ("BASELINE configs[2]: 200 Drugs-shaped synthetic molecules x 2*U{50..500} conformers = %d conformers "
              "(scripts/test.py:40-61,130-141), %d packed batches of <= %d atoms%s (this rank: %d batches, %d atoms, %d edges), "
              "%d warm-up + %d timed steps per batch, %s schedule, global branch active on %.0f%% of timed steps; "
              "ms_per_step = one step of every batch"
              % (G_job, len(batches), driver.sharded_capacity(args.max_atoms, world) if strong else args.max_atoms,
                 " cut into per-rank graph ranges" if strong else (" -- one such job per rank" if world > 1 else ""), len(per_batch), sum(r["atoms"] for r in per_batch),
                 sum(r["edges"] for r in per_batch), W, K, args.schedule, 100 * global_frac))
    else:
        copies = args.copies if kind != "large" else 1
        mols = args.mols if kind != "large" else args.mols * args.copies
        if strong:
            gb = build_batch(kind, mols, copies, args.seed)            # ONE global batch, cut by graph ranges
            b, (g0_, g1_), _ = shard_of(gb, rank, world)
            if b is None:
                raise SystemExit("strong scaling: rank %d got no graphs (%d graphs over %d ranks)" % (rank, gb["num_graphs"], world))
        else:
            b = build_batch(kind, mols, copies, args.seed + 1000 * rank)   # weak scaling: same shape per rank
        el, run, global_frac, gather, pr = timed_run(model, dev, b, cfg, W, K, args.schedule, skip, save_traj, args.seed,
                                                     rank, use_dist, profile=profile)
        G = b["num_graphs"]
        if pr is not None and pr[1] > 0:
            E_b = live_edges(run)
            prof_ms, prof_n, prof_flop, prof_edges = pr[0], pr[1], float(E_b) * FLOP_PER_EDGE_CFCONV * pr[1], float(E_b) * pr[1]
            ts = tile_stats(run)
            if ts:
                tile_acc.update({kk: vv * pr[1] for kk, vv in ts.items()})
                tile_acc["launches"] = pr[1]
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
            gt = torch.tensor([G], dtype=torch.int64, device=dev)
            dist.all_reduce(gt)
            G_total = int(gt.item())
        else:
            G_total = G
        run.check_nan()
        ms_per_step = el / K * 1e3
        value = G_total / (ms_per_step * JOB_STEPS / 1e3)
        if gather is not None:
            parts, any_nan = gather.result()
            assert len(parts) == world and parts[rank].shape[0] == run.topo.N and not any_nan
            assert torch.equal(parts[rank], run.pos), "all-gathered shard differs from the local positions"
        wl = ("%s-shaped synthetic molecules: %d molecules x %d conformers %s (this rank: %d atoms, %d edges, %d local "
              "edges), %s schedule, global branch active on %.0f%% of timed steps, %d-step job"
              % (kind, mols, copies, "in ONE global batch cut into per-rank graph ranges" if strong
                 else "per GPU", run.topo.N, live_edges(run), run.topo.L, args.schedule, 100 * global_frac,
                 JOB_STEPS))

    # ---- dominant operation = the two CFConvs of one InteractionBlock (encoder/schnet.py:136-162, 12 per forward).  With the
    # filter polynomials on it is ONE launch (k_cfconv_node: radius rows + local quad tiles per quad of targets), else one
    # k_cfconv_fused.  avg_launch_ms = HIP-event pairs around every such launch of the TIMED region (agdiff_profile_cfconv:
    # events on the launch stream, side-stream kernels running beside them as in the step); `achieved` prices the REFERENCE's
    # arithmetic (E x 90,112 FLOP per block) over that time.
    ws, topo, pk = run.ws, run.topo, run.pk
    stream = _lib.stream_ptr()
    E = live_edges(run)
    # (edges whose CFConv filters come from the filter MLPs: all of them without polynomials, else the local edges of types without a
    # coefficient set; E counts radius + local edges)
    if pk.poly_kt == 0:
        mlp_edges = E
    else:
        slotted = torch.tensor(sorted(pk.local_slots), dtype=torch.int32, device=topo.loc_type.device)
        mlp_edges = int((~torch.isin(topo.loc_type, slotted)).sum().item()) if topo.L else 0
    poly_info = {"mode": args.radius_poly, "poly_kt": pk.poly_kt, "terms": 32 * pk.poly_kt, "local_type_slots": int(pk.struct.poly_num_slots),
                 "pass_plan": int(pk.poly_plan), "one_pass_bound_of_high_terms": {k: float(v) for k, v in pk.poly_high_bound.items()},
                 "edges_through_filter_mlps_frac": mlp_edges / max(E, 1),
                 "cfconv_waves_per_simd": 4 if (int(ws.variant_log.item()) & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_FOUR"]) else (2 if pk.poly_kt >= 3 else 3),
                 "fit_errors_vs_float64_networks": {str(k): v for k, v in pk.poly_errors.items()}}
    roof = None
    P_, T_, W_ = ctypes.byref(pk.struct), ctypes.byref(topo.struct), ctypes.byref(ws.struct)
    node_path = pk.poly_kt > 0
    local_poly = bool(node_path and lib.agdiff_local_poly_enabled(P_, T_, W_))
    pads_info = None
    if tile_acc.get("launches"):
        nl = tile_acc["launches"]
        rt, lt_ = tile_acc["radius_tiles"] / nl, tile_acc["local_tiles"] / nl
        rl, ll = tile_acc["radius_rows_live"] / nl, tile_acc["local_rows_live"] / nl
        lt_used = lt_ if local_poly else 0.0
        pads_info = {"tiles_per_launch": rt + lt_used, "radius_tiles": rt, "local_tiles": lt_used,
                     "rows_live": rl + (ll if local_poly else 0.0), "rows_executed": 16.0 * (rt + lt_used),
                     "radius_pad_frac": 1.0 - rl / max(16.0 * rt, 1.0),
                     "local_pad_frac": (1.0 - ll / max(16.0 * lt_, 1.0)) if local_poly else None}
    if rank == 0 and prof_n > 0:
        avg_ms = prof_ms / prof_n
        t_s = prof_ms * 1e-3
        ref_priced = prof_flop / t_s / 1e12          # the REFERENCE's arithmetic for this op over the measured time
        pk_ = PEAK[args.precision]
        e_avg = prof_edges / prof_n
        quad_tiles = bool(int(ws.variant_log.item()) & _lib.DEFINES["AGDIFF_VAR_CFCONV_NODE_QUAD"])
        kname = "k_cfconv_quad" if quad_tiles else "k_cfconv_node"
        kern = [kname] if (node_path and local_poly) else ([kname, "k_cfconv_fused"] if node_path else ["k_cfconv_fused"])
        pmc = load_pmc(args.precision, e_avg, kern)
        passes = MFMA_PASSES[args.precision]
        n_avg = e_avg / max(E, 1) * topo.N
        # HBM bytes one launch has to move at least: per edge src + length + 2 scales (16 B), xs read once, one aggregate row
        # written per node (the node kernel; the MLP kernel also streams 512 B of edge_attr per edge)
        alg_bytes = e_avg * (16 if node_path else 528) + n_avg * 192 * 4 * 2
        if node_path:
            # What the kernel EXECUTES: per 16-row tile 12 channel tiles x poly_kt k-tiles x `passes` MFMAs of
            # 16x16x32 (16,384 FLOP each).  `achieved` = the MFMA FLOPs the ALGORITHM needs in this arithmetic mode (live rows
            # only: E x 192 channels x 32 poly_kt terms x 2 x passes) over the in-step launch time; the pad rows that complete a
            # target's / a type's last tile are executed too and reported beside it, as is the reference-priced figure of rounds 1-3.
            tiles = pads_info["tiles_per_launch"] if pads_info else None
            # MFMAs per 16-channel tile and 16 rows: poly_kt k-tiles x passes; with the pass plan (agdiff_params_t.poly_plan 1) two
            # at 32 terms (hi x hi of all terms + both cross terms of the low 16 in one instruction), 3 + 1 at 64 (mfma_per_channel_tile)
            mfma_ct = mfma_per_channel_tile(pk.poly_kt, pk.poly_plan, passes)
            issued = prof_edges * 192 * 32 * 2 * mfma_ct / t_s / 1e12
            executed = issued * (pads_info["rows_executed"] / max(pads_info["rows_live"], 1) if pads_info else 1.0)
            kernel = ("%s<NKT=%d> (one launch per InteractionBlock: radius rows %s%s)"
                      % (kname, pk.poly_kt, "in quad tiles" if quad_tiles else "by target",
                         " + local quad tiles, %d local types" % pk.struct.poly_num_slots if local_poly
                         else "; local edges through k_cfconv_fused on the padded local list, second launch"))
            # issue-slot model of one SIMD: an MFMA holds the issue port 8 cycles, any other VALU instruction 4 (wave64 on 16
            # lanes); VALU / tile from the SQ counter pass when one was taken on this workload
            cnts = pmc["kernels"][kern[0]].get("counters", {}) if pmc else {}
            issue_model = None
            if cnts.get("SQ_INSTS_VALU") and cnts.get("SQ_INSTS_MFMA") and tiles:
                n_mfma, n_other = cnts["SQ_INSTS_MFMA"], cnts["SQ_INSTS_VALU"] - cnts["SQ_INSTS_MFMA"]
                cyc = 8.0 * n_mfma + 4.0 * n_other
                issue_model = {"mfma_per_tile": n_mfma / tiles, "other_valu_per_tile": n_other / tiles,
                         "issue_cycles_per_tile": cyc / tiles, "simds": 1024, "clock_ghz_assumed": 2.4,
                         "issue_slot_frac": cyc / (1024 * 2.4e9 * avg_ms * 1e-3),
                         "source": "SQ_INSTS_VALU / SQ_INSTS_MFMA of profiles/%s_%s_pmc.json (separate counter passes of this command)" % (PROFILE_ROUND, args.precision)}
            note = ("frac = MFMA FLOPs the kernel's algorithm needs (a 32-term d-polynomial per filter channel and live row in "
                    "split 16-bit operands: mfma_per_channel_tile instructions of 16x16x32 per 16 rows x 16 channels -- three passes "
                    "per k-tile, or with the host-bounded pass plan two at 32 terms / four at 64, DESIGN.md 4c; fitted to the "
                    "reference's encoder + filter networks in float64 at load time, "
                    "accepted at <= 1e-6, DESIGN.md 4a) over the in-step launch time (HIP event pairs around every CFConv launch "
                    "of the timed region on its own stream; profiles/%s_*kernel_stats.csv holds rocprofv3's figure for the same "
                    "command) against the dense bf16 MFMA peak.  executed_mfma_frac adds the pad rows.  reference_priced_* is the "
                    "figure rounds 1-3 reported as frac: the REFERENCE's arithmetic for this op (it evaluates the 128->192->192 "
                    "filter network on every directed edge: E x 90,112 FLOP per block, SURVEY 8d) over the same time -- work this "
                    "kernel does not execute.  issue_model prices its instruction stream (SQ counters of profiles/%s_*_pmc.json)" % (PROFILE_ROUND, PROFILE_ROUND))
            ach = issued
        else:
            kernel = "k_cfconv_fused"
            issued = ref_priced * passes
            executed = issued
            issue_model = None
            ach = issued
            note = ("achieved = MFMA FLOPs issued: algorithmic FLOPs (E x 90,112: the reference evaluates the filter network on "
                    "every directed edge) x 3 bf16 passes (hi.hi + lo.hi + hi.lo; 1 in f32 mode), fp32 accumulate, over the "
                    "in-step launch time")
        roof = {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": pk_, "unit": "TFLOP/s", "frac": ach / pk_,
                "traffic": pmc["kernels"][kern[0]]["hbm_bytes_per_launch"] if pmc else None,
                "avg_launch_ms": avg_ms, "launches_timed": int(prof_n), "edges_per_launch": e_avg,
                "executed_mfma_tflops": executed, "executed_mfma_frac": executed / pk_,
                "useful_fp32_equivalent_tflops": (prof_edges * 192 * 32 * pk.poly_kt * 2 / t_s / 1e12) if node_path else issued / passes,
                "mfma_per_channel_tile": mfma_ct if node_path else None,
                "reference_priced_tflops": ref_priced, "reference_priced_frac": ref_priced / pk_,
                "hbm_frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": alg_bytes,
                "traffic_over_algorithmic": (pmc["kernels"][kern[0]]["hbm_bytes_per_launch"] / alg_bytes) if pmc else None,
                "hbm_frac_by_counters": (pmc["kernels"][kern[0]]["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if pmc else None,
                "valu_issue_busy": pmc["kernels"][kern[0]].get("valu_issue_busy") if pmc else None,
                "mfma_pipe_busy": pmc["kernels"][kern[0]].get("mfma_pipe_busy") if pmc else None,
                "issue_model": issue_model, "tiles": pads_info if node_path else None,
                "note": note}

    # ---- stand-alone CFConv aggregate (PyG propagate x_j * W, schnet.py:156-162) on the last batch's graph: the HBM-bound
    # "scatter" kernel BASELINE.json's north_star prices against the HBM roofline (unfused form: W[E,F] streamed)
    agg_roof = None
    if E > 0 and rank == 0:
        F = 128
        if run._fused_front():      # (the sampler kept radius rows only: the full destination-sorted list of the same positions)
            lib.agdiff_graph_build(T_, W_, run.pos_p, ctypes.c_float(cfg.cutoff), stream)
            E = int(ws.num_edges.item())
        Wt = torch.randn(E, F, device=dev)
        xin = torch.randn(topo.N, F, device=dev)
        outt = torch.empty(topo.N, F, device=dev)
        call = lambda: lib.agdiff_cfconv_aggregate(_lib.ptr(xin), _lib.ptr(Wt), _lib.ptr(ws.in_ptr), _lib.ptr(ws.e_src),
                                                   topo.N, F, _lib.ptr(outt), stream)
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = E * F * 4 + E * 4 + (topo.N + 1) * 4 + 2 * topo.N * F * 4      # fp32 516 B/edge + 1,028 B/node
        gbs = nbytes / (ms * 1e-3) / 1e9
        agg_roof = {"kernel": "k_cfconv_aggregate<128> (C-ABI op, not launched by the step: there the scatter-add is fused behind the filter)",
                    "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": ms, "algorithmic_bytes": nbytes}
        del Wt, xin, outt

    if args.breakdown and rank == 0:
        ops = {}

        def timeit(name, fn, reps=5):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ops[name] = e0.elapsed_time(e1) / reps
        P, Tp, Wp = P_, T_, W_
        et, lt = (topo.max_edges + _lib.TILE - 1) // _lib.TILE, (topo.L + _lib.TILE - 1) // _lib.TILE
        nc = cfg.num_convs
        timeit("graph_build", lambda: lib.agdiff_graph_build(Tp, Wp, run.pos_p, ctypes.c_float(cfg.cutoff), stream))
        timeit("edge_scales", lambda: lib.agdiff_edge_scales(P, Tp, Wp, 1, stream))
        timeit("edge_encoder", lambda: lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_canon), et, _lib.ptr(ws.c_len), _lib.ptr(ws.c_type), _lib.ptr(ws.e_attr), _lib.ptr(ws.l_attr_rows), _lib.ptr(ws.e_loc), _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), stream))
        timeit("node_stage_x%d" % (nc + 1), lambda: [lib.agdiff_schnet_node_stage(P, Tp, Wp, k, stream) for k in range(nc + 1)])
        timeit("cfconv_fused_x%d" % nc, lambda: [lib.agdiff_cfconv_fused(P, Tp, Wp, k, stream) for k in range(nc)])
        timeit("head_global", lambda: lib.agdiff_pair_head(ctypes.byref(pk.struct.head_global), _lib.ptr(ws.num_canon), et, _lib.ptr(ws.c_src), _lib.ptr(ws.c_dst), _lib.ptr(ws.h), _lib.ptr(ws.e_attr), None, _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), _lib.ptr(ws.e_inv_global), stream))
        ct = (topo.Lc + _lib.TILE - 1) // _lib.TILE
        timeit("local_lengths", lambda: lib.agdiff_local_lengths(Tp, Wp, run.pos_p, stream))
        timeit("local_encoder", lambda: lib.agdiff_edge_encoder(P, _lib.ptr(ws.num_local_canon), ct, _lib.ptr(ws.lc_len), _lib.ptr(topo.lc_type), None, _lib.ptr(ws.l_attr_rows), None, None, None, stream))
        timeit("gin_encoder_x%d" % cfg.num_convs_local, lambda: lib.agdiff_gin_encoder(P, Tp, Wp, 1, stream))
        timeit("local_head", lambda: lib.agdiff_pair_head(ctypes.byref(pk.struct.head_local), _lib.ptr(ws.num_local_canon), ct, _lib.ptr(topo.lc_src), _lib.ptr(topo.lc_dst), _lib.ptr(ws.hl), None, _lib.ptr(ws.l_attr_rows), _lib.ptr(topo.lc_pos), _lib.ptr(topo.lc_mir), _lib.ptr(ws.l_inv), stream))
        timeit("local_branch", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 0, stream))
        timeit("score_forward_global", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 1, stream))
        if node_path:       # the path the sampler runs (the entries above time the one-list MLP kernels on the same graph)
            lib.agdiff_graph_build_ex(Tp, Wp, run.pos_p, ctypes.c_float(cfg.cutoff), 1, stream)
            timeit("graph_build_sampler", lambda: lib.agdiff_graph_build_ex(Tp, Wp, run.pos_p, ctypes.c_float(cfg.cutoff), 1, stream))
            timeit("local_edge_rows", lambda: lib.agdiff_local_edge_rows(P, Tp, Wp, stream))
            timeit("scales_radius_rows", lambda: lib.agdiff_edge_scales_split(P, Tp, Wp, 0, stream))
            timeit("scales_local_tiles", lambda: lib.agdiff_edge_scales_split(P, Tp, Wp, 2, stream))
            timeit("cfconv_node_x%d" % nc, lambda: [lib.agdiff_cfconv_node(P, Tp, Wp, k, stream) for k in range(nc)])
            timeit("split_node_stage_x%d" % (nc + 1), lambda: [lib.agdiff_schnet_node_stage_split(P, Tp, Wp, k, 1, stream) for k in range(nc + 1)])
            timeit("head_poly", lambda: lib.agdiff_pair_head_poly(P, _lib.ptr(ws.num_canon), et, _lib.ptr(ws.c_src), _lib.ptr(ws.c_dst), _lib.ptr(ws.c_len), _lib.ptr(ws.h), _lib.ptr(ws.c_pos), _lib.ptr(ws.c_mir), _lib.ptr(ws.e_inv_global), stream))
            timeit("score_forward_global_sampler", lambda: lib.agdiff_score_forward(P, Tp, Wp, run.pos_p, 1 | 8, stream))
            ops.update(R=int(ws.rad_cnt.sum().item()), local_tiles=topo.T, poly_kt=pk.poly_kt, local_poly_slots=int(pk.struct.poly_num_slots))
        ops.update(N=topo.N, E=E, L=topo.L, G=int(run.topo.G), ms_per_step=ms_per_step)
        with open(args.breakdown, "w") as f:
            json.dump(ops, f, indent=1)

    # ---- `extra`: the reference's own schedule on the same job, with and without simplification (vii) (skipping
    # the global branch on steps whose result the sampler discards changes the work per step, not the outputs), and the
    # headline's fallback: the same saturated job with the filter polynomials off (every edge through the MLP kernels)
    def full_job(mdl, mcfg, batches_, confs_of_, schedule=None, extrapolate=True, which=None):
        """ONE complete sampling job, wall clock: one packed batch of the default job (`which`: its index; default the largest) x
        all 5000 denoising steps, trajectory kept on the device and copied to the host by finish() as the reference returns it
        (dualenc.py:545-547), the driver's default NaN / range polling (every 64 steps) -- next to the 20-step extrapolation;
        `ms_per_500_steps`: HIP events on the stream, no extra synchronisation (how the rate moves over a sustained run)."""
        bm = batches_[which] if which is not None else \
            max(batches_, key=lambda bb: sum(len(m_["atom_type"]) * confs_of_(m_["num_refs"]) for m_ in bb))
        b_ = driver.pack_batch(bm, confs_of_)
        Tt = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        at, bi, bt, ba = Tt(b_["atom_type"]), Tt(b_["bond_index"]), Tt(b_["bond_type"]), Tt(b_["batch"])
        g_ = torch.Generator(device="cpu").manual_seed(args.seed + 77)
        pos_init = torch.randn(at.shape[0], 3, generator=g_).to(dev)
        kw = dict(n_steps=JOB_STEPS, step_lr=1e-6, clip=1000.0, global_start_sigma=0.5, w_global=1.0, save_traj=save_traj,
                  skip_discarded_global=skip)
        schedule = schedule or args.schedule
        # short run of the same batch for the extrapolated figure (same code path as the headline)
        el_s = None
        if extrapolate:
            el_s, r_s, _, _, _ = timed_run(mdl, dev, b_, mcfg, W, K, schedule, skip, save_traj, args.seed + 77, rank, False)
            del r_s
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_ = mdl.begin_sampling(at, pos_init, bi, bt, ba, b_["num_graphs"], False, **kw)
        marks = [torch.cuda.Event(enable_timing=True)]
        marks[0].record()
        while run_.remaining() > 0:
            run_.advance(min(500, run_.remaining()))
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        torch.cuda.synchronize()
        t_loop = time.perf_counter() - t0
        pos_f, traj_f = run_.finish()          # trajectory D2H: n_steps x N x 12 B
        t_all = time.perf_counter() - t0
        finite = bool(torch.isfinite(pos_f).all().item())
        rec = {"schedule": schedule, "batch_atoms": int(at.shape[0]), "conformers": int(b_["num_graphs"]), "steps": JOB_STEPS,
               "global_branch_steps": int(run_.global_steps),
               "full_job_s": t_all, "denoising_loop_s": t_loop, "finish_trajectory_d2h_s": t_all - t_loop,
               "trajectory_bytes": int(len(traj_f)) * int(at.shape[0]) * 12,
               "conformers_per_s_full_job": b_["num_graphs"] / t_all, "nan_check_every": 64,
               "ms_per_500_steps": [round(marks[i].elapsed_time(marks[i + 1]), 1) for i in range(len(marks) - 1)],
               "positions_finite": finite, "max_abs_pos": float(pos_f.abs().max().item()),
               "radius_in_degree_at_the_end": float(run_.ws.rad_cnt.float().mean().item())}
        if el_s is not None:
            rec.update(extrapolated_s=el_s / K * JOB_STEPS, ratio_full_over_extrapolated=t_all / (el_s / K * JOB_STEPS))
        del run_, traj_f
        return rec

    def qm9_extra():
        """BASELINE configs[1] in the driver-run line: 200 QM9-shaped molecules x 2 * U{50..500} conformers, saturated
        schedule, the driver's batch plan, 2 + 10 steps per batch on every 4th batch."""
        rng = np.random.default_rng(args.seed + 9)
        molsq = []
        for i in range(200):
            at_, r_, c_, t_ = synth.random_molecule(rng, synth.sample_n_atoms(rng, "qm9"))
            molsq.append(dict(atom_type=at_, edge_index=np.stack([r_, c_]), edge_type=t_, num_refs=int(rng.integers(50, 501)),
                              name="q%d" % i, index=i))
        cq = driver.num_confs("2x")
        bq = driver.plan_batches(molsq, cq, args.max_atoms)
        cfgq = make_cfg("qm9", "saturated")
        mq = get_model(cfgq)
        mq.precision, mq.radius_poly, mq.poly_passes = args.precision, args.radius_poly, args.poly_passes
        mq.load_state_dict(synth.synth_state_dict(mq.state_dict()))
        mq = mq.to(dev).eval()
        tms, Gl, nb, pms, pn, pe = 0.0, 0, 0, 0.0, 0.0, 0.0
        for bidx, bm in enumerate(bq):
            if bidx % 4:
                continue
            b_ = driver.pack_batch(bm, cq)
            el_, r_, _, _, pr = timed_run(mq, dev, b_, cfgq, 2, 10, "saturated", True, save_traj, args.seed + bidx, rank, False, profile=True)
            r_.check_nan()
            tms += el_ / 10 * 1e3
            Gl += b_["num_graphs"]
            nb += 1
            if pr and pr[1] > 0:
                pms, pn, pe = pms + pr[0], pn + pr[1], pe + float(live_edges(r_)) * pr[1]
            del r_
        G_all = sum(cq(m_["num_refs"]) for m_ in molsq)
        passes = MFMA_PASSES[args.precision]
        return {"workload": "BASELINE configs[1]: 200 QM9-shaped synthetic molecules x 2*U{50..500} conformers = %d conformers, "
                            "%d packed batches of <= %d atoms, every 4th timed (2 warm-up + 10 steps), saturated schedule" % (G_all, len(bq), args.max_atoms),
                "value": Gl / (tms * JOB_STEPS / 1e3), "unit": "conformers/s", "ms_per_step_timed_batches": tms, "batches_timed": nb,
                "cfconv_avg_launch_ms": (pms / pn) if pn else None,
                "cfconv_mfma_frac": (pe * 192 * 32 * 2 * mfma_per_channel_tile(mq.packed().poly_kt, mq.packed().poly_plan, passes)
                                     / (pms * 1e-3) / 1e12 / PEAK[args.precision]) if pn else None}

    extra, value_full_job = None, None
    if rank == 0 and world == 1 and not args.no_extra and args.schedule == "saturated" and kind != "alanine":
        del run
        Ke, We = min(K, 200), min(W, 10)
        every = 4 if d200 else 1
        extra = {"note": "same job%s; reference schedule = beta_end 2e-3 (sigma < 0.5 on 2012 of 5000 steps), %d timed steps per "
                         "batch; fallback = saturated schedule with --radius-poly off" % (" on every 4th packed batch" if d200 else "", Ke)}

        def side_run(mdl, mcfg, schedule, sk):
            if d200:
                tms, Gl, recs, gf, r2, _ = run_job(mdl, mcfg, batches, confs_of, schedule, sk, We, Ke, False, every=every)
                del r2
                return Gl / (tms * JOB_STEPS / 1e3), tms, gf, len(recs)
            el2, r2, gf, _, _ = timed_run(mdl, dev, b, mcfg, We, Ke, schedule, sk, save_traj, args.seed, rank, False)
            r2.check_nan()
            ms2 = el2 / Ke * 1e3
            del r2
            return b["num_graphs"] / (ms2 * JOB_STEPS / 1e3), ms2, gf, 1
        # the reference's schedule on the synthetic checkpoint WITH a restoring force (agdiff_amd/synth.py: every local edge a
        # spring; reference-pinned by tests/golden/g14_*): molecules stay compact at every sigma, so the steps below sigma = 0.5
        # see the dense radius graph a trained model would see
        m2, cfg2 = make_model("default", weights="restoring")
        for name, sk in (("default_schedule_skip_discarded_global", True), ("default_schedule_no_skip", False)):
            v, ms2, gf2, nb = side_run(m2, cfg2, "default", sk)
            extra[name] = {"value": v, "unit": "conformers/s", "ms_per_step": ms2, "steps": Ke, "batches": nb,
                           "global_branch_share_of_steps": gf2, "checkpoint": "synthetic filler + restoring force",
                           "note": "timed steps drawn from the two ranges of the schedule (sigma < 0.5 / >= 0.5) in the job's own proportion 2012 : 2988; "
                                   "each batch starts from N(0, sigma_T^2) positions, so the few timed steps at low sigma still see a spread-out "
                                   "graph: the full-schedule figure for compact molecules is extra.default_schedule_full_job"}
        del m2
        if args.radius_poly != "off":
            m3, cfg3 = make_model(args.schedule, radius_poly="off")
            v, ms3, gf3, nb = side_run(m3, cfg3, args.schedule, skip)
            extra["fallback_filter_polynomials_off"] = {"value": v, "unit": "conformers/s", "ms_per_step": ms3, "steps": Ke,
                                                        "batches": nb}
            del m3
        if args.radius_poly == "auto":
            # the same job with 64-term polynomials (what a checkpoint whose first encoder layer is 2-8 x sharper gets)
            m4, cfg4 = make_model(args.schedule, radius_poly="kt2")
            v, ms4, gf4, nb = side_run(m4, cfg4, args.schedule, skip)
            extra["filter_polynomials_64_terms"] = {"value": v, "unit": "conformers/s", "ms_per_step": ms4, "steps": Ke,
                                                    "batches": nb, "fraction_of_headline": v / value}
            del m4
            # ... and the two rungs below it, with the pass plans the sharp checkpoints of tools/sharpness_sweep.py take there: 96 terms
            # (first layer ~32 x sharper) one pass from term 64 on, 128 terms (~64 x) one pass from term 96 on
            for terms, rp, pp in ((96, "kt3", "from64"), (128, "kt4", "from96")):
                m5, cfg5 = make_model(args.schedule, radius_poly=rp, poly_passes=pp if args.poly_passes == "auto" else None)
                v, ms5, gf5, nb = side_run(m5, cfg5, args.schedule, skip)
                extra["filter_polynomials_%d_terms" % terms] = {"value": v, "unit": "conformers/s", "ms_per_step": ms5, "steps": Ke, "batches": nb,
                                                                "pass_plan": int(m5.packed().poly_plan), "fraction_of_headline": v / value}
                del m5
        if d200 and not use_dist and not args.no_gather_extra:
            # cost of north_star's per-step collective on this box: the same batches with a one-rank RCCL group and the
            # all-gather of [positions | NaN flag] issued after every step (what every rank of a multi-GPU run does)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(_free_port())
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            try:
                tg = tn = 0.0
                calls0 = gcalls[0]
                for bidx, bm in enumerate(batches):
                    if bidx % every:
                        continue
                    b_ = driver.pack_batch(bm, confs_of)
                    el_, r_, _, ga, _ = timed_run(model, dev, b_, cfg, We, Ke, args.schedule, skip, save_traj, args.seed + bidx, rank, True)
                    r_.check_nan()
                    assert ga.calls == We + Ke
                    parts, any_nan = ga.result()
                    assert torch.equal(parts[0], r_.pos) and not any_nan
                    tg += el_ / Ke * 1e3
                    tn += per_batch[bidx]["ms_per_step"]
                    del r_, ga
                extra["all_gather_world1"] = {"ms_per_step_with_gather": tg, "ms_per_step_without": tn, "batches": len(batches[::every]),
                                              "note": "one all_gather_into_tensor of [pos | nan flag] per denoising step on a side stream (agdiff_amd/dist.py: StepAllGather), one-rank RCCL group"}
            finally:
                dist.destroy_process_group()
        if d200 and not args.no_full_job:
            # complete 5000-step jobs of three packed batches of different composition (fewest / median / most edges per step):
            # what a job delivers against what the 20 timed steps per batch extrapolate to
            order = sorted(range(len(per_batch)), key=lambda i_: per_batch[i_]["edges"])
            picks = sorted(set([order[0], order[-1]]) if args.end_to_end_batches > 0 else set([order[0], order[len(order) // 2], order[-1]]))
            jobs = [full_job(model, cfg, batches, confs_of, which=i_) for i_ in picks]
            ratio = sum(j["full_job_s"] for j in jobs) / sum(j["extrapolated_s"] for j in jobs)
            extra["full_job"] = max(jobs, key=lambda j: j["batch_atoms"])
            extra["full_jobs"] = {"batches": picks, "jobs": jobs, "ratio_full_over_extrapolated": ratio,
                                  "note": "wall clock of begin_sampling .. finish() (trajectory D2H, NaN / range polls every 64 steps included) "
                                          "over the extrapolation from that batch's timed steps; value_full_job = value / ratio"}
            value_full_job = value / ratio
            # ... and the job END TO END: agdiff_amd.driver.run_job -- the counterpart of scripts/test.py:116-176 --
            # over the first consecutive batches of the job's plan, every batch all 5000 steps: planning, packing, the static topology
            # of every batch (the first on the main thread, the next ones in the background while the GPU samples), host <-> device
            # copies, NaN / range polls, the .npz of every batch and the merged file.  Wall clock against what the same batches' timed
            # steps extrapolate to and against the full-job ratio above.
            if args.end_to_end_batches > 0 and len(batches) >= args.end_to_end_batches:
                import tempfile
                nb = args.end_to_end_batches
                use = [m_ for bm in batches[:nb] for m_ in bm]
                kw_job = dict(n_steps=JOB_STEPS, step_lr=1e-6, clip=1000.0, global_start_sigma=0.5, w_global=1.0, skip_discarded_global=skip)
                with tempfile.TemporaryDirectory() as d_:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    res_ = driver.run_job(model, use, d_, confs_of, args.max_atoms, kw_job, dev, log=lambda *_: None)
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t0
                confs_ = sum(confs_of(m_["num_refs"]) for m_ in use)
                got_ = sum(v.shape[0] for k_, v in res_.items() if k_.startswith("pos_gen_"))
                planned = driver.plan_batches(use, confs_of, args.max_atoms)
                same_plan = [sorted(m_["index"] for m_ in bm) for bm in planned] == [sorted(m_["index"] for m_ in bm) for bm in batches[:nb]]
                extrap = sum(per_batch[i_]["ms_per_step"] for i_ in range(nb)) * JOB_STEPS / 1e3
                extra["end_to_end"] = {
                    "what": "driver.run_job over the first %d consecutive batches of the plan x %d steps each: plan, pack, topology (background thread), "
                            "sample, polls, per-batch .npz + merged file; trajectories not saved (scripts/test.py default)" % (nb, JOB_STEPS),
                    "batches": nb, "same_batches_as_the_plan": bool(same_plan), "conformers": int(confs_), "conformers_written": int(got_),
                    "wall_s": wall, "conformers_per_s": confs_ / wall, "extrapolated_from_timed_steps_s": extrap,
                    "wall_over_extrapolated": wall / extrap, "wall_over_full_job_rate": wall / (extrap * ratio),
                    "note": "wall_over_full_job_rate = end-to-end wall clock over (these batches' extrapolation x the full-job ratio): what planning, "
                            "packing, topology, saving cost on top of sampling"}
            # ... and the REFERENCE's schedule end to end on the restoring-force checkpoint: 2012 steps with the global branch on
            # a dense radius graph, 2988 local-only steps -- what a trained model's 5000-step job costs
            m5, cfg5 = make_model("default", weights="restoring")
            extra["default_schedule_full_job"] = full_job(m5, cfg5, batches, confs_of, schedule="default", extrapolate=False)
            extra["default_schedule_full_job"]["checkpoint"] = "synthetic filler + restoring force (agdiff_amd/synth.py; tests/golden/g14_*)"
            del m5
        if d200 and not args.no_qm9_extra:
            extra["configs1_qm9"] = qm9_extra()
        if cpu is not None:
            extra["x_vs_cpu_whole_host"] = value / cpu["value"]
            extra["x_vs_cpu_single_process"] = value / cpu["single_process"]["value"]

    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        out = {
            "metric": "conformers/sec (whole node), GEOM-Drugs 5000-step sampling",
            "value": value, "value_full_job": value_full_job, "unit": "conformers/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,      # the reference publishes no number for this metric
            "dtype": args.precision, "data": "synthetic", "rccl_ranks": group_report["ranks"], "process_group": group_report,
            "config": {"workload": wl, "conformers_total": G_total, "parallelism": "dp%d" % world,
                       "all_gather_per_step": bool(use_dist), "all_gather_calls_rank0": int(gcalls[0]) if d200 else (gather.calls if use_dist else 0),
                       "trajectory_saved": save_traj, "nan_check_every": 64,
                       "headline": "value = conformers / (ms of the timed steps of every batch x 5000): the per-step rate extrapolated to "
                                   "the 5000-step job; value_full_job = value / (wall clock of three complete 5000-step jobs over their "
                                   "extrapolations: extra.full_jobs), i.e. with the trajectory copy, the NaN / range polls and the clock a "
                                   "sustained run holds (null when the full jobs were skipped)",
                       "skip_discarded_global": skip, "filter_polynomials": poly_info},
            "roofline": roof, "roofline_cfconv_aggregate": agg_roof, "cpu_baseline": cpu, "extra": extra,
        }
        if per_batch is not None:
            out["config"]["batches"] = per_batch
        if args.rehearse_on_one_gpu:
            out["config"]["rehearsal"] = "all %d ranks on ONE GPU, collectives over gloo: a run of the multi-rank code path, not a scaling figure" % world
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)         # (C stdio of the libraries: block-buffered on a pipe)
        os.write(record_fd, (json.dumps(out) + "\n").encode())       # the one line of this process's stdout


if __name__ == "__main__":
    main()
