"""Sampling driver: the counterpart of the reference's scripts/test.py:116-195 for this build.

scripts/test.py needs PyG `Data` pickles, rdkit and easydict and samples ONE molecule per call
(`repeat_data(data, 2 * num_refs)`, 100-1000 conformers, test.py:135-141), which leaves an MI355X
mostly idle.  This driver keeps its contract -- per molecule `num_confs(num_refs)` conformers,
5000-step Langevin sampling with the same arguments, at most one retry with `clip_local=20` when a NaN
appears in a molecule (test.py:143-181; only that molecule is re-sampled), results saved after every batch, `--resume` skips finished molecules --
but reads/writes PyG-free `.npz` files and packs many molecules into each batch.

Input .npz (see `save_testset`): for molecule i: `atom_type_i` [n], `edge_index_i` [2, e] and `edge_type_i` [e]
(bond graph; already extended to order 3 like `AddHigherOrderEdges` does unless --extend-order),
`num_refs_i` scalar, `name_i` string.  Output: `samples_<first>_<last>.npz` per batch (named by the molecule
indices it holds) with `pos_gen_<i>` [num_samples, n, 3] (+ `traj_<i>` [steps, num_samples, n, 3] with
--save-traj) and the merged `samples_all.npz`, written by rank 0 after a barrier.

    python -m agdiff_amd.driver --ckpt ckpt.pt --testset test.npz --out out_dir [--n-steps 5000]
    torchrun --nproc-per-node 8 -m agdiff_amd.driver ...      (each packed batch is sharded over the ranks by graph
                                                               ranges; one RCCL all-gather of positions per step)
"""
import argparse
import glob
import os

import numpy as np


def num_confs(spec):
    """scripts/test.py:15-24."""
    if str(spec).endswith("x"):
        return lambda x: x * int(str(spec)[:-1])
    if int(spec) > 0:
        return lambda x: int(spec)
    raise ValueError(spec)


def save_testset(path, molecules):
    """molecules: list of dicts with atom_type, edge_index, edge_type, num_refs, name."""
    out = {"count": np.int64(len(molecules))}
    for i, m in enumerate(molecules):
        out["atom_type_%d" % i] = np.asarray(m["atom_type"], dtype=np.int64)
        out["edge_index_%d" % i] = np.asarray(m["edge_index"], dtype=np.int64)
        out["edge_type_%d" % i] = np.asarray(m["edge_type"], dtype=np.int64)
        out["num_refs_%d" % i] = np.int64(m.get("num_refs", 1))
        out["name_%d" % i] = np.str_(m.get("name", "mol%d" % i))
    np.savez_compressed(path, **out)


def load_testset(path):
    z = np.load(path, allow_pickle=False)
    mols = []
    for i in range(int(z["count"])):
        mols.append(dict(atom_type=z["atom_type_%d" % i], edge_index=z["edge_index_%d" % i],
                         edge_type=z["edge_type_%d" % i], num_refs=int(z["num_refs_%d" % i]),
                         name=str(z["name_%d" % i]), index=i))
    return mols


def sharded_capacity(max_atoms, world):
    """Atoms of one global batch that `world` ranks share by contiguous graph ranges (dist.shard_graphs balances EDGES, so the
    ranks' atom counts differ by a few per cent: profiles/r05_shard_balance.json): 3 % below max_atoms x world, so that no rank's
    share crosses max_atoms -- 196,608 by default = three full rounds of the node kernels' workgroups; a rank 2 % above it ran a
    fourth round and 4 % longer than its peers."""
    return max_atoms * world if world <= 1 else int(max_atoms * world * 0.97)


def plan_batches(mols, confs_of, max_atoms):
    """First-fit decreasing: molecules sorted by their atom count x conformers (largest first), each put into the first
    batch it still fits into (a molecule larger than `max_atoms` gets a batch of its own).  Packing in input order
    leaves many half-empty batches when a few molecules have hundreds of conformers (GEOM test molecules carry 50-500
    references, utils/datasets.py:720-721), and a half-empty batch costs nearly a full one per denoising step.
    Results are keyed by molecule index, so the order inside the batches does not matter; every batch keeps its
    molecules in ascending index order."""
    need = [int(m["atom_type"].shape[0]) * confs_of(m["num_refs"]) for m in mols]
    order = sorted(range(len(mols)), key=lambda k: (-need[k], k))
    bins, room = [], []
    for k in order:
        for b in range(len(bins)):
            if need[k] <= room[b]:
                bins[b].append(k)
                room[b] -= need[k]
                break
        else:
            bins.append([k])
            room.append(max_atoms - need[k])
    batches = [[mols[k] for k in sorted(b)] for b in bins]
    batches.sort(key=lambda bm: bm[0].get("index", 0) if isinstance(bm[0], dict) else 0)
    return batches


def pack_batch(mols, confs_of):
    """repeat_data (utils/misc.py:88-90) for every molecule of the batch, concatenated."""
    from .synth import repeat_molecule
    ats, rs, cs, ts, bs, spans = [], [], [], [], [], []
    node_off, g_off = 0, 0
    for m in mols:
        g = confs_of(m["num_refs"])
        n = int(m["atom_type"].shape[0])
        a, r, c, t, b = repeat_molecule(m["atom_type"], m["edge_index"][0], m["edge_index"][1], m["edge_type"], g,
                                        node_off, g_off)
        ats.append(a); rs.append(r); cs.append(c); ts.append(t); bs.append(b)
        spans.append((node_off, n, g))
        node_off += n * g
        g_off += g
    return dict(atom_type=np.concatenate(ats), bond_index=np.stack([np.concatenate(rs), np.concatenate(cs)]),
                bond_type=np.concatenate(ts), batch=np.concatenate(bs), num_graphs=g_off, spans=spans)


def prepare_batch(model, bmols, confs_of, rank=0, world=1):
    """(packed batch, its BatchTopology on the host or None) -- for world > 1 the topology of THIS rank's graph range of the
    batch (dist.shard_of), as sample_batch_sharded samples it.  Host work only (no GPU call): run_job calls it for the next
    batch from a background thread.  None for models without prepare_topology (test stubs) or a rank without graphs."""
    packed = pack_batch(bmols, confs_of)
    if not hasattr(model, "prepare_topology"):
        return packed, None
    part = packed
    if world > 1:
        from .dist import shard_of
        part = shard_of(packed, rank, world)[0]
        if part is None:
            return packed, None
    try:
        return packed, model.prepare_topology(part["atom_type"], part["bond_index"], part["bond_type"], part["batch"], part["num_graphs"],
                                              extend_order=False, device="cpu")
    except Exception:
        # (a batch the topology refuses -- e.g. a molecule beyond the atom limit: the sampler builds it again and raises THERE, where
        # sample_batch_sharded keeps a failing rank's collectives in step with the others)
        return packed, None


def _topology_options(model):
    """What model.prepare_topology needs of the model (plain values: they travel to the worker process)."""
    return dict(order=int(model.config.edge_order), group_targets=getattr(model, "group_targets", None),
                radius_column=bool(getattr(model, "tuning", {}).get("group_radius_column", 1)))


def _prepare_in_worker(bmols, confs, rank, world, topo_opts):
    """prepare_batch in a worker PROCESS (run_job): the same packed batch and BatchTopology, built without the sampling
    process's interpreter lock -- a background THREAD hid 0.3 s of a batch's 1-4 s of numpy, because every Python-level step of
    the build waits for the launch loop's lock and vice versa.  Host work only: nothing here touches a GPU.  `confs`: conformers
    per molecule of the batch (the callable of scripts/test.py:15-24 does not pickle)."""
    from .topology import BatchTopology
    it = iter(confs)
    packed = pack_batch(bmols, lambda _num_refs: next(it))
    part = packed
    if world > 1:
        from .dist import shard_of
        part = shard_of(packed, rank, world)[0]
        if part is None:
            return packed, None
    try:
        return packed, BatchTopology(part["atom_type"], part["bond_index"], part["bond_type"], part["batch"], num_graphs=part["num_graphs"],
                                     extend_order=False, device="cpu", **topo_opts)
    except Exception:
        return packed, None               # (the sampler builds it again and raises there: prepare_batch)


def sample_batch(model, packed, device, sampler_kwargs, save_traj=False, max_retry=2, log=print, pos_init=None,
                 noise=None, topology=None):
    """test.py:143-181 for every molecule of a packed batch: a molecule in which a NaN appeared is sampled again
    (fresh pos_init) with clip_local=20, at most `max_retry` attempts in all, and dropped after that; the molecules
    packed with it keep their first result -- graphs are independent on the whole path, and the update kernel flags
    NaNs per graph (agdiff_ws_t.nan_flag).  Returns (pos [N,3] cpu, traj [steps,N,3] cpu or None, ok [num molecules]
    bool); rows of failed molecules are NaN.  `pos_init` [N,3] / `noise` [steps,N,3] replace the first attempt's
    random draws (parity tests).  `topology`: the batch's BatchTopology prepared ahead (prepare_batch; first attempt only).
    Models without begin_sampling (test stubs) take the reference's whole-batch retry."""
    import torch
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    spans = packed["spans"]
    n_mol = len(spans)
    if not hasattr(model, "begin_sampling"):
        at, bi, bt, ba = T(packed["atom_type"]), T(packed["bond_index"]), T(packed["bond_type"]), T(packed["batch"])
        clip_local = None
        for _ in range(max_retry):
            try:
                p0 = torch.randn(at.shape[0], 3).to(device) if pos_init is None else pos_init.to(device)
                pos_gen, traj = model.langevin_dynamics_sample_diffusion(
                    atom_type=at, pos_init=p0, bond_index=bi, bond_type=bt, batch=ba,
                    num_graphs=packed["num_graphs"], extend_order=False, clip_local=clip_local,
                    save_traj=save_traj, **sampler_kwargs)
                return pos_gen.cpu(), (torch.stack(traj) if save_traj else None), np.ones(n_mol, dtype=bool)
            except FloatingPointError:
                clip_local = 20
                log("Retrying with local clipping.")
        return None, None, np.zeros(n_mol, dtype=bool)

    N = packed["atom_type"].shape[0]
    pos_out = torch.full((N, 3), float("nan"))
    traj_out = None
    ok = np.zeros(n_mol, dtype=bool)
    # Passes still to run: (molecule slots of `packed`, clip_local, split-bf16?, attempts counted so far).  A molecule in which a
    # NaN appeared is sampled again with clip_local=20 and that pass counts against max_retry (test.py:143-181).  A molecule
    # whose ONLY fault was leaving the split-fp16 range (epsnet.check_nan) has not failed by the reference's lights: in fp32 it
    # would have been sampled once, without local clipping -- it is sampled again in split-bf16 (fp32's exponent range) with
    # the sampler settings unchanged, and that pass is not counted.
    passes = [(list(range(n_mol)), None, False, 0)]
    first = True
    while passes:
        todo, clip_local, wide, tries = passes.pop(0)
        sub = packed if len(todo) == n_mol else subset_batch(packed, todo)
        at, bi, bt, ba = T(sub["atom_type"]), T(sub["bond_index"]), T(sub["bond_type"]), T(sub["batch"])
        p0 = pos_init.to(device) if (first and pos_init is not None) else torch.randn(at.shape[0], 3).to(device)
        with _arithmetic(model, wide):
            extra = {"topology": topology} if (topology is not None and sub is packed) else {}
            run = model.begin_sampling(at, p0, bi, bt, ba, sub["num_graphs"], False, clip_local=clip_local,
                                       save_traj=save_traj, raise_on_nan=False,
                                       noise=(noise if first else None), **extra, **sampler_kwargs)
            topology = None                    # (moved to the device and owned by the run now)
            run.advance(run.remaining())
            pos, traj = run.finish()
        first = False
        pos = pos.cpu()
        bad_graph = run.nan_graphs().numpy()
        out_of_range = sorted(getattr(run, "range_graphs", ()))
        if save_traj:
            traj = torch.stack(traj)
            if traj_out is None:
                traj_out = torch.full((traj.shape[0], N, 3), float("nan"))
        nan_failed, range_failed = _sort_results(todo, sub["spans"], spans, bad_graph, out_of_range, wide, ok, pos_out, pos,
                                                 traj_out if save_traj else None, traj if save_traj else None)
        if out_of_range:
            SAMPLE_STATS["range_trips"] += len(out_of_range)
        if range_failed:
            SAMPLE_STATS["bf16x3_retries"] += 1
            log("%d conformers left the split-fp16 range: sampling their molecules (%d of %d) again in split-bf16."
                % (len(out_of_range), len(range_failed), len(sub["spans"])))
            passes.append((range_failed, clip_local, True, tries))
        if nan_failed:
            if tries + 1 < max_retry:
                log("NaN in %d of %d molecules: retrying those with local clipping." % (len(nan_failed), len(sub["spans"])))
                passes.append((nan_failed, 20, wide, tries + 1))
            else:
                SAMPLE_STATS["dropped"] += len(nan_failed)
    return pos_out, traj_out, ok


# what sample_batch / sample_batch_sharded met since the process started (run_job logs it with the job's summary)
SAMPLE_STATS = {"range_trips": 0, "bf16x3_retries": 0, "dropped": 0}


def _sort_results(todo, sub_spans, spans, bad_graph, out_of_range, wide, ok, pos_out, pos, traj_out, traj):
    """One pass's molecules: the good ones' rows go into pos_out / traj_out (rows of `packed`), the others are returned as
    (molecules with a NaN graph, molecules whose bad graphs all merely left the split-fp16 range).  In split-bf16 (`wide`)
    the range watch does not run, so nothing is range-only there."""
    in_range_set = set(int(g) for g in out_of_range)
    nan_failed, range_failed, g_off = [], [], 0
    for slot, (off_s, n, g) in zip(todo, sub_spans):
        off, _, _ = spans[slot]
        bad = [k for k in range(g_off, g_off + g) if bad_graph[k]]
        if bad:
            (range_failed if (not wide and all(k in in_range_set for k in bad)) else nan_failed).append(slot)
        else:
            ok[slot] = True
            pos_out[off:off + n * g] = pos[off_s:off_s + n * g]
            if traj_out is not None:
                traj_out[:, off:off + n * g] = traj[:, off_s:off_s + n * g]
        g_off += g
    return nan_failed, range_failed


def _arithmetic(model, wide):
    """`wide`: the model in split-bf16 for both branches (fp32's exponent range) for the duration of one attempt."""
    import contextlib
    if wide and hasattr(model, "arithmetic"):
        return model.arithmetic("bf16x3", "bf16x3")
    return contextlib.nullcontext(model)


def subset_batch(packed, slots):
    """The packed batch restricted to the molecule slots `slots` (re-based node / graph ids)."""
    keep_nodes, spans, batch = [], [], []
    node_off = g_off = 0
    shift = np.zeros(packed["atom_type"].shape[0], dtype=np.int64)
    gfirst = np.concatenate([[0], np.cumsum([g for (_, _, g) in packed["spans"]])])
    for s in slots:
        off, n, g = packed["spans"][s]
        idx = np.arange(off, off + n * g)
        keep_nodes.append(idx)
        shift[idx] = node_off - off
        batch.append(packed["batch"][idx] - gfirst[s] + g_off)
        spans.append((node_off, n, g))
        node_off += n * g
        g_off += g
    keep = np.concatenate(keep_nodes)
    mask = np.zeros(packed["atom_type"].shape[0], dtype=bool)
    mask[keep] = True
    bi = packed["bond_index"]
    esel = mask[bi[0]]
    return dict(atom_type=packed["atom_type"][keep], bond_index=bi[:, esel] + shift[bi[0][esel]][None, :],
                bond_type=packed["bond_type"][esel], batch=np.concatenate(batch), num_graphs=g_off, spans=spans)


def _done_indices(out_dir):
    done = set()
    for f in glob.glob(os.path.join(out_dir, "samples_[0-9]*.npz")):
        done.update(int(k.split("_")[-1]) for k in np.load(f).files if k.startswith("pos_gen_"))
    return done


def _batch_path(out_dir, bmols):
    """Output file of one batch, named by what it holds (smallest / largest molecule index of the batch), never by a
    batch counter: a --resume run plans its batches over the molecules still missing, and a counter restarting at 0
    would overwrite files of the earlier run that hold other, finished molecules."""
    base = os.path.join(out_dir, "samples_%05d_%05d" % (min(m["index"] for m in bmols), max(m["index"] for m in bmols)))
    path, k = base + ".npz", 0
    while os.path.exists(path):
        k += 1
        path = "%s_r%d.npz" % (base, k)
    return path


def _save_npz_atomic(path, arrays):
    # the temporary's name must not match the `samples_[0-9]*.npz` globs of _done_indices / merge_outputs: a run killed
    # mid-write leaves it behind, and np.load of a truncated archive would end the next --resume
    tmp = os.path.join(os.path.dirname(path), "." + os.path.basename(path) + ".tmp")
    with open(tmp, "wb") as f:
        np.savez_compressed(f, **arrays)
    os.replace(tmp, path)


def merge_outputs(out_dir):
    merged = {}
    for f in sorted(glob.glob(os.path.join(out_dir, "samples_[0-9]*.npz"))):
        z = np.load(f)
        merged.update({k: z[k] for k in z.files})
    _save_npz_atomic(os.path.join(out_dir, "samples_all.npz"), merged)
    return merged


def run_job(model, mols, out_dir, confs_of, max_atoms, sampler_kwargs, device, save_traj=False, resume=False,
            rank=0, world=1, shard=False, log=print):
    """Plan, sample and save (the loop of scripts/test.py:128-181 over packed batches).  Returns the merged result
    dict on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    os.makedirs(out_dir, exist_ok=True)
    done = set()
    if resume:
        # every rank must plan the same batches: rank 0 lists the finished molecules, the others take its list
        box = [sorted(_done_indices(out_dir)) if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        done = set(box[0])
    mols = [m for m in mols if m["index"] not in done]
    batches = plan_batches(mols, confs_of, sharded_capacity(max_atoms, world) if shard else max_atoms)
    mine = [bidx for bidx in range(len(batches)) if shard or bidx % world == rank]
    # The next batch is packed -- and its topology built -- while the GPU samples the current one, in a background thread (the
    # sampling loop spends its time inside library calls, which release the interpreter lock).  Measured on the MI355X box, four
    # 196 k-atom batches x 600 steps (tools/job_wall.py, profiles/r06_job_wall.json): sampling + saving alone 10.13 s, preparation
    # inline 11.3 s (0.29 s per batch), thread 10.8 s, worker process 11.8 s.
    # AGDIFF_PREPARE=process: a worker PROCESS (agdiff_amd/prep_worker.py, a fresh interpreter started as a child; host work only)
    # -- it shares nothing with the launch loop, but its reply (~100 MB of index arrays per batch) comes back through a pipe and a
    # pickle, which costs the main thread more than the thread's lock contention on a host with fast cores; for hosts where the
    # numpy build takes seconds per batch.  =inline: on the main thread, when the batch's turn comes (measurements, debugging).
    from concurrent.futures import Future, ThreadPoolExecutor
    prep = lambda bidx: prepare_batch(model, batches[bidx], confs_of, rank if shard else 0, world if shard else 1)
    how = os.environ.get("AGDIFF_PREPARE", "inline" if os.environ.get("AGDIFF_PREPARE_INLINE", "0") not in ("", "0") else "thread")
    inline = how == "inline"
    worker, pool = None, None
    if how == "process" and hasattr(model, "prepare_topology") and hasattr(model, "config") and len(mine) > 1:
        try:
            from .prep_worker import Client
            worker = Client()                         # (a child process: boots while the first batch is prepared here)
        except Exception as e:                        # (no worker: the thread)
            log("run_job: no worker process for the batch preparation (%s: %s); using a thread" % (type(e).__name__, e))
    if worker is None:
        pool = ThreadPoolExecutor(max_workers=1)
    opts = _topology_options(model) if worker is not None else None

    class _FromWorker:
        on_main = False

        def result(self):
            return worker.result()

    def submit(bidx, first=False):
        if inline or (first and worker is not None):
            f = Future()
            f.set_result(bidx)                 # (resolved on the main thread, when the batch's turn comes)
            f.on_main = True
            return f
        if worker is not None:
            worker.submit(batches[bidx], [confs_of(m["num_refs"]) for m in batches[bidx]], rank if shard else 0, world if shard else 1, opts)
            return _FromWorker()
        f = pool.submit(prep, bidx)
        f.on_main = False
        return f
    fut = submit(mine[0], first=True) if mine else None
    try:
        return _run_job_batches(model, batches, mine, fut, submit, prep, inline, shard, device, sampler_kwargs, save_traj, log, out_dir, rank,
                                world)
    finally:
        if worker is not None:
            worker.close()
        if pool is not None:
            pool.shutdown(wait=True)


def _run_job_batches(model, batches, mine, fut, submit, prep, inline, shard, device, sampler_kwargs, save_traj, log, out_dir, rank, world):
    import torch.distributed as dist
    for pos_in_mine, bidx in enumerate(mine):
        bmols = batches[bidx]
        # (first this batch's reply, THEN the next request: the worker writes a reply of ~100 MB into a pipe nobody reads until here,
        # and a request larger than the pipe's buffer sent before that would wait for a reader that is itself waiting)
        try:
            packed, topology = prep(fut.result()) if getattr(fut, "on_main", inline) else fut.result()
        except Exception as e:                   # (a worker that died: this batch on the main thread)
            log("run_job: the prepared batch did not arrive (%s: %s); preparing it here" % (type(e).__name__, e))
            packed, topology = prep(bidx)
        try:
            fut = submit(mine[pos_in_mine + 1]) if pos_in_mine + 1 < len(mine) else None
        except Exception as e:                   # (the worker's pipe is gone: the next batch on the main thread, when its turn comes)
            log("run_job: the next batch could not be handed to the worker (%s: %s)" % (type(e).__name__, e))
            from concurrent.futures import Future
            fut = Future()
            fut.set_result(mine[pos_in_mine + 1])
            fut.on_main = True
        if shard:
            from .dist import sample_batch_sharded
            pos, traj, ok = sample_batch_sharded(model, packed, device, sampler_kwargs, save_traj=save_traj, log=log,
                                                 topology=topology)
            if rank != 0:
                continue
        else:
            pos, traj, ok = sample_batch(model, packed, device, sampler_kwargs, save_traj=save_traj, log=log, topology=topology)
        if not ok.any():
            log("batch %d: every molecule failed twice (NaN); skipped: %s" % (bidx, [m["name"] for m in bmols]))
            continue
        out = {}
        for m, (off, n, g), good in zip(bmols, packed["spans"], ok):
            if not good:
                log("molecule %s failed twice (NaN); skipped" % m["name"])
                continue
            out["pos_gen_%d" % m["index"]] = pos[off:off + n * g].numpy().reshape(g, n, 3)
            out["name_%d" % m["index"]] = np.str_(m["name"])
            if traj is not None:
                out["traj_%d" % m["index"]] = traj[:, off:off + n * g].numpy().reshape(traj.shape[0], g, n, 3)
        _save_npz_atomic(_batch_path(out_dir, bmols), out)
        log("rank %d: batch %d/%d (%d of %d molecules, %d conformers) saved" % (rank, bidx + 1, len(batches),
                                                                               int(ok.sum()), len(bmols), packed["num_graphs"]))
    if world > 1:
        dist.barrier()                       # every rank's batch files are on disk
    if SAMPLE_STATS["range_trips"]:
        log("rank %d: %d conformers left the split-fp16 range; their molecules were sampled again in split-bf16 (%d extra passes)"
            % (rank, SAMPLE_STATS["range_trips"], SAMPLE_STATS["bf16x3_retries"]))
    if SAMPLE_STATS["dropped"]:
        log("rank %d: %d molecules were dropped (a NaN in every attempt)" % (rank, SAMPLE_STATS["dropped"]))
    return merge_outputs(out_dir) if rank == 0 else None


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True, help="reference checkpoint (dict with 'config' and 'model', train.py:219-231)")
    ap.add_argument("--testset", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--num-confs", default="2x")
    ap.add_argument("--start-idx", type=int, default=0)
    ap.add_argument("--end-idx", type=int, default=200)
    ap.add_argument("--n-steps", type=int, default=5000)
    ap.add_argument("--w-global", type=float, default=1.0)
    ap.add_argument("--global-start-sigma", type=float, default=0.5)
    ap.add_argument("--clip", type=float, default=1000.0)
    ap.add_argument("--save-traj", action="store_true")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--extend-order", action="store_true", help="input holds raw bonds: extend to order 3 first")
    ap.add_argument("--max-atoms", type=int, default=196608,
                    help="atoms per packed batch and GPU (≈200 k is where an MI355X samples fastest: fixed per-launch costs are "
                         "amortised and the node features still live in L2 / MALL; 196,608 = 3 full rounds of the node kernels' "
                         "256 x 16 x 16-node workgroups; --save-traj keeps n_steps x atoms x 12 bytes)")
    ap.add_argument("--seed", type=int, default=2021)
    ap.add_argument("--precision", default=None, choices=[None, "f32", "bf16x3", "f16x3"])
    ap.add_argument("--dist-mode", default="shard", choices=["shard", "batches"],
                    help="with several ranks: 'shard' = every packed batch (max-atoms x world atoms) is split into "
                         "contiguous graph ranges, one per rank, with an RCCL all-gather of the positions after each "
                         "denoising step (BASELINE.json north_star); 'batches' = whole batches dealt round-robin")
    ap.add_argument("--trust-ckpt", action="store_true", help="unpickle arbitrary classes from the checkpoint")
    args = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from . import compat, get_model
    from .synth import extend_graph_order_np

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(device)
    own_pg = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        own_pg = True
    torch.manual_seed(args.seed + rank)
    np.random.seed(args.seed + rank)
    ckpt = compat.load_checkpoint(args.ckpt, trust=args.trust_ckpt)
    cfg = compat.model_config(ckpt)
    model = get_model(cfg)
    if args.precision:
        model.precision = args.precision
    model.load_state_dict(ckpt["model"])
    model = model.to(device).eval()

    mols = [m for m in load_testset(args.testset) if args.start_idx <= m["index"] < args.end_idx]
    if args.extend_order:
        for m in mols:
            r, c, t = extend_graph_order_np(m["atom_type"].shape[0], m["edge_index"][0], m["edge_index"][1],
                                            m["edge_type"], order=cfg.edge_order)
            m["edge_index"], m["edge_type"] = np.stack([r, c]), t
    kw = dict(n_steps=args.n_steps, step_lr=1e-6, w_global=args.w_global, global_start_sigma=args.global_start_sigma,
              clip=args.clip)
    run_job(model, mols, args.out, num_confs(args.num_confs), args.max_atoms, kw, device, save_traj=args.save_traj,
            resume=args.resume, rank=rank, world=world, shard=(world > 1 and args.dist_mode == "shard"))
    if own_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
