"""GPU (MI355X): the denoising loop's steps as replayed HIP graphs (include/agdiff_hip.h: agdiff_step_graph_capture; epsnet.LangevinRun)
against the launch-by-launch loop: the same bits, on the reference's sampler fixtures, across the switch from local-only steps to
steps with the global branch, over several advance() calls, with the trajectory kept, and with a NaN appearing mid-run."""
import numpy as np
import pytest
import torch

from helpers import check_close, load_golden, sampler_case_cfg, sampler_case_kwargs, t

pytestmark = pytest.mark.gpu


def _model(cfg, head_scale=1e-3, precision="f16x3", graphs=True):
    from agdiff_amd import get_model
    from oracle import agdiff_oracle as O
    sd = O.synth_state_dict_for(cfg, head_scale=head_scale)
    m = get_model(cfg)
    m.precision = precision
    m.step_graphs = graphs
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    return m.to("cuda:0").eval()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("case", ["g5_sampler_top", "g5_sampler_lowT_global", "g5_sampler_mixed_cliplocal"])
def test_sampler_fixtures_by_graph_replay(case, precision):
    g = load_golden(case)
    cfg = sampler_case_cfg(g, case)
    args = [t(g[k]).cuda() for k in ("atom_type", "pos_init", "bond_index", "bond_type", "batch")]
    outs = {}
    for graphs in (True, False):
        m = _model(cfg, head_scale=float(g["head_scale"]), precision=precision, graphs=graphs)
        run = m.begin_sampling(args[0], args[1], args[2], args[3], args[4], int(g["num_graphs"]), False, n_steps=int(g["n_steps"]),
                               noise=t(g["noise"]).cuda(), **sampler_case_kwargs(g))
        run.advance(run.remaining())
        pos, traj = run.finish()
        assert (run.graph_steps > 0) == (graphs and int(g["n_steps"]) > 2), (run.graph_steps, run._use_graphs)
        outs[graphs] = (pos.cpu(), torch.stack(traj))
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    check_close("step_graphs traj[%s]" % case, outs[True][1].numpy(), g["traj"], precision)
    check_close("step_graphs pos[%s]" % case, outs[True][0].numpy(), g["pos_final"], precision)


def test_graphs_across_the_global_switch_and_several_advance_calls():
    """A schedule that starts with local-only steps and ends with the global branch on (four graphs: two parities of each kind),
    advanced in uneven pieces; without injected noise both runs draw the same normals from the same seed (one buffer refilled in
    place in both modes would be a different consumption of the generator: the launch-by-launch run is given the graph run's
    policy through noise injection instead)."""
    from agdiff_amd import drugs_model_config, synth
    cfg = drugs_model_config(num_diffusion_timesteps=60, beta_end=1e-2)
    b = synth.make_packed_batch("drugs", 3, 4, seed=15)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(8)
    pos_init = torch.randn(at.shape[0], 3, generator=gen).cuda()
    noise = torch.randn(60, at.shape[0], 3, generator=gen).cuda()
    probe = _model(cfg, graphs=False)
    sig = ((1.0 - probe.alphas).sqrt() / probe.alphas.sqrt()).cpu()
    start = float(sig[25])                                   # the global branch comes on 25 steps before the end
    kw = dict(extend_order=False, n_steps=60, w_global=1.0, global_start_sigma=start, clip=1000.0, noise=noise)
    outs, used = {}, {}
    for graphs in (True, False):
        m = _model(cfg, graphs=graphs)
        run = m.begin_sampling(at, pos_init, bi, bt, ba, b["num_graphs"], **kw)
        for piece in (7, 1, 20, 3, 29):
            run.advance(piece)
        assert run.remaining() == 0
        pos, traj = run.finish()
        outs[graphs] = (pos.cpu(), torch.stack(traj))
        used[graphs] = (run.graph_steps, run.global_steps, len(run._graphs))
    assert 0 < used[True][1] < 60 and used[True][0] >= 40 and used[True][2] >= 3 and used[False][0] == 0, used
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    # without injected noise: runs, finite, reproducible from the seed
    res = []
    for _ in range(2):
        torch.manual_seed(5)
        m = _model(cfg, graphs=True)
        pos, _ = m.langevin_dynamics_sample_diffusion(at, pos_init, bi, bt, ba, b["num_graphs"], extend_order=False, n_steps=60,
                                                      w_global=1.0, global_start_sigma=start, clip=1000.0)
        res.append(pos.cpu())
    assert torch.isfinite(res[0]).all() and torch.equal(res[0], res[1])


def test_nan_mid_run_raises_from_a_graph_run_too():
    from agdiff_amd import qm9_model_config, synth
    cfg = qm9_model_config(num_diffusion_timesteps=40)
    m = _model(cfg, graphs=True)
    b = synth.make_packed_batch("qm9", 3, 2, seed=3)
    at, bi, bt, ba = [t(b[k]).cuda() for k in ("atom_type", "bond_index", "bond_type", "batch")]
    gen = torch.Generator().manual_seed(1)
    pos_init = torch.randn(at.shape[0], 3, generator=gen)
    noise = torch.randn(40, at.shape[0], 3, generator=gen)
    noise[9, 4, 1] = float("nan")
    with pytest.raises(FloatingPointError):
        m.langevin_dynamics_sample_diffusion(at, pos_init.cuda(), bi, bt, ba, b["num_graphs"], extend_order=False, n_steps=40,
                                             noise=noise.cuda(), nan_check_every=8)
