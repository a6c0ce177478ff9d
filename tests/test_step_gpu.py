"""GPU parity of the module API and the training iterations against (a) the golden
vectors generated from the imported reference and (b) the oracle run live on the
host CPU with the same seeded inputs.  fp32 path; tolerances are stated per check
(reference fp32-vs-fp64 gap is ~1e-6 relative)."""
import math
import os

import pytest
import torch

from conftest import load_json, BN_SHADOWED, LOSS_TOL, GRADNORM_TOL, gap, check_state
from oracle import steps as osteps

pytestmark = pytest.mark.gpu


def close(a, b, rel, abs_=0.0):
    return abs(a - b) <= abs_ + rel * max(abs(a), abs(b))


@pytest.fixture(scope="module")
def T():
    from disentangle_mlp_amd import trainer
    return trainer


def test_kat0_forward(T):
    g = load_json("kat0.json")["fp64"]
    tr = T.BetaVAEGANTrainer()
    for k in ("features.0.weight", "x_to_mu.0.weight", "deconv1.weight"):
        assert close(float(tr.netEG.state_dict()[k].double().sum()), g["w_sum/" + k], 1e-5, 1e-5)
    gen = torch.Generator().manual_seed(1234)
    x = (torch.rand(4, 3, 64, 64, generator=gen) * 2 - 1).cuda()
    eps = torch.randn(4, 128, generator=gen).cuda()
    noise = torch.randn(4, 128, generator=gen).cuda()
    with torch.no_grad():
        recon, mu, lv = tr.netEG(x, eps)
        fake = tr.netEG.decode(noise)
        p_real, f_real = tr.netD(x)
        p_rec, f_rec = tr.netD(recon)
        p_fake, _ = tr.netD(fake)
    from disentangle_mlp_amd import functional as F
    rel = 2e-5
    assert close(float(recon.double().abs().sum()), g["recon"][1], rel)
    assert close(float(recon.double().sum()), g["recon"][0], rel * 5)
    assert close(float(fake.double().abs().sum()), g["fake"][1], rel)
    assert close(float(mu.double().abs().sum()), g["mu"][1], rel)
    assert close(float(lv.double().abs().sum()), g["logvar"][1], rel)
    for a, b in zip(p_real.tolist() + p_rec.tolist() + p_fake.tolist(), g["p_real"] + g["p_rec"] + g["p_fake"]):
        assert close(a, b, rel)
    for a, b in zip(recon[0, :, ::16, ::16].flatten().tolist(), g["recon_slice"]):
        assert close(a, b, 1e-4, 2e-6)
    for a, b in zip(f_rec[0, :8].tolist(), g["feat_slice"]):
        assert close(a, b, 1e-4, 2e-6)
    assert close(float(F.kld_loss(mu, lv, 1.0)), g["kl_beta1"], rel * 5)
    assert close(float(F.reconstruction_loss(recon, x)), g["mse"], rel)
    assert close(float(F.sim_loss(f_rec, f_real)), g["dis_l"], rel * 5)
    assert close(float(F.bce_loss(p_real, 0.9)), g["bce_real_0.9"], rel)
    assert close(float(F.bce_loss(p_fake, 0.1)), g["bce_fake_0.1"], rel)
    assert int(tr.netEG.state_dict()["features.1.num_batches_tracked"]) == g["bn_nbt"]


@pytest.mark.parametrize("batch", [4, 16])
def test_betavaegan_step_vs_golden(T, batch):
    """One full iteration vs the imported reference's golden vectors.  Phase-1 numbers are
    tight (2e-5).  Later phases follow Adam's first, sign-like update and are chaotic in ANY
    fp32 evaluation order: the reference's own CPU path moves kld by 0.5 % (55235 / 55388 /
    55501 at B=16) when only torch's thread count changes 1 / 3 / 8, and by 0.85 % between two
    hosts at B=4.  Stated tolerances: phase 2 losses 1e-3, mse_enc 2e-3, kld 3e-2; gradient
    norms 5e-3 (D), 1e-2 (EG phase 2), 0.5 (EG phase 3): one ReLU unit rounding to the other side of
    zero moves a layer's gradient by ~1e-3 of its norm."""
    gg = load_json(f"step_b{batch}.json")
    g, g64 = gg["fp32"], gg["fp64"]
    tr = T.BetaVAEGANTrainer(beta=25.0)
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(batch).items()}
    grads = {}

    def hook(ph, net):
        grads[ph] = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
    out = tr.step(b["data"], b["noise"], b["eps2"], b["eps3"], grad_hook=hook)
    losses = {k: float(v) for k, v in out.items()}
    for k, v in g["losses"].items():
        tol = max(LOSS_TOL[k], 5 * gap(v, g64["losses"][k]))
        if k == "D_x":
            assert close(losses["D_x_sum"] / batch, v, tol)
        else:
            assert close(losses[k], v, tol), (k, losses[k], v, tol)
    for ph, key in (("D", "d"), ("EG2", "eg"), ("EG3", "eg")):
        for k, v in g["grad_norms"][ph].items():
            if k in BN_SHADOWED[key]:
                continue
            tol = max(GRADNORM_TOL[ph], 5 * gap(v, g64["grad_norms"][ph][k]))
            assert close(grads[ph][k], v, tol, 1e-6), (ph, k, grads[ph][k], v, tol)
    check_state(tr.netEG.state_dict(), g["eg_state"], g64["eg_state"], BN_SHADOWED["eg"], 1e-3)
    check_state(tr.netD.state_dict(), g["d_state"], g64["d_state"], BN_SHADOWED["d"], 1e-3)
    sd = tr.netD.state_dict()
    assert int(sd["convs.1.num_batches_tracked"]) == 5
    assert int(tr.netEG.state_dict()["features.1.num_batches_tracked"]) == 2
    assert int(tr.netEG.state_dict()["act1.0.num_batches_tracked"]) == 3


@pytest.mark.parametrize("arith,loss_tol,grad_tol,batch", [("fp32", 2e-5, 3e-3, 8), ("bf16x6", 2e-5, 3e-3, 8),
                                                           ("bf16x3", 1e-4, 1e-2, 8), ("bf16x6", 2e-5, 3e-3, 128)])
def test_betavaegan_gradients_vs_live_oracle(T, arith, loss_tol, grad_tol, batch):
    """(Also run in the OPT-IN split-bf16 arithmetics: bf16x6 -- the exact 3-plane split -- at the fp32
    tolerances; bf16x3 -- 4.5e-6 per convolution instead of 5e-7 -- at 1e-4 for losses and 1e-2 for gradients.)
    Per-parameter gradients of all three phases vs the oracle (fp64, host CPU), B=8, with
    lr = 0 so that every phase differentiates at the same (initial) weights: this isolates the
    kernels from the chaotic sensitivity of Adam's first sign-like update.  Tolerance 3e-3
    relative L2 per tensor: one LeakyReLU/ReLU unit whose pre-activation rounds to the other
    side of 0 moves a weight gradient by ~1e-3 of its norm (the reference's own fp32 run shows
    3e-4 vs fp64 at B=16), a real indexing bug moves it by O(1).
    The last case is THE BENCHMARKED CONFIGURATION (BASELINE.json config 2: batch 128, default arithmetic) against
    the fp64 oracle on the host: the launches bench.py times -- ring kernels that are not K-split with their statistics
    epilogue, the one-pass BatchNorm backward, the 256-pixel transposed tiles -- none of which a batch of 8 reaches."""
    eg, d, oeg, od = osteps.build_nets(dtype=torch.float64)
    for o in (oeg, od):
        o.param_groups[0]["lr"] = 0.0
    b = osteps.synthetic_batch(batch, dtype=torch.float64)
    ref_g = {}
    ref_l = osteps.betavaegan_step(eg, d, oeg, od, b["data"], b["noise"], b["eps2"], b["eps3"], beta=25.0,
                                   grad_hook=lambda ph, net: ref_g.__setitem__(
                                       ph, {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    from disentangle_mlp_amd import ops
    tr = T.BetaVAEGANTrainer(beta=25.0, lr=0.0)
    got_g = {}
    prev_arith = ops.CONV_ARITH
    try:
        ops.CONV_ARITH = arith
        out = tr.step(*(b[k].float().cuda() for k in ("data", "noise", "eps2", "eps3")),
                      grad_hook=lambda ph, net: got_g.__setitem__(
                          ph, {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}))
    finally:
        ops.CONV_ARITH = prev_arith
    for k in ("errD_real", "errD_fake", "errG_fake", "errG_recon", "sim", "mse_dec", "kld", "mse_enc"):
        assert close(float(out[k]), ref_l[k], loss_tol), (k, float(out[k]), ref_l[k])
    for ph, key in (("D", "d"), ("EG2", "eg"), ("EG3", "eg")):
        for k, r in ref_g[ph].items():
            if k in BN_SHADOWED[key] and not (k == "x_to_mu.3.bias" and ph == "EG3"):
                continue        # analytically zero; the reference holds rounding noise there
            if float(r.norm()) == 0.0:
                continue
            e = float((got_g[ph][k].double() - r).norm() / float(r.norm()))
            assert e <= grad_tol, (ph, k, e)
    # BatchNorm running statistics after 5 / 2 / 3 forwards: order and count matter
    for net, ref in ((tr.netD, d), (tr.netEG, eg)):
        for (k, v), (_, r) in zip(net.state_dict().items(), ref.state_dict().items()):
            if "running" in k:
                e = float((v.cpu().double() - r).norm() / max(float(r.norm()), 1e-30))
                assert e <= (1e-4 if arith == "bf16x3" else 2e-5), (k, e)
            if "num_batches" in k:
                assert int(v) == int(r), k


def test_adam_step_vs_live_oracle(T):
    """The optimizer side: after the discriminator phase (no chaos yet) the updated D weights
    match the oracle's, except on the measure-zero set where Adam's m/sqrt(v) sign flips."""
    batch = 8
    eg, d, oeg, od = osteps.build_nets()
    b = osteps.synthetic_batch(batch)
    osteps.betavaegan_step(eg, d, oeg, od, b["data"], b["noise"], b["eps2"], b["eps3"], beta=25.0)
    tr = T.BetaVAEGANTrainer(beta=25.0)
    tr.step(*(b[k].cuda() for k in ("data", "noise", "eps2", "eps3")))
    for (k, v), (_, r) in zip(tr.netD.state_dict().items(), d.state_dict().items()):
        if k in BN_SHADOWED["d"] or "num_batches" in k or "running" in k:
            continue
        diff = (v.cpu().double() - r.double()).abs()
        # every element moved by exactly +-lr; a flipped sign costs 2*lr on that element
        assert float(diff.mean()) <= 5e-6, (k, float(diff.mean()))
        assert float((diff > 1.5e-3).double().mean()) <= 2e-3, k


def test_vae_and_gan_steps_vs_golden(T):
    g = load_json("vae_step_b16.json")["fp32"]
    tr = T.VAETrainer(beta=1.0, lr=3e-3)
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(16).items()}
    out = tr.step(b["data"], b["eps2"])
    assert close(float(out["mse"]), g["losses"]["mse"], 1e-4)
    assert close(float(out["kld"]), g["losses"]["kld"], 1e-4)
    g64 = load_json("vae_step_b16.json")["fp64"]
    check_state(tr.model.state_dict(), g["state"], g64["state"], BN_SHADOWED["eg"], 3e-3)
    gg = load_json("gan_step_b4.json")
    g, g64 = gg["fp32"], gg["fp64"]
    tg = T.GANTrainer(lr=3e-3)
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(4).items()}
    out = tg.step(b["data"], b["noise"])
    for k in ("errD_real", "errD_fake", "errG"):     # errG follows D's first (sign-like, lr 3e-3) Adam step: conftest.LOSS_TOL
        assert close(float(out[k]), g["losses"][k], max(LOSS_TOL[k], 5 * gap(g["losses"][k], g64["losses"][k]))), \
            (k, float(out[k]), g["losses"][k])
    check_state(tg.netG.state_dict(), g["g_state"], g64["g_state"], BN_SHADOWED["g"], 3e-3)
    check_state(tg.netD.state_dict(), g["d_state"], g64["d_state"], BN_SHADOWED["d"], 3e-3)


def test_checkpoint_roundtrip_with_oracle(T, tmp_path):
    """A HIP-side checkpoint has the reference's keys (incl. the 'module.' prefix on the
    discriminator) and loads into the oracle modules; and back."""
    tr = T.BetaVAEGANTrainer()
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(4).items()}
    tr.step(b["data"], b["noise"], b["eps2"], b["eps3"])
    path = tmp_path / "model_1.tar"
    tr.save(str(tmp_path / "legacy.tar"), 1, legacy_format=True)      # torch-1.3.1-readable pickle
    assert set(torch.load(str(tmp_path / "legacy.tar"), map_location="cpu")) >= {"epoch", "discriminator_model"}
    tr.save(str(path), 1)
    ck = torch.load(str(path), map_location="cpu")
    assert set(ck) == {"epoch", "encoder_decoder_model", "discriminator_model", "encoder_decoder_optimizer",
                       "discriminator_optimizer"}
    assert all(k.startswith("module.") for k in ck["discriminator_model"])
    eg, d, oeg, od = osteps.build_nets()
    eg.load_state_dict(ck["encoder_decoder_model"])
    torch.nn.DataParallel(d).load_state_dict(ck["discriminator_model"])
    oeg.load_state_dict(ck["encoder_decoder_optimizer"])
    od.load_state_dict(ck["discriminator_optimizer"])
    tr2 = T.BetaVAEGANTrainer(seed=1)
    assert tr2.load(str(path)) == 1
    for (k, a), (_, c) in zip(tr.netEG.state_dict().items(), tr2.netEG.state_dict().items()):
        assert torch.equal(a, c), k
    # the oracle continues from the checkpoint exactly where the HIP engine does
    with torch.no_grad():
        r_cpu, _, _ = eg(b["data"].cpu(), b["eps2"].cpu())
        r_gpu, _, _ = tr2.netEG(b["data"], b["eps2"])
    e = float((r_gpu.cpu().double() - r_cpu.double()).norm() / r_cpu.double().norm())
    assert e < 2e-5, e


def test_modules_reject_cpu_and_eval(T):
    from disentangle_mlp_amd.model import VAE
    m = VAE(T.ModelOpt())
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 64, 64))
    m = m.cuda().eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 64, 64).cuda())


def test_flat_gradient_buffers_match_plain_path(T):
    """data_parallel=True (flat fp32 gradient buffers, the layout RCCL all-reduces) at world
    size 1 must give the same iteration as the plain path."""
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(4).items()}
    res = []
    for dp in (False, True):
        tr = T.BetaVAEGANTrainer(beta=25.0, data_parallel=dp)
        out = tr.step(b["data"], b["noise"], b["eps2"], b["eps3"])
        res.append(({k: float(v) for k, v in out.items()}, tr))
    for k in ("errD_real", "errD_fake", "errG_fake", "errG_recon", "sim", "mse_dec"):
        assert close(res[0][0][k], res[1][0][k], 1e-6), k
    for (k, a), (_, c) in zip(res[0][1].netD.state_dict().items(), res[1][1].netD.state_dict().items()):
        assert float((a.double() - c.double()).abs().max()) <= 2.1e-3, k     # <= one Adam sign flip
        assert float((a.double() - c.double()).abs().mean()) <= 1e-6, k


def test_encoder_celeba_and_random_eps_paths(T):
    """Encoder_celeba (model.py:282-328) returns (z, per-sample kld); VAE.forward draws eps itself
    when none is injected (model.py:534)."""
    import oracle
    from disentangle_mlp_amd import model as M
    torch.manual_seed(5)
    ref = oracle.Encoder_celeba(oracle.OracleOpt())
    enc = M.Encoder_celeba(T.ModelOpt())
    enc.load_state_dict(ref.state_dict())
    enc = enc.cuda()
    g = torch.Generator().manual_seed(11)
    x = torch.rand(6, 3, 64, 64, generator=g) * 2 - 1
    eps = torch.randn(6, 128, generator=g)
    with torch.no_grad():
        z_ref, kld_ref = ref(x, eps)
    z, kld = enc(x.cuda(), eps.cuda())
    assert z.shape == (6, 128) and kld.shape == (6,)
    assert float((z.detach().cpu() - z_ref).norm() / z_ref.norm()) < 2e-5
    assert float((kld.detach().cpu() - kld_ref).norm() / kld_ref.norm()) < 2e-5
    z.sum().backward()                                   # gradient flows to the trunk
    assert enc.features[0].weight.grad is not None and float(enc.features[0].weight.grad.abs().sum()) > 0
    vae = M.VAE(T.ModelOpt()).cuda()
    r1, mu1, _ = vae(x.cuda())
    r2, mu2, _ = vae(x.cuda())
    assert r1.shape == (6, 3, 64, 64) and float((r1 - r2).abs().max()) > 0       # fresh eps each call
    assert float(r1.abs().max()) <= 1.0                                           # tanh range


def test_load_accepts_prefixless_discriminator_keys(T):
    """test.py-era checkpoints store netD without the DataParallel 'module.' prefix
    (utils/generate_samples_recons.py:23,31); load() takes both layouts."""
    tr = T.BetaVAEGANTrainer()
    ck = tr.checkpoint(3)
    ck["discriminator_model"] = {k[len("module."):]: v for k, v in ck["discriminator_model"].items()}
    tr2 = T.BetaVAEGANTrainer(seed=2)
    assert tr2.load(ck) == 3
    for (k, a), (_, b) in zip(tr.netD.state_dict().items(), tr2.netD.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("capturable", [False, True])
def test_hip_adam_matches_torch_adam_and_shares_checkpoints(T, capturable):
    """optim.HipAdam (vg_adam_step; capturable: vg_adam_prepare + vg_adam_step_dev, the scalars formed on the device)
    against torch.optim.Adam on the same gradients for 4 steps: parameters and moments agree to a few ulp; state_dict
    round-trips in both directions."""
    from disentangle_mlp_amd import optim as _optim
    import functools
    HipAdam = functools.partial(_optim.HipAdam, capturable=capturable)
    g = torch.Generator().manual_seed(21)
    shapes = [(7,), (33, 5), (256, 128, 5, 5), (2048, 1031), (3,), (64, 3, 5, 5)] + [(5, 5)] * 30     # > 24 tensors: 2 launches
    ps_a = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ps_b = [torch.nn.Parameter(p.detach().clone()) for p in ps_a]
    oa, ob = HipAdam(ps_a, lr=1e-3), torch.optim.Adam(ps_b, lr=1e-3)
    assert isinstance(oa, torch.optim.Adam) and isinstance(oa, _optim.HipAdam)
    for it in range(4):
        for pa, pb in zip(ps_a, ps_b):
            gr = torch.randn(*pa.shape, generator=g).cuda() * (10.0 ** (it - 2))
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if it == 2:
            ps_a[0].grad = ps_b[0].grad = None            # a parameter without a gradient is skipped, its step too
        oa.step()
        ob.step()
    for pa, pb in zip(ps_a, ps_b):
        assert float((pa - pb).abs().max()) <= 2e-6 * max(float(pb.abs().max()), 1.0)
        for key in ("exp_avg", "exp_avg_sq"):
            a, b = oa.state[pa][key], ob.state[pb][key]
            assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1e-30) + 1e-30, key
        assert float(oa.state[pa]["step"]) == float(ob.state[pb]["step"])
    # checkpoints: torch -> Hip and Hip -> torch
    ps_c = [torch.nn.Parameter(p.detach().clone()) for p in ps_b]
    oc = HipAdam(ps_c, lr=1e-3)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))        # as through torch.save / torch.load (live dicts share tensors)
    ps_d = [torch.nn.Parameter(p.detach().clone()) for p in ps_a]
    od = torch.optim.Adam(ps_d, lr=1e-3)
    od.load_state_dict(copy.deepcopy(oa.state_dict()))
    for pc, pb, pd, pa in zip(ps_c, ps_b, ps_d, ps_a):
        gr = torch.randn(*pc.shape, generator=g).cuda()
        pc.grad, pb.grad, pd.grad, pa.grad = gr.clone(), gr.clone(), gr.clone(), gr.clone()
    oc.step(); ob.step(); od.step(); oa.step()
    for pc, pb, pd, pa in zip(ps_c, ps_b, ps_d, ps_a):
        assert float((pc - pb).abs().max()) <= 2e-6 * max(float(pb.abs().max()), 1.0)
        assert float((pd - pa).abs().max()) <= 2e-6 * max(float(pa.abs().max()), 1.0)
    # weight decay is not implemented natively: falls back to torch's step, still correct
    pe, pf = torch.nn.Parameter(torch.ones(10).cuda()), torch.nn.Parameter(torch.ones(10).cuda())
    oe, of = HipAdam([pe], lr=1e-2, weight_decay=0.1), torch.optim.Adam([pf], lr=1e-2, weight_decay=0.1)
    pe.grad, pf.grad = torch.ones(10).cuda(), torch.ones(10).cuda()
    oe.step(); of.step()
    assert torch.allclose(pe, pf)


def test_captured_iteration_equals_eager_bit_for_bit(T):
    """trainer.BetaVAEGANTrainer(graph=True): from the third iteration of a batch shape on, `step` replays a HIP graph of
    the whole iteration (inputs and latents through static buffers, labels read from device memory, Adam's step count on
    the device).  Against the same trainer stepping eagerly: every loss of every iteration, every parameter, BatchNorm
    buffer (incl. num_batches_tracked, counted on the host) and Adam state afterwards -- identical bits.  Labels change
    between replays (the 5 % flips of new_betavaegan.py:89-90), latents are drawn by the trainer itself in one iteration,
    and a second batch shape gets its own capture."""
    from disentangle_mlp_amd.optim import HipAdam
    g = torch.Generator().manual_seed(3)
    batches = [16, 16, 16, 16, 16, 8, 8, 8, 8, 16, 16]
    labels = [(0.9, 0.1), (0.9, 0.1), (0.9, 0.1), (0.1, 0.1), (0.9, 0.9), (0.9, 0.1), (0.9, 0.1), (0.9, 0.1), (0.1, 0.9),
              (0.9, 0.1), (0.9, 0.1)]
    data = [torch.rand(b, 3, 64, 64, generator=g) * 2 - 1 for b in batches]
    lat = [[torch.randn(b, 128, generator=g) for _ in range(3)] for b in batches]
    res = {}
    for mode in ("eager", "graph"):
        tr = T.BetaVAEGANTrainer(beta=25.0, graph=(mode == "graph"), capturable=True)
        assert isinstance(tr.optimizerEG, HipAdam) and tr.optimizerEG.device_scalars
        losses = []
        for i, b in enumerate(batches):
            lt = [None, None, None] if i == 4 else [t.cuda() for t in lat[i]]      # iteration 4: the trainer draws them
            # iteration 9: a hooked iteration stays eager in both trainers -- the replay after it must find Adam's device
            # step counter current
            hook = (lambda ph, net: None) if i == 9 else None
            out = tr.step(data[i].cuda(), *lt, real_label=labels[i][0], fake_label=labels[i][1], grad_hook=hook)
            losses.append({k: v.clone() for k, v in out.items()})
        if mode == "graph":
            assert len(tr._graphs) == 2 and tr.graph                          # both shapes captured, no fallback
        assert tr.iteration == len(batches)
        # a checkpoint loaded in between: torch replaces Adam's moment tensors, so the captures made before must not be
        # replayed (HipAdam.state_generation): one more iteration, back to the checkpoint, four iterations from there
        import copy
        ckpt = copy.deepcopy(tr.checkpoint(1))
        x16, l16 = data[0].cuda(), [t.cuda() for t in lat[0]]
        after = {k: v.clone() for k, v in tr.step(x16, *l16).items()}
        tr.load(ckpt)
        again = [{k: v.clone() for k, v in tr.step(x16, *l16).items()} for _ in range(4)]
        assert all(torch.equal(after[k], again[0][k]) for k in after)         # the same iteration from the same state
        if mode == "graph":
            assert len(tr._graphs) == 3                                       # captured anew after the load
        losses += again
        sd = {f"{n}.{k}": v.detach().clone() for n, net in (("eg", tr.netEG), ("d", tr.netD)) for k, v in net.state_dict().items()}
        for n, o in (("oeg", tr.optimizerEG), ("od", tr.optimizerD)):
            osd = o.state_dict()
            for i, st in osd["state"].items():
                for k, v in st.items():
                    sd[f"{n}.{i}.{k}"] = v.detach().clone()
        res[mode] = (losses, sd)
    for i, (le, lg) in enumerate(zip(res["eager"][0], res["graph"][0])):
        for k in le:
            assert torch.equal(le[k], lg[k]), (i, k, float(le[k]), float(lg[k]))
    assert res["eager"][1].keys() == res["graph"][1].keys()
    for k, v in res["eager"][1].items():
        assert torch.equal(v.cpu(), res["graph"][1][k].cpu()), k
    n_steps = len(batches) + 4                      # + the four iterations after the reload (the one before it was undone)
    assert float(res["graph"][1]["oeg.0.step"]) == 2 * n_steps and float(res["graph"][1]["od.0.step"]) == n_steps
    assert int(res["graph"][1]["d.convs.1.num_batches_tracked"]) == 5 * n_steps


def test_captured_iteration_at_the_benchmarked_batch(T):
    """What bench.py times -- replays of the captured iteration at batch 128 -- against the same five iterations launched
    eagerly: every loss of every iteration and every weight afterwards, identical bits (replays 3..5 are graph launches)."""
    g = torch.Generator().manual_seed(4)
    data = [torch.rand(128, 3, 64, 64, generator=g) * 2 - 1 for _ in range(5)]
    lat = [[torch.randn(128, 128, generator=g) for _ in range(3)] for _ in range(5)]
    res = {}
    for mode in ("eager", "graph"):
        tr = T.BetaVAEGANTrainer(beta=25.0, graph=(mode == "graph"), capturable=True)
        losses = [{k: v.clone() for k, v in tr.step(data[i].cuda(), *(t.cuda() for t in lat[i])).items()} for i in range(5)]
        if mode == "graph":
            assert len(tr._graphs) == 1 and tr.graph
        res[mode] = (losses, {f"{n}.{k}": v.detach().clone() for n, net in (("eg", tr.netEG), ("d", tr.netD))
                              for k, v in net.state_dict().items()})
        del tr
        torch.cuda.empty_cache()
    for i, (le, lg) in enumerate(zip(res["eager"][0], res["graph"][0])):
        for k in le:
            assert torch.equal(le[k], lg[k]), (i, k, float(le[k]), float(lg[k]))
            assert bool(torch.isfinite(le[k]).all())
    for k, v in res["eager"][1].items():
        assert torch.equal(v, res["graph"][1][k]), k


def test_tuned_gemm_table_is_accepted_on_this_installation(T):
    """The shipped vendor-GEMM algorithm table (tuned_gemms.py) was measured on this image: its validators match, so a
    trainer switches TunableOp on in look-up mode (no tuning at run time)."""
    import torch.cuda.tunable as tunable
    from disentangle_mlp_amd import tuned_gemms
    T.BetaVAEGANTrainer(beta=25.0)
    if os.environ.get("VG_TUNED_GEMMS", "1") != "0" and not os.environ.get("PYTORCH_TUNABLEOP_ENABLED"):
        assert tuned_gemms.enable() is True
        assert tunable.is_enabled() and not tunable.tuning_is_enabled()
        assert len(tunable.get_results()) >= 20


def test_vae_and_gan_captured_steps_equal_eager(T):
    """VAETrainer (new_vae.py) and GANTrainer (new_gan.py) replay their iteration as a HIP graph the same way: losses
    of five iterations and the final weights bit for bit against eager stepping."""
    g = torch.Generator().manual_seed(4)
    data = [torch.rand(8, 3, 64, 64, generator=g) * 2 - 1 for _ in range(5)]
    lat = [torch.randn(8, 128, generator=g) for _ in range(5)]
    labels = [(0.9, 0.1), (0.9, 0.1), (0.9, 0.1), (0.1, 0.1), (0.9, 0.9)]
    for make, call, nets in (
            (lambda gr: T.VAETrainer(beta=1.0, graph=gr, capturable=True), lambda tr, i: tr.step(data[i].cuda(), lat[i].cuda()),
             lambda tr: (tr.model,)),
            (lambda gr: T.GANTrainer(graph=gr), lambda tr, i: tr.step(data[i].cuda(), lat[i].cuda(), real_label=labels[i][0],
                                                                      fake_label=labels[i][1]), lambda tr: (tr.netG, tr.netD))):
        res = {}
        for mode in (False, True):
            tr = make(mode)
            if not mode and hasattr(tr, "optimizerG"):       # eager twin of the GAN: the same device-scalar Adam arithmetic
                from disentangle_mlp_amd.optim import HipAdam
                tr.optimizerG = HipAdam(tr.netG.parameters(), lr=3e-3, capturable=True)
                tr.optimizerD = HipAdam(tr.netD.parameters(), lr=3e-3, capturable=True)
            losses = [{k: v.clone() for k, v in call(tr, i).items()} for i in range(5)]
            if mode:
                assert len(tr._graphs) == 1 and tr.graph
            res[mode] = (losses, {f"{j}.{k}": v.detach().clone() for j, n in enumerate(nets(tr)) for k, v in n.state_dict().items()})
        for i in range(5):
            for k, v in res[False][0][i].items():
                assert torch.equal(v, res[True][0][i][k]), (i, k)
        for k, v in res[False][1].items():
            assert torch.equal(v, res[True][1][k]), k


# ------------------------------------------------------------------ parity AWAY from the initial weights
def _tap_unit_signs(mods, store):
    """Forward hooks recording, per call, the signs of a layer's B x 2048 (or B x 16384) post-activation units and a
    signature of the call (the mean of its output: which batch went through)."""
    return [m.register_forward_hook(lambda mod, inp, o, i=i: store.append((i, float(o.detach().double().mean()),
                                                                           (o.detach() > 0).flatten().cpu())))
            for i, m in enumerate(mods)]


def _count_flips(a, b):
    """Units whose sign differs between two runs of the same iteration; calls matched per layer by their signature (the
    two implementations do not visit D's three batches in the same order)."""
    flips, b = 0, list(b)
    for (i, sa, ma) in a:
        j = min((k for k in range(len(b)) if b[k][0] == i and b[k][2].numel() == ma.numel()),
                key=lambda k: abs(b[k][1] - sa))
        flips += int((ma != b[j][2]).sum())
        b.pop(j)
    return flips


def _trained_oracle(k_iters, batch, seed=31, lr=3e-4):
    """The oracle trained for k iterations on the host (fp32, the reference's schedule).  lr 3e-4 rather than the
    reference's 1e-3: on synthetic images D saturates to D(x) = 1.0f within three lr = 1e-3 iterations (fp32
    sigmoid; the BCE then sits on its -100 clamp, where fp32 and fp64 differ by construction and the
    gradients vanish) -- the point here is a set of weights, Adam moments and BatchNorm statistics away from
    the initial ones, not that regime."""
    torch.set_num_threads(16)
    eg, d, oeg, od = osteps.build_nets()
    for o in (oeg, od):
        o.param_groups[0]["lr"] = lr
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(4 * batch, 3, 8, 8, generator=g)
    data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))     # smooth "images"
    for it in range(k_iters):
        x = data[(it % 4) * batch:(it % 4 + 1) * batch]
        no, e2, e3 = (torch.randn(batch, 128, generator=g) for _ in range(3))
        osteps.betavaegan_step(eg, d, oeg, od, x, no, e2, e3, beta=25.0)
    return eg, d, oeg, od, data, g


@pytest.mark.parametrize("batch", [16, 128])
def test_gradients_at_trained_weights_vs_oracle(T, batch):
    """`test_betavaegan_gradients_vs_live_oracle` differentiates at the initial weights only.  Here the ORACLE
    trains k = 2 iterations on the host; its checkpoint (both models + both Adam states, the reference's dict)
    is loaded into the HIP trainer, and ONE lr = 0 iteration on a fresh batch is compared at those trained
    weights: losses 1e-4 and every gradient tensor of all three phases against the fp64 oracle holding the
    same weights, to max(3e-3, 3 x the error of the reference's OWN fp32 arithmetic (the fp32 oracle, same weights,
    same inputs) against that fp64 run) -- ReLU units whose pre-activation rounds to the other side of zero move
    a gradient tensor by ~1e-3 each in either fp32 evaluation.  Phase 3 is thus checked tensor by tensor away from
    init (conftest.GRADNORM_TOL['EG3'] = 0.5 in the golden-vector test is only a chaos bound)."""
    # batch 128: the benchmarked configuration itself, away from the initial weights (round 4; ~2 min of host time for
    # the oracle's five iterations)
    import copy
    eg, d, oeg, od, data, g = _trained_oracle(2, batch)
    ck = {"epoch": 1, "encoder_decoder_model": eg.state_dict(),
          "discriminator_model": {"module." + k: v for k, v in d.state_dict().items()},
          "encoder_decoder_optimizer": oeg.state_dict(), "discriminator_optimizer": od.state_dict()}
    tr = T.BetaVAEGANTrainer(beta=25.0, seed=5)            # different init: everything must come from the checkpoint
    assert tr.load(ck) == 1
    for o in (tr.optimizerEG, tr.optimizerD):
        o.param_groups[0]["lr"] = 0.0
    assert float(tr.optimizerEG.state[next(iter(tr.netEG.parameters()))]["step"]) == 4     # 2 EG steps / iteration
    # fp64 twins of the trained oracle nets
    eg64, d64, oeg64, od64 = osteps.build_nets(dtype=torch.float64)
    eg64.load_state_dict(eg.state_dict())
    d64.load_state_dict(d.state_dict())
    for o in (oeg64, od64):
        o.param_groups[0]["lr"] = 0.0
    x = data[3 * batch:4 * batch]
    no, e2, e3 = (torch.randn(batch, 128, generator=g) for _ in range(3))
    ref_g, got_g, ref32_g = {}, {}, {}
    eg32, d32 = copy.deepcopy(eg), copy.deepcopy(d)            # the reference's own fp32 arithmetic at these weights
    o32 = [torch.optim.Adam(n.parameters(), lr=0.0) for n in (eg32, d32)]
    ref32_l = osteps.betavaegan_step(eg32, d32, o32[0], o32[1], x, no, e2, e3, beta=25.0,
                                     grad_hook=lambda ph, net: ref32_g.__setitem__(
                                         ph, {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}))
    # signs of the units behind the big Linear layers, in the fp64 oracle and in the build: a unit whose pre-activation
    # lands on the other side of zero moves every gradient behind it by ~0.8 / sqrt(units) (1.6e-3 at batch 128) in ANY
    # fp32 evaluation -- the reference's own (e32 below) included
    signs64, signs_hip = [], []
    taps = _tap_unit_signs((eg64.x_to_mu[2], eg64.x_to_logvar[2], eg64.preprocess[2], d64.lth_features[1]), signs64)
    ref_l = osteps.betavaegan_step(eg64, d64, oeg64, od64, x.double(), no.double(), e2.double(), e3.double(), beta=25.0,
                                   grad_hook=lambda ph, net: ref_g.__setitem__(
                                       ph, {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    taps += _tap_unit_signs((tr.netEG.x_to_mu[1], tr.netEG.x_to_logvar[1], tr.netEG.preprocess[1], tr.netD.lth_features[1]),
                            signs_hip)
    out = tr.step(x.cuda(), no.cuda(), e2.cuda(), e3.cuda(),
                  grad_hook=lambda ph, net: got_g.__setitem__(
                      ph, {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}))
    for h in taps:
        h.remove()
    assert 0 < len(signs_hip) <= len(signs64), (len(signs64), len(signs_hip))     # (the oracle may run a layer more often)
    flips = _count_flips(signs_hip, signs64)
    assert 1e-5 < ref_l["D_x"] < 1 - 1e-5, ref_l["D_x"]        # the fp32 sigmoid has not saturated to exactly 0 / 1
    # losses: 1e-4, or 3 x what the reference's own fp32 arithmetic loses against fp64 at these weights (D(x) close to 1
    # makes log(1 - p) sensitive to the fp32 rounding of p)
    for k in ("errD_real", "errD_fake", "errG_fake", "errG_recon", "sim", "mse_dec", "kld", "mse_enc"):
        tol = max(1e-4, 3 * gap(ref32_l[k], ref_l[k]))
        assert close(float(out[k]), ref_l[k], tol, 1e-7), (k, float(out[k]), ref_l[k], ref32_l[k])
    worst = {}
    for ph, key in (("D", "d"), ("EG2", "eg"), ("EG3", "eg")):
        for k, r in ref_g[ph].items():
            if (k in BN_SHADOWED[key] and not (k == "x_to_mu.3.bias" and ph == "EG3")) or float(r.norm()) == 0.0:
                continue
            e = float((got_g[ph][k].double() - r).norm() / float(r.norm()))
            e32 = float((ref32_g[ph][k] - r).norm() / float(r.norm()))
            # the stated 3e-3 (it covers one flipped unit at batch 128), the reference's own fp32 error, or -- when the
            # build and the fp64 oracle disagree on the sign of more than one counted unit -- 1.7e-3 per such unit
            worst[ph] = max(worst.get(ph, (0.0, "", 0.0)), (e / max(3e-3, 3 * e32, 1e-3 + 1.7e-3 * flips), k, e))
    assert all(w[0] <= 1.0 for w in worst.values()), (worst, flips)


def _smooth_batches(n_it, batch, seed=7):
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(4 * batch, 3, 8, 8, generator=g)
    data = torch.tanh(torch.nn.functional.interpolate(base, size=64, mode="bilinear"))
    rnd = [[torch.randn(batch, 128, generator=g) for _ in range(3)] for _ in range(n_it)]
    return data, rnd


def test_short_trajectory_tracks_the_oracle(T):
    """scripts/trajectory_vs_oracle.py as a test: both engines run 6 iterations FREELY on the same inputs (B = 16).
    After Adam's first sign-like step the two are chaotic twins (one LeakyReLU unit of D's 16 x 2048 Dis_l features
    on the other side of zero moves every gradient behind it by 0.8 / sqrt(32768) = 4.4e-3 -- measured between the
    reference's own fp32 and fp64 runs, scripts/diag_kl_signs.py -- and the next sign-like step turns that into tens of
    thousands of weights stepping the other way), so the bounds of a free run are on trends: D(x) within 0.02 absolute,
    reconstruction error within 8 % at every iteration; the first iteration's phase-1 numbers at 2e-5 and its KL at 3 %
    (conftest.LOSS_TOL).  The beta-weighted KL of LATER iterations of a FREE run cannot be bounded usefully: this
    build's iteration-1 value moved from 190 k to 692 k when nothing but the vendor library's algorithm for the Linear
    GEMMs changed (the reference's own three evaluations: 276 k / 361 k / 315 k) -- it is only required to stay finite
    here, and is bounded where it can be, one iteration ahead of a common state, by
    `test_one_iteration_ahead_of_the_oracle_state`."""
    n_it, batch = 6, 16
    torch.set_num_threads(16)
    data, rnd = _smooth_batches(n_it, batch)
    tr = T.BetaVAEGANTrainer(beta=25.0)
    eg, d, oeg, od = osteps.build_nets()
    for it in range(n_it):
        x = data[(it % 4) * batch:(it % 4 + 1) * batch]
        no, e2, e3 = rnd[it]
        out = tr.step(x.cuda(), no.cuda(), e2.cuda(), e3.cuda())
        ref = osteps.betavaegan_step(eg, d, oeg, od, x, no, e2, e3, beta=25.0)
        if it == 0:
            assert close(float(out["errD_real"]), ref["errD_real"], 2e-5) and close(float(out["errD_fake"]), ref["errD_fake"], 2e-5)
        assert abs(float(out["D_x_sum"]) / batch - ref["D_x"]) <= 0.02, (it, float(out["D_x_sum"]) / batch, ref["D_x"])
        assert close(float(out["mse_enc"]), ref["mse_enc"], 0.08), (it, float(out["mse_enc"]), ref["mse_enc"])
        assert close(float(out["mse_dec"]), ref["mse_dec"], 0.08), (it, float(out["mse_dec"]), ref["mse_dec"])
        kl = float(out["kld"])
        assert (abs(kl / ref["kld"] - 1) <= 0.03) if it == 0 else (math.isfinite(kl) and kl > 0), (it, kl, ref["kld"])


def test_one_iteration_ahead_of_the_oracle_state(T):
    """What CAN be bounded after the first Adam step: along the oracle's own trajectory (fp32, lr 1e-3, B = 16) the
    oracle's complete state -- weights, BatchNorm buffers, both Adam states: the reference's checkpoint dict -- is
    loaded into the HIP trainer before EVERY iteration, and into the oracle in fp64; all run the same iteration from
    the same state.  Every loss of that iteration, the beta-weighted KL and the encoder-phase reconstruction error
    after two more Adam steps included, must agree with the oracle's within max(the stated tolerance of
    conftest.LOSS_TOL, 1.5 x the reference's own fp64-vs-fp32 deviation from that state).  Measured
    (scripts/diag_resync.py): KL +0.9 % / -3.3 % / 0.0 % at iterations 0 / 1 / 2 where the reference's fp64 run
    deviates +0.1 % / -13 % / +31 %."""
    import copy
    n_it, batch = 4, 16
    torch.set_num_threads(16)
    data, rnd = _smooth_batches(n_it, batch)
    eg, d, oeg, od = osteps.build_nets()
    tr = T.BetaVAEGANTrainer(beta=25.0, seed=3)            # different init: everything comes from the loaded state
    worst = {}
    for it in range(n_it):
        x = data[(it % 4) * batch:(it % 4 + 1) * batch]
        state = copy.deepcopy({"epoch": it, "encoder_decoder_model": eg.state_dict(),
                               "discriminator_model": {"module." + k: v for k, v in d.state_dict().items()},
                               "encoder_decoder_optimizer": oeg.state_dict(), "discriminator_optimizer": od.state_dict()})
        eg64, d64, oeg64, od64 = osteps.build_nets(dtype=torch.float64)
        eg64.load_state_dict(state["encoder_decoder_model"])
        d64.load_state_dict({k[len("module."):]: v for k, v in state["discriminator_model"].items()})
        oeg64.load_state_dict(copy.deepcopy(state["encoder_decoder_optimizer"]))
        od64.load_state_dict(copy.deepcopy(state["discriminator_optimizer"]))
        r64 = osteps.betavaegan_step(eg64, d64, oeg64, od64, x.double(), *[t.double() for t in rnd[it]], beta=25.0)
        tr.load(copy.deepcopy(state))
        out = tr.step(x.cuda(), *[t.cuda() for t in rnd[it]])
        ref = osteps.betavaegan_step(eg, d, oeg, od, x, *rnd[it], beta=25.0)        # advances the trajectory
        assert abs(float(out["D_x_sum"]) / batch - ref["D_x"]) <= 2e-5 + 1.5 * abs(r64["D_x"] - ref["D_x"]), (it, "D_x")
        for k in ("errD_real", "errD_fake", "errG_fake", "errG_recon", "sim", "mse_dec", "kld", "mse_enc"):
            own = gap(r64[k], ref[k])                       # the reference against itself, from the same state
            tol = max(LOSS_TOL[k], 1.5 * own)
            dev = gap(float(out[k]), ref[k])
            worst[k] = max(worst.get(k, 0.0), dev)
            assert dev <= tol, (it, k, float(out[k]), ref[k], r64[k], tol)
    assert worst["kld"] <= 0.2 and worst["mse_enc"] <= 0.05, worst      # whatever the reference's own spread was


def test_train_epoch_on_device_loader_vs_oracle_loop(T):
    """train_epoch (new_betavaegan.py:77-201) on the device-resident loader against the oracle driven by the same
    batches and the same label stream: the four returned averages (two iterations, pre-chaos quantities at 1e-4)."""
    import numpy as np
    from disentangle_mlp_amd.data import DeviceImageDataset, DeviceLoader
    from oracle import data as odata
    rng = np.random.RandomState(2)
    imgs = rng.randint(0, 256, size=(10, 64, 64, 3), dtype=np.uint8)
    loader = DeviceLoader(DeviceImageDataset(imgs, device="cuda"), 8, shuffle=False)
    tr = T.BetaVAEGANTrainer(beta=25.0, seed=999)
    # the latents the trainer will draw (its own per-rank stream), replayed for the oracle
    from disentangle_mlp_amd.trainer import _latent_generator
    lg = _latent_generator(torch.device("cuda"), 999, 0)
    lat = [[torch.randn(b, 128, device="cuda", generator=lg).cpu() for _ in range(3)] for b in (8, 2)]
    enc, dec, dis, dx = tr.train_epoch(loader, label_rng=np.random.RandomState(9))
    eg, d, oeg, od = osteps.build_nets()
    lrng = np.random.RandomState(9)
    mse_sum = dx_sum = 0.0
    for i, (lo, hi) in enumerate(((0, 8), (8, 10))):
        x = torch.stack([odata.to_tensor_normalize(imgs[j]) for j in range(lo, hi)])
        fake = float(lrng.choice(a=[0.1, 0.9], p=[0.95, 0.05]))
        real = float(lrng.choice(a=[0.1, 0.9], p=[0.05, 0.95]))
        no, e2, e3 = lat[i]
        ref = osteps.betavaegan_step(eg, d, oeg, od, x, no, e2, e3, beta=25.0, real_label=real, fake_label=fake)
        mse_sum += ref["mse_enc"]
        dx_sum += ref["D_x"]
    assert enc == dec and dis == dx
    assert close(enc, mse_sum / 10, 5e-3), (enc, mse_sum / 10)        # mse_enc follows two Adam steps: conftest LOSS_TOL
    assert close(dx, dx_sum / 10, 2e-3), (dx, dx_sum / 10)


@pytest.mark.parametrize("batch", [8, 128])
def test_fused_conv_bn_equals_two_pass_batchnorm(T, batch):
    """Conv <-> BatchNorm fusion (model.FUSE_CONV_BN: statistics from the convolution epilogue, normalise + activation
    applied by the consumer while it loads) against the unfused path (every BatchNorm its own statistics and
    normalise passes): same losses (1e-5), same gradients (1e-3 relative L2 per tensor: only the summation order of
    the statistics differs -- a batch mean moves by ~1e-7 relative -- but at B = 8 one ReLU unit whose pre-activation
    rounds to the other side of zero moves a gradient tensor by up to ~1e-3), same BatchNorm buffers, lr = 0.
    Batch 8: every ring-kernel launch is K-split and takes the fallback statistics pass; batch 128 (the benchmark's):
    the statistics come from the ring kernels' own epilogue (the two-pass path it is compared with is pinned to the
    oracle at small batches)."""
    from disentangle_mlp_amd import model as M
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(batch).items()}
    res, masks = {}, {False: [], True: []}
    prev = M.FUSE_CONV_BN
    try:
        for fused in (False, True):
            M.FUSE_CONV_BN = fused
            tr = T.BetaVAEGANTrainer(beta=25.0, lr=0.0)
            grads = {}
            # signs of the B x 2048 units behind the big Linear layers (materialised on both paths), every call
            taps = [m.register_forward_hook(lambda mod, inp, o, f=fused: masks[f].append((o > 0).flatten().cpu()))
                    for m in (tr.netEG.x_to_mu[1], tr.netEG.x_to_logvar[1], tr.netEG.preprocess[1], tr.netD.lth_features[1])]
            out = tr.step(b["data"], b["noise"], b["eps2"], b["eps3"],
                          grad_hook=lambda ph, net: grads.__setitem__(ph, {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}))
            for h in taps:
                h.remove()
            res[fused] = ({k: float(v) for k, v in out.items()}, grads,
                          {k: v.detach().double().clone() for n in (tr.netEG, tr.netD) for k, v in n.state_dict().items() if "running" in k or "num_batches" in k})
    finally:
        M.FUSE_CONV_BN = prev
    assert len(masks[False]) == len(masks[True]) > 0
    flips = sum(int((a != c).sum()) for a, c in zip(masks[False], masks[True]))
    # what the epilogue feeds directly: the losses and every BatchNorm's running statistics
    for k, v in res[False][0].items():
        assert close(res[True][0][k], v, 1e-5, 1e-7), (k, res[True][0][k], v)
    for k, r in res[False][2].items():
        assert float((res[True][2][k] - r).abs().max()) <= 1e-5 * max(float(r.abs().max()), 1.0), k
    # gradients.  Batch 8: 1e-3.  Batch 128: the suite's stated per-tensor bound 3e-3 -- the two
    # paths differ by ~1e-6 in D's Dis_l features after four layers, so over the three D passes of 128 x 2048 LeakyReLU
    # units about one unit in two runs lands on the other side of zero, and ONE such unit moves every gradient behind
    # it by 0.8 / sqrt(128 * 2048) = 1.6e-3 (measured on the first run of this case: 1.67e-3 on every EG tensor).
    # Round 4: those units are COUNTED (`flips`: sign differences between the two paths over every call of the four
    # B x 2048-unit layers behind the big Linear layers); the stated 3e-3 holds for up to one of them, each further one
    # widens the bound by its quantum (two of them: 3.26e-3 measured on the first fp16x3 run).
    grad_tol = 1e-3 if batch == 8 else max(3e-3, 1e-3 + 1.7e-3 * flips)
    for ph, key in (("D", "d"), ("EG2", "eg"), ("EG3", "eg")):
        for k, r in res[False][1][ph].items():
            if float(r.norm()) == 0.0 or k in BN_SHADOWED[key]:      # shadowed biases: rounding noise on both sides
                continue
            e = float((res[True][1][ph][k] - r).norm() / r.norm())
            assert e <= grad_tol, (ph, k, e, "flipped units:", flips)
