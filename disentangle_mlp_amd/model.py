"""Drop-in module API of the reference's CelebA model zoo, computed on MI355X by
the HIP library.

Same class names, constructor signatures ``(opt[, representation_size=64])``,
method names, return orders / shapes and ``state_dict`` keys as
/root/reference/models/model.py:
    weights_init            model.py:8-14
    Encoder_celeba          model.py:282-328
    Generator_celeba        model.py:331-378
    Discriminator_celeba    model.py:381-416
    VAE                     model.py:419-571
so a reference checkpoint loads here and vice versa.  Layers are thin
subclasses of the torch layer classes (they only hold parameters / buffers and
keep ``weights_init``'s class-name matching and the seed -> weights recipe
bit-identical); their ``forward`` runs the hand-written kernels.  The 5x5
convolutions, BatchNorm+activation, reparameterisation, tanh / LeakyReLU /
sigmoid epilogues are HIP; the Linear GEMMs go to hipBLASLt through
``torch.nn.functional.linear``.

There is no CPU path: calling a module with CPU tensors raises.
"""
import torch
from torch import nn
import torch.nn.functional as tF

from . import functional as F
from . import ops


def weights_init(m):
    """model.py:8-14 (class-name substring match; works on the Hip* layers too)."""
    name = type(m).__name__
    if "Conv" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


# --------------------------------------------------------------------- layers
class HipConv2d(nn.Conv2d):
    """5x5 / pad 2 / stride 1|2 convolution.  ``bn_shadowed``: the bias feeds a train-mode
    BatchNorm, so its gradient is analytically zero and is returned as exact zeros."""

    def __init__(self, cin, cout, stride, bn_shadowed=True):
        super().__init__(cin, cout, 5, stride=stride, padding=2)
        self.bn_shadowed = bn_shadowed

    def forward(self, x):
        mode = F.BIAS_GRAD_ZERO if self.bn_shadowed else F.BIAS_GRAD_COMPUTE
        return F.conv5x5(x.contiguous(), self.weight, self.bias, self.stride[0], mode)


class HipConvTranspose2d(nn.ConvTranspose2d):
    """5x5 / pad 2 / stride 1|2 transposed convolution producing exactly stride*input."""

    def __init__(self, cin, cout, stride, bn_shadowed=True):
        super().__init__(cin, cout, 5, stride=stride, padding=2)
        self.bn_shadowed = bn_shadowed

    def forward(self, x, output_size=None):
        s = self.stride[0]
        if output_size is not None:
            want = tuple(output_size)[-2:]
            if want != (x.shape[2] * s, x.shape[3] * s):
                raise ValueError(f"requested output size {want} is not stride*input "
                                 f"{(x.shape[2] * s, x.shape[3] * s)}")
        mode = F.BIAS_GRAD_ZERO if self.bn_shadowed else F.BIAS_GRAD_COMPUTE
        return F.conv_transpose5x5(x.contiguous(), self.weight, self.bias, s, mode)


class _HipBatchNormMixin:
    """Train-mode batch norm fused with the following activation.  The reference never
    calls .eval(), so only batch statistics exist here;
    eval mode is rejected rather than silently approximated."""
    act = ops.ACT_NONE

    def _init_pending(self):
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(_flush_nbt)

    def _note_forward(self):
        if not self.training:
            raise RuntimeError("HipBatchNorm: eval-mode (running-statistics) normalisation is not part "
                               "of the reference's path and is not implemented")
        self._nbt_pending += 1     # num_batches_tracked, materialised lazily (no launch per call)

    def forward(self, x, stats=None):
        """``stats``: statistics slots left by the producing convolution (skips the statistics pass)."""
        self._note_forward()
        return F.batch_norm_act(x.contiguous(), self.weight, self.bias, self.running_mean, self.running_var,
                                self.eps, self.momentum, self.act, stats)

    def _load_from_state_dict(self, *a, **k):
        self._nbt_pending = 0
        return super()._load_from_state_dict(*a, **k)


def _flush_nbt(module, *args):
    if module._nbt_pending:
        module.num_batches_tracked += module._nbt_pending
        module._nbt_pending = 0


class HipBatchNorm2d(_HipBatchNormMixin, nn.BatchNorm2d):
    def __init__(self, c, act=ops.ACT_NONE):
        super().__init__(c)
        self.act = act
        self._init_pending()


class HipBatchNorm1d(_HipBatchNormMixin, nn.BatchNorm1d):
    def __init__(self, c, act=ops.ACT_NONE):
        super().__init__(c)
        self.act = act
        self._init_pending()


class FusedIntoBN(nn.Module):
    """Placeholder keeping the reference's nn.Sequential indices (e.g. ``features.2`` is the
    ReLU): the activation itself runs inside the preceding HipBatchNorm kernel."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def forward(self, x):
        return x

    def extra_repr(self):
        return f"{self.what} (fused into the preceding BatchNorm kernel)"


class HipLeakyReLU(nn.Module):
    def forward(self, x):
        return F.bias_act(x.contiguous(), None, ops.EW_LRELU)


class HipTanh(nn.Module):
    def forward(self, x):
        return F.bias_act(x.contiguous(), None, ops.EW_TANH)


class HipSigmoid(nn.Module):
    def forward(self, x):
        return F.bias_act(x.contiguous(), None, ops.EW_SIGMOID)


class HipLinear(nn.Linear):
    """Linear layer on the vendor fp32 GEMM (hipBLASLt / rocBLAS through torch; trainers load the measured
    algorithm table of tuned_gemms.py).  ``bn_shadowed``: the bias feeds a train-mode BatchNorm1d (x_to_mu.0,
    x_to_logvar.0, preprocess.0: model.py:461-462, 467-468, 491-492), so its gradient is analytically zero and is defined
    as exactly zero like the convolution biases' (SURVEY.md section 3.1 item 9).  Refuses CPU tensors like every other
    layer here."""

    def __init__(self, fin, fout, bn_shadowed=False):
        super().__init__(fin, fout)
        self.bn_shadowed = bn_shadowed

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("disentangle_mlp_amd modules need CUDA/ROCm tensors (no CPU fallback)")
        # big layers (and the shadowed ones) go through this package's Function: the batched weight gradient of
        # functional.deferred_wgrad(), no bias-gradient reduction where it is zero by construction
        if (self.weight.numel() >= F.DEFER_MIN_WEIGHTS or self.bn_shadowed) and x.dim() == 2:
            return F.linear(x, self.weight, self.bias, F.BIAS_GRAD_ZERO if self.bn_shadowed else F.BIAS_GRAD_COMPUTE)
        return tF.linear(x, self.weight, self.bias)


def shadowed_bias_params(net):
    """The convolution biases whose gradient is defined as exactly zero (they feed a train-mode BatchNorm: HipConv2d /
    HipConvTranspose2d with ``bn_shadowed``): their backward returns no gradient at all, see functional.BIAS_GRAD_ZERO."""
    return [m.bias for m in net.modules()
            if isinstance(m, (HipConv2d, HipConvTranspose2d, HipLinear)) and m.bn_shadowed and m.bias is not None]


# --------------------------------------------------------------- building blocks
# Conv <-> BatchNorm fusion: inside a chain conv -> BN -> act -> conv -> ... the statistics of every
# BatchNorm come from the producing convolution's epilogue, and its normalise + activation are applied by the
# CONSUMING convolution while it loads -- the normalised tensor of an inner layer is never written.  False: every
# BatchNorm runs its own two passes (the round-1 path; tests compare the two).
FUSE_CONV_BN = True
# The discriminator's head + BCE as one kernel each way (Discriminator_celeba.forward_with_bce); False /
# VG_FUSE_HEAD=0: Linear, Sigmoid and the BCE kernel one after the other (A/B timing, tests).
FUSE_HEAD_BCE = __import__("os").environ.get("VG_FUSE_HEAD", "1") != "0"


def _has_hooks(mods):
    """Forward (pre-)hooks on any of the modules, or registered globally: the fused chain calls the kernels of several
    modules at once and would never fire them, so a hooked chain runs module by module instead (feature taps,
    profilers and parity probes keep working; same numbers up to the summation order of the statistics)."""
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks for m in mods)


def run_conv_bn_chain(mods, x):
    """Forward of [conv | transposed conv | HipBatchNorm2d | FusedIntoBN placeholder] modules in order, fused as
    above; a BatchNorm that is last in the chain (its consumer is not one of these convolutions) is materialised,
    still without its statistics pass when the producer left statistics."""
    pending, stats = None, None          # BatchNorm not applied yet; statistics slots of the current x

    def flush(t):
        nonlocal pending, stats
        if pending is not None:
            t = pending(t, stats=stats)
            pending, stats = None, None
        return t
    for m in mods:
        if isinstance(m, (HipConv2d, HipConvTranspose2d)):
            tr = isinstance(m, HipConvTranspose2d)
            mode = F.BIAS_GRAD_ZERO if m.bn_shadowed else F.BIAS_GRAD_COMPUTE
            if pending is not None:
                bn = pending
                bn._note_forward()
                x, stats = F.bn_act_conv(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                                         bn.act, m.weight, m.bias, m.stride[0], tr, mode, stats)
                pending = None
            else:
                x, stats = F.conv_with_stats(x.contiguous(), m.weight, m.bias, m.stride[0], tr, mode)
        elif isinstance(m, HipBatchNorm2d):
            x = flush(x)
            pending = m
        elif isinstance(m, FusedIntoBN):
            continue
        else:
            x = flush(x)
            x, stats = m(x), None
    return flush(x)


class FusedChain(nn.Sequential):
    """nn.Sequential with the reference's indices (``features.0`` ... ``features.8``, ``convs.0`` ...), run fused.
    (The class name must contain neither "Conv" nor "BatchNorm": weights_init matches class names, model.py:8-14.)"""

    def forward(self, x):
        mods = list(self)
        if not FUSE_CONV_BN or _has_hooks(mods):
            return super().forward(x)
        return run_conv_bn_chain(mods, x)


def _enc_trunk(cin, width):
    chans = [cin, width, 2 * width, 4 * width]
    layers = []
    for a, b in zip(chans[:-1], chans[1:]):
        layers += [HipConv2d(a, b, 2), HipBatchNorm2d(b, ops.ACT_RELU), FusedIntoBN("ReLU")]
    return FusedChain(*layers)


def _latent_hw(opt):
    """Spatial size of the encoder's last feature map = opt.n_z[1:], (8, 8) for 64x64 inputs.  The
    reference hard-codes 8*8 (model.py:461) and the 16/32/64 output sizes (:558-564); deriving both
    from opt.n_z reproduces it exactly at n_z = [256, 8, 8] and gives the 128x128 / 256x256
    variants of BASELINE configs 4-5 (image side = 8 * n_z[1]) without new kernels."""
    n_z = getattr(opt, "n_z", None)
    return (8, 8) if n_z is None else (int(n_z[1]), int(n_z[2]))


def _enc_head(width, n_hidden, spatial=(8, 8)):
    return nn.Sequential(HipLinear(width * 4 * spatial[0] * spatial[1], 2048, bn_shadowed=True), HipBatchNorm1d(2048, ops.ACT_RELU),
                         FusedIntoBN("ReLU"), HipLinear(2048, n_hidden))


def _bn_relu(c):
    return nn.Sequential(HipBatchNorm2d(c, ops.ACT_RELU), FusedIntoBN("ReLU"))


class _DecoderMixin:
    def _build_decoder(self, n_hidden, n_z):
        dim = n_z[0] * n_z[1] * n_z[2]
        self.preprocess = nn.Sequential(HipLinear(n_hidden, dim, bn_shadowed=True), HipBatchNorm1d(dim, ops.ACT_RELU),
                                        FusedIntoBN("ReLU"))
        self.deconv1 = HipConvTranspose2d(n_z[0], 256, 2)
        self.act1 = _bn_relu(256)
        self.deconv2 = HipConvTranspose2d(256, 128, 2)
        self.act2 = _bn_relu(128)
        self.deconv3 = HipConvTranspose2d(128, 32, 2)
        self.act3 = _bn_relu(32)
        self.deconv4 = HipConvTranspose2d(32, 3, 1, bn_shadowed=False)
        self.activation = HipTanh()

    def _decode(self, code, n_z):
        bs = code.size(0)
        h = self.preprocess(code).view(-1, n_z[0], n_z[1], n_z[2])
        zh, zw = n_z[1], n_z[2]          # (8, 8) in the reference: the literals of model.py:558-564
        chain = [self.deconv1, *self.act1, self.deconv2, *self.act2, self.deconv3, *self.act3, self.deconv4]
        if FUSE_CONV_BN and not _has_hooks(chain + [self.act1, self.act2, self.act3]):
            # deconv1 -> act1 -> deconv2 -> act2 -> deconv3 -> act3 -> deconv4 as one fused chain
            h = run_conv_bn_chain(chain, h.contiguous())
            return self.activation(h)
        h = self.act1(self.deconv1(h, output_size=(bs, 256, 2 * zh, 2 * zw)))
        h = self.act2(self.deconv2(h, output_size=(bs, 128, 4 * zh, 4 * zw)))
        h = self.act3(self.deconv3(h, output_size=(bs, 32, 8 * zh, 8 * zw)))
        return self.activation(self.deconv4(h, output_size=(bs, 3, 8 * zh, 8 * zw)))


# -------------------------------------------------------------------- the zoo
class Encoder_celeba(nn.Module):
    """model.py:282-328.  forward -> (z, per-sample kld (B,))."""

    def __init__(self, opt, representation_size=64):
        super().__init__()
        self.input_channels = opt.input_channels
        self.n_hidden = opt.n_hidden
        self.features = _enc_trunk(self.input_channels, representation_size)
        self.x_to_mu = _enc_head(representation_size, self.n_hidden, _latent_hw(opt))
        self.x_to_logvar = _enc_head(representation_size, self.n_hidden, _latent_hw(opt))

    def reparameterize(self, x, eps=None):
        mu = self.x_to_mu(x)
        logvar = self.x_to_logvar(x)
        if eps is None:
            eps = torch.randn(mu.size(), device=mu.device)
        return F.KLRowsFn.apply(mu, logvar, eps)

    def forward(self, x, eps=None):
        bs = x.size(0)
        feat = self.features(x)
        return self.reparameterize(ops.keep_amax(feat, feat.view(bs, -1)), eps)


class Generator_celeba(nn.Module, _DecoderMixin):
    """model.py:331-378."""

    def __init__(self, opt):
        super().__init__()
        self.input_size = opt.n_hidden
        self.representation_size = opt.n_z
        self._build_decoder(self.input_size, self.representation_size)

    def forward(self, code):
        return self._decode(code, self.representation_size)


class Discriminator_celeba(nn.Module):
    """model.py:381-416.  forward -> (p (B,), lth features (B, 2048))."""

    def __init__(self, opt):
        super().__init__()
        self.representation_size = opt.n_z
        dim = opt.n_z[0] * opt.n_z[1] * opt.n_z[2]
        spec = [(opt.input_channels, 32, 1), (32, 128, 2), (128, 256, 2), (256, 256, 2)]
        layers = []
        for a, b, s in spec:
            layers += [HipConv2d(a, b, s), HipBatchNorm2d(b, ops.ACT_LRELU), FusedIntoBN("LeakyReLU(0.2)")]
        self.convs = FusedChain(*layers)
        self.lth_features = nn.Sequential(HipLinear(dim, 2048), HipLeakyReLU())
        self.sigmoid_output = nn.Sequential(HipLinear(2048, 1), HipSigmoid())

    def forward(self, x):
        bs = x.size(0)
        c = self.convs(x)
        feat = self.lth_features(ops.keep_amax(c, c.view(bs, -1)))
        p = self.sigmoid_output(feat)
        return p.squeeze(), feat.squeeze()

    def forward_with_bce(self, x, label, divisor=None):
        """``forward`` plus ``nn.BCELoss()(p, full(label))`` (new_betavaegan.py:101,118,153-154; ``divisor``: the batch the
        mean runs over -- the global batch under data parallelism) with the head -- Linear(2048 -> 1) + Sigmoid -- and the
        loss in ONE kernel each way.  Returns (p, features, bce)."""
        bs = x.size(0)
        c = self.convs(x)
        feat = self.lth_features(ops.keep_amax(c, c.view(bs, -1)))
        lin = self.sigmoid_output[0]
        if not FUSE_HEAD_BCE or lin._forward_hooks or lin._forward_pre_hooks or self.sigmoid_output[1]._forward_hooks:
            p = self.sigmoid_output(feat).squeeze()              # hooked head: module by module
            return p, feat.squeeze(), F.bce_loss(p, label, divisor)
        p, bce = F.dot_sigmoid_bce(feat, lin.weight, lin.bias, label, divisor)
        return p, feat.squeeze(), bce


class VAE(nn.Module, _DecoderMixin):
    """model.py:419-571.  ``forward(x, eps=None)``: eps may be injected for reproducible
    parity runs (the reference draws it inside ``reparameterize``, model.py:534)."""

    def __init__(self, opt, representation_size=64):
        super().__init__()
        self.input_channels = opt.input_channels
        self.n_hidden = opt.n_hidden
        self.features = _enc_trunk(self.input_channels, representation_size)
        self.x_to_mu = _enc_head(representation_size, self.n_hidden, _latent_hw(opt))
        self.x_to_logvar = _enc_head(representation_size, self.n_hidden, _latent_hw(opt))
        self.input_size = opt.n_hidden
        self.representation_size2 = opt.n_z
        self._build_decoder(self.input_size, self.representation_size2)

    def encode(self, x):
        bs = x.size(0)
        inner = self.features(x).view(bs, -1)
        return self.x_to_mu(inner), self.x_to_logvar(inner)

    def reparameterize(self, mu, logvar, eps=None):
        if eps is None:
            eps = torch.randn_like(mu)
        z, _ = F.reparam_kl(mu, logvar, eps, 0.0)
        return z

    def decode(self, code):
        return self._decode(code, self.representation_size2)

    def forward(self, x, eps=None):
        mu, logvar = self.encode(x)
        return self.decode(self.reparameterize(mu, logvar, eps)), mu, logvar

    def forward_with_kl(self, x, eps, beta):
        """Same as forward plus the fused beta*KL scalar (one kernel for reparam + KL)."""
        mu, logvar = self.encode(x)
        if eps is None:
            eps = torch.randn_like(mu)
        z, kl = F.reparam_kl(mu, logvar, eps, beta)
        return self.decode(z), mu, logvar, kl
