"""torch.autograd.Function wrappers: each forward / backward is one or two launches
of the HIP library (disentangle_mlp_amd.ops).  Work that autograd reports as not
needed (``ctx.needs_input_grad``) is skipped -- e.g. the discriminator's weight
gradients while it only relays gradients to the decoder (SURVEY.md section 3.1
item 3: those gradients are discarded by the reference's next ``zero_grad``).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops

# How the gradient of a convolution bias that feeds a train-mode BatchNorm is
# produced.  It is analytically zero (BN subtracts the batch mean); the reference
# accumulates pure rounding noise there.
# BIAS_GRAD_ZERO: the backward hands autograd NO gradient for that bias (None) -- defined as exactly zero without a
# fill + accumulate launch per pass; the trainers keep a persistent all-zero ``.grad`` on those parameters
# (trainer._zero_grads) so that optimizers and gradient exchange see zeros.
BIAS_GRAD_COMPUTE, BIAS_GRAD_ZERO = 0, 1


class Conv5x5Fn(Function):
    """nn.Conv2d(k=5, p=2, stride) -- /root/reference/models/model.py:450 etc."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, bias_grad):
        ctx.stride, ctx.bias_grad = stride, bias_grad
        ctx.save_for_backward(x, w)
        return ops.conv5x5_fwd(x, w, bias, stride)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if x.shape[2] % ctx.stride or x.shape[3] % ctx.stride:
                raise RuntimeError("conv5x5 data gradient needs input sizes divisible by the stride")
            gx = ops.convT5x5_fwd(gy, w, None, ctx.stride)
        if ctx.needs_input_grad[1]:
            gw = ops.conv5x5_wgrad(x, gy, ctx.stride)
        if ctx.needs_input_grad[2]:
            gb = None if ctx.bias_grad == BIAS_GRAD_ZERO else ops.channel_sum(gy)
        return gx, gw, gb, None, None


class ConvT5x5Fn(Function):
    """nn.ConvTranspose2d(k=5, p=2, stride, output_size=stride*in) -- model.py:495-507, :558-564."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, bias_grad):
        ctx.stride, ctx.bias_grad = stride, bias_grad
        ctx.save_for_backward(x, w)
        return ops.convT5x5_fwd(x, w, bias, stride)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.conv5x5_fwd(gy, w, None, ctx.stride)
        if ctx.needs_input_grad[1]:
            gw = ops.conv5x5_wgrad(gy, x, ctx.stride)       # roles swapped
        if ctx.needs_input_grad[2]:
            gb = None if ctx.bias_grad == BIAS_GRAD_ZERO else ops.channel_sum(gy)
        return gx, gw, gb, None, None


_defer_tls = __import__("threading").local()      # .current: the innermost active deferred_wgrad context of this thread


class deferred_wgrad:
    """Context for a phase that runs a network several times and then ONE backward (the discriminator phase: D(x) and
    D(fake), new_betavaegan.py:95-123): the weight gradient of a big Linear layer is computed once, over the
    concatenated batches, by whichever of its backward nodes runs last -- instead of one GEMM per pass (each writing the
    134 MB gradient of the 16384 x 2048 layer) plus autograd's additions.  The sum is the same up to fp32 summation
    order.  Every forward made inside the context must get its backward inside it (checked at exit).

    The bookkeeping (per big weight: forward passes that still owe a backward, and the (gy, x) pairs already seen)
    lives in the context object, which every forward made inside it captures: two trainers stepping in two threads, or
    a backward that runs after its context has closed, never see each other's state (such a late backward just
    computes its own weight gradient)."""

    def __init__(self):
        self.pending, self.stash, self.open = {}, {}, False

    def __enter__(self):
        self._outer = getattr(_defer_tls, "current", None)
        self.open = DEFER_WGRAD
        _defer_tls.current = self if DEFER_WGRAD else None
        return self

    def __exit__(self, *exc):
        _defer_tls.current = self._outer
        self.open = False
        left = sum(self.pending.values())
        self.pending, self.stash = {}, {}
        if left and exc[0] is None:
            raise RuntimeError(f"deferred_wgrad: {left} forward pass(es) of a Linear layer got no backward inside the context")
        return False


_acc_tls = __import__("threading").local()


class accumulate_param_grads:
    """Context for iterations that apply a layer several times before ONE backward (the discriminator on the real and on
    the generated batch, the decoder on the prior sample and on the reconstruction: new_betavaegan.py:99-121, 144-163).
    The first backward pass through a layer hands autograd its parameter gradient as usual and remembers the tensor; a
    later pass ADDS into that tensor inside its own kernel (`accumulate` of vg_conv5x5_wgrad*, vg_bn_act_bwd,
    vg_dot_sigmoid_bce_bwd, addmm_ for Linear) and hands autograd nothing -- instead of a second tensor plus the addition
    autograd would launch (29 of them per iteration).  Floating-point addition is commutative: same bits as autograd's sum.
    Forward passes capture the context object; `reset()` forgets the remembered tensors (call it where gradients are
    zeroed: a new backward must not add into the previous one's tensors).  Outside a context nothing changes."""

    def __init__(self):
        self.acc, self.open = {}, False

    def __enter__(self):
        self._outer = getattr(_acc_tls, "current", None)
        _acc_tls.current = self
        self.open = True
        return self

    def __exit__(self, *exc):
        _acc_tls.current = self._outer
        self.open, self.acc = False, {}
        return False

    def reset(self):
        self.acc = {}


def _acc_ctx():
    return getattr(_acc_tls, "current", None)


def _acc_get(actx, key):
    return actx.acc.get(key) if (actx is not None and actx.open) else None


def _acc_put(actx, key, value):
    """Remembers ALIASES (detach(): a new tensor object on the same storage) -- a second reference to the gradient tensor
    itself would make autograd's AccumulateGrad copy it instead of adopting it as ``.grad`` (71 copies per iteration)."""
    if actx is not None and actx.open:
        actx.acc[key] = tuple(t.detach() for t in value) if isinstance(value, tuple) else value.detach()


DEFER_MIN_WEIGHTS = 1 << 20
DEFER_WGRAD = __import__("os").environ.get("VG_DEFER_WGRAD", "1") != "0"      # 0: deferred_wgrad() does nothing


class LinearFn(Function):
    """nn.Linear (model.py:460-471, 402-408, 490-492).  Layers with >= 2^20 weights run on this package's fp16x3 GEMM
    under the default arithmetic (ops.linear_*; a GEMM whose reduction length is not a multiple of 32 -- the weight
    gradient at batches that are not -- goes to the vendor library), everything else on the vendor fp32 GEMMs (SURVEY.md
    K7; algorithm table: tuned_gemms.py).  Inside `deferred_wgrad()` the weight gradient of a layer with >= 2^20 weights
    is batched over its passes."""

    @staticmethod
    def forward(ctx, x, w, bias, bias_grad=BIAS_GRAD_COMPUTE):
        ctx.save_for_backward(x, w)
        ctx.bias_grad = bias_grad
        ctx.acc = _acc_ctx()
        dctx = getattr(_defer_tls, "current", None)
        ctx.defer = dctx if (dctx is not None and ctx.needs_input_grad[1] and w.numel() >= DEFER_MIN_WEIGHTS) else None
        if ctx.defer is not None:
            dctx.pending[id(w)] = dctx.pending.get(id(w), 0) + 1
        if ops.linear_split_ok(x.shape[1], w.numel()):
            return ops.linear_fwd(x, w, bias)
        return torch.nn.functional.linear(x, w, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.linear_dgrad(gy, w) if ops.linear_split_ok(gy.shape[1], w.numel()) else gy @ w
        if ctx.needs_input_grad[2] and ctx.bias_grad != BIAS_GRAD_ZERO:      # (a bias that feeds a BatchNorm1d: no gradient)
            gb = gy.sum(0)
        if ctx.needs_input_grad[1]:
            wg, wx = gy, x                                   # what the weight gradient is taken over
            d, k = ctx.defer, id(w)
            if d is not None and d.open and d.pending.get(k, 0) > 0:
                d.stash.setdefault(k, []).append((gy, x))
                d.pending[k] -= 1
                if d.pending[k] > 0:
                    wg = None                                # a later pass of this layer does it for all
                else:
                    pairs = d.stash.pop(k)
                    if len(pairs) > 1:
                        wg, wx = torch.cat([p[0] for p in pairs]), torch.cat([p[1] for p in pairs])
            if wg is not None:
                prev = _acc_get(ctx.acc, id(w))
                split = ops.linear_split_ok(wg.shape[0], w.numel())
                if prev is None:
                    gw = ops.linear_wgrad(wg, wx) if split else wg.t() @ wx
                    _acc_put(ctx.acc, id(w), gw)
                elif split:
                    prev.add_(ops.linear_wgrad(wg, wx))
                else:
                    prev.addmm_(wg.t(), wx)                  # the layer's second use: added in the GEMM's epilogue
        return gx, gw, gb, None


class BNActFn(Function):
    """Train-mode BatchNorm1d/2d + {none, ReLU, LeakyReLU(0.2)} -- model.py:451-452 etc.
    running_mean / running_var are updated in place by the kernel.  ``stats``: the statistics slots the producing
    convolution left (ops.conv5x5_fwd(..., want_stats=True)); then the statistics pass over x is skipped."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, act, stats=None):
        if stats is not None:
            count = x.numel() // x.shape[1]
            mean, invstd, scale, shift = ops.bn_finalize_stats(stats, count, gamma, beta, running_mean, running_var,
                                                               eps, momentum)
            y = ops.affine_act(x, scale, shift, act)
        else:
            y, mean, invstd = ops.bn_act_fwd(x, gamma, beta, running_mean, running_var, eps, momentum, act)
        ctx.act = act
        ctx.acc = _acc_ctx()
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        prev = _acc_get(ctx.acc, id(gamma)) if need_p else None
        gx, dg, db = ops.bn_act_bwd(gy.contiguous(), x, gamma, beta, mean, invstd, ctx.act, need_p, accumulate_into=prev)
        if prev is not None:
            dg = db = None                                   # added into the first pass's tensors
        elif need_p:
            _acc_put(ctx.acc, id(gamma), (dg, db))
        return gx, dg, db, None, None, None, None, None, None


class ConvStatsFn(Function):
    """conv / transposed conv that also returns the statistics slots of its output (a non-differentiable side
    product of the kernel's epilogue; an empty tensor when this layer's kernel cannot emit them)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, transposed, bias_grad):
        ctx.stride, ctx.transposed, ctx.bias_grad = stride, transposed, bias_grad
        ctx.acc = _acc_ctx()
        ctx.save_for_backward(x, w)
        conv = ops.convT5x5_fwd if transposed else ops.conv5x5_fwd
        y, stats = conv(x, w, bias, stride, want_stats=True)
        stats = stats if stats is not None else x.new_empty(0)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)      # or autograd zero-fills a "gradient" of the statistics slots every backward
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _):
        if gy is None:
            return (None,) * 6
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        s, tr = ctx.stride, ctx.transposed
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if not tr and (x.shape[2] % s or x.shape[3] % s):
                raise RuntimeError("conv5x5 data gradient needs input sizes divisible by the stride")
            gx = ops.conv5x5_fwd(gy, w, None, s) if tr else ops.convT5x5_fwd(gy, w, None, s)
        if ctx.needs_input_grad[1]:
            prev = _acc_get(ctx.acc, id(w))
            kw = dict(out=prev, accumulate=True) if prev is not None else {}
            gw = ops.conv5x5_wgrad(gy, x, s, **kw) if tr else ops.conv5x5_wgrad(x, gy, s, **kw)
            if prev is not None:
                gw = None                                    # added into the first pass's tensor
            else:
                _acc_put(ctx.acc, id(w), gw)
        if ctx.needs_input_grad[2]:
            gb = None if ctx.bias_grad == BIAS_GRAD_ZERO else ops.channel_sum(gy)
        return gx, gw, gb, None, None, None


class BNConvFn(Function):
    """[train-mode BatchNorm2d + activation] -> [5x5 conv / transposed conv] with the normalised, activated tensor
    NEVER materialised (model.py:451-456, 390-398, 496-505): the BatchNorm's statistics come from the
    slots the producing convolution left (``stats_in``, or one pass over x when there are none), its scale / shift
    and activation are applied by the consuming convolution while it stages its input -- forward, and again by the
    weight gradient in backward.  Backward: data gradient of the convolution (w.r.t. the activated tensor), then the
    ordinary BatchNorm backward against the saved raw x (vg_bn_act_bwd: mask recomputed from x).
    Returns (y, statistics slots of y -- empty when the kernel cannot emit them)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, bias, running_mean, running_var, eps, momentum, act, stride, transposed,
                bias_grad, stats_in):
        count = x.numel() // x.shape[1]
        # bound: max |act(BN(x))| <= |gamma| sqrt(count) + |beta| (None outside the fp16-plane arithmetic): what the
        # convolution -- and, in backward, the weight gradient -- scales the operand it reads through the BatchNorm by
        if stats_in is not None and stats_in.numel():
            mean, invstd, scale, shift, bound = ops.bn_finalize_stats(stats_in, count, gamma, beta, running_mean,
                                                                      running_var, eps, momentum, want_bound=True)
        else:
            mean, invstd, scale, shift, bound = ops.bn_stats(x, gamma, beta, running_mean, running_var, eps, momentum,
                                                             want_bound=True)
        conv = ops.convT5x5_fwd if transposed else ops.conv5x5_fwd
        y, stats = conv(x, w, bias, stride, in_affine=(scale, shift, act, bound), want_stats=True)
        stats = stats if stats is not None else x.new_empty(0)
        ctx.act, ctx.stride, ctx.transposed, ctx.bias_grad = act, stride, transposed, bias_grad
        ctx.acc = _acc_ctx()
        ctx.save_for_backward(x, gamma, beta, mean, invstd, scale, shift, w, bound)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)      # as ConvStatsFn
        return y, stats

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _):
        if gy is None:
            return (None,) * 14
        x, gamma, beta, mean, invstd, scale, shift, w, bound = ctx.saved_tensors
        gy = gy.contiguous()
        s, tr, act = ctx.stride, ctx.transposed, ctx.act
        need_bn = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gx = dg = db = gw = gb = None
        if need_bn:
            ga = ops.conv5x5_fwd(gy, w, None, s) if tr else ops.convT5x5_fwd(gy, w, None, s)    # grad w.r.t. act(BN(x))
            need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            prev = _acc_get(ctx.acc, id(gamma)) if need_p else None
            gx, dg, db = ops.bn_act_bwd(ga, x, gamma, beta, mean, invstd, act, need_p, accumulate_into=prev)
            if prev is not None:
                dg = db = None                               # added into the first pass's tensors
            elif need_p:
                _acc_put(ctx.acc, id(gamma), (dg, db))
        if ctx.needs_input_grad[3]:
            aff = (scale, shift, act, bound)
            prev = _acc_get(ctx.acc, id(w))
            kw = dict(out=prev, accumulate=True) if prev is not None else {}
            gw = ops.conv5x5_wgrad(gy, x, s, in_affine=aff, affine_on_gy=True, **kw) if tr \
                else ops.conv5x5_wgrad(x, gy, s, in_affine=aff, **kw)
            if prev is not None:
                gw = None
            else:
                _acc_put(ctx.acc, id(w), gw)
        if ctx.needs_input_grad[4]:
            gb = None if ctx.bias_grad == BIAS_GRAD_ZERO else ops.channel_sum(gy)
        return gx, dg, db, gw, gb, None, None, None, None, None, None, None, None, None


class BiasActFn(Function):
    """y = act(x + bias[c]) for LeakyReLU(0.2) / tanh / sigmoid -- model.py:404, 509, 408."""

    @staticmethod
    def forward(ctx, x, bias, kind):
        y = ops.bias_act_fwd(x, bias, kind)
        ctx.kind = kind
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = ops.act_bwd(gy.contiguous(), y, ctx.kind)
        gb = ops.channel_sum(gx) if ctx.needs_input_grad[1] else None
        return gx, gb, None


class ReparamKLFn(Function):
    """z = mu + eps*exp(logvar/2) and kl = beta*KL(q||N(0,1)) summed over the batch --
    model.py:532-535 + experiments/new_betavaegan.py:64-65, one fused kernel each way."""

    @staticmethod
    def forward(ctx, mu, logvar, eps, beta):
        z, kl, _ = ops.reparam_kl_fwd(mu, logvar, eps, beta)
        ctx.beta = beta
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(mu, logvar, eps)
        return z, kl

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, gkl):
        mu, logvar, eps = ctx.saved_tensors
        if gz is None and gkl is None:
            return None, None, None, None
        gz = gz.contiguous() if gz is not None else None
        gkl = gkl.contiguous() if gkl is not None else None
        gmu, glv = ops.reparam_kl_bwd(gz, mu, logvar, eps, gkl, ctx.beta)
        return gmu, glv, None, None


class KLRowsFn(Function):
    """Encoder_celeba's per-sample KL (model.py:321) together with z."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        z, _, rows = ops.reparam_kl_fwd(mu, logvar, eps, 1.0, want_rows=True)
        ctx.mark_non_differentiable(rows)
        ctx.save_for_backward(mu, logvar, eps)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" of the rows
        return z, rows

    @staticmethod
    @once_differentiable
    def backward(ctx, gz, _):
        if gz is None:
            return None, None, None
        mu, logvar, eps = ctx.saved_tensors
        gmu, glv = ops.reparam_kl_bwd(gz.contiguous(), mu, logvar, eps, None, 1.0)
        return gmu, glv, None


class SqDiffLossFn(Function):
    """scale * sum((a-b)^2), gradient to ``a`` only (``b`` is a target).
    scale 0.5 = Dis_l / SIM (new_betavaegan.py:67-69), 1.0 = pixel MSE (:71-75)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        loss, ga = ops.sqdiff_loss(a, b, scale, 1.0, want_grad=True)
        ctx.save_for_backward(ga)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (ga,) = ctx.saved_tensors
        return ops.scale_by_scalar(ga, gout.contiguous()), None, None


class BCELossFn(Function):
    """nn.BCELoss() vs a constant label (new_betavaegan.py:53,97,101); ``divisor`` is the
    batch the mean runs over (the global batch under data parallelism)."""

    @staticmethod
    def forward(ctx, p, target, divisor):
        loss, gp = ops.bce_loss(p, target, divisor, 1.0, want_grad=True)
        ctx.save_for_backward(gp)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (gp,) = ctx.saved_tensors
        return ops.scale_by_scalar(gp, gout.contiguous()), None, None


class DotSigmoidBCEFn(Function):
    """Linear(K -> 1) + Sigmoid + nn.BCELoss against a constant label in one launch each way (model.py:406-408,
    new_betavaegan.py:101,118,153-154).  Returns (p (B,), loss); p is not differentiable here (the iteration only reads
    it: mean D(x))."""

    @staticmethod
    def forward(ctx, feat, w, bias, target, divisor):
        p, loss, dlogit = ops.dot_sigmoid_bce_fwd(feat, w, bias, target, divisor, want_grad=True)
        ctx.save_for_backward(feat, w, dlogit)
        ctx.acc = _acc_ctx()
        ctx.has_bias = bias is not None
        ctx.mark_non_differentiable(p)
        ctx.set_materialize_grads(False)
        return p, loss

    @staticmethod
    @once_differentiable
    def backward(ctx, _gp, gloss):
        if gloss is None:
            return None, None, None, None, None
        feat, w, dlogit = ctx.saved_tensors
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        prev = _acc_get(ctx.acc, id(w)) if (need_w and need_b) else None
        gfeat, gw, gb = ops.dot_sigmoid_bce_bwd(dlogit, gloss.contiguous(), feat, w, ctx.needs_input_grad[0], need_w, need_b,
                                                accumulate_into=prev)
        if prev is not None:
            gw = gb = None                                   # added into the first pass's tensors
        elif need_w and need_b:
            _acc_put(ctx.acc, id(w), (gw, gb))
        return gfeat, gw, gb, None, None


# ------------------------------------------------------------ functional API
def conv5x5(x, w, bias, stride, bias_grad=BIAS_GRAD_COMPUTE):
    return Conv5x5Fn.apply(x, w, bias, stride, bias_grad)


def conv_transpose5x5(x, w, bias, stride, bias_grad=BIAS_GRAD_COMPUTE):
    return ConvT5x5Fn.apply(x, w, bias, stride, bias_grad)


def linear(x, w, bias, bias_grad=BIAS_GRAD_COMPUTE):
    return LinearFn.apply(x.contiguous(), w, bias, bias_grad)


def batch_norm_act(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, act=ops.ACT_NONE, stats=None):
    return ops.adopt_amax(BNActFn.apply(x, gamma, beta, running_mean, running_var, eps, momentum, act,
                                        stats if stats is not None and stats.numel() else None))


def conv_with_stats(x, w, bias, stride, transposed=False, bias_grad=BIAS_GRAD_COMPUTE):
    """(y, statistics slots of y) -- see ConvStatsFn."""
    return ConvStatsFn.apply(x, w, bias, stride, transposed, bias_grad)


def bn_act_conv(x, gamma, beta, running_mean, running_var, eps, momentum, act, w, bias, stride, transposed=False,
                bias_grad=BIAS_GRAD_COMPUTE, stats_in=None):
    """conv(act(BN_train(x))) with nothing materialised in between -- see BNConvFn.  Returns (y, stats of y)."""
    return BNConvFn.apply(x, gamma, beta, w, bias, running_mean, running_var, eps, momentum, act, stride, transposed,
                          bias_grad, stats_in)


def bias_act(x, bias, kind):
    return BiasActFn.apply(x, bias, kind)


def reparam_kl(mu, logvar, eps, beta):
    return ReparamKLFn.apply(mu, logvar, eps, float(beta))


def kld_loss(mu, logvar, beta):
    """KLD of experiments/new_betavaegan.py:64-65 (no sampling)."""
    _, kl = ReparamKLFn.apply(mu, logvar, torch.zeros_like(mu), float(beta))
    return kl


def sim_loss(sim_recon, sim_real):
    """SIM / Dis_l of new_betavaegan.py:67-69.  ``sim_real`` is a target (the reference
    leaves it attached, but only the discriminator -- whose gradients are discarded
    in that phase -- would receive anything through it)."""
    return SqDiffLossFn.apply(sim_recon, sim_real.detach(), 0.5)


def reconstruction_loss(recon_x, x):
    """new_betavaegan.py:71-75."""
    return SqDiffLossFn.apply(recon_x, x.detach(), 1.0)


def bce_loss(p, label_value, divisor=None):
    """``label_value``: a float, or a one-element device tensor (see ops.bce_loss)."""
    if not isinstance(label_value, torch.Tensor):
        label_value = float(label_value)
    return BCELossFn.apply(p.contiguous(), label_value, divisor)


def dot_sigmoid_bce(feat, w, bias, label_value, divisor=None):
    """(p, bce): see DotSigmoidBCEFn.  ``w`` (1, K) or (K,), ``bias`` (1,)."""
    if not isinstance(label_value, torch.Tensor):
        label_value = float(label_value)
    return DotSigmoidBCEFn.apply(feat.contiguous(), w, bias, label_value, divisor)

