"""Oracle (CPU, fp64 by default) per-op references: the ATen calls the reference's
nn.Modules dispatch to, restated with explicit arguments.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.
"""
import torch
import torch.nn.functional as F


def _d(t, dtype):
    return None if t is None else t.detach().to("cpu", dtype)


def conv5x5(x, w, b, stride, dtype=torch.float64):
    """nn.Conv2d(k=5, padding=2) forward, model.py:450."""
    return F.conv2d(_d(x, dtype), _d(w, dtype), _d(b, dtype), stride=stride, padding=2)


def convT5x5(x, w, b, stride, dtype=torch.float64):
    """nn.ConvTranspose2d(k=5, padding=2) with output_size = stride*input (model.py:558-564)."""
    return F.conv_transpose2d(_d(x, dtype), _d(w, dtype), _d(b, dtype), stride=stride, padding=2,
                              output_padding=stride - 1)


def conv5x5_grads(x, w, gy, stride, dtype=torch.float64):
    x, w, gy = _d(x, dtype).requires_grad_(), _d(w, dtype).requires_grad_(), _d(gy, dtype)
    F.conv2d(x, w, None, stride=stride, padding=2).backward(gy)
    return x.grad, w.grad


def convT5x5_grads(x, w, gy, stride, dtype=torch.float64):
    x, w, gy = _d(x, dtype).requires_grad_(), _d(w, dtype).requires_grad_(), _d(gy, dtype)
    F.conv_transpose2d(x, w, None, stride=stride, padding=2, output_padding=stride - 1).backward(gy)
    return x.grad, w.grad


def bn_act(x, gamma, beta, act, eps=1e-5, momentum=0.1, gy=None, dtype=torch.float64):
    """Train-mode batch norm + activation; returns y, running stats (from 0/1), and grads if gy."""
    x, gamma, beta = _d(x, dtype).requires_grad_(), _d(gamma, dtype).requires_grad_(), _d(beta, dtype).requires_grad_()
    C = x.shape[1]
    rm, rv = torch.zeros(C, dtype=dtype), torch.ones(C, dtype=dtype)
    z = F.batch_norm(x, rm, rv, gamma, beta, True, momentum, eps)
    y = {"none": lambda t: t, "relu": F.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2)}[act](z)
    out = dict(y=y.detach(), rm=rm, rv=rv)
    if gy is not None:
        y.backward(_d(gy, dtype))
        out.update(gx=x.grad, gw=gamma.grad, gb=beta.grad)
    return out
