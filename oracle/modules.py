"""Oracle (CPU, plain torch.nn) restatement of the CelebA model zoo.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Follows /root/reference/models/model.py:
  * weights_init            model.py:8-14
  * Encoder_celeba          model.py:282-328
  * Generator_celeba        model.py:331-378
  * Discriminator_celeba    model.py:381-416
  * VAE                     model.py:419-571

Attribute names and *registration order* are the reference's, so that the
same ``torch.manual_seed`` yields bit-identical initial weights and the same
``state_dict`` keys (``features.0.weight`` ... ``deconv4.bias``).
"""
from dataclasses import dataclass, field
from typing import List

import torch
from torch import nn


@dataclass
class OracleOpt:
    """The three fields of the reference's argparse namespace the models read
    (utils/envsetter.py:41-42,45; defaults from there)."""
    input_channels: int = 3
    n_hidden: int = 128
    n_z: List[int] = field(default_factory=lambda: [256, 8, 8])


def weights_init(m):
    """model.py:8-14: class-name substring match. 'Conv' hits Conv2d and
    ConvTranspose2d (weight ~ N(0, .02), bias untouched); 'BatchNorm' hits
    1d/2d (weight ~ N(1, .02), bias = 0). Linear keeps torch defaults."""
    name = type(m).__name__
    if "Conv" in name:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm" in name:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


def _enc_trunk(cin, width):
    # model.py:449-459 / 289-301: three [conv5x5 s2 p2 -> BN2d -> ReLU]
    chans = [cin, width, 2 * width, 4 * width]
    layers = []
    for a, b in zip(chans[:-1], chans[1:]):
        layers += [nn.Conv2d(a, b, 5, stride=2, padding=2), nn.BatchNorm2d(b), nn.ReLU()]
    return nn.Sequential(*layers)


def _latent_hw(opt):
    """Spatial size of the encoder's last feature map = opt.n_z[1:], (8, 8) for 64x64 inputs.  The
    reference hard-codes 8*8 (model.py:461) and the 16/32/64 output sizes (:558-564); deriving both
    from opt.n_z reproduces it exactly at n_z = [256, 8, 8] and gives the 128x128 / 256x256
    variants of BASELINE configs 4-5 (image side = 8 * n_z[1]) without new kernels."""
    n_z = getattr(opt, "n_z", None)
    return (8, 8) if n_z is None else (int(n_z[1]), int(n_z[2]))


def _enc_head(width, n_hidden, spatial=(8, 8)):
    # model.py:460-471: Linear(16384,2048) -> BN1d -> ReLU -> Linear(2048,n_hidden)
    return nn.Sequential(nn.Linear(width * 4 * spatial[0] * spatial[1], 2048), nn.BatchNorm1d(2048),
                         nn.ReLU(), nn.Linear(2048, n_hidden))


def _bn_relu(c):
    return nn.Sequential(nn.BatchNorm2d(c), nn.ReLU())


class _DecoderMixin:
    """Layers + forward of the decoder (model.py:490-509, 537-566 and
    :340-378). Mixed into both VAE and Generator_celeba."""

    def _build_decoder(self, n_hidden, n_z):
        dim = n_z[0] * n_z[1] * n_z[2]
        self.preprocess = nn.Sequential(nn.Linear(n_hidden, dim), nn.BatchNorm1d(dim), nn.ReLU())
        self.deconv1 = nn.ConvTranspose2d(n_z[0], 256, 5, stride=2, padding=2)
        self.act1 = _bn_relu(256)
        self.deconv2 = nn.ConvTranspose2d(256, 128, 5, stride=2, padding=2)
        self.act2 = _bn_relu(128)
        self.deconv3 = nn.ConvTranspose2d(128, 32, 5, stride=2, padding=2)
        self.act3 = _bn_relu(32)
        self.deconv4 = nn.ConvTranspose2d(32, 3, 5, stride=1, padding=2)
        self.activation = nn.Tanh()

    def _decode(self, code, n_z):
        bs = code.size(0)
        h = self.preprocess(code).view(-1, n_z[0], n_z[1], n_z[2])
        # literal output sizes (model.py:558-564) => output_padding=1 on the s2 layers
        zh, zw = n_z[1], n_z[2]          # (8, 8) in the reference: the literals of model.py:558-564
        h = self.act1(self.deconv1(h, output_size=(bs, 256, 2 * zh, 2 * zw)))
        h = self.act2(self.deconv2(h, output_size=(bs, 128, 4 * zh, 4 * zw)))
        h = self.act3(self.deconv3(h, output_size=(bs, 32, 8 * zh, 8 * zw)))
        return self.activation(self.deconv4(h, output_size=(bs, 3, 8 * zh, 8 * zw)))


class Encoder_celeba(nn.Module):
    """model.py:282-328. forward returns (z, per-sample kld of shape (B,))."""

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
        if eps is None:  # model.py:319 draws on CPU then moves
            eps = torch.randn(mu.size()).to(mu.device)
        z = mu + eps * torch.exp(0.5 * logvar)
        kld = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), 1)
        return z, kld

    def forward(self, x, eps=None):
        bs = x.size(0)
        feat = self.features(x).squeeze()
        return self.reparameterize(feat.view(bs, -1), eps)


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
    """model.py:381-416. forward returns (p:(B,), lth features:(B,2048))."""

    def __init__(self, opt):
        super().__init__()
        self.representation_size = opt.n_z
        dim = opt.n_z[0] * opt.n_z[1] * opt.n_z[2]
        spec = [(opt.input_channels, 32, 1), (32, 128, 2), (128, 256, 2), (256, 256, 2)]
        layers = []
        for a, b, s in spec:
            layers += [nn.Conv2d(a, b, 5, stride=s, padding=2), nn.BatchNorm2d(b), nn.LeakyReLU(0.2)]
        self.convs = nn.Sequential(*layers)
        self.lth_features = nn.Sequential(nn.Linear(dim, 2048), nn.LeakyReLU(0.2))
        self.sigmoid_output = nn.Sequential(nn.Linear(2048, 1), nn.Sigmoid())

    def forward(self, x):
        bs = x.size(0)
        feat = self.lth_features(self.convs(x).view(bs, -1))
        p = self.sigmoid_output(feat)
        return p.squeeze(), feat.squeeze()


class VAE(nn.Module, _DecoderMixin):
    """model.py:419-571: encoder and decoder in one Module."""

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
        inner = self.features(x).squeeze().view(bs, -1)
        return self.x_to_mu(inner), self.x_to_logvar(inner)

    def reparameterize(self, mu, logvar, eps=None):
        std = torch.exp(0.5 * logvar)
        if eps is None:
            eps = torch.randn_like(std)
        return mu + eps * std

    def decode(self, code):
        return self._decode(code, self.representation_size2)

    def forward(self, x, eps=None):
        mu, logvar = self.encode(x)
        return self.decode(self.reparameterize(mu, logvar, eps)), mu, logvar
