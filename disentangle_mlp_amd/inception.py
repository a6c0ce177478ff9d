"""Inception pool_3 feature extractor for FID on the device: the ``InceptionV3`` of
/root/reference/scoring/inception.py:16-160 with the FID patches of :163-310, forward only.

Contract kept from the reference: class name, constructor ``(output_blocks, resize_input, normalize_input,
requires_grad, use_fid_inception)``, ``BLOCK_INDEX_BY_DIM``, ``forward(inp in [0,1]) -> list of feature maps``, and the
weight file: a ``state_dict`` with torchvision's ``inception_v3`` names (``Conv2d_1a_3x3.conv.weight``,
``Mixed_5b.branch1x1.bn.running_mean``, ..., ``fc.weight``) -- the ``pt_inception-2015-12-05`` file the reference
downloads (scoring/inception.py:13).  There is no network here: pass ``weights=`` (a path or a state_dict); without
weights the constructor raises, it never runs on random parameters silently.

Device path (MI355X): every convolution is lowered to ONE fp32 GEMM -- 1x1 convolutions directly, the others through
``F.unfold`` (im2col; no JIT-compiled convolution library on the path) -- with the eval-mode BatchNorm
(eps = 0.001) folded into the GEMM's weights and bias once at load time and the ReLU applied in place.  This is an
evaluation-side component (5.7 GFLOP per image, 10 000 images per FID): library GEMMs (hipBLASLt) are
the right tool; the hand-written kernels of this package are the training path.

**Parity unpinned**: the pretrained weights and the reference's ``fid_stats_celeba.npz`` are not obtainable offline,
so absolute FID values cannot be compared; the architecture arithmetic is checked against the CPU oracle
(oracle/inception.py) on seeded random weights (tests/test_inception.py).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _ConvBN(nn.Module):
    """BasicConv2d parameters (conv without bias + BatchNorm2d(eps=0.001)); forward = folded GEMM + ReLU."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)
        self._folded = None

    def _load_from_state_dict(self, *a, **k):
        self._folded = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._folded = None                      # .to(device) / .float(): re-fold on first use
        return super()._apply(fn, *a, **k)

    def folded(self):
        if self._folded is None:
            with torch.no_grad():
                s = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
                w = (self.conv.weight * s.view(-1, 1, 1, 1)).reshape(self.conv.out_channels, -1).contiguous()
                b = (self.bn.bias - self.bn.running_mean * s).contiguous()
            self._folded = (w, b)
        return self._folded

    def forward(self, x):
        w, b = self.folded()
        B, C, H, W = x.shape
        kh, kw = self.conv.kernel_size
        sh, sw = self.conv.stride
        ph, pw = self.conv.padding
        OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        if (kh, kw, sh, sw, ph, pw) == (1, 1, 1, 1, 0, 0):
            cols = x.reshape(B, C, H * W)
        else:
            cols = F.unfold(x, (kh, kw), padding=(ph, pw), stride=(sh, sw))        # (B, C*kh*kw, OH*OW)
        y = torch.baddbmm(b.view(1, -1, 1), w.unsqueeze(0).expand(B, -1, -1), cols)
        return torch.relu_(y).view(B, -1, OH, OW)


def _avg3(x):       # Tensorflow's average pool: padded zeros are not counted (scoring/inception.py:203-205)
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=False)


class _A(nn.Module):        # FIDInceptionA
    def __init__(self, cin, pool_features):
        super().__init__()
        self.branch1x1 = _ConvBN(cin, 64, 1)
        self.branch5x5_1 = _ConvBN(cin, 48, 1)
        self.branch5x5_2 = _ConvBN(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = _ConvBN(cin, 64, 1)
        self.branch3x3dbl_2 = _ConvBN(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = _ConvBN(96, 96, 3, padding=1)
        self.branch_pool = _ConvBN(cin, pool_features, 1)

    def forward(self, x):
        return torch.cat([self.branch1x1(x), self.branch5x5_2(self.branch5x5_1(x)),
                          self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          self.branch_pool(_avg3(x))], 1)


class _B(nn.Module):        # torchvision InceptionB
    def __init__(self, cin):
        super().__init__()
        self.branch3x3 = _ConvBN(cin, 384, 3, stride=2)
        self.branch3x3dbl_1 = _ConvBN(cin, 64, 1)
        self.branch3x3dbl_2 = _ConvBN(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = _ConvBN(96, 96, 3, stride=2)

    def forward(self, x):
        return torch.cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))),
                          F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class _C(nn.Module):        # FIDInceptionC
    def __init__(self, cin, c7):
        super().__init__()
        self.branch1x1 = _ConvBN(cin, 192, 1)
        self.branch7x7_1 = _ConvBN(cin, c7, 1)
        self.branch7x7_2 = _ConvBN(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = _ConvBN(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = _ConvBN(cin, c7, 1)
        self.branch7x7dbl_2 = _ConvBN(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = _ConvBN(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = _ConvBN(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = _ConvBN(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = _ConvBN(cin, 192, 1)

    def forward(self, x):
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        return torch.cat([self.branch1x1(x), b7, bd, self.branch_pool(_avg3(x))], 1)


class _D(nn.Module):        # torchvision InceptionD
    def __init__(self, cin):
        super().__init__()
        self.branch3x3_1 = _ConvBN(cin, 192, 1)
        self.branch3x3_2 = _ConvBN(192, 320, 3, stride=2)
        self.branch7x7x3_1 = _ConvBN(cin, 192, 1)
        self.branch7x7x3_2 = _ConvBN(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = _ConvBN(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = _ConvBN(192, 192, 3, stride=2)

    def forward(self, x):
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([self.branch3x3_2(self.branch3x3_1(x)), b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


class _E(nn.Module):        # FIDInceptionE_1 (avg) / FIDInceptionE_2 (max: scoring/inception.py:299-303)
    def __init__(self, cin, pool):
        super().__init__()
        self.pool = pool
        self.branch1x1 = _ConvBN(cin, 320, 1)
        self.branch3x3_1 = _ConvBN(cin, 384, 1)
        self.branch3x3_2a = _ConvBN(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = _ConvBN(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = _ConvBN(cin, 448, 1)
        self.branch3x3dbl_2 = _ConvBN(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = _ConvBN(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = _ConvBN(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = _ConvBN(cin, 192, 1)

    def forward(self, x):
        t = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(t), self.branch3x3_2b(t)], 1)
        t = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(t), self.branch3x3dbl_3b(t)], 1)
        p = _avg3(x) if self.pool == "avg" else F.max_pool2d(x, kernel_size=3, stride=1, padding=1)
        return torch.cat([self.branch1x1(x), b3, bd, self.branch_pool(p)], 1)


class _FidInception(nn.Module):
    """``fid_inception_v3()`` of scoring/inception.py:163-185 (the module tree whose state_dict is the weight file)."""

    def __init__(self):
        super().__init__()
        self.Conv2d_1a_3x3 = _ConvBN(3, 32, 3, stride=2)
        self.Conv2d_2a_3x3 = _ConvBN(32, 32, 3)
        self.Conv2d_2b_3x3 = _ConvBN(32, 64, 3, padding=1)
        self.Conv2d_3b_1x1 = _ConvBN(64, 80, 1)
        self.Conv2d_4a_3x3 = _ConvBN(80, 192, 3)
        self.Mixed_5b, self.Mixed_5c, self.Mixed_5d = _A(192, 32), _A(256, 64), _A(288, 64)
        self.Mixed_6a = _B(288)
        self.Mixed_6b, self.Mixed_6c, self.Mixed_6d, self.Mixed_6e = _C(768, 128), _C(768, 160), _C(768, 160), _C(768, 192)
        self.Mixed_7a = _D(768)
        self.Mixed_7b, self.Mixed_7c = _E(1280, "avg"), _E(2048, "max")
        self.fc = nn.Linear(2048, 1008)          # in the weight file; not used for pool_3 features


class InceptionV3(nn.Module):
    """scoring/inception.py:16-160."""

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=(DEFAULT_BLOCK_INDEX,), resize_input=True, normalize_input=True, requires_grad=False,
                 use_fid_inception=True, weights=None):
        super().__init__()
        if not use_fid_inception:
            raise NotImplementedError("only the FID Inception (use_fid_inception=True) is part of the reference's path")
        if weights is None:
            raise RuntimeError(
                "InceptionV3 needs the pt_inception-2015-12-05 state_dict (scoring/inception.py:13): it cannot be "
                "downloaded here -- pass weights=<path to the .pth file or a state_dict>")
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"
        net = _FidInception()
        sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, bytes)) or hasattr(weights, "__fspath__") \
            else weights
        net.load_state_dict(sd)                   # strict: a wrong file fails loudly
        self.net = net.eval()
        for p in self.parameters():
            p.requires_grad = requires_grad

    def forward(self, inp):
        """inp (B,3,H,W) in [0,1] -> list of the requested blocks' feature maps (ascending block index)."""
        n, outp, x = self.net, [], inp
        if self.resize_input:
            x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
        if self.normalize_input:
            x = 2 * x - 1
        stages = (
            lambda t: F.max_pool2d(n.Conv2d_2b_3x3(n.Conv2d_2a_3x3(n.Conv2d_1a_3x3(t))), kernel_size=3, stride=2),
            lambda t: F.max_pool2d(n.Conv2d_4a_3x3(n.Conv2d_3b_1x1(t)), kernel_size=3, stride=2),
            lambda t: n.Mixed_6e(n.Mixed_6d(n.Mixed_6c(n.Mixed_6b(n.Mixed_6a(n.Mixed_5d(n.Mixed_5c(n.Mixed_5b(t)))))))),
            lambda t: F.adaptive_avg_pool2d(n.Mixed_7c(n.Mixed_7b(n.Mixed_7a(t))), (1, 1)),
        )
        for idx, stage in enumerate(stages):
            x = stage(x)
            if idx in self.output_blocks:
                outp.append(x)
            if idx == self.last_needed_block:
                break
        return outp


class InceptionFeatureExtractor:
    """``feature_extractor`` of fid.get_fid: images [n,h,w,3] with values 0..255 (what fid.py:68-105 feeds the
    network) -> pool_3 activations [n,2048], on the device, in chunks of ``batch_size``."""

    def __init__(self, weights, device="cuda", batch_size=50):
        self.device = torch.device(device)
        self.model = InceptionV3([InceptionV3.DEFAULT_BLOCK_INDEX], weights=weights).to(self.device)
        self.batch_size = int(batch_size)

    @torch.no_grad()
    def __call__(self, images):
        x = torch.as_tensor(images)
        outs = []
        for s in range(0, x.shape[0], self.batch_size):
            xb = x[s:s + self.batch_size].to(self.device, torch.float32).permute(0, 3, 1, 2).contiguous() / 255.0
            outs.append(self.model(xb)[0].reshape(xb.shape[0], -1))
        return torch.cat(outs)
