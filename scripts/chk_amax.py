"""Which bounds of max |tensor| (ops.amax_of) a Discriminator forward + backward finds attached by a producer ("cached") and
which it has to measure with a pass of its own.  (The conv-chain entries with a BatchNorm-on-load print MEASURED although their
bound travels in the in_affine tuple: the spy looks at the tensor attribute only.)"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from disentangle_mlp_amd import ops, model
from disentangle_mlp_amd.trainer import ModelOpt
opt = ModelOpt()
enc = model.Encoder_celeba(opt).cuda() if hasattr(model, "Encoder_celeba") else None
D = model.Discriminator_celeba(opt).cuda()
x = torch.randn(32, 3, 64, 64, device="cuda")
calls = []
lib = ops._lib.load()
orig = ops.amax_of
def spy(t, in_affine=None):
    known = getattr(t, "_vg_amax", None)
    calls.append((tuple(t.shape), known is not None and known[0] == t._version))
    return orig(t, in_affine)
ops.amax_of = spy
c = D.convs(x)
print("convs out has bound:", hasattr(c, "_vg_amax"), type(D.convs[-1]).__name__)
p, f = D(x)
(p.sum() + f.sum()).backward()
for s, k in calls: print(s, "cached" if k else "MEASURED")
