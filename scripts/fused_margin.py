"""How far the fused conv <-> BatchNorm path and the two-pass path are apart, gradient tensor by gradient tensor, at batch 8
(tests/test_step_gpu.py::test_fused_conv_bn_equals_two_pass_batchnorm[8] allows 1e-3): prints the eight largest relative L2
differences.  Normally ~1e-6 everywhere; a single activation unit landing on the other side of zero in one of the paths
shows up as a jump to ~1e-3 on the tensors behind it (the 16x16x32 form of the Linear GEMM)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from oracle import steps as osteps
from disentangle_mlp_amd import model as M, trainer as T
for batch in (8,):
    b = {k: v.cuda() for k, v in osteps.synthetic_batch(batch).items()}
    res = {}
    prev = M.FUSE_CONV_BN
    for fused in (False, True):
        M.FUSE_CONV_BN = fused
        tr = T.BetaVAEGANTrainer(beta=25.0, lr=0.0)
        grads = {}
        out = tr.step(b["data"], b["noise"], b["eps2"], b["eps3"],
                      grad_hook=lambda ph, net: grads.__setitem__(ph, {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}))
        res[fused] = grads
    M.FUSE_CONV_BN = prev
    worst = []
    for ph in res[False]:
        for k, r in res[False][ph].items():
            if float(r.norm()) == 0.0: continue
            worst.append((float((res[True][ph][k] - r).norm() / r.norm()), ph, k))
    worst.sort(reverse=True)
    print(batch, worst[:8])
