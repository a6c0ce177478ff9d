"""The pieces together, as the reference's experiment script uses them: ImageFolder tree -> get_data_loader
(HBM-resident uint8 cache) -> BetaVAEGANTrainer.step per batch -> sample / reconstruction grids -> checkpoint.
Synthetic images (smooth random patterns); prints losses and file sizes."""
import sys, os, tempfile, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from disentangle_mlp_amd import data, image_io
from disentangle_mlp_amd.trainer import BetaVAEGANTrainer

root = tempfile.mkdtemp(prefix="vg_e2e_")
rng = np.random.default_rng(0)
for split, n in (("train", 512), ("val", 64), ("test", 64)):
    d = os.path.join(root, split, "faces")
    os.makedirs(d)
    for i in range(n):
        low = rng.integers(0, 256, size=(6, 6, 3), dtype=np.uint8)
        Image.fromarray(low).resize((80, 96), Image.BICUBIC).save(os.path.join(d, f"{i:05d}.png"))
opt = types.SimpleNamespace(dataset="celebA", img_size=64, batch_size_train=128, batch_size_val=64, batch_size_test=64,
                            num_workers=8, **{f"image_root_{s}": os.path.join(root, s) for s in ("train", "val", "test")})
torch.manual_seed(999)
train, val, test = data.get_data_loader(opt)
tr = BetaVAEGANTrainer(beta=25.0)
for epoch in range(2):
    for i, (x, _) in enumerate(train):
        out = tr.step(x)
    print(f"epoch {epoch}: " + ", ".join(f"{k}={float(v):.3f}" for k, v in out.items()), flush=True)
    image_io.generate_samples(tr.netEG.decode, epoch, 64, 128, root)
    image_io.gen_reconstructions(lambda t: tr.netEG(t)[0], val, epoch, root, path_for_originals=root)
tr.save(os.path.join(root, "model_2.tar"), 2)
for f in sorted(os.listdir(root)):
    p = os.path.join(root, f)
    if os.path.isfile(p):
        print(f"{f:24s} {os.path.getsize(p):>12d} bytes")
