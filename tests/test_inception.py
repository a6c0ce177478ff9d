"""the Inception pool_3 network of scoring/inception.py.  **Parity unpinned**: the pretrained weights are
not obtainable offline, so these tests pin the ARCHITECTURE arithmetic (the torchvision block wiring, the four FID
pooling patches of scoring/inception.py:200-310, the eval-mode BatchNorm fold, the im2col + GEMM lowering) against the
CPU oracle on seeded random weights, the weight-file contract (torchvision's state_dict names) and the get_fid
plumbing."""
import numpy as np
import pytest
import torch

from oracle import fid as ofid
from oracle.inception import random_fid_inception


@pytest.fixture(scope="module")
def ref():
    return random_fid_inception(3)


def test_weight_file_contract_and_loud_failures(ref, tmp_path):
    from disentangle_mlp_amd.inception import InceptionV3, _FidInception
    # the module tree IS the weight-file layout: torchvision inception_v3 names, aux-less, fc 2048 -> 1008
    keys = list(_FidInception().state_dict())
    assert keys == list(ref.state_dict())
    assert keys[0] == "Conv2d_1a_3x3.conv.weight" and "Mixed_7c.branch_pool.bn.running_var" in keys and keys[-2:] == ["fc.weight", "fc.bias"]
    assert _FidInception().state_dict()["fc.weight"].shape == (1008, 2048)
    with pytest.raises(RuntimeError, match="pt_inception-2015-12-05"):
        InceptionV3()                                            # no weights: never random parameters silently
    bad = dict(ref.state_dict())
    bad.pop("Mixed_6e.branch7x7dbl_5.conv.weight")
    with pytest.raises(RuntimeError):
        InceptionV3(weights=bad)                                 # strict load
    path = tmp_path / "pt_inception-2015-12-05-test.pth"
    torch.save(ref.state_dict(), path)
    m = InceptionV3(weights=str(path))
    assert m.BLOCK_INDEX_BY_DIM[2048] == 3 and not any(p.requires_grad for p in m.parameters())


def test_architecture_matches_the_oracle_on_cpu_tensors(ref):
    """The product's lowering (folded BatchNorm, unfold + GEMM) against plain nn.Conv2d / nn.BatchNorm2d(eval): every
    block output, 75 x 75 inputs without resize (all pooling / padding edge cases), 3e-5 relative L2."""
    from disentangle_mlp_amd.inception import InceptionV3
    m = InceptionV3([0, 1, 2, 3], resize_input=False, weights=ref.state_dict())
    x = torch.rand(2, 3, 75, 75, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        outs = m(x)
        want = ref(x, resize_input=False)
    assert [o.shape[1] for o in outs] == [64, 192, 768, 2048]
    e = float((outs[3] - want).norm() / want.norm())
    assert e <= 3e-5, e


def test_get_fid_on_image_folders_end_to_end(ref, tmp_path):
    """get_fid(path_data, path_pretrained, inception=<weights>) on two folders of PNGs: activations from the product
    network, statistics and Frechet distance from the product arithmetic, against the oracle network + oracle FID."""
    from PIL import Image
    from disentangle_mlp_amd import fid
    rng = np.random.RandomState(4)
    folders = []
    for name, shift in (("a", 0), ("b", 40)):
        d = tmp_path / name
        d.mkdir()
        for i in range(6):
            img = np.clip(rng.randint(0, 200, size=(32, 32, 3)) + shift, 0, 255).astype(np.uint8)
            Image.fromarray(img).save(d / f"{i}.png")
        folders.append(d)
    wpath = tmp_path / "pt_inception-2015-12-05-rand.pth"
    torch.save(ref.state_dict(), wpath)
    got = fid.get_fid(str(folders[0]), str(folders[1]), inception=str(tmp_path), device="cpu")
    stats = []
    for d in folders:
        files = list(d.glob("*.jpg")) + list(d.glob("*.png"))
        x = torch.from_numpy(np.stack([np.asarray(Image.open(f).convert("RGB"), dtype=np.float32) for f in files]))
        with torch.no_grad():
            act = ref(x.permute(0, 3, 1, 2) / 255.0).reshape(len(files), -1).double().numpy()
        stats.append(ofid.activation_statistics(act))
    want = ofid.frechet_distance(*stats[0], *stats[1])
    assert abs(got - want) <= 1e-5 * max(abs(want), 1.0), (got, want)


@pytest.mark.gpu
def test_device_features_match_the_oracle(ref):
    """On the MI355X: pool_3 features of 64 x 64 images (resized to 299 x 299 as the reference does) vs the CPU oracle,
    1e-4 relative L2 (fp32 GEMMs of up to 3456-term dot products through 94 layers)."""
    from disentangle_mlp_amd.inception import InceptionFeatureExtractor
    ex = InceptionFeatureExtractor(ref.state_dict(), device="cuda", batch_size=4)
    imgs = torch.randint(0, 256, (6, 64, 64, 3), generator=torch.Generator().manual_seed(2)).float()
    got = ex(imgs).cpu()
    torch.set_num_threads(16)
    with torch.no_grad():
        want = ref(imgs.permute(0, 3, 1, 2) / 255.0).reshape(6, -1)
    assert got.shape == (6, 2048)
    e = float((got - want).norm() / want.norm())
    assert e <= 1e-4, e
