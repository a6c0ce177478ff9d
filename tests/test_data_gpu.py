"""GPU parity of the image I/O kernels (bit-exact vs the oracle's CPU restatement of the
reference's torchvision transforms / save_image) and the end-to-end get_data_loader path."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import data as OD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    from disentangle_mlp_amd import ops
    return ops


@pytest.mark.parametrize("N,Hs,Ws,C,B", [(50, 64, 64, 3, 128), (7, 5, 7, 3, 9), (20, 8, 8, 1, 4), (300, 64, 64, 3, 300)])
def test_gather_normalize_bit_exact(H, N, Hs, Ws, C, B):
    g = torch.Generator().manual_seed(3)
    u8 = torch.randint(0, 256, (N, Hs, Ws, C), dtype=torch.uint8, generator=g)
    if N >= 2:
        u8[0], u8[1] = 0, 255                                     # every byte value incl. the extremes
    idx = torch.randint(0, N, (B,), generator=g)
    out = H.u8_gather_normalize(u8.cuda(), idx.cuda(), 0.5, 0.5).cpu()
    ref = torch.stack([OD.to_tensor_normalize(u8[i].numpy(), 0.5, 0.5) for i in idx.tolist()])
    assert out.shape == (B, C, Hs, Ws) and torch.equal(out, ref)
    out2 = H.u8_gather_normalize(u8.cuda(), idx.cuda(), 0.485, 0.229).cpu()
    ref2 = torch.stack([OD.to_tensor_normalize(u8[i].numpy(), 0.485, 0.229) for i in idx.tolist()])
    assert torch.equal(out2, ref2)


def test_gather_normalize_all_byte_values(H):
    u8 = torch.arange(256, dtype=torch.uint8).repeat(3).reshape(1, 3, 256, 1).permute(0, 2, 3, 1).contiguous()  # [1,256,1,3]
    out = H.u8_gather_normalize(u8.cuda(), torch.zeros(1, dtype=torch.int64).cuda()).cpu()
    ref = OD.to_tensor_normalize(u8[0].numpy()).unsqueeze(0)
    assert torch.equal(out, ref) and float(out.min()) == -1.0 and float(out.max()) == 1.0


def test_gather_rejects_bad_inputs(H):
    u8 = torch.zeros(2, 4, 4, 3, dtype=torch.uint8)
    with pytest.raises(RuntimeError):
        H.u8_gather_normalize(u8, torch.zeros(1, dtype=torch.int64).cuda())
    with pytest.raises(RuntimeError):
        H.u8_gather_normalize(u8.cuda(), torch.zeros(1, dtype=torch.int32).cuda())


@pytest.mark.parametrize("shape,nrow,padding,normalize", [
    ((64, 3, 64, 64), 8, 2, True), ((5, 3, 7, 9), 2, 1, True), ((1, 3, 64, 64), 8, 2, True),
    ((10, 1, 6, 6), 4, 2, True), ((6, 3, 8, 8), 8, 0, False), ((3, 64, 64), 8, 2, True)])
def test_image_grid_bit_exact(H, shape, nrow, padding, normalize):
    g = torch.Generator().manual_seed(4)
    x = torch.tanh(torch.randn(*shape, generator=g) * 2)
    if not normalize:
        x = x * 0.7 + 0.4            # some values outside [0, 1]: clamped by the quantisation
    grid = H.image_grid_u8(x.cuda(), nrow=nrow, padding=padding, normalize=normalize).cpu().numpy()
    ref = OD.grid_to_u8(OD.make_grid(x, nrow=nrow, padding=padding, normalize=normalize))
    assert grid.shape == ref.shape and grid.dtype == np.uint8
    assert np.array_equal(grid, ref), f"{int((grid != ref).sum())} bytes differ"


def test_minmax(H):
    g = torch.Generator().manual_seed(5)
    for n in (1, 63, 4097, 128 * 3 * 64 * 64):
        x = torch.randn(n, generator=g)
        mm = H.minmax(x.cuda()).cpu()
        assert float(mm[0]) == float(x.min()) and float(mm[1]) == float(x.max())


def test_save_image_and_writers(tmp_path):
    from PIL import Image
    from disentangle_mlp_amd import image_io
    g = torch.Generator().manual_seed(6)
    x = torch.tanh(torch.randn(10, 3, 16, 16, generator=g))
    f = str(tmp_path / "grid.png")
    image_io.save_image(x.cuda(), f, nrow=4, normalize=True)
    ref = OD.grid_to_u8(OD.make_grid(x, nrow=4, normalize=True))
    assert np.array_equal(np.asarray(Image.open(f)), ref)
    dl = [(x, torch.zeros(10))]
    image_io.gen_reconstructions(lambda t: t * 0.5, dl, 3, str(tmp_path), nrow=5, path_for_originals=str(tmp_path))
    image_io.gen_fid_reconstructions(lambda t: t[:2], dl, 3, str(tmp_path))
    image_io.generate_samples(lambda z: torch.tanh(z[:, :3, None, None].expand(-1, 3, 8, 8)), 4, 6, 128, str(tmp_path))
    image_io.generate_fid_samples(lambda z: torch.tanh(z[:, :3, None, None].expand(-1, 3, 8, 8)), 4, 2, 128, str(tmp_path))
    for name in ("recon_3.pdf", "original_3.pdf", "recon_0_3.pdf", "recon_1_3.pdf", "sample_4.pdf", "sample_0_4.pdf",
                 "sample_1_4.pdf"):
        assert (tmp_path / name).stat().st_size > 0, name


def test_get_data_loader_end_to_end(tmp_path):
    """ImageFolder tree -> cache -> HBM -> batches: equal (bit for bit) to PIL + ToTensor + Normalize
    in torch DataLoader order."""
    from PIL import Image
    from torch.utils.data import DataLoader, TensorDataset
    from disentangle_mlp_amd import data as D
    rng = np.random.default_rng(1)
    roots = {}
    for split, n in (("train", 21), ("val", 6), ("test", 5)):
        root = tmp_path / split / "faces"
        os.makedirs(root)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, size=(48, 40, 3), dtype=np.uint8)).save(root / f"{i:03d}.png")
        roots[split] = str(tmp_path / split)
    opt = types.SimpleNamespace(dataset="celebA", img_size=32, image_root_train=roots["train"],
                                image_root_val=roots["val"], image_root_test=roots["test"], batch_size_train=8,
                                batch_size_val=4, batch_size_test=4, num_workers=1)
    torch.manual_seed(11)
    train, val, test = D.get_data_loader(opt)
    assert (len(train), len(val), len(test)) == (3, 2, 2) and len(train.dataset) == 21
    _, samples = OD.image_folder_samples(roots["train"])
    ref_imgs = torch.stack([OD.to_tensor_normalize(OD.load_resized_u8(p, 32)) for p, _ in samples])
    torch.manual_seed(11)
    ref_order = [b[0].tolist() for b in DataLoader(TensorDataset(torch.arange(21)), batch_size=8, shuffle=True)]
    torch.manual_seed(11)
    got = list(train)
    assert [tuple(d.shape) for d, _ in got] == [(8, 3, 32, 32), (8, 3, 32, 32), (5, 3, 32, 32)]
    for (d, lab), idx in zip(got, ref_order):
        assert d.is_cuda and torch.equal(d.cpu(), ref_imgs[idx]) and lab.tolist() == [0] * len(idx)
    _, vs = OD.image_folder_samples(roots["val"])
    vref = torch.stack([OD.to_tensor_normalize(OD.load_resized_u8(p, 32)) for p, _ in vs])
    vgot = torch.cat([d.cpu() for d, _ in val])
    assert torch.equal(vgot, vref)
