"""Host logic of the input pipeline (no GPU): ImageFolder order, cache build vs the oracle's
PIL restatement, DataLoader-identical batch order, data-parallel sharding."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

from disentangle_mlp_amd import data as D
from oracle import data as OD


class _FakeDataset:
    device = "cpu"

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def _make_tree(root, n_per_class=(5, 3), size=(40, 30)):
    from PIL import Image
    rng = np.random.default_rng(0)
    paths = []
    for ci, n in enumerate(n_per_class):
        d = os.path.join(root, f"class{ci}", "sub" if ci else "")
        os.makedirs(d, exist_ok=True)
        for i in range(n):
            arr = rng.integers(0, 256, size=(size[1], size[0], 3), dtype=np.uint8)
            ext = (".png", ".PNG", ".bmp")[i % 3]
            p = os.path.join(d, f"img_{9 - i}{ext}")
            Image.fromarray(arr).save(p)
            paths.append(p)
    with open(os.path.join(root, "class0", "notes.txt"), "w") as f:
        f.write("not an image")
    return paths


def test_image_folder_order_and_cache(tmp_path):
    root = str(tmp_path / "train")
    _make_tree(root)
    classes, samples = D.list_image_folder(root)
    oclasses, osamples = OD.image_folder_samples(root)
    assert classes == oclasses == ["class0", "class1"]
    assert samples == osamples and len(samples) == 8
    assert [os.path.basename(p) for p, _ in samples[:5]] == sorted(os.path.basename(p) for p, _ in samples[:5])
    img_file, lab_file = D.build_image_cache(root, 16, workers=1)
    imgs, labs = np.load(img_file), np.load(lab_file)
    assert imgs.shape == (8, 16, 16, 3) and imgs.dtype == np.uint8
    assert labs.tolist() == [0] * 5 + [1] * 3
    for i, (p, _) in enumerate(samples):
        assert np.array_equal(imgs[i], OD.load_resized_u8(p, 16)), p
    # unchanged tree: the cache is reused, not rebuilt
    mtime = os.path.getmtime(img_file)
    assert D.build_image_cache(root, 16, workers=1) == (img_file, lab_file)
    assert os.path.getmtime(img_file) == mtime
    # a new file invalidates it
    from PIL import Image
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(root, "class1", "zzz.png"))
    img_file2, _ = D.build_image_cache(root, 16, workers=1)
    assert np.load(img_file2).shape[0] == 9


def test_empty_root_raises(tmp_path):
    os.makedirs(tmp_path / "empty" / "cls")
    with pytest.raises(RuntimeError):
        D.list_image_folder(str(tmp_path / "empty"))
    with pytest.raises(FileNotFoundError):
        D.list_image_folder(str(tmp_path / "missing"))


@pytest.mark.parametrize("n,bs,shuffle", [(103, 16, True), (64, 16, True), (10, 4, False), (5, 8, True)])
def test_batch_order_matches_torch_dataloader(n, bs, shuffle):
    """Same torch.manual_seed => same batches as torch.utils.data.DataLoader, epoch after epoch
    (the loader consumes the global RNG exactly like iter(DataLoader))."""
    ds = TensorDataset(torch.arange(n))
    torch.manual_seed(123)
    ref = DataLoader(ds, batch_size=bs, shuffle=shuffle)
    ref_epochs = [[b[0].tolist() for b in ref] for _ in range(3)]
    ref_next = torch.rand(1).item()
    torch.manual_seed(123)
    ld = D.DeviceLoader(_FakeDataset(n), bs, shuffle=shuffle)
    got_epochs = [[c.tolist() for c in ld.index_batches(ld.epoch_order())] for _ in range(3)]
    assert got_epochs == ref_epochs
    assert torch.rand(1).item() == ref_next          # RNG left in the same state
    assert len(ld) == len(ref)


def test_private_generator_matches_dataloader():
    ds = TensorDataset(torch.arange(50))
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    ref = [b[0].tolist() for b in DataLoader(ds, batch_size=8, shuffle=True, generator=g1)]
    ld = D.DeviceLoader(_FakeDataset(50), 8, shuffle=True, generator=g2)
    assert [c.tolist() for c in ld.index_batches(ld.epoch_order())] == ref


def test_data_parallel_sharding():
    """Ranks see the same permutation and take the DataParallel chunk of every global batch; a tail
    that cannot feed every rank is dropped on all ranks."""
    n, gb, world = 72, 16, 4
    order = torch.arange(n)
    per_rank = []
    for r in range(world):
        ld = D.DeviceLoader(_FakeDataset(n), gb, shuffle=False, rank=r, world_size=world)
        per_rank.append([c.tolist() for c in ld.index_batches(order)])
    assert {len(x) for x in per_rank} == {5}              # 4 full batches + an 8-sample tail (2 per rank)
    for step in range(4):
        glob = sum((per_rank[r][step] for r in range(world)), [])
        assert glob == list(range(step * gb, (step + 1) * gb))
        assert glob == sum((c.tolist() for c in torch.arange(step * gb, (step + 1) * gb).chunk(world)), [])
    tail = [per_rank[r][4] for r in range(world)]
    assert sum(tail, []) == list(range(64, 72)) and all(len(t) == 2 for t in tail)
    # 70 samples: chunk() would split the 6-sample tail 2/2/2/0 -> rank 3 idle -> dropped everywhere
    for r in range(world):
        ld = D.DeviceLoader(_FakeDataset(70), gb, shuffle=False, rank=r, world_size=world)
        assert len(list(ld.index_batches(torch.arange(70)))) == 4
    # 65 samples: the 1-sample tail cannot be split over 4 ranks -> dropped everywhere
    for r in range(world):
        ld = D.DeviceLoader(_FakeDataset(65), gb, shuffle=False, rank=r, world_size=world)
        assert len(list(ld.index_batches(torch.arange(65)))) == 4
    with pytest.raises(ValueError):
        D.DeviceLoader(_FakeDataset(10), 6, world_size=4)


def test_oracle_make_grid_shapes_and_values():
    """The oracle's make_grid restatement: layout of the pinned torchvision 0.2.1 algorithm."""
    x = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).reshape(5, 3, 4, 6)
    g = OD.make_grid(x, nrow=2, padding=1, normalize=False, pad_value=-1.0)
    assert tuple(g.shape) == (3, 3 * 5 + 1, 2 * 7 + 1)
    assert torch.equal(g[:, 1:5, 1:7], x[0]) and torch.equal(g[:, 6:10, 8:14], x[3])
    assert float(g[0, 0, 0]) == -1.0 and float(g[0, 12, 9]) == -1.0          # padding / empty last cell
    gn = OD.make_grid(x, nrow=8, normalize=True)
    assert float(gn.max()) <= 1.0 and float(gn[:, 2:6, 2:8].min()) == 0.0
    one = OD.make_grid(x[0], normalize=True)
    assert tuple(one.shape) == (3, 4, 6)
    u8 = OD.grid_to_u8(gn)
    assert u8.dtype == np.uint8 and u8.shape == (gn.shape[1], gn.shape[2], 3)


def test_get_data_loader_rejects_other_datasets():
    class O:
        dataset = "birds"
    with pytest.raises(NotImplementedError):
        D.get_data_loader(O())
