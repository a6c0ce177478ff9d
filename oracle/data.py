"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the image pipeline either side of the step.

Input side: /root/reference/dataloader/dataset.py:37-50 (celebA branch): ImageFolder ->
Resize((s, s)) -> CenterCrop(s) -> ToTensor -> Normalize(0.5, 0.5), batched by a
torch.utils.data.DataLoader.  Output side: /root/reference/utils/utils.py:6-36, which calls
torchvision.utils.save_image(..., normalize=True).

torchvision is a third-party dependency that is NOT vendored under /root/reference and not
installed here (pinned torchvision==0.2.1, requirements.txt:30).  The functions below restate
its published 0.2.1 algorithms (datasets/folder.py make_dataset + pil_loader,
transforms/functional.py resize / to_tensor / normalize, utils.py make_grid / save_image) on
PIL + torch CPU ops; parity of this part is pinned by the reference's call sites only
("parity unpinned" against torchvision itself).  The DataLoader order is checked against the
real torch.utils.data.DataLoader in tests/.
"""
import math
import os

import numpy as np
import torch

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif")   # torchvision 0.2.1 folder.py


def image_folder_samples(root):
    """datasets.ImageFolder(root).samples: classes = sorted sub-directories, files in sorted
    os.walk order, filtered by extension (case-insensitive)."""
    classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    samples = []
    for ci, cname in enumerate(classes):
        for dirpath, _, fnames in sorted(os.walk(os.path.join(root, cname))):
            for fname in sorted(fnames):
                if fname.lower().endswith(IMG_EXTENSIONS):
                    samples.append((os.path.join(dirpath, fname), ci))
    return classes, samples


def load_resized_u8(path, size):
    """pil_loader (convert RGB) -> Resize((size, size), BILINEAR) -> CenterCrop(size) (identity
    after a square resize); returns the uint8 HWC array ToTensor starts from."""
    from PIL import Image
    with open(path, "rb") as f:
        img = Image.open(f).convert("RGB")
    img = img.resize((size, size), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def to_tensor_normalize(u8_hwc, mean=0.5, std=0.5):
    """ToTensor (float().div(255), HWC -> CHW) then Normalize: t.sub_(mean).div_(std)."""
    t = torch.from_numpy(np.array(u8_hwc, dtype=np.uint8, copy=True)).permute(2, 0, 1).contiguous().float().div(255)
    return t.sub_(mean).div_(std)


def make_grid(tensor, nrow=8, padding=2, normalize=False, pad_value=0):
    """torchvision 0.2.1 utils.make_grid (range=None, scale_each=False)."""
    tensor = tensor.detach().cpu().float()
    if tensor.dim() == 3:
        if tensor.size(0) == 1:
            tensor = torch.cat((tensor, tensor, tensor), 0)
        tensor = tensor.unsqueeze(0)
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if normalize:
        tensor = tensor.clone()
        lo, hi = float(tensor.min()), float(tensor.max())
        tensor.clamp_(min=lo, max=hi)
        tensor.add_(-lo).div_(hi - lo + 1e-5)
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((3, height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding) \
                .narrow(2, x * width + padding, width - padding).copy_(tensor[k])
            k += 1
    return grid


def grid_to_u8(grid):
    """save_image's quantisation in 0.2.1: grid.mul(255).clamp(0, 255).byte(), CHW -> HWC."""
    return grid.mul(255).clamp(0, 255).byte().permute(1, 2, 0).contiguous().numpy()
