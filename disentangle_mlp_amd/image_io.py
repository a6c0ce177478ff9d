"""Sample / reconstruction writers: /root/reference/utils/utils.py:6-36 with the same names and
arguments.  The reference moves every batch to the CPU and lets torchvision.utils.save_image
(pinned 0.2.1) min-max normalise, tile and quantise it; here that arithmetic runs on the device
(``vg_minmax`` + ``vg_image_grid_u8``, bit-identical to the CPU result) and only the finished
uint8 grid crosses PCIe for PIL to encode.  ``fn`` is any callable on device tensors
(``netEG.decode``, ``lambda x: netEG(x)[0]``...).
"""
import torch

from . import ops


def image_grid(tensor, nrow=8, padding=2, normalize=False, pad_value=0):
    """uint8 HWC grid (device tensor) of a (B,C,H,W) or (C,H,W) fp32 device tensor."""
    return ops.image_grid_u8(tensor.detach().float().contiguous(), nrow, padding, normalize, pad_value)


def save_image(tensor, filename, nrow=8, padding=2, normalize=False, pad_value=0):
    """torchvision.utils.save_image for device tensors (range=None, scale_each=False)."""
    from PIL import Image
    Image.init()      # registers every writer: the PDF plugin (the reference writes .pdf) encodes through the JPEG one
    grid = image_grid(tensor, nrow, padding, normalize, pad_value)
    Image.fromarray(grid.cpu().numpy()).save(filename)


def _first_batch(dl, device):
    orig_imgs, _ = next(iter(dl))
    return orig_imgs.to(device)


def gen_fid_reconstructions(fn, dl, epoch, results_path, device="cuda"):
    """utils.py:6-12: one file per reconstructed image of the loader's first batch."""
    with torch.no_grad():
        batch = fn(_first_batch(dl, device))
        for i, x in enumerate(batch):
            save_image(x, results_path + f"/recon_{i}_{str(epoch)}.pdf", normalize=True)


def gen_reconstructions(fn, dl, epoch, results_path, nrow=8, path_for_originals="", device="cuda"):
    """utils.py:14-21: one grid of the reconstructions (and optionally of the originals)."""
    with torch.no_grad():
        orig_imgs = _first_batch(dl, device)
        save_image(fn(orig_imgs), results_path + f"/recon_{str(epoch)}.pdf", nrow=nrow, normalize=True)
        if path_for_originals:
            save_image(orig_imgs, path_for_originals + f"/original_{str(epoch)}.pdf", nrow=nrow, normalize=True)


def generate_fid_samples(fn, epoch, n_samples, n_hidden, results_path, device="cuda"):
    """utils.py:23-29: n_samples decoded N(0,1) codes, one file each."""
    with torch.no_grad():
        sample = fn(torch.randn(n_samples, n_hidden).to(device))     # CPU draw, like the reference
        for i, x in enumerate(sample):
            save_image(x, results_path + f"/sample_{i}_{str(epoch)}.pdf", normalize=True)


def generate_samples(fn, epoch, n_samples, n_hidden, results_path, nrow=8, device="cuda"):
    """utils.py:31-36: one grid of decoded samples."""
    with torch.no_grad():
        sample = fn(torch.randn(n_samples, n_hidden).to(device))
        save_image(sample, results_path + f"/sample_{str(epoch)}.pdf", nrow=nrow, normalize=True)
