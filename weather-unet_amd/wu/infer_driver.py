"""Checkpoint interchange and the inference sweep of the reference's drivers.

* Checkpoints: the reference saves ``{'inference': G.state_dict(), 'discriminator': D.state_dict(), 'epoch',
  'global_step'}`` as ``{save_dir}/{name}/{name}_e{epoch:04d}_s{step}.pt`` (t_est_train.py:365-373) and resumes
  from the lexicographically last file (t_cls_train.py:158-166).  ``save_checkpoint`` / ``load_checkpoint`` /
  ``latest_checkpoint`` reproduce that format with the same keys and OIHW fp32 tensors, so files interchange with
  the reference in both directions.  Loading uses ``weights_only=True`` (tensors and plain containers only).
* Inference: ``inference/inf_transfer_c.py:108-121`` runs, per batch, one forward per class with a tiled one-hot
  row and saves each output with ``save_image(..., normalize=True)`` (per-image min-max).  ``class_sweep`` is that
  loop on GPU tensors; ``normalize_minmax`` is the min-max normalisation, done on the GPU.
"""
import glob
import os

import torch


def save_checkpoint(save_dir, name, inference, discriminator, epoch, global_step):
    os.makedirs(os.path.join(save_dir, name), exist_ok=True)
    path = os.path.join(save_dir, name, f"{name}_e{epoch:04d}_s{global_step}.pt")
    state = {"inference": {k: v.detach().cpu() for k, v in inference.state_dict().items()},
             "discriminator": {k: v.detach().cpu() for k, v in discriminator.state_dict().items()},
             "epoch": int(epoch), "global_step": int(global_step)}
    torch.save(state, path)
    return path


def latest_checkpoint(save_dir, name):
    """t_cls_train.py:158-160: sorted(glob(dir/*))[-1], or None."""
    found = sorted(glob.glob(os.path.join(save_dir, name, "*")))
    return found[-1] if found else None


def load_checkpoint(path, inference=None, discriminator=None, map_location="cpu"):
    """Load a reference-format checkpoint; returns (epoch, global_step).  Either module may be None."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    if inference is not None:
        inference.load_state_dict(sd["inference"])
    if discriminator is not None:
        discriminator.load_state_dict(sd["discriminator"])
    return int(sd.get("epoch", 0)), int(sd.get("global_step", 0))


def normalize_minmax(images, eps=1e-5):
    """torchvision.utils.save_image(normalize=True) per image: (x - min) / (max - min + eps), on the GPU."""
    flat = images.reshape(images.shape[0], -1)
    lo = flat.min(dim=1).values.view(-1, 1, 1, 1)
    hi = flat.max(dim=1).values.view(-1, 1, 1, 1)
    return ((images - lo) / (hi - lo + eps)).clamp_(0, 1)


@torch.no_grad()
def class_sweep(transfer, batch, num_classes=None, normalize=False, graphed=None):
    """inf_transfer_c.py:114-121: for every class i, ``transfer(batch, onehot[i] tiled)``.

    Returns a tensor (num_classes, B, 3, H, W).  ``graphed`` may be a ``GraphedUNet`` captured for this batch shape
    (one hipGraph replay per class instead of ~45 launches).  Note the reference never calls ``.eval()`` here, so its
    Dropout(0.3) is active; whether this sweep uses dropout follows ``transfer.training`` exactly as there.
    """
    nc = num_classes if num_classes is not None else transfer.adain1.num_classes
    bs = batch.shape[0]
    onehot = torch.eye(nc, device=batch.device)
    outs = []
    for i in range(nc):
        labels = onehot[i].unsqueeze(0).expand(bs, nc).contiguous()
        out = graphed(batch, labels, copy_out=True) if graphed is not None else transfer(batch, labels)
        outs.append(normalize_minmax(out) if normalize else out)
    return torch.stack(outs)
