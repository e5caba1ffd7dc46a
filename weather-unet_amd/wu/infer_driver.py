"""Checkpoint interchange and the inference sweep of the reference's drivers.

* Checkpoints: the reference saves ``{'inference': G.state_dict(), 'discriminator': D.state_dict(), 'epoch',
  'global_step'}`` as ``{save_dir}/{name}/{name}_e{epoch:04d}_s{step}.pt`` (t_est_train.py:365-373) and resumes
  from the lexicographically last file (t_cls_train.py:158-166).  ``save_checkpoint`` / ``load_checkpoint`` /
  ``latest_checkpoint`` reproduce that format with the same keys and OIHW fp32 tensors, so files interchange with
  the reference in both directions.  Loading uses ``weights_only=True`` (tensors and plain containers only).
* Inference: ``inference/inf_transfer_c.py:108-121`` runs, per batch, one forward per class with a tiled one-hot
  row and saves each output with ``save_image(..., normalize=True)`` (per-image min-max).  ``class_sweep`` is that
  loop on GPU tensors; ``signal_sweep`` the same loop over arbitrary conditioning rows (``inf_transfer_e.py:136-143``),
  ``transfer_rows`` the one-row-per-image call of ``inf_1year_signals.py:98-107``, ``axis_sweep`` the conditioning
  schedule of ``demo.py:67-82``; ``normalize_minmax`` / ``to_uint8`` are save_image's arithmetic, done on the GPU.
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
    """torchvision.utils.save_image(tensor, normalize=True) as the inference scripts call it -- on ONE (3,H,W) image at a time
    (inf_transfer_c.py:118-120), i.e. per-image min-max -- with the arithmetic of the pinned torchvision (<0.4, Pipfile:11;
    utils.make_grid.norm_ip): clamp to [min, max], then (x - min) / (max - min + 1e-5).  On the GPU, whole batch at once."""
    flat = images.reshape(images.shape[0], -1)
    lo = flat.min(dim=1).values.view(-1, 1, 1, 1)
    hi = flat.max(dim=1).values.view(-1, 1, 1, 1)
    return ((images - lo) / (hi - lo + eps)).clamp_(0, 1)


def to_uint8(images01):
    """The byte image save_image writes (torchvision <0.4: ``grid.mul(255).clamp(0, 255).byte()`` -- truncation, no +0.5),
    NHWC uint8 on the GPU, ready for a host-side encoder."""
    return images01.mul(255).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


def _run(transfer, batch, labels, graphed):
    return graphed(batch, labels, copy_out=True) if graphed is not None else transfer(batch, labels)


@torch.no_grad()
def signal_sweep(transfer, batch, rows, normalize=False, graphed=None):
    """inference/inf_transfer_e.py:136-143 (and t_cls_train.py:336-337): for every conditioning row r of ``rows`` (R, nc) --
    soft labels, standardised weather signals, scaled one-hot rows --, ``transfer(batch, r tiled B times)``.
    Returns (R, B, 3, H, W).  ``graphed``: a ``GraphedUNet`` captured for this batch shape (one hipGraph replay per row).
    The reference never calls ``.eval()`` in these scripts, so its Dropout(0.3) is active; whether this sweep uses dropout
    follows ``transfer.training`` exactly as there."""
    bs = batch.shape[0]
    rows = rows.to(device=batch.device, dtype=torch.float32)
    outs = []
    for i in range(rows.shape[0]):
        labels = rows[i].unsqueeze(0).expand(bs, rows.shape[1]).contiguous()        # torch.cat([row] * bs).view(-1, nc)
        out = _run(transfer, batch, labels, graphed)
        outs.append(normalize_minmax(out) if normalize else out)
    return torch.stack(outs)


@torch.no_grad()
def class_sweep(transfer, batch, num_classes=None, normalize=False, graphed=None):
    """inf_transfer_c.py:114-121: for every class i, ``transfer(batch, onehot[i] tiled)`` -- ``signal_sweep`` over the rows
    of the identity.  (The script's loop runs ``for i in range(bs)`` over ``onehot[i]``, i.e. it assumes batch_size ==
    num_classes; this sweep always covers all classes.)  Returns (num_classes, B, 3, H, W)."""
    nc = num_classes if num_classes is not None else transfer.adain1.num_classes
    return signal_sweep(transfer, batch, torch.eye(nc, device=batch.device), normalize, graphed)


@torch.no_grad()
def transfer_rows(transfer, batch, signals, normalize=False, graphed=None):
    """inference/inf_1year_signals.py:98-107: one conditioning row PER IMAGE (``transfer(batch, sig)``)."""
    out = _run(transfer, batch, signals.to(device=batch.device, dtype=torch.float32).contiguous(), graphed)
    return normalize_minmax(out) if normalize else out


@torch.no_grad()
def axis_sweep(transfer, batch, pred, thetas, alpha=1.0, graphed=None):
    """demo.py:67-82: for every angle theta and every class axis a, condition on the estimator's prediction ``pred`` (B, nc)
    with component a replaced by ``alpha * sin(theta)``:  c = onehot[a] * sin(theta) * alpha + (1 - onehot[a]) * pred.
    Returns (T, nc, B, 3, H, W) raw outputs (the script then maps (x + 1) * 127.5 and normalises per image for the GIF)."""
    nc = pred.shape[1]
    eye = torch.eye(nc, device=batch.device)
    pred = pred.to(device=batch.device, dtype=torch.float32)
    frames = []
    for theta in thetas:
        s = torch.sin(torch.as_tensor(float(theta), dtype=torch.float32, device=batch.device)) * alpha
        per_axis = []
        for a in range(nc):
            c = eye[a].unsqueeze(0) * s + (1.0 - eye[a]).unsqueeze(0) * pred               # :76-79
            per_axis.append(_run(transfer, batch, c.contiguous(), graphed))
        frames.append(torch.stack(per_axis))
    return torch.stack(frames)
