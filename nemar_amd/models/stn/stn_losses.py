"""Deformation-field regulariser — mirror of reference models/stn/stn_losses.py:4-30 on the fused gfx950 stencil
kernel (nemar_amd/csrc/smooth.hip): four directional |delta| means with optional bilateral weights
exp(-alpha*|delta I|) (mean over image channels; the image carries no gradient)."""
from ... import ops


def smoothness_loss(deformation, img=None, alpha=0.0, factor=1.0):
    """`factor` is this build's extension: the multi-resolution weight of UnetSTN folded into the kernel."""
    use_img = img if (img is not None and alpha > 0.0) else None
    return ops.smoothness(deformation, use_img, alpha, factor)
