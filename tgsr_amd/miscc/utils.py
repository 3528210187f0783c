"""The non-visual helpers of the reference's miscc/utils.py, so that its callers' import lines resolve when
`tgsr_amd.install_dropin()` has to supply the `miscc` package itself (no caller-side `miscc` on sys.path).

In scope (used by the training counterparts): `weights_init` (utils.py:454-464), `load_params` / `copy_G_params`
(:467-474: the generator EMA helpers), `mkdir_p` (:477-485).  Out of scope (attention-map
visualisation with PIL / skimage and a hard-coded Windows font path): `build_super_images*` import fine and raise when
called - a caller that wants the PNG panels keeps its own miscc/utils.py, which install_dropin() leaves in place.
"""
import errno
import os
from copy import deepcopy

import torch
import torch.nn as nn

from .config import cfg  # noqa: F401  (the reference module exposes it too, utils.py:15)


def weights_init(m):
    """utils.py:454-464: orthogonal conv / linear weights, N(1, 0.02) BatchNorm scale, zero biases."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
    elif classname.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find('Linear') != -1:
        nn.init.orthogonal_(m.weight.data, 1.0)
        if m.bias is not None:
            m.bias.data.fill_(0.0)


def load_params(model, new_param):
    """utils.py:467-469.  In-place under no_grad (not through `.data`): the copy bumps each parameter's version
    counter, which is what invalidates the packed-weight caches of the fused inference modules."""
    with torch.no_grad():
        for p, new_p in zip(model.parameters(), new_param):
            p.copy_(new_p)


def copy_G_params(model):
    """utils.py:472-474."""
    return deepcopy(list(p.data for p in model.parameters()))


def mkdir_p(path):
    """utils.py:477-485."""
    try:
        os.makedirs(path)
    except OSError as exc:
        if not (exc.errno == errno.EEXIST and os.path.isdir(path)):
            raise


def _visualisation(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("miscc.utils.%s draws attention-map panels with PIL/skimage (utils.py:74-451): "
                                  "visualisation is outside tgsr_amd's scope; keep the caller's own miscc/utils.py on "
                                  "sys.path (install_dropin() then leaves it in place)" % name)
    fn.__name__ = name
    return fn


build_super_images = _visualisation("build_super_images")
build_super_images2 = _visualisation("build_super_images2")
build_super_imagesall = _visualisation("build_super_imagesall")
