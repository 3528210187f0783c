"""tgsr_amd - MI355X-native (gfx950) implementation of the TGSR text-conditioned SR hot path.

The package mirrors the reference's module names (`model`, `util`, `GlobalAttention`, `miscc.config`,
`miscc.losses`).  `install_dropin()` registers them under those top-level names so unmodified callers
(`from model import RNN_ENCODER, G_SR_NET_low, NetG_highweight`, trainer_objective.py:8,75-88) import this
implementation.  See INTEGRATION.md.
"""
import sys

__version__ = "0.1.0"

_DROPIN = ("GlobalAttention", "util", "model", "models16", "miscc", "miscc.config", "miscc.losses")


def install_dropin():
    """Expose tgsr_amd.{model,util,GlobalAttention,miscc} as the top-level modules the reference's callers import."""
    import importlib
    for name in _DROPIN:
        sys.modules[name] = importlib.import_module("tgsr_amd." + name)
    return [sys.modules[n] for n in _DROPIN]
