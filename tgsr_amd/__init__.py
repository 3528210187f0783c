"""tgsr_amd - MI355X-native (gfx950) implementation of the TGSR text-conditioned SR hot path.

The package mirrors the reference's module names (`model`, `models16`, `util`, `GlobalAttention`, `miscc.config`,
`miscc.losses`).  `install_dropin()` registers them under those top-level names so unmodified callers
(`from model import G_SR_NET_low_stage1, RNN_ENCODER, Variable, torch, cfg`, trainer_objective.py:8;
`from miscc.losses import sent_loss, words_loss`, pretrain_DAMSM.py:5) import this implementation.
See INTEGRATION.md.
"""
import importlib
import importlib.util
import sys

__version__ = "0.2.0"

_DROPIN = ("GlobalAttention", "util", "model", "models16", "miscc.config", "miscc.losses")


def install_dropin():
    """Expose tgsr_amd.{GlobalAttention,util,model,models16} and tgsr_amd.miscc.{config,losses} as the top-level
    modules the reference's callers import.

    The `miscc` PACKAGE is not shadowed wholesale: it also holds `miscc/utils.py` (visualisation, `mkdir_p`), which the
    callers import (trainer_objective.py:7, pretrain_DAMSM.py:3-4) and which is theirs to keep.  If a `miscc` package
    is importable from sys.path (the caller's checkout) it stays the package and only its `config` / `losses`
    sub-modules are replaced; otherwise tgsr_amd.miscc (with the non-visual helpers of utils.py) serves as `miscc`.
    Returns the list of registered module names."""
    ours = {name: importlib.import_module("tgsr_amd." + name) for name in _DROPIN}
    pkg = sys.modules.get("miscc")
    if pkg is None or getattr(pkg, "__name__", "") == "tgsr_amd.miscc":
        spec = importlib.util.find_spec("miscc")
        if spec is not None and spec.submodule_search_locations is not None:
            pkg = importlib.util.module_from_spec(spec)
            sys.modules["miscc"] = pkg
            spec.loader.exec_module(pkg)
        else:
            pkg = importlib.import_module("tgsr_amd.miscc")
            importlib.import_module("tgsr_amd.miscc.utils")
            sys.modules["miscc"] = pkg
            sys.modules["miscc.utils"] = sys.modules["tgsr_amd.miscc.utils"]
    for name, mod in ours.items():
        sys.modules[name] = mod
        if name.startswith("miscc."):
            setattr(pkg, name.split(".", 1)[1], mod)
    return sorted(list(ours) + ["miscc"])
