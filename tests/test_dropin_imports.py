"""CPU: after tgsr_amd.install_dropin() the import statements of the reference's callers resolve.

The statements are restated here as strings (trainer_objective.py:7-8, 75-88; pretrain_DAMSM.py:3-6, 11; test1.py:4;
miscc/losses.py:5-7; util.py:11-12) and executed in a fresh interpreter, so that neither this process's sys.modules nor
the reference checkout is involved.  `datasets` (the callers' own data module) is not part of the drop-in."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

CALLER_IMPORTS = [
    "from miscc.utils import mkdir_p, build_super_imagesall",                          # trainer_objective.py:7
    "from model import G_SR_NET_low_stage1, RNN_ENCODER, Variable, torch, cfg",        # trainer_objective.py:8
    "from model import G_SR_NET_low",                                                  # trainer_objective.py:75
    "from models16 import G_SR_NET_low",                                               # :81
    "from model import NetG_highweight",                                               # :85
    "from models16 import NetG_highweight",                                            # :87
    "from miscc.utils import mkdir_p",                                                 # pretrain_DAMSM.py:3
    "from miscc.utils import build_super_images",                                      # :4
    "from miscc.losses import sent_loss, words_loss",                                  # :5
    "from miscc.config import cfg, cfg_from_file",                                     # :6, test1.py:4
    "from model import RNN_ENCODER, CNN_ENCODER",                                      # :11
    "from GlobalAttention import func_attention",                                      # miscc/losses.py:7
    "from GlobalAttention import GlobalAttentionGeneral as ATT_NET",                   # util.py:12
    "from util import *",                                                              # model.py:3
    "from model import *",                                                             # models16.py:2
]


def _run(code, extra_path=None, cwd=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([p for p in (extra_path, ROOT) if p])
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, cwd=cwd or "/", capture_output=True,
                          text=True, timeout=300)


def test_reference_import_lines_resolve():
    code = """
        import tgsr_amd
        names = tgsr_amd.install_dropin()
        for stmt in %r:
            exec(stmt, {})
        import model, miscc.config, miscc.losses, miscc.utils
        assert model.__name__ == "tgsr_amd.model" and miscc.config.__name__ == "tgsr_amd.miscc.config"
        assert miscc.config.cfg is model.cfg
        try:
            model.G_SR_NET_low_stage1()
        except NotImplementedError:
            pass
        else:
            raise SystemExit("G_SR_NET_low_stage1 must refuse construction")
        try:
            miscc.utils.build_super_imagesall(None, None, None, None, None)
        except NotImplementedError:
            pass
        else:
            raise SystemExit("visualisation helpers are out of scope and must say so")
        print("ok", len(names))
    """ % (CALLER_IMPORTS,)
    r = _run(code)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_callers_own_miscc_package_keeps_its_utils(tmp_path):
    """A caller-side `miscc` package (with its own utils.py) stays the package; only config / losses are replaced."""
    pkg = tmp_path / "miscc"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "utils.py").write_text("from miscc.config import cfg\nCALLER = 'mine'\ndef mkdir_p(p):\n    return 'caller'\n")
    (pkg / "config.py").write_text("raise ImportError('the caller-side config must be shadowed')\n")
    code = """
        import tgsr_amd
        tgsr_amd.install_dropin()
        from miscc.utils import mkdir_p, CALLER, cfg
        from miscc.config import cfg as cfg2, cfg_from_file
        import miscc, miscc.losses
        assert CALLER == 'mine' and mkdir_p('x') == 'caller' and cfg is cfg2
        assert miscc.config.__name__ == 'tgsr_amd.miscc.config' and miscc.losses.__name__ == 'tgsr_amd.miscc.losses'
        print("ok")
    """
    r = _run(code, extra_path=str(tmp_path))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
