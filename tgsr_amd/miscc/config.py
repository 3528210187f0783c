"""Configuration object with the reference's attribute names (miscc/config.py:10-67) on a plain namespace.

Modules read it at construction time exactly like the reference (`cfg.GAN.GF_DIM` model.py:37,
`cfg.TEXT.EMBEDDING_DIM` :38, `cfg.GAN.R_NUM` util.py:759, ...).  `cfg_from_file` keeps the reference's
semantics: every key of the YAML must already exist and have the same type (config.py:70-100).
"""


class _Node(dict):
    """dict with attribute access; no dependency on the (absent) easydict package."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _defaults():
    c = _Node()
    c.METHOD = 'S16'
    c.DATASET_NAME = 'birds'
    c.CONFIG_NAME = ''
    c.DATA_DIR = ''
    c.DATA_DIRIM = ''
    c.GPU_ID = 0
    c.CUDA = True
    c.WORKERS = 6
    c.RNN_TYPE = 'LSTM'
    c.B_VALIDATION = False
    c.TREE = _Node(BRANCH_NUM=5, BASE_SIZE=64)
    c.TRAIN = _Node(BATCH_SIZE=64, MAX_EPOCH=600, SNAPSHOT_INTERVAL=2000, DISCRIMINATOR_LR=2e-4,
                    GENERATOR_LR=2e-4, ENCODER_LR=2e-4, RNN_GRAD_CLIP=0.25, FLAG=False, NET_E='', NET_G='',
                    B_NET_D=True, SMOOTH=_Node(GAMMA1=5.0, GAMMA3=10.0, GAMMA2=5.0, LAMBDA=1.0))
    c.GAN = _Node(DF_DIM=64, GF_DIM=128, Z_DIM=100, CONDITION_DIM=100, R_NUM=2, B_ATTENTION=True, B_DCGAN=False)
    c.TEXT = _Node(CAPTIONS_PER_IMAGE=10, EMBEDDING_DIM=256, WORDS_NUM=18)
    return c


cfg = _defaults()


def _merge_a_into_b(a, b):
    """Same contract as config.py:70-100: unknown key -> KeyError, type mismatch -> ValueError."""
    if not isinstance(a, dict):
        return
    for k, v in a.items():
        if k not in b:
            raise KeyError('{} is not a valid config key'.format(k))
        if isinstance(v, dict):
            if not isinstance(b[k], dict):
                raise ValueError('Type mismatch for config key: {}'.format(k))
            _merge_a_into_b(v, b[k])
        else:
            if type(b[k]) is not type(v):
                raise ValueError('Type mismatch ({} vs. {}) for config key: {}'.format(type(b[k]), type(v), k))
            b[k] = v


def cfg_from_file(filename):
    """Load a YAML file and merge it into `cfg` (config.py:103-109; safe_load instead of the bare yaml.load
    that PyYAML >= 6 rejects)."""
    import yaml
    with open(filename, 'r', encoding='UTF-8') as f:
        _merge_a_into_b(yaml.safe_load(f) or {}, cfg)


def cfg_reset():
    """Restore the defaults in place (module-level `cfg` identity is kept)."""
    d = _defaults()
    cfg.clear()
    cfg.update(d)
