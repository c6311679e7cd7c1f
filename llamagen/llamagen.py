"""reference import path llamagen.llamagen: the GPT_models registry (reference llamagen/llamagen.py:475-504), backed by
sjd_amd.backbones.LlamaGenBackbone (same state-dict keys)."""
from sjd_amd.backbones import LlamaGenArgs, LlamaGenBackbone


def _mk(n_layer, n_head, dim):
    def ctor(**kwargs):
        kw = {k: v for k, v in kwargs.items() if k in LlamaGenArgs.__dataclass_fields__}
        return LlamaGenBackbone(LlamaGenArgs(n_layer=n_layer, n_head=n_head, dim=dim, **kw))
    return ctor


GPT_models = {'GPT-B': _mk(12, 12, 768), 'GPT-L': _mk(24, 16, 1024), 'GPT-XL': _mk(36, 20, 1280), 'GPT-XXL': _mk(48, 24, 1536),
              'GPT-XXXL': _mk(48, 40, 2560), 'GPT-1B': _mk(22, 32, 2048), 'GPT-3B': _mk(24, 32, 3200), 'GPT-7B': _mk(32, 32, 4096)}
Transformer, ModelArgs = LlamaGenBackbone, LlamaGenArgs
