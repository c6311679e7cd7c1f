"""reference import path llamagen.tokenizer.tokenizer_image.vq_model: the VQ_models registry (reference vq_model.py:415-423),
backed by the decode side in sjd_amd.detokenizers (same state-dict keys; encoder-side keys of a checkpoint are ignored)."""
from sjd_amd.detokenizers import LlamaGenVQ


def VQ_8(**kwargs):
    return LlamaGenVQ(ch_mult=(1, 2, 2, 4), **kwargs)


def VQ_16(**kwargs):
    return LlamaGenVQ(ch_mult=(1, 1, 2, 2, 4), **kwargs)


VQ_models = {'VQ-16': VQ_16, 'VQ-8': VQ_8}
