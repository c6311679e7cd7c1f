"""In-memory shims that let the upstream reference be imported in THIS container.

Used ONLY by tests/golden/make_golden.py (the fixture generator).  Nothing in the
product, the -m gpu tests, smoke() or bench.py imports this file: /root/reference
does not exist on the GPU box.  No reference file is modified or copied; we only
patch a few third-party symbols that changed between transformers 4.47.1 (the
reference's pin, environment.yaml:89) and the 5.x installed here.  Recipe follows
SURVEY.md Appendix A.
"""
import sys
import types

import torch

REF = "/root/reference"


def install():
    if REF not in sys.path:
        sys.path[:0] = [REF, REF + "/lumina_mgpt"]

    # absl is not installed (jacobi_iteration_lumina_mgpt.py:33)
    if "absl" not in sys.modules:
        absl = types.ModuleType("absl")
        absl_logging = types.ModuleType("absl.logging")
        absl_logging.info = lambda *a, **k: None
        absl.logging = absl_logging
        sys.modules["absl"] = absl
        sys.modules["absl.logging"] = absl_logging

    import transformers
    from transformers.generation import logits_process as lp
    from transformers.cache_utils import Cache
    from transformers.generation.utils import GenerationMixin

    # LogitsWarper was removed in transformers 5 (logit_processor_3dim.py:19)
    if not hasattr(lp, "LogitsWarper"):
        lp.LogitsWarper = lp.LogitsProcessor

    class LegacyCache(Cache):
        """4.47-style DynamicCache surface: key_cache/value_cache lists + cat update."""

        def __init__(self):
            self.key_cache = []
            self.value_cache = []

        def __len__(self):
            return len(self.key_cache)

        def get_seq_length(self, layer_idx=0):
            if len(self.key_cache) <= layer_idx:
                return 0
            return self.key_cache[layer_idx].shape[-2]

        def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
            if len(self.key_cache) <= layer_idx:
                self.key_cache.append(key_states)
                self.value_cache.append(value_states)
            else:
                self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], key_states], dim=-2)
                self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], value_states], dim=-2)
            return self.key_cache[layer_idx], self.value_cache[layer_idx]

    transformers.DynamicCache = LegacyCache

    # torch.cuda.Event / synchronize without a GPU (jacobi_iteration_lumina_mgpt.py:1052-1055)
    class _Evt:
        def __init__(self, *a, **k):
            pass

        def record(self):
            pass

        def elapsed_time(self, other):
            return 0.0

    torch.cuda.Event = _Evt
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.manual_seed_all = lambda *a, **k: None

    class CompatMixin:
        def _extract_past_from_model_output(self, outputs):
            return "past_key_values", outputs.past_key_values

        def _has_unfinished_sequences(self, this_peer_finished, synced_gpus, device=None, cur_len=None, max_length=None):
            return not this_peer_finished

    return LegacyCache, CompatMixin
