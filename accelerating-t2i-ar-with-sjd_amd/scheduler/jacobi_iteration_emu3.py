"""Mirror of reference scheduler/jacobi_iteration_emu3.py (Emu3 adapter): `renew_solver`, the [B,L,V] grammar
`EOLLogitProcessor3d` (as a kernel-rule descriptor) and `prepare_batch_cfg_model_inputs`."""
import torch

from .jacobi_iteration_lumina_mgpt import renew_sampler, renew_backbone  # noqa: F401
from .logit_processor_3dim import _Descriptor, get_double_cfg_input_ids


class Emu3PrefixConstrainedLogitsHelper:
    """reference emu3/mllm/utils_emu3.py:19-45 (constructor only; the per-token callback is replaced by kernel rules)."""

    def __init__(self, height, width, img_token, eoi_token, eos_token, eol_token, eof_token, pad_token, visual_tokens):
        self.height, self.width = height, width
        self.img_token, self.eoi_token, self.eos_token = img_token, eoi_token, eos_token
        self.eol_token, self.eof_token, self.pad_token = eol_token, eof_token, pad_token
        self.visual_tokens = visual_tokens
        self.offset_cache = {}


def renew_end_of_line_logit_processor_3d(model_class):
    """reference JE:41-151"""
    class EOLLogitProcessor3d(model_class, _Descriptor):
        __call__ = _Descriptor.__call__
        filter_value = "finfo.min"          # (direct calls: removed entries hold torch.finfo(dtype).min, JE:80)

    return EOLLogitProcessor3d


def renew_sampler_forward(model_class):
    class JacobiModel(model_class):
        """reference JE:153-368"""

        def _init_new_params(self, *args, use_chameleon_tokenizer=False, _init_doubled_attn_mask_cfg=True, visual_tokens=None,
                             **kwargs):
            super()._init_new_params(*args, use_chameleon_tokenizer=use_chameleon_tokenizer,
                                     _init_doubled_attn_mask_cfg=_init_doubled_attn_mask_cfg, **kwargs)
            # the reference keeps img_vocab = Chameleon image ids here as well
            self._init_doubled_attn_mask_cfg = _init_doubled_attn_mask_cfg

        def renew_attn_mask(self, batchsize, prefill_num, not_pad_mask=None, device='cuda'):
            """reference JE:177-186: 0/1 mask [B_cfg, prefill_num], zero on pad columns."""
            rows = batchsize * (2 if self.do_cfg else 1)
            return not_pad_mask.to(device=device, dtype=torch.float32).reshape(rows, prefill_num).clone()

        def prepare_batch_cfg_model_inputs(self, input_ids, neg_input_ids=None, attention_mask=None):
            """reference JE:234-278: stack the positive and negative prompts (left-padded to a common length) on the batch
            axis and derive the pad mask.  Returns {input_ids, pos_input_ids (when a negative prompt is given), attention_mask}."""
            pad = self.config.pad_token_id if hasattr(self, "config") else self.pad_token_id
            B = input_ids.shape[0]
            out = {"input_ids": input_ids, "attention_mask": attention_mask}
            if neg_input_ids is not None:
                stacked = get_double_cfg_input_ids(input_ids, neg_input_ids, pad_category=pad)
                out["input_ids"], out["pos_input_ids"] = stacked, stacked[:B]
                keep = stacked != pad
            else:
                rows = B * (2 if self.do_cfg else 1)
                keep = torch.zeros((rows, input_ids.shape[1]), dtype=torch.bool, device=input_ids.device)
                keep[:B] = input_ids != pad
            if attention_mask is None:
                out["attention_mask"] = self.renew_attn_mask(B, keep.shape[1], keep, input_ids.device)
            elif attention_mask.shape[0] == B:
                raise NotImplementedError
            return out

    return JacobiModel


def renew_solver(model, processor, **jacobi_param_dict):
    """reference JE:370-412 -> (model, LogitsProcessorList)"""
    h = jacobi_param_dict.pop('h', None)
    w = jacobi_param_dict.pop('w', None)
    jacobi_param_dict.pop('neg_inputs', None)
    jacobi_param_dict.pop('classifier_free_guidance', None)
    constrained_fn = processor.build_prefix_constrained_fn(h, w)
    constrained_fn.__class__ = renew_end_of_line_logit_processor_3d(constrained_fn.__class__)
    model.__class__ = renew_sampler(model.__class__)
    model._init_new_params(**jacobi_param_dict)
    model.__class__ = renew_sampler_forward(model.__class__)
    model._init_new_params(visual_tokens=constrained_fn.visual_tokens, **jacobi_param_dict)
    from transformers.generation.logits_process import LogitsProcessorList
    return model, LogitsProcessorList([constrained_fn])
