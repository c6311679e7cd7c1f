"""Mirror of reference llamagen/llamagen_solver.py: LlamaGenSolver, renew_llamagen, MaxlenCriteria and the baseline AR
sampler pieces (prefill / sample / top_k_top_p_filtering, which also produce the FIRST image token of the SJD path
from the global RNG, LS:75-104)."""
import torch
from torch.nn import functional as F

from .scheduler.logit_processor_3dim import TopKLogitsWarper, TopPLogitsWarper3d


def top_k_top_p_filtering(logits, top_k: int = 0, top_p: float = 1.0, filter_value: float = -float("Inf"), min_tokens_to_keep: int = 1):
    """Behaviour of reference LS:34-72 (used once per image, for the first token): keep the k largest logits, then the
    smallest descending-sorted prefix whose probability mass exceeds top_p (the token that crosses the threshold is kept)."""
    V = logits.size(-1)
    if top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), V)
        kth = logits.topk(k, dim=-1).values[..., -1:]
        logits.masked_fill_(logits < kth, filter_value)
    if top_p < 1.0:
        order = logits.argsort(dim=-1, descending=True)
        mass = logits.gather(-1, order).softmax(dim=-1).cumsum(dim=-1)
        drop_sorted = torch.zeros_like(mass, dtype=torch.bool)
        drop_sorted[..., 1:] = mass[..., :-1] > top_p          # shifted by one: the crossing token survives
        if min_tokens_to_keep > 1:
            drop_sorted[..., :min_tokens_to_keep] = False
        drop = torch.zeros_like(drop_sorted).scatter(-1, order, drop_sorted)
        logits.masked_fill_(drop, filter_value)
    return logits


def sample(logits, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, sample_logits=True):
    """Behaviour of reference LS:75-84: last row / temperature -> filtering -> softmax -> one draw from the GLOBAL generator."""
    last = logits[:, -1, :] / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        last = top_k_top_p_filtering(last, top_k=top_k, top_p=top_p)
    probs = last.softmax(dim=-1)
    idx = torch.multinomial(probs, num_samples=1) if sample_logits else probs.argmax(dim=-1, keepdim=True)
    return idx, probs


class MaxlenCriteria:
    """reference LS:341-347"""

    def __init__(self, max_seq_length):
        self.max_seq_length = max_seq_length

    def __call__(self, input_ids, scores, **kwargs):
        return input_ids.shape[-1] >= self.max_seq_length


def renew_llamagen(model_class):
    class WrappedLLamaGen(model_class):
        """reference LS:196-339.  The static KV cache already lives in the backbone; assign_kvcache/assign_past_key_values
        (the DynamicCache bridge) have no counterpart because rollback is a length update."""

        def _init_new_params(self, *args, **kwargs):
            self.is_encoder_decoder = False

        def clear_kvcache(self):
            if self.cache is not None:
                self.cache.k.zero_()
                self.cache.v.zero_()

    return WrappedLLamaGen


class LlamaGenSolver:
    """reference LS:349-470"""

    def __init__(self, model, image_top_k, image_top_p):
        self.model = model
        self.image_top_k = image_top_k
        self.image_top_p = image_top_p

    def create_logits_processor(self):
        from transformers.generation.logits_process import LogitsProcessorList
        return LogitsProcessorList([TopKLogitsWarper(top_k=self.image_top_k), TopPLogitsWarper3d(top_p=self.image_top_p)])

    @torch.no_grad()
    def prefill(self, cond_combined, cfg_scale, **sampling_kwargs):
        """reference LS:95-104 + 396-419: conditioning rows -> cache rows [0,T); first image token from the GLOBAL RNG."""
        model = self.model
        emb = model.embed_condition(cond_combined)
        Bc, T = emb.shape[0], emb.shape[1]
        pos = torch.arange(T, device=emb.device)[None].repeat(Bc, 1)
        ks = getattr(model, "_sjd_key_start", None)
        ks = torch.zeros(Bc, dtype=torch.int32, device=emb.device) if ks is None else torch.as_tensor(ks, dtype=torch.int32, device=emb.device)
        if hasattr(model.attn, "params"):
            model.attn.params = None
        logits = model.forward_embeds(emb, pos, 0, ks)
        if cfg_scale > 1.0:
            cond_logits, uncond_logits = torch.split(logits, len(logits) // 2, dim=0)
            logits = uncond_logits + (cond_logits - uncond_logits) * cfg_scale
        return sample(logits, **sampling_kwargs)[0], T

    @torch.no_grad()
    def generate(self, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, **sampling_kwargs):
        model = self.model
        if model.model_type == 'c2i':
            cond_combined = torch.cat([cond, torch.ones_like(cond) * model.num_classes]) if cfg_scale > 1.0 else cond
            T = 1
        elif model.model_type == 't2i':
            cond_combined = torch.cat([cond, torch.zeros_like(cond) + model.cls_embedding.uncond_embedding]) if cfg_scale > 1.0 else cond
            T = cond.shape[1]
        else:
            raise Exception("please check model type")
        Bc = cond_combined.shape[0]
        s_max = ((T + max_new_tokens + model.max_num_new_tokens + 32 + 31) // 32) * 32
        model.setup_cache(batch=Bc, s_max=s_max)
        for e in getattr(model, "_sjd_engines", {}).values():
            e.reset_graphs()
        if emb_masks is not None:
            # left-padded caption masks: the masked conditioning rows are a hidden key prefix (LS:403-412)
            masks = torch.cat([emb_masks, emb_masks]) if cfg_scale > 1.0 else emb_masks
            model._sjd_key_start = (masks.long().cumsum(-1) == 0).sum(-1).to(torch.int32)
        else:
            model._sjd_key_start = None
        next_token, T = self.prefill(cond_combined, cfg_scale, **sampling_kwargs)
        from transformers import GenerationConfig
        generation_config = GenerationConfig(max_new_tokens=T + max_new_tokens, max_length=T + max_new_tokens, temperature=1.0,
                                             top_k=None, do_sample=True, return_dict_in_generate=False)
        outputs = model._sample(input_ids=next_token, logits_processor=self.create_logits_processor(),
                                stopping_criteria=[MaxlenCriteria(max_new_tokens)], generation_config=generation_config,
                                synced_gpus=False, streamer=None, logits_warper=None, use_cache=True,
                                attention_mask=torch.ones((1, T + 1), device=cond.device), past_key_values=None,
                                cache_position=T + 1)
        generated = outputs[:, -max_new_tokens:]
        model.clear_kvcache()
        return generated
