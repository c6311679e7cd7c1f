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


def sample(logits, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, sample_logits=True, noise_device=None):
    """Behaviour of reference LS:75-84: last row / temperature -> filtering -> softmax -> one draw from the GLOBAL generator of the
    logits' device (noise_device="cpu": draw on the CPU generator instead -- replays the reference's CPU run on a GPU backbone)."""
    dev = logits.device
    if noise_device is not None:
        logits = logits.float().to(noise_device)
    last = logits[:, -1, :] / max(temperature, 1e-5)
    if top_k > 0 or top_p < 1.0:
        last = top_k_top_p_filtering(last, top_k=top_k, top_p=top_p)
    probs = last.softmax(dim=-1)
    idx = torch.multinomial(probs, num_samples=1) if sample_logits else probs.argmax(dim=-1, keepdim=True)
    return idx.to(dev), probs.to(dev)


def logits_to_probs(logits, temperature: float = 1.0, top_p: float = 1.0, top_k: int = None, **kwargs):
    """reference LS:86-91"""
    logits = logits / max(temperature, 1e-5)
    if (top_k or 0) > 0 or top_p < 1.0:
        logits = top_k_top_p_filtering(logits, top_k=top_k or 0, top_p=top_p)
    return logits.softmax(dim=-1)


def _ar_forward(model, tokens, pos, kv_len, key_start):
    """one AR step of the baseline decoder: a 1-row window through the same backbone (K3 append + K1 attention)"""
    if hasattr(model.attn, "params"):
        model.attn.params = None                      # kv_len by value: no SJD iteration blob in the plain AR loop
    return model.forward_window(tokens, pos, kv_len, key_start)


@torch.no_grad()
def generate(model, cond, max_new_tokens, emb_masks=None, cfg_scale=1.0, cfg_interval=-1, **sampling_kwargs):
    """The reference's plain auto-regressive LlamaGen decoder (LS:144-194: prefill + decode_n_tokens, one token per forward, every
    token drawn from the GLOBAL generator).  It is the non-SJD baseline `test_llamagen.py:20` imports next to the solver; here it
    runs on the same backbone and kernels (static cache, K1 with a 1-row window).  Returns LongTensor [B, max_new_tokens]."""
    if model.model_type == 'c2i':
        cond_combined = torch.cat([cond, torch.ones_like(cond) * model.num_classes]) if cfg_scale > 1.0 else cond
        T = 1
    elif model.model_type == 't2i':
        cond_combined = torch.cat([cond, torch.zeros_like(cond) + model.cls_embedding.uncond_embedding]) if cfg_scale > 1.0 else cond
        T = cond.shape[1]
    else:
        raise Exception("please check model type")
    B, dev = cond.shape[0], cond.device
    if B != 1:
        raise NotImplementedError("one prompt per call (the static cache rows of the CFG pair belong to one prompt)")
    Bc = cond_combined.shape[0]
    model.setup_cache(batch=Bc, s_max=((T + max_new_tokens + 32 + 31) // 32) * 32)
    for e in getattr(model, "_sjd_engines", {}).values():
        e.reset_graphs()
    if emb_masks is not None:
        assert emb_masks.shape[0] == B and emb_masks.shape[-1] == T                                     # LS:169-170
        masks = torch.cat([emb_masks, emb_masks]) if cfg_scale > 1.0 else emb_masks
        ks = (masks.long().cumsum(-1) == 0).sum(-1).to(device=dev, dtype=torch.int32)                  # masked (left-padded) cond rows
    else:
        ks = torch.zeros(Bc, dtype=torch.int32, device=dev)
    if getattr(model, "attn", None) is None:
        from . import ops
        model.attn = ops.HipWindowAttention()
    emb = model.embed_condition(cond_combined)
    pos = torch.arange(T, device=dev)[None].repeat(Bc, 1)
    if hasattr(model.attn, "params"):
        model.attn.params = None
    logits = model.forward_embeds(emb, pos, 0, ks)                                                       # prefill (LS:95-104)

    def combine(lg, use_cfg):
        if cfg_scale > 1.0:
            c, u = torch.split(lg, len(lg) // 2, dim=0)
            return u + (c - u) * cfg_scale if use_cfg else c
        return lg

    seq = torch.empty((B, max_new_tokens), dtype=torch.long, device=dev)
    tok = sample(combine(logits, True), **sampling_kwargs)[0]
    seq[:, 0:1] = tok
    cfg_flag = True
    for i in range(max_new_tokens - 1):                                                                  # decode_n_tokens (LS:123-142)
        if cfg_interval > -1 and i > cfg_interval:
            cfg_flag = False
        x = tok.view(-1, 1).repeat(Bc // B, 1)
        p_ = torch.full((Bc, 1), T + i, dtype=torch.long, device=dev)
        lg = _ar_forward(model, x, p_, T + i, ks)
        tok = sample(combine(lg, cfg_flag), **sampling_kwargs)[0]
        seq[:, i + 1:i + 2] = tok
    return seq


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

    def __init__(self, model, image_top_k, image_top_p, noise_device=None):
        self.model = model
        self.image_top_k = image_top_k
        self.image_top_p = image_top_p
        # None: every draw on the model's device, as the reference does.  "cpu": the first-token draw and the SJD noise streams come
        # from CPU generators (the golden fixtures were produced by CPU runs of the reference)
        self.noise_device = noise_device

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
        return sample(logits, noise_device=self.noise_device, **sampling_kwargs)[0], T

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
        model.sjd_noise_device = self.noise_device
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
