"""Mirror of reference scheduler/jacobi_iteration_lumina_mgpt.py -- the drop-in boundary of the SJD hot path.

Same entry points, argument names, defaults and error behaviour as the reference (file:line in each docstring);
behind them the decode loop is the HIP engine (sjd_amd/engine.py -> libsjd_hip.so), not ATen ops:

    renew_sampler(model_class)      -> class with _init_new_params(**kw) and the HF `_sample` hook      (JL:598-1251)
    renew_backbone(model_class)     -> class (the 3-D mask handling lives in kernel K1)                  (JL:1253-1338)
    renew_pipeline(model_class)     -> class with create_logits_processor(cfg, image_top_k, text_top_k)  (JL:432-468)
    renew_pipeline_sampler(pipe, **kw)                                                                   (JL:1340-1346)
    sampling_logits2tokens(...)     -> kernel K2 on CUDA tensors                                         (JL:82-132)
    SpeculativeSampler(...)(...)    -> kernel K4 on CUDA tensors                                         (JL:134-315)
    find_first_misaligned_token_inds, prefix_matching_next_tokens, check_is_force_no_cfg, set_seed

The model class handed to renew_sampler must be one of this package's backbones (sjd_amd/backbones.py: same
state-dict keys as the reference checkpoints) -- the transformer forward stays PyTorch-ROCm, its attention is K1.
"""
import ctypes
import random
from typing import Optional

import numpy as np
import torch

from .. import _lib as L
from .. import ops
from ..engine import SJDConfig, SJDEngine, WindowSpec
from .logit_processor_3dim import (MultiTokensInterleavedTopKLogitsWarper, MultiTokensVLLogitsProcessor,
                                   get_double_cfg_input_ids, grammar_from_processors)


def set_seed(seed: int):
    """reference JL:36-45"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def check_is_force_no_cfg(input_ids, image_start_token_id=None, image_end_token_id=None, guidance_scale=3., do_cfg=True):
    """reference JL:70-80 -- True when no image is open (CFG is then skipped).  Host integers, no device sync."""
    if (image_start_token_id is None) or (image_end_token_id is None):
        return False
    row = input_ids[0].tolist() if torch.is_tensor(input_ids) else list(input_ids[0])
    return row.count(image_start_token_id) == row.count(image_end_token_id)


def find_first_misaligned_token_inds(input_ids, next_tokens):
    """reference JL:317-333 (plain Jacobi decoding)."""
    a = input_ids.tolist() if torch.is_tensor(input_ids) else input_ids
    b = next_tokens.tolist() if torch.is_tensor(next_tokens) else next_tokens
    out = []
    for row, nxt in zip(a, b):
        idx = len(row)
        for i in range(1, len(row)):
            if row[i] != nxt[i - 1]:
                idx = i
                break
        out.append(idx)
    return out


def _draw(fill, shape, generator, dev):
    """Noise on the generator's own device (a CPU generator reproduces the reference's CPU stream on a GPU run)."""
    gdev = generator.device if generator is not None else dev
    return fill(torch.empty(shape, dtype=torch.float32, device=gdev)).to(dev)


def _rules_blob(device, n_rows, rules=(), resid=(), use_cfg=True, scheme=0):
    blob = ops.DeviceBlob(L.IterParams, device)
    blob.view.n_rows, blob.view.use_cfg, blob.view.scheme = n_rows, int(use_cfg), scheme
    for j, r in enumerate(rules):
        blob.view.rules[j] = r
    for j, r in enumerate(resid):
        blob.view.resid_rules[j] = r
    blob.upload()
    return blob


def sampling_logits2tokens(logits, all_collected_input_ids, unfinished_sequences, pad_token_id, output_token_num=1,
                           logits_processor=None, logits_warper=None, do_sample=True, has_eos_stopping_criteria=True,
                           do_cfg=False, guidance_scale=3., generator=None, is_force_no_cfg=False):
    """reference JL:82-132 on CUDA tensors: last `output_token_num` rows -> CFG -> grammar/top-k -> softmax ->
    multinomial, all inside kernel K2.  logits: [B_cfg, n, V] fp32.  Returns (next_tokens [1,n], probs [1,n,V])."""
    dev = logits.device
    n, V = output_token_num, logits.shape[-1]
    lg = logits[:, -n:, :].float().contiguous()
    # JL:107: the warpers (top-k / top-p / temperature) only apply when sampling
    procs = list(logits_processor or []) + (list(logits_warper or []) if do_sample else [])
    gr = grammar_from_processors(procs)
    ctx = all_collected_input_ids[0].tolist()
    gr.start(ctx)
    blob = _rules_blob(dev, n, gr.window_rules(n), use_cfg=do_cfg and not is_force_no_cfg)
    # do_sample=False (JL:127-129): no draw -- K2 still forms softmax(scores) (noise = 1: its token is the mode of p), the token is the argmax of
    # the processed SCORES as the reference takes it
    noise = (_draw(lambda t: t.exponential_(generator=generator), (n, V), generator, dev) if do_sample
             else torch.ones(n, V, dtype=torch.float32, device=dev))
    probs = torch.empty(n, V, dtype=torch.float32, device=dev)
    toks = torch.empty(n, dtype=torch.int64, device=dev)
    lu = lg[lg.shape[0] // 2] if (do_cfg and lg.shape[0] >= 2) else None
    ops.logits_to_probs_sample(lg[0], lu, guidance_scale, blob, noise, probs, ctypes.c_void_p(toks.data_ptr()))
    if not do_sample:
        # exactly torch.argmax(next_token_scores): the scores are the (CFG-combined) logits where the grammar allows a token -- p > 0 -- and -inf
        # elsewhere; two scores one ulp apart can round to the same p, so the argmax is taken on the scores themselves
        use_cfg = do_cfg and not is_force_no_cfg and lu is not None
        z = (guidance_scale * (lg[0] - lu) + lu) if use_cfg else lg[0]
        toks = torch.argmax(torch.where(probs > 0, z, torch.full_like(z, float("-inf"))), dim=-1)
    if has_eos_stopping_criteria and pad_token_id is not None:
        toks = toks * unfinished_sequences + pad_token_id * (1 - unfinished_sequences)     # JL:130
    return toks[None], probs[None]


class SpeculativeSampler:
    """reference JL:134-315.  __call__ runs kernel K4 (ballot over the accept tests, residual resample of the first
    rejected position).  draft_prob / advanced_prob: [1, L, V] fp32 CUDA tensors."""

    def __init__(self, collected_draft_logits=None, collected_advanced_logits=None, max_num_collected_logits=2,
                 generator=None, draft_type='jacobian_states', reject_sampling_relative_ids=None,
                 reject_sampling_draft_token_logits=None, sampling_last_draft_token=None):
        if draft_type != 'jacobian_states':
            raise NotImplementedError("only draft_type='jacobian_states' (target index = i-1) is used by the reference")
        self.generator = generator

    def __call__(self, draft_tokens, advanced_tokens, draft_prob, advanced_prob, logits_processor=None, logits_warper=None,
                 all_collected_input_ids=None, **kwargs):
        dev = advanced_prob.device
        Lw, V = draft_tokens.shape[1], advanced_prob.shape[-1]
        rs = _draw(lambda t: t.uniform_(0.0, 1.0, generator=self.generator), tuple(advanced_prob.shape), self.generator, dev)[0]   # JL:260
        win = draft_tokens[0].tolist()
        procs = list(logits_processor or []) + list(logits_warper or [])
        if procs:
            gr = grammar_from_processors(procs)
            gr.start(all_collected_input_ids[0].tolist())
            resid = gr.residual_rules(win)
        else:
            resid = [ops.make_rule() for _ in range(Lw - 1)]
        blob = _rules_blob(dev, Lw, resid=resid)
        state = ops.DeviceBlob(L.State, dev)
        for i in range(Lw):
            state.view.win_tok[i], state.view.tokens[i], state.view.q_src[i] = int(win[i]), int(advanced_tokens[0, i]), i
        state.dev.copy_(state.host)
        g_state = self.generator.get_state() if self.generator is not None else None
        noise2 = _draw(lambda t: t.exponential_(generator=self.generator), (1, V), self.generator, dev)[0]   # JL:237
        p = advanced_prob[0].float().contiguous()
        q = draft_prob[0].float().contiguous()
        ops.verify_accept(blob, state, p, q, rs.contiguous(), noise2, torch.empty(V, dtype=torch.float32, device=dev))
        st = state.download()
        if int(st.rejected) > 1:
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")      # torch.multinomial's message (JL:237)
        if g_state is not None and not st.rejected:
            self.generator.set_state(g_state)
        m = int(st.m)
        toks = torch.tensor([[st.tokens[i] for i in range(Lw)]], dtype=torch.int64, device=dev)
        scores = advanced_prob.clone()
        if m > 1:
            scores[0, :m - 1] = draft_prob[0, 1:m]                                                # JL:289
        return [m], toks, scores


def prefix_matching_next_tokens(model_input_ids, next_tokens, next_token_scores, is_prefilling_phase=False,
                                input_token_scores=None, prefix_token_sampler=None, **kwargs):
    """reference JL:335-376"""
    if is_prefilling_phase:
        return (model_input_ids.shape[1], next_tokens[:, -1:], next_tokens[:, next_tokens.shape[1]:],
                next_token_scores[:, -1:], next_token_scores[:, next_token_scores.shape[1]:])
    if prefix_token_sampler is not None:
        inds, next_tokens, next_token_scores = prefix_token_sampler(
            draft_tokens=model_input_ids, advanced_tokens=next_tokens, draft_prob=input_token_scores,
            advanced_prob=next_token_scores, **kwargs)
    else:
        inds = find_first_misaligned_token_inds(model_input_ids, next_tokens)
    m = min(inds)
    return m, next_tokens[:, :m], next_tokens[:, m:], next_token_scores[:, :m], next_token_scores[:, m:]


# ------------------------------------------------------------------------------------------------
def _stopping_to_limits(stopping_criteria, generation_config):
    eos, max_len = [], getattr(generation_config, "max_length", None) or (1 << 30)
    for c in (stopping_criteria or []):
        if hasattr(c, "eos_token_id"):
            e = c.eos_token_id
            eos += [int(t) for t in (e.tolist() if torch.is_tensor(e) else (e if isinstance(e, (list, tuple)) else [e]))]
        if hasattr(c, "max_length") and c.max_length is not None:
            max_len = min(max_len, int(c.max_length))
        if hasattr(c, "max_seq_length"):
            max_len = min(max_len, int(c.max_seq_length))
    return tuple(eos), max_len


class _EosStop:
    def __init__(self, eos_token_id):
        self.eos_token_id = [int(t) for t in (eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id])]


class _MaxLengthStop:
    def __init__(self, max_length):
        self.max_length = int(max_length)


def hf_generate(model, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None, streamer=None,
                attention_mask=None, neg_input_ids=None, input_ids=None, **kwargs):
    """What the reference drivers get from HF `GenerationMixin.generate` (transformers 4.47.1, third-party -- not in the reference
    tree) before it reaches the `_sample` hook, reduced to the arguments they use (test_emu3.py:163-169, IS:335-350, JA:137-272,
    ML:406-411): the generation length (max_new_tokens wins over max_length), EOS / max-length stopping criteria, the user's logits
    processors followed by a TopKLogitsWarper when `generation_config.top_k` is set, then
    `model._sample(input_ids, processors, criteria, generation_config, synced_gpus=False, streamer, attention_mask=, neg_input_ids=)`.
    `generation_config.temperature != 1` becomes a TemperatureLogitsWarper in front of that TopKLogitsWarper and `top_p < 1` a
    TopPLogitsWarper behind it, as in HF.  Anything that changes the distribution and has no kernel rule (beams, greedy) raises."""
    import copy
    from transformers import GenerationConfig
    from transformers.generation.logits_process import LogitsProcessorList
    from .logit_processor_3dim import TopKLogitsWarper
    ids = inputs if inputs is not None else input_ids
    if ids is None:
        raise ValueError("generate() needs input_ids")
    gc = copy.deepcopy(generation_config) if generation_config is not None else GenerationConfig(do_sample=True)
    for k in ("max_new_tokens", "max_length", "do_sample", "top_k", "top_p", "temperature", "eos_token_id", "pad_token_id", "num_beams"):
        if k in kwargs and kwargs[k] is not None:
            setattr(gc, k, kwargs.pop(k))
    if (getattr(gc, "num_beams", 1) or 1) != 1:
        raise NotImplementedError("the SJD hot path decodes one beam (num_beams=1)")
    P = ids.shape[1]
    if getattr(gc, "max_new_tokens", None) is not None:
        gc.max_length = P + int(gc.max_new_tokens)
    limit = getattr(getattr(model, "args", None), "max_position_embeddings", None)
    if limit is not None and (gc.max_length is None or gc.max_length > limit):
        import warnings
        warnings.warn(f"generate(): max_length {gc.max_length} exceeds the model's max_position_embeddings {limit}; the generation is cut "
                      f"at {limit} tokens (HF warns and goes on; rows past the context could never be positioned here)", stacklevel=2)
        gc.max_length = int(limit)
    procs = LogitsProcessorList(list(logits_processor or []))
    sampling = bool(getattr(gc, "do_sample", False))            # (greedy: HF builds no warpers; `_sample` takes the argmax of the processed scores, JL:127-129)
    if sampling and (getattr(gc, "temperature", None) or 1.0) != 1.0:       # HF appends the warpers after the user's processors: temperature first
        from .logit_processor_3dim import TemperatureLogitsWarper
        procs.append(TemperatureLogitsWarper(float(gc.temperature)))
    if sampling and getattr(gc, "top_k", None):
        procs.append(TopKLogitsWarper(int(gc.top_k)))
    if sampling and (getattr(gc, "top_p", None) or 1.0) < 1.0:              # ... then top-k, then top-p
        from .logit_processor_3dim import TopPLogitsWarper
        procs.append(TopPLogitsWarper(float(gc.top_p)))
    crit = list(stopping_criteria or [])
    if getattr(gc, "eos_token_id", None) is not None:
        crit.append(_EosStop(gc.eos_token_id))
    crit.append(_MaxLengthStop(gc.max_length))
    kw = {}
    if attention_mask is not None:
        kw["attention_mask"] = attention_mask
    if neg_input_ids is not None:
        kw["neg_input_ids"] = neg_input_ids
    return model._sample(ids, procs, crit, gc, False, streamer, **kw)


def renew_sampler(model_class):
    class JacobiSampler(model_class):
        """reference JL:598-1251"""

        def _init_new_params(self, jacobi_loop_interval_l=1, jacobi_loop_interval_r=(768 // 16) ** 2 + 768 // 16,
                             max_num_new_tokens=16, guidance_scale=3.0, seed=42, multi_token_init_scheme='random',
                             do_cfg=True, prefix_token_sampler_scheme='speculative_jacobi', use_chameleon_tokenizer=True,
                             _init_doubled_attn_mask_cfg=False, **kwargs):
            # reference JL:865-910.  img_vocab = Chameleon image ids 4..8195 for every model family (SURVEY.md
            # Appendix D); the reference reads them from ./ckpts/chameleon/tokenizer/text_tokenizer.json
            self.img_vocab_range = (4, 8196)
            self.jacobi_loop_interval_l = jacobi_loop_interval_l
            self.jacobi_loop_interval_r = jacobi_loop_interval_r
            self.max_num_new_tokens = max_num_new_tokens
            self.guidance_scale = guidance_scale
            self.seed = seed
            self.generator = None
            self.multi_token_init_scheme = multi_token_init_scheme
            self.do_cfg = do_cfg
            self.prefix_token_sampler_scheme = prefix_token_sampler_scheme
            self._init_doubled_attn_mask_cfg = _init_doubled_attn_mask_cfg
            self._sjd_engines = {}

        # ---- engine plumbing -------------------------------------------------------------------
        def _sjd_engine(self, n_batch, device):
            key = (n_batch, self.max_num_new_tokens, str(device))
            eng = self._sjd_engines.get(key)
            if getattr(self, "attn", None) is None:
                self.attn = ops.HipWindowAttention()           # K3 + K1; raises if libsjd_hip.so is missing
            if eng is None:
                eng = SJDEngine(self, self.vocab_size, device, max_window=self.max_num_new_tokens, n_batch=n_batch,
                                use_graph=getattr(self, "sjd_use_graph", True))
                self._sjd_engines[key] = eng
            return eng

        def _sjd_window_spec(self, input_ids, do_cfg, model_kwargs):
            """First-iteration inputs + per-batch-row visibility / position offsets from the HF-style kwargs
            (attention_mask [1,P] or pre-doubled [2,P]; neg_input_ids for pos/neg prompt CFG)."""
            dev = input_ids.device
            B = 2 if do_cfg else 1
            if hasattr(self, "tok_embeddings"):                        # LlamaGen: cond rows are already in the cache
                T = int(model_kwargs["attention_mask"].shape[1]) - 1   # prefill_num = T + 1 (LS:436-440)
                ks = getattr(self, "_sjd_key_start", None)
                ks = torch.zeros(B, dtype=torch.int32) if ks is None else torch.as_tensor(ks, dtype=torch.int32)
                return WindowSpec(first_tokens=input_ids[:, -1:].repeat(B, 1), first_positions=torch.full((B, 1), T, dtype=torch.long, device=dev),
                                  key_start=ks, pos_offset=torch.zeros(B, dtype=torch.long), kv_base=T)
            mask = model_kwargs.get("attention_mask")
            mask = torch.ones_like(input_ids) if mask is None else mask.to(dev)
            P = input_ids.shape[1]
            neg = model_kwargs.get("neg_input_ids")
            if do_cfg:
                if mask.shape[0] == 1:                                 # JL:1007-1014, 755-758
                    mask = mask.repeat(2, 1)
                    mask[1, :P - 1] = 0
                pad = getattr(getattr(self, "config", None), "pad_token_id", 0) or 0
                tokens = get_double_cfg_input_ids(input_ids, neg.to(dev), pad) if neg is not None else input_ids.repeat(2, 1)
            else:
                tokens = input_ids
            mask = mask[:, -tokens.shape[1]:].long()
            key_start = (mask.cumsum(-1) == 0).sum(-1).to(torch.int32)            # leading zeros = hidden prefix
            pos = mask.cumsum(-1) - 1                                             # JL:705-706
            pos = pos.masked_fill(mask == 0, 1)
            return WindowSpec(first_tokens=tokens, first_positions=pos, key_start=key_start,
                              pos_offset=-key_start.to(torch.long), kv_base=0)

        @torch.no_grad()
        def _sample(self, input_ids, logits_processor, stopping_criteria, generation_config, synced_gpus, streamer,
                    logits_warper=None, **model_kwargs):
            """reference JL:912-1249.  Returns LongTensor [1, P + N]."""
            assert not getattr(generation_config, "return_dict_in_generate", False)       # JL:1164
            if input_ids.shape[0] != 1:
                raise NotImplementedError("the reference supports one prompt per process (SURVEY.md Appendix B note 5)")
            if self.prefix_token_sampler_scheme not in ("speculative_jacobi", "jacobi"):
                raise ValueError(f"prefix_token_sampler_scheme: {self.prefix_token_sampler_scheme}")   # JL:1048
            dev = input_ids.device
            do_cfg = bool(self.do_cfg) and (self.guidance_scale != 1)                                  # JL:1002-1005
            do_sample = bool(getattr(generation_config, "do_sample", True))                            # JL:969
            procs = list(logits_processor or []) + (list(logits_warper or []) if do_sample else [])    # JL:107: warpers only when sampling
            eos, max_len = _stopping_to_limits(stopping_criteria, generation_config)
            grammar = grammar_from_processors(procs, prompt_len=input_ids.shape[1], max_length=max_len, vocab_size=getattr(self, "vocab_size", None))
            if getattr(grammar, "V", 0) is None:
                grammar.V = self.vocab_size
            spec = self._sjd_window_spec(input_ids, do_cfg, model_kwargs)
            cfg = SJDConfig(jacobi_loop_interval_l=self.jacobi_loop_interval_l, jacobi_loop_interval_r=self.jacobi_loop_interval_r,
                            max_num_new_tokens=self.max_num_new_tokens, guidance_scale=self.guidance_scale, seed=self.seed,
                            do_cfg=do_cfg, prefix_token_sampler_scheme=self.prefix_token_sampler_scheme,
                            multi_token_init_scheme=self.multi_token_init_scheme, img_vocab_lo=self.img_vocab_range[0],
                            img_vocab_n=self.img_vocab_range[1] - self.img_vocab_range[0], max_length=max_len, eos_token_ids=eos,
                            noise_device=getattr(self, "sjd_noise_device", None), do_sample=do_sample)
            B = 2 if do_cfg else 1
            if max_len >= (1 << 30):
                # no MaxLength criterion: bound the cache by the model's own context instead of asking for ~1e9 rows
                a_ = getattr(self, "args", None)
                lim = getattr(a_, "max_position_embeddings", None) or (getattr(a_, "block_size", 0) + getattr(a_, "cls_token_num", 0)) or None
                if lim is None:
                    raise ValueError("_sample needs a MaxLengthCriteria / generation_config.max_length (the model states no context limit)")
                max_len = cfg.max_length = int(lim)
            need = spec.kv_base + max(max_len, input_ids.shape[1]) + self.max_num_new_tokens + 32
            if self.cache is None or self.cache.k.shape[1] != B or self.cache.s_max < need:
                if spec.kv_base > 0:
                    # the conditioning rows are already IN the cache (LlamaGen prefill): re-allocating would decode from zeros.  The
                    # reference fails with a shape error when the CFG batch of `_sample` disagrees with the one `generate` prefilled.
                    have = None if self.cache is None else (int(self.cache.k.shape[1]), int(self.cache.s_max))
                    raise ValueError(f"prefilled KV cache (batch, rows) = {have} does not fit `_sample` (batch {B}, rows {need}): "
                                     "cfg_scale > 1 at prefill must match do_cfg / guidance_scale of the sampler")
                self.setup_cache(batch=B, s_max=((need + 31) // 32) * 32)
                for e in self._sjd_engines.values():
                    e.reset_graphs()
            eng = self._sjd_engine(B, dev)
            seq, stats = eng.decode(input_ids[0].tolist(), spec, grammar, cfg)
            if streamer is not None:
                for t in seq[input_ids.shape[1]:]:
                    streamer.put(torch.tensor([t]))
                streamer.end()
            self.last_sjd_stats = stats
            print("Time elapsed inner: ", stats.seconds)                                                 # JL:1218-1220
            print("gen loop num (NFE): ", stats.nfe)
            print("tokens length: ", len(seq))
            return torch.tensor([seq], dtype=torch.long, device=dev)

    if not hasattr(model_class, "generate"):
        # this package's backbones are plain nn.Modules: give them the HF-shaped entry point the drivers call
        # (`model.generate(input_ids, generation_config, logits_processor=..., attention_mask=..., neg_input_ids=...)`)
        JacobiSampler.generate = torch.no_grad()(hf_generate)
    return JacobiSampler


def renew_backbone(model_class):
    """reference JL:1253-1338 lifts the 3-D 0/1 mask to the 4-D additive mask inside `_update_causal_mask`.  Here the mask
    is never materialised: kernel K1 derives visibility from (key_start[b], kv_len, row), so the class is returned as is."""
    class JacobiBackbone(model_class):
        pass

    return JacobiBackbone


def renew_pipeline(model_class):
    class JacobiPipeline(model_class):
        """reference JL:432-468"""

        def _init_new_params(self, guidance_scale=3.0, image_top_k=2000, text_top_k=10, **kwargs):
            self.cfg = guidance_scale
            self.image_top_k = image_top_k
            self.text_top_k = text_top_k

        def create_logits_processor(self, cfg=3.0, image_top_k=2000, text_top_k=10):
            image_top_k = self.image_top_k if hasattr(self, 'image_top_k') else image_top_k
            text_top_k = self.text_top_k if hasattr(self, 'text_top_k') else text_top_k
            ip = getattr(self, "item_processor", None)
            tok = (lambda name, default: ip.token2id(getattr(ip, name))) if ip is not None else (lambda name, default: default)
            start, end, eol = tok("image_start_token", 8197), tok("image_end_token", 8196), tok("new_line_token", 8803)
            from transformers.generation.logits_process import LogitsProcessorList
            lp = LogitsProcessorList()
            lp.append(MultiTokensVLLogitsProcessor(image_start_token_id=start, image_end_token_id=end, image_next_line_token_id=eol,
                                                   patch_size=32, voc_size=self.model.config.vocab_size if hasattr(self.model, "config") else self.model.vocab_size,
                                                   device=getattr(self, "device", "cpu")))
            lp.append(MultiTokensInterleavedTopKLogitsWarper(image_top_k=image_top_k, text_top_k=text_top_k,
                                                             image_start_token_id=start, image_end_token_id=end))
            return lp

    return JacobiPipeline


def renew_pipeline_sampler(pipe_line, **kwargs):
    """reference JL:1340-1346"""
    pipe_line.__class__ = renew_pipeline(pipe_line.__class__)
    pipe_line._init_new_params(**kwargs)
    pipe_line.model.__class__ = renew_sampler(pipe_line.model.__class__)
    pipe_line.model._init_new_params(**kwargs)
    if hasattr(pipe_line.model, "model"):
        pipe_line.model.model.__class__ = renew_backbone(pipe_line.model.model.__class__)
    return pipe_line
