"""reference import path model_wrappers.model_loader (reference ML:25-574): `load_pretrained_model(model_name, **kw)` and
`get_forward_func(model_name, model, **kw) -> sample_fn(prompt)` with the reference's name-substring dispatch, keyword names and
defaults, in front of the MI355X SJD engine.

What differs, and why: the reference loaders pull multi-GB checkpoints, tokenizers and VQ models from the HF hub.  None of that
exists on the GPU box and none of it is on the SJD hot path, so every loader here takes its backbone
either from a LOCAL checkpoint directory (`model_name` / `cache_dir` pointing at one) or ready-made through `model=` (e.g.
synthetic weights), and the returned `sample_fn` works at the token-id level: it accepts a list / tensor of prompt token ids (or a
string when a tokenizer / item processor was supplied) and returns what the reference returns when the matching VQ decoder is
attached, else the generated token ids.  The SJD installation itself (`renew_*`, `_init_new_params`, interval formulas) is the
reference's, line by line.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lumina_mgpt.inference_solver import FlexARInferenceSolver  # noqa: E402
from scheduler.jacobi_iteration_lumina_mgpt import renew_pipeline_sampler  # noqa: E402
from scheduler.jacobi_iteration_anhole import renew_pipeline_sampler as renew_pipeline_sampler_anhole  # noqa: E402
from scheduler.jacobi_iteration_emu3 import renew_solver as renew_solver_emu3  # noqa: E402


def _backbone_from_dir(path, kind):
    if path is None or not os.path.isdir(path):
        raise FileNotFoundError(f"{kind}: no local checkpoint directory at {path!r} (the GPU box has no hub access); pass model=<backbone>")
    return FlexARInferenceSolver._load(path)


def load_lumina_mgpt(cache_dir="./ckpts", model_name="Alpha-VLLM/Lumina-mGPT-7B-768", target_size=768, seed=1, max_num_new_tokens=16,
                     multi_token_init_scheme='random', guidance_scale=7.0, device="cpu", model=None, item_processor=None, **kwargs):
    """reference ML:25-60"""
    path = model_name if os.path.isdir(model_name) else os.path.join(cache_dir, model_name)
    inference_solver = FlexARInferenceSolver(model_path=path, precision="bf16", target_size=target_size, cache_dir=cache_dir, device=device,
                                             model=model, item_processor=item_processor)
    return renew_pipeline_sampler(inference_solver, jacobi_loop_interval_l=1,
                                  jacobi_loop_interval_r=(target_size // 16) ** 2 + target_size // 16 - 10,
                                  max_num_new_tokens=max_num_new_tokens, guidance_scale=guidance_scale, seed=seed,
                                  multi_token_init_scheme=multi_token_init_scheme, do_cfg=True, **kwargs)


def load_anole(cache_dir="./ckpts", model_name="leloy/Anole-7b-v0.1-hf", target_size=512, seed=1, max_num_new_tokens=16,
               multi_token_init_scheme='random', guidance_scale=7.0, device="cpu", dtype=torch.bfloat16, image_top_k=2000, text_top_k=10,
               prefix_token_sampler_scheme='speculative_jacobi', model=None, processor=None, **kwargs):
    """reference ML:62-110 -> dict(processor, model).  `model`: a ChameleonBackbone (HF Chameleon state-dict keys)."""
    import sjd_amd.ops as ops
    if model is None:
        path = model_name if os.path.isdir(model_name) else os.path.join(cache_dir, model_name)
        model = _backbone_from_dir(path, "anole")
    model = model.to(device=device, dtype=dtype).eval()
    if model.attn is None and torch.device(device).type == "cuda":
        model.attn = ops.HipWindowAttention()
        model.enable_fused(ops, gemm="sjd")
    if not hasattr(model.model, "image_seq_length"):
        model.model.image_seq_length = getattr(processor, "image_seq_length", (target_size // 16) ** 2)
    model = renew_pipeline_sampler_anhole(model, processor, jacobi_loop_interval_l=1,
                                          jacobi_loop_interval_r=(target_size // 16) ** 2 + target_size // 16 - 10,
                                          max_num_new_tokens=max_num_new_tokens, guidance_scale=guidance_scale, seed=seed,
                                          multi_token_init_scheme=multi_token_init_scheme, do_cfg=True, image_top_k=image_top_k,
                                          text_top_k=text_top_k, prefix_token_sampler_scheme=prefix_token_sampler_scheme, **kwargs)
    return dict(processor=processor, model=model)


def load_emu3(cache_dir="./ckpts", model_name="BAAI/Emu3-Gen", target_size=720, seed=1, max_num_new_tokens=16, multi_token_init_scheme='random',
              guidance_scale=7.0, device="cpu", dtype=torch.bfloat16, image_top_k=2048, text_top_k=10,
              prefix_token_sampler_scheme='speculative_jacobi', model=None, processor=None, **kwargs):
    """reference ML:112-192 -> dict(processor, model, GENERATION_CONFIG, logits_processor).  `processor`: an
    emu3.mllm.processing_emu3.Emu3Processor (or anything with build_prefix_constrained_fn(h, w))."""
    from transformers.generation.configuration_utils import GenerationConfig
    import sjd_amd.ops as ops
    if model is None:
        path = model_name if os.path.isdir(model_name) else os.path.join(cache_dir, model_name)
        model = _backbone_from_dir(path, "emu3")
    if processor is None:
        raise ValueError("load_emu3 needs processor=<Emu3Processor> (tokenizer ids of the visual / control tokens)")
    model = model.to(device=device, dtype=dtype).eval()
    if model.attn is None and torch.device(device).type == "cuda":
        model.attn = ops.HipWindowAttention()
        model.enable_fused(ops, gemm="sjd")
    cfg = getattr(model, "config", None)
    eos = getattr(cfg, "eos_token_id", None) or processor.build_prefix_constrained_fn(1, 1).eos_token
    pad = getattr(cfg, "pad_token_id", None) or processor.build_prefix_constrained_fn(1, 1).pad_token
    if cfg is None:
        model.config = type("Cfg", (), dict(eos_token_id=eos, pad_token_id=pad, image_area=target_size * target_size))()
    GENERATION_CONFIG = GenerationConfig(use_cache=True, eos_token_id=eos, pad_token_id=pad, max_new_tokens=40960, do_sample=True,
                                         top_k=image_top_k)
    h, w = target_size // 8, target_size // 8
    gen_kwargs = dict(mode='G', ratio="1:1", image_area=model.config.image_area, return_tensors="pt")
    model, logits_processor = renew_solver_emu3(model, processor, h=h, w=w, jacobi_loop_interval_l=1, jacobi_loop_interval_r=h * (w + 1) - 1,
                                                max_num_new_tokens=max_num_new_tokens, guidance_scale=guidance_scale, seed=seed,
                                                multi_token_init_scheme=multi_token_init_scheme, do_cfg=True, image_top_k=image_top_k,
                                                text_top_k=text_top_k, prefix_token_sampler_scheme=prefix_token_sampler_scheme, **kwargs)
    return dict(processor=processor, model=model, GENERATION_CONFIG=GENERATION_CONFIG, logits_processor=logits_processor,
                processor_kwargs=gen_kwargs)


def load_llamagen(cache_dir="./ckpts", model_name="llamagen", target_size=512, seed=1, max_num_new_tokens=16, multi_token_init_scheme='random',
                  guidance_scale=7.5, device="cpu", dtype=torch.bfloat16, image_top_k=1000, text_top_k=10,
                  prefix_token_sampler_scheme='speculative_jacobi',
                  vq_params=dict(vq_model="VQ-16", codebook_size=16384, codebook_embed_dim=8, vq_ckpt="llamagen/vq_ds16_t2i.pt", downsample_size=16),
                  backbone_params=dict(gpt_model='GPT-XL', cls_token_num=120, gpt_type='t2i', t5_path='llamagen/t5-ckpt', t5_model_type='flan-t5-xl',
                                       t5_feature_max_len=120, no_left_padding=False),
                  is_compile=False, image_top_p=1.0, temperature=1.0, gpt_model=None, t5_model=None, vq_model=None, **kwargs):
    """reference ML:194-345 -> dict(model=LlamaGenSolver, gpt_model, t5_model, vq_model, ...).  Checkpoints are read from `cache_dir`
    when the files exist; `gpt_model=` / `t5_model=` / `vq_model=` take ready objects instead."""
    from llamagen.tokenizer.tokenizer_image.vq_model import VQ_models
    from llamagen.language.t5 import T5Embedder
    from llamagen.llamagen import GPT_models
    from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen
    from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler
    import sjd_amd.ops as ops
    latent_size = target_size // vq_params['downsample_size']
    if vq_model is None:
        vq_ckpt = os.path.join(cache_dir, vq_params['vq_ckpt'])
        if os.path.exists(vq_ckpt):
            vq_model = VQ_models[vq_params['vq_model']](codebook_size=vq_params['codebook_size'], codebook_embed_dim=vq_params['codebook_embed_dim'])
            vq_model.load_state_dict(torch.load(vq_ckpt, map_location="cpu")["model"])
            vq_model.to(device).eval()
    if gpt_model is None:
        gpt_model = GPT_models[backbone_params['gpt_model']](block_size=latent_size ** 2, cls_token_num=backbone_params['cls_token_num'],
                                                             model_type=backbone_params['gpt_type'])
        gpt_ckpt = os.path.join(cache_dir, "llamagen/t2i_XL_stage1_256.pt" if target_size == 256 else "llamagen/t2i_XL_stage2_512.pt")
        if not os.path.exists(gpt_ckpt):
            raise FileNotFoundError(f"{gpt_ckpt} not found (no hub access here); pass gpt_model=<LlamaGenBackbone>")
        ck = torch.load(gpt_ckpt, map_location="cpu")
        weights = ck.get("model") or ck.get("module") or ck.get("state_dict")
        if weights is None:
            raise Exception("please check model weight")
        gpt_model.load_state_dict(weights, strict=False)
    gpt_model = gpt_model.to(device=device, dtype=dtype).eval()
    if gpt_model.attn is None and torch.device(device).type == "cuda":
        gpt_model.attn = ops.HipWindowAttention()
    jacobi_param_dict = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=latent_size ** 2 - max_num_new_tokens - 2,
                             max_num_new_tokens=max_num_new_tokens, guidance_scale=guidance_scale, seed=seed,
                             multi_token_init_scheme=multi_token_init_scheme, do_cfg=True, image_top_k=image_top_k,
                             prefix_token_sampler_scheme=prefix_token_sampler_scheme, **kwargs)
    gpt_model.__class__ = renew_llamagen(gpt_model.__class__)
    gpt_model._init_new_params(**jacobi_param_dict)
    gpt_model.__class__ = renew_sampler(gpt_model.__class__)
    gpt_model._init_new_params(**jacobi_param_dict)
    if t5_model is None and gpt_model.model_type == 't2i':
        t5_model = T5Embedder(device=device, local_cache=True, cache_dir=os.path.join(cache_dir, backbone_params['t5_path']),
                              dir_or_name=backbone_params['t5_model_type'], torch_dtype=dtype, model_max_length=backbone_params['t5_feature_max_len'])
    model = LlamaGenSolver(model=gpt_model, image_top_k=image_top_k, image_top_p=image_top_p)
    return dict(model=model, gpt_model=gpt_model, t5_model=t5_model, vq_model=vq_model, vq_params=vq_params, backbone_params=backbone_params,
                latent_size=latent_size, guidance_scale=guidance_scale, temperature=temperature, image_top_k=image_top_k, image_top_p=image_top_p)


def load_pretrained_model(model_name="Alpha-VLLM/Lumina-mGPT-7B-768", **kwargs):
    """reference ML:347-359"""
    if 'lumina-mgpt' in model_name.lower():
        return load_lumina_mgpt(model_name=model_name, **kwargs)
    elif 'anole' in model_name.lower():
        return load_anole(model_name=model_name, **kwargs)
    elif 'llamagen' in model_name.lower():
        return load_llamagen(model_name=model_name, **kwargs)
    elif 'emu3' in model_name.lower():
        return load_emu3(model_name=model_name, **kwargs)
    else:
        raise NotImplementedError


def _ids(prompt, device):
    if torch.is_tensor(prompt):
        return prompt.to(device).view(1, -1).long()
    if isinstance(prompt, (list, tuple)) and all(isinstance(t, int) for t in prompt):
        return torch.tensor([list(prompt)], dtype=torch.long, device=device)
    return None


def get_lumina_mgpt_forward_func(inference_solver, guidance_scale=7.0, image_top_k=2000, max_gen_len=8192, temperature=1.0, target_size=768, **kwargs):
    """reference ML:362-387.  Token ids in -> generated ids out; a string prompt goes through the solver's item processor."""
    def sample_fn(prompts):
        lp = inference_solver.create_logits_processor(cfg=guidance_scale, image_top_k=image_top_k)
        ids = _ids(prompts, inference_solver.device)
        if ids is not None:
            return inference_solver.generate_ids(ids[0].tolist(), max_gen_len, logits_processor=lp, temperature=temperature)
        text = f"Generate an image of {target_size}x{target_size} according to the following prompt:\n" + prompts
        generated = inference_solver.generate(images=[], qas=[[text, None]], max_gen_len=max_gen_len, temperature=temperature, logits_processor=lp)
        return generated[1][0]
    return sample_fn


def get_anole_forward_func(inference_solver, **kwargs):
    """reference ML:389-425"""
    processor, model = inference_solver['processor'], inference_solver['model']

    def sample_fn(prompts):
        dev = next(model.parameters()).device
        ids = _ids(prompts, dev)
        if ids is None:
            ids = processor("Generate an image of " + prompts, padding=True, return_tensors="pt")["input_ids"].to(dev)
        n_new = model.model.image_seq_length + 2
        out = model.generate(ids, multimodal_generation_mode="image-only", max_new_tokens=n_new, do_sample=True)
        response = out[:, ids.shape[-1]:]
        if getattr(model.model, "vqmodel", None) is None:
            return response
        return model.decode_image_tokens(response[:, 1:-1])
    return sample_fn


def get_emu3_forward_func(inference_solver, not_decoded_imgs=False, **kwargs):
    """reference ML:427-496.  prompts: a string (needs the processor's tokenizer) or a (pos_ids, neg_ids) pair of id lists."""
    processor, model = inference_solver['processor'], inference_solver['model']
    gc, logits_processor = inference_solver['GENERATION_CONFIG'], inference_solver['logits_processor']

    def sample_fn(prompts):
        dev = next(model.parameters()).device
        if isinstance(prompts, (tuple, list)) and len(prompts) == 2 and _ids(prompts[0], dev) is not None:
            pos_ids, neg_ids = _ids(prompts[0], dev), _ids(prompts[1], dev)
        else:
            pk = inference_solver.get('processor_kwargs', {})
            pos_ids = torch.as_tensor(processor(text=prompts + " masterpiece, film grained, best quality.", **pk)["input_ids"]).to(dev)
            neg_ids = torch.as_tensor(processor(text="lowres, bad anatomy, bad hands, text, error, missing fingers, extra digit, fewer digits, cropped, "
                                                     "worst quality, low quality, normal quality, jpeg artifacts, signature, watermark, username, blurry.",
                                                **pk)["input_ids"]).to(dev)
        mi = model.prepare_batch_cfg_model_inputs(pos_ids, neg_input_ids=neg_ids, attention_mask=None)
        out = model.generate(mi['pos_input_ids'], gc, logits_processor=logits_processor, attention_mask=mi['attention_mask'], neg_input_ids=neg_ids)[0]
        if not_decoded_imgs or getattr(processor, "vision_tokenizer", None) is None or not hasattr(processor.vision_tokenizer, "decode"):
            return out
        images = processor.decode(out)
        return images[-1] if images else out
    return sample_fn


def get_llamagen_forward_func(inference_solver, use_ar_baseline=False, **kwargs):
    """reference ML:498-562.  prompts: a caption string (needs t5_model) or a (caption_embs [1,T,C], emb_masks [1,T]) pair / class-id tensor.
    The released reference calls the plain AR `generate` here (ML:546 `llamagen_original_generate`); use_ar_baseline=True does the same,
    the default runs the SJD solver that load_llamagen installed."""
    from llamagen.llamagen_solver import generate as llamagen_original_generate
    s = inference_solver
    model, gpt_model, vq_model, latent_size = s['model'], s['gpt_model'], s['vq_model'], s['latent_size']

    def sample_fn(prompts):
        if isinstance(prompts, str):
            embs, masks = s['t5_model'].get_text_embeddings([prompts])
            if not s['backbone_params']['no_left_padding']:                      # the reference's "naive left padding" (ML:527-539)
                new_masks = torch.flip(masks, dims=[-1])
                embs = torch.stack([torch.cat([e[int(m.sum()):], e[:int(m.sum())]]) for e, m in zip(embs, masks)])
                masks = new_masks
            c_indices, c_masks = embs * masks[:, :, None], masks
        elif isinstance(prompts, (tuple, list)):
            c_indices, c_masks = prompts
        else:
            c_indices, c_masks = prompts, None                                   # class ids (c2i)
        gen = (lambda *a, **k: llamagen_original_generate(gpt_model, *a, **k)) if use_ar_baseline else model.generate
        index_sample = gen(c_indices, latent_size ** 2, c_masks, cfg_scale=s['guidance_scale'], temperature=s['temperature'],
                           top_k=s['image_top_k'], top_p=s['image_top_p'], sample_logits=True)
        if vq_model is None:
            return index_sample
        qz = [len(c_indices), s['vq_params']['codebook_embed_dim'], latent_size, latent_size]
        return vq_model.decode_code(index_sample, qz).clamp(-1, 1)
    return sample_fn


def get_forward_func(model_name, model, **kwargs):
    """reference ML:564-574"""
    if 'lumina-mgpt' in model_name.lower():
        return get_lumina_mgpt_forward_func(model, **kwargs)
    elif 'anole' in model_name.lower():
        return get_anole_forward_func(model, **kwargs)
    elif 'llamagen' in model_name.lower():
        return get_llamagen_forward_func(model, **kwargs)
    elif 'emu3' in model_name.lower():
        return get_emu3_forward_func(model, **kwargs)
    else:
        raise NotImplementedError
