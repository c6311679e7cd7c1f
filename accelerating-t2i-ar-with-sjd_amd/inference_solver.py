"""Mirror of reference lumina_mgpt/inference_solver.py::FlexARInferenceSolver for the SJD hot path.

The reference class is ~90 % tokenizer / VQ-GAN glue (FlexARItemProcessor, decode_image, gradio helpers) around one call:
`self.model.generate(prompt_ids, generation_config, logits_processor=..., streamer=...)` (IS:347-350).  Those assets
(./ckpts/chameleon/tokenizer/*, multi-GB checkpoints) do not exist on the GPU box, so this mirror keeps the constructor /
attribute surface the SJD installers touch (`.model`, `.model.model`, `.item_processor`, `.device`, `.dtype`,
`create_logits_processor`, `generate`) and works on token ids: `generate_ids(prompt_ids, max_gen_len, logits_processor)`
is exactly IS:335-354 without the tokenizer, and `generate(images, qas, ...)` delegates to it when an `item_processor`
with the reference's `process_item` / `decode_image` methods is supplied.
"""
from typing import List, Optional

import torch

from . import backbones as BB
from . import ops


class _EosCriteria:
    def __init__(self, eos_token_id):
        self.eos_token_id = list(eos_token_id)


class AutoregressiveGuidance:
    """Stands where the reference puts LLMImageStartTriggeredUnbatchedClassifierFreeGuidanceLogitsProcessor (IS:16-132; same constructor names):
    it carries the guidance scale and the image token ids.  The reference's processor runs a second, unconditional forward per token on the
    context from the image-start token on and returns scale * (scores - uncond) + uncond (IS:124-127); here that forward is the uncond row of the
    one window forward and the combine is kernel K2's (u + scale * (c - u): the same fp32 operations)."""

    def __init__(self, guidance_scale, image_start_token_id, image_end_token_id, image_next_line_token_id, patch_size=32, model=None, **_):
        self.guidance_scale, self.patch_size = guidance_scale, patch_size
        self.image_start_token_id, self.image_end_token_id, self.image_next_line_token_id = image_start_token_id, image_end_token_id, image_next_line_token_id

    def __call__(self, input_ids, scores):
        raise RuntimeError("AutoregressiveGuidance is applied inside the window forward + kernel K2 (FlexARInferenceSolver.generate_ids), not called")


class FlexARInferenceSolver:
    """reference IS:273-450.  `model_path` may be a directory with `config.json` + `*.safetensors` / `pytorch_model.bin`
    holding reference (HF Chameleon) weights, or `model=` may pass a ready `sjd_amd.backbones.ChameleonBackbone`."""

    def __init__(self, model_path=None, precision="bf16", target_size=512, cache_dir=None, device="cpu", tokenizer=None,
                 model: Optional[BB.ChameleonBackbone] = None, item_processor=None, fused=True, gemm="sjd"):
        self.dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[precision]
        self.device = torch.device(device)
        self.target_size = target_size
        if model is None:
            model = self._load(model_path)
        self.model = model.to(device=self.device, dtype=self.dtype).eval()
        if self.model.attn is None and self.device.type == "cuda":
            self.model.attn = ops.HipWindowAttention()
        if fused and self.device.type == "cuda" and self.dtype != torch.float32 and getattr(self.model, "_ops", None) is None:
            self.model.enable_fused(ops, gemm=gemm)
        if item_processor is None:
            # IS:290-293: in a maintainer's checkout of the reference (`./lumina_mgpt/` on sys.path, test_lumina_mgpt.py:3-5, tokenizer and
            # VQ-GAN files under ./ckpts/) the reference's own item processor is importable -- build it exactly as the reference does, so
            # that test_lumina_mgpt.py:49 + :130 run verbatim there.  It does not exist on the GPU box (and is not part of the hot path):
            # then generate() says so and generate_ids() is the entry point.
            try:
                from data.item_processor import FlexARItemProcessor          # noqa: the reference's module, never vendored here
                kw = {} if tokenizer is None else {"tokenizer": tokenizer}
                item_processor = FlexARItemProcessor(with_decoder=True, target_size=target_size, device=device, **kw)
            except (ImportError, FileNotFoundError, OSError) as e:           # not a reference checkout, or its tokenizer / VQ-GAN assets are missing;
                import logging                                               # anything else (corrupt checkpoint, OOM, wrong device) propagates
                logging.getLogger("sjd_amd.inference_solver").info("reference item processor not available (%s: %s): generate_ids() is the entry point",
                                                                   type(e).__name__, e)
                item_processor = None
        self.item_processor = item_processor

    @staticmethod
    def _load(model_path):
        import json
        import os
        if model_path is None or not os.path.isdir(model_path):
            raise FileNotFoundError(f"{model_path!r}: no local checkpoint directory (there is no network access here); pass "
                                    "model=<ChameleonBackbone> or a directory with config.json and the weight files")
        cfg = json.load(open(os.path.join(model_path, "config.json")))
        args = BB.ChameleonArgs(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                                num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                                num_key_value_heads=cfg.get("num_key_value_heads", cfg["num_attention_heads"]),
                                rms_norm_eps=cfg.get("rms_norm_eps", 1e-5), rope_theta=cfg.get("rope_theta", 10000.0),
                                qk_norm=bool(cfg.get("qk_norm", str(cfg.get("model_type", "chameleon")).lower().startswith("chameleon"))),
                                max_position_embeddings=cfg.get("max_position_embeddings", 4096))
        model = BB.ChameleonBackbone(args)
        sd = {}
        for f in sorted(os.listdir(model_path)):
            if f.endswith(".safetensors"):
                from safetensors.torch import load_file
                sd.update(load_file(os.path.join(model_path, f)))
            elif f.endswith(".bin"):
                sd.update(torch.load(os.path.join(model_path, f), map_location="cpu"))
        own = model.state_dict()
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
        model.load_state_dict({k: sd[k] for k in own})      # VQ-VAE / vision keys of the checkpoint are not on the hot path
        return model

    def create_logits_processor(self, cfg=3.0, image_top_k=2000, text_top_k=10):
        """reference IS:417-450 -- the processors of the AUTOREGRESSIVE baseline (no renew_pipeline_sampler applied): the image-start-triggered
        classifier-free guidance (IS:16-132), the image grammar (IS:134-224) and the interleaved top-k (IS:226-270).  One token per forward is a
        draft window of ONE row, so the grammar and the top-k are the 3-dim processors of the SJD path at window 1 (same decisions per row:
        tests/test_oracle_golden.py::test_loop_lumina_autoregressive_baseline pins the whole AR loop on reference runs); the guidance processor
        is a marker that carries the scale and the ids -- its second forward is the uncond row of the window forward (AutoregressiveGuidance)."""
        from transformers.generation.logits_process import LogitsProcessorList
        from .scheduler.logit_processor_3dim import MultiTokensVLLogitsProcessor, MultiTokensInterleavedTopKLogitsWarper
        ip = self.item_processor
        tok = (lambda name, default: ip.token2id(getattr(ip, name))) if ip is not None else (lambda name, default: default)
        start, end, eol = tok("image_start_token", 8197), tok("image_end_token", 8196), tok("new_line_token", 8803)
        V = self.model.config.vocab_size if hasattr(self.model, "config") else self.model.vocab_size
        lp = LogitsProcessorList()
        lp.append(AutoregressiveGuidance(guidance_scale=cfg, image_start_token_id=start, image_end_token_id=end, image_next_line_token_id=eol, patch_size=32))
        lp.append(MultiTokensVLLogitsProcessor(image_start_token_id=start, image_end_token_id=end, image_next_line_token_id=eol, patch_size=32, voc_size=V,
                                               device=self.device))
        lp.append(MultiTokensInterleavedTopKLogitsWarper(image_top_k=image_top_k, text_top_k=text_top_k, image_start_token_id=start, image_end_token_id=end))
        return lp

    @torch.no_grad()
    def _generate_autoregressive(self, prompt, logits_processor, generation_config, streamer):
        """The reference's AR decode (HF _sample around IS:417-450's processors) on the SJD kernels: window 1 -- K5, one window forward over
        the cond / uncond rows, K2's draw, K4's single-row short-circuit (JL:344-350) per token -- with the uncond row's context starting at the
        image-start token (IS:62-63; the SJD sampler keeps the last prompt token only, JL:755-758).  Noise comes from the device's default
        generator, as HF's torch.multinomial takes it (torch.manual_seed(s) before the call makes a run repeatable).  The published AR-vs-SJD
        ratio is this path's step count and time against the renewed solver's on the same prompt (bench.py: `ar_baseline`)."""
        from .scheduler.jacobi_iteration_lumina_mgpt import renew_sampler, hf_generate
        procs = list(logits_processor)
        guide = [p for p in procs if isinstance(p, AutoregressiveGuidance)]
        scale = float(guide[0].guidance_scale) if guide else 1.0
        start = guide[0].image_start_token_id if guide else None
        ids = prompt[0].tolist()
        mask = torch.ones(1, len(ids), dtype=torch.long, device=prompt.device)
        if guide and scale != 1.0:
            if start not in ids:
                raise NotImplementedError("the AR baseline entry point decodes a prompt that ends inside an image (<image-start> h w): text-first "
                                          "prompts reach the guidance's context only once the model has emitted the image-start token")
            u0 = len(ids) - 1 - ids[::-1].index(start)                 # IS:108: the LAST image-start token
            mask = mask.repeat(2, 1)
            mask[1, :u0] = 0                                           # the uncond row sees the prompt from <image-start> on (IS:62-63)
        model, cls = self.model, self.model.__class__
        keep = {k: model.__dict__.get(k) for k in ("_sjd_engines",)}
        try:
            model.__class__ = renew_sampler(cls)
            model._init_new_params(jacobi_loop_interval_l=1, jacobi_loop_interval_r=1 << 20, max_num_new_tokens=1, guidance_scale=scale, seed=None,
                                   do_cfg=True, prefix_token_sampler_scheme="speculative_jacobi")
            if getattr(self, "_ar_engines", None):
                model._sjd_engines = self._ar_engines                  # (the window-1 engine and its captured graphs survive between calls)
            out = hf_generate(model, prompt, generation_config, logits_processor=[p for p in procs if not isinstance(p, AutoregressiveGuidance)],
                              streamer=streamer, attention_mask=mask)
            self._ar_engines, self.last_ar_stats = model._sjd_engines, model.last_sjd_stats
        finally:
            model.__class__ = cls
            for k, v in keep.items():
                if v is None:
                    model.__dict__.pop(k, None)
                else:
                    model.__dict__[k] = v
        return out

    @torch.no_grad()
    def generate_ids(self, prompt_ids: List[int], max_gen_len: int, logits_processor=None, streamer=None, temperature=1.0):
        """IS:335-354 on token ids: GenerationConfig(max_new_tokens, do_sample, eos 8710) -> model._sample."""
        from transformers import GenerationConfig
        if logits_processor is None:
            logits_processor = self.create_logits_processor()
        prompt = torch.tensor([prompt_ids], dtype=torch.int64, device=self.device)
        max_length = len(prompt_ids) + max_gen_len
        gc = GenerationConfig(max_new_tokens=max_gen_len, max_length=max_length, temperature=temperature, top_k=None, do_sample=True,
                              eos_token_id=[8710])
        if not hasattr(self.model, "_sample"):            # not renewed: the reference's autoregressive baseline (IS:347-350 -> HF _sample)
            out = self._generate_autoregressive(prompt, logits_processor, gc, streamer)
            ids = out[0, len(prompt_ids):].tolist()
            if ids and ids[-1] == 8710:
                ids = ids[:-1]
            return ids
        out = self.model._sample(prompt, logits_processor, [_EosCriteria(getattr(self, "eos_token_ids", [8710]))], gc, False, streamer,
                                 attention_mask=torch.ones_like(prompt))
        ids = out[0, len(prompt_ids):].tolist()
        if ids and ids[-1] == 8710:
            ids = ids[:-1]
        return ids

    def generate(self, images, qas, max_gen_len, temperature, logits_processor=None, streamer=None):
        """reference IS:299-354 -> (text, [PIL.Image]).  Needs the reference's item processor (tokenizer + VQ decoder)."""
        if self.item_processor is None:
            raise NotImplementedError("tokenizer / VQ-GAN assets are not part of the SJD hot path; use generate_ids(prompt_ids, ...) "
                                      "or construct the solver with item_processor=<reference FlexARItemProcessor>")
        conversations = [{"from": "human", "value": q} if i % 2 == 0 else {"from": "gpt", "value": a}
                         for q, a in qas for i in range(2)]
        item = {"image": images, "conversations": conversations}
        prompt = []
        for value in self.item_processor.process_item(item):          # IS:325-333: ints, or dicts holding an image's ids
            prompt += [value] if isinstance(value, int) else list(value["input_ids"])
        ids = self.generate_ids(prompt, max_gen_len, logits_processor, streamer, temperature)
        return self.decode_ids(ids)

    def get_streamer(self):
        """reference IS:295-296: HF's TextStreamer over the item processor's tokenizer (needs the reference's item processor)"""
        if self.item_processor is None:
            raise NotImplementedError("get_streamer() needs the reference's item processor (its tokenizer); pass streamer=<any object with put / end> instead")
        from transformers import TextStreamer
        return TextStreamer(self.item_processor.tokenizer)

    def decode_image(self, tokens: List[int]):
        """reference IS:402-403"""
        if self.item_processor is None:
            raise NotImplementedError("decode_image() needs the reference's item processor (VQ-GAN decoder); sjd_amd.detokenizers holds the decoders themselves")
        return self.item_processor.decode_image(tokens)

    @staticmethod
    def create_image_grid(images, rows, cols):
        """reference IS:405-415: paste rows x cols equally sized PIL images into one"""
        from PIL import Image
        width, height = images[0].size
        grid = Image.new("RGB", (cols * width, rows * height))
        for i, img in enumerate(images):
            grid.paste(img, ((i % cols) * width, (i // cols) * height))
        return grid

    def decode_ids(self, tokens: List[int]):
        """reference IS:356-400: split at <racm3:break>(8197) ... <eoss>(8196) spans; images go through item_processor.decode_image."""
        text_ids, images, i = [], [], 0
        while i < len(tokens):
            if tokens[i] == 8197 and 8196 in tokens[i:]:
                j = tokens.index(8196, i)
                images.append(self.item_processor.decode_image(tokens[i:j + 1]) if self.item_processor is not None else tokens[i:j + 1])
                i = j + 1
            else:
                text_ids.append(tokens[i])
                i += 1
        text = self.item_processor.tokenizer.decode(text_ids) if self.item_processor is not None else text_ids
        return text, images
