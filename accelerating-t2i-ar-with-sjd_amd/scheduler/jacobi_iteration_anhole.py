"""Mirror of reference scheduler/jacobi_iteration_anhole.py (Anole / HF-Chameleon adapter): builds the processor list of the
requested multimodal generation mode (JA:178-266) from the 3d descriptor classes and installs the SJD sampler."""
import torch

from .jacobi_iteration_lumina_mgpt import renew_sampler, renew_backbone, hf_generate
from .logit_processor_3dim import (AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d,
                                   AllowOnlyTokensInRelativeWindowLogitsProcessor3d,
                                   SuppressTokensAtBeginLogitsProcessor3d, SuppressTokensInIndexRangeLogitsProcessor3d,
                                   SuppressTokensLogitsProcessor3d, TopKLogitsWarper)


def image_only_processors(vocab_size, input_ids_length, max_length, image_seq_length, image_token_ids, boi_token_id=8197,
                          eoi_token_id=8196, eos_token_id=2, top_k=None, device="cpu"):
    """The processor list reference JA:183-232 hands to `_sample` in multimodal_generation_mode='image-only'."""
    from transformers.generation.logits_process import LogitsProcessorList
    allowed = set(image_token_ids) | {eos_token_id, boi_token_id, eoi_token_id}
    suppress = [t for t in range(vocab_size) if t not in allowed]
    procs = LogitsProcessorList([
        AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(trigger_token_id=boi_token_id, allowed_token_ids=[eoi_token_id],
                                                         offset=image_seq_length + 1, exclusive=True, device=device),
        AllowOnlyTokensInRelativeWindowLogitsProcessor3d(trigger_token_id=boi_token_id, allowed_token_ids=list(image_token_ids),
                                                         window_width=image_seq_length, exclusive=True, device=device),
        SuppressTokensInIndexRangeLogitsProcessor3d(suppress_tokens=[boi_token_id],
                                                    start_index=max_length - image_seq_length - 1, device=device),
        SuppressTokensLogitsProcessor3d(suppress_tokens=suppress, device=device),
        SuppressTokensAtBeginLogitsProcessor3d(begin_suppress_tokens=[eos_token_id], begin_index=input_ids_length, device=device),
    ])
    if top_k:
        procs.append(TopKLogitsWarper(top_k))
    return procs


def renew_pipeline_anole(model_class):
    class JacobiPipeline(model_class):
        """reference JA:97-288: `generate(..., multimodal_generation_mode=)` builds the mode's 3d processor list and hands over to HF
        generate -> `_sample`; `decode_image_tokens` maps the emitted BPE ids to VQ codes and decodes them."""

        def _init_new_params(self, guidance_scale=3.0, image_top_k=2000, text_top_k=10, **kwargs):
            self.cfg = guidance_scale
            self.image_top_k = image_top_k
            self.text_top_k = text_top_k
            # ChameleonImageVocabularyMapping of the reference (JA:56-95): contiguous image ids 4..8195, <racm3:break>, <eoss>
            if not hasattr(self, "image_token_ids"):
                self.image_token_ids = list(range(4, 8196))
            self.boi_token_id = getattr(self, "boi_token_id", 8197)
            self.eoi_token_id = getattr(self, "eoi_token_id", 8196)
            self.eos_token_id = getattr(getattr(self, "config", None), "eos_token_id", None) or getattr(self, "eos_token_id", 2)

        @torch.no_grad()
        def generate(self, inputs=None, generation_config=None, logits_processor=None, multimodal_generation_mode=None, **kwargs):
            """reference JA:137-272"""
            ids = inputs if inputs is not None else kwargs.get("input_ids")
            mode = multimodal_generation_mode or getattr(generation_config, "multimodal_generation_mode", None) or "text-only"
            L_img = self.model.image_seq_length if hasattr(self.model, "image_seq_length") else self.image_seq_length
            if mode == "image-only" and kwargs.get("max_length") is None and kwargs.get("max_new_tokens") is None and (
                    generation_config is None or (generation_config.max_length is None and generation_config.max_new_tokens is None)):
                kwargs["max_new_tokens"] = L_img + 2                                        # JA:116-126
            P = ids.shape[-1]
            mnt = kwargs.get("max_new_tokens", getattr(generation_config, "max_new_tokens", None))
            max_length = P + int(mnt) if mnt is not None else int(kwargs.get("max_length", getattr(generation_config, "max_length", 0)))
            if mode == "image-only":
                if max_length - P < L_img + 2:
                    import warnings
                    warnings.warn(f"the VQ decoder expects {L_img} image tokens wrapped in begin/end-of-image: max_new_tokens must be "
                                  f"at least {L_img + 2}, got {max_length - P}")
                procs = image_only_processors(self.vocab_size, P, max_length, L_img, self.image_token_ids, self.boi_token_id,
                                              self.eoi_token_id, self.eos_token_id, device=ids.device)
                procs = type(procs)(list(logits_processor or []) + list(procs))
            elif mode == "text-only":                  # JA:178-189
                from transformers.generation.logits_process import LogitsProcessorList
                procs = LogitsProcessorList(list(logits_processor or []) + [SuppressTokensLogitsProcessor3d(
                    suppress_tokens=list(self.image_token_ids) + [self.boi_token_id, self.eoi_token_id], device=ids.device)])
            elif mode == "interleaved-text-image":     # JA:233-260: the image-window processors without the global suppression
                from transformers.generation.logits_process import LogitsProcessorList
                procs = LogitsProcessorList(list(logits_processor or []) + [
                    AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(trigger_token_id=self.boi_token_id, allowed_token_ids=[self.eoi_token_id],
                                                                     offset=L_img + 1, exclusive=True, device=ids.device),
                    AllowOnlyTokensInRelativeWindowLogitsProcessor3d(trigger_token_id=self.boi_token_id, allowed_token_ids=list(self.image_token_ids),
                                                                     window_width=L_img, exclusive=True, device=ids.device),
                    SuppressTokensInIndexRangeLogitsProcessor3d(suppress_tokens=[self.boi_token_id], start_index=max_length - L_img - 1,
                                                                device=ids.device)])
            elif mode == "unrestricted":
                procs = logits_processor
            else:
                raise ValueError(f"Unknown multimodal generation mode: {mode}. Please choose one of 'unrestricted', 'text-only', 'image-only', "
                                 "or 'interleaved-text-image'.")          # JA:263-266
            kwargs.setdefault("do_sample", True)
            kwargs.pop("input_ids", None)
            return hf_generate(self, ids, generation_config, logits_processor=procs, **kwargs)

        def decode_image_tokens(self, bpe_tokens):
            """reference JA:274-316 (+ the truncating variant of renew_backbone_adapt_anole): BPE ids -> VQ codes -> pixels.
            Needs a VQ decoder on `self.model.vqmodel` (sjd_amd.detokenizers.ChameleonVQ with the checkpoint's weights)."""
            L_img = self.model.image_seq_length if hasattr(self.model, "image_seq_length") else self.image_seq_length
            if bpe_tokens.shape[1] != L_img:
                bpe_tokens = bpe_tokens[:, :L_img]
            vq = getattr(self.model, "vqmodel", None)
            if vq is None:
                raise RuntimeError("no VQ decoder attached: set model.model.vqmodel = sjd_amd.detokenizers.ChameleonVQ(...) with the checkpoint's weights")
            codes = (bpe_tokens - self.image_token_ids[0]).clamp_(0, len(self.image_token_ids) - 1)
            side = int(round(L_img ** 0.5))
            return vq.decode_code(codes, hw=(side, side))

    return JacobiPipeline


def renew_pipeline_sampler(pipe_line, processor, **kwargs):
    """reference JA:318-330"""
    if hasattr(pipe_line, "model"):
        pipe_line.model.__class__ = renew_backbone(pipe_line.model.__class__)
        if not hasattr(pipe_line.model, "image_seq_length") and processor is not None:
            pipe_line.model.image_seq_length = processor.image_seq_length
    pipe_line.__class__ = renew_pipeline_anole(pipe_line.__class__)
    pipe_line._init_new_params(**kwargs)
    pipe_line.__class__ = renew_sampler(pipe_line.__class__)
    pipe_line._init_new_params(**kwargs)
    return pipe_line
