"""Mirror of reference scheduler/jacobi_iteration_anhole.py (Anole / HF-Chameleon adapter): builds the image-only
processor list of JA:194-232 from the 3d descriptor classes and installs the SJD sampler."""
from .jacobi_iteration_lumina_mgpt import renew_sampler, renew_backbone
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
        """reference JA:97-288 (parameter plumbing; generate() of the HF pipeline calls `_sample`)."""

        def _init_new_params(self, guidance_scale=3.0, image_top_k=2000, text_top_k=10, **kwargs):
            self.cfg = guidance_scale
            self.image_top_k = image_top_k
            self.text_top_k = text_top_k

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
