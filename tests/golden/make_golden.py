#!/usr/bin/env python3
"""Generate the golden fixtures by IMPORTING THE REFERENCE in this container (CPU).

Run:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)

The reference (/root/reference, read-only) is imported with the in-memory shims of
_ref_shims.py.  Only inputs (seeds / small integer arrays) and the reference's
outputs are written; no reference source travels.  /root/reference does not exist
on the GPU box, so nothing at test time imports this script.

Fixtures
  fn_logits2tokens_lumina.npz   sampling_logits2tokens + MultiTokensVLLogitsProcessor +
                                MultiTokensInterleavedTopKLogitsWarper  (JL:82-132, LP:45-204)
  fn_logits2tokens_llamagen.npz TopKLogitsWarper + TopPLogitsWarper3d   (LS:458-470, LP:355-419)
  fn_emu3_grammar.npz           EOLLogitProcessor3d + TopK(2048)        (JE:41-151)
  fn_anole_grammar.npz          Anole image-only 3d processors          (JA:194-232, LP:207-353)
  fn_anole_modes.npz            Anole text-only / interleaved processors (JA:178-189, 233-260)
  fn_speculative_sampler.npz    SpeculativeSampler.__call__             (JL:247-315)
  fn_reguess.npz                get_multi_token_for_preparation('random') (JL:470-514)
  fn_temperature.npz            the same two with HF's TemperatureLogitsWarper in the processor list (GenerationConfig.temperature != 1)
  loop_llamagen.npz             whole _sample loop, tiny LlamaGen c2i   (JL:912-1249, LS:349-456)
  loop_lumina.npz               whole _sample loop, tiny Chameleon      (JL:912-1249, MC)
  loop_lumina_ar.npz            the AUTOREGRESSIVE baseline: HF generate() + the reference's non-SJD processors (IS:16-270, 417-450)
  vq_decoders.npz               image detokenizers: LlamaGen VQModel.decode_code and the Chameleon VQGAN decode of
                                image_tokenizer.pil_from_img_toks, small widths, per-key synthetic weights
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
# (ROOT must NOT be on sys.path: the repo root carries drop-in packages with the reference's own names -- scheduler/, llamagen/, emu3/ ... --
#  and, being regular packages, they would shadow the reference's __init__-less directories whatever the path order)
sys.path[:] = [p_ for p_ in sys.path if os.path.abspath(p_ or os.getcwd()) != ROOT]

import _ref_shims  # noqa: E402

LegacyCache, CompatMixin = _ref_shims.install()

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "sjd_synthetic", os.path.join(ROOT, "accelerating-t2i-ar-with-sjd_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)

import scheduler.jacobi_iteration_lumina_mgpt as JL  # noqa: E402  (the reference)
import scheduler.logit_processor_3dim as LP  # noqa: E402
assert JL.__file__.startswith("/root/reference/") and LP.__file__.startswith("/root/reference/"), (JL.__file__, LP.__file__)
from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def sample_cols(V, n=64, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, V, (n,), generator=g)


# ----------------------------------------------------------------------------------------------
# function-level vectors
# ----------------------------------------------------------------------------------------------
def lumina_context(P, h_grids, w_grids, n_img_tokens, seed):
    """prompt(P text ids) + <start> h w + n_img_tokens image/EOL tokens laid out per the grammar."""
    ids = synthetic.synthetic_prompt(P, seed, lo=8900, hi=9200)[0].tolist()
    ids += [8197, 8804 + h_grids, 8804 + w_grids]
    w = 2 * w_grids
    g = torch.Generator().manual_seed(seed + 1)
    k = 0
    while k < n_img_tokens:
        if (k + 1) % (w + 1) == 0:
            ids.append(8803)
        else:
            ids.append(int(torch.randint(4, 8196, (1,), generator=g)))
        k += 1
    return torch.tensor([ids], dtype=torch.long)


def gen_fn_logits2tokens_lumina():
    V, L = 9216, 16
    cases = []
    # (name, P, h_grids, w_grids, n_img_tokens, n_rows, text_mode)
    spec = [
        ("mid_row", 12, 4, 4, 11, 16, 0),       # window crosses one EOL
        ("row_start", 12, 4, 4, 9, 5, 0),       # short window, no EOL
        ("two_eol", 12, 4, 2, 3, 16, 0),        # w=4: several EOLs in one window
        ("end_of_image", 12, 2, 2, 12, 16, 0),  # window reaches the end-of-image slot (h=w=4: 20 tokens)
        ("after_start_0", 12, 4, 4, -3, 4, 0),  # context ends with <start> only (new_token_num=0 <2)
        ("after_start_1", 12, 4, 4, -2, 4, 0),  # <start> h
        ("after_start_2", 12, 4, 4, -1, 4, 0),  # <start> h w  (new_token_num=2)
        ("text_mode", 12, 4, 4, -4, 3, 1),      # no image open: identity grammar + text top-k 10
        ("single_row", 12, 4, 4, 20, 1, 0),
    ]
    out = {}
    meta = []
    for ci, (name, P, hg, wg, nimg, nrows, text_mode) in enumerate(spec):
        ctx = lumina_context(P, hg, wg, max(nimg, 0), seed=100 + ci)
        if nimg < 0:
            ctx = ctx[:, : P + 3 + (nimg + 1)] if nimg > -4 else ctx[:, :P]
        g = torch.Generator().manual_seed(1000 + ci)
        logits = torch.randn(2, nrows, V, generator=g) * 3.0
        proc = LogitsProcessorList([
            LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196,
                                            image_next_line_token_id=8803, patch_size=32, voc_size=V),
            LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10,
                                                      image_start_token_id=8197, image_end_token_id=8196),
        ])
        gen = torch.Generator().manual_seed(2000 + ci)
        is_force_no_cfg = JL.check_is_force_no_cfg(ctx, 8197, 8196)
        toks, probs = JL.sampling_logits2tokens(
            logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
            logits_processor=proc, logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
            do_cfg=True, guidance_scale=3.0, generator=gen, is_force_no_cfg=is_force_no_cfg)
        cols = sample_cols(V)
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_at_tok"] = probs[0].gather(-1, toks[0][:, None])[:, 0].numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, V=V, nrows=nrows, logits_seed=1000 + ci, noise_seed=2000 + ci,
                         logits_scale=3.0, guidance_scale=3.0, image_top_k=2000, text_top_k=10,
                         is_force_no_cfg=bool(is_force_no_cfg)))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_logits2tokens_lumina.npz"), **out)
    print("fn_logits2tokens_lumina ok", [m["name"] for m in meta])


def gen_fn_greedy():
    """sampling_logits2tokens(do_sample=False) (JL:127-129): argmax of the processed scores, no warper, probs = softmax(scores)"""
    V = 9216
    spec = [("mid_row", 12, 4, 4, 11, 16), ("two_eol", 12, 4, 2, 3, 16), ("end_of_image", 12, 2, 2, 12, 16), ("single_row", 12, 4, 4, 20, 1)]
    out, meta = {}, []
    for ci, (name, P, hg, wg, nimg, nrows) in enumerate(spec):
        ctx = lumina_context(P, hg, wg, nimg, seed=300 + ci)
        logits = torch.randn(2, nrows, V, generator=torch.Generator().manual_seed(3000 + ci)) * 3.0
        proc = LogitsProcessorList([LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196,
                                                                    image_next_line_token_id=8803, patch_size=32, voc_size=V)])
        warp = LogitsProcessorList([LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10, image_start_token_id=8197,
                                                                              image_end_token_id=8196)])     # must be IGNORED when not sampling
        no_cfg = JL.check_is_force_no_cfg(ctx, 8197, 8196)
        toks, probs = JL.sampling_logits2tokens(logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows, logits_processor=proc,
                                                logits_warper=warp, do_sample=False, has_eos_stopping_criteria=False, do_cfg=True,
                                                guidance_scale=3.0, generator=None, is_force_no_cfg=no_cfg)
        cols = sample_cols(V)
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, V=V, nrows=nrows, logits_seed=3000 + ci, logits_scale=3.0, guidance_scale=3.0, is_force_no_cfg=bool(no_cfg)))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_greedy.npz"), **out)
    print("fn_greedy ok", [m["name"] for m in meta])


def gen_fn_processor_calls():
    """The processors called DIRECTLY on score tensors, one at a time (LP:84-155, 190-204, 406-455; JE:41-151): which entries stay finite, and
    the values there -- what the mirror's descriptor objects must reproduce when a user calls them outside generate()."""
    from transformers.generation.logits_process import TemperatureLogitsWarper
    import scheduler.jacobi_iteration_emu3 as JE
    from emu3.mllm.utils_emu3 import Emu3PrefixConstrainedLogitsHelper
    V = 9216
    out, meta = {}, []
    cols = sample_cols(V)

    def put(name, ctx, scores, res, **kw):
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.finite"] = np.packbits((res > torch.finfo(res.dtype).min).numpy().reshape(-1))      # kept entries (-inf and finfo.min both mean "removed")
        out[f"{name}.vals"] = res[..., cols].numpy()
        out[f"{name}.rowmax"] = res.max(-1).values.numpy()
        meta.append(dict(name=name, shape=list(scores.shape), **kw))

    spec = [("mid_row", 12, 4, 4, 11, 16), ("two_eol", 12, 4, 2, 3, 16), ("end_of_image", 12, 2, 2, 12, 16), ("after_start_1", 12, 4, 4, -2, 4),
            ("text_mode", 12, 4, 4, -4, 3), ("two_dim", 12, 4, 4, 7, 1)]
    for ci, (name, P, hg, wg, nimg, nrows) in enumerate(spec):
        ctx = lumina_context(P, hg, wg, max(nimg, 0), seed=400 + ci)
        if nimg < 0:
            ctx = ctx[:, : P + 3 + (nimg + 1)] if nimg > -4 else ctx[:, :P]
        scores = torch.randn(1, nrows, V, generator=torch.Generator().manual_seed(4000 + ci)) * 3.0
        if name == "two_dim":
            scores = scores[:, 0]
        vl = LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196, image_next_line_token_id=8803, patch_size=32, voc_size=V)
        put("vl_" + name, ctx, scores, vl(ctx, scores.clone()), kind="vl", seed=4000 + ci, scale=3.0)
        tk = LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10, image_start_token_id=8197, image_end_token_id=8196)
        put("tk_" + name, ctx, scores, tk(ctx, scores.clone()), kind="tk", seed=4000 + ci, scale=3.0)
        tp = LP.TopPLogitsWarper3d(top_p=0.9)
        put("tp_" + name, ctx, scores, tp(ctx, scores.clone()), kind="tp", seed=4000 + ci, scale=3.0, top_p=0.9)
        tm = TemperatureLogitsWarper(0.7)
        put("tm_" + name, ctx, scores, tm(ctx, scores.clone()), kind="tm", seed=4000 + ci, scale=3.0, temperature=0.7)
    V2 = 12288
    cols2 = sample_cols(V2)
    vis_lo, vis_n = 3000, 8192
    tok = dict(img_token=200, eoi_token=201, eos_token=202, eol_token=203, eof_token=204, pad_token=205)
    H, W = 3, 5
    for ci, (n_after_img, nrows) in enumerate([(0, 16), (4, 16), (17, 8), (19, 6)]):
        helper = Emu3PrefixConstrainedLogitsHelper(H, W, visual_tokens=list(range(vis_lo, vis_lo + vis_n)), **tok)
        helper.__class__ = JE.renew_end_of_line_logit_processor_3d(helper.__class__)
        g = torch.Generator().manual_seed(4500 + ci)
        ctx = torch.cat([torch.randint(300, 2000, (1, 9), generator=g), torch.tensor([[tok["img_token"]]]),
                         torch.randint(vis_lo, vis_lo + vis_n, (1, n_after_img), generator=g)], dim=1)
        scores = torch.randn(1, nrows, V2, generator=g) * 3.0
        res = helper(ctx, scores.clone())
        name = f"emu3_c{ci}"
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.finite"] = np.packbits((res > torch.finfo(res.dtype).min).numpy().reshape(-1))      # kept entries (-inf and finfo.min both mean "removed")
        out[f"{name}.vals"] = res[..., cols2].numpy()
        out[f"{name}.rowmax"] = res.max(-1).values.numpy()
        meta.append(dict(name=name, kind="emu3", shape=list(scores.shape), seed=4500 + ci, scale=3.0, H=H, W=W, vis_lo=vis_lo, vis_n=vis_n,
                         n_after_img=n_after_img, **tok))
    # Anole's single-purpose processors (LP:207-353), each alone, on [1, L, V] and [1, V] scores
    img = list(range(4, 8196))
    an = [("at_fire", lambda: LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5), [9000, 8197, 5, 6, 7, 8], 4),
          ("at_idle", lambda: LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5), [9000, 9001, 5, 6, 7, 8], 4),
          ("at_excl_idle", lambda: LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5, exclusive=True), [9000, 9001, 5, 6, 7, 8], 3),
          ("at_excl_short", lambda: LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 9, exclusive=True), [9000, 8197, 5], 2),
          ("at_short", lambda: LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 9), [9000, 8197, 5], 2),
          ("win_fire", lambda: LP.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 4), [9000, 9001, 8197, 5, 6], 5),
          ("win_idle", lambda: LP.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 2), [9000, 8197, 5, 6, 7], 5),
          ("win_excl_idle", lambda: LP.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 2, exclusive=True), [9000, 8197, 5, 6, 7], 2),
          ("rng_on", lambda: LP.SuppressTokensInIndexRangeLogitsProcessor3d([8196, 8197], 3, 6), [1, 2, 3, 4], 3),
          ("rng_off", lambda: LP.SuppressTokensInIndexRangeLogitsProcessor3d([8196, 8197], 6, 9), [1, 2, 3, 4], 3),
          ("begin_on", lambda: LP.SuppressTokensAtBeginLogitsProcessor3d([2], 4), [1, 2, 3, 4], 2),
          ("begin_off", lambda: LP.SuppressTokensAtBeginLogitsProcessor3d([2], 4), [1, 2, 3, 4, 5, 6], 2),
          ("supp", lambda: LP.SuppressTokensLogitsProcessor3d(img + [8196, 8197]), [1, 2, 3], 1)]
    for ci, (name, mk, ctx_l, nrows) in enumerate(an):
        ctx = torch.tensor([ctx_l])
        scores = torch.randn(1, nrows, V, generator=torch.Generator().manual_seed(4800 + ci)) * 3.0
        if nrows == 1:
            scores = scores[:, 0]
        put("an_" + name, ctx, scores, mk()(ctx, scores.clone()), kind="anole", seed=4800 + ci, scale=3.0, case=name)
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = cols.numpy()
    out["cols2"] = cols2.numpy()
    np.savez_compressed(os.path.join(HERE, "fn_processor_calls.npz"), **out)
    print("fn_processor_calls ok", len(meta))


def gen_fn_logits2tokens_llamagen():
    from llamagen.llamagen_solver import LlamaGenSolver
    V = 16384
    out, meta = {}, []
    for ci, (nrows, top_k, top_p, cfg) in enumerate([(16, 1000, 1.0, 4.0), (16, 1000, 0.9, 4.0), (1, 50, 0.5, 7.5),
                                                     (7, 16384, 1.0, 1.5)]):
        solver = LlamaGenSolver(model=None, image_top_k=top_k, image_top_p=top_p)
        proc = solver.create_logits_processor()
        g = torch.Generator().manual_seed(3000 + ci)
        logits = torch.randn(2, nrows, V, generator=g) * 3.0
        gen = torch.Generator().manual_seed(4000 + ci)
        ctx = torch.randint(0, V, (1, 5 + ci), generator=g)
        toks, probs = JL.sampling_logits2tokens(
            logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
            logits_processor=proc, logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
            do_cfg=True, guidance_scale=cfg, generator=gen, is_force_no_cfg=False)
        cols = sample_cols(V)
        name = f"c{ci}"
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, V=V, nrows=nrows, logits_seed=3000 + ci, noise_seed=4000 + ci, logits_scale=3.0,
                         guidance_scale=cfg, top_k=top_k, top_p=top_p))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_logits2tokens_llamagen.npz"), **out)
    print("fn_logits2tokens_llamagen ok")


def gen_fn_emu3_grammar():
    import scheduler.jacobi_iteration_emu3 as JE
    from emu3.mllm.utils_emu3 import Emu3PrefixConstrainedLogitsHelper
    V = 12288
    vis_lo, vis_n = 3000, 8192  # a contiguous visual-token range, like Emu3's 32768 codes
    tok = dict(img_token=200, eoi_token=201, eos_token=202, eol_token=203, eof_token=204, pad_token=205)
    out, meta = {}, []
    H, W = 3, 5  # (W+1)*H = 18 tokens, then EOF EOI EOS pad...
    for ci, (n_after_img, nrows) in enumerate([(0, 16), (4, 16), (10, 16), (17, 8), (2, 3), (19, 6)]):
        helper = Emu3PrefixConstrainedLogitsHelper(H, W, visual_tokens=list(range(vis_lo, vis_lo + vis_n)), **tok)
        helper.__class__ = JE.renew_end_of_line_logit_processor_3d(helper.__class__)
        g = torch.Generator().manual_seed(5000 + ci)
        ctx = torch.cat([torch.randint(300, 2000, (1, 9), generator=g), torch.tensor([[tok["img_token"]]]),
                         torch.randint(vis_lo, vis_lo + vis_n, (1, n_after_img), generator=g)], dim=1)
        logits = torch.randn(2, nrows, V, generator=g) * 3.0
        proc = LogitsProcessorList([helper, TopKLogitsWarper(top_k=2048)])
        gen = torch.Generator().manual_seed(6000 + ci)
        toks, probs = JL.sampling_logits2tokens(
            logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
            logits_processor=proc, logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
            do_cfg=True, guidance_scale=3.0, generator=gen, is_force_no_cfg=False)
        name = f"c{ci}"
        cols = sample_cols(V)
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, V=V, nrows=nrows, H=H, W=W, vis_lo=vis_lo, vis_n=vis_n, logits_seed=5000 + ci,
                         noise_seed=6000 + ci, ctx_seed=5000 + ci, n_after_img=n_after_img, top_k=2048,
                         guidance_scale=3.0, logits_scale=3.0, **tok))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_emu3_grammar.npz"), **out)
    print("fn_emu3_grammar ok")


def gen_fn_anole_grammar():
    V = 9216
    img_ids = list(range(4, 8196))
    boi, eoi, eos = 8197, 8196, 2
    image_seq_length = 24
    out, meta = {}, []
    # context lengths relative to the BOI token: before any BOI, right after BOI, mid image, at the EOI slot
    for ci, (n_after_boi, nrows, prompt_len, max_length) in enumerate(
            [(-1, 4, 6, 40), (0, 16, 6, 40), (10, 16, 6, 40), (24, 4, 6, 40), (23, 16, 6, 40)]):
        g = torch.Generator().manual_seed(7000 + ci)
        ctx = torch.randint(8900, 9200, (1, prompt_len), generator=g)
        if n_after_boi >= 0:
            ctx = torch.cat([ctx, torch.tensor([[boi]]), torch.randint(4, 8196, (1, n_after_boi), generator=g)], dim=1)
        allowed = img_ids + [eos, boi, eoi]
        suppress = [t for t in range(V) if t not in set(allowed)]
        procs = LogitsProcessorList([
            LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(trigger_token_id=boi, allowed_token_ids=[eoi],
                                                                offset=image_seq_length + 1, exclusive=True),
            LP.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(trigger_token_id=boi, allowed_token_ids=img_ids,
                                                                window_width=image_seq_length, exclusive=True),
            LP.SuppressTokensInIndexRangeLogitsProcessor3d(suppress_tokens=[boi],
                                                           start_index=max_length - image_seq_length - 1),
            LP.SuppressTokensLogitsProcessor3d(suppress_tokens=suppress),
            LP.SuppressTokensAtBeginLogitsProcessor3d(begin_suppress_tokens=[eos], begin_index=prompt_len),
            TopKLogitsWarper(top_k=2000),
        ])
        logits = torch.randn(2, nrows, V, generator=g) * 3.0
        gen = torch.Generator().manual_seed(8000 + ci)
        toks, probs = JL.sampling_logits2tokens(
            logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
            logits_processor=procs, logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
            do_cfg=True, guidance_scale=3.0, generator=gen, is_force_no_cfg=False)
        name = f"c{ci}"
        cols = sample_cols(V)
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, V=V, nrows=nrows, prompt_len=prompt_len, max_length=max_length,
                         image_seq_length=image_seq_length, boi=boi, eoi=eoi, eos=eos, logits_seed=7000 + ci,
                         noise_seed=8000 + ci, top_k=2000, guidance_scale=3.0, logits_scale=3.0,
                         n_after_boi=n_after_boi))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_anole_grammar.npz"), **out)
    print("fn_anole_grammar ok")


def gen_fn_anole_modes():
    """The two other restricted modes of the Anole pipeline, processors exactly as JA:178-189 ("text-only") and JA:233-260
    ("interleaved-text-image") build them (+ the TopKLogitsWarper HF's generate appends), through the reference's sampling_logits2tokens."""
    V = 9216
    img_ids = list(range(4, 8196))
    boi, eoi, eos = 8197, 8196, 2
    L = 24
    out, meta = {}, []
    ci = 0
    for mode in ("text-only", "interleaved-text-image"):
        # contexts: no <boi> yet / <boi> is the last token / mid image / the <eoi> slot / one token after the image / no room for an image
        for n_after_boi, tail, nrows, prompt_len, max_length in [(-1, 0, 4, 6, 60), (0, 0, 16, 6, 60), (10, 0, 16, 6, 60), (24, 0, 4, 6, 60),
                                                                  (24, 2, 16, 6, 60), (-1, 0, 8, 40, 60)]:
            g = torch.Generator().manual_seed(7100 + ci)
            ctx = torch.randint(8900, 9200, (1, prompt_len), generator=g)
            if n_after_boi >= 0:
                ctx = torch.cat([ctx, torch.tensor([[boi]]), torch.randint(4, 8196, (1, n_after_boi), generator=g)], dim=1)
            if tail:
                ctx = torch.cat([ctx, torch.tensor([[eoi]]), torch.randint(8900, 9200, (1, tail - 1), generator=g)], dim=1)
            if mode == "text-only":
                procs = [LP.SuppressTokensLogitsProcessor3d(suppress_tokens=img_ids + [boi, eoi])]
            else:
                procs = [LP.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(trigger_token_id=boi, allowed_token_ids=[eoi], offset=L + 1, exclusive=True),
                         LP.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(trigger_token_id=boi, allowed_token_ids=img_ids, window_width=L, exclusive=True),
                         LP.SuppressTokensInIndexRangeLogitsProcessor3d(suppress_tokens=[boi], start_index=max_length - L - 1)]
            procs = LogitsProcessorList(procs + [TopKLogitsWarper(top_k=50)])
            logits = torch.randn(2, nrows, V, generator=g) * 3.0
            gen = torch.Generator().manual_seed(8100 + ci)
            toks, probs = JL.sampling_logits2tokens(
                logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
                logits_processor=procs, logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
                do_cfg=True, guidance_scale=3.0, generator=gen, is_force_no_cfg=False)
            name = f"m{ci}"
            cols = sample_cols(V)
            out[f"{name}.ctx"] = ctx.numpy()
            out[f"{name}.tokens"] = toks.numpy()
            out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
            out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
            out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
            meta.append(dict(name=name, mode=mode, V=V, nrows=nrows, prompt_len=prompt_len, max_length=max_length, image_seq_length=L,
                             boi=boi, eoi=eoi, eos=eos, logits_seed=7100 + ci, noise_seed=8100 + ci, top_k=50, guidance_scale=3.0,
                             logits_scale=3.0, n_after_boi=n_after_boi, tail=tail))
            ci += 1
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = sample_cols(V).numpy()
    np.savez_compressed(os.path.join(HERE, "fn_anole_modes.npz"), **out)
    print("fn_anole_modes ok")


def make_pq(V, L, seed, mode):
    """Draft rows q / target rows p shaped like real SJD iterations: top-k'd softmaxes; some q rows one-hot."""
    g = torch.Generator().manual_seed(seed)
    zl = torch.randn(L, V, generator=g) * 3.0
    # q[i] is what position i was drafted from; it is verified against p[i-1] (JL:153-155), so build it
    # as a perturbation of the target row i-1.
    zq = torch.roll(zl, 1, 0) + torch.randn(L, V, generator=g) * {"far": 3.0, "carried": 0.15}.get(mode, 0.3)

    def topk_softmax(z, k):
        kth = torch.topk(z, k)[0][..., -1, None]
        return torch.softmax(z.masked_fill(z < kth, -float("inf")), dim=-1)

    p = topk_softmax(zl, 500)
    q = topk_softmax(zq, 500)
    draft = torch.multinomial(q, 1, generator=g)[:, 0]
    n_onehot = {"carried": 0, "mixed": 5, "fresh": L - 1, "far": 3, "equal": 0}[mode]
    if mode == "equal":   # q rows == shifted p rows: ratio exactly 1 -> everything accepted
        q[1:] = p[:-1]
        draft[1:] = torch.multinomial(p[:-1], 1, generator=g)[:, 0]
    for i in range(L - n_onehot, L):
        t = int(torch.randint(0, V, (1,), generator=g))
        # bias fresh random drafts into p's support half of the time so both branches occur
        if i % 2 == 0:
            t = int(torch.multinomial(p[i - 1], 1, generator=g))
        q[i] = 0
        q[i, t] = 1.0
        draft[i] = t
    return p[None], q[None], draft[None]


def gen_fn_speculative_sampler():
    V = 9216
    out, meta = {}, []
    cases = [("carried", 16, None), ("mixed", 16, None), ("fresh", 16, None), ("far", 16, None), ("equal", 8, None),
             ("mixed", 16, "lumina"), ("fresh", 2, None), ("mixed", 16, "llamagen")]
    for ci, (mode, L, grammar) in enumerate(cases):
        p, q, draft = make_pq(V, L, 9000 + ci, mode)
        adv_tokens = torch.multinomial(p[0], 1, generator=torch.Generator().manual_seed(9500 + ci))[:, 0][None]
        gen = torch.Generator().manual_seed(9900 + ci)
        B = 1
        sampler = JL.SpeculativeSampler(
            generator=gen,
            reject_sampling_relative_ids=-torch.ones(B, dtype=torch.long),
            reject_sampling_draft_token_logits=torch.zeros((B, V), dtype=torch.long),
            sampling_last_draft_token=torch.zeros((B,), dtype=torch.long))
        proc, ctx = None, torch.randint(8900, 9200, (1, 7), generator=torch.Generator().manual_seed(ci))
        if grammar == "lumina":
            ctx = lumina_context(12, 4, 4, 3, seed=300 + ci)   # residual rows may land on an EOL slot
            proc = LogitsProcessorList([
                LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196,
                                                image_next_line_token_id=8803, patch_size=32, voc_size=V),
                LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10,
                                                          image_start_token_id=8197, image_end_token_id=8196)])
        elif grammar == "llamagen":
            from llamagen.llamagen_solver import LlamaGenSolver
            proc = LlamaGenSolver(None, 100, 1.0).create_logits_processor()
        inds, toks, scores = sampler(draft_tokens=draft, advanced_tokens=adv_tokens.clone(), draft_prob=q,
                                     advanced_prob=p, logits_processor=proc, logits_warper=None,
                                     all_collected_input_ids=ctx)
        name = f"c{ci}"
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.first_misaligned"] = np.array(inds)
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.adv_tokens"] = adv_tokens.numpy()
        out[f"{name}.draft"] = draft.numpy()
        meta.append(dict(name=name, V=V, L=L, mode=mode, grammar=grammar, pq_seed=9000 + ci, adv_seed=9500 + ci,
                         noise_seed=9900 + ci))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "fn_speculative_sampler.npz"), **out)
    print("fn_speculative_sampler ok", [(m["mode"], int(out[m["name"] + ".first_misaligned"][0])) for m in meta])


def gen_fn_temperature():
    """HF's TemperatureLogitsWarper in the processor list, as transformers' generate() leaves it there when
    GenerationConfig.temperature != 1 (behind the user's processors): sampling_logits2tokens (JL:82-132) and the residual call of
    SpeculativeSampler (JL:203-241), which runs the same list on log(max(p - q, 0))."""
    from transformers.generation.logits_process import TemperatureLogitsWarper
    V = 9216
    out, meta = {}, []

    def procs(T):
        return LogitsProcessorList([
            LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196,
                                            image_next_line_token_id=8803, patch_size=32, voc_size=V),
            LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10,
                                                      image_start_token_id=8197, image_end_token_id=8196),
            TemperatureLogitsWarper(T)])

    cols = sample_cols(V)
    for ci, (T, nrows, nimg) in enumerate([(0.7, 16, 11), (1.5, 16, 11), (0.35, 5, 9), (2.5, 16, 3)]):
        ctx = lumina_context(12, 4, 4, nimg, seed=400 + ci)
        logits = torch.randn(2, nrows, V, generator=torch.Generator().manual_seed(4000 + ci)) * 3.0
        gen = torch.Generator().manual_seed(4100 + ci)
        toks, probs = JL.sampling_logits2tokens(
            logits, ctx, torch.ones(1, dtype=torch.long), None, output_token_num=nrows,
            logits_processor=procs(T), logits_warper=None, do_sample=True, has_eos_stopping_criteria=False,
            do_cfg=True, guidance_scale=3.0, generator=gen, is_force_no_cfg=False)
        name = f"s{ci}"
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.nnz"] = (probs[0] > 0).sum(-1).numpy()
        out[f"{name}.pmax"] = probs[0].max(-1).values.numpy()
        out[f"{name}.p_at_tok"] = probs[0].gather(-1, toks[0][:, None])[:, 0].numpy()
        out[f"{name}.p_cols"] = probs[0][:, cols].numpy()
        meta.append(dict(name=name, kind="sample", V=V, nrows=nrows, temperature=T, logits_seed=4000 + ci, noise_seed=4100 + ci,
                         logits_scale=3.0, guidance_scale=3.0, image_top_k=2000, text_top_k=10))
    for ci, (T, mode, L) in enumerate([(0.7, "mixed", 16), (1.5, "far", 16), (0.5, "fresh", 16), (2.0, "mixed", 16)]):
        p, q, draft = make_pq(V, L, 9100 + ci, mode)
        adv_tokens = torch.multinomial(p[0], 1, generator=torch.Generator().manual_seed(9600 + ci))[:, 0][None]
        gen = torch.Generator().manual_seed(9950 + ci)
        sampler = JL.SpeculativeSampler(
            generator=gen, reject_sampling_relative_ids=-torch.ones(1, dtype=torch.long),
            reject_sampling_draft_token_logits=torch.zeros((1, V), dtype=torch.long),
            sampling_last_draft_token=torch.zeros((1,), dtype=torch.long))
        ctx = lumina_context(12, 4, 4, 3, seed=500 + ci)
        inds, toks, scores = sampler(draft_tokens=draft, advanced_tokens=adv_tokens.clone(), draft_prob=q, advanced_prob=p,
                                     logits_processor=procs(T), logits_warper=None, all_collected_input_ids=ctx)
        name = f"v{ci}"
        fm = int(inds[0])
        out[f"{name}.ctx"] = ctx.numpy()
        out[f"{name}.first_misaligned"] = np.array(inds)
        out[f"{name}.tokens"] = toks.numpy()
        out[f"{name}.adv_tokens"] = adv_tokens.numpy()
        out[f"{name}.draft"] = draft.numpy()
        if fm < L:       # the resampled row's distribution (scores[:, fm-1] is softmax(processors(log(max(p - q, 0)))))
            out[f"{name}.resid_p_cols"] = scores[0, fm - 1][cols].numpy()
            out[f"{name}.resid_nnz"] = np.array(int((scores[0, fm - 1] > 0).sum()))
        meta.append(dict(name=name, kind="verify", V=V, L=L, mode=mode, temperature=T, pq_seed=9100 + ci, adv_seed=9600 + ci,
                         noise_seed=9950 + ci))
    out["meta"] = np.array(json.dumps(meta))
    out["cols"] = cols.numpy()
    np.savez_compressed(os.path.join(HERE, "fn_temperature.npz"), **out)
    print("fn_temperature ok", [(m["name"], m["temperature"]) for m in meta],
          [int(out[m["name"] + ".first_misaligned"][0]) for m in meta if m["kind"] == "verify"])


def gen_fn_reguess():
    out, meta = {}, []
    img_vocab = torch.arange(4, 8196)
    for ci, n in enumerate([15, 7, 0, 1]):
        torch.manual_seed(1234 + ci)   # the reference draws from the GLOBAL CPU generator (JL:505)
        ids = torch.zeros(1, 9, dtype=torch.long)
        tcs = torch.zeros(1, 2, 9216)
        toks, scores = JL.get_multi_token_for_preparation(img_vocab, n, ids, tcs, "cpu", multi_token_init_scheme="random")
        out[f"c{ci}.tokens"] = toks.numpy()
        assert scores.shape == (1, n, 9216) and float(scores.sum()) == n
        if n:
            assert bool((scores[0].argmax(-1) == toks[0]).all())
        meta.append(dict(name=f"c{ci}", n=n, global_seed=1234 + ci))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "fn_reguess.npz"), **out)
    print("fn_reguess ok")


# ----------------------------------------------------------------------------------------------
# whole-loop traces
# ----------------------------------------------------------------------------------------------
class Tracer:
    """Wraps the reference's module-level hot-path functions to log per-iteration integers."""

    def __init__(self):
        self.windows, self.sampled, self.matched, self.final = [], [], [], []

    def install(self):
        self._pm, self._s2t = JL.prefix_matching_next_tokens, JL.sampling_logits2tokens
        tr = self

        def s2t(*a, **k):
            toks, probs = tr._s2t(*a, **k)
            tr.sampled.append(toks[0].tolist())
            return toks, probs

        def pm(model_input_ids, next_tokens, next_token_scores, **k):
            r = tr._pm(model_input_ids, next_tokens, next_token_scores, **k)
            tr.windows.append(model_input_ids[0].tolist())
            tr.matched.append(int(r[0]))
            tr.final.append(r[1][0].tolist())
            return r

        JL.sampling_logits2tokens, JL.prefix_matching_next_tokens = s2t, pm

    def remove(self):
        JL.sampling_logits2tokens, JL.prefix_matching_next_tokens = self._s2t, self._pm

    def pack(self, prefix, out):
        def ragged(lst):
            flat = np.array([x for r in lst for x in r], dtype=np.int64)
            offs = np.cumsum([0] + [len(r) for r in lst]).astype(np.int64)
            return flat, offs
        for key in ("windows", "sampled", "final"):
            f, o = ragged(getattr(self, key))
            out[f"{prefix}.{key}"], out[f"{prefix}.{key}_offs"] = f, o
        out[f"{prefix}.matched"] = np.array(self.matched, dtype=np.int64)


def gen_loop_llamagen():
    from llamagen.llamagen import Transformer, ModelArgs
    from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen
    out, meta = {}, []
    # (name, scheme, seed, class_id, cfg, top_k, top_p, window, latent, embed_token_scale)
    runs = [("spec_s7", "speculative_jacobi", 7, 207, 4.0, 1000, 1.0, 16, 16, 0.25),
            ("spec_s11_w8", "speculative_jacobi", 11, 3, 2.0, 200, 1.0, 8, 16, 0.5),
            ("jacobi_s7", "jacobi", 7, 207, 4.0, 1000, 1.0, 16, 16, 0.25),
            ("spec_s5_topp", "speculative_jacobi", 5, 42, 4.0, 1000, 0.95, 16, 8, 0.25),
            ("spec_s13_floor", "speculative_jacobi", 13, 999, 4.0, 1000, 1.0, 16, 8, 1.0)]
    for name, scheme, seed, class_id, cfg, top_k, top_p, window, latent, ets in runs:
        args = dict(dim=64, n_layer=2, n_head=4, vocab_size=16384, block_size=latent * latent, cls_token_num=1,
                    model_type="c2i", num_classes=1000, token_dropout_p=0.0, attn_dropout_p=0.0, resid_dropout_p=0.0,
                    ffn_dropout_p=0.0, drop_path_rate=0.0)
        model = Transformer(ModelArgs(**args)).eval()
        synthetic.fill_state_dict(model, seed=17, embed_token_scale=ets)
        jac = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=latent * latent - window - 2,
                   max_num_new_tokens=window, guidance_scale=cfg, seed=seed, multi_token_init_scheme="random",
                   do_cfg=True, image_top_k=top_k, text_top_k=10, prefix_token_sampler_scheme=scheme,
                   use_chameleon_tokenizer=False)
        model.config.is_encoder_decoder = False
        model.__class__ = renew_llamagen(model.__class__)
        model._init_new_params(**jac)
        model.__class__ = type("M", (CompatMixin, JL.renew_sampler(model.__class__)), {})
        model._init_new_params(**jac)
        model.img_vocab = torch.arange(4, 8196)
        solver = LlamaGenSolver(model=model, image_top_k=top_k, image_top_p=top_p)
        tr = Tracer()
        tr.install()
        torch.manual_seed(seed)   # prefill() samples from the GLOBAL generator (LS:81)
        try:
            toks = solver.generate(torch.tensor([class_id]), latent * latent, None, cfg_scale=cfg, temperature=1.0,
                                   top_k=top_k, top_p=top_p, sample_logits=True)
        finally:
            tr.remove()
        out[f"{name}.tokens"] = toks.numpy()
        tr.pack(name, out)
        meta.append(dict(name=name, model_args=args, weight_seed=17, embed_token_scale=ets, jacobi=jac,
                         class_id=class_id, cfg=cfg,
                         top_k=top_k, top_p=top_p, latent=latent, nfe=len(tr.matched)))
        print("loop_llamagen", name, "tokens", toks.shape, "NFE", len(tr.matched), "first", toks[0, :8].tolist())
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "loop_llamagen.npz"), **out)


def gen_loop_lumina(greedy=False):
    """greedy: GenerationConfig(do_sample=False) -- the argmax branch of sampling_logits2tokens (JL:127-129) inside the whole loop: no multinomial
    draw, the verify step's uniform and residual draws as before -> loop_lumina_greedy.npz"""
    from model.chameleon import ChameleonForConditionalGeneration, ChameleonConfig
    from transformers import GenerationConfig
    from transformers.generation.stopping_criteria import StoppingCriteriaList, EosTokenCriteria, MaxLengthCriteria
    out, meta = {}, []
    # (name, scheme, seed, hg, wg, window, l, r, P, kvh, embed_token_scale)
    runs = [("spec_s3", "speculative_jacobi", 3, 4, 4, 16, 3, 8 * 9 - 10, 12, 4, 0.25),
            ("spec_s9_w8_gqa", "speculative_jacobi", 9, 3, 5, 8, 3, 6 * 11 - 6, 20, 2, 0.5),
            ("jacobi_s3", "jacobi", 3, 4, 4, 16, 3, 8 * 9 - 10, 12, 4, 0.25),
            ("spec_s4_tail", "speculative_jacobi", 4, 4, 4, 16, 1, 8 * 9 + 1, 12, 4, 0.25)]
    if greedy:
        runs = [("greedy_s3", "speculative_jacobi", 3, 4, 4, 16, 3, 8 * 9 - 10, 12, 4, 0.25),
                ("greedy_s6_w8", "speculative_jacobi", 6, 3, 5, 8, 3, 6 * 11 - 6, 20, 2, 0.5),
                ("greedy_jacobi_s3", "jacobi", 3, 4, 4, 16, 3, 8 * 9 - 10, 12, 4, 0.25)]
    for name, scheme, seed, hg, wg, window, l, r, P, kvh, ets in runs:
        V = 9216
        cfg_kw = dict(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=kvh, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0,
                      swin_norm=False, mask_image_logits=False, vocabulary_map={"<image>": 8711},
                      vq_config=dict(embed_dim=32, num_embeddings=64, double_latent=False, latent_channels=32,
                                     resolution=32, in_channels=3, base_channels=32, channel_multiplier=[1, 1],
                                     num_res_blocks=1, attn_resolutions=None, dropout=0.0, attn_type="vanilla"),
                      attn_implementation="sdpa")
        cfg = ChameleonConfig(**cfg_kw)
        cfg.rope_scaling = None
        model = ChameleonForConditionalGeneration(cfg).eval()
        synthetic.fill_state_dict(model, seed=23, skip_prefixes=("model.vqmodel.",), embed_token_scale=ets)
        jac = dict(jacobi_loop_interval_l=l, jacobi_loop_interval_r=r, max_num_new_tokens=window, guidance_scale=3.0,
                   seed=seed, multi_token_init_scheme="random", do_cfg=True, image_top_k=2000, text_top_k=10,
                   prefix_token_sampler_scheme=scheme, use_chameleon_tokenizer=False)
        model.__class__ = type("M", (CompatMixin, JL.renew_sampler(model.__class__)), {})
        model._init_new_params(**jac)
        model.img_vocab = torch.arange(4, 8196)
        model.model.__class__ = JL.renew_backbone(model.model.__class__)
        procs = LogitsProcessorList([
            LP.MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196,
                                            image_next_line_token_id=8803, patch_size=32, voc_size=V),
            LP.MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10,
                                                      image_start_token_id=8197, image_end_token_id=8196)])
        prompt = torch.cat([synthetic.synthetic_prompt(P - 3, seed, lo=8900, hi=9200),
                            torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
        n_img = (2 * wg + 1) * 2 * hg
        max_len = P + n_img + 1 + 4
        stopping = StoppingCriteriaList([EosTokenCriteria([8196]), MaxLengthCriteria(max_len)])
        gc = GenerationConfig(max_length=max_len, do_sample=not greedy, temperature=1.0, top_k=None)
        gc._pad_token_tensor = torch.tensor(0)
        tr = Tracer()
        tr.install()
        try:
            seq = model._sample(input_ids=prompt, logits_processor=procs, stopping_criteria=stopping,
                                generation_config=gc, synced_gpus=False, streamer=None,
                                attention_mask=torch.ones_like(prompt), past_key_values=LegacyCache(), use_cache=True)
        finally:
            tr.remove()
        out[f"{name}.prompt"] = prompt.numpy()
        out[f"{name}.sequence"] = seq.numpy()
        tr.pack(name, out)
        cfg_kw.pop("attn_implementation")
        meta.append(dict(name=name, config=cfg_kw, weight_seed=23, embed_token_scale=ets, jacobi=jac, P=P, hg=hg,
                         wg=wg, max_len=max_len,
                         nfe=len(tr.matched)))
        gen = seq[0, P:].tolist()
        print("loop_lumina", name, "generated", len(gen), "NFE", len(tr.matched), "tail", gen[-4:],
              "eol@", [i for i, t in enumerate(gen) if t == 8803][:4])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "loop_lumina_greedy.npz" if greedy else "loop_lumina.npz"), **out)


def gen_loop_lumina_ar():
    """The reference's AUTOREGRESSIVE baseline (the denominator of its published speed-ups): FlexARInferenceSolver.create_logits_processor's three
    processors (IS:16-270, built as IS:417-450 builds them) around HF generate() on the same tiny Chameleon as loop_lumina.npz, the model class NOT
    renewed.  One token per forward; the unconditional branch is a second forward inside the CFG processor (IS:59-93) on the context from the
    image-start token on.  -> loop_lumina_ar.npz (prompt, sequence).  The multinomial of HF's _sample draws from the GLOBAL generator."""
    from model.chameleon import ChameleonForConditionalGeneration, ChameleonConfig
    from transformers import GenerationConfig
    sys.path.insert(0, "/root/reference/lumina_mgpt")
    import inference_solver as IS
    assert IS.__file__.startswith("/root/reference/"), IS.__file__
    out, meta = {}, []
    # (name, seed, hg, wg, P, kvh, embed_token_scale, guidance)
    runs = [("ar_s3", 3, 4, 4, 12, 4, 0.25, 3.0), ("ar_s9_gqa", 9, 3, 5, 20, 2, 0.5, 3.0), ("ar_s5_g1", 5, 3, 3, 9, 4, 0.25, 1.0)]
    for name, seed, hg, wg, P, kvh, ets, guidance in runs:
        V = 9216
        cfg_kw = dict(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=kvh, max_position_embeddings=512, rms_norm_eps=1e-5, rope_theta=10000.0,
                      swin_norm=False, mask_image_logits=False, vocabulary_map={"<image>": 8711},
                      vq_config=dict(embed_dim=32, num_embeddings=64, double_latent=False, latent_channels=32,
                                     resolution=32, in_channels=3, base_channels=32, channel_multiplier=[1, 1],
                                     num_res_blocks=1, attn_resolutions=None, dropout=0.0, attn_type="vanilla"),
                      attn_implementation="sdpa")
        cfg = ChameleonConfig(**cfg_kw)
        cfg.rope_scaling = None
        model = ChameleonForConditionalGeneration(cfg).eval()
        synthetic.fill_state_dict(model, seed=23, skip_prefixes=("model.vqmodel.",), embed_token_scale=ets)
        ids = dict(image_start_token_id=8197, image_end_token_id=8196)
        procs = LogitsProcessorList([
            IS.LLMImageStartTriggeredUnbatchedClassifierFreeGuidanceLogitsProcessor(guidance_scale=guidance, model=model, image_next_line_token_id=8803,
                                                                                   patch_size=32, **ids),
            IS.MultiModalLogitsProcessor(image_next_line_token_id=8803, patch_size=32, voc_size=V, device="cpu", **ids),
            IS.InterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10, **ids)])
        prompt = torch.cat([synthetic.synthetic_prompt(P - 3, seed, lo=8900, hi=9200),
                            torch.tensor([[8197, 8804 + hg, 8804 + wg]])], dim=1)
        n_img = (2 * wg + 1) * 2 * hg
        max_new = n_img + 1 + 4                  # the image, its end token, four text tokens behind it (text top-k 10, no CFG)
        gc = GenerationConfig(max_new_tokens=max_new, max_length=P + max_new, temperature=1.0, top_k=None, do_sample=True, eos_token_id=[8710],
                              pad_token_id=0)                                    # IS:335-345
        # HF generate() of the installed transformers cannot drive the reference's modeling code (written against 4.4x: no GenerationMixin on the
        # model class, another cache API), so the sampling loop of GenerationMixin._sample is spelled out here -- per step: the model on the new
        # token with the cache (called the way the reference's own CFG processor calls it, IS:86-91), logits[:, -1].float(), the processor list,
        # softmax, torch.multinomial(probs, 1) on the GLOBAL generator, stop at an EOS id or at max_length.  Model, processors and cache handling
        # are the reference's; both caches are the shim's legacy cache.
        procs[0].unconditional_context_backup["past_key_values"] = LegacyCache()
        torch.manual_seed(seed)
        seq, cache, cur = prompt.clone(), LegacyCache(), prompt
        while seq.shape[1] < gc.max_length:
            o = model(cur, attention_mask=torch.ones_like(seq), use_cache=True, past_key_values=cache)
            cache = o.get("past_key_values", cache)
            scores = procs(seq, o.logits[:, -1, :].clone().float())
            nxt = torch.multinomial(torch.softmax(scores, dim=-1), num_samples=1)
            seq, cur = torch.cat([seq, nxt], dim=1), nxt
            if int(nxt) in gc.eos_token_id:
                break
        out[f"{name}.prompt"] = prompt.numpy()
        out[f"{name}.sequence"] = seq.numpy()
        cfg_kw.pop("attn_implementation")
        meta.append(dict(name=name, config=cfg_kw, weight_seed=23, embed_token_scale=ets, seed=seed, guidance_scale=guidance, P=P, hg=hg, wg=wg,
                         max_new_tokens=max_new, image_top_k=2000, text_top_k=10))
        gen = seq[0, P:].tolist()
        print("loop_lumina_ar", name, "generated", len(gen), "tail", gen[-6:], "eol@", [i for i, t in enumerate(gen) if t == 8803][:4])
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "loop_lumina_ar.npz"), **out)


def gen_vq_decoders():
    """Reference decoders at small widths with weights that depend only on (state-dict key, shape): the fixture keeps the key/shape
    list (the test asserts the rewrite has exactly these), the codes and the decoded images."""
    import torch.nn as nn
    from llamagen.tokenizer.tokenizer_image import vq_model as LV
    out, meta = {}, {}

    class LG(nn.Module):               # the decode side of LV.VQModel with a configurable base width
        def __init__(self):
            super().__init__()
            self.quantize = LV.VectorQuantizer(96, 8, 0.25, 0.0, True, False)
            self.post_quant_conv = nn.Conv2d(8, 32, 1)
            self.decoder = LV.Decoder(z_channels=32, ch=32, ch_mult=(1, 2, 2))
    lg = synthetic.fill_state_dict_conv(LG().eval(), seed=11)
    codes = torch.randint(0, 96, (2 * 5 * 6,), generator=torch.Generator().manual_seed(1))
    img = lg.decoder(lg.post_quant_conv(lg.quantize.get_codebook_entry(codes, (2, 8, 5, 6), True)))
    out["llamagen_codes"], out["llamagen_image"] = codes.numpy(), img.numpy()
    meta["llamagen"] = dict(keys={k: list(v.shape) for k, v in lg.state_dict().items()}, seed=11, shape=[2, 8, 5, 6],
                            kwargs=dict(codebook_size=96, codebook_embed_dim=8, z_channels=32, ch=32, ch_mult=[1, 2, 2]))

    sys.path.insert(0, "/root/reference/lumina_mgpt")
    from model.chameleon_vae_ori import vqgan as CV
    dd = dict(double_z=False, z_channels=32, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2], num_res_blocks=2,
              attn_resolutions=[8], dropout=0.0)

    class CH(nn.Module):
        def __init__(self):
            super().__init__()
            self.quantize = CV.VectorQuantizer(80, 16, beta=0.25)
            self.post_quant_conv = nn.Conv2d(16, 32, 1)
            self.decoder = CV.Decoder(**dd)
    ch = synthetic.fill_state_dict_conv(CH().eval(), seed=12)
    codes = torch.randint(0, 80, (8 * 8,), generator=torch.Generator().manual_seed(2))
    img = ch.decoder(ch.post_quant_conv(ch.quantize.get_codebook_entry(codes, (1, 8, 8, 16))))
    out["chameleon_codes"], out["chameleon_image"] = codes.numpy(), img.numpy()
    meta["chameleon"] = dict(keys={k: list(v.shape) for k, v in ch.state_dict().items()}, seed=12,
                             kwargs=dict(n_embed=80, embed_dim=16, z_channels=32, ch=32, ch_mult=[1, 2, 2], num_res_blocks=2,
                                         attn_resolutions=[8], resolution=32))
    from emu3.tokenizer.modeling_emu3visionvq import Emu3VisionVQModel
    from emu3.tokenizer.configuration_emu3visionvq import Emu3VisionVQConfig
    ekw = dict(codebook_size=64, embed_dim=4, z_channels=4, ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[1])
    em = Emu3VisionVQModel(Emu3VisionVQConfig(**ekw)).eval()
    synthetic.fill_state_dict_conv(em, seed=13)
    codes = torch.randint(0, 64, (2, 4, 5), generator=torch.Generator().manual_seed(3))
    vcodes = torch.randint(0, 64, (1, 2, 3, 4), generator=torch.Generator().manual_seed(4))
    out["emu3_codes"], out["emu3_image"] = codes.numpy(), em.decode(codes).numpy()
    out["emu3_video_codes"], out["emu3_video"] = vcodes.numpy(), em.decode(vcodes).numpy()
    dec_keys = {k: list(v.shape) for k, v in em.state_dict().items() if not k.startswith(("encoder.", "quant_conv."))}
    meta["emu3"] = dict(keys=dec_keys, seed=13, kwargs=ekw)
    np.savez_compressed(os.path.join(HERE, "vq_decoders.npz"), meta=json.dumps(meta), **out)
    print("vq_decoders.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["fn", "loops", "vq"]
    if "vq" in which:
        gen_vq_decoders()
    if "fn" in which:
        gen_fn_logits2tokens_lumina()
        gen_fn_logits2tokens_llamagen()
        gen_fn_emu3_grammar()
        gen_fn_anole_grammar()
        gen_fn_speculative_sampler()
        gen_fn_reguess()
    if "fn" in which or "greedy" in which:
        gen_fn_greedy()
    if "fn" in which or "calls" in which:
        gen_fn_processor_calls()
    if "fn" in which or "temp" in which:
        gen_fn_temperature()
    if "fn" in which or "anole_modes" in which:
        gen_fn_anole_modes()
    if "loops" in which:
        gen_loop_llamagen()
        gen_loop_lumina()
    if "loops" in which or "greedy_loops" in which:
        gen_loop_lumina(greedy=True)
    if "loops" in which or "ar_loops" in which:
        gen_loop_lumina_ar()
