"""The mirror's logits processors called DIRECTLY on score tensors (outside generate()), one at a time, against fixtures the reference's own
processor objects produced (tests/golden/make_golden.py::gen_fn_processor_calls; LP:84-155, 190-204, 406-455, JE:41-151): the same entries stay
finite, with the same values.  CPU: the direct-call form is ATen ops over the rules the kernels apply (logit_processor_3dim.apply_row_rules).
The library is loaded for the rule structs only (no compute call)."""
import json
import os

import numpy as np
import pytest
import torch


def _load(golden_dir):
    d = np.load(os.path.join(golden_dir, "fn_processor_calls.npz"))
    return d, json.loads(str(d["meta"]))


def _check(d, m, res, cols):
    name = m["name"]
    fin = np.unpackbits(d[f"{name}.finite"])[:res.numel()].astype(bool).reshape(tuple(res.shape))
    got = (res > torch.finfo(res.dtype).min).numpy()          # kept entries: the Lumina processors remove with -inf, Emu3's with finfo.min (JE:80)
    assert (got == fin).all(), (name, int((got != fin).sum()))
    np.testing.assert_array_equal(res[..., cols].numpy(), d[f"{name}.vals"], err_msg=name)
    np.testing.assert_array_equal(res.max(-1).values.numpy(), d[f"{name}.rowmax"], err_msg=name)


def test_processors_called_directly_match_the_reference(golden_dir):
    from scheduler.logit_processor_3dim import (MultiTokensVLLogitsProcessor, MultiTokensInterleavedTopKLogitsWarper, TopPLogitsWarper3d,
                                                TemperatureLogitsWarper)
    from scheduler.jacobi_iteration_emu3 import renew_end_of_line_logit_processor_3d
    d, meta = _load(golden_dir)
    n = 0
    for m in meta:
        ctx = torch.from_numpy(d[f"{m['name']}.ctx"])
        if m["kind"] == "emu3":
            from emu3.mllm.utils_emu3 import Emu3PrefixConstrainedLogitsHelper
            g = torch.Generator().manual_seed(m["seed"])
            torch.randint(300, 2000, (1, 9), generator=g)
            torch.randint(m["vis_lo"], m["vis_lo"] + m["vis_n"], (1, m["n_after_img"]), generator=g)
            scores = torch.randn(*m["shape"], generator=g) * m["scale"]
            tok = {k: m[k] for k in ("img_token", "eoi_token", "eos_token", "eol_token", "eof_token", "pad_token")}
            helper = Emu3PrefixConstrainedLogitsHelper(m["H"], m["W"], visual_tokens=list(range(m["vis_lo"], m["vis_lo"] + m["vis_n"])), **tok)
            helper.__class__ = renew_end_of_line_logit_processor_3d(helper.__class__)
            _check(d, m, helper(ctx, scores.clone()), d["cols2"])
            n += 1
            continue
        if m["kind"] == "anole":
            from scheduler import logit_processor_3dim as M
            img = list(range(4, 8196))
            mk = {"at_fire": lambda: M.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5),
                  "at_idle": lambda: M.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5),
                  "at_excl_idle": lambda: M.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 5, exclusive=True),
                  "at_excl_short": lambda: M.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 9, exclusive=True),
                  "at_short": lambda: M.AllowOnlyTokensAtRelativeOffsetLogitsProcessor3d(8197, [8196], 9),
                  "win_fire": lambda: M.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 4),
                  "win_idle": lambda: M.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 2),
                  "win_excl_idle": lambda: M.AllowOnlyTokensInRelativeWindowLogitsProcessor3d(8197, img, 2, exclusive=True),
                  "rng_on": lambda: M.SuppressTokensInIndexRangeLogitsProcessor3d([8196, 8197], 3, 6),
                  "rng_off": lambda: M.SuppressTokensInIndexRangeLogitsProcessor3d([8196, 8197], 6, 9),
                  "begin_on": lambda: M.SuppressTokensAtBeginLogitsProcessor3d([2], 4),
                  "begin_off": lambda: M.SuppressTokensAtBeginLogitsProcessor3d([2], 4),
                  "supp": lambda: M.SuppressTokensLogitsProcessor3d(img + [8196, 8197])}[m["case"]]
            shape = m["shape"]
            full = torch.randn(1, shape[-2] if len(shape) == 3 else 1, shape[-1], generator=torch.Generator().manual_seed(m["seed"])) * m["scale"]
            scores = full if len(shape) == 3 else full[:, 0]
            _check(d, m, mk()(ctx, scores.clone()), d["cols"])
            n += 1
            continue
        shape = m["shape"]
        full = torch.randn(1, shape[-2] if len(shape) == 3 else 1, shape[-1], generator=torch.Generator().manual_seed(m["seed"])) * m["scale"]
        scores = full if len(shape) == 3 else full[:, 0]
        if m["kind"] == "vl":
            proc = MultiTokensVLLogitsProcessor(image_start_token_id=8197, image_end_token_id=8196, image_next_line_token_id=8803, patch_size=32,
                                                voc_size=shape[-1])
        elif m["kind"] == "tk":
            proc = MultiTokensInterleavedTopKLogitsWarper(image_top_k=2000, text_top_k=10, image_start_token_id=8197, image_end_token_id=8196)
        elif m["kind"] == "tp":
            proc = TopPLogitsWarper3d(top_p=m["top_p"])
        else:
            proc = TemperatureLogitsWarper(m["temperature"])
        res = proc(ctx, scores.clone())
        assert res.shape == scores.shape
        if m["kind"] == "tm":          # a division: the same fp32 operation
            np.testing.assert_array_equal(res[..., d["cols"]].numpy(), d[f"{m['name']}.vals"])
        else:
            _check(d, m, res, d["cols"])
        n += 1
    assert n == len(meta) == 41


