#!/usr/bin/env python3
"""PyTorch-ROCm SJD baseline: the reference's data flow restated with plain ATen ops on the GPU (no hand kernels).

BASELINE.md section 3 item 2: "the same restatement with device tensors on 1 MI355X ... the >=2x target is measured against
this".  It follows the op inventory of SURVEY.md 2.3 for one iteration of JacobiSampler._sample:
   window assembly with torch.cat + one-hot scatter rows (JL:505-514, 656-701), DynamicCache-style torch.cat of the
   whole K/V per layer (MC:547) + slice rollback (JL:47-54), additive 4-D mask rebuilt every iteration (JL:1308-1324),
   F.scaled_dot_product_attention with that mask (MC:567), clone/chunk/CFG, torch.where grammar mask + forced rows by
   index_put (LP:125-145), torch.topk(k=2000) threshold (LP:196-204), softmax, torch.multinomial (JL:111-118),
   torch.rand([1,n,V]) + a Python accept loop with one device->host sync per draft (JL:260-311), residual
   clamp/log/softmax/multinomial (JL:203-241).
Same synthetic Lumina-mGPT-7B workload as bench.py.  Timing tool only -- not part of the product, not used by tests.
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


class TorchCatAttention:
    """HF DynamicCache + masked SDPA, as the reference's ChameleonSdpaAttention uses them."""

    def __init__(self):
        self.k, self.v = {}, {}

    def rollback(self, n_drop):
        if n_drop > 0:
            for li in self.k:
                self.k[li] = self.k[li][..., :-n_drop, :]
                self.v[li] = self.v[li][..., :-n_drop, :]

    def __call__(self, layer, q, k, v, cache, kv_len, key_start_mask):
        B, n, H, D = q.shape
        q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
        if layer in self.k:
            self.k[layer] = torch.cat([self.k[layer], k], dim=-2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=-2)
        else:
            self.k[layer], self.v[layer] = k, v
        K, V = self.k[layer], self.v[layer]
        o = F.scaled_dot_product_attention(q.contiguous(), K.contiguous(), V.contiguous(), attn_mask=key_start_mask[..., :K.shape[-2]])
        return o.transpose(1, 2)


def build_mask(mask01, n, dtype, device):
    """_update_causal_mask (JL:1308-1324): 0/1 mask [2, n, S+n] -> additive [2,1,n,S+n]."""
    min_dtype = torch.finfo(dtype).min
    target = mask01.shape[-1]
    causal = torch.full((n, target), fill_value=min_dtype, dtype=dtype, device=device)
    if n != 1:
        causal = torch.triu(causal, diagonal=1)
    cache_position = torch.arange(target - n, target, device=device)
    causal *= torch.arange(target, device=device) > cache_position.reshape(-1, 1)
    causal = causal[None, None, :, :].expand(mask01.shape[0], 1, -1, -1).clone()
    pad = (causal + mask01[:, None, :, :].to(dtype)) == 0
    return causal.masked_fill(pad, min_dtype)


@torch.no_grad()
def run_baseline(model, context, P, grid, steps, warmup, seed=1234, window=16, g_scale=3.0, dev=None):
    """One PyTorch-ROCm SJD decode with the reference's data flow on `model` (a ChameleonBackbone whose forward takes the ATen path:
    model._ops is None and model.attn is a TorchCatAttention).  context = prompt (P ids, ending <start> h w) + already accepted
    image ids: the first iteration prefills all of it, so the timed steps run at that KV length.  -> dict(ms_per_step, ...)."""
    dev = dev or next(model.parameters()).device
    attn = model.attn
    V, W = model.vocab_size, window
    torch.manual_seed(seed)
    gen = torch.Generator(dev).manual_seed(seed)
    img_vocab = torch.arange(4, 8196, device=dev)
    suppress = torch.ones(V, dtype=torch.bool, device=dev)
    suppress[4:8196] = False
    ids = torch.tensor([list(context)], device=dev)
    C = ids.shape[1]
    tcs = torch.zeros(1, 1, V, device=dev)
    mask01 = torch.ones(2, C, device=dev)
    mask01[1, :P - 1] = 0
    add_tok, add_sc = None, None
    n, cur_len, it = 1, C, 0
    l_abs, r_abs = P, P + grid * grid + grid - 13
    t0, tok0 = None, C
    while it < warmup + steps:
        if it == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tok0 = ids.shape[1]
        first = it == 0
        # ---- window assembly (JL:606-701)
        if first:
            win = ids
            q_rows = None
        else:
            a_n = 0 if add_tok is None else min(add_tok.shape[1], n - 1)
            n_fresh = n - 1 - a_n
            rand = img_vocab[torch.randint(0, 8192, (1, n_fresh)).to(dev)]
            onehot = torch.zeros(1, n_fresh, V, device=dev).scatter(-1, rand.unsqueeze(-1), 1.0)
            parts_t = [ids[:, -1:]] + ([add_tok[:, :a_n]] if a_n else []) + [rand]
            parts_s = [tcs[:, -1:]] + ([add_sc[:, :a_n]] if a_n else []) + [onehot]
            win = torch.cat(parts_t, dim=-1)
            q_rows = torch.cat(parts_s, dim=1)
        nw = win.shape[1]
        S = 0 if first else mask01.shape[-1]
        if first:
            m3 = torch.tril(torch.ones(nw, nw, device=dev))[None].repeat(2, 1, 1) * mask01[:, None, :]
            pos = (mask01.long().cumsum(-1) - 1).masked_fill(mask01 == 0, 1)
        else:
            m3 = torch.ones(2, nw, S + nw, device=dev)                 # JL:831-838
            m3[:, :, :S] = mask01[:, None, :]
            m3[:, :, S:] = torch.tril(m3[0, :, S:])
            pos = (m3[:, -1, :].long().cumsum(-1) - 1)[:, -nw:]        # JL:705-712
        addmask = build_mask(m3, nw, torch.bfloat16, dev)
        logits = model.forward_window(win.repeat(2, 1), pos, 0, addmask)
        # ---- sampling_logits2tokens (JL:82-132) with the Lumina processors (LP:84-204)
        rows = 1 if first else n
        z = logits[:, -rows:, :].clone()
        c, u = z.chunk(2, dim=0)
        z = g_scale * (c - u) + u
        T = ids.shape[1] - P                       # image tokens so far (the prompt ends with <start> h w)
        n_start = (ids[0] == 8197).sum()
        n_end = (ids[0] == 8196).sum()
        if n_start == n_end + 1:                   # a sync, as in the reference
            z = torch.where(suppress, -float("inf"), z)
            for j in range(rows):
                if (T + 1 + j) % (grid + 1) == 0:
                    z[..., j, :] = -float("inf")
                    z[..., j, 8803] = 0
                if (T + 1 + j) % ((grid + 1) * grid + 1) == 0:
                    z[..., j, :] = -float("inf")
                    z[..., j, 8196] = 0
        kth = torch.topk(z, 2000)[0][..., -1, None]
        z = z.masked_fill(z < kth, -float("inf"))
        probs = torch.softmax(z, dim=-1)
        Y = torch.multinomial(probs.flatten(0, 1), 1, generator=gen).squeeze(1)[None]
        # ---- prefix matching (JL:247-376)
        if rows <= 1:
            m = nw
            emitted, tail_t, tail_s, keep = Y[:, -1:], None, None, probs[:, -1:]
        else:
            rs = torch.rand(probs.shape, device=dev, generator=gen)
            Yc, Pc = Y.clone(), probs.clone()
            m = rows
            for i in range(1, rows):
                x = win[0, i]
                ratio = (probs[0, i - 1, x] / q_rows[0, i, x]).clamp(max=1)
                if rs[0, i, x] < ratio:            # device->host sync per draft
                    Yc[0, i - 1] = x
                    Pc[0, i - 1, :] = q_rows[0, i, :]
                else:
                    d = (probs[0, i - 1] - q_rows[0, i]).clamp(min=0).log()
                    d = torch.where(suppress, -float("inf"), d)
                    kth2 = torch.topk(d, 2000)[0][..., -1, None]
                    d = d.masked_fill(d < kth2, -float("inf"))
                    Yc[0, i - 1] = torch.multinomial(torch.softmax(d, -1)[None], 1, generator=gen)[0, 0]
                    m = i
                    break
            emitted, tail_t, tail_s, keep = Yc[:, :m], Yc[:, m:], probs[:, m:], Pc[:, :m]
        n = min(W, r_abs - cur_len) if (l_abs <= cur_len < r_abs) else 1
        ids = torch.cat([ids, emitted], dim=-1)
        tcs = torch.cat([tcs[:, -1:], keep], dim=1)
        attn.rollback(nw - m)
        if not first:
            mask01 = torch.cat([mask01, torch.ones(2, m, device=dev)], dim=-1)
        add_tok, add_sc = (tail_t, tail_s) if (rows > 1 and m < rows) else (None, None)
        cur_len = ids.shape[1]
        it += 1
        if int(ids[0, -1]) == 8196:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_steps = it - warmup
    toks = ids.shape[1] - tok0
    return dict(kind="pytorch_sjd_baseline", steps=n_steps, ms_per_step=round(dt / n_steps * 1e3, 3),
                tokens_per_step=round(toks / n_steps, 4), tokens_per_s=round(toks / dt, 2), kv_len_start=int(C), kv_len_end=int(mask01.shape[-1]))


def run_on_engine_model(model, context, P, grid, steps, warmup, seed=1234, window=16):
    """bench.py's `torch_baseline` leg: the SAME weights as the engine just used, forward through the plain ATen path
    (hipBLASLt GEMMs, ATen norms/RoPE, torch.cat KV cache, masked SDPA): the fused HIP path is switched off for the duration."""
    saved = (getattr(model, "_ops", None), model.attn)
    model._ops, model.attn = None, TorchCatAttention()
    try:
        return run_baseline(model, context, P, grid, steps, warmup, seed=seed, window=window)
    finally:
        model._ops, model.attn = saved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--embed-token-scale", type=float, default=0.7)
    ap.add_argument("--model", default="lumina7b")
    ap.add_argument("--image-prefix", type=int, default=0, help="already accepted image tokens prefilled before the timed steps (mid-image KV length)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import sjd_amd.backbones as BB
    import sjd_amd.synthetic as synthetic
    from sjd_amd.frontends import lumina_prompt
    margs = BB.LUMINA_7B if a.model == "lumina7b" else BB.ChameleonArgs(hidden_size=1024, intermediate_size=2048, num_hidden_layers=4,
                                                                       num_attention_heads=8, num_key_value_heads=8)
    with torch.device(dev):
        model = BB.ChameleonBackbone(margs, attn=TorchCatAttention()).to(torch.bfloat16).eval()
    synthetic.fill_state_dict_device(model, seed=0, embed_token_scale=a.embed_token_scale)
    P, grid = 64, 48
    context = lumina_prompt(P, grid, grid, seed=1234)
    g = torch.Generator().manual_seed(7)
    for t in range(a.image_prefix):                      # synthetic accepted prefix with the line tokens where the grammar puts them
        context.append(8803 if (t + 1) % (grid + 1) == 0 else 4 + int(torch.randint(0, 8192, (1,), generator=g)))
    print(json.dumps(run_baseline(model, context, P, grid, a.steps, a.warmup, dev=dev)))


if __name__ == "__main__":
    main()
