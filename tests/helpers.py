"""Shared test helpers: seeded inputs identical to tests/golden/make_golden.py, tiny backbones, and the
forward callbacks the oracle loop drives."""
import numpy as np
import torch

import sjd_amd.backbones as BB
import sjd_amd.synthetic as synthetic


def make_pq(V, L, seed, mode):
    """Same construction as make_golden.make_pq (draft rows q / target rows p; some q rows one-hot)."""
    g = torch.Generator().manual_seed(seed)
    zl = torch.randn(L, V, generator=g) * 3.0
    zq = torch.roll(zl, 1, 0) + torch.randn(L, V, generator=g) * {"far": 3.0, "carried": 0.15}.get(mode, 0.3)

    def topk_softmax(z, k):
        kth = torch.topk(z, k)[0][..., -1, None]
        return torch.softmax(z.masked_fill(z < kth, -float("inf")), dim=-1)

    p = topk_softmax(zl, 500)
    q = topk_softmax(zq, 500)
    draft = torch.multinomial(q, 1, generator=g)[:, 0]
    n_onehot = {"carried": 0, "mixed": 5, "fresh": L - 1, "far": 3, "equal": 0}[mode]
    if mode == "equal":
        q[1:] = p[:-1]
        draft[1:] = torch.multinomial(p[:-1], 1, generator=g)[:, 0]
    for i in range(L - n_onehot, L):
        t = int(torch.randint(0, V, (1,), generator=g))
        if i % 2 == 0:
            t = int(torch.multinomial(p[i - 1], 1, generator=g))
        q[i] = 0
        q[i, t] = 1.0
        draft[i] = t
    return p[None], q[None], draft[None]


def make_llamagen(model_args, weight_seed, embed_token_scale, attn, dtype=torch.float32, device="cpu"):
    keys = {k: v for k, v in model_args.items() if k in BB.LlamaGenArgs.__dataclass_fields__}
    model = BB.LlamaGenBackbone(BB.LlamaGenArgs(**keys), attn=attn).eval()
    synthetic.fill_state_dict(model, seed=weight_seed, embed_token_scale=embed_token_scale)
    return model.to(device=device, dtype=dtype)


def make_chameleon(config, weight_seed, embed_token_scale, attn, dtype=torch.float32, device="cpu"):
    args = BB.ChameleonArgs(vocab_size=config["vocab_size"], hidden_size=config["hidden_size"],
                            intermediate_size=config["intermediate_size"],
                            num_hidden_layers=config["num_hidden_layers"],
                            num_attention_heads=config["num_attention_heads"],
                            num_key_value_heads=config["num_key_value_heads"], rms_norm_eps=config["rms_norm_eps"],
                            rope_theta=config["rope_theta"], qk_norm=True,
                            max_position_embeddings=config["max_position_embeddings"])
    model = BB.ChameleonBackbone(args, attn=attn).eval()
    synthetic.fill_state_dict(model, seed=weight_seed, embed_token_scale=embed_token_scale)
    return model.to(device=device, dtype=dtype)


def llamagen_prefill_sample(logits, cfg_scale, temperature, top_k, top_p):
    """First image token: reference prefill()+sample()+top_k_top_p_filtering (llamagen_solver.py:34-104);
    draws from the GLOBAL torch generator."""
    cond, uncond = torch.split(logits, len(logits) // 2, dim=0)
    lg = uncond + (cond - uncond) * cfg_scale if cfg_scale > 1.0 else logits
    lg = lg[:, -1, :] / max(temperature, 1e-5)
    if top_k > 0:
        k = min(max(top_k, 1), lg.size(-1))
        lg[lg < torch.topk(lg, k)[0][..., -1, None]] = -float("inf")
    if top_p < 1.0:
        sl, si = torch.sort(lg, descending=True)
        cp = torch.cumsum(torch.softmax(sl, dim=-1), dim=-1)
        rm = cp > top_p
        rm[..., 1:] = rm[..., :-1].clone()
        rm[..., 0] = 0
        lg[rm.scatter(1, si, rm)] = -float("inf")
    probs = torch.softmax(lg, dim=-1)
    return torch.multinomial(probs, num_samples=1)


@torch.no_grad()
def llamagen_forward_fn(model, class_id, cfg_scale, top_k, top_p, max_new_tokens, seed, device="cpu"):
    """Returns (forward_fn for the oracle loop, first image token).  Mirrors LlamaGenSolver.generate
    (llamagen_solver.py:371-456): cond||null-class prefill at cache row 0, then _sample from row T=1."""
    T = 1
    model.setup_cache(batch=2, s_max=T + max_new_tokens + 40)
    cond = torch.tensor([class_id, model.num_classes], device=device)
    torch.manual_seed(seed)
    zeros = torch.zeros(2, dtype=torch.long)
    logits = model.forward_embeds(model.embed_condition(cond), torch.zeros(2, 1, dtype=torch.long, device=device), 0, zeros)
    first = int(llamagen_prefill_sample(logits.cpu(), cfg_scale, 1.0, top_k, top_p)[0, 0])

    @torch.no_grad()
    def fwd(win, kv_len):
        n = len(win)
        toks = torch.tensor([win, win], dtype=torch.long, device=device)
        pos = (T + kv_len + torch.arange(n, device=device))[None].repeat(2, 1)
        lg = model.forward_window(toks, pos, T + kv_len, zeros)
        return lg[0].cpu().numpy(), lg[1].cpu().numpy()

    return fwd, first


@torch.no_grad()
def lumina_forward_fn(model, P, s_max, device="cpu", uncond_start=None):
    """cond||uncond batch; the uncond half is blind to prompt[:P-1] and its RoPE positions are shifted by -(P-1)
    (jacobi_iteration_lumina_mgpt.py:703-712, 755-758).
    uncond_start: index of the first prompt token the uncond half sees (default P - 1, the SJD sampler's; the reference's AUTOREGRESSIVE CFG
    processor keeps the context from the image-start token on, IS:62-63: P - 3 for a prompt that ends <image-start> h w)."""
    model.setup_cache(batch=2, s_max=s_max)
    if uncond_start is not None and uncond_start != P - 1:
        u0 = int(uncond_start)
        key_start = torch.tensor([0, u0])

        @torch.no_grad()
        def fwd_u(win, kv_len):
            n = len(win)
            toks = torch.tensor([win, win], dtype=torch.long, device=device)
            if kv_len == 0:
                assert n == P
                pos = torch.stack([torch.arange(P), torch.tensor([1] * u0 + list(range(P - u0)))]).to(device)
            else:
                base = kv_len + torch.arange(n)
                pos = torch.stack([base, base - u0]).to(device)
            lg = model.forward_window(toks, pos, kv_len, key_start)
            return lg[0].cpu().numpy(), lg[1].cpu().numpy()

        return fwd_u
    key_start = torch.tensor([0, P - 1])

    @torch.no_grad()
    def fwd(win, kv_len):
        n = len(win)
        toks = torch.tensor([win, win], dtype=torch.long, device=device)
        if kv_len == 0:
            assert n == P
            pos = torch.stack([torch.arange(P), torch.tensor([1] * (P - 1) + [0])]).to(device)
        else:
            base = kv_len + torch.arange(n)
            pos = torch.stack([base, base - (P - 1)]).to(device)
        lg = model.forward_window(toks, pos, kv_len, key_start)
        return lg[0].cpu().numpy(), lg[1].cpu().numpy()

    return fwd


class Emu3StubTokenizer:
    """stand-in for the hub tokenizer: only the special-token attributes and `encode` of single tokens are used"""
    bos_token, boi_token, img_token, eoi_token, eos_token, eol_token, eof_token, pad_token = "<b>", "<boi>", "<img>", "<eoi>", "<eos>", "<eol>", "<eof>", "<pad>"
    table = {"<b>": 1, "<boi>": 2, "<img>": 200, "<eoi>": 201, "<eos>": 202, "<eol>": 203, "<eof>": 204, "<pad>": 205}

    def encode(self, s):
        if s in self.table:
            return [self.table[s]]
        if s.startswith("<|visual token "):
            return [3000 + int(s[len("<|visual token "):-2])]
        out, i = [], 0
        while i < len(s):                               # greedy special tokens, every other character = one text id
            hit = next((t for t in self.table if s.startswith(t, i)), None)
            out.append(self.table[hit] if hit else 300 + (ord(s[i]) % 1000))
            i += len(hit) if hit else 1
        return out
