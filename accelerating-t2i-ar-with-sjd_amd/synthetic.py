"""Deterministic synthetic weights / prompts (no checkpoints reach the GPU box).

Every tensor is drawn from its own CPU ``torch.Generator`` seeded by
``(seed, crc32(name))`` so the values depend only on the tensor's state-dict
name and shape -- not on module construction order.  The golden-fixture
generator (tests/golden/make_golden.py) loads these values into the *reference*
models; the tests load the same values into this repo's backbones, which share
the reference checkpoints' state-dict keys.
"""
import zlib

import torch


def _gen_for(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    return g


def synthetic_tensor(name: str, shape, seed: int = 0, head_gain: float = 3.0,
                     embed_token_scale: float = 1.0) -> torch.Tensor:
    """fp32 CPU tensor for state-dict entry ``name``.

    * 1-D ``...norm...weight`` / ``weight`` of norms -> 1 + 0.1*N(0,1);  biases -> 0.02*N(0,1)
    * embeddings -> c + embed_token_scale*N(0,1) with one shared N(0,1) row ``c`` (if scale < 1).  A small
      ``embed_token_scale`` makes the next-token distribution depend only weakly on the previous token, which
      is what gives Speculative Jacobi Decoding its acceptance rate on real image models; with 1.0 (plain
      random weights) acceptance sits at the floor of ~1 token/step.
    * 2-D linear weights [out,in] -> N(0, 1/in); the LM head gets ``head_gain``/sqrt(in) so that
      logits have std ~= head_gain (a peaked, "alive" distribution instead of near-uniform).
    """
    shape = tuple(int(s) for s in shape)
    g = _gen_for(name, seed)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    lname = name.lower()
    if lname.endswith("bias"):
        return 0.02 * x
    if len(shape) == 1 or "norm" in lname:
        return 1.0 + 0.1 * x
    if "embed" in lname:  # tok_embeddings / embed_tokens / embedding_table
        if embed_token_scale >= 1.0:
            return x
        common = torch.randn(shape[-1:], generator=_gen_for(name + "#common", seed), dtype=torch.float32)
        return common + embed_token_scale * x
    fan_in = shape[-1]
    gain = head_gain if (lname.startswith("output.") or lname.startswith("lm_head.")) else 1.0
    return x * (gain / fan_in ** 0.5)


def fill_state_dict(module: torch.nn.Module, seed: int = 0, skip_prefixes=(), head_gain: float = 3.0,
                    embed_token_scale: float = 1.0):
    """In-place deterministic fill of every parameter/buffer tensor in ``module.state_dict()``."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if any(name.startswith(p) for p in skip_prefixes):
                continue
            if not torch.is_floating_point(t):
                continue
            t.copy_(synthetic_tensor(name, t.shape, seed, head_gain, embed_token_scale).to(t.dtype))
    return module


def fill_state_dict_conv(module: torch.nn.Module, seed: int = 0):
    """Deterministic per-key fill for convolutional nets (the VQ decoders): weights N(0, 1/fan_in) with fan_in = prod(shape[1:]),
    GroupNorm gains 1 + 0.1 N(0,1), biases 0.02 N(0,1), codebooks N(0,1).  Depends only on (key, shape, seed)."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not torch.is_floating_point(t):
                continue
            x = torch.randn(tuple(t.shape), generator=_gen_for(name, seed), dtype=torch.float32)
            if name.endswith("bias"):
                x = 0.02 * x
            elif t.dim() == 1:
                x = 1.0 + 0.1 * x
            elif "embedding" not in name:
                fan_in = 1
                for d in t.shape[1:]:
                    fan_in *= int(d)
                x = x / fan_in ** 0.5
            t.copy_(x.to(t.dtype))
    return module


def synthetic_prompt(length: int, seed: int, lo: int = 8900, hi: int = 60000) -> torch.Tensor:
    """Prompt token ids uniform in a text-id range (never grammar tokens)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.randint(lo, hi, (1, length), generator=g, dtype=torch.long)


def device_tensor_fill(name: str, t: torch.Tensor, seed: int = 0, head_gain: float = 3.0, embed_token_scale: float = 1.0):
    """In-place fill of ONE state-dict tensor on its own device (the per-tensor rule of fill_state_dict_device; the values depend
    only on (name, shape, seed, embed_token_scale), so a single tensor -- e.g. the embedding table -- can be re-drawn alone)."""
    g = torch.Generator(device=t.device)
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2**63 - 1))
    lname = name.lower()
    x = torch.randn(t.shape, generator=g, device=t.device, dtype=torch.float32)
    if lname.endswith("bias"):
        x = 0.02 * x
    elif t.dim() == 1 or "norm" in lname:
        x = 1.0 + 0.1 * x
    elif "embed" in lname:
        if embed_token_scale < 1.0:
            common = torch.randn(t.shape[-1:], generator=g, device=t.device, dtype=torch.float32)
            x = common + embed_token_scale * x
    else:
        gain = head_gain if (lname.startswith("output.") or lname.startswith("lm_head.")) else 1.0
        x = x * (gain / t.shape[-1] ** 0.5)
    t.copy_(x.to(t.dtype))


def fill_state_dict_device(module: torch.nn.Module, seed: int = 0, head_gain: float = 3.0, embed_token_scale: float = 1.0):
    """Same distribution as fill_state_dict but drawn ON the tensor's device (fast for 7B-parameter benches).
    Values differ from the CPU variant; use fill_state_dict when CPU/GPU weight equality matters."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if torch.is_floating_point(t):
                device_tensor_fill(name, t, seed, head_gain, embed_token_scale)
    return module


def refill_embeddings_device(module: torch.nn.Module, seed: int = 0, embed_token_scale: float = 1.0):
    """Re-draw only the token-embedding tables in place (same storage: captured hipGraphs stay valid).  embed_token_scale = 1.0 is
    the "floor" regime of SURVEY.md 8(d): plain random weights, ~1 accepted token per SJD step."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if torch.is_floating_point(t) and "embed" in name.lower() and t.dim() == 2:
                device_tensor_fill(name, t, seed, 3.0, embed_token_scale)
    return module
