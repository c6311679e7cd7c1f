"""Image detokenizers: VQ token ids -> pixels, for the two VQGAN families of the reference.

Not part of the per-token hot path: one call per finished image, plain PyTorch-ROCm convolutions (MIOpen).  The two decoders
are the same network family -- 3x3 conv, a middle (res, attention, res), then per resolution level a run of residual blocks
(with single-head spatial attention at selected levels) and a nearest-neighbour 2x upsample + conv, GroupNorm(32) + swish in
front of every conv -- and differ in three places that `VQDecoderSpec` captures: how the sub-modules are NAMED in the checkpoint,
where attention sits, and how codes are looked up.  Module / parameter names follow the reference so that its checkpoints load
with `load_state_dict(strict=True)`:

  LlamaGen   llamagen/tokenizer/tokenizer_image/vq_model.py:28-62 (VQModel.decode_code), :128-194 (Decoder),
             :258-275 (get_codebook_entry: l2-normalised codebook rows)
  Chameleon  lumina_mgpt/model/chameleon_vae_ori/vqgan.py:410-529 (Decoder), :532-600 (VQModel.decode_code),
             :131-146 (get_codebook_entry) -- used by Lumina-mGPT and Anole through chameleon_vae_ori/image_tokenizer.py

Parity: tests/test_detokenizers.py loads per-key synthetic weights (the fixture lists the reference's state-dict keys and shapes)
and compares the decoded image with the one the imported reference produced (tests/golden/make_golden.py::gen_vq_decoders).
"""
from dataclasses import dataclass, field
from typing import Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


def _gn(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class _Res(nn.Module):
    """GroupNorm-swish-conv twice plus a skip (1x1 `nin_shortcut` when the width changes)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.norm1, self.conv1 = _gn(c_in), nn.Conv2d(c_in, c_out, 3, padding=1)
        self.norm2, self.conv2 = _gn(c_out), nn.Conv2d(c_out, c_out, 3, padding=1)
        if c_in != c_out:
            self.nin_shortcut = nn.Conv2d(c_in, c_out, 1)

    def forward(self, x):
        y = self.conv1(F.silu(self.norm1(x)))
        y = self.conv2(F.silu(self.norm2(y)))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + y


class _Attn(nn.Module):
    """single-head self-attention over the h*w positions (1x1 conv projections)"""

    def __init__(self, c):
        super().__init__()
        self.norm = _gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.norm(x)
        q, k, v = (m(y).flatten(2).transpose(1, 2) for m in (self.q, self.k, self.v))      # [b, hw, c]
        o = F.scaled_dot_product_attention(q, k, v)                                         # scale c^-0.5, softmax over keys
        return x + self.proj_out(o.transpose(1, 2).reshape(b, c, h, w))


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


@dataclass
class VQDecoderSpec:
    z_channels: int = 256
    ch: int = 128
    ch_mult: Sequence[int] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    out_channels: int = 3
    naming: str = "llamagen"                       # "llamagen": mid.{0,1,2}, conv_blocks.{i}.res / .attn (coarse level first)
    #                                                "taming":   mid.block_1 / attn_1 / block_2, up.{i}.block / .attn (fine level = 0)
    attn_resolutions: Tuple[int, ...] = ()         # taming: feature-map sizes that carry attention; llamagen: the coarsest level does
    resolution: int = 512                          # taming: output size (fixes the feature-map size of every level)


class VQDecoder(nn.Module):
    def __init__(self, spec: VQDecoderSpec):
        super().__init__()
        self.spec = spec
        n_lvl = len(spec.ch_mult)
        c = spec.ch * spec.ch_mult[-1]
        self.conv_in = nn.Conv2d(spec.z_channels, c, 3, padding=1)
        res = spec.resolution // 2 ** (n_lvl - 1)
        levels = []
        for lvl in reversed(range(n_lvl)):             # coarse -> fine
            c_out = spec.ch * spec.ch_mult[lvl]
            with_attn = (lvl == n_lvl - 1) if spec.naming == "llamagen" else (res in spec.attn_resolutions)
            blocks, attns = nn.ModuleList(), nn.ModuleList()
            for _ in range(spec.num_res_blocks + 1):
                blocks.append(_Res(c, c_out))
                c = c_out
                if with_attn:
                    attns.append(_Attn(c))
            level = nn.Module()
            if spec.naming == "llamagen":
                level.res, level.attn = blocks, attns
            else:
                level.block, level.attn = blocks, attns
            if lvl != 0:
                level.upsample = _Up(c)
                res *= 2
            levels.append(level)
        c_mid = spec.ch * spec.ch_mult[-1]
        if spec.naming == "llamagen":
            self.mid = nn.ModuleList([_Res(c_mid, c_mid), _Attn(c_mid), _Res(c_mid, c_mid)])
            self.conv_blocks = nn.ModuleList(levels)                       # index 0 = coarsest
        else:
            self.mid = nn.Module()
            self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _Res(c_mid, c_mid), _Attn(c_mid), _Res(c_mid, c_mid)
            self.up = nn.ModuleList(list(reversed(levels)))                # index 0 = finest
        self.norm_out = _gn(c)
        self.conv_out = nn.Conv2d(c, spec.out_channels, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        if self.spec.naming == "llamagen":
            for m in self.mid:
                h = m(h)
            levels = list(self.conv_blocks)
        else:
            h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
            levels = list(reversed(self.up))
        for level in levels:                           # coarse -> fine
            blocks = level.res if self.spec.naming == "llamagen" else level.block
            for i, blk in enumerate(blocks):
                h = blk(h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if hasattr(level, "upsample"):
                h = level.upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class _Codebook(nn.Module):
    def __init__(self, n_e, e_dim):
        super().__init__()
        self.embedding = nn.Embedding(n_e, e_dim)


def _decode_side(state_dict):
    """drop what only the encoder half of a VQGAN checkpoint needs (the reference loads the full model, strict)"""
    drop = ("encoder.", "quant_conv.", "loss.", "quantize.codebook_used")
    return {k: v for k, v in state_dict.items() if not k.startswith(drop)}


class LlamaGenVQ(nn.Module):
    """`vq_model.decode_code(index_sample, qzshape)` of the reference's test_llamagen.py: ids [b*h*w] (or [b, h*w]) + shape
    (b, e_dim, h, w) -> image [b, 3, 16h or 8h, ...] in [-1, 1].  Codebook rows are l2-normalised at lookup."""

    def __init__(self, codebook_size=16384, codebook_embed_dim=8, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), l2_norm=True):
        super().__init__()
        self.l2_norm = l2_norm
        self.quantize = _Codebook(codebook_size, codebook_embed_dim)
        self.post_quant_conv = nn.Conv2d(codebook_embed_dim, z_channels, 1)
        self.decoder = VQDecoder(VQDecoderSpec(z_channels=z_channels, ch=ch, ch_mult=tuple(ch_mult), naming="llamagen"))

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(_decode_side(state_dict), strict=strict, **kw)

    @torch.no_grad()
    def decode_code(self, code_b, shape=None, channel_first=True):
        w = self.quantize.embedding.weight
        if self.l2_norm:
            w = F.normalize(w, p=2, dim=-1)
        z = w[code_b]
        if shape is not None:
            z = z.reshape(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2).contiguous() if channel_first else z.view(shape)
        return self.decoder(self.post_quant_conv(z))


class ChameleonVQ(nn.Module):
    """The Chameleon / Lumina-mGPT / Anole image tokenizer's decode side: ids [b, h*w] (or [b, h, w]) -> image in [-1, 1]
    (reference ImageTokenizer.pil_from_img_toks -> VQModel.decode(quantize.get_codebook_entry(...)))."""

    def __init__(self, n_embed=8192, embed_dim=256, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2,
                 attn_resolutions=(), resolution=512, out_ch=3):
        super().__init__()
        self.embed_dim = embed_dim
        self.quantize = _Codebook(n_embed, embed_dim)
        self.post_quant_conv = nn.Conv2d(embed_dim, z_channels, 1)
        self.decoder = VQDecoder(VQDecoderSpec(z_channels=z_channels, ch=ch, ch_mult=tuple(ch_mult), num_res_blocks=num_res_blocks,
                                               out_channels=out_ch, naming="taming", attn_resolutions=tuple(attn_resolutions),
                                               resolution=resolution))

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(_decode_side(state_dict), strict=strict, **kw)

    @torch.no_grad()
    def pil_from_img_toks(self, tokens, h_latent_dim=32, w_latent_dim=32):
        """reference ImageTokenizer.pil_from_img_toks (image_tokenizer.py:117-121): flat VQ ids of one image -> PIL.Image"""
        from PIL import Image
        img = self.decode_code(torch.as_tensor(tokens).reshape(1, -1), hw=(h_latent_dim, w_latent_dim))
        return Image.fromarray(to_uint8(img, truncate=True)[0].cpu().numpy())

    @torch.no_grad()
    def decode_code(self, code_b, hw=None):
        code_b = torch.as_tensor(code_b)
        if code_b.dim() == 2:
            b, n = code_b.shape
            h, w = hw if hw is not None else (int(n ** 0.5), int(n ** 0.5))
            code_b = code_b.reshape(b, h, w)
        z = self.quantize.embedding(code_b).permute(0, 3, 1, 2).contiguous()          # [b, h, w, e] -> [b, e, h, w]
        return self.decoder(self.post_quant_conv(z))


def to_uint8(img, truncate=False):
    """[-1, 1] float image [b, 3, H, W] -> uint8 [b, H, W, 3].  truncate=True is the Chameleon tokenizer's conversion
    (image_tokenizer.py:95-106: (clamp(x) + 1) / 2 * 255 cast to uint8); the default rounds."""
    x = (img.float().clamp(-1, 1) + 1) * 127.5
    x = x.floor() if truncate else x.round()
    return x.to(torch.uint8).permute(0, 2, 3, 1).contiguous()
