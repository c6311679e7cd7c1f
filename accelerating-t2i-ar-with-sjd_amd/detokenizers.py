"""Image detokenizers: VQ token ids -> pixels, for the three VQGAN families of the reference.

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

  Emu3       emu3/tokenizer/modeling_emu3visionvq.py:596-721 (decoder), :790-812 (decode): the same decoder with a causal
             temporal 3-D-conv stack in front and norms modulated by the quantised latent (classes below the Chameleon one)

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


# ------------------------------------------------------------------------------------------------ Emu3 VisionVQ (decode side)
class _CausalConv3d(nn.Module):
    """3-D conv over [b, c, t, h, w]: "same" padding in h/w, two frames of left padding in t (never looks at later frames)"""

    def __init__(self, c_in, c_out, kernel=(3, 1, 1)):
        super().__init__()
        self.conv = nn.Conv3d(c_in, c_out, kernel)
        ph, pw = kernel[1] - 1, kernel[2] - 1
        self.pad = (pw // 2 + pw % 2, pw // 2, ph // 2 + ph % 2, ph // 2, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, self.pad))


class _TemporalRes(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm1, self.conv1 = nn.BatchNorm3d(c), _CausalConv3d(c, c, (3, 3, 3))
        self.norm2, self.conv2 = nn.BatchNorm3d(c), _CausalConv3d(c, c, (3, 3, 3))

    def forward(self, x):
        return x + self.conv2(F.silu(self.norm2(self.conv1(F.silu(self.norm1(x))))))


class _TemporalUp(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _CausalConv3d(c, c, (3, 3, 3))

    def forward(self, x):
        return self.conv(x.repeat_interleave(2, dim=2))             # nearest-neighbour x2 along t


class _SpatialNorm(nn.Module):
    """GroupNorm whose scale and shift are 1x1 convs of the (resized) quantised latent"""

    def __init__(self, c, zq_ch):
        super().__init__()
        self.norm_layer = _gn(c)
        self.conv_y, self.conv_b = nn.Conv2d(zq_ch, c, 1), nn.Conv2d(zq_ch, c, 1)

    def forward(self, x, zq):
        zq = F.interpolate(zq, size=x.shape[-2:], mode="nearest")
        return self.norm_layer(x) * self.conv_y(zq) + self.conv_b(zq)


class _ResZ(nn.Module):
    def __init__(self, c_in, c_out, zq_ch):
        super().__init__()
        self.norm1, self.conv1 = _SpatialNorm(c_in, zq_ch), nn.Conv2d(c_in, c_out, 3, padding=1)
        self.norm2, self.conv2 = _SpatialNorm(c_out, zq_ch), nn.Conv2d(c_out, c_out, 3, padding=1)
        if c_in != c_out:
            self.nin_shortcut = nn.Conv2d(c_in, c_out, 1)

    def forward(self, x, zq):
        y = self.conv1(F.silu(self.norm1(x, zq)))
        y = self.conv2(F.silu(self.norm2(y, zq)))
        return (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x) + y


class _AttnZ(nn.Module):
    def __init__(self, c, zq_ch):
        super().__init__()
        self.norm = _SpatialNorm(c, zq_ch)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x, zq):
        b, c, h, w = x.shape
        y = self.norm(x, zq)
        q, k, v = (m(y).flatten(2).transpose(1, 2) for m in (self.q, self.k, self.v))
        o = F.scaled_dot_product_attention(q, k, v)
        return x + self.proj_out(o.transpose(1, 2).reshape(b, c, h, w))


class _Emu3Decoder(nn.Module):
    def __init__(self, z_channels, embed_dim, ch, ch_mult, num_res_blocks, attn_levels, temporal_factor, out_channels):
        super().__init__()
        self.time_res_stack = nn.Sequential(*[_TemporalRes(z_channels) for _ in range(num_res_blocks)])
        n_up, f = 0, temporal_factor
        while f > 1:
            n_up, f = n_up + 1, f // 2
        self.time_conv = nn.ModuleList([_TemporalUp(z_channels) for _ in range(n_up)])
        c = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, c, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _ResZ(c, c, embed_dim), _AttnZ(c, embed_dim), _ResZ(c, c, embed_dim)
        levels = []
        for lvl in reversed(range(len(ch_mult))):          # coarse -> fine; stored fine-first as `up`
            level = nn.Module()
            level.block, level.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                level.block.append(_ResZ(c, ch * ch_mult[lvl], embed_dim))
                c = ch * ch_mult[lvl]
                if lvl in attn_levels:
                    level.attn.append(_AttnZ(c, embed_dim))
            if lvl != 0:
                level.upsample = _Up(c)
            levels.append(level)
        self.up = nn.ModuleList(list(reversed(levels)))
        self.norm_out = _SpatialNorm(c, embed_dim)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z, zq):
        """z (after post_quant_conv) and zq (raw codebook vectors): [b, t, c, h, w]; both run through the temporal stack together"""
        x = torch.cat((z, zq), dim=0).permute(0, 2, 1, 3, 4)           # [2b, c, t, h, w]
        x = self.time_res_stack(x)
        for up in self.time_conv:
            x = F.silu(up(x))
        h, zq = torch.chunk(x.permute(0, 2, 1, 3, 4), 2, dim=0)
        h, zq = h.reshape(-1, *h.shape[2:]), zq.reshape(-1, *zq.shape[2:])     # frames become batch entries
        h = self.conv_in(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h, zq), zq), zq)
        for level in reversed(self.up):
            for i, blk in enumerate(level.block):
                h = blk(h, zq)
                if len(level.attn) > 0:
                    h = level.attn[i](h, zq)
            if hasattr(level, "upsample"):
                h = level.upsample(h)
        return self.conv_out(F.silu(self.norm_out(h, zq)))


class Emu3VisionVQ(nn.Module):
    """Decode side of the Emu3 VisionVQ tokenizer (reference emu3/tokenizer/modeling_emu3visionvq.py:596-721 decoder, :790-812 decode):
    a causal temporal stack (4 output frames per latent frame) in front of a VQGAN decoder whose norms are modulated by the
    quantised latent.  ids [b, h, w] -> image [b, 3, 8h, 8w] (frame 0), ids [b, t, h, w] -> video [b, 4t, 3, 8h, 8w]."""

    def __init__(self, codebook_size=32768, embed_dim=4, z_channels=4, out_channels=3, temporal_downsample_factor=4, ch=256,
                 ch_mult=(1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(3,)):
        super().__init__()
        self.t_factor, self.out_channels, self.scale = temporal_downsample_factor, out_channels, 2 ** (len(ch_mult) - 1)
        self.quantize = _Codebook(codebook_size, embed_dim)
        self.post_quant_conv = _CausalConv3d(embed_dim, z_channels)
        self.decoder = _Emu3Decoder(z_channels, embed_dim, ch, tuple(ch_mult), num_res_blocks, tuple(attn_resolutions),
                                    temporal_downsample_factor, out_channels)

    def load_state_dict(self, state_dict, strict=True, **kw):
        return super().load_state_dict(_decode_side(state_dict), strict=strict, **kw)

    @torch.no_grad()
    def decode(self, x):
        image = x.dim() == 3
        if image:
            x = x.unsqueeze(1)
        b, t, h, w = x.shape
        zq = self.quantize.embedding(x).permute(0, 4, 1, 2, 3).contiguous()        # [b, c, t, h, w]
        z = self.post_quant_conv(zq)
        video = self.decoder(z.permute(0, 2, 1, 3, 4), zq.permute(0, 2, 1, 3, 4))
        video = video.reshape(b, t * self.t_factor, self.out_channels, h * self.scale, w * self.scale)
        return video[:, 0] if image else video


def to_uint8(img, truncate=False):
    """[-1, 1] float image [b, 3, H, W] -> uint8 [b, H, W, 3].  truncate=True is the Chameleon tokenizer's conversion
    (image_tokenizer.py:95-106: (clamp(x) + 1) / 2 * 255 cast to uint8); the default rounds."""
    x = (img.float().clamp(-1, 1) + 1) * 127.5
    x = x.floor() if truncate else x.round()
    return x.to(torch.uint8).permute(0, 2, 3, 1).contiguous()
