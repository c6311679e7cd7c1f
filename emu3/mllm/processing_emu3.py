"""reference import path emu3.mllm.processing_emu3.Emu3Processor (test_emu3.py:16, model_loader.py load_emu3).

The reference class is string templating around the hub tokenizer plus VisionVQ encode/decode (out of scope
except the grammar helper).  This version keeps what touches the SJD hot path at the token-id level:
  * `build_prefix_constrained_fn(h, w)` -> Emu3PrefixConstrainedLogitsHelper with the tokenizer's special ids (reference :255-290);
  * `calculate_generate_size(ratio, image_area, spatial_scale_factor)` (reference :246-253);
  * `__call__(text=..., mode='G', ratio=..., image_area=...)` -> {input_ids, image_size}: bos + text + boi + "H*W" + img token, built
    from the tokenizer the caller supplies (reference :81-182, generation mode only);
  * `decode(token_ids)` -> PIL images through a VisionVQ decoder (sjd_amd.detokenizers.Emu3VisionVQ) when one is attached.
`tokenizer` is any object with `encode(str | list[str]) -> ids` and the special-token attributes of Emu3's tokenizer; a
`SimpleNamespace` of ids works for token-id drivers (see tests/test_gpu_api.py)."""
from functools import partial

import torch

from .utils_emu3 import Emu3PrefixConstrainedLogitsHelper


class Emu3Processor:
    def __init__(self, image_processor=None, vision_tokenizer=None, tokenizer=None,
                 chat_template="You are a helpful assistant. USER: {image_prompt}{text_prompt}. ASSISTANT:", prefix_template="{H}*{W}",
                 visual_template=("<|visual token {token_id:0>6d}|>", r"<\|visual token (\d+)\|>"), **kwargs):
        assert vision_tokenizer is not None, "image tokenizer can not be None"
        self.image_processor, self.vision_tokenizer, self.tokenizer = image_processor, vision_tokenizer, tokenizer
        self.chat_template, self.prefix_template, self.visual_template = chat_template, prefix_template, visual_template
        self.const_helper = self.build_const_helper()

    def _id(self, token):
        ids = self.tokenizer.encode(token)
        return int(ids[0] if isinstance(ids, (list, tuple)) else ids)

    def build_const_helper(self):
        tk = self.tokenizer
        n_codes = getattr(getattr(self.vision_tokenizer, "config", self.vision_tokenizer), "codebook_size", 32768)
        vis_start = self._id(self.visual_template[0].format(token_id=0))
        vis_end = self._id(self.visual_template[0].format(token_id=n_codes - 1))
        return partial(Emu3PrefixConstrainedLogitsHelper, img_token=self._id(tk.img_token), eoi_token=self._id(tk.eoi_token),
                       eos_token=self._id(tk.eos_token), eol_token=self._id(tk.eol_token), eof_token=self._id(tk.eof_token),
                       pad_token=self._id(tk.pad_token), visual_tokens=list(range(vis_start, vis_end + 1)))

    def build_prefix_constrained_fn(self, height, width):
        return self.const_helper(height=height, width=width)

    @staticmethod
    def calculate_generate_size(ratio, image_area, spatial_scale_factor):
        w, h = map(int, ratio.split(":"))
        target_ratio = (image_area / (h * w)) ** 0.5
        return int(round(h * target_ratio / spatial_scale_factor)), int(round(w * target_ratio / spatial_scale_factor))

    def __call__(self, text=None, image=None, *, mode="G", ratio="1:1", image_area=518400, return_tensors=None, **kwargs):
        if mode != "G" or image is not None:
            raise NotImplementedError("understanding mode needs the VisionVQ encoder (pre-processing, not on the SJD hot path)")
        text = [text] if isinstance(text, str) else list(text)
        if len(text) != 1:
            raise ValueError("`text` can only be `str` in generation mode")
        tk = self.tokenizer
        h, w = self.calculate_generate_size(ratio, image_area, getattr(self.vision_tokenizer, "spatial_scale_factor", 8))
        prompt = tk.bos_token + text[0] + tk.boi_token + self.prefix_template.format(H=h, W=w) + tk.img_token
        ids = list(tk.encode(prompt))
        out = {"input_ids": torch.tensor([ids]) if return_tensors == "pt" else [ids], "image_size": [[h, w]]}
        return type("BatchFeature", (dict,), {"__getattr__": dict.__getitem__})(out)

    @torch.no_grad()
    def decode(self, token_ids):
        """ids of one generated sequence -> list of PIL images (reference :189-226, restated on ids instead of on the decoded string)."""
        from PIL import Image
        from sjd_amd.detokenizers import to_uint8
        helper = self.const_helper(height=0, width=0)
        lo, n = helper.visual_tokens[0], len(helper.visual_tokens)
        ids = [int(t) for t in (token_ids.tolist() if torch.is_tensor(token_ids) else token_ids)]
        rows, cur, images = [], [], []
        for t in ids:
            if lo <= t < lo + n:
                cur.append(t - lo)
            elif t == helper.eol_token and cur:
                rows.append(cur)
                cur = []
            elif t in (helper.eof_token, helper.eoi_token) and rows:
                codes = torch.tensor(rows, dtype=torch.long, device=next(self.vision_tokenizer.parameters()).device)
                img = self.vision_tokenizer.decode(codes[None]).float()
                images.append(Image.fromarray(to_uint8(img)[0].cpu().numpy()))
                rows = []
        return images
