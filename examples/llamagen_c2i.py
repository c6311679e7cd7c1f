#!/usr/bin/env python3
"""End-to-end class-conditional LlamaGen generation with Speculative Jacobi Decoding on one MI355X: the flow of the reference's
test_llamagen.py (model registry -> renew_llamagen / renew_sampler -> LlamaGenSolver.generate -> VQ decode -> image file), with the
reference's import lines.  Checkpoints: pass --gpt-ckpt / --vq-ckpt (the reference's files load unchanged: same state-dict keys);
without them both networks get synthetic weights, which exercises every step but of course draws noise.

    python examples/llamagen_c2i.py --class-id 207 --out sample.png
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from llamagen.llamagen import GPT_models  # noqa: E402                                  (test_llamagen.py:19)
from llamagen.llamagen_solver import LlamaGenSolver, renew_llamagen  # noqa: E402       (test_llamagen.py:20)
from llamagen.tokenizer.tokenizer_image.vq_model import VQ_models  # noqa: E402         (test_llamagen.py:17)
from scheduler.jacobi_iteration_lumina_mgpt import renew_sampler  # noqa: E402          (test_llamagen.py:21)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpt-model", default="GPT-B", choices=list(GPT_models))
    ap.add_argument("--gpt-ckpt", default=None)
    ap.add_argument("--vq-model", default="VQ-16", choices=list(VQ_models))
    ap.add_argument("--vq-ckpt", default=None)
    ap.add_argument("--image-size", type=int, default=256)
    ap.add_argument("--class-id", type=int, default=207)
    ap.add_argument("--cfg-scale", type=float, default=4.0)
    ap.add_argument("--top-k", type=int, default=1000)
    ap.add_argument("--top-p", type=float, default=1.0)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="sample.png")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    import sjd_amd.ops as ops
    import sjd_amd.synthetic as synthetic
    latent = a.image_size // (16 if a.vq_model == "VQ-16" else 8)

    vq = VQ_models[a.vq_model](codebook_size=16384, codebook_embed_dim=8).to(dev).eval()
    if a.vq_ckpt:
        vq.load_state_dict(torch.load(a.vq_ckpt, map_location="cpu")["model"])
    else:
        synthetic.fill_state_dict_conv(vq, seed=1)

    gpt = GPT_models[a.gpt_model](block_size=latent ** 2, cls_token_num=1, model_type="c2i", num_classes=1000).to(dev, torch.bfloat16).eval()
    gpt.attn = ops.HipWindowAttention()
    if a.gpt_ckpt:
        ck = torch.load(a.gpt_ckpt, map_location="cpu")
        gpt.load_state_dict(ck.get("model", ck), strict=False)
    else:
        synthetic.fill_state_dict_device(gpt, seed=0, embed_token_scale=0.5)
    jac = dict(jacobi_loop_interval_l=1, jacobi_loop_interval_r=latent ** 2 - a.window - 2, max_num_new_tokens=a.window,
               guidance_scale=a.cfg_scale, seed=a.seed, multi_token_init_scheme="random", do_cfg=True, image_top_k=a.top_k,
               text_top_k=10, prefix_token_sampler_scheme="speculative_jacobi")
    gpt.__class__ = renew_llamagen(gpt.__class__)
    gpt._init_new_params(**jac)
    gpt.__class__ = renew_sampler(gpt.__class__)
    gpt._init_new_params(**jac)

    solver = LlamaGenSolver(model=gpt, image_top_k=a.top_k, image_top_p=a.top_p)
    torch.manual_seed(a.seed)
    t0 = time.time()
    index_sample = solver.generate(torch.tensor([a.class_id], device=dev), latent ** 2, None, cfg_scale=a.cfg_scale, temperature=1.0,
                                   top_k=a.top_k, top_p=a.top_p, sample_logits=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st = gpt.last_sjd_stats
    print(f"{index_sample.shape[1]} image tokens in {st.nfe} forward passes ({index_sample.shape[1] / max(st.nfe, 1):.2f} tokens/step), {dt:.2f} s")

    samples = vq.decode_code(index_sample.reshape(-1), (1, 8, latent, latent))          # [-1, 1]   (test_llamagen.py:182)
    from sjd_amd.detokenizers import to_uint8
    from PIL import Image
    Image.fromarray(to_uint8(samples)[0].cpu().numpy()).save(a.out)
    print(f"wrote {a.out} ({samples.shape[-1]}x{samples.shape[-2]})")


if __name__ == "__main__":
    main()
