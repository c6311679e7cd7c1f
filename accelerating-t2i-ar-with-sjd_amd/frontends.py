"""Model-family front ends: what each family hands to the SJD engine at the `_sample` boundary (WindowSpec).

  lumina_*   : Lumina-mGPT / Anole (Chameleon arch).  cond||uncond batch, the uncond half is blind to prompt[:P-1]
               and its RoPE positions are shifted by -(P-1)   (reference jacobi_iteration_lumina_mgpt.py:703-712,
               742-770, 1000-1014).
  llamagen_* : LlamaGen.  class/caption conditioning is prefilled into cache rows [0,T) by the solver, `_sample`
               starts from the first image token at row T (reference llamagen/llamagen_solver.py:371-456).
  emu3_*     : Emu3.  positive||negative prompts left-padded to a common length; pads are masked keys
               (reference jacobi_iteration_emu3.py:234-278, logit_processor_3dim.py:422-440).
"""
import torch

from .engine import WindowSpec
from . import synthetic


def lumina_prompt(P, h_latent, w_latent, seed, text_lo=8900, text_hi=60000):
    """P ids: text ids then <start>=8197, grid tokens 8804 + latent/2 (reference item_processor.py:102-104)."""
    text = synthetic.synthetic_prompt(P - 3, seed, lo=text_lo, hi=text_hi)[0].tolist()
    return text + [8197, 8804 + h_latent // 2, 8804 + w_latent // 2]


def lumina_window_spec(prompt, device):
    P = len(prompt)
    ids = torch.tensor([prompt, prompt], dtype=torch.long, device=device)
    pos = torch.stack([torch.arange(P), torch.tensor([1] * (P - 1) + [0])]).to(device)
    return WindowSpec(first_tokens=ids, first_positions=pos, key_start=torch.tensor([0, P - 1], dtype=torch.int32),
                      pos_offset=torch.tensor([0, -(P - 1)], dtype=torch.long), kv_base=0)


def llamagen_window_spec(first_token, T, device, key_start=None):
    ids = torch.tensor([[first_token], [first_token]], dtype=torch.long, device=device)
    pos = torch.full((2, 1), T, dtype=torch.long, device=device)
    ks = torch.zeros(2, dtype=torch.int32) if key_start is None else torch.as_tensor(key_start, dtype=torch.int32)
    return WindowSpec(first_tokens=ids, first_positions=pos, key_start=ks, pos_offset=torch.zeros(2, dtype=torch.long),
                      kv_base=T)


def emu3_window_spec(pos_ids, neg_ids, pad_token_id, device):
    """get_double_cfg_input_ids + renew_attn_mask: left-pad to max length; pad columns are invisible keys."""
    Pp, Pn = len(pos_ids), len(neg_ids)
    Pm = max(Pp, Pn)
    rows = [[pad_token_id] * (Pm - Pp) + list(pos_ids), [pad_token_id] * (Pm - Pn) + list(neg_ids)]
    ids = torch.tensor(rows, dtype=torch.long, device=device)
    pads = []
    for r in rows:
        n = 0
        while n < Pm and r[n] == pad_token_id:
            n += 1
        pads.append(n)
    pos = torch.stack([(torch.arange(Pm) - p).clamp_min(0) for p in pads])
    for b, p in enumerate(pads):
        pos[b, :p] = 1                     # position_ids.masked_fill_(attention_mask == 0, 1) (JL:705-706)
    return WindowSpec(first_tokens=ids, first_positions=pos.to(device), key_start=torch.tensor(pads, dtype=torch.int32),
                      pos_offset=torch.tensor([-p for p in pads], dtype=torch.long), kv_base=0)
