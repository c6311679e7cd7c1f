"""fp64 reference of the draft-window attention + KV append (kernels K1/K3).

TEST INFRASTRUCTURE ONLY (see sjd_oracle.c header).  Restates, for one layer,
  DynamicCache.update / KVCache.update   (reference modeling_chameleon.py:547, llamagen.py:210-219)
  repeat_kv                              (modeling_chameleon.py:222-232)
  additive causal/padding mask           (jacobi_iteration_lumina_mgpt.py:1308-1324)
  softmax(QK^T/sqrt(D) + mask) V         (modeling_chameleon.py:567, llamagen.py:269)
as:  key j is visible to window row i of batch b  iff  key_start[b] <= j <= kv_len + i.
Rows with no visible key return zeros (the reference returns an unused uniform average there).
"""
import math

import torch


class OracleWindowAttention:
    def __init__(self, compute_dtype=torch.float64):
        self.compute_dtype = compute_dtype

    def __call__(self, layer, q, k, v, cache, kv_len, key_start):
        """q [B,n,H,D]; k,v [B,n,Hkv,D]; cache.k/v [layers,B,Hkv,S,D]; kv_len int; key_start [B] ints -> [B,n,H,D]"""
        kv_len = int(kv_len)
        B, n, H, D = q.shape
        Hkv = k.shape[2]
        cache.k[layer, :, :, kv_len:kv_len + n] = k.transpose(1, 2).to(cache.k.dtype)
        cache.v[layer, :, :, kv_len:kv_len + n] = v.transpose(1, 2).to(cache.v.dtype)
        total = kv_len + n
        K = cache.k[layer, :, :, :total].to(self.compute_dtype)       # [B,Hkv,total,D]
        Vv = cache.v[layer, :, :, :total].to(self.compute_dtype)
        g = H // Hkv
        K = K.repeat_interleave(g, dim=1)
        Vv = Vv.repeat_interleave(g, dim=1)
        Q = q.transpose(1, 2).to(self.compute_dtype)                  # [B,H,n,D]
        S = Q @ K.transpose(-1, -2) / math.sqrt(D)                    # [B,H,n,total]
        j = torch.arange(total, device=q.device)[None, None, None, :]
        i = torch.arange(n, device=q.device)[None, None, :, None]
        ks = torch.as_tensor(key_start, device=q.device).view(B, 1, 1, 1)
        visible = (j >= ks) & (j <= kv_len + i)
        S = S.masked_fill(~visible, float("-inf"))
        m = S.max(dim=-1, keepdim=True).values
        m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
        E = torch.exp(S - m)
        denom = E.sum(-1, keepdim=True)
        Pm = torch.where(denom > 0, E / denom.clamp_min(1e-300), torch.zeros_like(E))
        O = Pm @ Vv                                                   # [B,H,n,D]
        return O.transpose(1, 2).to(q.dtype)
