"""Host-side RoPE table for the fused qk-norm/RoPE/cache-write kernel.

Same numbers as the reference's complex128 `freqs` buffer (rope_params, wan/modules/model.py:28-35,
concatenated as wan/modules/causal_model.py:639-645: 22 temporal + 21 row + 21 column frequency pairs
for head_dim 128), computed in float64 and stored as float32 (cos, sin) pairs: [1024, head_dim/2, 2].
"""
import torch


def _axis_angles(max_pos, dim, theta=10000.0):
    inv = 1.0 / torch.pow(torch.tensor(theta, dtype=torch.float64),
                          torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    return torch.outer(torch.arange(max_pos, dtype=torch.float64), inv)


def rope_cos_sin_table(head_dim, max_pos=1024):
    d = head_dim
    ang = torch.cat([_axis_angles(max_pos, d - 4 * (d // 6)), _axis_angles(max_pos, 2 * (d // 6)),
                     _axis_angles(max_pos, 2 * (d // 6))], dim=1)
    return torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).to(torch.float32).contiguous()
