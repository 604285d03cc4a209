import numpy as np
import torch


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max())


def vt_key_pos(k):
    """Position of key k in the PRIMX_HEADS_VT layout (csrc/common.h vt_key_pos)."""
    quad = (k >> 2) & 3
    perm = ((quad & 1) << 1) | (quad >> 1)
    return (k & ~15) | (perm << 2) | (k & 3)


def unpack_rows(buf, n, dh):
    """[B, H, n_pad, DP] -> [B, n, H, dh]"""
    return buf[:, :, :n, :dh].permute(0, 2, 1, 3)


def unpack_vt(buf, n, dh):
    """[B, H, DP, n_pad] (quad-permuted keys) -> [B, n, H, dh]"""
    pos = torch.tensor([vt_key_pos(k) for k in range(n)], device=buf.device)
    return buf[:, :, :dh, :].index_select(3, pos).permute(0, 3, 1, 2)
