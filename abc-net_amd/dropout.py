"""Bit-exact torch mirror of the device dropout hash (csrc/common.hpp: abc_drop_keep), so that
tests can drive the oracle with the identical keep-mask (torch RNG parity is impossible, K7)."""
import torch


def keep_mask(idx: torch.Tensor, seed: int, p: float) -> torch.Tensor:
    """idx: int64 tensor of element indices (pixel*ld + channel).  Returns bool keep-mask."""
    M = 0xFFFFFFFF
    h = ((idx * 0x9E3779B1) & M) ^ (seed & M)
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    h = h ^ (h >> 16)
    u = (h >> 8).to(torch.float32) * (1.0 / 16777216.0)
    return u >= p


def head_keep_masks(B, h, w, n_heads, seed, p):
    """keep-masks [B,128,h,w] per head, for the feature layout the engine uses (NHWC rows of n_heads*128)"""
    ld = 128 * n_heads
    pix = torch.arange(B * h * w, dtype=torch.int64).view(B, h, w, 1)
    out = []
    for i in range(n_heads):
        ch = torch.arange(128, dtype=torch.int64).view(1, 1, 1, 128) + 128 * i
        out.append(keep_mask(pix * ld + ch, seed, p).permute(0, 3, 1, 2).float().contiguous())
    return out
