"""Architecture tables of the two reference U-Nets: parameter/buffer names, shapes and
order exactly as the reference ``state_dict()`` (checkpoint row),
and the block topology the engine lowers to kernel launches.

  variant "unet"  : /root/reference/src/unet.py:77-119   (261 state-dict entries)
  variant "unet2" : /root/reference/src/unet2.py:129-173 (353 entries; CBAM + residual)
"""
from __future__ import annotations

from collections import OrderedDict

DEFAULT_HEADS = [1, 21, 5, 1, 4, 2]  # unet.py:78 default argument
TRAIN_HEADS = [1, 14, 3, 2, 1, 360, 60, 60]  # train.py:47


def blocks(variant: str, in_channels: int):
    """(name, kind, cin, cout, k) in registration/forward order."""
    if variant == "unet":
        stem, k, d1 = 16, 3, 16
    elif variant == "unet2":
        stem, k, d1 = 32, 5, 32
    else:
        raise ValueError("unknown variant %r" % variant)
    return [
        ("inc1", "dc", in_channels, stem, k), ("inc2", "dc", stem, stem, k),
        ("down1", "down", d1, 32, 3), ("down2", "down", 32, 64, 3), ("inc3", "dc", 64, 64, 3),
        ("down3", "down", 64, 128, 3), ("down4", "down", 128, 256, 3), ("down5", "down", 256, 512, 3),
        ("up1", "up", 512, 256, 3), ("up2", "up", 256, 128, 3), ("up3", "up", 128, 128, 3),
        ("dconv1", "dc", 128, 128, 3), ("dconv2", "dc", 128, 128, 3),
    ]


def dc_prefix(name: str, kind: str) -> str:
    return {"dc": name, "down": name + ".maxpool_conv.1", "up": name + ".conv"}[kind]


def _bn(prefix, c):
    return [(prefix + ".weight", (c,), "param"), (prefix + ".bias", (c,), "param"),
            (prefix + ".running_mean", (c,), "buffer"), (prefix + ".running_var", (c,), "buffer"),
            (prefix + ".num_batches_tracked", (), "counter")]


def _conv(prefix, co, ci, k):
    return [(prefix + ".weight", (co, ci, k, k), "param"), (prefix + ".bias", (co,), "param")]


def _dc(variant, prefix, ci, co, k):
    p = prefix + ".double_conv"
    e = _conv(p + ".0", co, ci, k) + _bn(p + ".1", co) + _conv(p + ".3", co, co, k) + _bn(p + ".4", co)
    if variant == "unet2":
        mid = co // 16
        m = p + ".5.channel_attention.shared_MLP"
        e += [(m + ".0.weight", (mid, co), "param"), (m + ".0.bias", (mid,), "param"),
              (m + ".2.weight", (co, mid), "param"), (m + ".2.bias", (co,), "param")]
        e += _conv(p + ".5.spatial_attention.conv2d", 1, 2, 7)
        if ci != co:
            e += _conv(prefix + ".res_conv", co, ci, 1)
    return e


def state_table(variant: str, in_channels: int, heads):
    """Ordered list of (name, shape, role) with role in {param, buffer, counter}."""
    t = [("s", (10,), "param")]
    for name, kind, ci, co, k in blocks(variant, in_channels):
        if kind == "up":
            t += [(name + ".up.weight", (ci, ci // 2, 3, 3), "param"), (name + ".up.bias", (ci // 2,), "param")]
        t += _dc(variant, dc_prefix(name, kind), ci, co, k)
    for i, h in enumerate(heads):
        p = "out_modules.%d" % i
        t += _conv(p + ".conv1", 128, 128, 3) + _bn(p + ".bn", 128) + _conv(p + ".conv2", h, 128, 1)
    return t


def numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def arena_layout(table, role):
    """name -> (offset, numel) inside the flat arena holding every tensor of `role`, densely
    packed in state-dict order (so the arena's numel equals the reference parameter count)."""
    out, off = OrderedDict(), 0
    for name, shape, r in table:
        if r != role:
            continue
        n = numel(shape)
        out[name] = (off, n)
        off += n
    return out, off
