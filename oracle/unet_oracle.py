"""Oracle (test infrastructure, not product): functional torch-CPU restatement of
the reference U-Nets.

Follows, op for op and in the same order:
  variant "unet"  -> /root/reference/src/unet.py:6-119
  variant "unet2" -> /root/reference/src/unet2.py:6-173

The network is described by a table of named tensors whose names, shapes, dtypes
and ORDER equal the reference ``state_dict()`` (261 entries for unet, 353 for
unet2 with heads [1,14,3,2,1,360,60,60]); the forward is a plain function over
that dict, so that autograd on CPU gives reference gradients too.

Pinned by tests/golden/*.npz, generated from the reference import
(tests/golden/make_golden.py) and checked in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

HEADS = [1, 14, 3, 2, 1, 360, 60, 60]  # src/train.py:47
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (prefix, kind, cin, cout, kernel) in forward/registration order.
#   unet.py:83-98 / unet2.py:135-150


def block_plan(variant: str, in_channels: int):
    if variant == "unet":
        stem, k = 16, 3
        d1_in = 16
    elif variant == "unet2":
        stem, k = 32, 5
        d1_in = 32
    else:
        raise ValueError(variant)
    return [
        ("inc1", "dc", in_channels, stem, k),
        ("inc2", "dc", stem, stem, k),
        ("down1", "down", d1_in, 32, 3),
        ("down2", "down", 32, 64, 3),
        ("inc3", "dc", 64, 64, 3),
        ("down3", "down", 64, 128, 3),
        ("down4", "down", 128, 256, 3),
        ("down5", "down", 256, 512, 3),
        ("up1", "up", 512, 256, 3),
        ("up2", "up", 256, 128, 3),
        ("up3", "up", 128, 128, 3),
        ("dconv1", "dc", 128, 128, 3),
        ("dconv2", "dc", 128, 128, 3),
    ]


def _bn_entries(prefix, c):
    return [
        (prefix + ".weight", (c,), torch.float32, "bn_w"),
        (prefix + ".bias", (c,), torch.float32, "bn_b"),
        (prefix + ".running_mean", (c,), torch.float32, "bn_rm"),
        (prefix + ".running_var", (c,), torch.float32, "bn_rv"),
        (prefix + ".num_batches_tracked", (), torch.int64, "bn_n"),
    ]


def _conv_entries(prefix, cout, cin, k):
    return [
        (prefix + ".weight", (cout, cin, k, k), torch.float32, "w"),
        (prefix + ".bias", (cout,), torch.float32, "b"),
    ]


def _dc_entries(variant, prefix, cin, cout, k):
    p = prefix + ".double_conv"
    e = []
    e += _conv_entries(p + ".0", cout, cin, k)
    e += _bn_entries(p + ".1", cout)
    e += _conv_entries(p + ".3", cout, cout, k)
    e += _bn_entries(p + ".4", cout)
    if variant == "unet2":
        mid = cout // 16
        ca = p + ".5.channel_attention.shared_MLP"
        e += [
            (ca + ".0.weight", (mid, cout), torch.float32, "w"),
            (ca + ".0.bias", (mid,), torch.float32, "b"),
            (ca + ".2.weight", (cout, mid), torch.float32, "w"),
            (ca + ".2.bias", (cout,), torch.float32, "b"),
        ]
        e += _conv_entries(p + ".5.spatial_attention.conv2d", 1, 2, 7)
        if cin != cout:
            e += _conv_entries(prefix + ".res_conv", cout, cin, 1)
    return e


def param_table(variant: str, in_channels: int = 1, heads=None):
    """Ordered (name, shape, dtype, role) list == reference state_dict order."""
    heads = list(HEADS if heads is None else heads)
    t = [("s", (10,), torch.float32, "s")]
    for prefix, kind, cin, cout, k in block_plan(variant, in_channels):
        if kind == "dc":
            t += _dc_entries(variant, prefix, cin, cout, k)
        elif kind == "down":
            t += _dc_entries(variant, prefix + ".maxpool_conv.1", cin, cout, k)
        else:  # up: ConvTranspose2d weight is (Cin, Cin//2, 3, 3)  (unet.py:44)
            t += [
                (prefix + ".up.weight", (cin, cin // 2, 3, 3), torch.float32, "wT"),
                (prefix + ".up.bias", (cin // 2,), torch.float32, "b"),
            ]
            t += _dc_entries(variant, prefix + ".conv", cin, cout, k)
    for i, h in enumerate(heads):
        p = "out_modules.%d" % i
        t += _conv_entries(p + ".conv1", 128, 128, 3)
        t += _bn_entries(p + ".bn", 128)
        t += _conv_entries(p + ".conv2", h, 128, 1)
    return t


def _fan_in(shape, role):
    if role == "wT":  # ConvTranspose2d: torch computes fan_in from dim 1
        return shape[1] * shape[2] * shape[3]
    n = 1
    for d in shape[1:]:
        n *= d
    return max(n, 1)


def filled_state(variant: str, in_channels: int = 1, heads=None, seed: int = 0):
    """Name-keyed deterministic fill: every tensor is drawn from a CPU generator
    seeded by crc32(name) ^ seed, so any box regenerates identical weights without
    the reference.  BN affine / running stats are deliberately non-trivial (some
    gamma < 0) so that folding and pool/affine ordering bugs cannot hide."""
    table = param_table(variant, in_channels, heads)
    fan = {}
    for name, shape, dt, role in table:
        if role in ("w", "wT"):
            fan[name.rsplit(".", 1)[0]] = _fan_in(shape, role)
    sd = OrderedDict()
    for name, shape, dt, role in table:
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)

        def uni(lo, hi):
            return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

        if role in ("w", "wT"):
            b = 1.0 / math.sqrt(fan[name.rsplit(".", 1)[0]])
            v = uni(-b, b)
        elif role == "b":
            b = 1.0 / math.sqrt(fan[name.rsplit(".", 1)[0]])
            v = uni(-b, b)
        elif role == "bn_w":
            v = uni(-0.4, 1.6)
        elif role == "bn_b":
            v = uni(-0.3, 0.3)
        elif role == "bn_rm":
            v = uni(-0.2, 0.2)
        elif role == "bn_rv":
            v = uni(0.5, 1.5)
        elif role == "bn_n":
            v = torch.zeros((), dtype=torch.int64)
        elif role == "s":
            v = uni(-0.02, 0.02)
        else:
            raise AssertionError(role)
        sd[name] = v
    return sd


CALIB_STEPS, CALIB_SIZE, CALIB_BATCH, CALIB_SEED0 = 60, 256, 2, 1000


def bn_stat_keys(sd):
    return [k for k in sd if k.endswith(("running_mean", "running_var"))]


def calibrated_state(variant: str, in_channels: int = 1, heads=None, seed: int = 0, steps: int = CALIB_STEPS,
                     size: int = CALIB_SIZE, batch: int = CALIB_BATCH, stats=None):
    """filled_state() whose BatchNorm running statistics MATCH its activations: the statistics the reference module
    ends up with after `steps` train-mode forwards (nn.BatchNorm2d's own update, momentum 0.1, unet.py:13,16,67) over
    seeded Bernoulli images (seed CALIB_SEED0 + i).  With the random running statistics of filled_state() the eval
    forward collapses (atom map spanning 0.05); with these the eval maps have the range of the train-mode ones.
    `stats` (a flat f32 vector in bn_stat_keys order, e.g. tests/golden/calibrated_*.npz: produced by the REFERENCE
    import) is loaded instead of recomputed when given."""
    sd = filled_state(variant, in_channels, heads, seed)
    if stats is not None:
        off = 0
        flat = torch.as_tensor(stats, dtype=torch.float32)
        for k in bn_stat_keys(sd):
            n = sd[k].numel()
            sd[k] = flat[off:off + n].clone()
            off += n
        assert off == flat.numel()
        for k in sd:
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.tensor(steps, dtype=torch.int64)
        return sd
    with torch.no_grad():
        for i in range(steps):
            forward(variant, sd, synthetic_image(batch, size, seed=CALIB_SEED0 + i, in_channels=in_channels), train=True)
    return sd


def synthetic_image(batch: int, size: int, seed: int = 7, in_channels: int = 1, p: float = 0.1):
    """Bernoulli(p) ink image in {0,1} f32 (contract of utils_for_test.py:26-39)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand((batch, in_channels, size, size), generator=g) < p).float()


# --------------------------------------------------------------------------
# forward


def _bn(sd, p, x, train):
    if train:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        train, BN_MOMENTUM, BN_EPS)


def _conv(sd, p, x, k):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=(k - 1) // 2)


def _cbam(sd, p, x):
    # unet2.py:19-22 (channel attention) and :30-35 (spatial attention), :43-46
    ca = p + ".channel_attention.shared_MLP"
    n, c = x.shape[0], x.shape[1]

    def mlp(v):
        v = F.relu(F.linear(v, sd[ca + ".0.weight"], sd[ca + ".0.bias"]))
        return F.linear(v, sd[ca + ".2.weight"], sd[ca + ".2.bias"])

    avg = mlp(F.adaptive_avg_pool2d(x, 1).view(n, -1)).unsqueeze(2).unsqueeze(3)
    mx = mlp(F.adaptive_max_pool2d(x, 1).view(n, -1)).unsqueeze(2).unsqueeze(3)
    out = torch.sigmoid(avg + mx) * x
    a = torch.mean(out, dim=1, keepdim=True)
    m, _ = torch.max(out, dim=1, keepdim=True)
    sa = torch.sigmoid(F.conv2d(torch.cat([a, m], dim=1), sd[p + ".spatial_attention.conv2d.weight"],
                                sd[p + ".spatial_attention.conv2d.bias"], padding=3))
    return sa * out


def _double_conv(variant, sd, prefix, x, k, train):
    p = prefix + ".double_conv"
    if variant == "unet":  # unet.py:11-18
        y = F.relu(_bn(sd, p + ".1", _conv(sd, p + ".0", x, k), train))
        return F.relu(_bn(sd, p + ".4", _conv(sd, p + ".3", y, k), train))
    # unet2.py:54-74
    y = F.relu(_bn(sd, p + ".1", _conv(sd, p + ".0", x, k), train))
    y = _bn(sd, p + ".4", _conv(sd, p + ".3", y, k), train)
    y = _cbam(sd, p + ".5", y)
    if (prefix + ".res_conv.weight") in sd:
        r = F.conv2d(x, sd[prefix + ".res_conv.weight"], sd[prefix + ".res_conv.bias"])
    else:
        r = x
    return F.relu(y + r)


def _up(variant, sd, prefix, x1, x2, train):
    # unet.py:48-60: transposed conv (k3, s2) then pad by floor-div halves
    # (torch>=1.13 tensor // is floor => diff=-1 crops FIRST row/col), cat skip first.
    x1 = F.conv_transpose2d(x1, sd[prefix + ".up.weight"], sd[prefix + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return _double_conv(variant, sd, prefix + ".conv", torch.cat([x2, x1], dim=1), 3, train)


def forward(variant: str, sd, x, train: bool = False, dropout_masks=None, dropout_p: float = 0.2,
            return_trunk: bool = False):
    """Returns the list of head maps (unet.py:100-119 / unet2.py:152-173).

    ``sd`` is mutated like a module in train mode (running stats, counters).
    ``dropout_masks``: optional list (one per head) of {0,1} keep-masks shaped
    like the head's 128-channel feature; used as x*mask/(1-p) (nn.Dropout
    semantics with an injected mask).  None => no dropout (p=0 / eval)."""
    plan = {b[0]: b for b in block_plan(variant, x.shape[1])}

    def dc(name, t):
        return _double_conv(variant, sd, name, t, plan[name][4], train)

    def down(name, t):
        return _double_conv(variant, sd, name + ".maxpool_conv.1", F.max_pool2d(t, 2), 3, train)

    x1 = dc("inc2", dc("inc1", x))
    x2 = down("down1", x1)
    x3 = dc("inc3", down("down2", x2))
    x4 = down("down3", x3)
    x5 = down("down4", x4)
    x6 = down("down5", x5)
    t = _up(variant, sd, "up1", x6, x5, train)
    t = _up(variant, sd, "up2", t, x4, train)
    t = _up(variant, sd, "up3", t, x3, train)
    t = dc("dconv2", dc("dconv1", t))
    outs = []
    i = 0
    while ("out_modules.%d.conv1.weight" % i) in sd:
        p = "out_modules.%d" % i
        f = F.leaky_relu(_bn(sd, p + ".bn", _conv(sd, p + ".conv1", t, 3), train), 0.01)
        if variant == "unet" and dropout_masks is not None:  # unet.py:69,73 (unet2 has no dropout)
            f = f * dropout_masks[i] / (1.0 - dropout_p)
        outs.append(F.conv2d(f, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]))
        i += 1
    if return_trunk:
        return outs, t
    return outs


def clone_state(sd, requires_grad: bool = False):
    out = OrderedDict()
    for k, v in sd.items():
        c = v.detach().clone()
        if requires_grad and c.is_floating_point() and not k.endswith(("running_mean", "running_var")):
            c.requires_grad_(True)
        out[k] = c
    return out
