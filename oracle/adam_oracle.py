"""Oracle (test infrastructure, not product): one step of torch.optim.Adam as the
reference configures it (/root/reference/src/train.py:55:
``Adam(lr=2.5e-4, weight_decay=1e-8)``, defaults betas=(0.9,0.999), eps=1e-8,
L2 decay added to the gradient, bias-corrected, eps added AFTER the sqrt of the
bias-corrected second moment).  Written out explicitly so that it can be checked
against torch.optim.Adam itself (tests/test_oracle_golden.py) and then serve as
the checker for the fused HIP optimiser.
"""
import math
import torch


def adam_step(p, g, m, v, step, lr=2.5e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-8):
    """In-place on p, m, v (all fp32, same shape); ``step`` is the 1-based count."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
    return p, m, v
