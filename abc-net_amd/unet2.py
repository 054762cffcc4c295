"""``from unet2 import UNet`` replacement (reference: /root/reference/src/unet2.py:129-173): CBAM + residual variant."""
from .model import UNetBase


class UNet(UNetBase):
    VARIANT = "unet2"
