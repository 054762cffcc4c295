"""``from unet import UNet`` replacement (reference: /root/reference/src/unet.py:77-119)."""
from .model import UNetBase


class UNet(UNetBase):
    VARIANT = "unet"
