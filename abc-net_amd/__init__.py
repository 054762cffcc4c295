"""abcnet_amd: MI355X-native (gfx950) implementation of ABC-Net's U-Net hot path.

Loaded as ``abcnet_amd`` through the ``abcnet_amd.py`` shim at the repo root.
Heavy submodules (the HIP library) load lazily so that CPU-only tooling can import
the package; any compute entry point fails loudly when the HIP library is absent.
"""
__all__ = ["synthetic"]
