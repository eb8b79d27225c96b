"""kaamer_amd — MI355X-native k-mer search path for kaamer.

The product is libkaamer_hip.so (HIP kernels + the C ABI of include/kaamer_hip.h).
This package is the thin Python plumbing around it: a ctypes binding (`abi`),
and `Index` / `Workspace` helpers that hand torch-owned device memory and HIP
streams to the C ABI.  There is no CPU fallback: importing `kaamer_amd.abi`
fails loudly when the library is missing.
"""
from . import abi  # noqa: F401
from .api import Image, Index, Workspace, build_image_from_proteins, pack_sequences  # noqa: F401

__all__ = ["abi", "Image", "Index", "Workspace", "build_image_from_proteins", "pack_sequences"]
