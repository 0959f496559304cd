"""onet_amd: MI355X-native (gfx950) implementation of the Onet twin-U-Net hot path.

    from onet_amd import Onet            # same surface as the reference's Onet_vanilla_20240606.Onet

Python host code + a C-ABI library of hand-written HIP kernels (include/onet_hip.h).
PyTorch-ROCm supplies device memory, streams, autograd orchestration and torch.distributed
(RCCL) -- plumbing only.  There is no CPU compute fallback."""
from .ops import set_sync_bn
from .modules import (BatchNormReLU2d, BilinearUp2x, Conv3x3, ConvT2x2, DoubleConv, Down, MaxPool2, Onet, UNet,
                      Up, invalidate_packed)

__all__ = ["Onet", "UNet", "Up", "Down", "DoubleConv", "Conv3x3", "ConvT2x2", "BatchNormReLU2d", "MaxPool2",
           "BilinearUp2x", "invalidate_packed", "set_sync_bn"]
__version__ = "0.1.0"
