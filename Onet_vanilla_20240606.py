"""Drop-in shim: ``import Onet_vanilla_20240606 as onet_vanilla_model`` (TS:29, TZ:21, EN:25, TP:25 of
the reference) resolves to the MI355X-native implementation when /root/repo precedes the
reference on sys.path.  Same public names as the reference module; import-time RNG seeding
mirrors OV:33-35 because the reference trainers rely on it."""
import numpy as np
import torch

from onet_amd.modules import DoubleConv, Down, Onet, UNet, Up  # noqa: F401

torch.manual_seed(1981)
np.random.seed(1981)
torch.set_default_dtype(torch.float32)
