"""Stand-in for the three MONAI factories the reference's unet_blocks.py /
swin_unetr.py import (MONAI is not installable here).  Used ONLY by
gen_golden.py in the build container.  Own code; maps each factory to the stock
torch module MONAI resolves it to for the arguments the reference passes:

  get_act_layer('leakyrelu')          -> nn.LeakyReLU()            (slope 0.01)
  get_norm_layer('batch', 3, C)       -> nn.BatchNorm3d(C)
  Convolution(3, Cin, Cout, strides=1, kernel_size=3, conv_only=True)
                                      -> nn.Sequential with child 'conv' =
                                         nn.Conv3d(Cin, Cout, 3, 1, padding=1, bias=True)

UnetrBasicBlock / UnetrUpBlock are not restated: constructing them raises.
"""
import sys
import types

import torch.nn as nn


def get_act_layer(name):
    if name != "leakyrelu":
        raise NotImplementedError(name)
    return nn.LeakyReLU()


def get_norm_layer(name, spatial_dims, channels):
    if name != "batch" or spatial_dims != 3:
        raise NotImplementedError((name, spatial_dims))
    return nn.BatchNorm3d(channels)


class Convolution(nn.Sequential):
    def __init__(self, spatial_dims, in_channels, out_channels, strides=1, kernel_size=3,
                 act=None, norm=None, conv_only=False, is_transposed=False, **kw):
        super().__init__()
        if spatial_dims != 3 or not conv_only or is_transposed:
            raise NotImplementedError("stand-in covers only the reference's call")
        k = tuple(kernel_size) if not isinstance(kernel_size, int) else (kernel_size,) * 3
        pad = tuple((v - 1) // 2 for v in k)
        self.add_module("conv", nn.Conv3d(in_channels, out_channels, k, tuple(strides), padding=pad, bias=True))


class _Unavailable(nn.Module):
    def __init__(self, *a, **kw):
        raise NotImplementedError("MONAI block not available in this image")


def install():
    monai = types.ModuleType("monai")
    networks = types.ModuleType("monai.networks")
    blocks = types.ModuleType("monai.networks.blocks")
    layers = types.ModuleType("monai.networks.layers")
    utils = types.ModuleType("monai.networks.layers.utils")
    blocks.Convolution = Convolution
    blocks.UnetrBasicBlock = _Unavailable
    blocks.UnetrUpBlock = _Unavailable
    utils.get_act_layer = get_act_layer
    utils.get_norm_layer = get_norm_layer
    monai.networks, networks.blocks, networks.layers, layers.utils = networks, blocks, layers, utils
    for m in (monai, networks, blocks, layers, utils):
        sys.modules[m.__name__] = m
