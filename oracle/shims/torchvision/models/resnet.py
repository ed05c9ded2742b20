import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1


class Bottleneck(nn.Module):
    expansion = 4


def conv1x1(*a, **k):
    raise NotImplementedError


def conv3x3(*a, **k):
    raise NotImplementedError
