"""Post-processing network of ``mode: train_postprocess`` -- mirrors graphs/layers/post_processing_networks.py of the
reference: ``PostProcessResidual`` (:39-52) and ``PostProcessingiWave`` (:54-77, config.postprocess == "iwave"): a 3x3
conv 3 -> 64*clrch, ``resnetlevel`` residual blocks (conv, ReLU, conv, + input), a 3x3 conv, + the first conv's output, a
3x3 conv back to 3 channels, + the input image.  Same parameter names (``convFilter``, ``interConvFilter``,
``outputConvFilter``, ``resNetList.{i}.resNet.{0,2}``) and the reference's initialisation of the residual blocks
(trunc_normal std 0.01, zero bias, :6-12).  Every conv runs on the MFMA conv engine (ReLU and the residual adds in its
epilogue); in training the engine's backward-data / weight-gradient kernels are used through autograd.ConvFn.

The other choices of the reference's agent (DnCNN with BatchNorm, IRCNN with dilated convs, DIDN, DUDnCNN;
agents/liftingDWT_agent.py:27-36) are not built: SURVEY.md 8f.3 names this file's :54-77.
"""
import torch
from torch import nn

from ... import autograd as ag
from ... import ops
from ...packed_cache import PackedOwnerMixin, cached


def init_weights(m):
    """post_processing_networks.py:6-12."""
    if type(m) == nn.Conv2d:
        torch.nn.init.trunc_normal_(m.weight, std=0.01)
        m.bias.data.fill_(0)


def _conv(m, x, act=ops.ACT_NONE, residual=None, train=False):
    """x (1,B,C,H,W) -> conv3x3 of module m (+ residual, activation) on the engine."""
    if train:
        return ag.conv(x, m.weight[None], m.bias[None], 3, act=act, residual=residual)
    w, b, packed = cached(m, ("pp",), [m.weight, m.bias], lambda: (
        m.weight.detach()[None].contiguous(), m.bias.detach()[None].contiguous(), ops.conv_pack(m.weight.detach()[None].contiguous(), 3)))
    return ops.conv2d(x, w, b, 3, act=act, residual=residual, packed=packed)


class PostProcessResidual(PackedOwnerMixin, nn.Module):
    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.channelNumber = config.clrch * 64
        self.resWeight = 1
        self.resNet = nn.Sequential(nn.Conv2d(self.channelNumber, self.channelNumber, (3, 3), stride=1, padding="same"), nn.ReLU(),
                                    nn.Conv2d(self.channelNumber, self.channelNumber, (3, 3), stride=1, padding="same"))
        self.resNet.apply(init_weights)

    def forward_pm(self, x, train):
        t = _conv(self.resNet[0], x, ops.ACT_RELU, train=train)
        return _conv(self.resNet[2], t, residual=x, train=train)              # tmp + inputImage (:50-51)

    def forward(self, inputImage):
        return self.forward_pm(inputImage[None].contiguous(), torch.is_grad_enabled() and self.training)[0]


class PostProcessingiWave(PackedOwnerMixin, nn.Module):
    def __init__(self, config):
        super().__init__()
        self._init_packed_owner()
        self.channelNumber = config.clrch * 64
        self.numberOfResnets = config.resnetlevel
        self.convFilter = nn.Conv2d(3, self.channelNumber, (3, 3), stride=1, padding="same")
        self.interConvFilter = nn.Conv2d(self.channelNumber, self.channelNumber, (3, 3), stride=1, padding="same")
        self.outputConvFilter = nn.Conv2d(self.channelNumber, 3, (3, 3), stride=1, padding="same")
        self.resNetList = nn.ModuleList()
        self.resWeight = 1
        for _ in range(self.numberOfResnets):
            self.resNetList.append(PostProcessResidual(config))

    def forward(self, inputImage):
        """(B,3,H,W) reconstructed RGB -> (B,3,H,W) (:68-77)."""
        train = torch.is_grad_enabled() and self.training
        x = inputImage[None].contiguous()
        t1 = _conv(self.convFilter, x, train=train)
        t2 = t1
        for blk in self.resNetList:
            t2 = blk.forward_pm(t2, train)
        t2 = _conv(self.interConvFilter, t2, residual=t1, train=train)          # interConv(tmp_2) + tmp_1
        return _conv(self.outputConvFilter, t2, residual=x, train=train)[0]     # outputConv(tmp_2) + inputImage


def make_postprocess(config):
    """The dispatch of agents/liftingDWT_agent.py:27-36."""
    kind = config.postprocess
    if kind == "iwave":
        return PostProcessingiWave(config)
    raise NotImplementedError("postprocess %r is not built (only 'iwave', post_processing_networks.py:54-77; the reference "
                              "also offers DnCNN, IRCNN, DIDN, DUDnCNN)" % kind)
