"""Drop-in for ha/conv.py (DWConv1d, ConvEncoder: the strided front-end of the attention ASR encoder),
forward only, on csrc/conv.hip + the GEMMs.  Same constructor arguments and state-dict names
(``conv.0.{weight,bias}``, ``conv.{i}.depthwise.*``, ``conv.{i}.pointwise.*``).  Internally activations
are channels-last [N, T, C]; ``forward`` keeps the reference's channels-first signature, ``forward_cl`` is
what AudioEncoder uses (no transposes)."""
import torch
import torch.nn as nn

from . import _lib, ops
from ._linear import WeightImages, linear, linear_dw, linear_dx


class DWConv1d(nn.Module):
    "Depthwise separable convolution (ha/conv.py:6-22)"
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, bias=True):
        super().__init__()
        if dilation != 1:
            raise NotImplementedError('dilated depthwise convolutions are not built (the reference never uses them)')
        self.kernel_size = kernel_size,
        self.stride = stride,
        self.padding = padding,
        self.dilation = dilation,
        self.bias = bias,
        self.depthwise = nn.Conv1d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                   dilation=dilation, groups=in_channels, bias=bias)
        self.pointwise = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=bias)
        self._images = WeightImages()

    def forward_cl(self, x, gelu=False):
        """x [N, T, C] -> [N, T', C_out]; ``gelu`` fuses F.gelu into the pointwise GEMM."""
        dw = self.depthwise
        y = ops.dwconv1d_cl(x.contiguous(), dw.weight.detach().reshape(dw.weight.shape[0], -1).contiguous(),
                            dw.bias.detach() if dw.bias is not None else None, dw.stride[0], dw.padding[0])
        N, To, C = y.shape
        pw = self.pointwise
        out = linear(self._images, y.view(N * To, C), pw.weight,
                     bias=pw.bias.detach() if pw.bias is not None else None, gelu='erf' if gelu else False)
        return out.view(N, To, -1)

    def forward(self, x):
        return self.forward_cl(x.mT.contiguous()).mT

    # training: keep the depthwise output and the pointwise pre-activation
    def _forward_cl_train(self, x):
        dw, pw = self.depthwise, self.pointwise
        y = ops.dwconv1d_cl(x, dw.weight.detach().reshape(dw.weight.shape[0], -1).contiguous(),
                            dw.bias.detach() if dw.bias is not None else None, dw.stride[0], dw.padding[0])
        N, To, C = y.shape
        a = linear(self._images, y.view(N * To, C), pw.weight, bias=pw.bias.detach() if pw.bias is not None else None)
        out = ops.gelu_fwd(a, exact=True)
        return out.view(N, To, -1), (x, y, a)

    def _backward_cl(self, saved, dout, put, want_dx=True):
        x, y, a = saved
        dw, pw = self.depthwise, self.pointwise
        N, To, C = y.shape
        da = ops.gelu_bwd(dout.reshape(N * To, -1), a, exact=True)
        put(pw.weight, linear_dw(da, y.view(N * To, C)).view_as(pw.weight))
        put(pw.bias, ops.colsum(da) if pw.bias is not None else None)
        dy = linear_dx(self._images, da, pw.weight).view(N, To, C)
        dx, dww, dwb = ops.dwconv1d_cl_bwd(dy, x, dw.weight.detach().reshape(C, -1).contiguous(), dw.stride[0], dw.padding[0],
                                           want_dx=want_dx, has_bias=dw.bias is not None)
        put(dw.weight, dww.view_as(dw.weight)); put(dw.bias, dwb)
        return dx


class ConvEncoder(nn.Module):
    def __init__(self, *, input_dim: int, hidden_dim: int, output_dim: int, strides: tuple, kernel_size: int = 3):
        super().__init__()
        conv = [nn.Conv1d(input_dim, hidden_dim, kernel_size=kernel_size, stride=strides[0], padding=1)]
        for stride in strides[1:-1]:
            conv.append(DWConv1d(hidden_dim, hidden_dim, kernel_size=kernel_size, stride=stride, padding=1))
        conv.append(DWConv1d(hidden_dim, output_dim, kernel_size=kernel_size, stride=strides[-1], padding=1))
        self.conv = nn.ModuleList(conv)
        self._images = WeightImages()

    def subsampled_lengths(self, input_lengths):
        # ha/conv.py:35-42, float floor per layer, int32 result
        o = input_lengths
        for conv in self.conv:
            p, k, s = conv.padding[0], conv.kernel_size[0], conv.stride[0]
            o = o + 2 * p - k
            o = torch.floor(o / s + 1)
        return o.int()

    def forward_cl(self, x):
        """x [N, T, F] channels-last -> [N, T', C_out]: gelu(conv(x)) per layer (ha/conv.py:44-47)."""
        if not x.is_cuda:
            raise _lib.HaloError('haloop_amd.conv.ConvEncoder runs on the HIP device only (no CPU path)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError('haloop_amd.conv.ConvEncoder has no autograd of its own: train it through '
                                      'haloop_amd.transformer.AudioEncoder, or call it under torch.no_grad()')
        first = self.conv[0]
        col, To = ops.im2col_cl(x.float().contiguous(), first.kernel_size[0], first.stride[0], first.padding[0])
        y = linear(self._images, col, first.weight, bias=first.bias.detach(), gelu='erf').view(x.shape[0], To, -1)
        for conv in list(self.conv)[1:]:
            y = conv.forward_cl(y, gelu=True)
        return y

    def _forward_cl_train(self, x):
        first = self.conv[0]
        x = x.float().contiguous()
        col, To = ops.im2col_cl(x, first.kernel_size[0], first.stride[0], first.padding[0])
        a = linear(self._images, col, first.weight, bias=first.bias.detach())
        y = ops.gelu_fwd(a, exact=True).view(x.shape[0], To, -1)
        saved = [(col, a)]
        for conv in list(self.conv)[1:]:
            y, sv = conv._forward_cl_train(y)
            saved.append(sv)
        return y, saved

    def _backward_cl(self, saved, dout, put):
        """Parameter gradients only: the input is data (mel features)."""
        convs = list(self.conv)
        for i in range(len(convs) - 1, 0, -1):
            dout = convs[i]._backward_cl(saved[i], dout, put)
        col, a = saved[0]
        first = convs[0]
        da = ops.gelu_bwd(dout.reshape(a.shape), a, exact=True)
        put(first.weight, linear_dw(da, col).view_as(first.weight))
        put(first.bias, ops.colsum(da))

    def forward(self, x, input_lengths):
        "x [N, F, T] channels-first like the reference"
        return self.forward_cl(x.mT).mT, self.subsampled_lengths(input_lengths)
