"""VSRN region-relationship reasoning (itr/modalmodule/vsrn_.py:6-71): parameters in the reference's module tree
(`g`, `theta`, `phi` = Conv1d(k=1), `W` = Sequential(Conv1d(k=1), BatchNorm1d) with gamma = beta = 0 at init), arithmetic
on the HIP kernels.  The towers keep regions as ROWS ((B, N, D), a kernel-size-1 convolution over (B, D, N) is a Linear
over the last axis), so the reference's permutes disappear."""

import torch
from torch import nn

from .. import ops
from ..settings import SETTINGS


class Rs_GCN(nn.Module):
    def __init__(self, in_channels, inter_channels, bn_layer=True):
        super().__init__()
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        if self.inter_channels is None:
            self.inter_channels = max(in_channels // 2, 1)
        self.g = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        if bn_layer:
            self.W = nn.Sequential(nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1, stride=1, padding=0),
                                   nn.BatchNorm1d(self.in_channels))
            nn.init.constant_(self.W[1].weight, 0)
            nn.init.constant_(self.W[1].bias, 0)
        else:
            self.W = nn.Conv1d(self.inter_channels, self.in_channels, kernel_size=1, stride=1, padding=0)
            nn.init.constant_(self.W.weight, 0)
            nn.init.constant_(self.W.bias, 0)
        self.theta = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self.phi = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self._folded = None

    def train(self, mode=True):
        self._folded = None
        return super().train(mode)

    def load_state_dict(self, *a, **k):
        self._folded = None
        return super().load_state_dict(*a, **k)

    def _weights(self):
        """Parameter preparation (once per eval phase, not hot path): theta | phi | g stacked for one GEMM; eval-mode
        BatchNorm folded into the W convolution,  BN(Wy + b) = (s W) y + (s (b - mean) + beta),  s = gamma / sqrt(var + eps)."""
        if self._folded is None or self._folded[0].device != self.g.weight.device:
            with torch.no_grad():
                w3 = torch.cat([self.theta.weight[:, :, 0], self.phi.weight[:, :, 0], self.g.weight[:, :, 0]], 0).contiguous()
                b3 = torch.cat([self.theta.bias, self.phi.bias, self.g.bias], 0).contiguous()
                if isinstance(self.W, nn.Sequential):
                    conv, bn = self.W[0], self.W[1]
                    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    ww = (conv.weight[:, :, 0] * s[:, None]).contiguous()
                    wb = (s * (conv.bias - bn.running_mean) + bn.bias).contiguous()
                else:
                    ww, wb = self.W.weight[:, :, 0].contiguous(), self.W.bias.contiguous()
            self._folded = (w3, b3, ww, wb)
        return self._folded

    def forward(self, v):
        """v (B, N, D) -> v* (B, N, D)."""
        if self.training:
            return self.forward_train(v)      # batch statistics, gradients: the tape implementation
        w3, b3, ww, wb = self._weights()
        B, N, D = v.shape
        tpg = ops.linear(v.reshape(B * N, D), w3, b3)                        # [B*N, 3C]: theta | phi | g
        y = ops.gcn_relation(tpg, B, N, self.inter_channels)                 # (theta phi^T / N) g
        if not SETTINGS.vsrn_residual_in_epilogue:                           # cross-check form: copy v, accumulate onto the copy (round 2)
            out = v.reshape(B * N, D).clone()
            ops.gemm_acc(y, y.shape[1], B * N, self.inter_channels, ww, wb, out)
        else:
            out = ops.gemm_residual(y, ww, wb, v.reshape(B * N, D))          # v + BN(W y): the residual read by the GEMM's epilogue
        return out.view(B, N, D)

    def forward_train(self, v):
        """The layer on the autograd tape (VSRN.train_emb): BatchNorm with batch statistics (channel = feature, rows = B * N)."""
        from .. import autograd as ag
        B, N, D = v.shape
        C = self.inter_channels
        v2 = v.reshape(B * N, D)
        theta = ag.linear(v2, self.theta.weight[:, :, 0], self.theta.bias).view(B, N, C)
        phi = ag.linear(v2, self.phi.weight[:, :, 0], self.phi.bias).view(B, N, C)
        g = ag.linear(v2, self.g.weight[:, :, 0], self.g.bias).view(B, N, C)
        R = ag.bmm_nt(theta, phi) * (1.0 / N)                                   # R_div_C = theta phi^T / N
        y = ag.bmm_nn(R, g).reshape(B * N, C)
        if isinstance(self.W, nn.Sequential):
            wy = ag.batch_norm_train(ag.linear(y, self.W[0].weight[:, :, 0], self.W[0].bias), self.W[1])
        else:
            wy = ag.linear(y, self.W.weight[:, :, 0], self.W.bias)
        return (wy + v2).view(B, N, D)
