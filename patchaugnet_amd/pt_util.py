"""Shared-MLP building block with the reference's parameter names.

Mirror of ``utils/model_util/pt_util.py:16-41`` (SharedMLP) and ``:98-152`` (_ConvBase) for the only
configuration the hot path uses: 1x1 convolution WITHOUT bias -> BatchNorm -> ReLU, post-activation.
State-dict keys are identical to the reference's (``layer{i}.conv.weight``, ``layer{i}.bn.bn.*``) so its
checkpoints load unchanged.  The 1x1 convolution is evaluated as a matmul over the channel axis.
"""
import torch
import torch.nn as nn

from . import train_ops


class _BN(nn.Module):
    """Holder that reproduces the reference's ``bn.bn`` nesting (pt_util.py:72-90)."""

    def __init__(self, channels, dims):
        super().__init__()
        self.bn = (nn.BatchNorm2d if dims == 2 else nn.BatchNorm1d)(channels)

    def forward(self, x):
        return self.bn(x)


class ConvBNReLU(nn.Module):
    """One SharedMLP layer: ``conv`` (1x1, no bias, kaiming-normal init) -> ``bn`` -> ReLU."""

    def __init__(self, c_in, c_out, dims=2):
        super().__init__()
        self.conv = (nn.Conv2d if dims == 2 else nn.Conv1d)(c_in, c_out, kernel_size=1, bias=False)
        nn.init.kaiming_normal_(self.conv.weight)
        self.bn = _BN(c_out, dims)
        self.activation = nn.ReLU(inplace=True)

    def forward(self, x):
        w = self.conv.weight.flatten(1)                       # (O, C)
        y = torch.matmul(w, x.flatten(2)).view(x.shape[0], w.shape[0], *x.shape[2:])
        return self.activation(self.bn(y))


class SharedMLP(nn.Sequential):
    """args = [c0, c1, ..., cL]; children are named layer0..layer{L-1} like the reference."""

    def __init__(self, args, *, bn=True, dims=2):
        super().__init__()
        assert bn, "the hot path only uses the batch-normalised variant"
        self.channels = list(args)
        for i in range(len(args) - 1):
            self.add_module(f"layer{i}", ConvBNReLU(args[i], args[i + 1], dims))

    def _chain(self, x, pool):
        """On the MI355X, train() and eval(): the whole stack, forward and backward, on the MFMA GEMM kernels of csrc/train_gemm.hip
        (BatchNorm + ReLU applied inside the next layer's operand loader; train(): statistics accumulated in the epilogue,
        eval(): the running statistics)."""
        layers = [train_ops.BNLayer(l.conv.weight, l.bn.bn) for l in self]
        return train_ops.chain_train(x.flatten(2), layers, pool, training=self.training)

    def forward(self, x):
        if train_ops.on_device(x):
            return self._chain(x, 0).view(x.shape[0], self.channels[-1], *x.shape[2:])
        return super().forward(x)                   # CPU form

    def forward_maxpool(self, x):
        """(B, C, m, k) -> max over k of the stack's output, (B, C_out, m) (patch_aug_net.py:236): fused into the last BatchNorm pass."""
        if train_ops.on_device(x):
            return self._chain(x, x.shape[3])
        return super().forward(x).max(dim=3)[0]     # CPU form
