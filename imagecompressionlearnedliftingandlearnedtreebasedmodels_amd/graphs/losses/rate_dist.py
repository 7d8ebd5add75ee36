"""Rate-distortion loss on the HIP reductions -- mirrors graphs/losses/rate_dist.py:14-71 of the reference.

``forward3(x, x_hat, rate1, rate2list)``: mse = mean((x - x_hat)^2); rate = sum(bits) / numel(x) * 3 (bits per PIXEL,
numel counts the 3 channels); loss = rate1 + rate2 + lambda * mse (rate_dist.py:35-42).  Sums are accumulated in
float64 on the device (lldwt_sq_err_sum / lldwt_sum).
"""
import torch
from torch import nn

from ... import ops


def _sum(t):
    acc = torch.zeros(1, dtype=torch.float64, device=t.device)
    ops.sum_into(t.contiguous(), acc)
    return acc


class TrainRDLoss(nn.Module):
    def __init__(self, lambda_):
        super().__init__()
        self.lambda_ = lambda_

    def _mse(self, x, x_hat):
        acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        ops.sq_err_sum(x.contiguous(), x_hat.contiguous(), acc)
        return (acc / x.numel()).float()[0]

    def _terms(self, x, x_hat, rate1, rate2list):
        self.mse = self._mse(x, x_hat)
        n = x.numel()
        self.rate1 = (_sum(rate1) / n * 3).float()[0]
        r2 = torch.zeros(1, dtype=torch.float64, device=x.device)
        for r in rate2list:
            ops.sum_into(r.contiguous(), r2)
        self.rate2 = (r2 / n * 3).float()[0]

    def forward3(self, x, x_hat, rate1, rate2list):
        self._terms(x, x_hat, rate1, rate2list)
        self.loss = self.rate1 + self.rate2 + self.lambda_ * self.mse
        return self.loss, self.mse, self.rate1, self.rate2

    def forward2(self, x, x_hat, rate1, rate2):
        return self.forward3(x, x_hat, rate1, [rate2])

    def forward(self, x, x_hat, rate):
        loss, mse, r1, _ = self.forward3(x, x_hat, rate, [])
        self.rate = r1
        return loss, mse, r1


    def forward3_train(self, x, x_hat, rate1, rate2list):
        """Differentiable forward3 (rate_dist.py:35-42): float64 device sums with hand-written gradients."""
        from ... import autograd as ag
        n = x.numel()
        self.mse = (ag.SqErrSumFn.apply(x, x_hat) / n)[0]
        self.rate1 = (ag.SumFn.apply(rate1) / n * 3)[0]
        r2 = 0
        for r in rate2list:
            r2 = r2 + ag.SumFn.apply(r)
        self.rate2 = (r2 / n * 3)[0]
        self.loss = self._combine()
        return self.loss, self.mse, self.rate1, self.rate2

    def _combine(self):
        return self.rate1 + self.rate2 + self.lambda_ * self.mse


class TrainDLoss(TrainRDLoss):
    """lambda * MSE only (rate_dist.py:45-71); the rates are still reported."""

    def forward3(self, x, x_hat, rate1, rate2list):
        self._terms(x, x_hat, rate1, rate2list)
        self.loss = self.lambda_ * self.mse
        return self.loss, self.mse, self.rate1, self.rate2

    def _combine(self):
        return self.lambda_ * self.mse
