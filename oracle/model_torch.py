"""TEST INFRASTRUCTURE — CPU restatement of one training step of ``chemprop.models.MPNN`` with a regression predictor:
``training_step`` (``models/model.py:148-161``) over ``fingerprint`` (``:126-134``), ``RegressionFFN.train_step``
(``nn/predictors.py:161-169``), the criterion's batch value (``nn/metrics.py:78-127``; MSE ``:137-141``, MAE ``:146-148``,
bounded ``:157-163``; BCE with logits ``:292-295`` behind ``BinaryClassificationFFN.train_step``, ``predictors.py:235-247``) and ``torch.optim.Adam`` (``model.py:208-209``).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.

Built from the other restatements (``dmpnn_torch.forward``, ``agg_torch``, ``ffn_torch.mlp_forward``) plus
``torch.nn.functional.batch_norm`` — the ATen op ``nn.BatchNorm1d`` executes.  Pinned: ``tests/test_model.py`` checks losses,
every gradient and the parameters after two Adam steps against goldens frozen from the EXECUTED reference
(``tests/golden/model/*.npz``, ``tests/golden/make_golden_model.py``: the reference's own ``MPNN.training_step``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import agg_torch, dmpnn_torch, ffn_torch


def criterion(preds: Tensor, targets: Tensor, weights: Optional[Tensor], task_weights: Optional[Tensor], lt_mask: Optional[Tensor],
              gt_mask: Optional[Tensor], kind: str = "mse", v_kl: float = 0.2, eps: float = 1e-8, alpha: float = 0.1) -> Tensor:
    """``mask = targets.isfinite(); targets = targets.nan_to_num(nan=0.0)`` (model.py:152-153), then ``ChempropMetric.update`` +
    ``compute`` on this batch alone (metrics.py:78-127): ``sum(L * w[:, None] * task_weights * mask) / mask.sum()``."""
    mask = targets.isfinite()
    targets = targets.nan_to_num(nan=0.0)
    if kind.startswith("bounded"):  # metrics.py:157-161
        preds = torch.where((preds < targets) & lt_mask, targets, preds)
        preds = torch.where((preds > targets) & gt_mask, targets, preds)
    if kind == "mve":       # MVELoss, metrics.py:203-219, on MveFFN.train_step's [b, t, 2] (predictors.py:173-190)
        mean, var = torch.unbind(preds, dim=-1)
        L = (mean - targets) ** 2 / (2 * var) + (2 * torch.pi * var).log() / 2
    elif kind == "evidential":   # EvidentialLoss, metrics.py:222-262, on EvidentialFFN.train_step's [b, t, 4] (predictors.py:193-212)
        mean, v, alpha, beta = torch.unbind(preds, dim=-1)
        residuals = targets - mean
        twoBlambda = 2 * beta * (1 + v)
        L_nll = (0.5 * (torch.pi / v).log() - alpha * twoBlambda.log() + (alpha + 0.5) * torch.log(v * residuals**2 + twoBlambda)
                 + torch.lgamma(alpha) - torch.lgamma(alpha + 0.5))
        L = L_nll + v_kl * ((2 * v + alpha) * residuals.abs() - eps)
    elif kind == "quantile":     # QuantileLoss, metrics.py:589-610, on QuantileFFN.train_step's [b, t, 2] = (mean, interval) (predictors.py:215-232)
        mean, interval = torch.unbind(preds, dim=-1)
        bounds = torch.tensor([-1 / 2, 1 / 2]).view(-1, 1, 1)
        tau = torch.tensor([[alpha / 2, 1 - alpha / 2], [alpha / 2 - 1, -alpha / 2]]).view(2, 2, 1, 1)
        L = (tau * (targets - (mean + bounds * interval))).amax(0).sum(0)
    elif kind == "ce":      # metrics.py:298-304 on the logits [b, t, c] of MulticlassClassificationFFN.train_step (predictors.py:313-314)
        L = F.cross_entropy(preds.transpose(1, 2), targets.long(), reduction="none")
    elif kind == "bce":     # metrics.py:292-295 on the raw logits of BinaryClassificationFFN.train_step (predictors.py:246-247)
        L = F.binary_cross_entropy_with_logits(preds, targets, reduction="none")
    else:
        L = (preds - targets).abs() if kind.endswith("mae") else F.mse_loss(preds, targets, reduction="none")
    w = torch.ones(targets.shape[0]) if weights is None else weights
    tw = torch.ones(1, targets.shape[1]) if task_weights is None else task_weights.view(1, -1)
    L = L * w.view(-1, 1) * tw * mask
    return L.sum() / mask.sum()


class Model:
    """Parameters (leaf tensors keyed like the reference's state dict) + the forward of ``MPNN`` in ``train()`` / ``eval()``."""

    def __init__(self, state: dict, cfg: dict):
        self.cfg = cfg
        self.p = {}
        self.buf = {}
        for k, v in state.items():
            t = torch.as_tensor(v).clone()
            if k.endswith(("running_mean", "running_var", "num_batches_tracked", "task_weights", "criterion.bounds", "criterion.tau")):
                self.buf[k] = t
            else:
                self.p[k] = t.float().requires_grad_(True)
        self.training = True

    def parameters(self):
        return list(self.p.values())

    def forward(self, bmg) -> Tensor:
        cfg, p = self.cfg, self.p
        mpc = cfg["mp"]
        w = dmpnn_torch.MPWeights(p["message_passing.W_i.weight"], p["message_passing.W_h.weight"], p["message_passing.W_o.weight"],
                                  p["message_passing.W_o.bias"], p.get("message_passing.W_i.bias"), p.get("message_passing.W_h.bias"))
        Hv = dmpnn_torch.forward(bmg.V, bmg.E, bmg.edge_index, bmg.rev_edge_index, w, depth=mpc.get("depth", 3),
                                 activation=mpc.get("activation", "relu"))
        H = dict(norm=agg_torch.norm, mean=agg_torch.mean, sum=agg_torch.sum_)[cfg["agg"]](Hv, bmg.batch)       # model.py:131
        if cfg["bn"]:                                                                                             # model.py:132
            H = F.batch_norm(H, self.buf["bn.running_mean"], self.buf["bn.running_var"], p["bn.weight"], p["bn.bias"], self.training, 0.1, 1e-5)
            if self.training:
                self.buf["bn.num_batches_tracked"] += 1
        ws, bs, i = [], [], 0
        while f"predictor.ffn.{i}.{0 if i == 0 else 2}.weight" in p:
            j = 0 if i == 0 else 2
            ws.append(p[f"predictor.ffn.{i}.{j}.weight"])
            bs.append(p[f"predictor.ffn.{i}.{j}.bias"])
            i += 1
        P = ffn_torch.mlp_forward(H, ws, bs, cfg["ffn"].get("activation", "relu"))                              # predictors.py:166-169
        if cfg.get("predictor") == "multiclass":                                                                  # predictors.py:313-314
            P = P.reshape(P.shape[0], -1, cfg["ffn"]["n_classes"])
        if cfg.get("predictor") == "mve":                                                                         # predictors.py:177-188
            mean, var = torch.chunk(P, 2, 1)
            P = torch.stack((mean, F.softplus(var)), dim=2)
        if cfg.get("predictor") == "evidential":                                                                  # predictors.py:197-210
            mean, v, alpha, beta = torch.chunk(P, 4, 1)
            P = torch.stack((mean, F.softplus(v), F.softplus(alpha) + 1, F.softplus(beta)), dim=2)
        if cfg.get("predictor") == "quantile":                                                                    # predictors.py:219-230
            lower, upper = torch.chunk(P, 2, 1)
            P = torch.stack(((lower + upper) / 2, upper - lower), dim=2)
        return P

    def loss(self, bmg, targets, weights, lt_mask, gt_mask) -> Tensor:
        return criterion(self.forward(bmg), targets, weights, self.buf.get("predictor.criterion.task_weights"), lt_mask, gt_mask,
                         self.cfg.get("criterion", "mse"), alpha=float(self.cfg.get("alpha", 0.1)))

    def state(self) -> dict:
        out = {k: v.detach().clone() for k, v in self.p.items()}
        out.update({k: v.clone() for k, v in self.buf.items()})
        return out


def train_steps(state: dict, cfg: dict, bmg, targets, weights, lt_mask, gt_mask, lr: float, steps: int):
    """``steps`` times: ``opt.zero_grad(); loss = training_step(batch); loss.backward(); opt.step()`` with ``torch.optim.Adam``.
    Returns ``(model, [loss], [{name: grad}])``."""
    m = Model(state, cfg)
    names = list(m.p.keys())
    opt = torch.optim.Adam(m.parameters(), lr)
    losses, grads = [], []
    for _ in range(steps):
        opt.zero_grad()
        l = m.loss(bmg, targets, weights, lt_mask, gt_mask)
        l.backward()
        losses.append(float(l.detach()))
        grads.append({k: (m.p[k].grad.clone() if m.p[k].grad is not None else torch.zeros_like(m.p[k])) for k in names})
        opt.step()
    return m, losses, grads
