"""f4: the model around the block — ``chemprop.models.MPNN`` (``models/model.py:60-161``) with a regression
predictor (``nn/predictors.py:101-169``) — and its training step as ONE C call.

:class:`MPNN` mirrors the reference's attribute names (``message_passing``, ``agg``, ``bn``, ``predictor``; ``predictor.ffn``,
``predictor.criterion.task_weights``), so a state dict moves between the two for those keys, and its ``forward`` /
``fingerprint`` / ``loss`` run on the engine's kernels through autograd like any torch module (the drop-in path).

:class:`FusedTrainer` is the MI355X-native form of ``training_step`` + ``optimizer.step()`` (``model.py:148-161,208-231``):
every parameter lives in ONE flat buffer, every gradient in another (``distributed.GradSync`` / ``optim.FlatAdam``), and a
step is ``dmpnn_train_step`` — K0, the block's forward with kept tensors, aggregation, batch norm, the predictor's layers, the
loss and its gradient, the backward pass of all of it and the Adam update are enqueued by one call (``csrc/dmpnn_head.hip``);
nothing of the step runs in Python or in ATen.  With more than one rank the gradient all-reduce sits between the backward pass
and the update (the step is then two calls around ``GradSync.allreduce``).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Iterable, Optional

import torch
from torch import Tensor, nn

from . import _lib, engine
from .agg import MODES, Aggregation, NormAggregation, note_batch
from .distributed import GradSync, force_collective
from .ffn import MLP
from .nn import BondMessagePassing, classify_activation
from .optim import FlatAdam

__all__ = ["MSE", "MAE", "BCE", "CE", "MVE", "Evidential", "Quantile", "RegressionFFN", "BinaryClassificationFFN", "MulticlassClassificationFFN", "MveFFN", "EvidentialFFN", "QuantileFFN",
           "MPNN", "FusedTrainer", "masked_loss"]


def masked_loss(preds: Tensor, targets: Tensor, weights: Optional[Tensor] = None, task_weights: Optional[Tensor] = None,
                lt_mask: Optional[Tensor] = None, gt_mask: Optional[Tensor] = None, kind: str = "mse", v_kl: float = 0.2, eps: float = 1e-8,
                alpha: float = 0.1) -> Tensor:
    """``ChempropMetric.update`` + ``compute`` on one batch (``nn/metrics.py:78-127``) with the masking of
    ``MPNN.training_step`` (``models/model.py:152-156``): torch ops, differentiable — the module path's criterion."""
    mask = targets.isfinite()
    targets = targets.nan_to_num(nan=0.0)
    if kind in ("bce", "ce", "mve", "evidential", "quantile"):
        lt_mask = gt_mask = None     # (only the Bounded* criteria apply the masks: metrics.py:157-177)
    if kind in ("mve", "evidential", "quantile"):   # preds [b, t, 2 | 4]: what MveFFN / EvidentialFFN / QuantileFFN.train_step stack (predictors.py:173-232)
        if kind == "mve":               # MVELoss, metrics.py:203-219
            mean, var = torch.unbind(preds, dim=-1)
            L = (mean - targets) ** 2 / (2 * var) + (2 * torch.pi * var).log() / 2
        elif kind == "quantile":        # QuantileLoss, metrics.py:589-610
            mean, interval = torch.unbind(preds, dim=-1)
            bounds = torch.tensor([-0.5, 0.5], device=preds.device).view(-1, 1, 1)
            tau = torch.tensor([[alpha / 2, 1 - alpha / 2], [alpha / 2 - 1, -alpha / 2]], device=preds.device).view(2, 2, 1, 1)
            L = (tau * (targets - (mean + bounds * interval))).amax(0).sum(0)
        else:                           # EvidentialLoss, metrics.py:222-262
            mean, v, alpha, beta = torch.unbind(preds, dim=-1)
            residuals = targets - mean
            two_b_lambda = 2 * beta * (1 + v)
            L_nll = (0.5 * (torch.pi / v).log() - alpha * two_b_lambda.log() + (alpha + 0.5) * torch.log(v * residuals**2 + two_b_lambda)
                     + torch.lgamma(alpha) - torch.lgamma(alpha + 0.5))
            L = L_nll + v_kl * ((2 * v + alpha) * residuals.abs() - eps)
        w = torch.ones(targets.shape[0], dtype=torch.float, device=targets.device) if weights is None else weights
        tw = 1.0 if task_weights is None else task_weights.view(1, -1)
        return (L * w.view(-1, 1) * tw * mask).sum() / mask.sum()
    if kind == "ce":                 # preds [b, t, c] logits, targets [b, t] class indices (metrics.py:298-304)
        L = torch.nn.functional.cross_entropy(preds.transpose(1, 2), targets.long(), reduction="none")
        w = torch.ones(targets.shape[0], dtype=torch.float, device=targets.device) if weights is None else weights
        tw = 1.0 if task_weights is None else task_weights.view(1, -1)
        return (L * w.view(-1, 1) * tw * mask).sum() / mask.sum()
    if lt_mask is not None:
        preds = torch.where((preds < targets) & lt_mask, targets, preds)
    if gt_mask is not None:
        preds = torch.where((preds > targets) & gt_mask, targets, preds)
    if kind == "bce":     # nn/metrics.py:292-295 (on logits: the classification predictor's train_step, predictors.py:246-247)
        L = torch.nn.functional.binary_cross_entropy_with_logits(preds, targets, reduction="none")
    else:
        L = (preds - targets).abs() if kind == "mae" else torch.nn.functional.mse_loss(preds, targets, reduction="none")
    w = torch.ones(targets.shape[0], dtype=torch.float, device=targets.device) if weights is None else weights
    tw = 1.0 if task_weights is None else task_weights.view(1, -1)
    L = L * w.view(-1, 1) * tw * mask
    return L.sum() / mask.sum()


class MSE(nn.Module):
    """The criterion's state (``nn/metrics.py:60-76``: a ``task_weights`` buffer of shape ``[1, t]``) and its batch value."""

    kind = "mse"

    def __init__(self, task_weights=1.0):
        super().__init__()
        self.register_buffer("task_weights", torch.as_tensor(task_weights, dtype=torch.float).view(1, -1))

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        t = targets if mask is None else torch.where(mask, targets, torch.full_like(targets, float("nan")))
        return masked_loss(preds, t, weights, self.task_weights, lt_mask, gt_mask, self.kind)


class MAE(MSE):
    kind = "mae"


class BCE(MSE):
    """``chemprop.nn.metrics.BCELoss`` (``metrics.py:292-295``): binary cross entropy on LOGITS; no bounds."""

    kind = "bce"

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        return super().forward(preds, targets, mask, weights, None, None)


class RegressionFFN(nn.Module):
    """``chemprop.nn.predictors.RegressionFFN`` (``predictors.py:101-169``): ``ffn = MLP.build(input_dim, n_tasks, hidden_dim,
    n_layers, dropout, activation)``, criterion MSE, identity output transform while training."""

    n_targets = 1

    def __init__(self, n_tasks: int = 1, input_dim: int = 300, hidden_dim: int = 300, n_layers: int = 1, dropout: float = 0.0,
                 activation="relu", criterion: Optional[nn.Module] = None, task_weights: Optional[Tensor] = None):
        super().__init__()
        self.hparams = dict(n_tasks=n_tasks, input_dim=input_dim, hidden_dim=hidden_dim, n_layers=n_layers, dropout=dropout,
                            activation=activation, cls=self.__class__)
        self.ffn = MLP.build(input_dim, n_tasks * self.n_targets, hidden_dim, n_layers, dropout, activation)
        self.criterion = criterion if criterion is not None else MSE(torch.ones(n_tasks) if task_weights is None else task_weights)
        self.output_transform = nn.Identity()

    @property
    def input_dim(self) -> int:
        return self.ffn.input_dim

    @property
    def output_dim(self) -> int:
        return self.ffn.output_dim

    @property
    def n_tasks(self) -> int:
        return self.output_dim // self.n_targets

    def forward(self, Z: Tensor) -> Tensor:
        return self.output_transform(self.ffn(Z))

    train_step = forward


class CE(MSE):
    """``chemprop.nn.metrics.CrossEntropyLoss`` (``metrics.py:298-304``): ``preds [b, t, c]`` logits against class indices; no bounds."""

    kind = "ce"

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        return super().forward(preds, targets, mask, weights, None, None)


class MVE(MSE):
    """``chemprop.nn.metrics.MVELoss`` (``metrics.py:203-219``): ``preds [b, t, 2]`` = (mean, variance); no bounds."""

    kind = "mve"

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        return super().forward(preds, targets, mask, weights, None, None)


class Evidential(MSE):
    """``chemprop.nn.metrics.EvidentialLoss`` (``metrics.py:222-262``): ``preds [b, t, 4]`` = (mean, v, alpha, beta); no bounds."""

    kind = "evidential"

    def __init__(self, task_weights=1.0, v_kl: float = 0.2, eps: float = 1e-8):
        super().__init__(task_weights)
        self.v_kl, self.eps = v_kl, eps

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        t = targets if mask is None else torch.where(mask, targets, torch.full_like(targets, float("nan")))
        return masked_loss(preds, t, weights, self.task_weights, None, None, self.kind, self.v_kl, self.eps)


class Quantile(MSE):
    """``chemprop.nn.metrics.QuantileLoss`` (``metrics.py:589-610``): ``preds [b, t, 2]`` = (mean, interval); no bounds."""

    kind = "quantile"

    def __init__(self, task_weights=1.0, alpha: float = 0.1):
        super().__init__(task_weights)
        self.alpha = alpha
        # (the reference's buffers, metrics.py:594-600: the state dicts match)
        self.register_buffer("bounds", torch.tensor([-1 / 2, 1 / 2]).view(-1, 1, 1))
        self.register_buffer("tau", torch.tensor([[alpha / 2, 1 - alpha / 2], [alpha / 2 - 1, -alpha / 2]]).view(2, 2, 1, 1))

    def forward(self, preds, targets, mask=None, weights=None, lt_mask=None, gt_mask=None):
        t = targets if mask is None else torch.where(mask, targets, torch.full_like(targets, float("nan")))
        return masked_loss(preds, t, weights, self.task_weights, None, None, self.kind, alpha=self.alpha)


class _MultiTargetFFN(RegressionFFN):
    """A regression predictor with ``n_targets`` values per task (``predictors.py:132``: the MLP is ``n_tasks * n_targets`` wide)."""

    _default_criterion = MSE

    def __init__(self, n_tasks: int = 1, input_dim: int = 300, hidden_dim: int = 300, n_layers: int = 1, dropout: float = 0.0,
                 activation="relu", criterion: Optional[nn.Module] = None, task_weights: Optional[Tensor] = None):
        super().__init__(n_tasks, input_dim, hidden_dim, n_layers, dropout, activation,
                         criterion if criterion is not None else self._default_criterion(torch.ones(n_tasks) if task_weights is None else task_weights))


class MveFFN(_MultiTargetFFN):
    """``chemprop.nn.predictors.MveFFN`` (``predictors.py:173-190``): an MLP ``2 n_tasks`` wide, chunked into means and raw variances
    (``softplus``), stacked ``[b, t, 2]``; ``train_step`` is ``forward``.  (The output transform is the identity here, as the
    reference's ``UnscaleTransform`` is while training.)"""

    n_targets = 2
    _default_criterion = MVE

    def forward(self, Z: Tensor) -> Tensor:
        mean, var = torch.chunk(self.ffn(Z), 2, 1)
        return torch.stack((mean, torch.nn.functional.softplus(var)), dim=2)

    train_step = forward


class EvidentialFFN(_MultiTargetFFN):
    """``chemprop.nn.predictors.EvidentialFFN`` (``predictors.py:193-212``): ``4 n_tasks`` wide — mean | v | alpha | beta,
    ``v = softplus``, ``alpha = softplus + 1``, ``beta = softplus``, stacked ``[b, t, 4]``."""

    n_targets = 4
    _default_criterion = Evidential

    def forward(self, Z: Tensor) -> Tensor:
        sp = torch.nn.functional.softplus
        mean, v, alpha, beta = torch.chunk(self.ffn(Z), 4, 1)
        return torch.stack((mean, sp(v), sp(alpha) + 1, sp(beta)), dim=2)

    train_step = forward


class QuantileFFN(_MultiTargetFFN):
    """``chemprop.nn.predictors.QuantileFFN`` (``predictors.py:215-232``): ``2 n_tasks`` wide — lower | upper bounds, stacked as
    ``(mean, interval) [b, t, 2]``."""

    n_targets = 2
    _default_criterion = Quantile

    def forward(self, Z: Tensor) -> Tensor:
        lower, upper = torch.chunk(self.ffn(Z), 2, 1)
        return torch.stack(((lower + upper) / 2, upper - lower), dim=2)

    train_step = forward


class MulticlassClassificationFFN(RegressionFFN):
    """``chemprop.nn.predictors.MulticlassClassificationFFN`` (``predictors.py:271-314``): an MLP ``n_tasks * n_classes`` wide;
    ``forward`` predicts class probabilities ``[b, t, c]`` (softmax), ``train_step`` hands the logits ``[b, t, c]`` to ``CrossEntropyLoss``."""

    def __init__(self, n_classes: int, n_tasks: int = 1, input_dim: int = 300, hidden_dim: int = 300, n_layers: int = 1, dropout: float = 0.0,
                 activation="relu", criterion: Optional[nn.Module] = None, task_weights: Optional[Tensor] = None):
        super().__init__(n_tasks * n_classes, input_dim, hidden_dim, n_layers, dropout, activation,
                         criterion if criterion is not None else CE(torch.ones(n_tasks) if task_weights is None else task_weights))
        self.n_classes = n_classes
        # (predictors.py:271-314 records n_tasks AND n_classes; the base class saw their product — `cls(**hparams)` must rebuild this width)
        self.hparams.update(n_tasks=n_tasks, n_classes=n_classes)

    @property
    def n_tasks(self) -> int:
        return self.output_dim // (self.n_targets * self.n_classes)

    def forward(self, Z: Tensor) -> Tensor:
        return self.ffn(Z).reshape(Z.shape[0], -1, self.n_classes).softmax(-1)

    def train_step(self, Z: Tensor) -> Tensor:
        return self.ffn(Z).reshape(Z.shape[0], -1, self.n_classes)


class BinaryClassificationFFN(RegressionFFN):
    """``chemprop.nn.predictors.BinaryClassificationFFN`` (``predictors.py:235-247``): the same MLP; ``forward`` predicts
    probabilities (``sigmoid``), ``train_step`` hands the raw logits to ``BCELoss``."""

    def __init__(self, n_tasks: int = 1, input_dim: int = 300, hidden_dim: int = 300, n_layers: int = 1, dropout: float = 0.0,
                 activation="relu", criterion: Optional[nn.Module] = None, task_weights: Optional[Tensor] = None):
        super().__init__(n_tasks, input_dim, hidden_dim, n_layers, dropout, activation,
                         criterion if criterion is not None else BCE(torch.ones(n_tasks) if task_weights is None else task_weights))

    def forward(self, Z: Tensor) -> Tensor:
        return self.ffn(Z).sigmoid()

    def train_step(self, Z: Tensor) -> Tensor:
        return self.ffn(Z)


class MPNN(nn.Module):
    """``chemprop.models.MPNN`` (``models/model.py:60-146``) without Lightning: the four sub-modules under the reference's
    names and ``fingerprint`` / ``forward``; ``loss(batch)`` is the arithmetic of ``training_step`` (``model.py:148-161``)."""

    def __init__(self, message_passing: nn.Module, agg: nn.Module, predictor: nn.Module, batch_norm: bool = False):
        super().__init__()
        if message_passing.output_dim != predictor.input_dim:
            raise ValueError(f"message passing output dim {message_passing.output_dim} != predictor input dim {predictor.input_dim}")
        self.message_passing = message_passing
        self.agg = agg
        self.bn = nn.BatchNorm1d(message_passing.output_dim) if batch_norm else nn.Identity()
        self.predictor = predictor

    @property
    def criterion(self):
        return self.predictor.criterion

    def fingerprint(self, bmg, V_d: Optional[Tensor] = None, X_d: Optional[Tensor] = None) -> Tensor:
        H = self.bn(self.agg(self.message_passing(bmg, V_d), bmg.batch))
        return H if X_d is None else torch.cat((H, X_d), dim=1)

    def forward(self, bmg, V_d: Optional[Tensor] = None, X_d: Optional[Tensor] = None) -> Tensor:
        return self.predictor(self.fingerprint(bmg, V_d, X_d))

    def loss(self, bmg, targets: Tensor, weights: Optional[Tensor] = None, lt_mask: Optional[Tensor] = None,
             gt_mask: Optional[Tensor] = None, V_d: Optional[Tensor] = None, X_d: Optional[Tensor] = None) -> Tensor:
        if X_d is None and torch.is_grad_enabled() and self.training:
            # the module path of a training step: the block through its own autograd node, everything behind it — aggregation, batch
            # norm, predictor, criterion AND their backward pass — as ONE more (head_loss); None: the head kernels do not take this model
            l = head_loss(self, self.message_passing(bmg, V_d), bmg.batch, len(bmg), targets, weights, lt_mask, gt_mask)
            if l is not None:
                return l
        preds = self.predictor.train_step(self.fingerprint(bmg, V_d, X_d))
        c = self.criterion
        return masked_loss(preds, targets, weights, getattr(c, "task_weights", None), lt_mask, gt_mask, getattr(c, "kind", "mse"),
                           float(getattr(c, "v_kl", 0.2)), float(getattr(c, "eps", 1e-8)), float(getattr(c, "alpha", 0.1)))


def _mro_names(obj) -> set:
    return {c.__name__ for c in type(obj).__mro__}


def aggregation_mode(agg) -> Optional[str]:
    """``"sum" | "mean" | "norm"`` for this package's aggregations (``.mode``) and for the reference's own classes / their HIP
    subclasses (``chemprop/nn/agg.py:66-113``, told by class name); ``None`` for anything else (attentive, custom)."""
    mode = getattr(agg, "mode", None)
    if mode in MODES:
        return mode
    names = _mro_names(agg)
    for cls, m in (("NormAggregation", "norm"), ("MeanAggregation", "mean"), ("SumAggregation", "sum")):  # (Norm derives from Sum)
        if cls in names:
            return m
    return None


def criterion_kind(crit) -> tuple[Optional[str], bool]:
    """``(kind, bounded)``: ``"mse" | "mae"`` and whether the criterion applies ``lt_mask`` / ``gt_mask``.  This package's criteria
    carry ``.kind`` (bounded iff masks are handed over); the reference's are told by class name — ``MSE`` / ``MAE`` ignore the masks
    (``nn/metrics.py:139-150``), ``BoundedMSE`` / ``BoundedMAE`` apply them (``:158-177``); RMSE, MVE, ... are not built in."""
    kind = getattr(crit, "kind", None)
    if kind in _lib.LOSS:
        return kind, kind in ("mse", "mae")
    names = _mro_names(crit)
    if "RMSE" in names:
        return None, False
    if "MVELoss" in names:            # nn/metrics.py:203-219
        return "mve", False
    if "EvidentialLoss" in names:     # nn/metrics.py:222-262
        return "evidential", False
    if "QuantileLoss" in names and "PointQuantileLoss" not in names:   # nn/metrics.py:589-610 (the interval form; the point form takes one value per task)
        return "quantile", False
    if "BCELoss" in names:       # nn/metrics.py:292-295 (binary classification: chemprop's second task type)
        return "bce", False
    if "CrossEntropyLoss" in names:   # nn/metrics.py:298-304 (multiclass: logits [b, t, c] against class indices)
        return "ce", False
    for cls, k in (("MSE", "mse"), ("MAE", "mae")):
        if cls in names:
            return k, "BoundedMixin" in names
    return None, False


class HeadSpec:
    """What ``dmpnn_head`` needs to know of the model around the block — aggregation, batch norm, the predictor's layers and
    activation, the criterion — taken once from the modules; raises ``NotImplementedError`` for what the head kernels do not
    implement.  Shared by :class:`FusedTrainer` (the whole step as one C call) and :func:`head_loss` (the module path's ONE
    autograd node for everything behind the block)."""

    def __init__(self, model):
        agg, pred = model.agg, model.predictor
        mode = aggregation_mode(agg)
        if mode is None:
            raise NotImplementedError(f"sum / mean / norm aggregation (got {type(agg).__name__})")
        blocks = list(pred.ffn)
        if len(blocks) > _lib.MAX_FFN_LAYERS:
            raise NotImplementedError(f"at most {_lib.MAX_FFN_LAYERS} predictor layers")
        f_act, f_slope = "none", 0.0
        for b in blocks[1:]:
            code, sl, _ = classify_activation(b[0])
            if code in ("custom", "prelu") or b[1].p > 0:
                raise NotImplementedError("predictor with a built-in activation (not PReLU) and dropout 0")
            f_act, f_slope = code, sl
        # (the reference's UnscaleTransform IS the identity in training mode, transforms.py:45-50: what a scaled regression run carries)
        if not (isinstance(pred.output_transform, nn.Identity) or "UnscaleTransform" in _mro_names(pred.output_transform)):
            raise NotImplementedError("the output transform is the identity while training (predictors.py:166-169)")
        kind, self.bounded = criterion_kind(pred.criterion)
        if kind is None:
            raise NotImplementedError("MSE / MAE criterion (bounded or not), BCE, cross entropy, MVE, evidential or quantile")
        # values per task (predictors.py: n_targets): 1, or — round 6 — the mean-variance / evidential pairs of predictor and criterion
        # (MveFFN + MVELoss: 2, EvidentialFFN + EvidentialLoss: 4, QuantileFFN + QuantileLoss: 2; the predictors' transforms of the
        # raw outputs live in the criterion kernel); Dirichlet heads train through torch ops
        self.n_targets = int(getattr(pred, "n_targets", 1))
        want_targets = {"mve": 2, "evidential": 4, "quantile": 2}.get(kind, 1)
        want_pred = {"mve": "MveFFN", "evidential": "EvidentialFFN", "quantile": "QuantileFFN"}.get(kind)
        if self.n_targets != want_targets or (want_pred is not None and want_pred not in _mro_names(pred)):
            raise NotImplementedError("one value per task, or MveFFN with MVELoss / EvidentialFFN with EvidentialLoss / QuantileFFN with QuantileLoss")
        if want_targets > 1 and int(blocks[-1][-1].out_features) % want_targets:
            raise NotImplementedError("the output width must be n_tasks * n_targets")
        self.v_kl, self.eps = float(getattr(pred.criterion, "v_kl", 0.2)), float(getattr(pred.criterion, "eps", 1e-8))
        self.q_alpha = float(getattr(pred.criterion, "alpha", 0.1))
        # multiclass (predictors.py:271-314): the output layer holds n_classes logits per task, the criterion is the cross entropy
        # over them — every other pairing of a class dimension and a criterion (Dirichlet heads, ...) trains through torch ops
        self.n_classes = int(getattr(pred, "n_classes", 0) or 0)
        if (kind == "ce") != (self.n_classes >= 2):
            raise NotImplementedError("a multiclass predictor (n_classes >= 2) with CrossEntropyLoss, or neither")
        if kind == "ce" and int(blocks[-1][-1].out_features) % self.n_classes:
            raise NotImplementedError("the output width must be n_tasks * n_classes")
        self.agg_mode, self.agg_norm = MODES[mode], float(getattr(agg, "norm", 1.0))
        self.f_act, self.f_slope, self.kind = f_act, f_slope, kind
        self.layers = [b[-1] for b in blocks]
        self.bn = model.bn if isinstance(model.bn, nn.BatchNorm1d) else None
        if self.bn is not None and (self.bn.momentum is None or not self.bn.affine or not self.bn.track_running_stats):
            raise NotImplementedError("nn.BatchNorm1d with a fixed momentum, affine, running statistics")
        self.criterion = pred.criterion

    @property
    def n_out(self) -> int:
        """Width of the output layer (= columns of the predictions the kernels write)."""
        return int(self.layers[-1].out_features)

    @property
    def n_tasks(self) -> int:
        """Columns of ``targets``: the output width, or — multiclass — that over ``n_classes``."""
        return self.n_out // self.n_classes if self.n_classes >= 2 else self.n_out // self.n_targets

    def params(self) -> list:
        """The head's parameters in the order ``fill`` asks ``gptr`` about them."""
        ps = [] if self.bn is None else [self.bn.weight, self.bn.bias]
        for lin in self.layers:
            ps.append(lin.weight)
            if lin.bias is not None:
                ps.append(lin.bias)
        return ps

    def fill(self, h, nV: int, n_mols: int, d_out: int, batch: Tensor, T: Tensor, weights, lt_mask, gt_mask, gptr, bn_training: bool = True) -> list:
        """Fill ``h`` (a ``_lib.HeadArgs``) but for ``preds / loss_out / gHv / ws``; ``gptr(param)`` gives the address the gradient of
        ``param`` goes to (``None``: not wanted).  Returns the tensors that must stay alive until the call has been enqueued."""
        dev = T.device
        h.n_atoms, h.n_mols, h.d_h = nV, n_mols, d_out
        h.batch = batch.data_ptr()
        h.agg_mode, h.agg_norm = self.agg_mode, self.agg_norm
        bn = self.bn
        if bn is not None:
            h.bn_weight, h.bn_bias = bn.weight.data_ptr(), bn.bias.data_ptr()
            h.bn_running_mean, h.bn_running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            h.bn_eps, h.bn_momentum, h.bn_training = float(bn.eps), float(bn.momentum), 1 if bn_training else 0
            h.g_bn_weight, h.g_bn_bias = gptr(bn.weight), gptr(bn.bias)
            nbt = bn.num_batches_tracked
            if bn_training and nbt is not None and nbt.dtype == torch.int64 and nbt.device == dev:
                h.bn_num_batches_tracked = nbt.data_ptr()  # (counted by the batch-norm kernel: no launch of its own)
        h.n_layers, h.act, h.act_slope = len(self.layers), _lib.ACT[self.f_act], float(self.f_slope)
        h.dims[0] = d_out
        for l, lin in enumerate(self.layers):
            h.W[l], h.b[l] = lin.weight.data_ptr(), (None if lin.bias is None else lin.bias.data_ptr())
            h.dims[l + 1] = lin.out_features
            h.gW[l], h.gb[l] = gptr(lin.weight), gptr(lin.bias)
        h.loss = _lib.LOSS[self.kind]
        h.n_classes = self.n_classes
        h.evid_v_kl, h.evid_eps, h.quantile_alpha = self.v_kl, self.eps, self.q_alpha
        h.targets = T.data_ptr()
        keep = []
        if weights is not None:
            wt = engine._f32c(weights.reshape(-1, 1), "weights").reshape(-1).contiguous()
            h.weights = wt.data_ptr()
            keep.append(wt)
        tw = getattr(self.criterion, "task_weights", None)
        if tw is not None:
            tw = tw.reshape(-1).float()
            if tw.numel() == 1 and self.n_tasks > 1:  # (task_weights = 1.0 broadcasts over the tasks, metrics.py:69-70)
                tw = tw.expand(self.n_tasks)
            tw = tw.contiguous()
            h.task_weights = tw.data_ptr()
            keep.append(tw)
        for name, m in (("lt_mask", lt_mask), ("gt_mask", gt_mask)):
            if m is not None and self.bounded:   # (the reference's plain MSE / MAE ignore the masks: nn/metrics.py:139-150)
                m8 = m.to(torch.uint8).contiguous()
                setattr(h, name, m8.data_ptr())
                keep.append(m8)
        return keep


class _HeadLoss(torch.autograd.Function):
    """Everything behind the block — aggregation, batch norm, the predictor's layers, the criterion — AND its backward pass as ONE
    ``dmpnn_head`` call in the forward of one autograd node: the loss comes back with the gradient of ``H_v`` and of every head
    parameter already computed (for a unit upstream gradient); ``backward`` scales them by the upstream gradient (one launch over
    one flat buffer) and hands them to autograd.  Replaces ~40 torch launches and autograd nodes of the module path."""

    @staticmethod
    def forward(ctx, spec, Hv, batch, n_mols, T, weights, lt_mask, gt_mask, *params):
        lib = _lib.load()
        dev = Hv.device
        Hv = engine._f32c(Hv, "H_v")
        nV, d_out = int(Hv.shape[0]), int(Hv.shape[1])
        # one flat buffer for the head's parameter gradients (16-byte aligned pieces) + gH_v
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        flat = torch.empty(n + nV * d_out, dtype=torch.float32, device=dev)
        ptr = {id(p): flat.data_ptr() + 4 * o for p, o in zip(params, offs)}
        want = {id(p) for p, need in zip(params, ctx.needs_input_grad[8:]) if need}
        gH = flat[n:].view(nV, d_out)
        h = _lib.HeadArgs()
        keep = spec.fill(h, nV, n_mols, d_out, batch, T, weights, lt_mask, gt_mask,
                         lambda p: None if (p is None or id(p) not in want) else ptr[id(p)], bn_training=spec.bn is None or spec.bn.training)
        preds = torch.empty(n_mols, spec.n_out, dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        h.preds, h.loss_out = preds.data_ptr(), loss.data_ptr()
        h.gHv, h.ldg = gH.data_ptr(), d_out
        nb = int(lib.dmpnn_head_ws_bytes(C.byref(h)))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        h.ws, h.ws_bytes = ws.data_ptr(), nb
        with engine._OnDevice(dev):
            _lib.check(lib.dmpnn_head(C.byref(h), Hv.data_ptr(), Hv.stride(0), engine._stream_ptr(dev)), "dmpnn_head")
        del keep
        ctx.flat, ctx.n, ctx.offs, ctx.shapes, ctx.want = flat, n, offs, [tuple(p.shape) for p in params], [id(p) in want for p in params]
        ctx.hv_shape = (nV, d_out)
        ctx.preds = preds
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        flat = ctx.flat * gl   # (ONE launch scales every gradient of the head by the upstream gradient)
        n = ctx.n
        gH = flat[n:].view(ctx.hv_shape) if ctx.needs_input_grad[1] else None
        gp = [flat[o:o + math.prod(sh)].view(sh) if w else None for o, sh, w in zip(ctx.offs, ctx.shapes, ctx.want)]
        return (None, gH, None, None, None, None, None, None, *gp)


def head_loss(model, Hv: Tensor, batch: Tensor, n_mols: int, targets: Tensor, weights: Optional[Tensor] = None,
              lt_mask: Optional[Tensor] = None, gt_mask: Optional[Tensor] = None) -> Optional[Tensor]:
    """``criterion(predictor.train_step(bn(agg(H_v, batch))), targets, ...)`` (``models/model.py:152-157``) as ONE autograd node on the
    head kernels (:class:`_HeadLoss`); ``None`` when the head kernels do not implement this model or these inputs (the caller then
    runs the torch modules)."""
    # (cached on the identities of what the spec was read from — swapping model.predictor / criterion / bn / agg, or changing a
    #  predictor dropout p, after the first step must not leave the old modules in use: round-4 ADVICE)
    pred = model.predictor
    key = (id(model.agg), id(model.bn), id(pred), id(getattr(pred, "criterion", None)), id(getattr(pred, "ffn", None)),
           id(getattr(pred, "output_transform", None)),
           tuple(float(getattr(m, "p", 0.0)) for m in pred.modules() if isinstance(m, nn.Dropout)) if isinstance(pred, nn.Module) else ())
    cached = model.__dict__.get("_dmpnn_head_spec")
    if cached is not None and cached[0] == key:
        spec = cached[1]
    else:
        try:
            spec = HeadSpec(model)
        except NotImplementedError as e:
            spec = str(e)
        model.__dict__["_dmpnn_head_spec"] = (key, spec)
    if isinstance(spec, str) or Hv.device.type != "cuda" or Hv.dtype != torch.float32:
        return None
    if batch is None or batch.dtype != torch.int64 or not batch.is_contiguous() or batch.numel() != Hv.shape[0] or batch.device != Hv.device:
        return None
    T = targets if (targets.dtype == torch.float32 and targets.is_contiguous()) else targets.float().contiguous()
    if T.dim() != 2 or T.shape[0] != n_mols or T.shape[1] != spec.n_tasks or (spec.bn is not None and spec.bn.training and n_mols < 2):
        return None
    if weights is not None and weights.numel() != n_mols:
        return None
    return _HeadLoss.apply(spec, Hv, batch, int(n_mols), T, weights, lt_mask, gt_mask, *spec.params())


class FusedTrainer:
    """``training_step`` + ``Adam.step`` of an :class:`MPNN` as one ``dmpnn_train_step`` call per batch.

    Takes what the kernels implement and refuses the rest loudly (those models train through the module path): a
    :class:`~chemprop_amd.nn.BondMessagePassing` block with a built-in activation, no ``V_d``, directed, dropout 0 or — with a
    ReLU-class activation — ``nn.Dropout`` inside the tile kernels (hash mask, one seed per step from torch's CPU generator); sum /
    mean / norm aggregation; optional ``nn.BatchNorm1d``; an MLP predictor with a built-in activation and dropout 0; MSE / MAE.
    """

    def __init__(self, model: MPNN, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, group=None,
                 tile_plan: bool = True):
        mp, agg, pred = model.message_passing, model.agg, model.predictor
        # a bond block: this package's mirror, or the subclass of the reference's own class (integration.HipBondMessagePassing):
        # W_h is [d_h, d_h] there (the atom variant's takes d_e + d_h columns, the mol-atom-bond ones have a second read-out)
        bond = (all(isinstance(getattr(mp, n, None), nn.Linear) for n in ("W_i", "W_h", "W_o"))
                and mp.W_h.in_features == mp.W_h.out_features and mp.W_o.in_features > mp.W_h.out_features)
        if not bond:
            raise NotImplementedError("FusedTrainer: a BondMessagePassing block (W_i / W_h [d_h, d_h] / W_o)")
        act, slope, slope_t = classify_activation(mp.tau)
        if act in ("custom", "prelu") or mp.undirected or mp.W_d is not None:
            raise NotImplementedError("FusedTrainer: built-in activation (not PReLU), directed, no V_d — other blocks train "
                                      "through the module path (MPNN.loss + autograd)")
        if mp.dropout.p > 0 and not (type(mp.dropout) is nn.Dropout and act in ("relu", "leakyrelu")):
            # (active dropout lives inside the tile kernels for ReLU-class activations: dmpnn_fwd_args.dropout_p; a dropout module
            #  that is not exactly nn.Dropout has its own semantics and stays on the module path)
            raise NotImplementedError("FusedTrainer: dropout inside the block needs nn.Dropout and a ReLU / LeakyReLU activation")
        try:
            self.head = HeadSpec(model)
        except NotImplementedError as e:
            raise NotImplementedError(f"FusedTrainer: {e}") from None
        self.model, self.mp = model, mp
        self.act, self.slope = act, slope
        self.layers, self.bn, self.bounded = self.head.layers, self.head.bn, self.head.bounded
        params = [p for p in model.parameters() if p.requires_grad]
        engine._require_device(params[0], "model parameters")
        self.sync = GradSync(params, modules=[model], group=group)
        self.opt = FlatAdam(self.sync, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.dev = params[0].device
        self._views = {id(p): v for p, v in zip(self.sync.params, self.sync.views)}
        # the two slices of the flat gradient buffer a data-parallel step exchanges one after the other (see step())
        blk = [p for p in mp.parameters() if p.requires_grad]
        blk_ids = {id(p) for p in blk}
        rest = [p for p in params if id(p) not in blk_ids]
        self._block_range = self.sync.range_of(blk) if blk else (0, 0)
        self._head_range = self.sync.range_of(rest) if rest else (0, 0)
        self._checked = 0
        self._level = None
        self.tile_plan = bool(tile_plan)   # False: always the full (CSR) plan — what every round before used; tests keep both alive
        self.last_route = None

    # ---- helpers ----
    def _gv(self, p: Optional[Tensor]) -> Optional[int]:
        if p is None or not p.requires_grad:
            return None
        return self._views[id(p)].data_ptr()

    def _world(self) -> int:
        d = torch.distributed
        return d.get_world_size(self.sync.group) if (d.is_available() and d.is_initialized()) else 1

    def step(self, bmg, targets: Tensor, weights: Optional[Tensor] = None, lt_mask: Optional[Tensor] = None,
             gt_mask: Optional[Tensor] = None, lr: Optional[float] = None, clip: Optional[tuple] = None) -> Tensor:
        """One optimisation step on ``(bmg, targets, ...)`` (a ``TrainingBatch`` without ``V_d`` / ``X_d``); returns the device
        tensor ``[loss, number of finite targets]`` of THIS step (no host sync).  ``model.train()`` semantics (batch norm uses
        and updates batch statistics).  ``clip = (value, "norm" | "value")``: Lightning's ``gradient_clip_val`` / ``_algorithm``
        (``cli/train.py:1937``), applied between the backward pass and the update inside the same call."""
        from .nn import _VALIDATE_FIRST_N, _route, _training_plan_kind

        lib = _lib.load()
        mp, dev = self.mp, self.dev
        engine._require_device(bmg.V, "bmg.V")
        if not self.model.training:
            # (batch norm would update its running statistics while the block's dropout follows model.training: the two switches
            #  must not disagree — and a training step of a model in eval mode is a bug of the caller, not a mode)
            raise RuntimeError("FusedTrainer.step: the model is in eval mode — call model.train() first")
        batch = bmg.batch
        n_mols = len(bmg)
        nV, nE = int(bmg.V.shape[0]), int(bmg.E.shape[0])
        n_tasks = self.head.n_tasks
        T = engine._f32c(targets, "targets")
        if T.dim() != 2 or T.shape[0] != n_mols or T.shape[1] != n_tasks or not T.is_contiguous():
            raise ValueError(f"targets must be a contiguous [{n_mols}, {n_tasks}] matrix, got {tuple(targets.shape)}")
        # the head kernels read these through raw pointers: a wrong dtype / size would be an out-of-bounds device read, not an error
        if batch is None or batch.dtype != torch.int64 or not batch.is_contiguous() or batch.device != bmg.V.device or batch.numel() != nV:
            raise ValueError(f"bmg.batch must be a contiguous int64 vector of {nV} molecule ids on {bmg.V.device}")
        if weights is not None and weights.numel() != n_mols:
            raise ValueError(f"weights must hold one value per molecule ({n_mols}), got {tuple(weights.shape)}")
        for name, m in (("lt_mask", lt_mask), ("gt_mask", gt_mask)):
            if m is not None and tuple(m.shape) != (n_mols, n_tasks):
                raise ValueError(f"{name} must have the targets' shape [{n_mols}, {n_tasks}], got {tuple(m.shape)}")
        if self.bn is not None and n_mols == 1:
            raise ValueError("Expected more than 1 value per channel when training (batch norm on a batch of one molecule)")
        validate = _lib.opt("DMPNN_VALIDATE", "first") != "never" and self._checked < _VALIDATE_FIRST_N
        world = self._world()
        self.sync.wait()

        # ---- K0: a launched plan while the first batches are validated (host read of the verdict), else inside the C call ----
        # The kind of plan: once the first batches are validated (on full plans: their verdict on the graph invariants is read on
        # the host), a batch bound for the tile kernels gets the TILE plan — K0 is then the 11 us tile table instead of the 28 us
        # CSR plan, the kept tensors stay in the caller's edge order and the backward tile kernel reads the batch's own index
        # arrays (DMPNN_F_TILE_PLAN; every tile checks itself, a molecule beyond the tile takes the kernels' generic path).
        no_mega = getattr(mp, "_dmpnn_no_mega", False) or (n_mols > 0 and nE > 30 * n_mols)
        oversize = getattr(bmg, "oversize", None)
        if oversize is None and mp.dropout.p > 0 and not no_mega:
            # (dropout lives inside the tile kernels only; their generic path for a molecule beyond the tile has none and answers NaN —
            #  which this step would feed to Adam.  A foreign batch is counted on the device: nn.batch_oversize)
            from .nn import batch_oversize

            oversize = batch_oversize(bmg, n_mols)
        level = 1 if (no_mega or oversize is True) else 2
        # (ONE rule for "this training forward runs on the tile plan", the module path's: shapes of the tile kernel — d_h <= 320, even
        #  d_v / d_e —, the environment switches, a plan the library can build; anything else keeps the full plan and the per-step routes)
        kind = _training_plan_kind(mp, bmg) if (self.tile_plan and not validate and level == 2) else False
        plan = engine.GraphPlan.from_bmg(bmg, light=kind, launch=validate)
        plan.oversize = oversize
        if validate:
            self._checked += 1
            level = _route(mp, plan, n_mols, batch)
            if plan.oversize is True:
                level = min(level, 1)
        self._last_plan_tiles = bool(plan.tiles_only)
        if plan.tiles_only and level < 2:  # (cannot happen: the tile plan was only asked for at level 2)
            raise RuntimeError("FusedTrainer: a tile plan without the tile kernels")
        note_batch(batch, n_mols)

        # ---- argument blocks of the block's forward / backward (workspace allocated, nothing enqueued) ----
        W = lambda lin, n: getattr(getattr(mp, lin), n)
        drop = None
        if mp.dropout.p > 0 and self.model.training:
            # one seed per step from torch's CPU generator (torch.manual_seed fixes the run), like the module path's fused dropout
            drop = (float(mp.dropout.p), int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()))
            self.last_dropout_seed = drop[1]
        try:
            out, st = engine.forward(plan, bmg.V, bmg.E, W("W_i", "weight"), W("W_h", "weight"), W("W_o", "weight"), W("W_o", "bias"),
                                     W("W_i", "bias"), W("W_h", "bias"), depth=mp.depth, act=self.act, slope=self.slope, keep=True,
                                     max_level=level, launch=False, dropout=drop)
        except engine.RouteUnavailable as e:
            raise NotImplementedError(f"FusedTrainer: this batch does not take the tile kernels ({e}); dropout on the other routes "
                                      "runs through the module path (MPNN.loss + autograd)") from None
        self.last_route = st.route
        d_out = int(out.shape[1])
        gout = torch.empty(nV, d_out, dtype=torch.float32, device=dev)
        need, views = {}, {}
        for k, (lin, n) in dict(W_i=("W_i", "weight"), b_i=("W_i", "bias"), W_h=("W_h", "weight"), b_h=("W_h", "bias"),
                                W_o=("W_o", "weight"), b_o=("W_o", "bias")).items():
            p = W(lin, n)
            need[k] = p is not None and p.requires_grad
            if need[k]:
                views[k] = self._views[id(p)]
        grads, b, keep_b = engine.backward(st, gout, need, out=views, launch=False)
        for k, g in grads.items():  # (a view the engine did not take would silently drop the gradient)
            if g is not None and g is not views.get(k):
                raise RuntimeError(f"FusedTrainer: the gradient view of {k} was not accepted (dtype / layout)")

        # ---- the head ----
        h = _lib.HeadArgs()
        keep = [T, gout, keep_b, st, plan]
        keep += self.head.fill(h, nV, n_mols, d_out, batch, T, weights, lt_mask, gt_mask, self._gv)
        bn = self.bn
        t = int(self.layers[-1].out_features)
        preds = torch.empty(n_mols, t, dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        h.preds, h.loss_out = preds.data_ptr(), loss.data_ptr()
        h.gHv, h.ldg = gout.data_ptr(), d_out
        nb = int(lib.dmpnn_head_ws_bytes(C.byref(h)))
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        h.ws, h.ws_bytes = ws.data_ptr(), nb

        s = _lib.StepArgs()
        s.edge_index, s.rev_edge_index = plan.edge_index.data_ptr(), plan.rev_edge_index.data_ptr()
        bt = batch if (batch.dtype == torch.int64 and batch.is_contiguous()) else None
        s.batch = None if bt is None else bt.data_ptr()
        s.plan_bytes, s.plan_ready = plan.buf.numel() * 4, (1 if validate else 0)
        s.bwd, s.head = b, h
        opt = self.opt
        fused_update = world == 1 and not force_collective()   # (forced: the staged data-parallel step also on one rank)
        if fused_update:
            k = opt.steps + 1  # (committed below, once the call has returned OK: a refused step must not advance Adam's bias correction)
            b1, b2 = opt.betas
            s.p, s.g, s.m, s.v, s.n_params = opt.flat.data_ptr(), self.sync.flat.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr(), opt.flat.numel()
            s.lr, s.beta1, s.beta2, s.eps, s.weight_decay = float(opt.lr if lr is None else lr), b1, b2, opt.eps, opt.weight_decay
            s.bias_corr1, s.sqrt_bias_corr2, s.grad_scale = 1.0 - b1 ** k, math.sqrt(1.0 - b2 ** k), 1.0
            if clip is not None and clip[0] is not None and float(clip[0]) > 0:
                from .optim import CLIP_MODES

                s.clip_val, s.clip_mode, s.clip_ws = float(clip[0]), CLIP_MODES[clip[1] or "norm"], opt.clip_ws.data_ptr()
        with engine._OnDevice(dev):
            if fused_update:
                _lib.check(lib.dmpnn_train_step(C.byref(s), engine._stream_ptr(dev)), "dmpnn_train_step")
                opt.steps = k
            else:
                # data parallel: the head's gradients (predictor, batch norm) are final before the block's backward pass starts —
                # their slice of the flat buffer goes out on the communication stream while that pass runs; the block's slice
                # follows it; the update waits (stream dependency) for both
                s.stages = _lib.STEP_FORWARD
                _lib.check(lib.dmpnn_train_step(C.byref(s), engine._stream_ptr(dev)), "dmpnn_train_step(forward)")
                self.sync.allreduce(*self._head_range)
                s.stages = _lib.STEP_BACKWARD
                _lib.check(lib.dmpnn_train_step(C.byref(s), engine._stream_ptr(dev)), "dmpnn_train_step(backward)")
                self.sync.allreduce(*self._block_range)
        if bn is not None and not h.bn_num_batches_tracked and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        self.preds = preds
        if fused_update:
            for p in self.sync.params:  # (the engine's weight caches key on the autograd version)
                torch.autograd.graph.increment_version(p)
            self.sync.new_step()
        else:
            opt.step(lr, clip=None if (clip is None or clip[0] is None or not float(clip[0]) > 0) else (float(clip[0]), clip[1] or "norm"))
        return loss
