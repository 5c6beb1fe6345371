"""Metric arithmetic of reproducibility/metrics.py, restated in numpy (no sklearn dependency).

``retrieval_metrics`` (metrics.py:5-15): p@10 / p@50 = share of queries whose target index is among the
first 10 / 50 retrieved indices.  ``eval_metrics`` (metrics.py:19-75): accuracy, support-weighted
precision / recall / F1, multi-class Matthews correlation, AUC for a binary problem with scores, and the
binary confusion counts the reference derives for integer labels 1 (positive) and 0 (negative).
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np


def retrieval_metrics(y_target: Sequence[int], y_predictions) -> Dict[str, float]:
    pred = np.asarray(y_predictions)
    tgt = np.asarray(y_target).reshape(-1, 1)
    n = len(tgt)
    if n == 0:
        raise ZeroDivisionError("retrieval_metrics of an empty query set")
    hit10 = (pred[:, :10] == tgt).any(axis=1).sum()
    hit50 = (pred[:, :50] == tgt).any(axis=1).sum()
    return {"p@10": float(hit10) / n, "p@50": float(hit50) / n}


def _confusion(y_true, y_pred):
    labels, inv = np.unique(np.concatenate([np.asarray(y_true, dtype=object).astype(str),
                                            np.asarray(y_pred, dtype=object).astype(str)]), return_inverse=True)
    n = len(y_true)
    t, p = inv[:n], inv[n:]
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    np.add.at(cm, (t, p), 1)
    return cm


def _binary_auc(y_true, score) -> float:
    from scipy.stats import rankdata
    y = np.asarray(y_true)
    pos = y == np.unique(y).max()                      # sklearn's default positive label for {0,1} / {-1,1}
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    r = rankdata(np.asarray(score, dtype=np.float64))  # average ranks == trapezoidal ROC area with ties
    return float((r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def eval_metrics(y_true, y_pred, y_pred_proba=None, average_method: str = "weighted", verbose: bool = False):
    assert len(y_true) == len(y_pred)
    if average_method != "weighted":
        raise NotImplementedError("only the reference's default, support-weighted averaging, is provided")
    n = len(y_true)
    cm = _confusion(y_true, y_pred)
    tp_c = np.diag(cm).astype(np.float64)
    support = cm.sum(axis=1).astype(np.float64)          # true instances per class
    predicted = cm.sum(axis=0).astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        prec_c = np.where(predicted > 0, tp_c / predicted, 0.0)
        rec_c = np.where(support > 0, tp_c / support, 0.0)
        f1_c = np.where(prec_c + rec_c > 0, 2 * prec_c * rec_c / (prec_c + rec_c), 0.0)
    w = support / max(n, 1)
    acc = float(tp_c.sum() / n) if n else float("nan")
    # multi-class MCC (Gorodkin's R_K): (c s - sum p_k t_k) / sqrt((s^2 - sum p_k^2)(s^2 - sum t_k^2))
    c, s = tp_c.sum(), float(n)
    den = np.sqrt((s * s - (predicted ** 2).sum()) * (s * s - (support ** 2).sum()))
    mcc = float((c * s - (predicted * support).sum()) / den) if den > 0 else 0.0

    if y_pred_proba is None:
        auroc = float("nan")
    elif len(np.unique(np.asarray(y_true))) > 2:
        print("Multiclass AUC is not currently available.")
        auroc = float("nan")
    else:
        auroc = _binary_auc(y_true, y_pred_proba)

    # binary counts exactly as the reference derives them: only integer-like labels 1 / 0 participate
    yt, yp = np.asarray(y_true, dtype=object), np.asarray(y_pred, dtype=object)
    is1_t, is1_p = np.array([v == 1 for v in yt], bool), np.array([v == 1 for v in yp], bool)
    is0_t, is0_p = np.array([v == 0 for v in yt], bool), np.array([v == 0 for v in yp], bool)
    eq = np.array([a == b for a, b in zip(yt, yp)], bool)
    tp, fp = int((eq & is1_p & is1_t).sum()), int((is1_p & ~eq).sum())
    tn, fn = int((eq & is0_p & is0_t).sum()), int((is0_p & ~eq).sum())
    nan = float("nan")
    perf = {"Accuracy": acc, "AUC": auroc, "WF1": float((f1_c * w).sum()), "precision": float((prec_c * w).sum()),
            "recall": float((rec_c * w).sum()), "mcc": mcc, "tp": tp, "fp": fp, "tn": tn, "fn": fn,
            "sensitivity": tp / (tp + fn) if tp + fn else nan, "specificity": tn / (tn + fp) if tn + fp else nan,
            "ppv": tp / (tp + fp) if tp + fp else nan, "npv": tn / (tn + fn) if tn + fn else nan,
            "hitrate": (tp + tn) / (tp + tn + fp + fn) if tp + tn + fp + fn else nan, "instances": n}
    if verbose:
        for k, v in perf.items():
            print(f"{k:12s} {v}")
    return perf
