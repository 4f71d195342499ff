"""Validation metrics used by `evaluate` (medseg/common_utils/metrics.py:12-54) and Dice (measure.py:52-99)."""
import numpy as np


class runningScore(object):
    """Confusion-matrix accumulator: overall / mean accuracy, mean IoU, frequency-weighted accuracy."""

    def __init__(self, n_classes):
        self.n_classes = n_classes
        self.confusion_matrix = np.zeros((n_classes, n_classes))

    def _fast_hist(self, label_true, label_pred, n_class):
        keep = (label_true >= 0) & (label_true < n_class)
        idx = n_class * label_true[keep].astype(int) + label_pred[keep].astype(int)
        return np.bincount(idx, minlength=n_class ** 2).reshape(n_class, n_class)

    def update(self, label_trues, label_preds):
        """Device tensors (both on the GPU) are accumulated by the HIP kernel into an int64 matrix that stays on the device: no
        host round trip per batch; numpy inputs follow the upstream host path."""
        import torch
        if torch.is_tensor(label_trues) and torch.is_tensor(label_preds) and label_trues.is_cuda and label_preds.is_cuda:
            from . import ops
            self._dev_hist = ops.confusion_hist(label_trues, label_preds, self.n_classes, getattr(self, "_dev_hist", None))
            return
        for lt, lp in zip(label_trues, label_preds):
            self.confusion_matrix += self._fast_hist(np.asarray(lt).flatten(), np.asarray(lp).flatten(), self.n_classes)

    def _total(self):
        h = self.confusion_matrix
        if getattr(self, "_dev_hist", None) is not None:
            h = h + self._dev_hist.cpu().numpy().astype(np.float64)          # the one synchronisation of an evaluation
        return h

    def get_scores(self):
        h = self._total()
        with np.errstate(divide="ignore", invalid="ignore"):
            acc = np.diag(h).sum() / h.sum()
            acc_cls = np.nanmean(np.diag(h) / h.sum(axis=1))
            iu = np.diag(h) / (h.sum(axis=1) + h.sum(axis=0) - np.diag(h))
            freq = h.sum(axis=1) / h.sum()
        return ({"Overall Acc: \t": acc, "Mean Acc : \t": acc_cls, "FreqW Acc : \t": (freq[freq > 0] * iu[freq > 0]).sum(),
                 "Mean IoU : \t": np.nanmean(iu)}, dict(zip(range(self.n_classes), iu)))

    def reset(self):
        self.confusion_matrix = np.zeros((self.n_classes, self.n_classes))
        self._dev_hist = None


def dice_from_confusion(confusion) -> np.ndarray:
    """Per-class Dice of one volume from its confusion matrix: 2*h_cc / (row_c + col_c) == dc(pred == c, gt == c) (measure.py:52-99)."""
    h = np.asarray(confusion, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return 2.0 * np.diag(h) / (h.sum(axis=1) + h.sum(axis=0))


def dice(result, reference) -> float:
    """2|A&B| / (|A|+|B|) on binarised inputs; NaN when both are empty."""
    a, b = np.asarray(result).astype(bool), np.asarray(reference).astype(bool)
    den = int(a.sum()) + int(b.sum())
    return float("nan") if den == 0 else 2.0 * int((a & b).sum()) / den
