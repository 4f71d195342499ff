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


def _border(mask: np.ndarray, connectivity: int) -> np.ndarray:
    from scipy.ndimage import binary_erosion, generate_binary_structure
    return mask ^ binary_erosion(mask, structure=generate_binary_structure(mask.ndim, connectivity), iterations=1)


def surface_distances(result, reference, voxelspacing=None, connectivity=1) -> np.ndarray:
    """Distances from every surface voxel of `result` to the nearest surface voxel of `reference` (the surface-distance
    construction of medpy 0.4.0 `metric.binary`, carried by measure.py:1096-1128): surface = mask XOR its erosion, distances from
    the Euclidean distance transform of the reference surface's complement.  Host code (scipy), as upstream."""
    from scipy.ndimage import distance_transform_edt
    a, b = np.atleast_1d(np.asarray(result).astype(bool)), np.atleast_1d(np.asarray(reference).astype(bool))
    if not a.any():
        raise RuntimeError("The first supplied array does not contain any binary object.")
    if not b.any():
        raise RuntimeError("The second supplied array does not contain any binary object.")
    if voxelspacing is not None:
        voxelspacing = np.ascontiguousarray(np.broadcast_to(np.asarray(voxelspacing, dtype=np.float64), (a.ndim,)))
    return distance_transform_edt(~_border(b, connectivity), sampling=voxelspacing)[_border(a, connectivity)]


def hd(result, reference, voxelspacing=None, connectivity=1) -> float:
    """Symmetric Hausdorff distance (measure.py:333-378)."""
    return max(surface_distances(result, reference, voxelspacing, connectivity).max(),
               surface_distances(reference, result, voxelspacing, connectivity).max())


def hd_2D_stack(result, reference, pixelspacing=None, connectivity=1) -> float:
    """Mean in-plane Hausdorff distance over the slices where both masks are non-empty; -1 when there is none
    (measure.py:381-399)."""
    vals = [hd(r, g, pixelspacing, connectivity) for r, g in zip(result, reference) if r.sum() > 0 and g.sum() > 0]
    return sum(vals) / len(vals) if vals else -1


def asd(result, reference, voxelspacing=None, connectivity=1) -> float:
    """Directed average surface distance result -> reference; 1e100 when either mask is empty (measure.py:458-548)."""
    if np.sum(result) > 0 and np.sum(reference) > 0:
        return surface_distances(result, reference, voxelspacing, connectivity).mean()
    return 1e100


class runningMySegmentationScore(object):
    """Patient-wise scores of 3-D predictions (metrics.py:139-291): one row per patient, one column per (foreground class, metric).

    'Dice', 'VolError' and 'VolSim' are functions of three voxel counts per class (|pred|, |gt|, |pred & gt|); for device tensors
    those come from the confusion-matrix kernel (one launch pair and one 2*n^2-word readback per patient instead of 2*(n-1)
    full-volume host copies and masks).  The surface-distance metrics 'HD' and 'ASD' are scipy distance transforms on the host,
    as upstream (measure.py:333-548); asking for them brings the two label volumes to the host once per patient."""
    SUPPORTED = ("Dice", "VolError", "VolSim", "HD", "ASD")

    def __init__(self, n_classes, idx2cls_dict=None, metrics_list=("Dice",), foreground_only=False):
        self.n_classes, self.metrics, self.foreground_only = n_classes, list(metrics_list), foreground_only
        if idx2cls_dict is None:
            idx2cls_dict = {1: "foreground"} if foreground_only else {c: str(c) for c in range(n_classes)}
        self.idx2cls_dict = idx2cls_dict
        self.multi_scores, self.tables, self.header = {}, [], ["patient_id"]
        for c, name in idx2cls_dict.items():
            if c > 0:
                for m in self.metrics:
                    if m not in self.SUPPORTED:
                        raise NotImplementedError(f"metric {m!r}: only {self.SUPPORTED} are computed by this build")
                    self.multi_scores[name + "_" + m] = []
                    self.header.append(name + "_" + m)

    def _counts(self, preds, gts):
        """-> (pred_count[c], gt_count[c], intersection[c]) as the reference's per-class binarisation counts them
        (metrics.py:205-223): a ground-truth label outside [0, n) belongs to no class, the predicted voxel under it still counts."""
        import torch
        n = self.n_classes
        if torch.is_tensor(preds) and torch.is_tensor(gts) and preds.is_cuda and gts.is_cuda:
            from . import ops
            both = torch.stack([ops.confusion_hist(gts, preds, n), ops.confusion_hist(preds, preds, n)]).cpu().numpy()
            h, hp = both[0], both[1]
            if self.foreground_only and int(h.sum()) != gts.numel():
                raise ValueError("foreground_only with labels outside [0, n_classes): pass host arrays")
            pc, gc, ic = np.diag(hp).copy(), h.sum(axis=1), np.diag(h).copy()
        else:
            p, g = np.asarray(preds).reshape(-1).astype(np.int64), np.asarray(gts).reshape(-1).astype(np.int64)
            if self.foreground_only:                       # gt > 0 / pred > 0 (any positive label is foreground upstream)
                p, g = (p > 0).astype(np.int64), (g > 0).astype(np.int64)
            pc = np.bincount(p[(p >= 0) & (p < n)], minlength=n)
            gc = np.bincount(g[(g >= 0) & (g < n)], minlength=n)
            same = p[(p == g) & (p >= 0) & (p < n)]
            ic = np.bincount(same, minlength=n)
            return pc, gc, ic
        if self.foreground_only:                           # everything that is not background, on both sides
            fg_i = h[1:, 1:].sum()
            pc, gc, ic = np.array([0, pc[1:].sum()]), np.array([0, gc[1:].sum()]), np.array([0, fg_i])
        return pc, gc, ic

    def update(self, pid, preds, gts, voxel_spacing=None):
        """preds / gts: integer volumes [n_slices, H, W] (numpy, or both torch tensors on the GPU); returns the patient's row."""
        if tuple(preds.shape) != tuple(gts.shape):
            raise AssertionError(f"pid :{pid} shape not consistent: pred {tuple(preds.shape)} vs gt {tuple(gts.shape)}")
        if voxel_spacing is not None and len(voxel_spacing) != 3:
            raise AssertionError(f"check voxel spacing, {voxel_spacing}")
        pc, gc, ic = self._counts(preds, gts)
        surf = [m for m in self.metrics if m in ("HD", "ASD")]
        if surf:
            if voxel_spacing is None:
                raise ValueError("'HD' / 'ASD' need the voxel spacing (x, y, z) of the volume")
            if "HD" in surf:      # metrics.py:225-229: the in-plane pair is voxel_spacing[:2]; upstream's own guard on the axis order
                assert voxel_spacing[0] >= voxel_spacing[2], "z spacing should be in last dim in the cardiac imaging"
            p_h = preds.detach().cpu().numpy() if hasattr(preds, "detach") else np.asarray(preds)
            g_h = gts.detach().cpu().numpy() if hasattr(gts, "detach") else np.asarray(gts)
        row = [str(pid)]
        for c, name in self.idx2cls_dict.items():
            if c == 0:
                continue
            v1, v2, inter = int(pc[c]), int(gc[c]), int(ic[c])
            for m in self.metrics:
                if m == "Dice":                            # medpy dc: ZeroDivisionError -> 0.0
                    score = 2.0 * inter / float(v1 + v2) if v1 + v2 else 0.0
                elif m in ("HD", "ASD"):
                    pm = (p_h > 0) if self.foreground_only else (p_h == c)
                    gm = (g_h > 0) if self.foreground_only else (g_h == c)
                    if m == "HD":                          # 2-D stack, 8-neighbourhood surfaces (metrics.py:224-230)
                        score = hd_2D_stack(pm, gm, pixelspacing=voxel_spacing[:2], connectivity=2)
                    else:
                        score = asd(pm, gm, voxelspacing=voxel_spacing, connectivity=2)
                    score = float(score)
                elif m == "VolError":                      # (pred - gt) / gt, numpy float division (inf / nan on an empty gt)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        score = float(np.float64(v1 - v2) / np.float64(1.0 * v2))
                else:                                      # VolSim, measure.py:668-722
                    if v2 == 0:
                        raise RuntimeError("The second supplied array does not contain any binary object.")
                    score = float(1 - np.abs(v1 - v2) / np.abs(float(v2 + v1)))
                self.multi_scores[name + "_" + m].append(score)
                row.append(score)
        self.tables.append(row)
        return row

    def get_scores(self, save_path=None):
        """-> ({col_mean, col_std}, [[means as '%.3f'], [stds]], header); written as csv when save_path is given."""
        summary, rows, header = {}, [[], []], []
        for k, vals in self.multi_scores.items():
            mean, std = np.mean(vals), np.std(vals)
            summary[k + "_mean"], summary[k + "_std"] = mean, std
            rows[0].append("{:.3f}".format(mean))
            rows[1].append("{:.3f}".format(std))
            header.append(k)
        if save_path is not None:
            import pandas as pd
            pd.DataFrame(rows, columns=header).to_csv(save_path, index=False)
        return summary, rows, header

    def save_patient_wise_result_to_csv(self, save_path):
        import pandas as pd
        df = pd.DataFrame(self.tables, columns=self.header)
        if save_path is not None:
            df.to_csv(save_path, index=False)
        return df

    def reset(self):
        for k in self.multi_scores:
            self.multi_scores[k] = []
        self.tables = []
