#!/usr/bin/env python3
"""Golden vectors for the SURVEY 8(f) rows 1 and 3 (validation metrics, input pipeline), produced by the REAL reference functions
(read-only at /root/reference) on CPU.  Build container only:  python tools/gen_golden_io.py  ->  tests/golden/io_cases.pt
Only inputs / expected outputs are written; the reference's missing third-party imports are stubbed as in tools/gen_golden.py."""
import os
import sys
from unittest.mock import MagicMock

for _m in ["numpy.lib.function_base", "SimpleITK", "medpy", "medpy.metric", "medpy.metric.binary", "IPython",
           "IPython.display", "skimage", "skimage.transform", "seaborn"]:
    sys.modules[_m] = MagicMock()
sys.path.insert(0, os.environ.get("CTL_REFERENCE", "/root/reference"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from medseg.common_utils.basic_operations import crop_or_pad, rescale_intensity  # noqa: E402
import medseg.common_utils.metrics as ref_metrics  # noqa: E402
from medseg.common_utils.measure import dc as ref_dc  # noqa: E402
from medseg.common_utils.metrics import runningMySegmentationScore, runningScore  # noqa: E402

# metrics.py:5 takes `dc` from medpy 0.4.0 (absent here); measure.py:52-99 is the reference's own copy of that function and differs
# from it only when both masks are empty (NaN instead of medpy's 0.0) -- the cases below keep every class non-empty somewhere.
ref_metrics.dc = ref_dc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "io_cases.pt")


def main():
    g = torch.Generator().manual_seed(7)
    rng = np.random.RandomState(7)
    cases = {"rescale": [], "crop_or_pad": [], "running_score": [], "noise_clamp": [], "patient_scores": [], "surface_scores": []}
    # rescale_intensity (basic_operations.py:232-245): N*C*H*W, per (n, c) plane; includes a constant plane (max == min)
    for shape, lo, hi in [((3, 1, 17, 23), 0.0, 1.0), ((2, 2, 32, 32), -1.0, 2.5), ((16, 1, 64, 64), 0.0, 1.0)]:
        x = torch.randn(shape, generator=g) * 3.0 + 1.0
        x[0, 0] = 0.75
        cases["rescale"].append({"x": x, "new_min": lo, "new_max": hi, "y": rescale_intensity(x.clone(), lo, hi)})
    # crop_or_pad (basic_operations.py:173-220): 3-D image + label, crop / pad / mixed, odd differences
    for (n, h, w), (nh, nw) in [((2, 11, 14), (8, 8)), ((3, 5, 6), (8, 9)), ((2, 13, 5), (6, 10)), ((1, 7, 7), (7, 7)), ((2, 192, 180), (192, 192))]:
        img = rng.randn(n, h, w).astype(np.float32)
        lab = rng.randint(0, 4, (n, h, w)).astype(np.int64)
        out = crop_or_pad(img.copy(), (nh, nw), lab.copy())
        cases["crop_or_pad"].append({"image": torch.from_numpy(img), "label": torch.from_numpy(lab), "size": (nh, nw),
                                     "image_out": torch.from_numpy(np.ascontiguousarray(out[0])),
                                     "label_out": torch.from_numpy(np.ascontiguousarray(out[1]))})
    # runningScore (metrics.py:12-54): two updates, labels with out-of-range entries (ignored), a class never predicted
    rs = runningScore(4)
    batches = []
    for k in range(2):
        lt = rng.randint(-1, 5, (3, 24, 20)).astype(np.int64)       # -1 and 4 are outside [0, 4)
        lp = rng.randint(0, 3, (3, 24, 20)).astype(np.uint8)        # class 3 never predicted
        rs.update(label_trues=lt, label_preds=lp)
        batches.append((torch.from_numpy(lt), torch.from_numpy(lp)))
    score, cls_iu = rs.get_scores()
    cases["running_score"].append({"batches": batches, "confusion": torch.from_numpy(rs.confusion_matrix.copy()),
                                   "score": {k: float(v) for k, v in score.items()}, "cls_iu": {int(k): float(v) for k, v in cls_iu.items()}})
    # input noise (train_adv_supervised_segmentation_triplet.py:185-187) with the noise tensor recorded
    clean = torch.rand(4, 1, 32, 32, generator=g)
    noise = 0.05 * torch.randn(4, 1, 32, 32, generator=g)
    cases["noise_clamp"].append({"clean": clean, "noise": noise, "out": torch.clamp(clean + noise, 0, 1)})
    # patient-wise 3-D scores (metrics.py:139-291): three volumes, per-class and foreground-only bookkeeping
    for fg in (False, True):
        ms = runningMySegmentationScore(n_classes=4, idx2cls_dict=None if fg else {0: "BG", 1: "LV", 2: "MYO", 3: "RV"},
                                        metrics_list=["Dice", "VolError", "VolSim"], foreground_only=fg)
        vols, rows = [], []
        for k in range(3):
            gt = rng.randint(0, 4, (5 + k, 24, 20)).astype(np.int64)
            pr = np.where(rng.rand(*gt.shape) < 0.7, gt, rng.randint(0, 4, gt.shape)).astype(np.uint8)
            rows.append(ms.update(pid="p%d" % k, preds=pr.copy(), gts=gt.copy(), voxel_spacing=[1.25, 1.25, 10.0]))
            vols.append((torch.from_numpy(pr), torch.from_numpy(gt)))
        summary, summary_list, header = ms.get_scores()
        cases["patient_scores"].append({"foreground_only": fg, "idx2cls": ms.idx2cls_dict, "volumes": vols,
                                        "rows": [[r[0]] + [float(v) for v in r[1:]] for r in rows],
                                        "summary": {k: float(v) for k, v in summary.items()}, "summary_list": summary_list,
                                        "header": header, "table_header": ms.header})
    # surface-distance metrics (measure.py:333-548 through metrics.py:224-236): ring phantoms, prediction = shifted / eroded rings,
    # one slice without the class in the prediction, one volume per bookkeeping mode
    import contextlib, io
    yy, xx = np.mgrid[0:40, 0:36]
    def rings(cy, cx, scale):
        r = np.sqrt(((yy - cy) / scale) ** 2 + ((xx - cx) / (0.8 * scale)) ** 2)
        return np.where(r < 4, 1, np.where(r < 7, 2, np.where(r < 10, 3, 0))).astype(np.int64)
    for fg in (False, True):
        ms = runningMySegmentationScore(n_classes=4, idx2cls_dict=None if fg else {0: "BG", 1: "LV", 2: "MYO", 3: "RV"},
                                        metrics_list=["Dice", "HD", "ASD"], foreground_only=fg)
        vols, rows = [], []
        for k in range(2):
            gt = np.stack([rings(20 + 0.5 * z, 18 - 0.3 * z, 1.0 + 0.05 * z) for z in range(6)])
            pr = np.stack([rings(20.8 + 0.5 * z + k, 17.1 - 0.3 * z, 1.1 + 0.04 * z) for z in range(6)]).astype(np.uint8)
            pr[0][pr[0] == 1] = 2                                      # class 1 missing from the first predicted slice
            with contextlib.redirect_stdout(io.StringIO()):            # hd_2D_stack prints the slice index
                rows.append(ms.update(pid="s%d" % k, preds=pr.copy(), gts=gt.copy(), voxel_spacing=[1.5, 1.25, 1.0]))
            vols.append((torch.from_numpy(pr), torch.from_numpy(gt)))
        cases["surface_scores"].append({"foreground_only": fg, "idx2cls": ms.idx2cls_dict, "volumes": vols, "spacing": [1.5, 1.25, 1.0],
                                        "rows": [[r[0]] + [float(v) for v in r[1:]] for r in rows], "table_header": ms.header})
    torch.save(cases, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
