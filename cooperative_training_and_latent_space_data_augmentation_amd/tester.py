"""Patient-wise evaluation harness: volume in -> per-class Dice out.

Mirror of `TestSegmentationNetwork` (medseg/test_basic_segmentation_solver.py:29-172) for the part that sits on the accelerated
path: the volume is uploaded once, predicted in chunks of <= `maximum_batch_size` slices (`predict`, model.py:375-394), the arg-max
and the per-patient voxel counts are taken on the device, and one 2*n^2-word readback per patient replaces the reference's two
full-volume `.cpu().numpy()` copies per chunk.  The dataset only has to offer what the reference's tester reads from it:
`patient_number`, `get_patient_data_for_testing(i, crop_size=)` -> {'image': [n,1,H,W], 'label': [n,H,W]}, `get_id()`,
`get_voxel_spacing()` and `formalized_label_dict`.  Writing nrrd files (SimpleITK) is outside the path and not offered."""
import os

import numpy as np
import torch

from . import ops
from .metrics import runningMySegmentationScore


PASS_FLOATS_PER_PIXEL = 512      # generous bound of what one slice of a `predict` pass keeps on the device, in floats per input pixel (FTN + STN
                                 # encoder / decoder workspaces: a bs16 256^2 decoder pass holds 289 MB = 69 floats per pixel and slice)


def max_slices_per_pass(h: int, w: int, widest_channels: int = 16, device=None) -> int:
    """The kernels address a tensor with 32-bit byte offsets (< 2 GiB, include/ctl_hip.h); the widest tensor of a pass is the
    full-resolution 16-channel fp32 feature map: 64 bytes per pixel.  With `device` the pass is also bounded by HALF of the device memory
    that is free right now (`maximum_batch_size` exists upstream to bound GPU memory: a coalesced pass must not defeat that -- ADVICE r4)."""
    limit = max(1, (2 ** 31 - 1) // (h * w * widest_channels * 4))
    if device is not None:
        free, _ = torch.cuda.mem_get_info(device)
        limit = max(1, min(limit, int(free // 2) // (h * w * 4 * PASS_FLOATS_PER_PIXEL)))
    return limit


COALESCE_CHUNKS = True      # run consecutive <= `maximum_batch_size`-slice chunks of a volume as one pass (see predict_volume)


def predict_volume(segmentation_model, image_d: torch.Tensor, n_iter=None, chunk=None, out: torch.Tensor = None, coalesce=None) -> torch.Tensor:
    """uint8 label volume [n,H,W] of a device volume [n,1,H,W]: `predict` (model.py:375-394) + arg-max on the device.
    chunk = 10 is the reference's loop (test_basic_segmentation_solver.py:85-114): a GPU-MEMORY workaround, not part of the computation --
    `predict` puts every network in eval mode (running BatchNorm statistics), so slices are independent and batching changes nothing but
    the number of launches (bitwise equal logits and labels, tests/test_engine_gpu.py::test_whole_volume_pass_is_bitwise_the_chunked_loop).
    With `coalesce` (default: tester.COALESCE_CHUNKS = True) consecutive chunks therefore run as ONE pass, as many as the 2 GiB
    addressing limit of a tensor AND half of the currently free device memory allow (max_slices_per_pass); coalesce=False is the
    reference's literal loop.  chunk = None:
    the whole volume per pass either way."""
    n, _, h, w = image_d.shape
    limit = max_slices_per_pass(h, w, device=image_d.device)
    coalesce = COALESCE_CHUNKS if coalesce is None else bool(coalesce)
    if chunk is not None and int(chunk) < 1:
        raise ValueError("chunk must be positive")
    chunk = limit if (chunk is None or coalesce) else min(int(chunk), limit)
    pred = out if out is not None else torch.empty((n, h, w), dtype=torch.uint8, device=image_d.device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        pred[lo:hi] = ops.argmax_c(segmentation_model.predict(input=image_d[lo:hi], softmax=False, n_iter=n_iter))
    return pred


class TestSegmentationNetwork(object):
    __test__ = False                     # not a pytest class

    def __init__(self, test_dataset, crop_size, segmentation_model, use_gpu=True, save_path="", summary_report_file_name="result.csv",
                 detailed_report_file_name="details.csv", patient_wise=True, metrics_list=("Dice", "HD"), foreground_only=False,
                 save_soft_prediction=False, keep_results=True):
        if not use_gpu:
            raise ValueError("this build has no CPU path")
        self.test_dataset, self.crop_size, self.segmentation_model = test_dataset, crop_size, segmentation_model
        self.num_classes = segmentation_model.num_classes
        self.segmentation_metric = runningMySegmentationScore(n_classes=self.num_classes,
                                                              idx2cls_dict=getattr(test_dataset, "formalized_label_dict", None),
                                                              metrics_list=metrics_list, foreground_only=foreground_only)
        self.save_path, self.patient_wise = save_path, patient_wise
        self.summary_report_file_name, self.detailed_report_file_name = summary_report_file_name, detailed_report_file_name
        self.save_soft_prediction, self.keep_results = save_soft_prediction, keep_results
        if save_path and not os.path.exists(save_path):
            os.makedirs(save_path)
        self.df, self.result_dict = None, {}

    def run(self):
        n = self.test_dataset.patient_number if self.patient_wise else len(self.test_dataset)
        for i in range(n):
            pack = self.test_dataset.get_patient_data_for_testing(i, crop_size=self.crop_size) if self.patient_wise else self.test_dataset[i]
            pid, result = self.evaluate(i, pack, n)
            if self.keep_results:
                self.result_dict[pid] = result
        join = os.path.join
        self.segmentation_metric.get_scores(save_path=join(self.save_path, self.summary_report_file_name) if self.save_path else None)
        self.df = self.segmentation_metric.save_patient_wise_result_to_csv(
            save_path=join(self.save_path, self.detailed_report_file_name) if self.save_path else None)
        return self.df

    def evaluate(self, i, data_tensor_pack, total_number, maximum_batch_size=10, coalesce=None):
        """One patient: `predict` over the volume, device arg-max into one uint8 volume, metric update from the device tensors.
        maximum_batch_size = 10 is upstream's default argument (a GPU-memory workaround); the chunks it asks for are run as one pass
        (tester.COALESCE_CHUNKS, exact: see predict_volume; 2x the slices/s on a 40-slice volume); coalesce=False: the literal loop."""
        dev = torch.device("cuda", torch.cuda.current_device())
        image = data_tensor_pack["image"]
        if image.dim() == 5:                              # DataLoader(batch_size=1) adds a leading axis upstream
            image = image[0]
        limit = max_slices_per_pass(image.shape[-2], image.shape[-1], device=dev)
        assert maximum_batch_size is None or int(maximum_batch_size) > 0
        if maximum_batch_size is None or (COALESCE_CHUNKS if coalesce is None else coalesce):
            maximum_batch_size = min(int(image.shape[0]), limit)
        else:
            maximum_batch_size = min(int(maximum_batch_size), limit)
        label = torch.as_tensor(data_tensor_pack["label"]).reshape(-1, image.shape[-2], image.shape[-1])
        assert image.size(1) == 1, "currently only support gray images, found: {}".format(image.size(1))
        image_d = image.to(dev, dtype=torch.float32, non_blocking=True)
        label_d = label.to(dev, dtype=torch.int64, non_blocking=True)
        pid = self.test_dataset.get_id()
        total = image_d.size(0)
        pred_d = torch.empty((total, image_d.shape[-2], image_d.shape[-1]), dtype=torch.uint8, device=dev)
        soft = [] if (self.save_soft_prediction or self.keep_results) else None
        for lo in range(0, total, maximum_batch_size):
            hi = min(total, lo + maximum_batch_size)
            logit = self.segmentation_model.predict(input=image_d[lo:hi], softmax=False)
            pred_d[lo:hi] = ops.argmax_c(logit)
            if soft is not None:
                soft.append(logit)
        spacing = self.test_dataset.get_voxel_spacing() if hasattr(self.test_dataset, "get_voxel_spacing") else None
        self.segmentation_metric.update(pid=pid, preds=pred_d, gts=label_d, voxel_spacing=spacing)
        result = None
        if soft is not None:
            soft_np = torch.cat(soft, 0).float().cpu().numpy()
            result = {"image": image.numpy().reshape(-1, image.shape[-2], image.shape[-1]), "label": label.numpy(),
                      "pred": pred_d.cpu().numpy(), "soft_pred": soft_np}
            if total == 1:
                result = {k: v[0] for k, v in result.items()}
        if self.save_soft_prediction and self.save_path:
            out = os.path.join(self.save_path, "pred_npy")
            os.makedirs(out, exist_ok=True)
            tag = str(pid).replace("/", "_")
            for name, key in (("soft_pred", "soft_pred"), ("gt", "label"), ("image", "image")):
                np.save(os.path.join(out, "{}_{}.npy".format(tag, name)), result[key])
        return pid, result

    def get_top_k_results(self, topk=5, attribute="MYO_Dice", order=0):
        assert self.df is not None, "please run evaluation before saving"
        if order == 0:
            return self.df.nlargest(topk, attribute)
        if order == 1:
            return self.df.nsmallest(topk, attribute)
        raise ValueError(order)
