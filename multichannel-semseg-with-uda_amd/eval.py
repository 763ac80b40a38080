"""Segmentation metrics of the reference's ``eval.py`` (:21-47, :96-175) with the confusion matrix accumulated on
the MI355X (``mcdseg_confusion_hist``) instead of per-image numpy bincounts over PNG round-trips.

``fast_hist`` keeps the reference's name and argument order (ground truth first) but takes GPU label tensors and
returns an int64 GPU tensor; the scalar metrics run on the n x n matrix on the host exactly as the reference's do
(same formulas, same nan conventions).  The palette plots / pandas tables of the reference are outside this build.
"""
import numpy as np
import torch

from mcdseg import ops


def fast_hist(a, b, n, out=None):
    """confusion matrix rows = ground truth ``a``, columns = prediction ``b``; entries with a outside [0,n) are dropped
    (eval.py:21-23).  ``out`` (int64 [n,n] on the GPU) is accumulated into when given."""
    if out is None:
        out = torch.zeros((n, n), dtype=torch.int64, device=a.device)
    return ops.confusion_hist_(out, a.reshape(-1), b.reshape(-1))


def _np(hist):
    return hist.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(hist) else np.asarray(hist, dtype=np.float64)


def per_class_iu(hist):
    hist = _np(hist)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


def calc_fw_iu(hist):
    hist = _np(hist)
    pred_per_class, gt_per_class = hist.sum(0), hist.sum(1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nansum((gt_per_class * np.diag(hist)) / (pred_per_class + gt_per_class - np.diag(hist))) / gt_per_class.sum()


def calc_pixel_accuracy(hist):
    hist = _np(hist)
    return np.diag(hist).sum() / hist.sum(1).sum()


def calc_mean_accuracy(hist):
    hist = _np(hist)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.nanmean(np.diag(hist) / hist.sum(1))


class ConfusionMeter(object):
    """Device-side accumulation of eval.py:96-138 (``calc_all_metrics``): feed (prediction, ground truth) label batches,
    read the reference's summary numbers at the end."""

    def __init__(self, n_class, background_id=255, consider_background_loss=False, device=None):
        self.n_class, self.background_id, self.consider_background = n_class, background_id, consider_background_loss
        self.hist = torch.zeros((n_class, n_class), dtype=torch.int64, device=device or torch.device("cuda", torch.cuda.current_device()))

    def update(self, pred, gt):
        """``gt`` still carries ``background_id`` (255): those pixels fall outside [0,n) and are dropped by fast_hist,
        which is what eval.py:123-129 does by hand.  With ``consider_background_loss`` they count as class n-1
        (eval.py:100-102 ``bg_mapping``)."""
        gt = gt.to(self.hist.device).long()
        if self.consider_background:
            gt = torch.where(gt == self.background_id, torch.full_like(gt, self.n_class - 1), gt)
        fast_hist(gt, pred.to(self.hist.device), self.n_class, out=self.hist)

    def summary(self):
        hist = _np(self.hist)
        used = np.where(hist.sum(1) != 0)[0]  # only classes present in the ground truth (eval.py:143-144)
        sub = hist[used][:, used]
        iou = per_class_iu(sub)
        return {"used_class_ids": used.tolist(), "IoU": (iou * 100).tolist(), "pixAcc": 100 * calc_pixel_accuracy(sub),
                "mAcc": 100 * calc_mean_accuracy(sub), "fwIoU": 100 * calc_fw_iu(sub), "mIoU": 100 * float(iou.mean()) if len(iou) else float("nan"),
                "pred_distribution": hist.sum(0)[used].tolist(), "gt_distribution": hist.sum(1)[used].tolist()}
