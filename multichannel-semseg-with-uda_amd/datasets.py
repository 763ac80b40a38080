"""Dataset layer of the hot path.

The reference reads nine on-disk datasets from hard-coded lab paths (datasets.py:981-992); none of that is
on the MI355X path.  What the trainers need from it is kept with the reference's names and values --
``AVAILABLE_DATASET_LIST``, ``get_n_class`` (:1011-1024), ``get_img_shape`` (:1027-1043, W,H order),
``ConcatDataset`` (:20-28), ``check_src_tgt_ok`` (:1004-1008) -- plus ``SyntheticRGBD``, the seeded
stand-in the BASELINE configs run on: N(0,1) images in place of ImageNet-normalised
RGB+HHA (transform.py:307) and uniform labels with class n_class-1 as the zero-weight background
(transform.py:319-325 maps 255 there).
"""
import torch
from torch.utils import data

AVAILABLE_DATASET_LIST = ["gta", "city", "test", "ir", "city16", "synthia", "2d3d", "sun", "suncg", "nyu"]

_IMG_SIZE = {  # (W, H) class constants of the reference's dataset classes
    "gta": [1280, 720], "synthia": [1280, 760], "city": [2048, 1024], "city16": [2048, 1024], "ir": [640, 480],
    "2d3d": [1080, 1080], "sun": [640, 480], "suncg": [640, 480], "nyu": [640, 480],
}


class ConcatDataset(data.Dataset):
    def __init__(self, *datasets):
        self.datasets = datasets

    def __getitem__(self, i):
        return tuple(d[i] for d in self.datasets)

    def __len__(self):
        return min(len(d) for d in self.datasets)


class SyntheticRGBD(data.Dataset):
    """(image [C,H,W] fp32, label [H,W] int64) generated from (seed, index)."""

    def __init__(self, length, input_ch, img_shape_wh, n_class, seed, test=False):
        self.length, self.ch, self.n_class, self.seed, self.test = length, input_ch, n_class, seed, test
        self.w, self.h = int(img_shape_wh[0]), int(img_shape_wh[1])

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        img = torch.randn(self.ch, self.h, self.w, generator=g)
        lbl = torch.randint(0, self.n_class, (self.h, self.w), generator=g, dtype=torch.int64)
        if self.test:
            return img, lbl, "synthetic_%06d.png" % i
        return img, lbl


def get_dataset(dataset_name, split, img_transform, label_transform, test, input_ch=3, joint_transform=None,
                synthetic=None):
    """``synthetic`` = dict(length, img_shape, n_class, seed) selects the generated data; the on-disk datasets of
    the reference are outside this build."""
    assert dataset_name in AVAILABLE_DATASET_LIST
    if synthetic is None:
        raise NotImplementedError("on-disk dataset %r is outside the MI355X hot-path build; run with --synthetic"
                                  % dataset_name)
    return SyntheticRGBD(synthetic["length"], input_ch, synthetic["img_shape"], synthetic["n_class"], synthetic["seed"], test)


def check_src_tgt_ok(src_dataset_name, tgt_dataset_name):
    if src_dataset_name == "synthia" and not tgt_dataset_name == "city16":
        raise AssertionError("you must use synthia-city16 pair")
    elif src_dataset_name == "city16" and not tgt_dataset_name == "synthia":
        raise AssertionError("you must use synthia-city16 pair")


def get_n_class(src_dataset_name):
    if src_dataset_name in ["synthia", "city16"]:
        return 16
    elif src_dataset_name in ["gta", "city", "ir", "test"]:
        return 19 + 1
    elif src_dataset_name in ["2d3d", "sun"]:
        return 13 + 1
    elif src_dataset_name in ["suncg", "nyu"]:
        return 40 + 1
    raise NotImplementedError("You have to define the class of %s dataset" % src_dataset_name)


def get_img_shape(dataset, is_train):
    shape = list(_IMG_SIZE[dataset])
    if is_train and dataset in ("city", "city16"):
        shape = [s / 2 for s in shape]
    return shape
