"""Dataset layer of the hot path.

The reference reads nine on-disk datasets from hard-coded lab paths (datasets.py:981-992); none of that is
on the MI355X path.  What the trainers need from it is kept with the reference's names and values --
``AVAILABLE_DATASET_LIST``, ``get_n_class`` (:1011-1024), ``get_img_shape`` (:1027-1043, W,H order),
``ConcatDataset`` (:20-28), ``check_src_tgt_ok`` (:1004-1008) -- plus ``SyntheticRGBD``, the seeded
stand-in the BASELINE configs run on: N(0,1) images in place of ImageNet-normalised
RGB+HHA (transform.py:307) and uniform labels with class n_class-1 as the zero-weight background
(transform.py:319-325 maps 255 there) -- and ``FileListRGBD``, a loader for real data: image / HHA / label files named in a list
are opened with PIL (``default_loader``, datasets.py:20-28) and handed over as raw uint8 arrays; everything the reference's
transforms then do (Scale, ToTensor, Normalize, ToLabel, ReLabel; transform.py:302-325) runs on the MI355X
(``DeviceInputPipeline``).
"""
import numpy as np
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
    """(image [C,H,W] fp32, label [H,W] int64) generated from (seed, index).

    ``raw=True`` yields what a loader holds BEFORE the reference's transforms: a uint8 [H,W,C] image (RGB then HHA
    bytes) and a uint8 label map whose background pixels carry ``background_id`` -- the trainer then runs
    ToTensor/Normalize/ReLabel on the MI355X (``DeviceInputPipeline``)."""

    def __init__(self, length, input_ch, img_shape_wh, n_class, seed, test=False, raw=False, background_id=255):
        self.length, self.ch, self.n_class, self.seed, self.test = length, input_ch, n_class, seed, test
        self.w, self.h = int(img_shape_wh[0]), int(img_shape_wh[1])
        self.raw, self.background_id = raw, background_id

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        if self.raw:
            img = torch.randint(0, 256, (self.h, self.w, self.ch), generator=g, dtype=torch.uint8)
            lbl = torch.randint(0, self.n_class, (self.h, self.w), generator=g, dtype=torch.int64)
            lbl = torch.where(lbl == self.n_class - 1, torch.full_like(lbl, self.background_id), lbl).to(torch.uint8)
        else:
            img = torch.randn(self.ch, self.h, self.w, generator=g)
            lbl = torch.randint(0, self.n_class, (self.h, self.w), generator=g, dtype=torch.int64)
        if self.test:
            return img, lbl, "synthetic_%06d.png" % i
        return img, lbl


IMAGENET_MEAN = [.485, .456, .406, .485, .485, .485]  # 6-channel RGB+HHA statistics of transform.py:307
IMAGENET_STD = [.229, .224, .225, .229, .229, .229]
CITY_MEAN = [0.290101, 0.328081, 0.286964]            # transform.py:311
CITY_STD = [0.182954, 0.186566, 0.184475]


def default_loader(path):
    """datasets.py:27-28 of the reference"""
    from PIL import Image
    return Image.open(path)


class FileListRGBD(data.Dataset):
    """Real data for the trainers: ``list_file`` has one sample per line, ``rgb_path [hha_path] label_path`` (whitespace
    separated; relative paths are taken from the list's directory).  ``__getitem__`` returns what the reference's loaders hold
    BEFORE their transforms -- a uint8 [H,W,3 or 6] image (RGB, then HHA) and a uint8 [H,W] label map straight from the PNGs
    (``default_loader``; palette / grey label PNGs are read as their index / grey values, as ``ToLabel`` sees them) -- so that
    resize, normalisation and relabelling run on the device.  All files of one list must share one size (batches are stacked)."""

    def __init__(self, list_file, input_ch=6, test=False):
        import os
        self.root = os.path.dirname(os.path.abspath(list_file))
        self.items = []
        for line in open(list_file):
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            if len(parts) not in (2, 3):
                raise ValueError("FileListRGBD: expected 'rgb [hha] label' per line, got %r" % line)
            self.items.append([p if os.path.isabs(p) else os.path.join(self.root, p) for p in parts])
        if not self.items:
            raise ValueError("FileListRGBD: %s names no samples" % list_file)
        self.input_ch, self.test = input_ch, test
        if input_ch > 3 and any(len(it) != 3 for it in self.items):
            raise ValueError("FileListRGBD: input_ch=%d needs an HHA / depth file per sample" % input_ch)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        it = self.items[i]
        rgb = np.array(default_loader(it[0]).convert("RGB"), dtype=np.uint8)
        if self.input_ch > 3:
            extra = default_loader(it[1])
            extra = np.array(extra.convert("RGB") if self.input_ch == 6 else extra.convert("L"), dtype=np.uint8)
            extra = extra.reshape(extra.shape[0], extra.shape[1], -1)[:, :, :self.input_ch - 3]
            img = np.concatenate([rgb, extra], axis=2)
        else:
            img = rgb[:, :, :self.input_ch]
        lbl = np.array(default_loader(it[-1]), dtype=np.uint8)  # mode "P" / "L": the class indices themselves
        if lbl.ndim != 2:
            raise ValueError("FileListRGBD: label image %s is not single-channel" % it[-1])
        out = (torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(np.ascontiguousarray(lbl)))
        return out + (it[-1],) if self.test else out


class DeviceInputPipeline(object):
    """``get_img_transform`` / ``get_lbl_transform`` (transform.py:302-325) on the GPU: pinned uint8 batches go over PCIe as
    bytes (4x fewer than fp32); ``Scale`` (bilinear for images, nearest for label maps: Pillow's 8-bit arithmetic, bit for bit),
    ToTensor+Normalize (HWC->NCHW) and ToLabel+ReLabel(background_id -> n_class-1) each are one kernel.  ``img_shape`` = (W, H)
    as the reference passes it to ``Scale``; None leaves the size alone (the loader already produced it)."""

    def __init__(self, input_ch, n_class, device, normalize_way="imagenet", background_id=255, img_shape=None):
        from mcdseg import ops
        self._ops = ops
        self.input_ch, self.n_class, self.device, self.background_id = input_ch, n_class, device, background_id
        self.img_shape = None if img_shape is None else (int(img_shape[0]), int(img_shape[1]))
        if normalize_way == "imagenet":
            mean, std = IMAGENET_MEAN[:input_ch], IMAGENET_STD[:input_ch]
        elif normalize_way == "city":
            mean, std = CITY_MEAN[:input_ch], CITY_STD[:input_ch]
        else:  # "No normalization..." (transform.py:314-315): ToTensor only
            mean, std = [0.0] * input_ch, [1.0] * input_ch
        self.mean = torch.tensor(mean, dtype=torch.float32, device=device)
        self.std = torch.tensor(std, dtype=torch.float32, device=device)

    def images(self, *parts):
        """one or more uint8 [N,H,W,c_i] tensors (e.g. RGB and HHA) -> fp32 [N, sum c_i, H, W]"""
        parts = [p.to(self.device, non_blocking=True) for p in parts]
        if self.img_shape is not None:
            parts = [self._ops.resize_u8(p, self.img_shape) for p in parts]
        n, h, w = parts[0].shape[:3]
        c = sum(p.shape[3] for p in parts)
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=self.device)
        off = 0
        for p in parts:
            self._ops.normalize_u8_(out, p, self.mean[off:off + p.shape[3]], self.std[off:off + p.shape[3]], c_off=off)
            off += p.shape[3]
        return out

    def labels(self, lbl_u8):
        lbl_u8 = lbl_u8.to(self.device, non_blocking=True)
        if self.img_shape is not None:
            lbl_u8 = self._ops.resize_u8(lbl_u8, self.img_shape, nearest=True)
        return self._ops.relabel_u8(lbl_u8, self.background_id, self.n_class - 1)


def get_dataset(dataset_name, split, img_transform, label_transform, test, input_ch=3, joint_transform=None,
                synthetic=None, file_list=None):
    """``synthetic`` = dict(length, img_shape, n_class, seed) selects the generated data, ``file_list`` a list file of real
    samples (``FileListRGBD``); the reference's own on-disk dataset classes (hard-coded lab paths) are outside this build."""
    assert dataset_name in AVAILABLE_DATASET_LIST
    if file_list is not None:
        return FileListRGBD(file_list, input_ch=input_ch, test=test)
    if synthetic is None:
        raise NotImplementedError("on-disk dataset %r is outside the MI355X hot-path build; run with --synthetic, or "
                                  "--src_file_list / --tgt_file_list for real data" % dataset_name)
    return SyntheticRGBD(synthetic["length"], input_ch, synthetic["img_shape"], synthetic["n_class"], synthetic["seed"], test,
                         raw=synthetic.get("raw", False), background_id=synthetic.get("background_id", 255))


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
