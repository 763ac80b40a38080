#!/usr/bin/env python3
"""Golden vectors for the resize step of the input pipeline: ``Scale(img_shape, Image.BILINEAR)`` for images and
``Scale(img_shape, Image.NEAREST)`` for label maps (/root/reference/transform.py:303, 320).  ``Scale`` is torchvision's (absent
here) thin wrapper over ``PIL.Image.resize``, so the vectors are produced by calling the real Pillow (version recorded in the
file) on seeded uint8 images; the oracle restatement (oracle/ref_io.py) is asserted against them while generating.
Only data is written: tests/golden/resize_small.npz.     python tests/golden/make_golden_resize.py
"""
import os
import sys

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_io  # noqa: E402

# (tag, in H, in W, out W, out H): down, up, mixed, non-integer factors, the NYU -> 240x320 case of BASELINE config 1
CASES = [("down2", 48, 64, 32, 24), ("up", 15, 20, 47, 33), ("mixed", 30, 17, 40, 11), ("frac", 37, 53, 31, 29),
         ("same_w", 24, 32, 32, 17), ("cfg1", 96, 128, 64, 48)]


def main():
    rng = np.random.RandomState(31337)
    out = {"pillow_version": np.array(PIL.__version__)}
    for tag, h, w, ow, oh in CASES:
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        img[:4, :4] = 255
        img[-3:, -3:] = 0
        lbl = rng.randint(0, 41, size=(h, w)).astype(np.uint8)
        lbl[rng.rand(h, w) < 0.1] = 255
        ref_img = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        ref_lbl = np.asarray(Image.fromarray(lbl).resize((ow, oh), Image.NEAREST))
        assert ref_img.shape == (oh, ow, 3) and ref_lbl.shape == (oh, ow)
        got_img, got_lbl = ref_io.resize_bilinear_u8(img, (ow, oh)), ref_io.resize_nearest_u8(lbl, (ow, oh))
        assert np.array_equal(got_img, ref_img), (tag, "bilinear", int(np.abs(got_img.astype(int) - ref_img).max()))
        assert np.array_equal(got_lbl, ref_lbl), (tag, "nearest")
        out.update({"img_" + tag: img, "lbl_" + tag: lbl, "size_" + tag: np.array([ow, oh]), "rimg_" + tag: ref_img, "rlbl_" + tag: ref_lbl})
        print("  %-7s %dx%d -> %dx%d ok" % (tag, w, h, ow, oh))
    np.savez_compressed(os.path.join(HERE, "resize_small.npz"), **out)
    print("resize_small.npz written (Pillow %s)" % PIL.__version__)


if __name__ == "__main__":
    main()
