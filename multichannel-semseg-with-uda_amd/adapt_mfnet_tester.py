#!/usr/bin/env python3
"""MFNet inference -- the reference's ``adapt_mfnet_tester.py`` (:17-141): RGB and HHA encoders, fused classifier F1,
label PNGs + mean entropy; shares the loop of ``adapt_tester.py`` (folded-BN convolutions, argmax/entropy kernel,
device-side confusion matrix).

    python adapt_mfnet_tester.py nyu train_output/.../pth/MCD-normal-drn_d_38-1.pth.tar --synthetic
"""
import adapt_tester


def main(argv=None):
    return adapt_tester.main(argv, mfnet=True)


if __name__ == "__main__":
    main()
