"""Command-line flags of the trainers -- names, defaults and choices of the reference's ``argmyparse.py``
(:35-137), plus the switches this build adds (synthetic data, no download, solver choice, distributed)."""
import argparse
import os

from datasets import AVAILABLE_DATASET_LIST, get_img_shape, get_n_class

AVAILABLE_NET_LIST = ["fcn", "psp", "segnet", "fcnvgg", "drn_c_26", "drn_c_42", "drn_c_58", "drn_d_22", "drn_d_38",
                      "drn_d_54", "drn_d_105", "unet", "fusenet"]
AVAILABLE_NET_LIST = AVAILABLE_NET_LIST + [x + "_ver2" for x in AVAILABLE_NET_LIST if "drn" in x]
AVAILABLE_NET_LIST = AVAILABLE_NET_LIST + [x + "_fusenet" for x in AVAILABLE_NET_LIST if "drn" in x]


def add_additional_params_to_args(args):
    """n_class / machine / default image shapes derived from the dataset names (argmyparse.py:14-32)."""
    keys = vars(args)
    dataset = args.src_dataset if "src_dataset" in keys else args.tgt_dataset
    args.n_class = get_n_class(dataset)
    args.machine = os.uname()[1]
    if "src_dataset" in keys and "train_img_shape" in keys and args.train_img_shape is None:
        args.train_img_shape = get_img_shape(args.src_dataset, is_train=True)
        print("args.train_img_shape is set to %s" % args.train_img_shape)
    if "tgt_dataset" in keys and "test_img_shape" in keys and args.test_img_shape is None:
        args.test_img_shape = get_img_shape(args.tgt_dataset, is_train=False)
        print("args.test_img_shape is set to %s" % args.test_img_shape)
    return args


def add_mi355x_flags(parser):
    g = parser.add_argument_group("MI355X build")
    g.add_argument("--synthetic", action="store_true", help="seeded synthetic RGB+HHA batches instead of on-disk datasets")
    g.add_argument("--synthetic_len", type=int, default=64, help="samples per synthetic dataset")
    g.add_argument("--synthetic_raw", action="store_true",
                   help="synthetic uint8 HWC images + uint8 labels (background 255): ToTensor/Normalize/ReLabel run on the GPU")
    g.add_argument("--src_file_list", default=None,
                   help="real data: a list file with 'rgb_path [hha_path] label_path' per line (datasets.FileListRGBD); files are "
                        "read with PIL, Scale / ToTensor / Normalize / ReLabel run on the GPU at --train_img_shape")
    g.add_argument("--tgt_file_list", default=None, help="the same for the target domain (its labels are loaded but not used)")
    g.add_argument("--seed", type=int, default=1234)
    g.add_argument("--no_pretrained", action="store_true", help="He-normal init instead of ImageNet weights (no network)")
    g.add_argument("--solver", choices=["fused", "dropin"], default="fused",
                   help="fused: solvers.solver (one loss kernel per phase, step-B generator backward elided); "
                        "dropin: the reference's statement-for-statement loop over the drop-in modules")
    g.add_argument("--no_tflog", action="store_true", help="do not require tensorboard_logger")
    g.add_argument("--dtype", choices=["f32", "f16"], default="f32",
                   help="arithmetic of the convolutions: f32 = fp32-grade (operands as two scaled fp16 pieces, three MFMA terms; the default, "
                        "within 1e-3 of the reference), f16 = reduced precision (BASELINE config 5's 'bf16': the leading fp16 piece only, one MFMA term, and "
                        "2-byte activation storage inside the trunk)")
    return parser


def get_common_training_parser(parser):
    parser.add_argument("--savename", type=str, default="normal", help="save name(Do NOT use '-')")
    parser.add_argument("--base_outdir", type=str, default="train_output", help="base output dir")
    parser.add_argument("--epochs", type=int, default=40, help="number of epochs to train (default: 40)")
    parser.add_argument("--max_iter", type=int, default=5000)
    parser.add_argument("--net", type=str, default="drn_d_38", help="network structure", choices=AVAILABLE_NET_LIST)
    parser.add_argument("--res", type=str, default="50", metavar="ResnetLayerNum", choices=["18", "34", "50", "101", "152"])
    parser.add_argument("--is_data_parallel", action="store_true", help="wrap models so checkpoints carry the module. prefix")
    parser.add_argument("--opt", type=str, default="sgd", choices=["sgd", "adam"], help="network optimizer")
    parser.add_argument("--lr", type=float, default=1e-3, help="learning rate (default: 0.001)")
    parser.add_argument("--adjust_lr", action="store_true", help="whether you change lr")
    parser.add_argument("--momentum", type=float, default=0.9, help="momentum sgd (default: 0.9)")
    parser.add_argument("--weight_decay", type=float, default=2e-5, help="weight_decay (default: 2e-5)")
    parser.add_argument("-b", "--batch_size", type=int, default=1, help="batch_size (per GPU)")
    parser.add_argument("--normalize_way", type=str, default="imagenet", choices=["imagenet", "None"])
    parser.add_argument("--crop_size", type=int, default=-1)
    parser.add_argument("--rotate_angle", type=int, default=0)
    parser.add_argument("--loss_weights_file", type=str, default=None)
    parser.add_argument("--add_bg_loss", action="store_true")
    parser.add_argument("--fix_bn", action="store_true")
    parser.add_argument("--no_dropout", action="store_true")
    parser.add_argument("--input_ch", type=int, default=3, choices=[1, 3, 4, 6])
    parser.add_argument("--train_img_shape", default=None, nargs=2, metavar=("W", "H"), type=int, help="W H")
    parser.add_argument("--background_id", type=int, default=255)
    parser.add_argument("--resume", type=str, default=None, metavar="PTH.TAR", help="model(pth) path")
    return add_mi355x_flags(parser)


def get_src_only_training_parser(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(description="PyTorch Segmentation Adaptation")
    parser.add_argument("src_dataset", type=str, choices=AVAILABLE_DATASET_LIST)
    parser.add_argument("--split", type=str, default="train")
    return get_common_training_parser(parser)


def get_da_base_training_parser(parser=None):
    if parser is None:
        parser = argparse.ArgumentParser(description="PyTorch Segmentation Adaptation")
    parser.add_argument("src_dataset", type=str, choices=AVAILABLE_DATASET_LIST)
    parser.add_argument("tgt_dataset", type=str, choices=AVAILABLE_DATASET_LIST)
    parser.add_argument("--src_split", type=str, default="train")
    parser.add_argument("--tgt_split", type=str, default="train")
    return get_common_training_parser(parser)


def get_da_mcd_training_parser():
    parser = get_da_base_training_parser()
    parser.add_argument("--method", type=str, default="MCD", help="Method Name")
    parser.add_argument("--num_k", type=int, default=4, help="how many steps to repeat the generator update")
    parser.add_argument("--num_multiply_d_loss", type=int, default=1)
    parser.add_argument("--d_loss", type=str, default="diff",
                        choices=["jsd", "mysymkl", "spatial_jsd", "symkl", "diff", "nmlsymkl", "strange_kl", "mis_symkl"])
    parser.add_argument("--uses_one_classifier", action="store_true", help="separate f1, f2")
    return parser


def get_da_mcd_testing_parser():
    """flags of the reference's testers (argmyparse.py:146-160)"""
    parser = argparse.ArgumentParser(description="Adapt tester for validation data")
    parser.add_argument("tgt_dataset", type=str, choices=AVAILABLE_DATASET_LIST)
    parser.add_argument("trained_checkpoint", type=str, metavar="PTH.TAR")
    parser.add_argument("--split", type=str, default="val", help="'val' or 'test') is used")
    parser.add_argument("--outdir", type=str, default="test_output", help="output directory")
    parser.add_argument("--test_img_shape", default=None, nargs=2, type=int, help="W H")
    parser.add_argument("--saves_prob", action="store_true", help="whether you save probability tensors")
    parser.add_argument("--use_f2", action="store_true", help="whether you use f2")
    g = parser.add_argument_group("MI355X build")
    g.add_argument("--synthetic", action="store_true")
    g.add_argument("--synthetic_len", type=int, default=4)
    g.add_argument("--seed", type=int, default=4321)
    g.add_argument("-b", "--batch_size", type=int, default=1)
    return parser
