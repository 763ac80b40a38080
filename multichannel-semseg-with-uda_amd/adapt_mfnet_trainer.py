#!/usr/bin/env python3
"""MCD trainer for the two-encoder MFNet (RGB encoder + HHA/depth encoder, fused classifiers) -- the reference's
``adapt_mfnet_trainer.py`` (:22-270) on the MI355X HIP kernels.

    python adapt_mfnet_trainer.py suncg nyu --input_ch 6 --method_detail MFNet-ScoreAddFusion -b 16 --synthetic --no_pretrained
"""
import os

import torch
import tqdm

from argmyparse import add_additional_params_to_args, get_da_mcd_training_parser
from datasets import check_src_tgt_ok
from loss import CrossEntropyLoss2d, ProbCrossEntropyLoss2d, get_prob_distance_criterion
from models.model_util import fix_batchnorm_when_training, fix_dropout_when_training, get_models, get_optimizer
from solvers.solver import MFNetMCDSolver
from trainer_common import Run, make_loader
from util import adjust_learning_rate, emphasize_str, get_class_weight_from_file, load_checkpoint, mkdir_if_not_exist


def build(args, detailed_method):
    g3, g1, f1, f2 = get_models(net_name=args.net, res=args.res, input_ch=args.input_ch, n_class=args.n_class,
                                method=detailed_method, is_data_parallel=args.is_data_parallel)
    optimizer_g = get_optimizer(list(g3.parameters()) + list(g1.parameters()), lr=args.lr, opt=args.opt,
                                momentum=args.momentum, weight_decay=args.weight_decay)
    optimizer_f = get_optimizer(list(f1.parameters()) + list(f2.parameters()), lr=args.lr, opt=args.opt,
                                momentum=args.momentum, weight_decay=args.weight_decay)
    return g3, g1, f1, f2, optimizer_g, optimizer_f


def main(argv=None):
    parser = get_da_mcd_training_parser()
    parser.add_argument("--method_detail", type=str, default="MFNet-AddFusion",
                        help="MFNet-{Add,Gate,Concat,ConcatConv}Fusion, MFNet-Score{Add,Gate}Fusion")
    args = parser.parse_args(argv)
    args = add_additional_params_to_args(args)
    check_src_tgt_ok(args.src_dataset, args.tgt_dataset)
    run = Run(args)
    detailed_method = args.method_detail

    resume_flg = bool(args.resume)
    start_epoch = 0
    if args.resume:
        if not os.path.exists(args.resume):
            raise OSError("%s does not exist!" % args.resume)
        cli = args
        checkpoint = load_checkpoint(args.resume)
        start_epoch = checkpoint["epoch"]
        args = checkpoint["args"]
        for k in ("synthetic", "synthetic_raw", "synthetic_len", "seed", "no_pretrained", "solver", "no_tflog", "method_detail"):
            if k not in vars(args):
                setattr(args, k, getattr(cli, k))
        detailed_method = args.method_detail
        g3, g1, f1, f2, optimizer_g, optimizer_f = build(args, detailed_method)
        g3.load_state_dict(checkpoint["g_3ch_state_dict"])
        g1.load_state_dict(checkpoint["g_1ch_state_dict"])
        f1.load_state_dict(checkpoint["f1_state_dict"])
        if not args.uses_one_classifier:
            f2.load_state_dict(checkpoint["f2_state_dict"])
        for m in (g3, g1, f1, f2):
            m.to(run.device)
        optimizer_g.load_state_dict(checkpoint["optimizer_g"])
        optimizer_f.load_state_dict(checkpoint["optimizer_f"])
    else:
        g3, g1, f1, f2, optimizer_g, optimizer_f = build(args, detailed_method)
    if args.uses_one_classifier:
        print("f1 and f2 are same!")
        f2 = f1

    mode = "%s-%s2%s-%s_%sch_MFNet" % (args.src_dataset, args.src_split, args.tgt_dataset, args.tgt_split, args.input_ch)
    if args.net in ["fcn", "psp"]:
        model_name = "%s-%s-%s-res%s" % (detailed_method, args.savename, args.net, args.res)
    else:
        model_name = "%s-%s-%s" % (detailed_method, args.savename, args.net)
    outdir = os.path.join(args.base_outdir, mode)
    pth_dir = os.path.join(outdir, "pth")
    if run.is_main:
        mkdir_if_not_exist(pth_dir)
    run.configure_logger(os.path.join(outdir, "tflog", model_name), args)
    run.save_params(args, os.path.join(outdir, "param-%s%s.json" % (model_name, "_resume" if resume_flg else "")))

    train_loader = make_loader(args, run, [(args.src_dataset, args.src_split), (args.tgt_dataset, args.tgt_split)])
    weight = get_class_weight_from_file(n_class=args.n_class, weight_filename=args.loss_weights_file, add_bg_loss=args.add_bg_loss)
    for m in (g3, g1, f1, f2):
        m.to(run.device)
    weight = weight.to(run.device)
    run.sync_replicas([g3, g1, f1, f2])
    # the gated fusions are paired with the probability-input criterion (adapt_mfnet_trainer.py:149)
    criterion = CrossEntropyLoss2d(weight) if "Gate" not in detailed_method else ProbCrossEntropyLoss2d(weight)
    criterion_d = get_prob_distance_criterion(args.d_loss, n_class=args.n_class)  # symkl needs the row length (the reference passes none and fails there)
    for m in (g3, g1, f1, f2):
        m.train()
    if args.no_dropout:
        for m in (g3, g1, f1, f2):
            fix_dropout_when_training(m)
    if args.fix_bn:
        emphasize_str("BN layers are NOT trained!")
        for m in (g3, g1, f1, f2):
            fix_batchnorm_when_training(m)

    solver = MFNetMCDSolver(g3, g1, f1, f2, optimizer_g, optimizer_f, criterion, criterion_d, num_k=args.num_k)
    for epoch in range(start_epoch, args.epochs):
        d_loss_per_epoch = 0.0
        c_loss_per_epoch = 0.0
        it = enumerate(train_loader)
        for ind, (source, target) in (tqdm.tqdm(it) if run.is_main else it):
            src_imgs = run.images(source[0])
            src_lbls = run.labels(source[1])
            tgt_imgs = run.images(target[0])
            c_loss, d_loss = solver.step(src_imgs, src_lbls, tgt_imgs)
            c_loss, d_loss = float(c_loss), float(d_loss)
            c_loss_per_epoch += c_loss
            d_loss_per_epoch += d_loss
            if ind % 100 == 0 and run.is_main:
                print("iter [%d] DLoss: %.4f CLoss: %.4f" % (ind, d_loss, c_loss))
            if ind > args.max_iter:
                break
        if run.is_main:
            print("Epoch [%d] DLoss: %.4f CLoss: %.4f" % (epoch, d_loss_per_epoch, c_loss_per_epoch))
        run.log_value("c_loss", c_loss_per_epoch, epoch)
        run.log_value("d_loss", d_loss_per_epoch, epoch)
        run.log_value("lr", args.lr, epoch)
        if args.adjust_lr:
            args.lr = adjust_learning_rate(optimizer_g, args.lr, args.weight_decay, epoch, args.epochs)
            args.lr = adjust_learning_rate(optimizer_f, args.lr, args.weight_decay, epoch, args.epochs)
        checkpoint_fn = os.path.join(pth_dir, "%s-%s.pth.tar" % (model_name, epoch + 1))
        args.start_epoch = epoch + 1
        save_dic = {
            "epoch": epoch + 1,
            "args": args,
            "g_3ch_state_dict": g3.state_dict(),
            "g_1ch_state_dict": g1.state_dict(),
            "f1_state_dict": f1.state_dict(),
            "optimizer_g": optimizer_g.state_dict(),
            "optimizer_f": optimizer_f.state_dict(),
        }
        if not args.uses_one_classifier:
            save_dic["f2_state_dict"] = f2.state_dict()
        run.save(save_dic, checkpoint_fn)
    return 0


if __name__ == "__main__":
    main()
