#!/usr/bin/env python3
"""MCD + HHA-regression multitask trainer -- the reference's ``adapt_multitask_trainer.py`` (:21-273) on the
MI355X HIP kernels: RGB encoder, two segmentation decoders + one depth decoder, learned task weights.

    python adapt_multitask_trainer.py suncg nyu --input_ch 6 -b 8 --synthetic --no_pretrained
"""
import os

import torch
import tqdm

from argmyparse import add_additional_params_to_args, get_da_mcd_training_parser
from datasets import check_src_tgt_ok
from loss import CrossEntropyLoss2d, get_prob_distance_criterion
from models.model_util import fix_batchnorm_when_training, fix_dropout_when_training, get_multitask_models, get_optimizer
from solvers.solver import MultiTaskMCDSolver
from trainer_common import Run, make_loader
from util import adjust_learning_rate, emphasize_str, get_class_weight_from_file, load_checkpoint, mkdir_if_not_exist


def build(args, criterion, criterion_d):
    model_enc, model_dec = get_multitask_models(net_name=args.net, input_ch=args.input_ch, n_class=args.n_class,
                                                is_data_parallel=args.is_data_parallel, semseg_criterion=criterion,
                                                discrepancy_criterion=criterion_d)
    optimizer_enc = get_optimizer(model_enc.parameters(), lr=args.lr, momentum=args.momentum, opt=args.opt,
                                  weight_decay=args.weight_decay)
    optimizer_dec = get_optimizer(model_dec.parameters(), opt=args.opt, lr=args.lr, momentum=args.momentum,
                                  weight_decay=args.weight_decay)
    return model_enc, model_dec, optimizer_enc, optimizer_dec


def main(argv=None):
    args = get_da_mcd_training_parser().parse_args(argv)
    args = add_additional_params_to_args(args)
    check_src_tgt_ok(args.src_dataset, args.tgt_dataset)
    if args.input_ch <= 3:
        raise SystemExit("the multitask trainer regresses the channels after RGB: --input_ch must be 4 or 6")
    run = Run(args)

    def criteria(a):
        w = get_class_weight_from_file(n_class=a.n_class, weight_filename=a.loss_weights_file, add_bg_loss=a.add_bg_loss)
        return CrossEntropyLoss2d(w), get_prob_distance_criterion(a.d_loss, n_class=a.n_class)

    resume_flg = bool(args.resume)
    start_epoch = 0
    if args.resume:
        if not os.path.exists(args.resume):
            raise OSError("%s does not exist!" % args.resume)
        cli = args
        checkpoint = load_checkpoint(args.resume)
        start_epoch = checkpoint["epoch"]
        args = checkpoint["args"]
        for k in ("synthetic", "synthetic_raw", "synthetic_len", "src_file_list", "tgt_file_list", "seed", "no_pretrained", "solver", "no_tflog"):
            if k not in vars(args):
                setattr(args, k, getattr(cli, k))
        criterion, criterion_d = criteria(args)
        model_enc, model_dec, optimizer_enc, optimizer_dec = build(args, criterion, criterion_d)
        model_enc.load_state_dict(checkpoint["enc_state_dict"])
        model_dec.load_state_dict(checkpoint["dec_state_dict"])
        model_enc.to(run.device), model_dec.to(run.device)
        optimizer_enc.load_state_dict(checkpoint["optimizer_enc"])
        optimizer_dec.load_state_dict(checkpoint["optimizer_dec"])
    else:
        criterion, criterion_d = criteria(args)
        model_enc, model_dec, optimizer_enc, optimizer_dec = build(args, criterion, criterion_d)

    mode = "%s-%s2%s-%s_%sch_MCDmultitask" % (args.src_dataset, args.src_split, args.tgt_dataset, args.tgt_split, args.input_ch)
    if args.net in ["fcn", "psp"]:
        model_name = "%s-%s-%s-res%s" % (args.method, args.savename, args.net, args.res)
    else:
        model_name = "%s-%s-%s" % (args.method, args.savename, args.net)
    outdir = os.path.join(args.base_outdir, mode)
    pth_dir = os.path.join(outdir, "pth")
    if run.is_main:
        mkdir_if_not_exist(pth_dir)
    run.configure_logger(os.path.join(outdir, "tflog", model_name), args)
    run.save_params(args, os.path.join(outdir, "param-%s%s.json" % (model_name, "_resume" if resume_flg else "")))

    train_loader = make_loader(args, run, [(args.src_dataset, args.src_split), (args.tgt_dataset, args.tgt_split)])
    model_enc.to(run.device), model_dec.to(run.device)
    run.sync_replicas([model_enc, model_dec])
    model_enc.train(), model_dec.train()
    if args.no_dropout:
        fix_dropout_when_training(model_enc), fix_dropout_when_training(model_dec)
    if args.fix_bn:
        emphasize_str("BN layers are NOT trained!")
        fix_batchnorm_when_training(model_enc), fix_batchnorm_when_training(model_dec)

    solver = MultiTaskMCDSolver(model_enc, model_dec, optimizer_enc, optimizer_dec, num_k=args.num_k,
                                num_multiply_d_loss=args.num_multiply_d_loss)
    for epoch in range(start_epoch, args.epochs):
        sums = dict(c=0.0, d=0.0, seg=0.0, sdep=0.0, tdep=0.0)
        it = enumerate(train_loader)
        for ind, (source, target) in (tqdm.tqdm(it) if run.is_main else it):
            src_imgs = run.images(source[0])
            src_gt = run.labels(source[1])
            tgt_imgs = run.images(target[0])
            c_loss, d_loss, parts = solver.step(src_imgs, src_gt, tgt_imgs)
            c_loss, d_loss = float(c_loss), float(d_loss)
            sums["c"] += c_loss
            sums["d"] += d_loss
            sums["seg"] += float(parts[0]); sums["sdep"] += float(parts[1]); sums["tdep"] += float(parts[2])
            if ind % 100 == 0 and run.is_main:
                print("iter [%d] DLoss: %.6f CLoss: %.4f" % (ind, d_loss, c_loss))
            if ind > args.max_iter:
                break
        dec = model_dec.module if hasattr(model_dec, "module") else model_dec
        std_semseg, std_depth = dec.get_task_weights()
        if run.is_main:
            print("std_semseg: %.4f, std_depth: %.4f" % (float(std_semseg.reshape(-1)[0]), float(std_depth.reshape(-1)[0])))  # (1-element arrays, as the reference returns them)
            print("Epoch [%d] DLoss: %.4f CLoss: %.4f" % (epoch, sums["d"], sums["c"]))
        for name, key in (("c_loss", "c"), ("d_loss", "d"), ("src_semseg_loss", "seg"), ("src_depth_loss", "sdep"),
                          ("tgt_depth_loss", "tdep")):
            run.log_value(name, sums[key], epoch)
        run.log_value("lr", args.lr, epoch)
        if args.adjust_lr:
            args.lr = adjust_learning_rate(optimizer_enc, args.lr, args.weight_decay, epoch, args.epochs)
            args.lr = adjust_learning_rate(optimizer_dec, args.lr, args.weight_decay, epoch, args.epochs)
        checkpoint_fn = os.path.join(pth_dir, "%s-%s.pth.tar" % (model_name, epoch + 1))
        args.start_epoch = epoch + 1
        run.save({"epoch": epoch + 1, "args": args, "enc_state_dict": model_enc.state_dict(),
                  "dec_state_dict": model_dec.state_dict(), "optimizer_enc": optimizer_enc.state_dict(),
                  "optimizer_dec": optimizer_dec.state_dict()}, checkpoint_fn)
    return 0


if __name__ == "__main__":
    main()
