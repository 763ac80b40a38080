#!/usr/bin/env python3
"""Source-only supervised trainer -- the reference's ``source_trainer.py`` (:21-165): DRNSeg wrapped so that
the checkpoint carries DataParallel's ``module.`` prefix, CE loss, SGD; BASELINE config 1 geometry is
``suncg --net drn_d_38 --input_ch 6 -b 2 --train_img_shape 320 240 --synthetic --no_pretrained``."""
import os

import torch
import tqdm

from argmyparse import add_additional_params_to_args, get_src_only_training_parser
from loss import CrossEntropyLoss2d
from models.model_util import fix_batchnorm_when_training, get_full_model, get_optimizer
from trainer_common import Run, make_loader
from util import adjust_learning_rate, emphasize_str, get_class_weight_from_file, load_checkpoint, mkdir_if_not_exist


def main(argv=None):
    args = get_src_only_training_parser().parse_args(argv)
    args = add_additional_params_to_args(args)
    run = Run(args)
    start_epoch = 0
    if args.resume:
        if not os.path.exists(args.resume):
            raise OSError("%s does not exist!" % args.resume)
        cli = args
        checkpoint = load_checkpoint(args.resume)
        args = checkpoint["args"]
        for k in ("synthetic", "synthetic_raw", "synthetic_len", "src_file_list", "tgt_file_list", "seed", "no_pretrained", "solver", "no_tflog"):
            if k not in vars(args):
                setattr(args, k, getattr(cli, k))
        model = get_full_model(net=args.net, res=args.res, n_class=args.n_class, input_ch=args.input_ch)
        optimizer = get_optimizer(model.parameters(), opt=args.opt, lr=args.lr, momentum=args.momentum,
                                  weight_decay=args.weight_decay)
        model.load_state_dict(checkpoint["state_dict"])
        model.to(run.device)
        optimizer.load_state_dict(checkpoint["optimizer"])
        start_epoch = checkpoint["epoch"]
        json_fn = os.path.join(args.outdir, "param_%s_resume.json" % args.savename)
    else:
        model = get_full_model(net=args.net, res=args.res, n_class=args.n_class, input_ch=args.input_ch)
        optimizer = get_optimizer(model.parameters(), opt=args.opt, lr=args.lr, momentum=args.momentum,
                                  weight_decay=args.weight_decay)
        args.outdir = os.path.join(args.base_outdir, "%s-%s_only_%sch" % (args.src_dataset, args.split, args.input_ch))
        args.pth_dir = os.path.join(args.outdir, "pth")
        model_name = "%s-%s-res%s" % (args.savename, args.net, args.res) if args.net in ["fcn", "psp"] else \
            "%s-%s" % (args.savename, args.net)
        args.tflog_dir = os.path.join(args.outdir, "tflog", model_name)
        json_fn = os.path.join(args.outdir, "param-%s.json" % model_name)
    if run.is_main:
        mkdir_if_not_exist(args.pth_dir)
    run.configure_logger(args.tflog_dir, args)
    run.save_params(args, json_fn)

    train_loader = make_loader(args, run, [(args.src_dataset, args.split)])
    weight = get_class_weight_from_file(n_class=args.n_class, weight_filename=args.loss_weights_file, add_bg_loss=args.add_bg_loss)
    model.to(run.device)
    run.sync_replicas([model])
    criterion = CrossEntropyLoss2d(weight.to(run.device))
    model.train()
    if args.fix_bn:
        emphasize_str("BN layers are NOT trained!")
        fix_batchnorm_when_training(model)

    for epoch in range(start_epoch, args.epochs):
        epoch_loss = 0.0
        it = enumerate(train_loader)
        for ind, (images, labels) in (tqdm.tqdm(it) if run.is_main else it):
            imgs = run.images(images)
            lbls = run.labels(labels)
            optimizer.zero_grad()
            preds = model(imgs)
            loss = criterion(preds, lbls)
            loss.backward()
            epoch_loss += float(loss)
            optimizer.step()
            if ind > args.max_iter:
                break
        if run.is_main:
            print("Epoch [%d] Loss: %.4f" % (epoch + 1, epoch_loss))
        run.log_value("loss", epoch_loss, epoch)
        run.log_value("lr", args.lr, epoch)
        if args.adjust_lr:
            args.lr = adjust_learning_rate(optimizer, args.lr, args.weight_decay, epoch, args.epochs)
        if args.net in ("fcn", "psp"):
            checkpoint_fn = os.path.join(args.pth_dir, "%s-%s-res%s-%s.pth.tar" % (args.savename, args.net, args.res, epoch + 1))
        else:
            checkpoint_fn = os.path.join(args.pth_dir, "%s-%s-%s.pth.tar" % (args.savename, args.net, epoch + 1))
        args.start_epoch = epoch + 1
        run.save({"args": args, "epoch": epoch + 1, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()},
                 checkpoint_fn)
    return 0


if __name__ == "__main__":
    main()
