#!/usr/bin/env python3
"""MCD early-fusion trainer -- entry point with the reference's CLI, output layout and checkpoint format
(adapt_trainer.py:21-245), running on the MI355X HIP kernels.

    python adapt_trainer.py suncg nyu --input_ch 6 -b 16 --synthetic --no_pretrained
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 adapt_trainer.py suncg nyu ... (data parallel)

Three-step update per iteration (adapt_trainer.py:155-220): ``--solver fused`` (default) uses
``solvers.solver.MCDSolver``; ``--solver dropin`` runs the reference's statements over the drop-in modules.
"""
import os

import torch
import tqdm

from argmyparse import add_additional_params_to_args, get_da_mcd_training_parser
from datasets import check_src_tgt_ok
from loss import CrossEntropyLoss2d, get_prob_distance_criterion
from models.model_util import fix_batchnorm_when_training, fix_dropout_when_training, get_models, get_optimizer
from solvers.solver import MCDSolver
from trainer_common import Run, make_loader
from util import adjust_learning_rate, emphasize_str, get_class_weight_from_file, load_checkpoint, mkdir_if_not_exist


def build(args):
    model_g, model_f1, model_f2 = get_models(net_name=args.net, res=args.res, input_ch=args.input_ch, n_class=args.n_class,
                                             method=args.method, is_data_parallel=args.is_data_parallel)
    optimizer_g = get_optimizer(model_g.parameters(), lr=args.lr, momentum=args.momentum, opt=args.opt,
                                weight_decay=args.weight_decay)
    optimizer_f = get_optimizer(list(model_f1.parameters()) + list(model_f2.parameters()), opt=args.opt, lr=args.lr,
                                momentum=args.momentum, weight_decay=args.weight_decay)
    return model_g, model_f1, model_f2, optimizer_g, optimizer_f


def dropin_step(model_g, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d, src_imgs, src_lbls, tgt_imgs,
                num_k, num_multiply_d_loss):
    """adapt_trainer.py:163-214, statement for statement"""
    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    outputs = model_g(src_imgs)
    loss = criterion(model_f1(outputs), src_lbls) + criterion(model_f2(outputs), src_lbls)
    loss.backward()
    c_loss = loss.detach()
    optimizer_g.step()
    optimizer_f.step()
    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    outputs = model_g(src_imgs)
    loss = criterion(model_f1(outputs), src_lbls) + criterion(model_f2(outputs), src_lbls)
    outputs = model_g(tgt_imgs)
    loss = loss - criterion_d(model_f1(outputs), model_f2(outputs))
    loss.backward()
    optimizer_f.step()
    for _ in range(num_k):
        optimizer_g.zero_grad()
        outputs = model_g(tgt_imgs)
        loss = criterion_d(model_f1(outputs), model_f2(outputs)) * num_multiply_d_loss
        loss.backward()
        optimizer_g.step()
    return c_loss, loss.detach() / num_k


def main(argv=None):
    parser = get_da_mcd_training_parser()
    args = parser.parse_args(argv)
    args = add_additional_params_to_args(args)
    check_src_tgt_ok(args.src_dataset, args.tgt_dataset)
    run = Run(args)

    resume_flg = bool(args.resume)
    start_epoch = 0
    if args.resume:
        print("=> loading checkpoint '{}'".format(args.resume))
        if not os.path.exists(args.resume):
            raise OSError("%s does not exist!" % args.resume)
        infn = os.path.split(args.resume)[1]
        cli = args
        checkpoint = load_checkpoint(args.resume)
        start_epoch = checkpoint["epoch"]
        args = checkpoint["args"]  # the pickled namespace replaces the CLI one (adapt_trainer.py:40-43)
        args.savename = infn.split("-")[0] if "savename" not in vars(args) else args.savename
        for k in ("synthetic", "synthetic_raw", "synthetic_len", "src_file_list", "tgt_file_list", "seed", "no_pretrained", "solver", "no_tflog"):
            if k not in vars(args):
                setattr(args, k, getattr(cli, k))
        model_g, model_f1, model_f2, optimizer_g, optimizer_f = build(args)
        model_g.load_state_dict(checkpoint["g_state_dict"])
        model_f1.load_state_dict(checkpoint["f1_state_dict"])
        if not args.uses_one_classifier:
            model_f2.load_state_dict(checkpoint["f2_state_dict"])
        for m in (model_g, model_f1, model_f2):
            m.to(run.device)
        optimizer_g.load_state_dict(checkpoint["optimizer_g"])
        optimizer_f.load_state_dict(checkpoint["optimizer_f"])
        print("=> loaded checkpoint '{}'".format(args.resume))
    else:
        model_g, model_f1, model_f2, optimizer_g, optimizer_f = build(args)
    if args.uses_one_classifier:
        print("f1 and f2 are same!")
        model_f2 = model_f1

    mode = "%s-%s2%s-%s_%sch" % (args.src_dataset, args.src_split, args.tgt_dataset, args.tgt_split, args.input_ch)
    if args.net in ["fcn", "psp"]:
        model_name = "%s-%s-%s-res%s" % (args.method, args.savename, args.net, args.res)
    else:
        model_name = "%s-%s-%s" % (args.method, args.savename, args.net)
    outdir = os.path.join(args.base_outdir, mode)
    pth_dir = os.path.join(outdir, "pth")
    if run.is_main:
        mkdir_if_not_exist(pth_dir)
    run.configure_logger(os.path.join(outdir, "tflog", model_name), args)
    json_fn = os.path.join(outdir, "param-%s%s.json" % (model_name, "_resume" if resume_flg else ""))
    run.save_params(args, json_fn)

    train_loader = make_loader(args, run, [(args.src_dataset, args.src_split), (args.tgt_dataset, args.tgt_split)])
    weight = get_class_weight_from_file(n_class=args.n_class, weight_filename=args.loss_weights_file, add_bg_loss=args.add_bg_loss)
    for m in (model_g, model_f1, model_f2):
        m.to(run.device)
    weight = weight.to(run.device)
    run.sync_replicas([model_g, model_f1, model_f2])

    criterion = CrossEntropyLoss2d(weight)
    criterion_d = get_prob_distance_criterion(args.d_loss, n_class=args.n_class)  # symkl needs the row length (the reference passes none and fails there)
    for m in (model_g, model_f1, model_f2):
        m.train()
    if args.no_dropout:
        print("NO DROPOUT")
        for m in (model_g, model_f1, model_f2):
            fix_dropout_when_training(m)
    if args.fix_bn:
        emphasize_str("BN layers are NOT trained!")
        for m in (model_g, model_f1, model_f2):
            fix_batchnorm_when_training(m)

    solver = None
    if args.solver == "fused" and args.d_loss == "diff":
        solver = MCDSolver(model_g, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d, num_k=args.num_k,
                           num_multiply_d_loss=args.num_multiply_d_loss)

    for epoch in range(start_epoch, args.epochs):
        d_loss_per_epoch = 0.0
        c_loss_per_epoch = 0.0
        it = enumerate(train_loader)
        for ind, (source, target) in (tqdm.tqdm(it) if run.is_main else it):
            src_imgs = run.images(source[0])
            src_lbls = run.labels(source[1])
            tgt_imgs = run.images(target[0])
            if solver is not None:
                c_loss, d_loss = solver.step(src_imgs, src_lbls, tgt_imgs)
            else:
                c_loss, d_loss = dropin_step(model_g, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d,
                                             src_imgs, src_lbls, tgt_imgs, args.num_k, args.num_multiply_d_loss)
            c_loss, d_loss = float(c_loss), float(d_loss)
            c_loss_per_epoch += c_loss
            d_loss_per_epoch += d_loss
            if ind % 100 == 0 and run.is_main:
                print("iter [%d] DLoss: %.6f CLoss: %.4f" % (ind, d_loss, c_loss))
            if ind > args.max_iter:
                break
        if run.is_main:
            print("Epoch [%d] DLoss: %.4f CLoss: %.4f" % (epoch, d_loss_per_epoch, c_loss_per_epoch))
        run.log_value("c_loss", c_loss_per_epoch, epoch)
        run.log_value("d_loss", d_loss_per_epoch, epoch)
        run.log_value("lr", args.lr, epoch)
        if args.adjust_lr:  # the reference passes weight_decay as the decay rate (adapt_trainer.py:228-230)
            args.lr = adjust_learning_rate(optimizer_g, args.lr, args.weight_decay, epoch, args.epochs)
            args.lr = adjust_learning_rate(optimizer_f, args.lr, args.weight_decay, epoch, args.epochs)

        checkpoint_fn = os.path.join(pth_dir, "%s-%s.pth.tar" % (model_name, epoch + 1))
        args.start_epoch = epoch + 1
        save_dic = {
            "epoch": epoch + 1,
            "args": args,
            "g_state_dict": model_g.state_dict(),
            "f1_state_dict": model_f1.state_dict(),
            "optimizer_g": optimizer_g.state_dict(),
            "optimizer_f": optimizer_f.state_dict(),
        }
        if not args.uses_one_classifier:
            save_dic["f2_state_dict"] = model_f2.state_dict()
        run.save(save_dic, checkpoint_fn)
    return 0


if __name__ == "__main__":
    main()
