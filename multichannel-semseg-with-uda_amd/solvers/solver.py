"""The three-step Maximum-Classifier-Discrepancy update as a solver object.

The reference has no working solver (``solvers/solver.py`` there is dead LSGAN code); the update is
written inline in each trainer.  ``MCDSolver.step`` follows ``adapt_trainer.py:155-220`` and
``MFNetMCDSolver.step`` follows ``adapt_mfnet_trainer.py:174-244``:

  A  G, F1, F2 <- min  CE(F1(G(xs)), ys) + CE(F2(G(xs)), ys)
  B  F1, F2    <- min  CE1 + CE2 (source)  -  Diff(F1(G(xt)), F2(G(xt)))
  C  G         <- min  Diff(target) * num_multiply_d_loss          (num_k times)

Differences from running the same statements through the drop-in modules -- none of them changes a
result:
  * each phase evaluates its loss terms and both logit gradients with ONE fused kernel launch
    (``mcdseg.ops.mcd_losses``) instead of separate CE / CE / Diff criteria;
  * step B never back-propagates through G: only ``optimizer_f.step()`` follows and the generator
    gradients are zeroed before any use (adapt_trainer.py:187-205), so G runs without saving
    activations there -- its BatchNorm running statistics still move as on the reference's 7 forwards per step;
  * step B's generator forward on the target batch IS step C's first one: the generator does not change in between
    (only ``optimizer_f.step()`` runs), so the reference computes the same features twice (adapt_trainer.py:196 and :209).
    Here that forward runs once, with its tape, and each of its BatchNorm layers applies the running-statistics update twice
    (``ops.bn_running_updates``) -- weights, losses and running statistics are bit for bit those of the literal schedule
    (``MCDSEG_REUSE_TARGET_FORWARD=0`` or ``solver.reuse_tgt = False`` runs that one: 7 generator forwards instead of 6);
    switched off by itself when a generator holds dropout or a BatchNorm that does not run through the fused groups;
  * step C does not form the (unused) classifier weight gradients;
  * when both classifiers are the bare x8 up-sampler (``DRNSegPixelClassifier`` ver1 -- the MCD configuration), the loss
    kernel forms their logits on the fly from the generator's score map (``mcdseg.ops.up8_mcd_losses``): the two
    full-resolution logit tensors are never written or read.  Logits, losses and gradients are those of the two-pass form
    (bit for bit, up to the order of the loss's block partial sums).
"""
import contextlib
import os

import torch

from mcdseg import ops

REUSE_TARGET_FORWARD = os.environ.get("MCDSEG_REUSE_TARGET_FORWARD", "1") != "0"
# MFNet: the two modality encoders share nothing and run on two streams -- each one's HBM-bound BatchNorm passes beside the other's
# convolutions, forward and (autograd runs a node's backward on its forward's stream) backward: BASELINE config 3 409.4 -> 398.2 ms per
# step, bit-identical results (tests/test_model_gpu.py::test_mfnet_encoders_on_two_streams_are_bitwise).  "0": one after the other.
MFNET_TWO_STREAMS = os.environ.get("MCDSEG_MFNET_TWO_STREAMS", "1") != "0"
_SECOND = {}


def _second_stream(device, cur):
    """a stream for the second encoder, distinct from ``cur`` (the fork of step B may have made the side stream current)"""
    key = (device.index, cur.cuda_stream)
    if key not in _SECOND:
        _SECOND[key] = torch.cuda.Stream(device=device)
    return _SECOND[key]


def _detached(t):
    """``t.detach()`` that keeps the pre-split companion the producer attached (same storage, same version counter); maps over
    tuples / lists of tensors"""
    if isinstance(t, (tuple, list)):
        return type(t)(_detached(v) for v in t)
    d = t.detach()
    for a in ("_mcd_cb", "_mcd_virtual"):
        if hasattr(t, a):
            setattr(d, a, getattr(t, a))
    return d


# generators whose forward is a pure function of (weights, batch) plus BatchNorm running-statistics updates that honour
# ``ops.bn_running_updates``: the classes of models/dilated_fcn.py built on the fused conv+BN groups.  Anything else -- a user's
# own generator, functional dropout / noise this module cannot see -- gets the reference's literal schedule.
_REPEATABLE_GENERATORS = {"DRNSegBase", "MultiTaskEncoder"}


def _forward_is_repeatable(modules):
    """two forwards of these generators on one batch give the same features and the same running statistics as one forward whose
    BatchNorm layers apply their update twice: a whitelisted generator class holding no active dropout and no BatchNorm outside
    the fused groups.  Decided once, when the solver is built."""
    from models.drn import BatchNorm2d
    for g in modules:
        inner = getattr(g, "module", g) if type(g).__name__ == "DataParallel" else g
        if type(inner).__name__ not in _REPEATABLE_GENERATORS or not type(inner).__module__.startswith("models."):
            return False
        for m in inner.modules():
            if isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.p > 0:
                return False
            if isinstance(m, torch.nn.modules.batchnorm._NormBase) and not isinstance(m, BatchNorm2d):
                return False
    return True


def _params(modules):
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


class _frozen:
    """temporarily mark parameters as not requiring grad (their gradient would be discarded anyway)"""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]

    def __enter__(self):
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p in self.params:
            p.requires_grad_(True)


class MCDSolver:
    def __init__(self, model_g, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d, num_k=4,
                 num_multiply_d_loss=1):
        self.g, self.f1, self.f2 = model_g, model_f1, model_f2
        self.opt_g, self.opt_f = optimizer_g, optimizer_f
        self.class_weight = getattr(criterion, "weight", None)
        self.ignore_index = getattr(criterion, "ignore_index", -100)
        # the gated MFNet fusions emit probabilities and are trained with ProbCrossEntropyLoss2d (adapt_mfnet_trainer.py:149):
        # that criterion runs as its own kernel per head; everything else takes the fused CE/CE kernel
        self.prob_criterion = criterion if type(criterion).__name__ == "ProbCrossEntropyLoss2d" else None
        if type(criterion_d).__name__ != "Diff2d":
            raise NotImplementedError("the fused solver implements d_loss='diff' (loss.py:93-100)")
        self.num_k = num_k
        self.mult = float(num_multiply_d_loss)
        # step B's target forward doubles as step C's first (see the module docstring) -- for generator classes known to be repeatable
        repeatable = _forward_is_repeatable(self._generators())
        self.reuse_tgt = REUSE_TARGET_FORWARD and repeatable
        # step B's two generator passes side by side on two streams (ops.ForwardFork) -- for the same generator classes only: what orders
        # the two passes' BatchNorm running-statistics updates is the per-layer event inside the fused groups; a generator with a plain
        # nn.BatchNorm2d (or any other state outside those groups) would update it from both streams at once
        self.fork_ok = repeatable
        self.fused_up = (self.prob_criterion is None and ops.FUSED_UP_LOSS
                         and all(type(f).__name__ == "DRNSegPixelClassifier" and getattr(f, "ver", None) == "ver1"
                                 and type(getattr(f, "up", None)).__name__ == "Up8" for f in (model_f1, model_f2)))

    def _loss_backward(self, feats, labels, ce_coef=0.0, diff_coef=0.0):
        """Heads + loss + backward down to ``feats`` (and into the classifiers' parameters); returns losses[4]."""
        if self.fused_up and len(feats) == 1:
            s = feats[0]
            w1, w2 = self.f1.up.weight, self.f2.up.weight
            losses, g1, g2 = ops.up8_mcd_losses(s, w1, s, w2, labels, self.class_weight, self.ignore_index, ce_coef=ce_coef,
                                                diff_coef=diff_coef)
            sd = s.detach()
            ds = None
            for w, g in ((w1, g1), (w2, g2)):
                dx, dw = ops.up8_backward(g, sd, w.detach(), s.requires_grad, w.requires_grad)
                if dw is not None:
                    w.grad = dw if w.grad is None else w.grad.add_(dw)
                if dx is not None:
                    ds = dx if ds is None else ds.add_(dx)
            if ds is not None:
                torch.autograd.backward([s], [ds])
            return losses
        o1, o2 = self._heads(feats)
        if labels is not None:
            losses, g1, g2 = self._ce(o1, o2, labels)
        else:
            losses, g1, g2 = ops.mcd_losses(o1, o2, None, None, diff_coef=diff_coef)
        torch.autograd.backward([o1, o2], [g1, g2])
        return losses

    # -- hooks the MFNet variant overrides
    def _features(self, x):
        return (self.g(x),)

    def _heads(self, feats):
        return self.f1(*feats), self.f2(*feats)

    def _generators(self):
        return [self.g]

    def _ce(self, o1, o2, labels, want_grad=True):
        if self.prob_criterion is not None:
            vals, grads = [], []
            for o in (o1, o2):
                od = o.detach().requires_grad_(want_grad)
                val = self.prob_criterion(od, labels)
                if want_grad:
                    val.backward()
                vals.append(val.detach())
                grads.append(od.grad)
            return torch.stack(vals + [torch.zeros_like(vals[0])]), grads[0], grads[1]
        return ops.mcd_losses(o1, o2, labels, self.class_weight, self.ignore_index, ce_coef=1.0, want_grad=want_grad)

    def step(self, src_imgs, src_lbls, tgt_imgs):
        # ---- A: generator and classifiers on source
        self.opt_g.zero_grad()
        self.opt_f.zero_grad()
        losses = self._loss_backward(self._features(src_imgs), src_lbls, ce_coef=1.0)
        c_loss = losses[0] + losses[1]
        self.opt_g.step()
        self.opt_f.step()

        # ---- B: classifiers only (generator forward without a tape)
        self.opt_g.zero_grad()
        self.opt_f.zero_grad()
        # the two generator passes of this step are independent (same weights): they run side by side on two streams, every BatchNorm's
        # running statistics still updated source-first (ops.ForwardFork); the losses follow in the reference's order
        fork = ops.forward_fork(src_imgs.device) if self.fork_ok else None
        with (fork.lead() if fork is not None else torch.no_grad()):
            feats_src = self._features(src_imgs)
        if fork is None:
            self._loss_backward(feats_src, src_lbls, ce_coef=1.0)
        taped = None
        with (fork.follow() if fork is not None else contextlib.nullcontext()):
            if self.reuse_tgt and self.num_k > 0:
                with ops.bn_running_updates(2):  # this forward is also the first one of step C
                    taped = self._features(tgt_imgs)
                feats = tuple(_detached(f) for f in taped)
            else:
                with torch.no_grad():
                    feats = self._features(tgt_imgs)
        if fork is not None:
            fork.join(feats_src)
            self._loss_backward(feats_src, src_lbls, ce_coef=1.0)
        del feats_src
        self._loss_backward(feats, None, diff_coef=-1.0)
        del feats
        self.opt_f.step()
        self._after_b()

        # ---- C: generator only, num_k times
        d_last = None
        with _frozen(_params([self.f1, self.f2])):
            for k in range(self.num_k):
                self.opt_g.zero_grad()
                feats = taped if (k == 0 and taped is not None) else self._features(tgt_imgs)
                taped = None
                losses = self._loss_backward(feats, None, diff_coef=self.mult)
                del feats
                d_last = losses[2] * self.mult
                self.opt_g.step()
        d_loss = d_last / self.num_k  # only the last inner loss is logged (adapt_trainer.py:214)
        return c_loss, d_loss

    def _after_b(self):
        pass


class MFNetMCDSolver(MCDSolver):
    """Two modality encoders (RGB: channels 0-2, HHA/depth: the rest), classifiers take both feature maps."""

    def __init__(self, model_g_3ch, model_g_1ch, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d,
                 num_k=4):
        self.g_3ch, self.g_1ch = model_g_3ch, model_g_1ch  # (before the base constructor: it asks _generators())
        super().__init__(model_g_3ch, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d, num_k=num_k,
                         num_multiply_d_loss=1)  # adapt_mfnet_trainer.py:226-235 applies no multiplier

    def _features(self, x):
        timing = ops.LAUNCH_TIMER is not None and ops.LAUNCH_TIMER.wants("conv_wgrad")
        if not (MFNET_TWO_STREAMS and x.is_cuda) or timing:  # (a step whose launches carry HIP-event pairs runs every kernel alone)
            return self.g_3ch(x[:, :3, :, :]), self.g_1ch(x[:, 3:, :, :])
        # the two encoders share nothing: the HHA one runs on a stream of its own (forward here; autograd runs a node's backward on its
        # forward's stream), so that each encoder's HBM-bound BatchNorm passes overlap the other's convolutions
        cur = torch.cuda.current_stream(x.device)
        other = _second_stream(x.device, cur)
        other.wait_stream(cur)
        a = self.g_3ch(x[:, :3, :, :])
        with torch.cuda.stream(other):
            b = self.g_1ch(x[:, 3:, :, :])
        cur.wait_stream(other)
        b.record_stream(cur)
        for c in (getattr(b, "_mcd_cb", None) or ())[:2]:
            if torch.is_tensor(c):
                c.record_stream(cur)
        return a, b

    def _generators(self):
        return [self.g_3ch, self.g_1ch]

    def _after_b(self):
        self.opt_f.zero_grad()  # adapt_mfnet_trainer.py:222


class MultiTaskMCDSolver:
    """Three-step update of the multitask variant (segmentation + HHA regression decoders on an RGB encoder),
    following ``adapt_multitask_trainer.py:166-239`` through the decoder's own loss methods.

    Elisions that change no result: step B only steps the decoder optimizer, so the encoder runs there without a
    tape; the ``semseg_forward(src_fet)`` whose result the reference throws away (``:208``) still runs -- it moves
    the BatchNorm running statistics of both segmentation decoders -- but builds no graph; step B's encoder forward on the
    target batch is step C's first one (see the module docstring; ``reuse_tgt``)."""

    def __init__(self, model_enc, model_dec, optimizer_enc, optimizer_dec, num_k=4, num_multiply_d_loss=1):
        self.enc, self.dec = model_enc, model_dec
        self.opt_enc, self.opt_dec = optimizer_enc, optimizer_dec
        self.num_k, self.mult = num_k, num_multiply_d_loss
        repeatable = _forward_is_repeatable([model_enc])
        self.reuse_tgt = REUSE_TARGET_FORWARD and repeatable
        self.fork_ok = repeatable  # (as in MCDSolver: the fork's ordering exists inside the fused groups only)

    def step(self, src_imgs, src_gt_semseg, tgt_imgs):
        enc, dec = self.enc, self.dec
        src_rgbs, src_depths = src_imgs[:, :3, :, :], src_imgs[:, 3:, :, :].contiguous()
        tgt_rgbs, tgt_depths = tgt_imgs[:, :3, :, :], tgt_imgs[:, 3:, :, :].contiguous()

        self.opt_enc.zero_grad()
        self.opt_dec.zero_grad()
        src_fet = enc(src_rgbs)
        tgt_fet = enc(tgt_rgbs)
        src_semseg_loss, src_depth_loss = dec.get_loss(src_fet, src_gt_semseg, src_depths, separately_returning=True)
        tgt_depth_loss = dec.get_depth_loss(tgt_fet, tgt_depths)
        loss = src_semseg_loss + src_depth_loss + tgt_depth_loss
        loss.backward()
        c_loss = loss.detach()
        self.opt_enc.step()
        self.opt_dec.step()

        self.opt_enc.zero_grad()
        self.opt_dec.zero_grad()
        # the encoder's two passes of this step side by side (ops.ForwardFork, as in MCDSolver.step); the decoders follow in the
        # reference's order
        fork = ops.forward_fork(src_rgbs.device) if self.fork_ok else None
        with (fork.lead() if fork is not None else torch.no_grad()):
            src_fet = enc(src_rgbs)
        taped = None
        with (fork.follow() if fork is not None else contextlib.nullcontext()):
            if fork is None:
                with torch.no_grad():
                    dec.semseg_forward(src_fet)
                src_semseg_loss, src_depth_loss = dec.get_loss(src_fet, src_gt_semseg, src_depths, separately_returning=True)
            if self.reuse_tgt and self.num_k > 0:
                with ops.bn_running_updates(2):  # this forward is also the first one of step C
                    taped = enc(tgt_rgbs)
                tgt_fet = _detached(taped)
            else:
                with torch.no_grad():
                    tgt_fet = enc(tgt_rgbs)
        if fork is not None:
            fork.join([src_fet])
            with torch.no_grad():
                dec.semseg_forward(src_fet)
            src_semseg_loss, src_depth_loss = dec.get_loss(src_fet, src_gt_semseg, src_depths, separately_returning=True)
        tgt_depth_loss = dec.get_depth_loss(tgt_fet, tgt_depths)
        tgt_discrepancy = dec.get_cls_descrepancy(tgt_fet)
        loss = src_semseg_loss + src_depth_loss + tgt_depth_loss - tgt_discrepancy
        loss.backward()
        self.opt_dec.step()
        parts = (src_semseg_loss.detach(), src_depth_loss.detach(), tgt_depth_loss.detach())

        for k in range(self.num_k):
            self.opt_enc.zero_grad()
            tgt_fet = taped if (k == 0 and taped is not None) else enc(tgt_rgbs)
            taped = None
            loss = dec.get_cls_descrepancy(tgt_fet) * self.mult
            loss.backward()
            self.opt_enc.step()
        return c_loss, loss.detach() / self.num_k, parts
