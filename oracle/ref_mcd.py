"""Oracle restatement of the three-step MCD update (plain torch autograd, CPU).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The reference has no solver function; the update is inline in the trainers.  Each function
below follows one of those loops statement by statement:

* ``mcd_step``        adapt_trainer.py:155-220        (G, F1, F2; early fusion)
* ``mfnet_mcd_step``  adapt_mfnet_trainer.py:174-244  (G_rgb, G_hha, F1, F2; no d-loss multiplier,
                                                       extra optimizer_f.zero_grad() after step B)
* ``source_step``     source_trainer.py:115-141       (supervised CE only)

Every forward runs in train mode, so BN running statistics move 7 times per ``mcd_step``
(src, src, tgt, then tgt x num_k) exactly as in the reference.
"""


def mcd_step(model_g, model_f1, model_f2, optimizer_g, optimizer_f, criterion, criterion_d,
             src_imgs, src_lbls, tgt_imgs, num_k=4, num_multiply_d_loss=1):
    # step A: G and F on source
    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    feat = model_g(src_imgs)
    loss = criterion(model_f1(feat), src_lbls) + criterion(model_f2(feat), src_lbls)
    loss.backward()
    c_loss = float(loss.detach())
    optimizer_g.step()
    optimizer_f.step()

    # step B: F only, keep source accuracy and maximise the discrepancy on target
    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    feat = model_g(src_imgs)
    loss = criterion(model_f1(feat), src_lbls) + criterion(model_f2(feat), src_lbls)
    feat = model_g(tgt_imgs)
    loss = loss - criterion_d(model_f1(feat), model_f2(feat))
    loss.backward()
    optimizer_f.step()

    # step C: G only, minimise the discrepancy on target, num_k times
    for _ in range(num_k):
        optimizer_g.zero_grad()
        feat = model_g(tgt_imgs)
        loss = criterion_d(model_f1(feat), model_f2(feat)) * num_multiply_d_loss
        loss.backward()
        optimizer_g.step()
    d_loss = float(loss.detach()) / num_k  # only the last inner loss is logged (adapt_trainer.py:214)
    return c_loss, d_loss


def mfnet_mcd_step(model_g_3ch, model_g_1ch, model_f1, model_f2, optimizer_g, optimizer_f, criterion,
                   criterion_d, src_imgs, src_lbls, tgt_imgs, num_k=4):
    def heads(x):
        a = model_g_3ch(x[:, :3])
        b = model_g_1ch(x[:, 3:])
        return model_f1(a, b), model_f2(a, b)

    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    o1, o2 = heads(src_imgs)
    loss = criterion(o1, src_lbls) + criterion(o2, src_lbls)
    loss.backward()
    c_loss = float(loss.detach())
    optimizer_g.step()
    optimizer_f.step()

    optimizer_g.zero_grad()
    optimizer_f.zero_grad()
    o1, o2 = heads(src_imgs)
    loss = criterion(o1, src_lbls) + criterion(o2, src_lbls)
    o1, o2 = heads(tgt_imgs)
    loss = loss - criterion_d(o1, o2)
    loss.backward()
    optimizer_f.step()
    optimizer_f.zero_grad()

    for _ in range(num_k):
        optimizer_g.zero_grad()
        o1, o2 = heads(tgt_imgs)
        loss = criterion_d(o1, o2)
        loss.backward()
        optimizer_g.step()
    d_loss = float(loss.detach()) / num_k
    return c_loss, d_loss


def source_step(model, optimizer, criterion, imgs, lbls):
    optimizer.zero_grad()
    loss = criterion(model(imgs), lbls)
    loss.backward()
    optimizer.step()
    return float(loss.detach())
