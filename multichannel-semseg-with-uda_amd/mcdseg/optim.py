"""Flat-buffer SGD (momentum + weight decay) driven by one HIP kernel per step.

Drop-in for ``torch.optim.SGD`` as the reference configures it (models/model_util.py:289-292:
dampening 0, no Nesterov): same constructor arguments, ``param_groups``, ``zero_grad``, ``step`` and a
``state_dict`` in torch's own layout (``state[i]['momentum_buffer']`` + ``param_groups``), so
checkpoints interchange with the reference (adapt_trainer.py:232-245, 29-59).

Parameters are re-homed, on first use, into one contiguous fp32 buffer (each ``p.data`` becomes a view
of it); momentum lives in a second flat buffer and gradients are gathered into a third, which is also
the single payload of the data-parallel all-reduce (``mcdseg.dist``).  Parameters whose ``grad`` is
None are skipped exactly as torch does (no weight decay, no momentum update).
"""
import os

import torch

from . import dist as mdist
from . import ops

_ALIGN = 64  # floats; keeps every view 256-byte aligned for the float4 kernels
# Data parallelism, optional: MCDSEG_DP_OVERLAP=1 all-reduces the flat gradient buffer in buckets of MCDSEG_DP_BUCKET_MB (default 25)
# while the backward pass is still running -- a gradient is copied into its flat view by a post-accumulate hook, and once every
# parameter of a bucket has arrived the bucket's slice is reduced asynchronously (RCCL's own stream).  OFF by default: the five
# 104 MB all-reduces of an MCD step are ~2 % of its time on xGMI; the switch exists so that a measured
# scaling curve can be acted on without new code.  Results are those of the single all-reduce (sums of the same values).
# The ORDER in which the buckets' collectives are started is the same on every rank whatever the ranks' timing: a bucket whose last
# gradient has arrived is only "ready", and collectives are started strictly in bucket order (``_drain``) -- WHEN a gradient arrives
# depends on rank-local state (a weight gradient left on the side stream arrives early through ``_early_grad``, one kept on the main
# stream for lack of memory only at the end of the pass), WHETHER a bucket becomes ready in a pass depends on the graph alone.
DP_OVERLAP = os.environ.get("MCDSEG_DP_OVERLAP", "0") == "1"
DP_BUCKET_MB = float(os.environ.get("MCDSEG_DP_BUCKET_MB", "25"))


class FlatSGD(torch.optim.Optimizer):
    # The update kernel is HIP-only.  Tests of the host-side logic (flat layout, run detection, the gloo
    # all-reduce of the data-parallel path) clear this flag and substitute a reference update for the kernel.
    _require_gpu = True

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("FlatSGD: negative hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=0, nesterov=False))
        self._flat = None

    # ------------------------------------------------------------------ flat storage
    def _all_params(self):
        seen, out = set(), []
        for group in self.param_groups:
            for p in group["params"]:
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    def _flatten(self):
        params = self._all_params()
        if not params:
            raise ValueError("FlatSGD: no parameters")
        dev = params[0].device
        if dev.type != "cuda" and self._require_gpu:
            raise RuntimeError("FlatSGD: parameters must be on the GPU before the first step (no CPU fallback)")
        offs, total = [], 0
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise RuntimeError("FlatSGD: all parameters must be fp32 on one device")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        fp = torch.zeros(total, dtype=torch.float32, device=dev)
        fv = torch.zeros(total, dtype=torch.float32, device=dev)
        fg = torch.zeros(total, dtype=torch.float32, device=dev)
        views = {}
        for p, o in zip(params, offs):
            n = p.numel()
            pv = fp[o:o + n].view(p.shape)
            pv.copy_(p.data)
            p.data = pv
            vv = fv[o:o + n].view(p.shape)
            st = self.state.get(p)
            if st and st.get("momentum_buffer") is not None:
                vv.copy_(st["momentum_buffer"])
                st["momentum_buffer"] = vv
            views[id(p)] = (o, n, vv, fg[o:o + n].view(p.shape))
        self._flat = dict(p=fp, v=fv, g=fg, params=params, offs=offs, views=views, total=total)
        ops.bump_weight_epoch()
        self._setup_overlap()

    # ------------------------------------------------------------------ bucketed all-reduce during backward (optional)
    def _setup_overlap(self):
        fl = self._flat
        for h in getattr(self, "_hooks", []):
            h.remove()
        for p in fl["params"]:
            if getattr(p, "_mcd_grad_sink", None) is not None:
                del p._mcd_grad_sink
        self._hooks, fl["buckets"], fl["bucket_of"] = [], [], {}
        if not (DP_OVERLAP and mdist.is_distributed()):
            return
        limit = int(DP_BUCKET_MB * (1 << 20) / 4)
        cur = None
        # buckets in REVERSE parameter order: the backward pass produces the last layers' gradients first
        for p, o in reversed(list(zip(fl["params"], fl["offs"]))):
            n = p.numel()
            if cur is None or cur["hi"] - o > limit:
                cur = dict(lo=o, hi=o + n, ids=set(), arrived=0, work=None, dirty=False, early=set(), early_seen={}, ready=False,
                           launched=False, unsunk=set(), events=[], index=len(fl["buckets"]))
                fl["buckets"].append(cur)
            cur["lo"] = o
            cur["ids"].add(id(p))
            fl["bucket_of"][id(p)] = cur
        fl["next"] = 0  # the first bucket whose collective has not been started in this pass (``_drain``)
        for p in fl["params"]:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._grad_arrived))
            p._mcd_grad_sink = self._early_grad  # (ops._conv_backward: a weight gradient left on the side stream arrives here first)

    def _early_grad(self, p, dw):
        """A weight gradient whose kernels have just been enqueued on the side stream (ops._conv_backward, MCDSEG_OVERLAP_WGRAD=2);
        the CURRENT stream is that side stream.  Copy it into the flat view there and, when it completes its bucket, mark the bucket
        ready -- its all-reduce then starts from there as soon as every earlier bucket's has (``_drain``): RCCL's stream waits for the
        side stream only, and the main stream's backward pass goes on.  The gradient reaches ``p.grad`` through its ``_LateGrad`` node
        at the end of the pass; ``_grad_arrived`` then finds the parameter in ``early`` and has nothing left to do.  Anything but
        "first gradient of this parameter since the last step / zero_grad, nothing accumulated yet" leaves the bucket to step()'s
        copy-and-reduce path.  ``dw`` None: ops reports a contribution to this parameter's gradient that does NOT come through here
        (a second use of the weight in one graph, a weight gradient kept on the main stream) -- what ``p.grad`` will hold is then
        more than an early copy, so an early copy of the same pass must not stand for it."""
        fl = self._flat
        b = fl["bucket_of"].get(id(p)) if fl is not None else None
        if b is None:
            return
        if dw is None:
            b["unsunk"].add(id(p))
            if id(p) in b["early"]:
                self._spoil(b)
            return
        if (b["ready"] or b["dirty"] or id(p) in b["early"] or id(p) in b["unsunk"] or p.grad is not None
                or b["arrived"] >= len(b["ids"])):
            self._spoil(b)
            return
        fl["views"][id(p)][3].copy_(dw)
        self._copied(b, dw)
        b["early"].add(id(p))
        b["arrived"] += 1
        if b["arrived"] == len(b["ids"]):
            b["ready"] = True
            self._drain()

    @staticmethod
    def _copied(b, t):
        if t.is_cuda:
            ev = torch.cuda.Event()
            ev.record()  # (on the current stream: the one the copy was enqueued on)
            b["events"].append(ev)

    def _spoil(self, b):
        """the bucket's slice no longer equals the gradients step() will find in ``p.grad``: wait for its collective if one is in
        flight (step()'s copies must not race it) and leave the bucket to step()'s copy-and-reduce path.  A bucket that was ready
        keeps its place in the launch order (``_drain`` starts its -- now useless -- collective all the same: another rank may have
        started it before the bucket was spoilt there)."""
        if b["work"] is not None:
            b["work"].wait()
            b["work"] = None
        b["dirty"] = True

    def _drain(self, flush=False):
        """start the all-reduce of every ready bucket whose predecessors have all been started -- strictly in bucket order, so that
        all ranks issue the same sequence of collectives (module comment).  ``flush`` (step()): also past buckets that did not
        become ready in this pass; which those are depends on the graph only."""
        fl = self._flat
        bs, i = fl["buckets"], fl["next"]
        while i < len(bs):
            b = bs[i]
            if not b["ready"]:
                if not flush:
                    break
            elif not b["launched"]:
                self._launch_reduce(b)
            i += 1
        fl["next"] = i

    def _launch_reduce(self, b):
        """every gradient of the bucket sits in its flat view: start the all-reduce of the slice.  When some of them were copied on the
        side stream (``_early_grad``) the exchange is started FROM the side stream, behind those copies and behind the current stream's
        own: RCCL's stream then waits for the side stream, and the stream running the backward pass waits for nobody."""
        fl = self._flat
        flat = fl["g"][b["lo"]:b["hi"]]
        b["launched"] = True
        if flat.is_cuda:
            # the copies into this slice ran on the main stream (``_grad_arrived``) and / or on the side stream (``_early_grad``), and a
            # bucket may be started from either (``_drain`` runs where the LAST bucket in line became ready): the launching stream
            # waits for the event behind every copy
            stream = ops._side_stream(flat.device) if b["early"] else torch.cuda.current_stream(flat.device)
            for ev in b["events"]:
                stream.wait_event(ev)
            with torch.cuda.stream(stream):
                b["work"] = mdist.all_reduce_sum_async(flat)
        else:
            b["work"] = mdist.all_reduce_sum_async(flat)
        if b["dirty"] and b["work"] is not None:  # (spoilt before its turn came: nobody will wait for it in step())
            b["work"].wait()
            b["work"] = None

    def _grad_arrived(self, p):
        fl = self._flat
        if fl is None or id(p) not in fl["bucket_of"] or p.grad is None:
            return
        b = fl["bucket_of"][id(p)]
        if (id(p) in b["early"] and not b["dirty"] and id(p) not in b["unsunk"] and b["early_seen"].get(id(p), 0) == 0):
            b["early_seen"][id(p)] = 1  # the gradient the side stream delivered (``_early_grad``) has now reached p.grad: nothing to do
            return
        if b["ready"] or b["dirty"] or b["arrived"] >= len(b["ids"]) or id(p) in b["early"]:
            # A SECOND backward pass before step() (MCDSolver's step B: two loss.backward() calls, then optimizer_f.step()): p.grad
            # now holds the accumulated local gradient, while the bucket's slice is being -- or has been -- summed over the ranks
            # from the first pass alone.  (Or: p.grad holds more than the early copy, ``unsunk``.)  Wait for the collective in flight
            # (the copy below must not race it), and leave the bucket to step()'s copy-and-reduce path, which reads the accumulated
            # p.grad of every parameter.
            self._spoil(b)
            return
        fl["views"][id(p)][3].copy_(p.grad)
        self._copied(b, p.grad)
        b["arrived"] += 1
        if b["arrived"] == len(b["ids"]):
            b["ready"] = True
            self._drain()

    def _finish_overlap(self, ps):
        """waits for the collectives the hooks started; True when every gradient of ``ps`` has been reduced that way (else the
        caller copies and reduces the whole run again -- correct, merely redundant for the buckets that were complete)"""
        fl = self._flat
        self._drain(flush=True)
        buckets = {id(b): b for b in (fl["bucket_of"].get(id(p)) for p in ps) if b is not None}
        for b in buckets.values():
            if b["work"] is not None:
                b["work"].wait()
        return all(fl["bucket_of"].get(id(p)) is not None and fl["bucket_of"][id(p)]["work"] is not None and
                   not fl["bucket_of"][id(p)]["dirty"] for p in ps)

    def _reset_overlap(self):
        for b in (self._flat or {}).get("buckets", []):
            if b["work"] is not None:  # (zero_grad() without step(): the reference's literal loop computes G's gradients in step B and
                b["work"].wait()       # never applies them -- the next pass's copies must not race a collective still in flight)
            b["arrived"], b["work"], b["dirty"], b["ready"], b["launched"] = 0, None, False, False, False
            b["early"].clear(), b["early_seen"].clear(), b["unsunk"].clear(), b["events"].clear()
        if self._flat is not None and "next" in self._flat:
            self._flat["next"] = 0

    def _ensure_flat(self):
        if self._flat is None:
            self._flatten()
            return
        # .cuda()/.to()/load_state_dict(assign) may have re-homed a parameter: rebuild if any view moved
        fl = self._flat
        for p in fl["params"]:
            o, n, _, _ = fl["views"][id(p)]
            if p.data_ptr() != fl["p"].data_ptr() + 4 * o:
                self._flat = None
                self._flatten()
                return

    # ------------------------------------------------------------------ torch.optim API
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_flat()
        fl = self._flat
        group_of = {}
        for group in self.param_groups:
            for p in group["params"]:
                group_of[id(p)] = group
        # maximal runs of consecutive parameters that have a gradient and share hyper-parameters
        runs, cur = [], None
        for p in fl["params"]:
            g = group_of[id(p)]
            if p.grad is None:
                cur = None
                continue
            key = (g["lr"], g["momentum"], g["weight_decay"])
            if cur is None or cur["key"] != key:
                cur = dict(key=key, params=[])
                runs.append(cur)
            cur["params"].append(p)
        world = mdist.world_size()
        for run in runs:
            ps = run["params"]
            o0 = fl["views"][id(ps[0])][0]
            o1, n1 = fl["views"][id(ps[-1])][0], fl["views"][id(ps[-1])][1]
            lo, hi = o0, o1 + n1
            gflat = fl["g"][lo:hi]
            reduced = bool(fl.get("buckets")) and self._finish_overlap(ps)  # the hooks copied and reduced these during backward
            if not reduced:
                gviews = [fl["views"][id(p)][3] for p in ps]
                grads = [p.grad if p.grad.dtype == torch.float32 else p.grad.float() for p in ps]
                torch._foreach_copy_(gviews, grads)
                if mdist.is_distributed():
                    mdist.all_reduce_sum_(gflat)
            lr, mu, wd = run["key"]
            ops.sgd_momentum_flat_(fl["p"][lo:hi], gflat, fl["v"][lo:hi], lr, mu, wd, 1.0 / world, params=ps)
            if mu != 0:
                for p in ps:
                    self.state[p]["momentum_buffer"] = fl["views"][id(p)][2]
        self._reset_overlap()
        return loss

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        if self._flat is not None:
            self._reset_overlap()

    def state_dict(self):
        sd = super().state_dict()
        for st in sd["state"].values():
            if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                st["momentum_buffer"] = st["momentum_buffer"].clone()  # detach from the flat storage
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._flat is not None:
            fl = self._flat
            fl["v"].zero_()
            for p in fl["params"]:
                st = self.state.get(p)
                if st and st.get("momentum_buffer") is not None:
                    vv = fl["views"][id(p)][2]
                    vv.copy_(st["momentum_buffer"])
                    st["momentum_buffer"] = vv

    def flat_buffers(self):
        """(params, grads, momentum) flat tensors -- for tests and the bench's byte accounting."""
        self._ensure_flat()
        return self._flat["p"], self._flat["g"], self._flat["v"]
