"""BertAdam and the LR schedules of the reference, with the update running as ONE fused multi-tensor
HIP kernel sequence over the flat parameter arena (visualbert_amd/csrc/optim.hip).

API mirrors visualbert/pytorch_pretrained_bert/optimization.py:
  _LRSchedule / ConstantLR / WarmupCosineSchedule / WarmupConstantSchedule / WarmupLinearSchedule (:37-173),
  BertAdam(params, lr, warmup, t_total, schedule, b1, b2, e, weight_decay, max_grad_norm) (:185-304).
The device kernel evaluates the schedules the reference's training path uses -- 'warmup_linear' (the default,
models/model_wrapper.py:136-139) and 'none' -- from its per-tensor step counters; for every other schedule
(cosine, constant-after-warmup, the restart variants, any _LRSchedule subclass) the multiplier is evaluated on the host
from a mirror of the (common) step count and handed to the same kernel as the learning rate of that step.
"""
import math

import torch
from torch.optim import Optimizer

from . import _lib, ops


class _LRSchedule(object):
    warn_t_total = False

    def __init__(self, warmup=0.002, t_total=-1, **kw):
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        warmup = max(warmup, 0.)
        self.warmup, self.t_total = float(warmup), float(t_total)

    def get_lr(self, step, nowarn=False):
        if self.t_total < 0:
            return 1.
        return self.get_lr_(float(step) / self.t_total)

    def get_lr_(self, progress):
        return 1.


class ConstantLR(_LRSchedule):
    def get_lr_(self, progress):
        return 1.


class WarmupCosineSchedule(_LRSchedule):
    warn_t_total = True

    def __init__(self, warmup=0.002, t_total=-1, cycles=.5, **kw):
        super(WarmupCosineSchedule, self).__init__(warmup=warmup, t_total=t_total, **kw)
        self.cycles = cycles

    def get_lr_(self, progress):
        if progress < self.warmup:
            return progress / self.warmup
        progress = (progress - self.warmup) / (1 - self.warmup)
        return 0.5 * (1. + math.cos(math.pi * self.cycles * 2 * progress))


class WarmupCosineWithHardRestartsSchedule(WarmupCosineSchedule):
    """optimization.py:113-129: `cycles` cosine decays with hard restarts after the warm-up."""

    def __init__(self, warmup=0.002, t_total=-1, cycles=1., **kw):
        super(WarmupCosineWithHardRestartsSchedule, self).__init__(warmup=warmup, t_total=t_total, cycles=cycles, **kw)
        assert cycles >= 1.

    def get_lr_(self, progress):
        if progress < self.warmup:
            return progress / self.warmup
        progress = (progress - self.warmup) / (1 - self.warmup)
        return 0.5 * (1. + math.cos(math.pi * ((self.cycles * progress) % 1)))


class WarmupCosineWithWarmupRestartsSchedule(WarmupCosineWithHardRestartsSchedule):
    """optimization.py:132-150: training is cut into `cycles` equal parts, each with its own warm-up and cosine decay."""

    def __init__(self, warmup=0.002, t_total=-1, cycles=1., **kw):
        assert warmup * cycles < 1.
        warmup = warmup * cycles if warmup >= 0 else warmup
        super(WarmupCosineWithWarmupRestartsSchedule, self).__init__(warmup=warmup, t_total=t_total, cycles=cycles, **kw)

    def get_lr_(self, progress):
        progress = progress * self.cycles % 1.
        if progress < self.warmup:
            return progress / self.warmup
        progress = (progress - self.warmup) / (1 - self.warmup)
        return 0.5 * (1. + math.cos(math.pi * progress))


class WarmupConstantSchedule(_LRSchedule):
    def get_lr_(self, progress):
        if progress < self.warmup:
            return progress / self.warmup
        return 1.


class WarmupLinearSchedule(_LRSchedule):
    warn_t_total = True

    def get_lr_(self, progress):
        if progress < self.warmup:
            return progress / self.warmup
        return max((progress - 1.) / (self.warmup - 1.), 0.)


SCHEDULES = {None: ConstantLR, "none": ConstantLR, "warmup_cosine": WarmupCosineSchedule,
             "warmup_constant": WarmupConstantSchedule, "warmup_linear": WarmupLinearSchedule}      # optimization.py:176-182
_SCHEDULE_CLASSES = {c.__name__: c for c in (ConstantLR, WarmupCosineSchedule, WarmupCosineWithHardRestartsSchedule,
                                             WarmupCosineWithWarmupRestartsSchedule, WarmupConstantSchedule,
                                             WarmupLinearSchedule)}


class BertAdam(Optimizer):
    """BERT's Adam: per-tensor gradient clipping, no bias correction, decoupled weight decay, eps outside
    the sqrt, LR schedule evaluated from a per-tensor step counter (optimization.py:239-304).

    All parameters must live in one visualbert_amd ParameterArena (every model built by
    visualbert_amd.modeling does).  State (exp_avg = 'next_m', exp_avg_sq = 'next_v', step counters)
    is held as flat device buffers.

    `if p.grad is None: continue` (optimization.py:254-255) follows the zero_grad semantics of the reference's era
    (torch < 2: gradients are zeroed in place, so a parameter is skipped only until its first backward): a tensor that has
    ever received a gradient keeps stepping (weight decay, moment decay, counter) in steps that do not reach it.  Under
    torch >= 2's zero_grad(set_to_none=True) the reference itself would skip it again; `zero_grad()` here is one memset of
    the gradient arena either way.  load_state_dict keeps the moments / counters of parameters the checkpoint omits."""

    def __init__(self, params, lr=None, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0, **kwargs):
        if lr is None or lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not isinstance(schedule, _LRSchedule) and schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        if not isinstance(schedule, _LRSchedule):
            schedule = SCHEDULES[schedule](warmup=warmup, t_total=t_total)
        defaults = dict(lr=lr, schedule=schedule, b1=b1, b2=b2, e=e, weight_decay=weight_decay,
                        max_grad_norm=max_grad_norm)
        super(BertAdam, self).__init__(params, defaults)
        self._fused = None

    # -- fused state -------------------------------------------------------------------------------
    def _build(self):
        arena = None
        member = {}
        g0 = self.param_groups[0]
        for gi, group in enumerate(self.param_groups):
            for k in ("lr", "b1", "b2", "e", "max_grad_norm"):
                if group[k] != g0[k]:
                    raise NotImplementedError("fused BertAdam: groups may differ in weight_decay only (as in "
                                              "models/model_wrapper.py:108-112); %s differs" % k)
            if type(group["schedule"]) is not type(g0["schedule"]) or \
                    group["schedule"].warmup != g0["schedule"].warmup or group["schedule"].t_total != g0["schedule"].t_total:
                raise NotImplementedError("fused BertAdam: one schedule for all groups")
            for p in group["params"]:
                a = getattr(p, "_vb_arena", None)
                if a is None:
                    raise RuntimeError("visualbert_amd.BertAdam: parameter is not arena-managed (build the model "
                                       "with visualbert_amd.modeling; no per-tensor fallback exists)")
                if arena is None:
                    arena = a
                elif a is not arena:
                    raise RuntimeError("visualbert_amd.BertAdam: parameters from different arenas")
                member[id(p)] = group["weight_decay"]
        wds = sorted(set(v for v in member.values() if v > 0.0))
        if len(wds) > 1:
            raise NotImplementedError("fused BertAdam: a single non-zero weight_decay value")
        sch = g0["schedule"]
        if type(sch) is WarmupLinearSchedule:
            code = 1                                     # evaluated on the device from the per-tensor step counters
        elif type(sch) is ConstantLR or sch.t_total < 0:
            code = 0
        else:
            code = -1                                    # evaluated on the host, see step()
        opt_flags = [id(p) in member for p in arena.params]
        dec_flags = [member.get(id(p), 0.0) > 0.0 for p in arena.params]
        tt, ct, nt, nc = arena.tables(opt_flags, dec_flags)
        dev = arena.device
        self._fused = dict(arena=arena, tt=tt, ct=ct, nt=nt, nc=nc, code=code, wd=wds[0] if wds else 0.0,
                           m=torch.zeros_like(arena.data), v=torch.zeros_like(arena.data),
                           norm2=torch.zeros(nt + nc, dtype=torch.float32, device=dev),
                           steps=torch.zeros(nt, dtype=torch.int32, device=dev), host_step=0,
                           opt_flags=opt_flags, dec_flags=dec_flags)
        return self._fused

    def _touched_flags(self, f):
        """The reference's step() skips a parameter whose .grad is None -- no moment update, no weight decay
        (optimization.py:254-255): a head's unused tensors (cls.* under `flickr`, seq_relationship under `vqa_advanced`), the
        visual tables on a text-only batch, ...  Here every gradient is a view into the flat arena and always exists, so
        the backward pass records which parameters it wrote (ParameterArena.touched) and hands the optimizer KERNEL one
        flag per tensor; the kernel skips a tensor iff its flag is 0 and its gradient norm is 0.  Nothing is cached on the
        host between steps: under data parallelism the flags are all-reduced with the gradients
        (DataParallelGradSync.finish_step -> arena.touched_synced), so every rank takes the same decision every step."""
        a = f["arena"]
        synced = getattr(a, "touched_synced", None)
        if synced is not None:                                  # this step's flags, already combined over the ranks
            a.touched_synced = None
            return synced
        return a.touched_flags()

    def fused(self):
        f = self._fused
        if f is None or f["arena"] is not getattr(self.param_groups[0]["params"][0], "_vb_arena", None):
            old = f
            f = self._build()
            if old is not None and int(old["steps"].max()) > 0:
                # the model was moved / re-laid-out after training had started (TrainVisualBERTObjective._apply rebuilds the
                # arena): the moments and step counters follow their parameters by NAME instead of silently restarting
                a0, a1 = old["arena"], f["arena"]
                where = {n: (o, p.numel(), i) for i, (n, p, o) in enumerate(zip(a0.names, a0.params, a0.offsets))}
                steps0, steps1 = old["steps"].tolist(), f["steps"].tolist()
                for i, (n, p, o) in enumerate(zip(a1.names, a1.params, a1.offsets)):
                    if n not in where or where[n][1] != p.numel():
                        raise RuntimeError("visualbert_amd.BertAdam: parameter %r changed under a running optimizer" % n)
                    o0, cnt, i0 = where[n]
                    f["m"][o:o + cnt].copy_(old["m"][o0:o0 + cnt])
                    f["v"][o:o + cnt].copy_(old["v"][o0:o0 + cnt])
                    steps1[i] = steps0[i0]
                f["steps"].copy_(torch.tensor(steps1, dtype=torch.int32))
                f["host_step"] = old["host_step"]
        return f

    def zero_grad(self, set_to_none=False):
        """one memset of the flat gradient arena; .grad stays a view into it (set_to_none is ignored on
        purpose: the kernels accumulate straight into these views)."""
        self.fused()["arena"].zero_grad()

    def get_lr(self):
        f = self.fused()
        g = self.param_groups[0]
        steps = f["steps"].tolist()
        flags = f["tt"].view(-1, 4)[:, 3].tolist()
        return [g["lr"] * g["schedule"].get_lr(s) for s, fl in zip(steps, flags) if fl & 1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        f = self.fused()
        a = f["arena"]
        g = self.param_groups[0]
        sch = g["schedule"]
        L = _lib.lib()
        touched = self._touched_flags(f)
        lr, code = float(g["lr"]), f["code"]
        if code < 0:
            # every optimised tensor takes every step, so one host counter mirrors the device's per-tensor counters
            # (optimization.py:283-284 evaluates the schedule at state['step'] BEFORE incrementing it)
            lr, code = lr * float(sch.get_lr(f["host_step"])), 0
        rc = L.vb_bert_adam_step(_lib.ptr(a.data), _lib.ptr(a.grad), _lib.ptr(f["m"]), _lib.ptr(f["v"]),
                                 _lib.ptr(a.shadow), _lib.ptr(f["ct"]), f["nc"], _lib.ptr(f["tt"]), f["nt"],
                                 _lib.ptr(touched), _lib.ptr(f["norm2"]), _lib.ptr(f["steps"]), lr, float(g["b1"]),
                                 float(g["b2"]), float(g["e"]), float(f["wd"]), float(g["max_grad_norm"]),
                                 float(sch.warmup), float(sch.t_total), code, _lib.stream_ptr())
        _lib.check(rc, "vb_bert_adam_step")
        f["host_step"] += 1
        # the kernel refreshed the bf16 shadows of every optimised 2-D parameter
        shadowed = f.get("shadowed")
        if shadowed is None:                         # (the fused state -- and this list with it -- is rebuilt by load_state_dict)
            member = set()
            for group in self.param_groups:
                for p in group["params"]:
                    member.add(id(p))
            shadowed = f["shadowed"] = [p for p in a.params if p.dim() == 2 and id(p) in member]
        for p in shadowed:
            p._vb_shadow_ver = p._version
        a.refresh_transposed()                       # W^T copies for the next backward (one launch)
        ops.bump_x3_epoch()                          # split (bf16x3) images of the weights are re-made at their next use
        return loss

    # -- checkpoint compatibility: per-parameter {'step', 'next_m', 'next_v'} like the reference ------
    def state_dict(self):
        f = self.fused()
        a = f["arena"]
        steps = f["steps"].tolist()
        state = {}
        idx = {id(p): i for i, p in enumerate(a.params)}
        k = 0
        groups = []
        for group in self.param_groups:
            ids = []
            for p in group["params"]:
                i = idx[id(p)]
                o, n = a.offsets[i], p.numel()
                state[k] = dict(step=steps[i], next_m=f["m"][o:o + n].view(p.shape).clone(),
                                next_v=f["v"][o:o + n].view(p.shape).clone())
                ids.append(k)
                k += 1
            gd = {kk: vv for kk, vv in group.items() if kk not in ("params", "schedule")}
            gd["params"] = ids
            sch = group["schedule"]
            # a plain dict, not the object and not a bare class name: torch.save-able by either side and enough to
            # rebuild the schedule on load
            gd["schedule"] = {"class": type(sch).__name__, "warmup": float(getattr(sch, "warmup", -1)),
                              "t_total": float(getattr(sch, "t_total", -1)), "cycles": getattr(sch, "cycles", None)}
            groups.append(gd)
        return dict(state=state, param_groups=groups)

    def load_state_dict(self, sd):
        # hyper-parameters and the schedule travel with the checkpoint (saved as {"class", "warmup", "t_total", "cycles"});
        # a checkpoint written by the reference carries the schedule OBJECT of its own module -- rebuilt here by class name
        for group, saved in zip(self.param_groups, sd.get("param_groups", [])):
            for k, v in saved.items():
                if k in ("params", "schedule"):
                    continue
                group[k] = v
            sch = saved.get("schedule")
            if sch is not None and not isinstance(sch, dict):
                sch = {"class": type(sch).__name__, "warmup": float(getattr(sch, "warmup", -1)),
                       "t_total": float(getattr(sch, "t_total", -1)), "cycles": getattr(sch, "cycles", None)}
            if isinstance(sch, dict) and sch.get("class") in _SCHEDULE_CLASSES:
                kw = dict(warmup=sch["warmup"], t_total=sch["t_total"])
                if sch.get("cycles") is not None:
                    kw["cycles"] = sch["cycles"]
                group["schedule"] = _SCHEDULE_CLASSES[sch["class"]](**kw)
        # the fused state caches weight decay, the decay / optimise flags and the device tables: rebuild it from the
        # hyper-parameters just restored (the moments and counters are overwritten from the checkpoint right below)
        old = self._fused
        self._fused = None
        f = self.fused()
        if old is not None and old["m"].shape == f["m"].shape:
            # parameters the checkpoint does not mention keep the moments and counters they had (the reference's
            # Optimizer.load_state_dict replaces `state` wholesale only for the ids it lists; a partial checkpoint must not
            # silently zero the rest)
            f["m"].copy_(old["m"])
            f["v"].copy_(old["v"])
            f["steps"].copy_(old["steps"])
        sch0 = self.param_groups[0]["schedule"]
        f["code"] = 1 if type(sch0) is WarmupLinearSchedule else (0 if (type(sch0) is ConstantLR or sch0.t_total < 0) else -1)
        a = f["arena"]
        idx = {id(p): i for i, p in enumerate(a.params)}
        steps = f["steps"].tolist()
        k = 0
        for group in self.param_groups:
            for p in group["params"]:
                st = sd["state"].get(k)
                if st is not None:
                    i = idx[id(p)]
                    o, n = a.offsets[i], p.numel()
                    f["m"][o:o + n].view(p.shape).copy_(st["next_m"])
                    f["v"][o:o + n].view(p.shape).copy_(st["next_v"])
                    steps[i] = int(st["step"])
                k += 1
        f["steps"].copy_(torch.tensor(steps, dtype=torch.int32))
        flags = f["tt"].view(-1, 4)[:, 3].tolist()
        taken = sorted(set(st for st, fl in zip(steps, flags) if fl & 1))
        if f["code"] < 0 and len(taken) > 1:
            raise NotImplementedError("fused BertAdam: %s is evaluated from ONE step count, the checkpoint holds %s"
                                      % (type(self.param_groups[0]["schedule"]).__name__, taken))
        f["host_step"] = taken[-1] if taken else 0
