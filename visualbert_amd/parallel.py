"""Data-parallel gradient synchronisation: one process per GPU, RCCL over xGMI.

The reference's only parallelism is single-process nn.DataParallel (models/model_wrapper.py:146):
per step it broadcasts all weights GPU0 -> others, scatters the batch, gathers one loss per replica,
takes loss.mean() (model_wrapper.py:75) and reduce-adds gradients onto GPU 0.  Here every rank owns
a full replica and its own shard of the minibatch, computes its own mean loss, and the flat fp32
gradient arena is all-reduced with AVERAGING -- which reproduces the reference's mean of per-replica
means exactly -- in L+2 contiguous buckets launched while backward is still running:

    heads+pooler | layer L-1 | ... | layer 0 | embeddings (incl. the tied decoder weight, last)

Each bucket is handed to RCCL (torch.distributed backend "nccl" == RCCL on ROCm) from an autograd
hook the moment its layer's wgrad kernels have been enqueued; the collective runs on RCCL's own
stream behind an event, so it overlaps the remaining backward compute.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU): a ring all-reduce moves 2(N-1)/N x payload per GPU and is per-link
bound, so buckets are whole layers (28 MB fp32 at BERT-base) -- large enough to run at link speed,
small enough that the last one (embeddings, 94 MB + heads) is the only exposed tail.

Compute units for the collective.  The persistent GEMM kernels launch one 160-KB-LDS workgroup per CU, and such a
workgroup cannot share a CU with an RCCL workgroup: while the collective is resident on c CUs, c workgroups of a GEMM
launch wait for a free CU.  Reserving CUs up front (VB_COMM_CUS=c: GEMMs use CUs - c workgroups while buckets are in
flight, RCCL capped to c channels via NCCL_MAX_NCHANNELS) was measured on one GPU and is OFF by default: the N=768 GEMMs
have 246 output tiles, so ANY grid below 246 workgroups costs them a second round (30 -> 47 us, 78 -> 128 us at 224
workgroups) -- the same price the un-reserved launch pays only while the collective really is resident.
"""
import os

import torch
import torch.distributed as dist

from . import _lib


def comm_cus():
    return int(os.environ.get("VB_COMM_CUS", "0"))


def configure_rccl_env():
    """call before init_process_group: with VB_COMM_CUS=c keep the collective on at most c channels (= CUs)."""
    if comm_cus() > 0:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(comm_cus()))


class DataParallelGradSync(object):
    def __init__(self, objective, process_group=None, overlap=True):
        """objective: visualbert_amd.modeling.TrainVisualBERTObjective (owns the ParameterArena)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.obj = objective
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.overlap = overlap
        self.backend = dist.get_backend(process_group)
        self._works = []
        self._done = set()
        self._reserved = False
        from . import ops
        ops.set_replica(dist.get_rank(process_group))       # replicas must not repeat each other's dropout masks
        self._install()

    def _install(self):
        self.buckets = self.obj.bucket_ranges()              # [(name, lo, hi)] in completion order
        self._by_name = {n: (lo, hi) for n, lo, hi in self.buckets}
        for layer in self.obj.bert.encoder.layer:
            layer.grad_ready_hook = self._layer_ready if self.overlap else None

    def broadcast_parameters(self, src=0):
        """all ranks start from rank `src`'s weights (the reference replicates GPU0's weights every forward)."""
        dist.broadcast(self.obj.arena.data, src=src, group=self.pg)
        self.obj.arena.refresh_shadows() if self.obj.arena.data.is_cuda else None

    # -- per step ---------------------------------------------------------------------------------
    def begin_step(self):
        if self.obj.arena.grad.data_ptr() != getattr(self, "_grad_ptr", None):
            self._install()                                   # arena was rebuilt (e.g. .to(device))
            self._grad_ptr = self.obj.arena.grad.data_ptr()
        self._works = []
        self._done = set()

    def _reduce(self, name):
        if name in self._done:
            return
        self._done.add(name)
        lo, hi = self._by_name[name]
        if lo is None:
            return
        view = self.obj.arena.grad[lo:hi]
        self._reserve_cus(True)
        if self.backend == "nccl":
            w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        else:                                                 # gloo (CPU tests): SUM then scale
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append((w, view))

    def _reserve_cus(self, on):
        """leave comm_cus() CUs to RCCL while buckets are in flight (see the module docstring)."""
        if self.backend != "nccl" or self.world == 1 or comm_cus() <= 0 or on == self._reserved:
            return
        self._reserved = on
        wgs = 0
        if on:
            cus = torch.cuda.get_device_properties(self.obj.arena.grad.device).multi_processor_count
            wgs = max(8, (cus - comm_cus()) // 8 * 8)
        _lib.check(_lib.lib().vb_gemm_set_persistent_wgs(wgs), "vb_gemm_set_persistent_wgs")

    def _layer_ready(self, layer_index):
        # everything above this layer in the graph has finished enqueuing its backward
        self._reduce("heads")
        self._reduce("layer%d" % layer_index)

    def finish_step(self):
        for name, _, _ in self.buckets:                       # whatever the hooks did not cover
            self._reduce(name)
        for w, view in self._works:
            w.wait()
            if self.backend != "nccl":
                view.div_(self.world)
        self._works = []
        self._reserve_cus(False)
