"""Data-parallel gradient synchronisation: one process per GPU, RCCL over xGMI.

The reference's only parallelism is single-process nn.DataParallel (models/model_wrapper.py:146):
per step it broadcasts all weights GPU0 -> others, scatters the batch, gathers one loss per replica,
takes loss.mean() (model_wrapper.py:75) and reduce-adds gradients onto GPU 0.  Here every rank owns
a full replica and its own shard of the minibatch, computes its own mean loss, and the flat fp32
gradient arena is all-reduced with AVERAGING -- which reproduces the reference's mean of per-replica
means exactly -- in L+2 contiguous buckets launched while backward is still running:

    heads+pooler | layer L-1 | ... | layer 0 | embeddings (incl. the tied decoder weight, last)

Each bucket is handed to RCCL from an autograd hook the moment its layer's wgrad kernels have been
enqueued.  Default path: the C ABI's communicator (include/visualbert_hip.h: vb_comm_init from an RCCL
unique id shipped through torch.distributed's store, vb_allreduce_bucket on a side HIP stream behind an
event recorded on the compute stream) -- launch, stream ordering and the compute-unit reservation below
are then one mechanism owned by this class.  `use_abi_comm=False` (and every non-RCCL backend, e.g. the
gloo CPU tests) goes through torch.distributed's all_reduce instead.  xGMI is point-to-point
(7 links x ~153 GB/s per GPU): a ring all-reduce moves 2(N-1)/N x payload per GPU and is per-link
bound, so buckets are whole layers (28 MB fp32 at BERT-base) -- large enough to run at link speed,
small enough that the last one (embeddings, 94 MB + heads) is the only exposed tail.

Compute units for the collective.  Read from the code objects (hipcc -Rpass-analysis=kernel-resource-usage, launch sizes in
csrc/gemm.hip / attention.hip): the persistent 256x256 GEMM and the grouped weight-gradient kernel launch 512 threads x 256
VGPRs with 163,840 B of LDS -- the WHOLE register file and LDS of a CU; two workgroups of the 256x128 kernel (256 threads x 256
VGPRs, 81,920 B each) fill it the same way; the one-pass attention backward holds 12 waves x 168 VGPRs and 124 KB.  None of them
can share a CU with an RCCL workgroup (registers alone forbid it): while the collective is resident on c CUs, c workgroups of a
persistent launch wait for a CU -- the launch is stretched by up to the collective's remaining residency -- and c CUs host one
256x128 workgroup (or none) instead of two.  Reserving CUs up front (VB_COMM_CUS=c: RCCL capped to c channels via
NCCL_MAX_NCHANNELS; while buckets are in flight the GEMM dispatcher sends what would have been persistent 256x256 launches to
the one-workgroup-per-tile kernel and runs the weight-gradient kernel on CUs - c workgroups; bitwise results then differ from a run
without the reservation) is OFF by default: on one GPU the persistent GEMM's N = 768 shapes paid a second
round for ANY grid below 246 workgroups (30 -> 47 us, 78 -> 128 us at 224) -- which is why the reservation now pins the
per-tile kernel for them.  DESIGN.md section 6 carries the bound this puts on the scaling prediction.
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


class RcclCommunicator(object):
    """the C ABI's communicator (vb_comm_*): rank 0 draws the RCCL unique id, torch.distributed's process group carries
    its 128 bytes to the other ranks (a host-side object broadcast), every rank joins on its CURRENT device."""

    def __init__(self, process_group=None):
        import ctypes
        L = _lib.lib()
        rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
        idbuf = (ctypes.c_char * _lib.VB_COMM_ID_BYTES)()
        payload = [None]
        if rank == 0:
            _lib.check(L.vb_comm_unique_id(idbuf), "vb_comm_unique_id")
            payload = [bytes(idbuf.raw)]
        dist.broadcast_object_list(payload, src=0, group=process_group)
        idbuf.raw = payload[0]
        handle = ctypes.c_void_p()
        _lib.check(L.vb_comm_init(idbuf, rank, world, ctypes.byref(handle)), "vb_comm_init")
        self.handle, self.rank, self.world = handle, rank, world

    def allreduce(self, t, average, stream):
        """in place on `t` (contiguous fp32 / bf16), enqueued on `stream` (a torch.cuda.Stream)."""
        import ctypes
        _lib.check(_lib.lib().vb_allreduce_bucket(self.handle, _lib.ptr(t), t.numel(), _lib.dtype_code(t.dtype),
                                                  1 if average else 0, ctypes.c_void_p(stream.cuda_stream)),
                   "vb_allreduce_bucket")

    def close(self):
        if self.handle is not None:
            _lib.lib().vb_comm_destroy(self.handle)
            self.handle = None


class DataParallelGradSync(object):
    def __init__(self, objective, process_group=None, overlap=True, use_abi_comm=True):
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
        self.comm = None
        self.comm_stream = None
        if use_abi_comm and self.backend == "nccl":
            self.comm = RcclCommunicator(process_group)
            self.comm_stream = torch.cuda.Stream(device=objective.arena.grad.device)
        self.comm_kind = "C-ABI vb_allreduce_bucket" if self.comm is not None else "torch.distributed %s" % self.backend
        from . import ops
        ops.set_replica(dist.get_rank(process_group))       # replicas must not repeat each other's dropout masks
        self._install()

    def close(self):
        if self.comm is not None:
            torch.cuda.synchronize()
            self.comm.close()
            self.comm = None

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
    def begin_step(self, sync=True):
        """sync=False: a micro-step whose gradients no optimizer step will read (gradient_accumulation_steps > 1: the reference's
        ModelWrapper.step zeroes the gradients at the top of EVERY call, models/model_wrapper.py:64, so only the last micro-batch
        of a group reaches the optimizer) -- the hooks and finish_step() then move nothing over xGMI."""
        self._sync = bool(sync)
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
        if self.comm is not None:
            # the bucket's producers are enqueued on the compute stream: the side stream waits for an event recorded there
            # now, then RCCL averages the bucket in place while the compute stream goes on with the next layer's backward
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(view.device))
            self.comm_stream.wait_event(ready)
            self.comm.allreduce(view, True, self.comm_stream)
            return
        if self.backend == "nccl":
            w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
        else:                                                 # gloo (CPU tests): SUM then scale
            w = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._works.append((w, view))

    def _reserve_cus(self, on):
        """leave comm_cus() CUs to RCCL while buckets are in flight (see the module docstring): a launch option of the
        COMPUTE stream (vb_stream_set_opts), not a process-wide switch."""
        if self.backend != "nccl" or self.world == 1 or comm_cus() <= 0 or on == self._reserved:
            return
        self._reserved = on
        import ctypes
        sp = _lib.stream_ptr()
        if on:
            # only persistent_workgroups changes: whatever else the user attached to this stream (nt_kernel for an A/B run,
            # attn_two_pass) is read back first and restored when the reservation ends
            cur = _lib.StreamOpts(0, 0, 0, 0)
            _lib.check(_lib.lib().vb_stream_get_opts(sp, ctypes.byref(cur)), "vb_stream_get_opts")
            self._saved_opts = (cur.persistent_workgroups, cur.nt_kernel, cur.attn_two_pass, cur.reserved)
            cus = torch.cuda.get_device_properties(self.obj.arena.grad.device).multi_processor_count
            # only the workgroup budget is lowered; the GEMM dispatcher (csrc/gemm.hip: dispatch_pipe) reads it and moves exactly the
            # launches that would have taken the persistent 256x256 kernel -- whose grid IS the CU count: short of CUs it pays a whole
            # second round on the N = 768 shapes, see the module docstring -- to the one-workgroup-per-TILE kernel (a CU that RCCL holds
            # simply takes fewer tiles); the small-problem kernels of small per-GPU batches keep their rule, measured against the
            # reduced budget; the grouped weight-gradient kernel, persistent by construction, gets the reduced workgroup count.
            # NOTE: a GEMM that changes kernel changes its summation order -- with VB_COMM_CUS set, results differ bitwise from a run
            # without it (same bounds; replicas still agree with each other: every rank reserves the same way).
            o = _lib.StreamOpts(max(8, (cus - comm_cus()) // 8 * 8), cur.nt_kernel, cur.attn_two_pass, cur.reserved)
            _lib.check(_lib.lib().vb_stream_set_opts(sp, ctypes.byref(o)), "vb_stream_set_opts")
        else:
            saved = getattr(self, "_saved_opts", (0, 0, 0, 0))
            if any(saved):
                o = _lib.StreamOpts(*saved)
                _lib.check(_lib.lib().vb_stream_set_opts(sp, ctypes.byref(o)), "vb_stream_set_opts")
            else:
                _lib.check(_lib.lib().vb_stream_set_opts(sp, None), "vb_stream_set_opts")

    def _layer_ready(self, layer_index):
        if not getattr(self, "_sync", True):
            return
        # everything above this layer in the graph has finished enqueuing its backward
        self._reduce("heads")
        self._reduce("layer%d" % layer_index)

    def _reduce_touched(self):
        """which parameters received a gradient on ANY rank this step: the per-tensor flags the fused optimizer skips on
        (optimization.BertAdam._touched_flags) ride one more tiny all-reduce, so that every rank takes the same decision --
        a rank-local flag would let replicas diverge silently (weight decay applied on one rank only)."""
        a = self.obj.arena
        flags = a.touched_flags().clone()
        if self.comm is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(flags.device))
            self.comm_stream.wait_event(ready)
            self.comm.allreduce(flags, True, self.comm_stream)      # mean > 0  <=>  touched somewhere
            flags.record_stream(self.comm_stream)
        elif self.backend == "nccl":
            self._works.append((dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.pg, async_op=True), None))
        else:
            self._works.append((dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None))
        a.touched_synced = flags

    def finish_step(self):
        if not getattr(self, "_sync", True):
            return
        for name, _, _ in self.buckets:                       # whatever the hooks did not cover
            self._reduce(name)
        self._reduce_touched()
        for w, view in self._works:
            w.wait()
            if self.backend != "nccl" and view is not None:
                view.div_(self.world)
        self._works = []
        if self.comm is not None:                             # the optimizer (compute stream) waits for the last bucket
            done = torch.cuda.Event()
            done.record(self.comm_stream)
            torch.cuda.current_stream(self.obj.arena.grad.device).wait_event(done)
        self._reserve_cus(False)

    def measure_allreduce(self, barrier, reps=5):
        """stand-alone all-reduce of the whole gradient arena in the step's buckets (nothing to overlap with): payload,
        milliseconds and bus bandwidth 2(N-1)/N x bytes / time -- to be read against xGMI's ~153 GB/s per link."""
        import time
        g = self.obj.arena.grad
        nbytes = g.numel() * g.element_size()
        saved = g.clone()

        def once():
            self.begin_step()
            self.finish_step()

        once()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            once()
        barrier()
        dt = (time.perf_counter() - t0) / reps
        g.copy_(saved)
        self.obj.arena.touched_synced = None
        n = self.world
        return dict(payload_bytes=nbytes, buckets=len(self.buckets), ms=round(dt * 1e3, 3),
                    bus_GBps=round(2.0 * (n - 1) / n * nbytes / dt / 1e9, 2) if n > 1 else 0.0,
                    algo_GBps=round(nbytes / dt / 1e9, 2), ranks=n, path=self.comm_kind)
