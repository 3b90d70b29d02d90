"""Synthetic COCO-shaped pre-training batches and pinned-host staging of region features.

The reference's dataloaders (visualbert/dataloaders/coco_dataset.py:169-233, bert_field.py:79-109,
vcr.py:457-475) need AllenNLP, COCO on disk and a vocabulary file; none exist here.  This module
produces batches with the same keys / dtypes / shapes that VisualBERTFixedImageEmbedding.forward takes
(visualbert/models/model.py:234-260).  Detectron region features stay
pre-extracted off-GPU: FeatureStager keeps them in pinned host memory and streams each batch with an
asynchronous host-to-device copy (hipMemcpyAsync underneath) on a side stream, double-buffered.
"""
import torch


def pin_batch(batch):
    """move a host batch into pinned memory (what a feature store / loader worker would hand over)."""
    return {k: v.pin_memory() for k, v in batch.items()}


def synthetic_pretraining_batch(B, T=128, R=36, Dv=2048, vocab=30522, seed=0, device="cpu", ragged=False):
    g = torch.Generator().manual_seed(4000 + seed)
    ids = torch.randint(0, vocab, (B, T), generator=g, dtype=torch.int64)
    if ragged:
        lens = torch.randint(max(T // 2, 3), T + 1, (B,), generator=g, dtype=torch.int64)
        dims = torch.randint(max(R // 2, 1), R + 1, (B,), generator=g, dtype=torch.int64)
    else:
        lens = torch.full((B,), T, dtype=torch.int64)
        dims = torch.full((B,), R, dtype=torch.int64)
    ar = torch.arange(T).unsqueeze(0)
    mask = (ar < lens.unsqueeze(1)).to(torch.int64)
    ids = ids * mask
    type_ids = ((ar >= (lens.unsqueeze(1) // 2)).to(torch.int64)) * mask
    feats = 5.0 * torch.rand((B, R, Dv), generator=g, dtype=torch.float32)      # fc6-after-ReLU-like, >= 0
    feats = feats * (torch.arange(R).unsqueeze(0) < dims.unsqueeze(1)).unsqueeze(-1).float()
    sel = (torch.rand((B, T), generator=g) < 0.15) & (mask == 1)
    sel[:, 1] = True                                                            # >= 1 masked token per sample
    labels = torch.where(sel, ids, torch.full_like(ids, -1))
    nxt = torch.randint(0, 2, (B,), generator=g, dtype=torch.int64)
    batch = dict(bert_input_ids=ids, bert_input_mask=mask, bert_input_type_ids=type_ids, image_dim_variable=dims,
                 image_feat_variable=feats, masked_lm_labels=labels, is_random_next=nxt)
    return {k: v.to(device) for k, v in batch.items()}


def synthetic_batch(head, B, T, R, Dv=2048, vocab=30522, seed=0, device="cpu", ragged=False):
    """synthetic batch for BASELINE.json's three workloads, with the kwargs of VisualBERTFixedImageEmbedding.forward:
    "pretraining" (configs[1]), "vqa" (configs[3]: label = float32 [B, 3129] soft scores, vqa_dataset's answer scores),
    "nlvr" (configs[4]: two images, visual_embeddings_type 0 / 1 per half, label int64 [B]; nlvr_dataset.py:98-114)."""
    batch = synthetic_pretraining_batch(B, T, R, Dv, vocab, seed, "cpu", ragged)
    if head == "pretraining":
        return {k: v.to(device) for k, v in batch.items()}
    g = torch.Generator().manual_seed(5000 + seed)
    batch.pop("masked_lm_labels")
    batch.pop("is_random_next")
    if head == "vqa":
        lab = torch.zeros((B, 3129), dtype=torch.float32)
        idx = torch.randint(0, 3129, (B, 3), generator=g)
        lab.scatter_(1, idx, torch.tensor([[1.0, 0.6, 0.3]]).expand(B, 3).contiguous())
        batch["label"] = lab
    elif head == "nlvr":
        batch["label"] = torch.randint(0, 2, (B,), generator=g, dtype=torch.int64)
        vt = torch.zeros((B, R), dtype=torch.int64)
        vt[:, R // 2:] = 1
        batch["visual_embeddings_type"] = vt
    else:
        raise ValueError("synthetic_batch: head %r" % (head,))
    return {k: v.to(device) for k, v in batch.items()}


def mask_tokens(input_ids, maskable, vocab_size, mask_id, probability=0.15, generator=None, uniforms=None,
                random_ids=None):
    """Vectorised BERT masking of a whole batch on the host.

    Replaces the per-token Python loop `random_word` (pytorch_pretrained_bert/fine_tuning.py:272-308), which also rebuilds
    `list(tokenizer.vocab.items())` for every replaced token: a token is selected with `probability`; of the selected ones
    80 % become [MASK], 10 % a uniformly random vocabulary id, 10 % stay; the label is the ORIGINAL id at selected
    positions and -1 elsewhere.  The reference draws one uniform per token and reuses it (prob /= probability) to pick
    among the three outcomes -- so does this function, which makes it element-for-element identical to the loop when
    given the same uniforms (`uniforms`, `random_ids`: injected draws for tests; otherwise from `generator`).

    input_ids: int64 [B, T]; maskable: bool [B, T] (False for [CLS]/[SEP]/padding, which the reference never passes to
    random_word).  Returns (masked_ids, labels)."""
    ids = input_ids
    if uniforms is None:
        uniforms = torch.rand(ids.shape, generator=generator, dtype=torch.float64)
    if random_ids is None:
        random_ids = torch.randint(0, vocab_size, ids.shape, generator=generator, dtype=torch.int64)
    sel = (uniforms < probability) & maskable
    u2 = uniforms / probability
    to_mask = sel & (u2 < 0.8)
    to_rand = sel & (u2 >= 0.8) & (u2 < 0.9)
    out = torch.where(to_mask, torch.full_like(ids, mask_id), ids)
    out = torch.where(to_rand, random_ids, out)
    labels = torch.where(sel, ids, torch.full_like(ids, -1))
    return out, labels


def collate_pretraining(ids_a, ids_b, is_correct, features, vocab_size, mask_id, cls_id, sep_id, probability=0.15,
                        generator=None, uniforms=None, random_ids=None, pin=True):
    """One pre-training batch from tokenised caption pairs and per-image region features, built with whole-batch tensor
    operations straight into (pinned) pre-padded buffers (SURVEY 8f / N2 + N3).

    Replaces, for a batch at a time, convert_one_example_to_features_pretraining (dataloaders/bert_data_utils.py:168-247:
    [CLS] a [SEP] b [SEP], segment ids, lm labels with -1 on the special tokens), the per-field AllenNLP padding
    (dataloaders/bert_field.py:79-100: ids / mask / type ids with 0, lm labels with -1; coco_dataset.py:176-181: region
    features zero-padded to the most regions in the batch, image_dim_variable = region count) and the per-token
    random_word loop (via mask_tokens).  is_random_next = int(is_correct), as the reference's fields carry it
    (bert_data_utils.py:81, 258-259).

    ids_a / ids_b: lists of 1-D int64 tensors (ids_b entries may be None or empty: single-sentence examples);
    features: list of float32 [r_i, Dv] tensors.  Returns the kwargs dict of VisualBERTFixedImageEmbedding.forward on
    the host; with pin=True every tensor sits in pinned memory, ready for FeatureStager's asynchronous copies."""
    B = len(ids_a)
    la = torch.tensor([int(x.numel()) for x in ids_a], dtype=torch.int64)
    lb = torch.tensor([0 if (y is None) else int(y.numel()) for y in ids_b], dtype=torch.int64)
    has_b = lb > 0
    lens = la + 2 + torch.where(has_b, lb + 1, torch.zeros_like(lb))
    T = int(lens.max())

    def new(shape, dtype, fill):
        t = torch.empty(shape, dtype=dtype, pin_memory=bool(pin))
        return t.fill_(fill) if fill is not None else t

    ids = new((B, T), torch.int64, 0)
    ar = torch.arange(T).unsqueeze(0)
    # token scatter: a at columns 1..la, b at la+2..la+1+lb
    flat_a = torch.cat([x.reshape(-1) for x in ids_a]) if int(la.sum()) else torch.zeros(0, dtype=torch.int64)
    rows_a = torch.repeat_interleave(torch.arange(B), la)
    cols_a = 1 + torch.arange(int(la.sum())) - torch.repeat_interleave(torch.cumsum(la, 0) - la, la)
    ids[rows_a, cols_a] = flat_a
    if int(lb.sum()):
        flat_b = torch.cat([y.reshape(-1) for y in ids_b if y is not None and y.numel()])
        rows_b = torch.repeat_interleave(torch.arange(B), lb)
        cols_b = torch.repeat_interleave(la + 2, lb) + torch.arange(int(lb.sum())) - \
            torch.repeat_interleave(torch.cumsum(lb, 0) - lb, lb)
        ids[rows_b, cols_b] = flat_b
    ids[:, 0] = cls_id
    ids[torch.arange(B), la + 1] = sep_id
    ids[torch.arange(B)[has_b], (la + lb + 2)[has_b]] = sep_id
    in_a = (ar >= 1) & (ar <= la.unsqueeze(1))
    in_b = (ar >= (la + 2).unsqueeze(1)) & (ar <= (la + lb + 1).unsqueeze(1)) & has_b.unsqueeze(1)
    real = ar < lens.unsqueeze(1)
    masked, labels = mask_tokens(ids, in_a | in_b, vocab_size, mask_id, probability, generator, uniforms, random_ids)
    out = {
        "bert_input_ids": new((B, T), torch.int64, None).copy_(masked),
        "bert_input_mask": new((B, T), torch.int64, None).copy_(real.to(torch.int64)),
        "bert_input_type_ids": new((B, T), torch.int64, None).copy_(((ar > (la + 1).unsqueeze(1)) & real).to(torch.int64)),
        "masked_lm_labels": new((B, T), torch.int64, None).copy_(labels),
        "is_random_next": new((B,), torch.int64, None).copy_(torch.as_tensor([int(bool(c)) for c in is_correct])),
    }
    dims = torch.tensor([int(f.shape[0]) for f in features], dtype=torch.int64)
    R, Dv = int(dims.max()), int(features[0].shape[1])
    feats = new((B, R, Dv), torch.float32, 0.0)
    for b, f in enumerate(features):                       # one contiguous block copy per image into the padded slab
        feats[b, :f.shape[0]].copy_(f)
    out["image_dim_variable"] = new((B,), torch.int64, None).copy_(dims)
    out["image_feat_variable"] = feats
    return out


class RegionFeatureStore(object):
    """Pre-extracted detectron region features on disk, one .npy file of float32 [regions, Dv] per image, as the
    reference reads them (dataloaders/coco_dataset.py:144-155, image_feature_type "vqa_fix_100":
    np.load(folder/COCO_{split}2014_{image_id:012d}.npy), image_dim_variable = shape[0]).  Files are memory-mapped
    and copied ONCE, straight into a pinned pre-padded [B, R, Dv] slab -- the layout FeatureStager streams to HBM -- so the
    per-field AllenNLP padding of the reference (ArrayField -> torch.stack of padded copies, bert_field.py:79-109) and its
    pageable intermediate tensors disappear."""

    def __init__(self, folder, split_name="train", pattern="COCO_{split}2014_{image_id:012d}.npy"):
        self.folder = folder
        self.split_name = split_name
        self.pattern = pattern

    def path(self, image_id):
        import os
        return os.path.join(self.folder, self.pattern.format(split=self.split_name, image_id=int(image_id)))

    def load(self, image_id):
        """float32 [regions, Dv] view of one image's features (memory-mapped: no copy until it is read)."""
        import numpy as np
        a = np.load(self.path(image_id), mmap_mode="r")
        if a.ndim != 2:
            raise ValueError("%s: expected a [regions, Dv] array, got shape %s" % (self.path(image_id), a.shape))
        return a

    def read_batch(self, image_ids, regions=None, out=None, pin=True, wait=None):
        """-> dict(image_feat_variable float32 [B, R, Dv] zero padded, image_dim_variable int64 [B]) in pinned memory.
        regions: pad / truncate to this many regions (None: the most regions in the batch, like the reference's padding);
        out: a previous result to overwrite in place (a slot of a ring);
        wait: the event FeatureStager.stage() returned when `out` was last handed to it -- the DMA out of the slab must have
              executed before the host rewrites it, so it is synchronised HERE, before the first byte is written (a wait
              inside the next stage() call would come after the refill, i.e. after the race)."""
        import numpy as np
        if wait is not None:
            wait.synchronize()
        arrays = [self.load(i) for i in image_ids]
        B = len(arrays)
        Dv = int(arrays[0].shape[1])
        R = int(regions) if regions is not None else max(int(a.shape[0]) for a in arrays)
        if out is not None and tuple(out["image_feat_variable"].shape) == (B, R, Dv):
            feats, dims = out["image_feat_variable"], out["image_dim_variable"]
        else:
            feats = torch.empty((B, R, Dv), dtype=torch.float32, pin_memory=bool(pin))
            dims = torch.empty((B,), dtype=torch.int64, pin_memory=bool(pin))
        fv = feats.numpy()
        for b, a in enumerate(arrays):
            if a.shape[1] != Dv:
                raise ValueError("feature width differs inside a batch: %d vs %d" % (a.shape[1], Dv))
            r = min(int(a.shape[0]), R)
            np.copyto(fv[b, :r], a[:r], casting="same_kind")
            if r < R:
                fv[b, r:] = 0.0
            dims[b] = r
        return {"image_feat_variable": feats, "image_dim_variable": dims}


class FeatureStager(object):
    """Double-buffered pinned-host -> HBM streaming of a batch dict (the features are 295 KB/sample).

    Each slot owns its device tensors and (for pageable inputs) its pinned staging buffers.  Hazards handled here, not
    by the caller:
      * the consumer runs raw-pointer HIP kernels on the compute stream, so the caching allocator must not hand a staged
        tensor's memory to the next copy while a queued kernel still reads it: the slot's device tensors are REUSED, and
        before a slot is overwritten the copy stream waits for an event recorded on the compute stream at that moment --
        everything the consumer enqueued for the slot's previous contents is ahead of that event;
      * a slot's OWN pinned staging buffer (pageable inputs) is rewritten by the host only after the slot's previous copy has
        completed.
    Contract for inputs that are ALREADY pinned (RegionFeatureStore slabs): they are copied from in place, so the caller must
    not rewrite such a slab until the event returned by the stage() call that consumed it has completed -- pass that event as
    `wait=` to RegionFeatureStore.read_batch(out=slab, ...) or call `stager.wait_host(slot)` before refilling.
    `strict_pinned=True` turns the most common violation into an error: a pinned slab that is staged AGAIN while the copy of its
    previous contents is still in flight was, in a refill loop, rewritten under that copy (a caller that re-stages an unchanged
    slab on purpose, like bench.py's streaming leg, keeps the default)."""

    def __init__(self, device, strict_pinned=False):
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.strict_pinned = bool(strict_pinned)
        self._pinned = {}
        self._dev = {}
        self._copied = {}
        self._slab_events = {}                          # data_ptr of a caller-pinned input -> event of the last copy out of it

    def stage(self, host_batch, slot=0):
        """enqueue async copies of `host_batch` (CPU tensors) on the side stream; returns (device batch, event).
        The returned tensors stay valid until the next stage() call for the same slot."""
        out = {}
        consumed = torch.cuda.Event()
        consumed.record(torch.cuda.current_stream(self.device))      # the slot's previous contents are consumed before this
        prev = self._copied.get(slot)
        # entries whose copy has completed carry no information any more (and a freed slab's address may be reused by another
        # tensor, which must not inherit the old event): drop them before looking anything up
        self._slab_events = {p: e for p, e in self._slab_events.items() if not e.query()}
        pinned = {k: v.is_pinned() for k, v in host_batch.items()}   # a driver query: once per tensor and call
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(consumed)
            for k, v in host_batch.items():
                if pinned[k]:                           # the feature store already lives in pinned memory: copied from in
                    pin = v                             # place (the caller keeps it intact until `ev` completes, see above)
                    last = self._slab_events.get(v.data_ptr())
                    if self.strict_pinned and last is not None and not last.query():
                        raise RuntimeError("FeatureStager.stage: pinned input %r is staged again while the copy of its previous "
                                           "contents is still in flight -- refill a slab only after the event stage() returned "
                                           "has completed (read_batch(..., wait=ev) or stager.wait_host(slot))" % k)
                else:
                    key = (slot, k)
                    pin = self._pinned.get(key)
                    if pin is None or pin.shape != v.shape or pin.dtype != v.dtype:
                        pin = torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
                        self._pinned[key] = pin
                    elif prev is not None:
                        prev.synchronize()              # the previous copy out of this buffer has finished
                    pin.copy_(v)
                key = (slot, k)
                d = self._dev.get(key)
                if d is None or d.shape != pin.shape or d.dtype != pin.dtype:
                    d = torch.empty(pin.shape, dtype=pin.dtype, device=self.device)
                    self._dev[key] = d
                d.copy_(pin, non_blocking=True)
                out[k] = d
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._copied[slot] = ev
        for k, v in host_batch.items():
            if pinned[k]:
                self._slab_events[v.data_ptr()] = ev
        return out, ev

    def wait_host(self, slot=0):
        """block until the last stage() of `slot` has finished reading host memory: after this the caller may rewrite the
        pinned tensors it passed to that call."""
        ev = self._copied.get(slot)
        if ev is not None:
            ev.synchronize()
