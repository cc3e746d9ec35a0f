"""Host side of the data-parallel training step (mirror of train.py:259-332, 344-382).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Parameters are
replicated, the batch is sharded along axis 0 (reference: `NamedSharding(mesh, P("batch"))`,
train.py:238-244, 280), and the ONLY exchange step is the mean all-reduce of the flat fp32 gradient
bucket with the loss and the finite flags folded into it — the explicit form of the gradient
all-reduce XLA's GSPMD inserts implicitly in the reference.  ``DataParallelStep`` is the one
implementation of that protocol: bench.py times it, tests/test_dp_cpu.py drives it under gloo with
two ranks, a training loop calls it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

def create_learning_rate_schedule(base_learning_rate: float, warmup_steps: int, cosine_decay_steps: int) -> Callable[[int], float]:
    """train.py:455-466: linear warm-up from 0 to ``base_learning_rate`` over ``warmup_steps`` updates, then a cosine decay
    to 0 over ``cosine_decay_steps`` (optax.join_schedules of linear_schedule and cosine_decay_schedule).  The returned
    function maps the update COUNT (0 for the first optimizer update, as optax counts) to the base learning rate; the
    per-layer factors layer_lr_decay ** (max_depth - depth) (train.py:697-713) multiply it inside amt_adamw_step, which is
    exact because both pieces of the schedule are linear in their base value.  Use:
    ``model.adamw_step(grads, lr_base=schedule(step - 1), step=step)`` with ``step`` counting from 1."""
    base = float(base_learning_rate)
    warm, decay = int(warmup_steps), int(cosine_decay_steps)

    def schedule(count: int) -> float:
        c = int(count)
        if c < warm:
            return base * min(max(c / warm, 0.0), 1.0)
        if decay <= 0:
            return 0.0
        return base * 0.5 * (1.0 + math.cos(math.pi * min(c - warm, decay) / decay))

    return schedule


@dataclass
class LossScaleController:
    """train.py:344, 369-382: halve on non-finite (and roll back), double while scaled_loss < 1e4.
    bf16 needs no scaling; the controller is kept because it is part of the step's observable
    behaviour.  `enabled=False` pins the scale to 1."""
    scale: float = 1.0
    enabled: bool = True
    growth_threshold: float = 10_000.0

    def update(self, grads_valid: bool, loss: float, scaled_loss: float) -> bool:
        """Returns True if the step must be skipped and state restored (train.py:369-377)."""
        finite = grads_valid and loss == loss and abs(loss) != float("inf")
        if not finite:
            if self.enabled:
                self.scale = self.scale / 2
            return True
        if self.enabled and scaled_loss < self.growth_threshold:
            self.scale = self.scale * 2
        return False


# ======================================================================================================
# The device-resident data-parallel step (the form bench.py times and a training loop runs).
#
# Everything stays on the device between steps: gradients, loss and the finite flags live in ONE flat fp32
# bucket, the collectives average it over the ranks, and the optimizer is gated by the all-reduced flag ON THE
# DEVICE (amt_adamw_step_gated), so a step has no host synchronisation and can be replayed as hipGraphs.
# bench.py, the gloo world-2 test (tests/test_dp_cpu.py) and a training loop all go through DataParallelStep, so
# the protocol that is tested is the protocol that is timed.
# ======================================================================================================

class GradBucket:
    """``[ valid_all, 0, 0, 0 | gradients (amt flat layout, n) | loss, valid_first ]`` in one fp32 buffer.

    * ``valid_all``  : 1.0 iff every gradient of this rank's step is finite (train.py:320-322);
    * ``valid_first``: the same flag for the first half of the two-phase step (decoder, transformer, final CNN norm);
    * ``loss``       : mean over this rank's clips of the summed BCE (train.py:60-62).

    The four head floats keep the gradient range 16-byte aligned.  After the mean all-reduce a flag is 1.0 iff it
    was 1 on every rank, and loss / gradients are the means over the global batch (train.py:62)."""

    HEAD = 4
    TAIL = 2

    def __init__(self, n_params: int, device, buf: Optional[torch.Tensor] = None):
        self.n = int(n_params)
        if buf is None:
            buf = torch.zeros(self.HEAD + self.n + self.TAIL, dtype=torch.float32, device=device)
        if buf.dtype != torch.float32 or buf.numel() != self.HEAD + self.n + self.TAIL:
            raise ValueError("a gradient bucket is HEAD + n + TAIL float32 values")
        self.buf = buf
        self.loss_div = 1.0       # the loss word holds loss * loss_div (in-place protocol: the SUM over ranks)
        self.head = self.buf[:self.HEAD]
        self.grads = self.buf[self.HEAD:self.HEAD + self.n]
        self.tail = self.buf[self.HEAD + self.n:]

    @classmethod
    def of_model(cls, model) -> "GradBucket":
        """The model's OWN gradient accumulator as the bucket (include/amt.h: amt_grad_bucket): the in-place protocol -
        backward sums, collectives and optimizer all work on this one buffer, nothing is copied or unscaled in between."""
        buf, head, n, tail = model.grad_bucket()
        if head != cls.HEAD or tail != cls.TAIL:
            raise RuntimeError("amt_grad_bucket layout differs from GradBucket's")
        return cls(n, buf.device, buf=buf)

    def first_range(self, split: int) -> torch.Tensor:
        """complete after phase 1: gradients [split, n), loss, valid_first"""
        return self.buf[self.HEAD + split:]

    def second_range(self, split: int) -> torch.Tensor:
        """complete after phase 2: valid_all and gradients [0, split)"""
        return self.buf[:self.HEAD + split]

    @property
    def valid_all(self) -> torch.Tensor:      # float32[1] view: the optimizer gate
        return self.buf[0:1]


class InputBlock:
    """The audio ``[B, 2, T]`` and the labels ``[B, S, V]`` of one batch as views of ONE contiguous byte buffer (device, or
    pinned host): a batch travels in one copy.  The reference hands both over as float16 (grain_loader.py:88)."""

    def __init__(self, audio_shape, audio_dtype, labels_shape, labels_dtype, device="cpu", pin: bool = False):
        na = int(torch.Size(audio_shape).numel()) * torch.empty((), dtype=audio_dtype).element_size()
        ny = int(torch.Size(labels_shape).numel()) * torch.empty((), dtype=labels_dtype).element_size()
        self.label_offset = (na + 255) // 256 * 256
        self.buf = torch.empty(self.label_offset + ny, dtype=torch.uint8, device=device, pin_memory=pin)
        self.audio = self.buf[:na].view(audio_dtype).view(audio_shape)
        self.labels = self.buf[self.label_offset:self.label_offset + ny].view(labels_dtype).view(labels_shape)
        self.audio._amt_block = self
        self.labels._amt_block = self

    @staticmethod
    def like(other: "InputBlock", device=None, pin: bool = False):
        return InputBlock(other.audio.shape, other.audio.dtype, other.labels.shape, other.labels.dtype,
                          device=other.buf.device if device is None else device, pin=pin)

    def same_layout(self, other) -> bool:
        return (isinstance(other, InputBlock) and self.buf.numel() == other.buf.numel() and self.audio.dtype == other.audio.dtype
                and self.labels.dtype == other.labels.dtype and self.audio.shape == other.audio.shape
                and self.labels.shape == other.labels.shape)

    def __iter__(self):          # (audio, labels) = block
        return iter((self.audio, self.labels))

    def __getitem__(self, i):
        return (self.audio, self.labels)[i]


class HipTrainBackend:
    """Per-rank forward/backward/optimizer on the HIP library for fixed, device-resident inputs."""

    def __init__(self, model, audio: torch.Tensor, labels: torch.Tensor, bucket: GradBucket, scale: float = 1.0,
                 use_optimizer: bool = True, accum_steps: int = 1, fused: bool = False, optimizer_args: Optional[dict] = None):
        # once, here: no conversion launch inside a captured step.  Audio in the context's input type, labels as given if they
        # are float16 (the reference's batches), float32 otherwise; both in ONE device block (InputBlock): a batch is one upload
        blk = InputBlock(audio.shape, model.audio_dtype, labels.shape,
                         torch.float16 if labels.dtype == torch.float16 else torch.float32, device=audio.device)
        blk.audio.copy_(audio)
        blk.labels.copy_(labels)
        audio, labels = blk.audio, blk.labels
        self.model, self.audio, self.labels, self.bucket, self.scale = model, audio, labels, bucket, float(scale)
        self.use_optimizer = use_optimizer
        self.optimizer_args = dict(optimizer_args or {})      # layer_lr_decay, b1, b2, eps, weight_decay, clip_norm (model.adamw_step)
        # fused = the fused protocol (include/amt.h): the gradients never leave the context's accumulator, AdamW reads the
        # raw sums (norm pass + commit pass, one launch) and leaves the accumulator zeroed.  With more than one rank the
        # bucket must BE that accumulator (GradBucket.of_model): the collectives then run on it in place and the rank runs
        # the single-rank program plus the collectives (in_place).
        self.fused = bool(fused) and use_optimizer
        self.in_place = self.fused and bucket.buf.data_ptr() == model.grad_bucket()[0].data_ptr()
        self.world = 1            # set by DataParallelStep: 1 / world multiplies the SUM-reduced gradients in the optimizer
        # train.py:275-310: the rank's batch in `accum_steps` minibatches, gradients accumulated in fp32 ON THE DEVICE (the
        # context's accumulator: amt_forward_backward_accum), grads / (grad_scale * steps) in the last minibatch's final pass
        self.accum_steps = int(accum_steps)
        if self.accum_steps < 1 or audio.shape[0] % self.accum_steps:
            raise ValueError("the rank's batch must be divisible by accum_steps")
        # input sets: the step reads FIXED device tensors (its launches are captured); a second set (add_input_set) lets the
        # upload of the next batch land where the NEXT step reads it while this step still reads its own (InputFeeder)
        self._sets = [blk]
        self.active = 0
        self._mb = self._minibatches(audio, labels)
        dev = audio.device
        self.loss_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        self.valid_buf = torch.ones(1, dtype=torch.int32, device=dev)
        self.split = model.grad_bucket_split()
        if use_optimizer:
            model.init_optimizer(self.optimizer_args.get("layer_lr_decay", 0.7))

    def _minibatches(self, audio, labels):
        mb = audio.shape[0] // self.accum_steps
        return [(audio[i * mb:(i + 1) * mb], labels[i * mb:(i + 1) * mb]) for i in range(self.accum_steps)]

    def add_input_set(self):
        """A further pair of input tensors (a copy of set 0, so that rehearsal runs on it are finite); returns its index."""
        blk = InputBlock.like(self._sets[0])
        blk.buf.copy_(self._sets[0].buf)
        self._sets.append(blk)
        return len(self._sets) - 1

    def use_set(self, i: int):
        """The input set the NEXT issued (or captured) step reads."""
        self.active = int(i)
        self.audio, self.labels = self._sets[self.active].audio, self._sets[self.active].labels
        self._mb = self._minibatches(self.audio, self.labels)

    def full(self, seed: int):
        b, k = self.bucket, self.accum_steps
        # Every minibatch of a step draws its masks from the SAME key, as the reference does: compute_training_step splits
        # the key once per step (train.py:274) and minibatch_scan_body hands that one `key` to compute_loss in every scan
        # iteration (train.py:283-290), which splits it per sample of the minibatch (train.py:53).
        for i, (a, y) in enumerate(self._mb):
            self.model.loss_and_grads(a, y, scale=self.scale, seed=seed, train=True, grads_out=None if self.fused else b.grads,
                                      sync=False, accum=(i, k), loss_out=self.loss_buf, valid_out=self.valid_buf,
                                      fused_optimizer=self.fused)
        if self.in_place:       # the rank's loss next to its gradient sums: reduced with them (the flags come out of the optimizer)
            self.model.pack_tail(self.loss_buf, self.valid_buf, tail_loss=b.tail[0:1])
        elif not self.fused:    # (fused: the flag is an output of the optimizer's commit pass, packed by apply())
            self.model.pack_tail(self.loss_buf, self.valid_buf, head_valid=b.head[0:1], tail_loss=b.tail[0:1], tail_valid=b.tail[1:2])

    def discard(self):
        """Drop the gradients of a step that will not be applied (warm-up / capture rehearsal runs in prepare())."""
        if self.fused:
            self.model.reset_grad_accumulator()

    def half(self, phase: int, seed: int):
        b, k = self.bucket, self.accum_steps
        if self.fused and not self.in_place:
            raise RuntimeError("the two-phase step of the fused protocol needs the model's own bucket (GradBucket.of_model)")
        gout = None if self.in_place else b.grads      # in place: raw sums stay in the accumulator, which IS the bucket
        if phase == 1:      # all minibatches but the last run whole; the last one is the step that is cut in two
            for i, (a, y) in enumerate(self._mb[:-1]):
                self.model.loss_and_grads(a, y, scale=self.scale, seed=seed, train=True, grads_out=gout, sync=False,
                                          accum=(i, k), loss_out=self.loss_buf, valid_out=self.valid_buf,
                                          fused_optimizer=self.in_place)
        a, y = self._mb[-1]
        self.model.loss_and_grads_phase(phase, a, y, gout, self.loss_buf, self.valid_buf,
                                        scale=self.scale, seed=seed, train=True, accum=(k - 1, k))
        if self.in_place:
            if phase == 1:
                self.model.pack_tail(self.loss_buf, self.valid_buf, tail_loss=b.tail[0:1])
        elif phase == 1:    # captured with the half: no eager launches between the graphs and the collectives
            self.model.pack_tail(self.loss_buf, self.valid_buf, tail_loss=b.tail[0:1], tail_valid=b.tail[1:2])
        else:
            self.model.pack_tail(self.loss_buf, self.valid_buf, head_valid=b.head[0:1])

    def full_signal(self, seed: int):
        """The whole step as ONE launch sequence that publishes "first half complete" on the device behind the first CNN
        Block chain (include/amt.h: amt_first_half_wait): the overlapped form without cutting the step in two host calls.
        In-place protocol only; the library writes the rank's loss into the bucket's tail in front of the signal."""
        if not self.in_place:
            raise RuntimeError("full_signal needs the in-place protocol (fused=True over GradBucket.of_model)")
        k = self.accum_steps
        for i, (a, y) in enumerate(self._mb[:-1]):
            self.model.loss_and_grads(a, y, scale=self.scale, seed=seed, train=True, grads_out=None, sync=False,
                                      accum=(i, k), loss_out=self.loss_buf, valid_out=self.valid_buf, fused_optimizer=True)
        a, y = self._mb[-1]
        self.model.loss_and_grads_phase(3, a, y, None, self.loss_buf, self.valid_buf, scale=self.scale, seed=seed, train=True,
                                        accum=(k - 1, k))

    def wait_first_half(self, timeout_ms: int = 10000) -> bool:
        """True: the signal arrived.  False: it did not within ``timeout_ms``; the compute stream has then been synchronised
        (the first range IS complete, its collective may be issued) and the signal counters re-aligned (amt_first_half_sync),
        so a later wait pairs with the right signal again."""
        try:
            self.model.first_half_wait(int(timeout_ms))
            return True
        except RuntimeError:
            self.model.first_half_sync()      # hipStreamSynchronize + "every signal so far is consumed"
            return False

    def sync_signals(self):
        self.model.first_half_sync()

    def advance_seed(self):
        self.model.advance_seed(1)      # one key per step (train.py:274), whatever the number of minibatches

    def apply(self, step: int, lr_base: float):
        # Gated on the all-reduced flag, on the device.  step = -1: the bias corrections come from the context's DEVICE
        # count of applied updates, which a skipped (non-finite) step does not advance - optax's count behaves the same
        # way and the reference rolls its optimizer state back (train.py:369-377); `step` on the host would drift.
        if self.in_place:
            # the accumulator holds the SUM over ranks of the raw sums, the loss word the sum of the ranks' losses: the
            # optimizer multiplies by 1 / world and gates on isfinite(|update|^2) && isfinite(loss sum) - the same verdict on
            # every rank, no flag collective (include/amt.h: amt_adamw_step_reduced)
            b = self.bucket
            b.loss_div = float(self.world)
            self.model.adamw_step_reduced(lr_base, -1, 1.0 / self.world, b.tail[0:1], valid=self.valid_buf, **self.optimizer_args)
            self.model.pack_tail(b.tail[0:1], self.valid_buf, head_valid=b.head[0:1], tail_valid=b.tail[1:2])
        elif self.fused:
            b = self.bucket
            self.model.adamw_step(None, lr_base=lr_base, step=-1, valid=self.valid_buf, **self.optimizer_args)
            self.model.pack_tail(self.loss_buf, self.valid_buf, head_valid=b.head[0:1], tail_loss=b.tail[0:1], tail_valid=b.tail[1:2])
        elif self.use_optimizer:
            self.model.adamw_step(self.bucket.grads, lr_base=lr_base, step=-1, valid_mean=self.bucket.valid_all, **self.optimizer_args)

    def applied_steps(self) -> int:
        return self.model.optimizer_count() if self.use_optimizer else 0

    def generation(self) -> int:
        return self.model.plan_generation()


class InputFeeder:
    """A fresh batch per step from pinned host memory, uploaded on a copy stream while the previous step computes
    (the reference uploads each minibatch inside its step: ``jax.device_put(batch, batch_sharding)``, train.py:280).

    The step (a captured hipGraph) reads FIXED device tensors, and it reads them until its very end (the Stem's backward
    re-reads the audio), so batch k + 1 cannot be uploaded over batch k while step k runs.  Two forms:

    * ``InputFeeder(backend=be)`` (bench.py, train()): the backend gets a SECOND input set and the step is captured once per
      set (DataParallelStep.prepare); the upload of batch k + 1 goes straight into the set step k does not read, ``take()``
      makes the compute stream wait for it and switches the backend to that set.  Nothing is copied on the device.
    * ``InputFeeder(audio_dev, labels_dev)``: one input pair, two staging pairs; ``take()`` copies staging -> inputs on the
      device in front of the step (6.6 MB at the reference configuration: two small launches, ~1 % of the step).

    ``prefetch()`` may be called right after the step has been issued.  Which way the streams wait for each other matters
    on this platform (scratch/feeder_cost.py, profiles/r04_notes.md): the compute stream waiting for an event of the copy
    stream costs ~20 us per step, and ANOTHER stream waiting for an event recorded on the busy compute stream costs
    0.12 ms per step (round 4's first form: 3.15 against 3.02 ms).  So both hand-overs are HOST waits on events that are long
    complete when they are waited for - "this buffer may be overwritten" on an event the compute stream recorded a step
    earlier, "the upload has landed" on the copy stream's event, 0.13 ms after it was issued a step ago - and no stream ever
    waits for another."""

    def __init__(self, audio_dev: Optional[torch.Tensor] = None, labels_dev: Optional[torch.Tensor] = None, backend=None):
        self.backend = backend
        if backend is not None:
            while len(backend._sets) < 2:
                backend.add_input_set()
            self.stage = list(backend._sets[:2])                         # uploads land where a step reads them (InputBlocks)
            self.audio, self.labels = backend._sets[0].audio, backend._sets[0].labels
            self.up = 1 - backend.active                                 # first upload: the set the NEXT step will read
        else:
            self.audio, self.labels = audio_dev, labels_dev
            self.stage = [(torch.empty_like(audio_dev), torch.empty_like(labels_dev)) for _ in range(2)]
            self.up = 0
        self.down = self.up                                              # next pair to take from
        self.copy_stream = torch.cuda.Stream(device=self.audio.device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]         # upload into pair i finished (copy stream)
        self.consumed = [None, None]                                   # pair i has been read for the last time (compute stream)
        self.pending = 0
        self.bytes_per_batch = self.audio.numel() * self.audio.element_size() + self.labels.numel() * self.labels.element_size()
        self._ring, self._ring_ev, self._slot = None, None, 0      # two pinned host pairs for prefetch_from_host

    @staticmethod
    def pin(audio, labels, audio_dtype, labels_dtype=torch.float32):
        """host batch -> ONE pinned block in the dtypes the device takes (the reference's loader hands over float16 samples AND
        float16 labels, grain_loader.py:88); returns its (audio, labels) views: ``prefetch(*pin(...))`` is one copy."""
        a, y = torch.as_tensor(audio), torch.as_tensor(labels)
        blk = InputBlock(a.shape, audio_dtype, y.shape, labels_dtype, pin=True)
        blk.audio.copy_(a)
        blk.labels.copy_(y)
        return blk.audio, blk.labels

    def prefetch(self, host_audio: torch.Tensor, host_labels: torch.Tensor):
        if self.pending >= (1 if self.backend is not None else 2):
            raise RuntimeError("the staging pairs hold batches that have not been taken")
        i = self.up
        self.up ^= 1
        if self.consumed[i] is not None:
            self.consumed[i].synchronize()               # HOST wait (see the class comment); recorded a step ago
        src = getattr(host_audio, "_amt_block", None)
        dst = self.stage[i]
        with torch.cuda.stream(self.copy_stream):
            if src is not None and src is getattr(host_labels, "_amt_block", None) and isinstance(dst, InputBlock) and dst.same_layout(src):
                dst.buf.copy_(src.buf, non_blocking=True)        # audio and labels: one contiguous transfer
            else:
                dst[0].copy_(host_audio, non_blocking=True)
                dst[1].copy_(host_labels, non_blocking=True)
            self.ready[i].record(self.copy_stream)
        self.pending += 1

    def prefetch_from_host(self, audio, labels):
        """The same for an arbitrary host batch (numpy / pageable torch): converted into one of two persistent pinned pairs
        first (no pinned allocation per step); a pair is rewritten only after the upload that read it has finished."""
        if self._ring is None:
            self._ring = [InputBlock(self.audio.shape, self.audio.dtype, self.labels.shape, self.labels.dtype, pin=True)
                          for _ in range(2)]
            self._ring_ev = [None, None]
        k = self._slot
        self._slot ^= 1
        if self._ring_ev[k] is not None:
            self._ring_ev[k].synchronize()
        ha, hy = self._ring[k]
        ha.copy_(torch.as_tensor(audio).reshape(ha.shape))
        hy.copy_(torch.as_tensor(labels).reshape(hy.shape))
        self.prefetch(ha, hy)
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        self._ring_ev[k] = ev

    def take(self):
        """In front of a step, on the compute stream: the step issued next reads the batch uploaded last."""
        if not self.pending:
            return
        i = self.down
        self.down ^= 1
        cur = torch.cuda.current_stream(self.audio.device)
        if self.backend is not None:
            # everything issued so far - the last step that read the OTHER set included - is in front of this event: once
            # it has completed, that set may be overwritten by the upload after next
            j = self.backend.active
            if self.consumed[j] is None:
                self.consumed[j] = torch.cuda.Event()
            self.consumed[j].record(cur)
            self.ready[i].synchronize()      # HOST wait for the upload (issued a step ago, 0.13 ms long): no stream waits at all
            self.backend.use_set(i)
        else:
            self.ready[i].synchronize()
            self.audio.copy_(self.stage[i][0], non_blocking=True)
            self.labels.copy_(self.stage[i][1], non_blocking=True)
            if self.consumed[i] is None:
                self.consumed[i] = torch.cuda.Event()
            self.consumed[i].record(cur)
        self.pending -= 1


class DataParallelStep:
    """One optimizer step of the data-parallel loop (train.py:259-332 with the GSPMD all-reduce made explicit).

    ``overlap=True``: the collective of the range the first half of the backward pass completes (37 of the 46 MB at the
    reference configuration, with the loss behind it) runs on a side stream while the CNN backward executes, the rest
    follows on the main stream: two collectives per step.  With the in-place protocol on a GPU the step stays ONE launch
    sequence that publishes the first half's completion to the host (``signal``, include/amt.h: amt_first_half_wait), which
    then issues the first collective; otherwise it runs in two host calls (amt_forward_backward_phase).  ``overlap=False``: one collective over the whole bucket after the backward pass.
    The reduction is a MEAN over ranks: ``ReduceOp.AVG`` where the backend has it (RCCL), SUM followed by a scale
    of the same range otherwise (gloo).  On CUDA/HIP devices the per-rank work is captured into hipGraphs unless
    ``graph=False``; the dropout seed of a replay is ``seed + seed_word`` with the device word advanced inside the
    graph, so replays draw fresh masks."""

    def __init__(self, backend, bucket: GradBucket, split: int, dist=None, overlap: bool = False, graph: bool = True,
                 lr_base: float = 1e-4, signal: Optional[bool] = None, wire: str = "fp32"):
        self.backend, self.bucket, self.split, self.dist = backend, bucket, int(split), dist
        self.world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
        # in-place protocol (HipTrainBackend(fused=True) over GradBucket.of_model): the collectives are SUMs over the
        # gradient sums and the loss word on the context's own accumulator; 1 / world goes into the optimizer
        self.in_place = bool(getattr(backend, "in_place", False))
        if hasattr(backend, "world"):
            backend.world = self.world
        # overlap without splitting the step: one launch sequence + a device-side signal (HipTrainBackend.full_signal)
        # (the default where it is available; signal=False forces the two-call form: 3.05 against 3.15 ms per step at the
        # reference configuration with no collective running - a stream waiting for an event of the busy compute stream
        # costs 0.12 ms on this platform, include/amt.h, profiles/r04_notes.md)
        can_signal = bool(overlap) and self.in_place and hasattr(backend, "full_signal") and bucket.buf.is_cuda
        if signal and not can_signal:
            raise ValueError("the signalled overlap form needs overlap=True and the in-place protocol on a GPU")
        self.signal = can_signal if signal is None else bool(signal)
        if getattr(backend, "fused", False) and self.world > 1 and not self.in_place:
            raise ValueError("more than one rank with the fused optimizer needs the model's own bucket: GradBucket.of_model(model)")
        # the optimizer gate tests the MEAN of the ranks' 0/1 flags against 0.999 (csrc/head.hip: adamw_gate_open): one
        # bad rank among N gives (N - 1) / N, which must stay below the threshold
        if self.world >= 1000:
            raise ValueError("the device-side optimizer gate (mean flag >= 0.999) supports fewer than 1000 ranks")
        self.overlap = bool(overlap)
        # wire = "bf16": the gradients cross the links as bfloat16 - 23.2 MB per rank and direction, the size of the
        # reference's own message (XLA reduces the fp16 outputs of the backward pass, /root/reference/train.py:292-297 casts
        # AFTER the reduction point) - with fp32 ACCUMULATION ON ARRIVAL: an all-to-all of bf16 shards (every rank receives
        # shard r of every rank), a local fp32 sum, an all-gather of the reduced shards as bf16.  On a fully connected xGMI
        # node both steps use all seven links at once and move 2 x 7/8 x 23.2 MB per rank against 2 x 7/8 x 46.4 MB for the
        # fp32 all-reduce.  The loss word travels in fp32 (a 4-byte all-reduce).  Opt-in; "fp32" (one all-reduce per range)
        # is the default.  NOT measured on hardware (no multi-GPU node in rounds 1-4); covered under gloo.
        if wire not in ("fp32", "bf16"):
            raise ValueError("wire must be 'fp32' or 'bf16'")
        self.wire = wire
        # the first real --gpus N run must not be able to hang or to re-exec: prepare() runs ONE trial step of the chosen form
        # (collectives included, its gradients discarded) when there is more than one rank; a signal time-out or a collective
        # error on ANY rank makes EVERY rank fall back to the one-collective in-place form (overlap = False) in this process
        self.fallback = None              # None, or the reason the overlapped form was given up
        self.signal_timeout_ms = 10000
        self._signal_failed = False
        self._inject_failure = None       # tests: "signal" | "collective" makes this rank's trial step fail that way
        self._wire_buf = {}
        self.lr_base = lr_base
        self.cuda = bucket.buf.is_cuda
        self.graph = bool(graph) and self.cuda
        self.step_no = 0
        self.applied = 0          # optimizer updates actually applied (device count, refreshed by result())
        self.optimizer_timeouts = 0   # barrier time-outs of the one-launch optimizer (no step is lost to one; refreshed by result())
        self._graphs = None
        self._graph_sets = []
        self._half_done = None
        self._gen = None
        self._comm = torch.cuda.Stream(device=bucket.buf.device) if self.cuda else None
        if self.overlap and self.world > 1 and not (0 < self.split < bucket.n):
            raise ValueError("overlap needs 0 < split < n_params")
        self._avg = None
        if self.dist is not None and self.world >= 1 and self.dist.is_initialized():
            self._avg = self.dist.ReduceOp.AVG if self.dist.get_backend() == "nccl" else None

    # ---- collectives -------------------------------------------------------------------------------
    def _reduce_bf16_wire(self, t: torch.Tensor, mean: bool):
        """SUM (or mean) of the fp32 range ``t`` over the ranks, in place, with bfloat16 on the wire and fp32 accumulation
        on arrival (see ``wire`` in __init__)."""
        W, n = self.world, t.numel()
        m = (n + W - 1) // W
        key = (t.data_ptr(), n)
        if key not in self._wire_buf:
            self._wire_buf[key] = (torch.zeros(m * W, dtype=torch.bfloat16, device=t.device),
                                   torch.empty(m * W, dtype=torch.bfloat16, device=t.device),
                                   torch.empty(m * W, dtype=torch.bfloat16, device=t.device))
        send, recv, out = self._wire_buf[key]
        send[:n].copy_(t)                                              # fp32 -> bf16 (round to nearest even)
        self.dist.all_to_all_single(recv, send)                        # recv[k m : (k + 1) m] = rank k's copy of MY shard
        red = recv.view(W, m).to(torch.float32).sum(0)                 # fp32 accumulation on arrival
        if mean:
            red.mul_(1.0 / W)
        self.dist.all_gather_into_tensor(out, red.to(torch.bfloat16))  # every rank gets every reduced shard (identical bits)
        t.copy_(out[:n])

    def _mean_allreduce(self, t: torch.Tensor):
        if self.dist is None or not self.dist.is_initialized():
            return
        if self.wire == "bf16" and self.world > 1:
            # the control / loss words of the range keep fp32: they sit at its ends (GradBucket layout)
            b = self.bucket
            lo, hi = t.data_ptr(), t.data_ptr() + 4 * t.numel()
            g0, g1 = b.grads.data_ptr(), b.grads.data_ptr() + 4 * b.n
            i0, i1 = max(lo, g0), min(hi, g1)
            head, body, tail = t[:(i0 - lo) // 4], t[(i0 - lo) // 4:(i1 - lo) // 4], t[(i1 - lo) // 4:]
            for ctl in (head, tail):
                if ctl.numel():
                    self.dist.all_reduce(ctl, op=self.dist.ReduceOp.SUM)
                    if not self.in_place:
                        ctl.mul_(1.0 / self.world)
            if body.numel():
                self._reduce_bf16_wire(body, mean=not self.in_place)
            return
        if self.in_place:          # raw sums: 1 / world is applied by the optimizer (and by result() to the loss)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            return
        if self._avg is not None:
            self.dist.all_reduce(t, op=self._avg)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
            t.mul_(1.0 / self.world)

    # what a collective covers.  Legacy protocol: the control words travel with the gradients.  In place: the gradient sums
    # and the loss word only (the flags are outputs of the optimizer, identical on every rank by construction).
    def _whole_range(self):
        b = self.bucket
        return b.buf[b.HEAD:b.HEAD + b.n + 1] if self.in_place else b.buf

    def _first_range(self):
        b = self.bucket
        return b.buf[b.HEAD + self.split:b.HEAD + b.n + 1] if self.in_place else b.first_range(self.split)

    def _second_range(self):
        b = self.bucket
        return b.buf[b.HEAD:b.HEAD + self.split] if self.in_place else b.second_range(self.split)

    # ---- capture -----------------------------------------------------------------------------------
    def prepare(self, seed: int = 0, trial: Optional[bool] = None):
        """Run once eagerly (plans the workspace: every allocation happens here), capture the graphs, and - with more than
        one rank (or ``trial=True``) - run one trial step of the overlapped form; see ``fallback``."""
        self._prepare_graphs(seed)
        if trial is None:
            trial = self.world > 1 and (self.overlap or self.signal)
        if trial and (self.overlap or self.signal):
            reason = self._trial_step(seed)
            if reason is not None:
                self._fall_back(reason, seed)

    def _trial_step(self, seed: int) -> Optional[str]:
        """One step of the chosen form whose gradients are discarded; returns None or why it failed ON SOME RANK."""
        be = self.backend
        why = None
        try:
            self._issue(seed, apply=False)
            if self.cuda:
                torch.cuda.synchronize()
            if self._signal_failed:
                why = "first-half signal did not arrive within %d ms" % self.signal_timeout_ms
            elif self._inject_failure == "collective":      # (tests: reported after taking part in every collective - peers must not hang)
                why = "collective error: injected"
        except RuntimeError as e:      # a collective refused the call (in-place range, stream, backend)
            why = "collective error: " + str(e).splitlines()[0][:160]
        self._inject_failure = None
        getattr(be, "discard", lambda: None)()
        if hasattr(be, "sync_signals") and self.signal:
            be.sync_signals()
        # every rank must take the same decision: MAX of the failure flags over a plain all-reduce of one word
        if self.dist is not None and self.dist.is_initialized() and self.world > 1:
            flag = torch.tensor([1.0 if why else 0.0], device=self.bucket.buf.device if self.dist.get_backend() == "nccl" else "cpu")
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if float(flag.item()) > 0 and why is None:
                why = "another rank's trial step failed"
        self.step_no = 0
        self._signal_failed = False
        return why

    def _fall_back(self, reason: str, seed: int):
        self.fallback = reason
        self.overlap, self.signal = False, False
        self._graphs, self._graph_sets = None, []
        self._prepare_graphs(seed)

    def _prepare_graphs(self, seed: int = 0):
        be = self.backend
        discard = getattr(be, "discard", lambda: None)      # fused protocol: rehearsal gradients must not reach the optimizer
        be.full(seed)
        discard()
        if self.signal:
            be.full_signal(seed)
            discard()
            be.sync_signals()      # the rehearsal's signal has no wait: declared consumed
        elif self.overlap:
            be.half(1, seed); be.half(2, seed)
            discard()
        if self.cuda:
            torch.cuda.synchronize()
        if not self.graph:
            return
        side = torch.cuda.Stream(device=self.bucket.buf.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up on a side stream, as torch's capture rules want
            be.full(seed)
            discard()
            if self.signal:
                be.full_signal(seed)
                discard()
                be.sync_signals()
            elif self.overlap:
                be.half(1, seed); be.half(2, seed)
                discard()
        torch.cuda.current_stream().wait_stream(side)
        # one capture per input set of the backend (HipTrainBackend.add_input_set): the same launches on other input pointers
        nsets = len(getattr(be, "_sets", [None]))
        active = getattr(be, "active", 0)
        self._graph_sets = []
        for i in range(nsets):
            if nsets > 1:
                be.use_set(i)
            self._graph_sets.append(self._capture(seed))
        if nsets > 1:
            be.use_set(active)
        self._graphs = self._graph_sets[active if nsets > 1 else 0]
        self._gen = be.generation() if hasattr(be, "generation") else None

    def _capture(self, seed: int):
        be = self.backend
        if self.signal:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                be.advance_seed()
                be.full_signal(seed)
            return (g,)
        if self.overlap:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                be.advance_seed()
                be.half(1, seed)
            with torch.cuda.graph(gb, pool=ga.pool()):
                be.half(2, seed)
            return (ga, gb)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            be.advance_seed()
            be.full(seed)
        return (g,)

    def _check_generation(self):
        if self._graphs is not None and self._gen is not None and self.backend.generation() != self._gen:
            raise RuntimeError("the workspace was re-planned after capture (amt_plan_generation changed): "
                               "the captured graphs are stale, call prepare() again")

    # ---- one step ----------------------------------------------------------------------------------
    def step(self, seed: Optional[int] = None):
        """Issue one step (asynchronous on a device).  ``seed`` is used by eager steps; graph replays advance the
        device seed word instead."""
        self.step_no += 1
        self._issue(self.step_no if seed is None else seed, apply=True)

    def _issue(self, seed: int, apply: bool):
        be, b = self.backend, self.bucket
        self._check_generation()
        nsets = len(getattr(be, "_sets", [None]))
        if self._graphs is not None and nsets != max(1, len(self._graph_sets)):
            # (InputFeeder(backend=...) adds a second input set: constructed after prepare() its uploads would land in a set
            # no captured graph reads, and every other batch would silently be the stale one)
            raise RuntimeError("the backend has %d input sets but %d were captured: construct InputFeeder(backend=...) "
                               "BEFORE DataParallelStep.prepare()" % (nsets, max(1, len(self._graph_sets))))
        if self._graphs is not None and len(self._graph_sets) > 1:
            self._graphs = self._graph_sets[be.active]      # the graph captured on the input set this step reads
        if self.signal:
            # ONE graph (the single-rank step's launch sequence).  The host waits for the signal the graph publishes behind
            # its first CNN Block chain (mapped host memory), then issues the collective of [split, n) + loss on the
            # communication stream - which therefore needs no dependency on the compute stream - while the CNN backward
            # runs; the CNN range follows on the compute stream, which finally waits for the communication stream
            if self._graphs is not None:
                self._graphs[0].replay()
            else:
                be.full_signal(seed)
            arrived = self._inject_failure != "signal" and be.wait_first_half(self.signal_timeout_ms)
            if not arrived:
                # no signal: the host has waited for the whole launch sequence instead (wait_first_half synchronised), so the
                # collectives below are still correct - issued anyway, or the peers would sit in theirs - and the failure is
                # recorded: prepare() turns it into the fall-back, a later step into an error
                if self._inject_failure == "signal":
                    torch.cuda.current_stream().synchronize()
                    be.sync_signals()
                self._signal_failed = True
            with torch.cuda.stream(self._comm):
                self._mean_allreduce(self._first_range())
            self._mean_allreduce(self._second_range())
            torch.cuda.current_stream().wait_stream(self._comm)
        elif self.overlap:
            if self._graphs is not None:
                self._graphs[0].replay()
            else:
                be.half(1, seed)
            if self.cuda:
                # The second half is issued FIRST; the host then waits for an event recorded between the halves and issues
                # the first collective on the communication stream.  (The stream-side form - the communication stream waiting
                # for that event - costs 0.12 ms per step on this platform: any stream waiting for an event of the busy
                # compute stream does; profiles/r04_notes.md.)
                if self._half_done is None:
                    self._half_done = torch.cuda.Event()
                self._half_done.record(torch.cuda.current_stream())
                if self._graphs is not None:
                    self._graphs[1].replay()
                else:
                    be.half(2, seed)
                self._half_done.synchronize()
                with torch.cuda.stream(self._comm):
                    self._mean_allreduce(self._first_range())           # transformer + decoder range, loss, first flag
                self._mean_allreduce(self._second_range())               # whole-step flag + CNN range
                torch.cuda.current_stream().wait_stream(self._comm)
            else:
                self._mean_allreduce(self._first_range())
                if self._graphs is not None:
                    self._graphs[1].replay()
                else:
                    be.half(2, seed)
                self._mean_allreduce(self._second_range())
        else:
            if self._graphs is not None:
                self._graphs[0].replay()
            else:
                be.full(seed)
            if self.world > 1:
                self._mean_allreduce(self._whole_range())                 # ONE bucket: flag, gradients, loss, flag
        if apply:
            if self._signal_failed:
                self._signal_failed = False
                raise RuntimeError("the first-half signal of a step did not arrive within %d ms (the step itself completed: the "
                                   "host synchronised instead); run with signal=False / overlap=False" % self.signal_timeout_ms)
            be.apply(self.step_no, self.lr_base)

    def result(self) -> Tuple[float, bool]:
        """(loss, grads_valid) of the last step; synchronises.  grads_valid is the AND over ranks and halves (and is 0 for
        a non-finite loss: amt_pack_tail folds isfinite(loss) into the flags, train.py:369).  At this sync point the
        host's step count is re-aligned with the device's count of APPLIED updates, so the learning-rate schedule
        (``schedule(applied)``) does not run ahead after skipped steps."""
        v = self.bucket.buf[[0, self.bucket.HEAD + self.bucket.n, self.bucket.HEAD + self.bucket.n + 1]].tolist()
        v[1] /= self.bucket.loss_div
        valid = v[0] >= 0.999 and v[2] >= 0.999
        if hasattr(self.backend, "applied_steps"):
            self.applied = int(self.backend.applied_steps())
        m = getattr(self.backend, "model", None)
        # a barrier time-out of the one-launch optimizer (something else held compute units): that step was committed by
        # one workgroup, nothing was lost, and the context has taken the two launches since; reported, not acted upon
        self.optimizer_timeouts = int(m.optimizer_status()) if (m is not None and hasattr(m, "optimizer_status")) else 0
        if m is not None and hasattr(m, "team_status") and m.team_status() != 0:     # polled where the loop synchronises anyway
            raise RuntimeError("an in-launch XCD-team hand-off timed out (amt_team_status != 0): the outputs of that step are "
                               "undefined; the context has switched its CNN stages back to one launch per Block")
        return float(v[1]), bool(valid)


# ======================================================================================================
# train(): the reference's training loop (train.py:211-452) over the pieces above.
# ======================================================================================================
def train(model, data_loader, *, num_steps: int, base_learning_rate: float = 1e-4, warmup_steps: int = 1000,
          layer_lr_decay: float = 0.7, weight_decay: float = 0.005, dist=None, overlap: Optional[bool] = None,
          accum_steps: int = 1, loss_scaling: bool = False, grad_scale: float = 1.0, check_every: Optional[int] = None,
          checkpoint_dir=None, checkpoint_every: int = 20, checkpoints_to_keep: int = 3, print_every: int = 1000,
          seed: int = 1234, log: Callable[[str], None] = print, resume: bool = False, save_optimizer: bool = True):
    """``train(...)`` of the reference (train.py:211-452, called from main at :872) on the HIP path.

    ``data_loader`` yields ``(events, audio)`` host batches - the reference's order (train.py:355) -: events
    ``[B, frames, 90]`` soft labels, audio ``[B, 2, T]``; every batch has the shape of the first one (the loader crops / pads
    to the batch, grain_loader.py:60-88).  ``B`` is the RANK's batch; ``accum_steps`` minibatches of it are accumulated on
    the device (train.py:275-310).  What the reference's loop does, and where it lives here:

    * learning rate (train.py:455-466): ``create_learning_rate_schedule(base, warmup_steps, num_steps)`` evaluated at the
      count of APPLIED updates (optax's count; a skipped step does not advance it) -> ``DataParallelStep.lr_base``;
    * the step (train.py:259-332): ``DataParallelStep`` over ``HipTrainBackend(fused=True)`` on the model's own bucket:
      forward, loss, backward, (collectives,) AdamW + clip, gated on the device - one hipGraph replay + one optimizer launch;
    * a fresh batch per step (train.py:280): ``InputFeeder`` (pinned host memory, copy stream, double-buffered);
    * non-finite step (train.py:369-377): the device gate leaves parameters, m, v and the update count untouched, so there is
      nothing to roll back; with ``loss_scaling`` the scale is halved (and doubled while scaled_loss < 1e4, train.py:379-382:
      ``LossScaleController``), which re-captures the step - bf16 needs no scaling, so it is off by default and the scale
      stays ``grad_scale``.  A barrier time-out of the one-launch optimizer (``DataParallelStep.optimizer_timeouts``) loses
      nothing - that step is committed by one workgroup - and is only logged;
    * checkpoints (train.py:384-394, 799-812): every ``checkpoint_every`` steps the flat weight file + metadata
      (``infer.save_checkpoint``), the newest ``checkpoints_to_keep`` are kept; rank 0 only;
    * ``check_every``: how often the host synchronises to read (loss, valid) - the reference does every step (it decides on
      the host); with the device gate that is only needed when ``loss_scaling`` is on.  Default: 1 with ``loss_scaling``,
      otherwise the smaller of ``print_every`` and ``checkpoint_every`` (the step stays asynchronous in between);
    * ``resume`` (train.py:799-831: the reference restores model AND optimizer state through its checkpoint manager, and
      ``train(start_step=...)`` continues the count): with a checkpoint in ``checkpoint_dir`` the parameters, AdamW's moments
      and the count of applied updates are restored and the loop continues at the checkpoint's step + 1 - schedule and bias
      corrections go on where they stopped.  A checkpoint without optimizer state (``save_optimizer=False``, or written by
      ``infer.save_checkpoint`` alone) restores the parameters and the step; the moments then restart from zero (logged).

    Out of scope, as in SURVEY.md section 2: test-set evaluation, TensorBoard, the (no-op) genetic recombination.
    Returns a dict: steps issued, updates applied, skipped steps, final scale, per-check (step, loss) history."""
    import itertools
    import os as _os
    from . import infer as _infer
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    if overlap is None:
        overlap = world > 1
    it = iter(data_loader)
    try:
        events0, audio0 = next(it)
    except StopIteration:
        raise ValueError("the data loader is empty") from None
    dev = model.device
    audio_dev = torch.as_tensor(audio0).to(device=dev, dtype=model.audio_dtype).contiguous()
    ev0 = torch.as_tensor(events0)     # float16 labels (the reference's batches, grain_loader.py:88) stay float16 on the device
    labels_dev = ev0.to(device=dev, dtype=torch.float16 if ev0.dtype == torch.float16 else torch.float32).contiguous()
    model.init_optimizer(layer_lr_decay)
    if check_every is None:
        cands = [c for c in (print_every, checkpoint_every if checkpoint_dir is not None else 0) if c and c > 0]
        check_every = 1 if loss_scaling else (min(cands) if cands else max(1, num_steps))
    start_step = 0
    if resume and checkpoint_dir is not None and _os.path.isdir(str(checkpoint_dir)):
        try:
            flat, _conf, ck_step, opt = _infer.load_checkpoint(str(checkpoint_dir), with_optimizer=True)
        except FileNotFoundError:
            flat = None
        if flat is not None:
            model.load_flat(flat)
            start_step = ck_step
            if opt is not None:
                model.load_optimizer_state(opt[0], opt[1], opt[2])
                log(f"Resumed from step {ck_step}: parameters, AdamW moments and {opt[2]} applied updates restored")
            else:
                log(f"Resumed from step {ck_step}: parameters restored; no optimizer state in the checkpoint - moments and "
                    "the update count restart from zero")
    bucket = GradBucket.of_model(model)
    controller = LossScaleController(scale=float(grad_scale), enabled=bool(loss_scaling))
    schedule = create_learning_rate_schedule(base_learning_rate, warmup_steps, num_steps)
    hyper = dict(layer_lr_decay=layer_lr_decay, weight_decay=weight_decay)

    def build(scale, old=None):
        be = HipTrainBackend(model, audio_dev, labels_dev, bucket, scale=scale, fused=True, accum_steps=accum_steps,
                             optimizer_args=hyper)
        if old is None:
            be.add_input_set()             # two input sets, the step captured on each: uploads land where a step reads them
        else:                              # a re-capture (new loss scale) keeps the input tensors and the one in use
            be._sets = old._sets
            be.use_set(old.active)
        step = DataParallelStep(be, bucket, be.split, dist=dist, overlap=overlap, graph=True, lr_base=schedule(0))
        step.prepare(seed=seed)
        return be, step

    be, dp = build(controller.scale)
    feeder = InputFeeder(backend=be)
    history, skipped, issued, timeouts_seen = [], 0, 0, 0
    loss_sum, loss_n = 0.0, 0
    saved = []
    applied = model.optimizer_count()
    have_batch = True                      # batch 1 is already in the input tensors
    for step_no in range(start_step + 1, num_steps + 1):
        if not have_batch:
            break
        feeder.take()                      # (batch k + 1 was sent on its way while step k was being issued)
        dp.lr_base = schedule(applied)     # optax evaluates the schedule at its count BEFORE the increment
        dp.step(seed=seed + step_no)
        issued += 1
        try:
            ev_h, au_h = next(it)
            feeder.prefetch_from_host(au_h, ev_h)
        except StopIteration:
            have_batch = False
        applied += 1                       # optimistic; re-aligned with the device count at every check
        if step_no % check_every == 0 or step_no == num_steps:
            loss, valid = dp.result()      # synchronises; dp.applied = the device's count of applied updates
            applied = dp.applied
            new_timeouts = dp.optimizer_timeouts - timeouts_seen
            timeouts_seen = dp.optimizer_timeouts
            if new_timeouts:
                log(f"step {step_no}: the one-launch optimizer timed out at its grid barrier (step applied; two launches from now on)")
            if controller.update(valid, loss, loss * controller.scale):
                skipped += 1
                log(f"Encountered NAN/inf loss at step {step_no}, loss = {loss}. Skipped on the device; grad scale -> {controller.scale}")
                if controller.enabled:
                    be, dp = build(controller.scale, be)   # the scale is an argument of the captured launches
                    feeder.backend = be
                continue
            elif controller.enabled and be.scale != controller.scale:
                log(f"Grad scale: {be.scale} -> {controller.scale}")
                be, dp = build(controller.scale, be)
                feeder.backend = be
            if valid:
                loss_sum += loss; loss_n += 1
                history.append((step_no, loss))
        if print_every and step_no % print_every == 0 and loss_n:
            log(f"Step {step_no}/{num_steps}, Loss: {loss_sum / loss_n}, LR = {schedule(applied)}")
            loss_sum, loss_n = 0.0, 0
        if checkpoint_dir is not None and rank == 0 and checkpoint_every and step_no % checkpoint_every == 0:
            torch.cuda.synchronize()
            opt = None
            if save_optimizer:
                om, ov, oc = model.optimizer_state()
                opt = (om.cpu().numpy(), ov.cpu().numpy(), oc)
            _infer.save_checkpoint(checkpoint_dir, model.weights_tensor().cpu().numpy(), model.conf, step=step_no, optimizer=opt)
            saved.append(step_no)
            while len(saved) > checkpoints_to_keep:
                old = saved.pop(0)
                for name in (f"metadata_{old}.json", f"params_{old}.f32", f"opt_{old}.f32"):     # metadata first: never half a checkpoint
                    try:
                        _os.remove(_os.path.join(str(checkpoint_dir), name))
                    except OSError:
                        pass
    torch.cuda.synchronize()
    return {"steps": issued, "applied": model.optimizer_count(), "skipped": skipped, "grad_scale": controller.scale,
            "history": history, "checkpoints": saved, "start_step": start_step}
