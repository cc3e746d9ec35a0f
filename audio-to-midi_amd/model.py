"""Host-side mirror of the reference's model API for the hot path.

Mirrors, name for name:
  * ``model_config``                         model.py:20-34
  * ``OutputSequenceGenerator(conf, key)``   model.py:673-738
      ``__call__(samples, state, rope_freqs, key, enable_dropout) -> ((logits, probs), state)``
                                              model.py:740-769
      ``predict(state, samples, rope_freqs) -> (logits, probs)``   model.py:771-773
  * ``compute_loss`` (value + grads)         train.py:49-62   -> ``loss_and_grads``

The reference calls the model under ``jax.vmap`` over the leading batch/window axis
(train.py:56-58, infer.py:40); here the batch axis is part of the call: ``samples`` is
``[B, 2, T]`` and outputs are ``[B, S, 90]``.  All compute happens in libamt_hip.so (HIP, gfx950);
torch is used for device memory and streams only.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import numpy as np
import torch

from . import lib as _lib
from . import params as _params
from .rope import RopeFreqs, precompute_frequencies

# model.py:20-34
model_config = {
    "dims": [4 * (2 ** i) for i in range(7)],
    "depths": [3, 3, 3, 3, 3, 21, 3],
    "cnn_hidden_expansion": 2.0,
    "num_transformer_layers": 8,
    "num_transformer_heads": 4,
    "attention_size": 64,
    "compressed_attention_q_size": 64,
    "compressed_attention_kv_size": 64,
    "transformer_dropout_rate": 0.1,
    "transformer_hidden_expansion": 2.0,
    "sdd_rate": 0.1,
}

_TORCH_DTYPE = {"float32": torch.float32, "bfloat16": torch.bfloat16}
# dtype of the audio a context takes (include/amt.h, Conventions): bf16 contexts read IEEE half samples - what the reference
# feeds its model (grain_loader.py:88) - so the input is not quantised a second time
_AUDIO_DTYPE = {"float32": torch.float32, "bfloat16": torch.float16}


class StepOutput(NamedTuple):
    loss: float                 # mean over the batch of the summed BCE (train.py:62, /grad_scale)
    scaled_loss: float          # loss * scale
    grads_flat: torch.Tensor    # fp32 [n_params], device layout, already divided by scale
    grads_valid: bool           # train.py:320-322
    logits: Optional[torch.Tensor]


class OutputSequenceGenerator:
    """The reference's model class, backed by the HIP library.

    ``key`` (a JAX PRNG key in the reference) is an integer seed for the random init here;
    pass ``key=None`` to start from zeros and load weights explicitly.
    """

    def __init__(self, conf=None, key: Optional[int] = None, *, dtype: str = "bfloat16", device="cuda:0",
                 gamma: float = 1e-6):
        conf = dict(model_config if conf is None else conf)
        if dtype not in _TORCH_DTYPE:
            raise ValueError("dtype must be 'float32' or 'bfloat16'")
        self.conf = conf
        self.dtype = dtype
        self.torch_dtype = _TORCH_DTYPE[dtype]
        self.audio_dtype = _AUDIO_DTYPE[dtype]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("OutputSequenceGenerator runs on an AMD GPU only (device must be cuda:N)")
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the HIP hot path has no CPU fallback")
        self._L = _lib.load()
        self._cfg = _lib.make_config(conf, dtype)
        self.n_params = int(self._L.amt_param_count(self._cfg))
        if self.n_params == 0:
            raise ValueError("configuration rejected by libamt_hip (see include/amt.h for constraints)")
        self.n_events = int(self._cfg.n_events)
        h = C.c_void_p()
        torch.cuda.set_device(self.device)
        if self._L.amt_create(self._cfg, self.device.index or 0, C.byref(h)) != 0:
            raise RuntimeError("amt_create failed: " + self._L.amt_last_error(None).decode())
        self._h = h
        self._rope_key = None
        if key is not None:
            self.load_equinox_params(_params.init_equinox(conf, seed=int(key), gamma=gamma))

    # ------------------------------------------------------------------ lifetime / errors
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.amt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self._L.amt_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ parameters
    def load_flat(self, flat):
        """flat fp32 parameters in device layout (numpy array or torch tensor)."""
        if isinstance(flat, np.ndarray):
            flat = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32))
        flat = flat.to(torch.float32).contiguous()
        if flat.numel() != self.n_params:
            raise ValueError(f"expected {self.n_params} parameters, got {flat.numel()}")
        on_dev = 1 if flat.is_cuda else 0
        self._check(self._L.amt_load_weights(self._h, _lib.ptr(flat), flat.numel(), on_dev, self._stream()),
                    "amt_load_weights")
        torch.cuda.current_stream(self.device).synchronize()

    def load_equinox_params(self, p):
        """Parameters as the reference's pytree (dict of Equinox-shaped leaves)."""
        self.load_flat(_params.equinox_to_flat(p, self.conf))

    def grads_to_equinox(self, flat):
        return _params.flat_to_equinox(np.asarray(flat), self.conf)

    def weights_tensor(self) -> torch.Tensor:
        """Zero-copy fp32 view of the context's master parameters (device layout)."""
        ptr = self._L.amt_weights_ptr(self._h)
        return _DevView(ptr, self.n_params, self.device).tensor()

    def refresh_weights(self):
        self._check(self._L.amt_refresh_weights(self._h, self._stream()), "amt_refresh_weights")

    def set_rope(self, rope_freqs: RopeFreqs):
        cos = np.ascontiguousarray(rope_freqs.cos_freq, dtype=np.float32)
        sin = np.ascontiguousarray(rope_freqs.sin_freq, dtype=np.float32)
        if cos.shape != sin.shape or cos.ndim != 2 or cos.shape[1] != self._cfg.head_dim // 2:
            raise ValueError("rope_freqs must hold (max_pos, head_dim/2) cos/sin tables")
        key = (cos.shape[0], float(cos[-1, -1]), float(sin[-1, 0]))
        if key == self._rope_key:
            return
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self._L.amt_set_rope(self._h, cos.ctypes.data_as(C.c_void_p), sin.ctypes.data_as(C.c_void_p),
                                         cos.shape[0]), "amt_set_rope")
        self._rope_key = key

    def set_attn_fp8(self, enable: bool = True):
        """Global-attention forward on fp8 (e4m3) MFMA with delayed per-tensor scaling (include/amt.h:
        amt_set_attn_fp8; BASELINE configs[4]).  bf16 models only; not part of the reference (model.py:333-386
        computes attention in the activation dtype), so it is off unless asked for."""
        self._check(self._L.amt_set_attn_fp8(self._h, 1 if enable else 0), "amt_set_attn_fp8")

    # ------------------------------------------------------------------ shape checks
    def _prep_audio(self, samples):
        if not torch.is_tensor(samples):
            samples = torch.as_tensor(np.asarray(samples))
        if samples.dim() == 2:
            samples = samples[None]
        if samples.dim() != 3 or samples.shape[1] != 2:
            raise ValueError(f"samples must be [B, 2, T] (stereo), got {tuple(samples.shape)}")
        T = samples.shape[2]
        if T % 320 != 0 or T // 320 < 16:
            raise ValueError("T must be a multiple of 320 samples and at least 16 frames long")
        return samples.to(device=self.device, dtype=self.audio_dtype).contiguous()

    def _ensure_rope(self, rope_freqs, S):
        if rope_freqs is not None:
            if rope_freqs.cos_freq.shape[0] < S:
                # the reference would fail to broadcast here (rope.py:40-41); refuse as well
                raise ValueError(f"rope_freqs has {rope_freqs.cos_freq.shape[0]} positions but the clip has {S} frames")
            self.set_rope(rope_freqs)

    # ------------------------------------------------------------------ inference
    def predict(self, state, samples, rope_freqs: Optional[RopeFreqs] = None):
        """model.predict(state, samples, rope_freqs) under vmap -> (logits, probs), fp32 [B,S,90]."""
        del state
        a = self._prep_audio(samples)
        B, _, T = a.shape
        S = T // 320
        self._ensure_rope(rope_freqs, S)
        logits = torch.empty(B, S, self.n_events, dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        self._check(self._L.amt_forward(self._h, _lib.ptr(a), B, T, _lib.ptr(logits), _lib.ptr(probs),
                                        self._stream()), "amt_forward")
        return logits, probs

    def __call__(self, samples, state=None, rope_freqs: Optional[RopeFreqs] = None, key=None,
                 enable_dropout: bool = False):
        """model.py:740-769 under vmap: ``((logits, probs), state)``.  ``enable_dropout=True`` is the forward pass of the
        training step (the call compute_loss makes, train.py:56-58): attention-weight and FFN-output dropout with masks
        drawn from ``key`` (an integer seed here; a JAX PRNG key in the reference) - the masks ``loss_and_grads(seed=key,
        train=True)`` draws, so its logits are the ones that loss is computed from.  The CNN's stochastic depth stays
        inactive in both modes, as in the reference (eqx.nn.Sequential forwards only ``key``, model.py:757-758, 160)."""
        if not enable_dropout:
            return self.predict(state, samples, rope_freqs), state
        a = self._prep_audio(samples)
        B, _, T = a.shape
        S = T // 320
        self._ensure_rope(rope_freqs, S)
        logits = torch.empty(B, S, self.n_events, dtype=torch.float32, device=self.device)
        probs = torch.empty_like(logits)
        self._check(self._L.amt_forward_train(self._h, _lib.ptr(a), B, T, int(0 if key is None else key), _lib.ptr(logits),
                                              _lib.ptr(probs), self._stream()), "amt_forward_train")
        return (logits, probs), state

    # ------------------------------------------------------------------ training forward + backward
    def loss_and_grads(self, audio, labels, scale: float = 1.0, seed: int = 0, train: bool = True,
                       rope_freqs: Optional[RopeFreqs] = None, return_logits: bool = False,
                       grads_out: Optional[torch.Tensor] = None, sync: bool = True, accum=(0, 1),
                       loss_out: Optional[torch.Tensor] = None, valid_out: Optional[torch.Tensor] = None,
                       fused_optimizer: bool = False):
        """compute_loss + the gradient half of compute_training_step (train.py:49-62, 292-322).

        ``accum=(i, k)``: minibatch i of k of an accumulated step (train.py:275-310): call i = 0 .. k - 1 in order with
        the same ``grads_out`` / ``loss_out`` / ``valid_out`` tensors; gradients are summed in the context's fp32
        accumulator and only call k - 1 writes ``grads_out = sum / (scale * k)``, the mean loss and the flag."""
        a = self._prep_audio(audio)
        B, _, T = a.shape
        S = T // 320
        # float16 labels go in as they are (the reference's batches: grain_loader.py:88), anything else as float32
        y = torch.as_tensor(labels)
        y = y.to(device=self.device, dtype=torch.float16 if y.dtype == torch.float16 else torch.float32).contiguous()
        self._label_dtype(y.dtype)
        if tuple(y.shape) != (B, S, self.n_events):
            raise ValueError(f"labels must be [{B}, {S}, {self.n_events}], got {tuple(y.shape)}")
        self._ensure_rope(rope_freqs, S)
        if fused_optimizer:      # the gradients stay in the context's accumulator for adamw_step(None, ...) (include/amt.h)
            if grads_out is not None:
                raise ValueError("fused_optimizer=True leaves the gradients in the context: pass no grads_out")
            grads = None
        else:
            grads = grads_out if grads_out is not None else torch.empty(self.n_params, dtype=torch.float32,
                                                                        device=self.device)
        loss = loss_out if loss_out is not None else torch.empty(2, dtype=torch.float32, device=self.device)
        valid = valid_out if valid_out is not None else torch.empty(1, dtype=torch.int32, device=self.device)
        logits = torch.empty(B, S, self.n_events, dtype=torch.float32, device=self.device) if return_logits else None
        ai, ak = int(accum[0]), int(accum[1])
        self._check(self._L.amt_forward_backward_accum(self._h, _lib.ptr(a), _lib.ptr(y), B, T, float(scale), int(seed),
                                                       1 if train else 0, _lib.ptr(loss), _lib.ptr(grads), _lib.ptr(valid),
                                                       _lib.ptr(logits), 0, ai, ak, self._stream()), "amt_forward_backward")
        if ai != ak - 1 or fused_optimizer:
            return loss, grads, valid, logits          # nothing has been written yet: the last minibatch finalises
        if not sync:
            return loss, grads, valid, logits
        lv = loss.cpu().numpy()
        return StepOutput(float(lv[0]), float(lv[1]), grads, bool(int(valid.item())), logits)

    def _label_dtype(self, dtype):
        """Tell the context the element type of the labels of the calls that follow (include/amt.h: amt_set_label_dtype)."""
        code = 2 if dtype == torch.float16 else 0          # AMT_F16 / AMT_F32
        if getattr(self, "_label_code", 0) != code:
            self._check(self._L.amt_set_label_dtype(self._h, code), "amt_set_label_dtype")
            self._label_code = code

    # ------------------------------------------------------------------ the same step in two halves (DP overlap)
    def grad_bucket_split(self) -> int:
        """First flat index of the gradients completed by ``loss_and_grads_phase(1)`` (decoder, transformer,
        final CNN norm); ``[0, split)`` (the CNN stages) is completed by phase 2."""
        return int(self._L.amt_grad_bucket_split(self._h))

    def loss_and_grads_phase(self, phase: int, audio, labels, grads_out: torch.Tensor, loss_out: torch.Tensor,
                             valid_out: torch.Tensor, scale: float = 1.0, seed: int = 0, train: bool = True, accum=(0, 1)):
        """Asynchronous half of ``loss_and_grads`` (include/amt.h: amt_forward_backward_phase).  All tensors are
        caller-owned device tensors (audio in ``model.audio_dtype``, labels float32 or float16); nothing is synchronised."""
        if int(phase) not in (1, 2, 3):
            raise RuntimeError("amt_forward_backward_phase failed: phase must be 1, 2 or 3")
        if int(phase) == 3 and grads_out is not None:
            raise ValueError("phase 3 (whole step + first-half signal) belongs to the fused protocol: grads_out must be None")
        if audio.dtype != self.audio_dtype or not audio.is_cuda or not audio.is_contiguous():
            raise ValueError(f"audio must be a contiguous device tensor of dtype {self.audio_dtype} (model.audio_dtype): "
                             "this entry point converts nothing")
        if labels.dtype not in (torch.float32, torch.float16) or not labels.is_cuda or not labels.is_contiguous():
            raise ValueError("labels must be a contiguous device tensor of dtype float32 or float16")
        self._label_dtype(labels.dtype)
        self._check(self._L.amt_forward_backward_accum(self._h, _lib.ptr(audio), _lib.ptr(labels), audio.shape[0],
                                                       audio.shape[2], float(scale), int(seed), 1 if train else 0,
                                                       _lib.ptr(loss_out), _lib.ptr(grads_out), _lib.ptr(valid_out), None,
                                                       int(phase), int(accum[0]), int(accum[1]), self._stream()),
                    "amt_forward_backward_phase")

    def init_optimizer(self, layer_lr_decay=0.7):
        """Allocate AdamW state (train.py:845 setup_optimizers).  Call before hipGraph capture."""
        self._check(self._L.amt_optimizer_init(self._h, float(layer_lr_decay)), "amt_optimizer_init")

    def adamw_step(self, grads_flat, lr_base, step, layer_lr_decay=0.7, b1=0.9, b2=0.999, eps=1e-3,
                   weight_decay=0.005, clip_norm=1.0, valid: Optional[torch.Tensor] = None,
                   valid_mean: Optional[torch.Tensor] = None):
        """optimizer half of compute_training_step (train.py:324-325, 646-729).  ``step`` counts from 1 (optax);
        ``step=-1`` takes the count of applied updates from the device (skipped steps do not advance it).

        ``valid`` (device int32[1], the grads_valid word of ``loss_and_grads``) and ``valid_mean`` (device
        float32[1], that flag after a mean all-reduce: 1.0 iff every rank's is 1) gate the update ON THE DEVICE:
        with a closed gate parameters, m and v stay untouched (the skip of train.py:369-377) and no host sync is
        needed, so the step stays hipGraph-capturable."""
        if valid is not None and (valid.dtype != torch.int32 or not valid.is_cuda):
            raise ValueError("valid must be a device int32 tensor")
        # grads_flat=None: the fused protocol - the gradients are what loss_and_grads(fused_optimizer=True) left in the
        # context; `valid` is then an OUTPUT (1 iff the update was applied)
        if valid_mean is not None and (valid_mean.dtype != torch.float32 or not valid_mean.is_cuda):
            raise ValueError("valid_mean must be a device float32 tensor")
        self._check(self._L.amt_adamw_step_gated(self._h, _lib.ptr(grads_flat), float(lr_base), float(layer_lr_decay),
                                                 float(b1), float(b2), float(eps), float(weight_decay), float(clip_norm),
                                                 int(step), _lib.ptr(valid), _lib.ptr(valid_mean), self._stream()),
                    "amt_adamw_step")

    def adamw_step_reduced(self, lr_base, step, grad_mul: float, loss_sum: torch.Tensor, layer_lr_decay=0.7, b1=0.9,
                           b2=0.999, eps=1e-3, weight_decay=0.005, clip_norm=1.0, valid: Optional[torch.Tensor] = None):
        """The fused protocol's optimizer behind a SUM all-reduce that ran on the context's own accumulator
        (include/amt.h: amt_adamw_step_reduced): reads sums * grad_mul (1 / world size), gate = isfinite(|update|^2) and
        isfinite(loss_sum[0]); ``valid`` (device int32[1]) is an output, 1 iff the update was applied."""
        if valid is not None and (valid.dtype != torch.int32 or not valid.is_cuda):
            raise ValueError("valid must be a device int32 tensor")
        if loss_sum.dtype != torch.float32 or not loss_sum.is_cuda:
            raise ValueError("loss_sum must be a device float32 tensor")
        self._check(self._L.amt_adamw_step_reduced(self._h, float(lr_base), float(layer_lr_decay), float(b1), float(b2),
                                                   float(eps), float(weight_decay), float(clip_norm), int(step),
                                                   float(grad_mul), _lib.ptr(loss_sum), _lib.ptr(valid), self._stream()),
                    "amt_adamw_step_reduced")

    def grad_bucket(self):
        """(tensor, head, n, tail): the context's gradient accumulator as ONE float32 device tensor
        ``[head control floats | n gradient sums | tail control floats]`` (include/amt.h: amt_grad_bucket) - collectives run
        on views of it, nothing is copied.  The tensor does not own the memory: keep the model alive."""
        h, n, t = C.c_uint64(), C.c_uint64(), C.c_uint64()
        ptr = self._L.amt_grad_bucket(self._h, C.byref(h), C.byref(n), C.byref(t))
        if not ptr:
            raise RuntimeError("amt_grad_bucket returned NULL")
        total = int(h.value + n.value + t.value)
        return _DevView(ptr, total, self.device).tensor(), int(h.value), int(n.value), int(t.value)

    def first_half_wait(self, timeout_ms: int = 0):
        """Block the calling host thread until the next "first half complete" signal of a ``loss_and_grads_phase(3, ...)``
        step has been published (include/amt.h: amt_first_half_wait): the collective of the first range may be issued."""
        self._check(self._L.amt_first_half_wait(self._h, int(timeout_ms)), "amt_first_half_wait")

    def first_half_sync(self):
        """Declare every first-half signal published so far as consumed (after unpaired rehearsal steps)."""
        self._check(self._L.amt_first_half_sync(self._h, self._stream()), "amt_first_half_sync")

    def first_half_status(self) -> int:
        """Number of first-half waits that gave up (0 in normal operation; synchronises)."""
        return int(self._L.amt_first_half_status(self._h))

    def optimizer_status(self) -> int:
        """Barrier time-outs of the one-launch optimizer seen so far (no step is lost to one; include/amt.h: amt_optimizer_status)."""
        return int(self._L.amt_optimizer_status(self._h))

    def reset_grad_accumulator(self):
        """Zero the context's gradient accumulator (after a fused backward pass whose gradients are not to be applied)."""
        self._check(self._L.amt_grad_accumulator_reset(self._h, self._stream()), "amt_grad_accumulator_reset")

    def optimizer_count(self) -> int:
        """Number of optimizer updates applied so far (device counter; synchronises).  include/amt.h: amt_optimizer_count."""
        return int(self._L.amt_optimizer_count(self._h))

    def optimizer_state(self):
        """(m, v, count): zero-copy device views of AdamW's moments (device layout, like the parameters) and the count of
        applied updates - what a checkpoint needs to resume (train.py:799-831 restores the optimizer state)."""
        pm, pv = self._L.amt_optimizer_moment_ptr(self._h, 0), self._L.amt_optimizer_moment_ptr(self._h, 1)
        if not pm or not pv:
            raise RuntimeError("no optimizer state: call init_optimizer first")
        return (_DevView(pm, self.n_params, self.device).tensor(), _DevView(pv, self.n_params, self.device).tensor(),
                self.optimizer_count())

    def load_optimizer_state(self, m, v, count: int):
        mm, vv, _ = self.optimizer_state()
        mm.copy_(torch.as_tensor(m).reshape(-1))
        vv.copy_(torch.as_tensor(v).reshape(-1))
        self._check(self._L.amt_optimizer_set_count(self._h, int(count)), "amt_optimizer_set_count")

    def workspace_bytes(self) -> int:
        return int(self._L.amt_workspace_bytes(self._h))

    def advance_seed(self, delta: int = 1):
        """*seed_word += delta on the current stream (include/amt.h: amt_advance_seed).  Every dropout mask is drawn
        from ``seed + seed_word``; capturing this call at the head of a step graph gives each replay fresh masks."""
        self._check(self._L.amt_advance_seed(self._h, int(delta), self._stream()), "amt_advance_seed")

    def pack_tail(self, loss: torch.Tensor, valid: torch.Tensor, head_valid: Optional[torch.Tensor] = None,
                  tail_loss: Optional[torch.Tensor] = None, tail_valid: Optional[torch.Tensor] = None):
        """(loss float32[>=1], grads_valid int32[1]) -> control words of a gradient bucket, one launch (include/amt.h:
        amt_pack_tail).  Each destination is a float32 view of ONE element (or None)."""
        def p(t, dt):
            if t is None:
                return None
            assert t.dtype == dt and t.is_contiguous() and t.numel() >= 1 and t.device == loss.device
            return C.c_void_p(t.data_ptr())
        self._check(self._L.amt_pack_tail(self._h, p(loss, torch.float32), p(valid, torch.int32), p(head_valid, torch.float32),
                                          p(tail_loss, torch.float32), p(tail_valid, torch.float32), self._stream()),
                    "amt_pack_tail")

    def seed_word(self) -> torch.Tensor:
        """The context's device seed word as an int64[1] tensor view (write to it to set the offset directly)."""
        ptr = self._L.amt_seed_word(self._h)
        if not ptr:
            raise RuntimeError("amt_seed_word returned NULL")
        return _DevView(ptr, 1, self.device, typestr="<i8").tensor()

    def plan_generation(self) -> int:
        """Changes exactly when a first-time plan had to replace the workspace, which invalidates every hipGraph
        captured from this model (include/amt.h: amt_plan_generation).  Record at capture, compare before replay."""
        return int(self._L.amt_plan_generation(self._h))

    def team_status(self) -> int:
        """0 unless an in-launch XCD-team hand-off has timed out (fatal; include/amt.h: amt_team_status)."""
        return int(self._L.amt_team_status(self._h))

    def tlayer_active(self) -> bool:
        """True when training-size calls run the transformer stack's forward as ONE XCD-team launch (csrc/tlayer.hip;
        include/amt.h: amt_tlayer_active); AMT_TLAYER=0 selects one launch per operator chain."""
        return bool(self._L.amt_tlayer_active(self._h))

    def tlayer_stamps(self, n_layers: int):
        """Per-phase wall-clock stamps (100 MHz) of the last team launch as an int64 array [256 workgroups][n_layers][32]
        (contexts created under AMT_TLAYER_STAMPS=1; None otherwise)."""
        import ctypes

        import numpy as np
        n = 256 * n_layers * 32
        buf = np.zeros(n, dtype=np.uint64)
        got = int(self._L.amt_tlayer_stamps(self._h, buf.ctypes.data_as(ctypes.c_void_p), n))
        if got <= 0:
            return None
        return buf.astype(np.int64).reshape(256, n_layers, 32)


class _DevView:
    """Wrap a raw device pointer as a torch tensor through __cuda_array_interface__."""

    def __init__(self, ptr, n, device, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3}
        self._device = device

    def tensor(self):
        return torch.as_tensor(self, device=self._device)
