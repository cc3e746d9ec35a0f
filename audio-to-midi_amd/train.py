"""Host side of the data-parallel training step (mirror of train.py:259-332, 344-382).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Parameters are
replicated, the batch is sharded along axis 0 (reference: `NamedSharding(mesh, P("batch"))`,
train.py:238-244, 280), and the ONLY exchange step is one sum-all-reduce of the flat fp32 gradient
bucket with the loss and the finite flag folded into its tail — the explicit form of the gradient
all-reduce XLA's GSPMD inserts implicitly in the reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import torch

BUCKET_TAIL = 3  # loss, scaled_loss, grads_valid


def new_bucket(n_params: int, device) -> torch.Tensor:
    """[grads (n_params) | loss | scaled_loss | valid] in one contiguous fp32 buffer."""
    return torch.zeros(n_params + BUCKET_TAIL, dtype=torch.float32, device=device)


def pack_tail(bucket: torch.Tensor, loss, scaled_loss, valid) -> None:
    n = bucket.numel() - BUCKET_TAIL
    bucket[n] = loss
    bucket[n + 1] = scaled_loss
    bucket[n + 2] = valid


def allreduce_bucket(bucket: torch.Tensor, dist=None) -> Tuple[torch.Tensor, float, float, bool]:
    """Sum over ranks, divide by the world size (the reference takes the MEAN over the global
    batch, train.py:62, and every rank holds the mean over its shard).  The finite flag becomes
    "all ranks finite" (train.py:320-322).  Returns (grads view, loss, scaled_loss, grads_valid)."""
    world = 1
    if dist is not None and dist.is_initialized():
        world = dist.get_world_size()
        if world > 1:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
            bucket.mul_(1.0 / world)
    n = bucket.numel() - BUCKET_TAIL
    tail = bucket[n:].tolist()
    valid = abs(tail[2] - 1.0) < 0.5 / max(world, 1)   # mean of 0/1 flags is 1 only if all are 1
    return bucket[:n], float(tail[0]), float(tail[1]), bool(valid)


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


def compute_training_step(model, audio, labels, *, grad_scale: float, seed: int, bucket: torch.Tensor,
                          lr_base: float, step: int, dist=None, accum_steps: int = 1,
                          apply_update: bool = True, local_step: Optional[Callable] = None):
    """One optimizer step: per-rank forward/backward on this rank's shard (optionally split into
    `accum_steps` minibatches accumulated in fp32, train.py:275-310), ONE all-reduce, AdamW.

    `local_step(audio, labels, scale, seed) -> (loss, scaled_loss, grads_flat, valid)` defaults to
    the HIP path (`model.loss_and_grads`); tests inject the oracle there to exercise the collective
    protocol on CPU tensors."""
    n = bucket.numel() - BUCKET_TAIL
    B = audio.shape[0]
    if B % accum_steps:
        raise ValueError("batch not divisible by accum_steps")
    mb = B // accum_steps
    acc = bucket[:n]
    tot_loss = tot_scaled = 0.0
    all_valid = True
    for i in range(accum_steps):
        a, y = audio[i * mb:(i + 1) * mb], labels[i * mb:(i + 1) * mb]
        if local_step is None:
            out = model.loss_and_grads(a, y, scale=grad_scale, seed=seed + i, train=True,
                                       grads_out=acc if accum_steps == 1 else None)
            loss, scaled, g, valid = out.loss, out.scaled_loss, out.grads_flat, out.grads_valid
        else:
            loss, scaled, g, valid = local_step(a, y, grad_scale, seed + i)
        if accum_steps == 1:
            if g.data_ptr() != acc.data_ptr():
                acc.copy_(g)
        elif i == 0:
            acc.copy_(g)
        else:
            acc.add_(g)
        tot_loss += loss
        tot_scaled += scaled
        all_valid = all_valid and valid
    if accum_steps > 1:
        acc.mul_(1.0 / accum_steps)
    pack_tail(bucket, tot_loss / accum_steps, tot_scaled / accum_steps, 1.0 if all_valid else 0.0)
    grads, loss, scaled_loss, valid = allreduce_bucket(bucket, dist)
    if apply_update and valid and model is not None:
        model.adamw_step(grads, lr_base=lr_base, step=step)
    return loss, scaled_loss, valid
