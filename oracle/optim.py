"""Oracle for the optimizer half of the training step: numpy-float64 restatement of the reference's Optax chain.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED against the reference itself (optax is not
vendored and cannot be imported here, requirements.txt:6 leaves it unpinned); the Adam / decoupled-weight-decay core is
pinned against ``torch.optim.AdamW`` in tests/test_oracle_kat.py, the schedule and the clipping by known answers.

Reference (train.py):
  * 455-466   create_learning_rate_schedule: optax.join_schedules([linear_schedule(0 -> base, warmup_steps),
              cosine_decay_schedule(base, cosine_decay_steps)], boundaries=[warmup_steps])
  * 646-724   setup_optimizers: one optax.adamw(schedule_d, weight_decay, eps=1e-3, b1=0.9, b2=0.999) per label under
              optax.multi_transform; label "conv_layer|d" for every leaf under ``layers`` with d = sum(depths[:stage]) +
              index in the stage's Sequential, schedule base = base_lr * layer_lr_decay ** (max_conv_depth - d);
              label "default|0" (base_lr) for everything else
  * 725       optax.chain(tx, optax.clip_by_global_norm(1.0)): the clip acts on the UPDATES of all labels together
  * 324-325   updates, opt_state = tx.update(grads, opt_state, params); model = eqx.apply_updates(model, updates)

Optax semantics restated (published algorithm, optax/_src/{transform,schedule,clipping,alias}.py):
  * scale_by_adam: mu = b1 mu + (1 - b1) g; nu = b2 nu + (1 - b2) g^2; count += 1;
    u = (mu / (1 - b1^count)) / (sqrt(nu / (1 - b2^count)) + eps)                      (eps_root = 0)
  * add_decayed_weights(wd): u += wd * p;   scale_by_learning_rate(schedule): u *= -schedule(count_before_increment)
    -> the FIRST update uses schedule(0), which is 0 under a warm-up from 0
  * linear_schedule(init, end, T)(c) = init + (end - init) * clip(c / T, 0, 1)  (T <= 0: end)
  * cosine_decay_schedule(init, T, alpha=0)(c) = init * 0.5 * (1 + cos(pi * min(c, T) / T))
  * join_schedules(schedules, [b])(c) = schedules[0](c) if c < b else schedules[1](c - b)
  * clip_by_global_norm(m): n = sqrt(sum u^2 over every leaf); u if n < m else u / n * m
"""
from __future__ import annotations

import math
from typing import Sequence

import numpy as np

ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-3      # train.py:692-694


def learning_rate(count: int, base: float, warmup_steps: int, cosine_decay_steps: int) -> float:
    """train.py:455-466 evaluated at update count ``count`` (0 for the first update)."""
    if count < warmup_steps:
        frac = min(max(count / warmup_steps, 0.0), 1.0) if warmup_steps > 0 else 1.0
        return base * frac
    c = count - warmup_steps
    if cosine_decay_steps <= 0:
        return 0.0
    return base * 0.5 * (1.0 + math.cos(math.pi * min(c, cosine_decay_steps) / cosine_decay_steps))


def lr_scale_from_depths(depths: np.ndarray, layer_lr_decay: float) -> np.ndarray:
    """Per-parameter learning-rate factor: layer_lr_decay ** (max_conv_depth - d) for conv leaves (d >= 0), 1 for the
    default label (d < 0); train.py:697-713."""
    depths = np.asarray(depths)
    max_depth = int(depths.max()) if (depths >= 0).any() else 0
    return np.where(depths >= 0, float(layer_lr_decay) ** (max_depth - np.maximum(depths, 0)), 1.0)


class AdamWChain:
    """State + update of the reference's chain over a FLAT float64 parameter vector with a per-parameter LR factor."""

    def __init__(self, n: int, lr_scale: Sequence[float] | None = None, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS,
                 weight_decay=0.005, clip_norm=1.0):
        self.mu = np.zeros(n)
        self.nu = np.zeros(n)
        self.count = 0
        self.lr_scale = np.ones(n) if lr_scale is None else np.asarray(lr_scale, dtype=np.float64)
        self.b1, self.b2, self.eps, self.wd, self.clip = b1, b2, eps, weight_decay, clip_norm

    def update(self, params: np.ndarray, grads: np.ndarray, lr: float) -> np.ndarray:
        """One tx.update + apply_updates with ``lr`` = the base schedule's value for this update; returns new params."""
        p = np.asarray(params, dtype=np.float64)
        g = np.asarray(grads, dtype=np.float64)
        self.mu = self.b1 * self.mu + (1 - self.b1) * g
        self.nu = self.b2 * self.nu + (1 - self.b2) * g * g
        self.count += 1
        mhat = self.mu / (1 - self.b1 ** self.count)
        vhat = self.nu / (1 - self.b2 ** self.count)
        u = mhat / (np.sqrt(vhat) + self.eps)
        u = u + self.wd * p
        u = -(lr * self.lr_scale) * u
        if self.clip is not None and self.clip > 0:
            n = math.sqrt(float(np.sum(u * u)))
            if not n < self.clip:
                u = u / n * self.clip
        return p + u
