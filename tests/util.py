"""Helpers shared by the GPU parity tests: ctypes calls on torch tensors + numpy mirrors."""
import ctypes as C

import numpy as np
import torch

import audio_to_midi_amd as amt
from audio_to_midi_amd import lib as L

TD = {"float32": torch.float32, "bfloat16": torch.bfloat16}
DT = {"float32": L.AMT_F32, "bfloat16": L.AMT_BF16}
# storage type of the forward RESIDUAL STREAM per activation dtype (csrc/common.h: Res<T>): IEEE half in bf16 contexts
RES_TD = {"float32": torch.float32, "bfloat16": torch.float16}
# (abs + rel*|ref|) tolerances per activation dtype for single ops
TOL = {"float32": (2e-5, 2e-5), "bfloat16": (2e-2, 2e-2)}


def dev(x, dtype="float32"):
    return torch.as_tensor(np.ascontiguousarray(x)).to(device="cuda", dtype=TD[dtype] if isinstance(dtype, str) else dtype).contiguous()


def dev_res(x, dtype):
    """a residual-stream tensor (block / layer input x, residual operand R) on the device, in its storage type"""
    return dev(x, RES_TD[dtype])


def f32(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def rounded(x, dtype):
    """value of x after storage in the activation dtype (numpy float64 out)"""
    t = torch.as_tensor(np.asarray(x, dtype=np.float64)).to(TD[dtype])
    return t.to(torch.float64).numpy()


_KEEP = []


def P(t):
    """device pointer of a tensor; the tensor is kept alive until the next call() returns
    (a temporary passed inline would otherwise be freed, and its block reused, before the launch)"""
    if t is None:
        return None
    _KEEP.append(t)
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    fn = getattr(L.load(), name)
    rc = fn(*args)
    torch.cuda.synchronize()
    _KEEP.clear()
    assert rc == 0, f"{name} returned {rc}"


def host(t):
    return t.detach().to(torch.float64).cpu().numpy()


def check(got, ref, dtype, scale=1.0, what=""):
    a, r = TOL[dtype]
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = np.abs(got - ref)
    bound = (a * scale) + r * np.abs(ref)
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} off; max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} (ref {ref.flat[err.argmax()]:.4f})"


def rel_l2(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30))


# ---- numpy mirror of the device's counter-based dropout RNG (csrc/common.h: rng_u32) ----
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def rng_u32(seed, stream_id, idx):
    idx = np.asarray(idx, dtype=np.uint32)
    s0 = np.uint32(seed & 0xFFFFFFFF); s1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        key = _mix32(_mix32(np.asarray(s0 ^ np.uint32((stream_id * 0x85ebca6b) & 0xFFFFFFFF), dtype=np.uint32)) + s1 + np.uint32(0x165667b1))
        h = _mix32((idx * np.uint32(0x9e3779b1)) ^ key)
    return h


def keep_mask(seed, stream_id, idx, p):
    thr = int(np.rint(np.float32(p) * np.float32(16777216.0)))
    return ((rng_u32(seed, stream_id, idx) >> np.uint32(8)) >= np.uint32(thr)).astype(np.float64), 1.0 - thr / 16777216.0


def dropout_masks_for(seed, conf, B, S, n_heads=4):
    """The device's dropout masks of one training step rebuilt on the host, in the form oracle/torch_model.py takes:
    per layer dict(attn = keep mask of the attention weights, ffn = keep mask of the FFN output, keep = keep
    probability).  Streams: 2 * layer for the attention weights, 2 * layer + 1 for the FFN output; element index =
    flat index of [B][H][nwin][16][16] (local layers), [B][H][S][S] (global layers), [B][S][256] (FFN); csrc/common.h."""
    import torch
    from oracle import np_model
    p = conf["transformer_dropout_rate"]
    _, _, nwin, _ = np_model.local_window_geometry(S)
    H = n_heads
    masks = []
    for li in range(2 * conf["num_transformer_layers"]):
        if li % 2 == 0:
            k_, kp = keep_mask(seed, 2 * li, np.arange(B * H * nwin * 256).reshape(B, H, nwin, 16, 16), p)
            attn = torch.tensor(k_).permute(0, 2, 1, 3, 4)
        else:
            k_, kp = keep_mask(seed, 2 * li, np.arange(B * H * S * S).reshape(B, H, S, S), p)
            attn = torch.tensor(k_)
        f_, _ = keep_mask(seed, 2 * li + 1, np.arange(B * S * 256).reshape(B, S, 256), p)
        masks.append(dict(attn=attn, ffn=torch.tensor(f_), keep=kp))
    return masks


def bf16_round_gemm_weights(params):
    """The parameters a bf16 context computes with (oracle/torch_model.py: round_gemm_weights)."""
    from oracle import torch_model
    return torch_model.round_gemm_weights(params)
