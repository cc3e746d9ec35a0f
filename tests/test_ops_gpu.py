"""Op-level parity: every HIP kernel (through the C ABI of include/amt_ops.h) against a float64
torch/numpy reference of the same reference op.  fp32 mode pins the indexing tightly; bf16 mode
is the production dtype (inputs are pre-rounded so only the kernel's own rounding shows)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from util import (DT, RES_TD, TD, P, call, check, dev, dev_res, f32, host, keep_mask, rel_l2, rounded, stream)  # noqa: E402

DTYPES = ["float32", "bfloat16"]
EPI = dict(STORE=0, STORE_F32=1, RESID=2, GELU=3, SCALE_RESID=4, GLU=5, GLU_BWD=6, GELU_BWD=7, STORE_RES=8)


def gelu(x):
    return 0.5 * x * (1 + np.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def gelu_grad(x):
    t = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    F.gelu(t, approximate="tanh").sum().backward()
    return t.grad.numpy()


def gemm_nt(dtype, epi, X, W, bias=None, Y=None, Y2=None, R=None, U=None, gamma=None, seed=0, sid=0, p=0.0,
            ldx=None, ldy=None):
    N, K = X.shape[0], W.shape[1]
    M = W.shape[0]
    call("amt_op_gemm_nt", DT[dtype], epi, P(X), ldx or X.stride(0), P(W), W.stride(0), N, M, K, P(bias), P(Y),
         ldy or (Y.stride(0) if Y is not None else 0), P(Y2), Y2.stride(0) if Y2 is not None else 0, P(R),
         R.stride(0) if R is not None else 0, P(U), U.stride(0) if U is not None else 0, P(gamma), seed, sid, p,
         stream())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,M,K", [(100, 64, 32), (300, 320, 256), (257, 96, 96), (1000, 8, 4), (64, 4, 8),
                                   (77, 256, 320), (4000, 1024, 256), (33000, 64, 32), (130, 512, 64)])
def test_gemm_nt_store(dtype, N, M, K):
    rng = np.random.default_rng(N + M + K)
    X = rounded(rng.standard_normal((N, K)), dtype)
    W = rounded(rng.standard_normal((M, K)) / math.sqrt(K), dtype)
    b = rng.standard_normal(M).astype(np.float32)
    Xd, Wd = dev(X, dtype), dev(W, dtype)
    Y = torch.full((N, M), float("nan"), device="cuda", dtype=Xd.dtype)
    gemm_nt(dtype, EPI["STORE"], Xd, Wd, f32(b), Y)
    check(host(Y), X @ W.T + b, dtype, scale=1.0, what="gemm store")
    Yf = torch.full((N, M), float("nan"), device="cuda", dtype=torch.float32)
    gemm_nt(dtype, EPI["STORE_F32"], Xd, Wd, f32(b), Yf)
    check(host(Yf), X @ W.T + b, "float32", scale=1.0 if dtype == "float32" else 3.0, what="gemm store f32")


def test_gemm_nt_exact_integers():
    """asymmetric small-integer operands: any swapped row/col or k-permutation bug shows exactly"""
    N, M, K = 96, 128, 64
    X = (np.arange(N * K).reshape(N, K) % 7 - 3).astype(np.float64)
    W = ((np.arange(M * K).reshape(M, K) * 3) % 5 - 2).astype(np.float64)
    for dtype in DTYPES:
        Y = torch.zeros((N, M), device="cuda", dtype=torch.float32)
        gemm_nt(dtype, EPI["STORE_F32"], dev(X, dtype), dev(W, dtype), None, Y)
        assert np.array_equal(host(Y), X @ W.T)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_strided_operands(dtype):
    """X is a column slice of a wider matrix (c = QC[:, 256:320]) and Y a column slice (dc into dQC)"""
    rng = np.random.default_rng(5)
    N = 150
    QC = rounded(rng.standard_normal((N, 320)), dtype)
    W = rounded(rng.standard_normal((512, 64)) / 8, dtype)
    QCd = dev(QC, dtype)
    Y = torch.zeros((N, 512), device="cuda", dtype=QCd.dtype)
    gemm_nt(dtype, EPI["STORE"], QCd[:, 256:], dev(W, dtype), None, Y, ldx=320)
    check(host(Y), QC[:, 256:] @ W.T, dtype, what="strided X")
    Wt = rounded(rng.standard_normal((64, 512)) / 20, dtype)
    KV = rounded(rng.standard_normal((N, 512)), dtype)
    out = torch.zeros((N, 320), device="cuda", dtype=QCd.dtype)
    gemm_nt(dtype, EPI["STORE"], dev(KV, dtype), dev(Wt, dtype), None, out[:, 256:], ldy=320)
    ref = np.zeros((N, 320)); ref[:, 256:] = KV @ Wt.T
    check(host(out), ref, dtype, what="strided Y")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_epilogues(dtype):
    rng = np.random.default_rng(11)
    N, K = 200, 64
    X = rounded(rng.standard_normal((N, K)), dtype)
    Xd = dev(X, dtype)
    td = Xd.dtype
    # RESID
    M = 64
    W = rounded(rng.standard_normal((M, K)) / 8, dtype); b = rng.standard_normal(M).astype(np.float32)
    R = rounded(rng.standard_normal((N, M)), dtype)
    Y = torch.zeros((N, M), device="cuda", dtype=RES_TD[dtype])       # residual operand and output: residual-stream storage type
    gemm_nt(dtype, EPI["RESID"], Xd, dev(W, dtype), f32(b), Y, R=dev_res(R, dtype))
    check(host(Y), R + X @ W.T + b, dtype, what="resid")
    # RESID + dropout: mask must be the counter-based one
    Y.zero_()
    gemm_nt(dtype, EPI["RESID"], Xd, dev(W, dtype), f32(b), Y, R=dev_res(R, dtype), seed=1234567890123, sid=7, p=0.1)
    keep, kp = keep_mask(1234567890123, 7, np.arange(N * M).reshape(N, M), 0.1)
    assert 0.85 < keep.mean() < 0.95
    check(host(Y), R + (X @ W.T + b) * keep / kp, dtype, what="resid+dropout")
    # GELU (pre-activation rounded to the activation dtype, as stored)
    M = 128
    W = rounded(rng.standard_normal((M, K)) / 8, dtype); b = rng.standard_normal(M).astype(np.float32)
    U = torch.zeros((N, M), device="cuda", dtype=td); Hh = torch.zeros_like(U)
    gemm_nt(dtype, EPI["GELU"], Xd, dev(W, dtype), f32(b), U, Y2=Hh)
    u = X @ W.T + b
    check(host(U), u, dtype, what="gelu u")
    check(host(Hh), gelu(host(U)), dtype, what="gelu h")
    # SCALE_RESID
    M = 64
    W = rounded(rng.standard_normal((M, K)) / 8, dtype); b = rng.standard_normal(M).astype(np.float32)
    g = rng.standard_normal(M).astype(np.float32)
    Z = torch.zeros((N, M), device="cuda", dtype=td); Y = torch.zeros((N, M), device="cuda", dtype=RES_TD[dtype])
    gemm_nt(dtype, EPI["SCALE_RESID"], Xd, dev(W, dtype), f32(b), Y, Y2=Z, R=dev_res(R, dtype), gamma=f32(g))
    check(host(Z), X @ W.T + b, dtype, what="z")
    check(host(Y), R + g * host(Z), dtype, what="scale resid")
    # STORE_RES (Downsample conv: the output starts a stage's residual stream)
    Y = torch.zeros((N, M), device="cuda", dtype=RES_TD[dtype])
    gemm_nt(dtype, EPI["STORE_RES"], Xd, dev(W, dtype), f32(b), Y)
    check(host(Y), X @ W.T + b, dtype, what="store res")
    # GELU_BWD
    M = 128
    W = rounded(rng.standard_normal((M, K)) / 8, dtype)
    Us = rounded(rng.standard_normal((N, M)), dtype)
    D = torch.zeros((N, M), device="cuda", dtype=td)
    gemm_nt(dtype, EPI["GELU_BWD"], Xd, dev(W, dtype), None, D, U=dev(Us, dtype))
    check(host(D), (X @ W.T) * gelu_grad(Us), dtype, what="gelu bwd")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_glu_fwd_bwd(dtype):
    from audio_to_midi_amd.params import glu_row_perm
    rng = np.random.default_rng(3)
    N, K, inter = 150, 256, 512
    perm = glu_row_perm(inter)
    X = rounded(rng.standard_normal((N, K)), dtype)
    W1 = rounded(rng.standard_normal((2 * inter, K)) / 16, dtype)
    b1 = rng.standard_normal(2 * inter).astype(np.float32)
    U = torch.zeros((N, 2 * inter), device="cuda", dtype=dev(X, dtype).dtype)
    G = torch.zeros((N, inter), device="cuda", dtype=U.dtype)
    gemm_nt(dtype, EPI["GLU"], dev(X, dtype), dev(W1[perm], dtype), f32(b1[perm]), G, Y2=U)
    u = X @ W1.T + b1
    Uh = host(U)
    check(Uh, u[:, perm], dtype, what="glu u (permuted)")
    u_r = np.zeros_like(u); u_r[:, perm] = Uh
    check(host(G), gelu(u_r[:, :inter]) * u_r[:, inter:], dtype, what="glu g")
    # backward through the GLU: dg = dY . W2 ; du1 = dg u2 gelu'(u1) ; du2 = dg gelu(u1)
    Hd = 256
    dY = rounded(rng.standard_normal((N, Hd)), dtype)
    W2 = rounded(rng.standard_normal((Hd, inter)) / 16, dtype)
    dU = torch.zeros_like(U)
    gemm_nt(dtype, EPI["GLU_BWD"], dev(dY, dtype), dev(W2.T.copy(), dtype), None, dU, U=U)
    dg = dY @ W2
    du = np.concatenate([dg * u_r[:, inter:] * gelu_grad(u_r[:, :inter]), dg * gelu(u_r[:, :inter])], axis=1)
    check(host(dU), du[:, perm], dtype, scale=3.0, what="glu bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,M,Kp,mstore", [(500, 64, 64, 0), (4000, 320, 256, 0), (1000, 96, 256, 90), (3000, 8, 4, 0),
                                           (3000, 4, 8, 0), (777, 1024, 256, 0), (129, 512, 64, 0), (20000, 16, 8, 0),
                                           # the 32-clip sizes (token splits capped at 48, ragged last stage)
                                           (9001, 320, 256, 0), (16000, 1024, 256, 0)])
def test_gemm_tn_wgrad(dtype, N, M, Kp, mstore):
    rng = np.random.default_rng(N + M)
    G = rounded(rng.standard_normal((N, M)), dtype)
    A = rounded(rng.standard_normal((N, Kp)), dtype)
    Ms = mstore or M
    dW = torch.ones((Ms, Kp), device="cuda", dtype=torch.float32)
    db = torch.ones((Ms,), device="cuda", dtype=torch.float32)
    call("amt_op_gemm_tn_wgrad", DT[dtype], P(dev(G, dtype)), M, P(dev(A, dtype)), Kp, N, M, Kp, P(dW), Kp, P(db),
         mstore, stream())
    ref = G.T @ A
    sc = math.sqrt(N)
    check(host(dW), 1 + ref[:Ms], "float32", scale=sc * (1 if dtype == "float32" else 2), what="dW")
    check(host(db), 1 + G.sum(0)[:Ms], "float32", scale=sc, what="dbias")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [4, 8, 16, 32, 64, 128, 256])
def test_layernorm(dtype, C):
    rng = np.random.default_rng(C)
    N = 1000 + C
    X = rounded(rng.standard_normal((N, C)) * 2 + 0.5, dtype)
    w = rng.standard_normal(C).astype(np.float32); b = rng.standard_normal(C).astype(np.float32)
    Xd = dev_res(X, dtype)                    # every LayerNorm of the model reads the residual stream
    Y = torch.zeros((N, C), device="cuda", dtype=TD[dtype]); mean = torch.zeros(N, device="cuda"); rstd = torch.zeros(N, device="cuda")
    call("amt_op_layernorm_fwd", DT[dtype], P(Xd), P(f32(w)), P(f32(b)), P(Y), P(mean), P(rstd), N, C, stream())
    xt = torch.tensor(X, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64, requires_grad=True); bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = F.layer_norm(xt, (C,), wt, bt, 1e-5)
    check(host(Y), yt.detach().numpy(), dtype, scale=3.0, what="ln fwd")
    dY = rounded(rng.standard_normal((N, C)), dtype)
    dR = rounded(rng.standard_normal((N, C)), dtype)
    yt.backward(torch.tensor(dY))
    dX = torch.zeros_like(Y); dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    call("amt_op_layernorm_bwd", DT[dtype], P(dev(dY, dtype)), P(Xd), P(mean), P(rstd), P(f32(w)), P(dev(dR, dtype)),
         P(dX), P(dw), P(db), N, C, stream())
    check(host(dX), dR + xt.grad.numpy(), dtype, scale=4.0, what="ln dX")
    check(host(dw), wt.grad.numpy(), "float32", scale=math.sqrt(N) * 4, what="ln dw")
    check(host(db), bt.grad.numpy(), "float32", scale=math.sqrt(N) * 4, what="ln db")


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem(dtype):
    rng = np.random.default_rng(1)
    B, T = 3, 320 * 20 + 320 * 3
    L = T // 5
    a = rounded(rng.standard_normal((B, 2, T)), dtype)
    w = (rng.standard_normal((4, 2, 5)) * 0.3).astype(np.float32); b = rng.standard_normal(4).astype(np.float32)
    lw = rng.standard_normal(4).astype(np.float32); lb = rng.standard_normal(4).astype(np.float32)
    ad = dev_res(a, dtype)                                               # the audio has the residual stream's storage type
    Y = torch.zeros((B, L, 4), device="cuda", dtype=RES_TD[dtype])      # the Stem's output starts the residual stream
    call("amt_op_stem_fwd", DT[dtype], P(ad), P(f32(w)), P(f32(b)), P(f32(lw)), P(f32(lb)), P(Y), B, T, stream())
    tw = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (w, b, lw, lb)]
    c = F.conv1d(torch.tensor(a), tw[0], tw[1], stride=5)
    y = F.layer_norm(c.transpose(1, 2), (4,), tw[2], tw[3], 1e-5)
    check(host(Y), y.detach().numpy(), dtype, scale=3.0, what="stem fwd")
    dY = rounded(rng.standard_normal((B, L, 4)), dtype)
    y.backward(torch.tensor(dY))
    outs = [torch.zeros(n, device="cuda") for n in (40, 4, 4, 4)]
    call("amt_op_stem_bwd", DT[dtype], P(ad), P(dev(dY, dtype)), P(f32(w)), P(f32(b)), P(f32(lw)), *[P(o) for o in outs],
         B, T, stream())
    for o, t_, nm in zip(outs, tw, ["dw", "db", "dlnw", "dlnb"]):
        check(host(o), t_.grad.numpy().reshape(-1), "float32", scale=math.sqrt(B * L) * 4, what="stem " + nm)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,L", [(4, 700), (8, 333), (16, 200), (32, 150), (64, 100), (128, 90), (256, 50)])
def test_block_conv_pieces(dtype, C, L):
    """dwconv+LN forward, its backward, the depthwise-conv backward and the layer-scale backward"""
    rng = np.random.default_rng(C + L)
    B = 2
    X = rounded(rng.standard_normal((B, L, C)), dtype)
    wdw = (rng.standard_normal((C, 1, 7)) * 0.4).astype(np.float32); bdw = rng.standard_normal(C).astype(np.float32)
    lw = rng.standard_normal(C).astype(np.float32); lb = rng.standard_normal(C).astype(np.float32)
    wdev = f32(wdw[:, 0, :].T.copy())
    Xd = dev_res(X, dtype)                    # Block input: residual stream
    Y = torch.zeros((B, L, C), device="cuda", dtype=TD[dtype]); mean = torch.zeros(B * L, device="cuda"); rstd = torch.zeros(B * L, device="cuda")
    call("amt_op_dwln_fwd", DT[dtype], P(Xd), P(wdev), P(f32(bdw)), P(f32(lw)), P(f32(lb)), P(Y), P(mean), P(rstd), B, L,
         C, stream())
    xt = torch.tensor(X, requires_grad=True)
    tw = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (wdw, bdw, lw, lb)]
    d = F.conv1d(xt.transpose(1, 2), tw[0], tw[1], padding=3, groups=C)
    d.retain_grad()
    y = F.layer_norm(d.transpose(1, 2), (C,), tw[2], tw[3], 1e-5)
    check(host(Y), y.detach().numpy(), dtype, scale=3.0, what="dwln fwd")
    dY = rounded(rng.standard_normal((B, L, C)), dtype)
    y.backward(torch.tensor(dY))
    dD = torch.zeros_like(Y); dlw = torch.zeros(C, device="cuda"); dlb = torch.zeros(C, device="cuda")
    call("amt_op_dwln_bwd", DT[dtype], P(dev(dY, dtype)), P(Xd), P(wdev), P(f32(bdw)), P(mean), P(rstd), P(f32(lw)), P(dD),
         P(dlw), P(dlb), B, L, C, stream())
    dD_ref = d.grad.transpose(1, 2).numpy()
    check(host(dD), dD_ref, dtype, scale=4.0, what="dwln dD")
    sc = math.sqrt(B * L) * 4
    check(host(dlw), tw[2].grad.numpy(), "float32", scale=sc, what="dlnw")
    check(host(dlb), tw[3].grad.numpy(), "float32", scale=sc, what="dlnb")
    # depthwise conv backward from the (rounded) dD the kernel stored
    dDr = host(dD)
    xt2 = torch.tensor(X, requires_grad=True)
    w2 = torch.tensor(wdw, dtype=torch.float64, requires_grad=True); b2 = torch.tensor(bdw, dtype=torch.float64, requires_grad=True)
    d2 = F.conv1d(xt2.transpose(1, 2), w2, b2, padding=3, groups=C)
    d2.backward(torch.tensor(dDr).transpose(1, 2))
    dR = rounded(rng.standard_normal((B, L, C)), dtype)
    dX = torch.zeros_like(Y); dwd = torch.zeros((7, C), device="cuda"); dbd = torch.zeros(C, device="cuda")
    call("amt_op_dwconv_bwd", DT[dtype], P(dD), P(Xd), P(wdev), P(dev(dR, dtype)), P(dX), P(dwd), P(dbd), B, L, C,
         stream())
    check(host(dX), dR + xt2.grad.numpy(), dtype, scale=4.0, what="dwconv dX")
    check(host(dwd), w2.grad.numpy()[:, 0, :].T, "float32", scale=sc, what="dwconv dw")
    check(host(dbd), b2.grad.numpy(), "float32", scale=sc, what="dwconv db")
    # fused tail (dwln_bwd + dwconv_bwd + dgamma in one kernel, dD kept in LDS) against the same references
    Zt = rounded(rng.standard_normal((B, L, C)), dtype)
    dX2 = torch.zeros_like(Y); outs = [torch.zeros(n, device="cuda") for n in (C, C, 7 * C, C, C)]
    call("amt_op_dwblock_bwd", DT[dtype], P(dev(dY, dtype)), P(Xd), P(dev(dR, dtype)), P(dev(Zt, dtype)), P(wdev),
         P(f32(bdw)), P(mean), P(rstd), P(f32(lw)), P(dX2), *[P(o) for o in outs], B, L, C, stream())
    check(host(dX2), dR + xt2.grad.numpy(), dtype, scale=4.0, what="fused dX")
    check(host(outs[0]), tw[2].grad.numpy(), "float32", scale=sc, what="fused dlnw")
    check(host(outs[1]), tw[3].grad.numpy(), "float32", scale=sc, what="fused dlnb")
    check(host(outs[2]).reshape(7, C), w2.grad.numpy()[:, 0, :].T, "float32", scale=sc, what="fused dw")
    check(host(outs[3]), b2.grad.numpy(), "float32", scale=sc, what="fused db")
    check(host(outs[4]), (dR * Zt).sum((0, 1)), "float32", scale=sc, what="fused dgamma")
    # layer scale
    Z = rounded(rng.standard_normal((B * L, C)), dtype); g = rng.standard_normal(C).astype(np.float32)
    dZ = torch.zeros((B * L, C), device="cuda", dtype=Y.dtype); dg = torch.zeros(C, device="cuda")
    dYf = dY.reshape(B * L, C)
    call("amt_op_scale_bwd", DT[dtype], P(dev(dYf, dtype)), P(dev(Z, dtype)), P(f32(g)), P(dZ), P(dg), B * L, C, stream())
    check(host(dZ), g * dYf, dtype, scale=3.0, what="scale dZ")
    check(host(dg), (dYf * Z).sum(0), "float32", scale=sc, what="dgamma")


# ----------------------------------------------------------------------------- attention
def _rope_t(x, cos, sin):
    x1, x2 = x[..., 0::2], x[..., 1::2]
    c, s = cos[None, :, None, :], sin[None, :, None, :]
    return torch.stack([x1 * c - x2 * s, x1 * s + x2 * c], dim=-1).flatten(-2)


def _attn_ref(qc, kv, cos, sin, B, S, H, local, keep=None, kp=1.0):
    """float64 torch reference on the raw projections (model.py:241-257, 353-367, 409-471)."""
    from oracle.np_model import local_window_geometry
    q = qc[:, :, :256].reshape(B, S, H, 64)
    k = kv[:, :, :256].reshape(B, S, H, 64)
    v = kv[:, :, 256:].reshape(B, S, H, 64)

    def sdpa(q, k, v, pos_cos, pos_sin, km):
        q = _rope_t(q, pos_cos, pos_sin).transpose(1, 2) / 8.0
        k = _rope_t(k, pos_cos, pos_sin).transpose(1, 2)
        w = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        if km is not None:
            w = w * km / kp
        return (w @ v.transpose(1, 2)).transpose(1, 2)

    if not local:
        return sdpa(q, k, v, cos[:S], sin[:S], keep).reshape(B, S, 256)
    W = 16
    pl, pr, nwin, stride = local_window_geometry(S, W)
    pad = lambda t: F.pad(t, (0, 0, 0, 0, pl, pr))
    qp, kp_, vp = pad(q), pad(k), pad(v)
    out = torch.zeros(B, S, H, 64, dtype=qc.dtype)
    cnt = torch.zeros(S, dtype=qc.dtype)
    for wi in range(nwin):
        st = wi * stride
        km = None if keep is None else keep[:, wi]
        ow = sdpa(qp[:, st:st + W], kp_[:, st:st + W], vp[:, st:st + W], cos[:W], sin[:W], km)
        for off in range(W):
            if st + off < S:
                out[:, st + off] = out[:, st + off] + ow[:, off]
                cnt[st + off] += 1
    return (out / cnt[None, :, None, None]).reshape(B, S, 256)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("local,S,p", [(0, 250, 0.0), (0, 500, 0.0), (0, 100, 0.1), (1, 250, 0.0), (1, 256, 0.0),
                                       (1, 500, 0.0), (1, 50, 0.1), (0, 64, 0.0), (1, 16, 0.0)])
def test_attention(dtype, local, S, p):
    from audio_to_midi_amd.rope import precompute_frequencies
    from oracle.np_model import local_window_geometry
    rng = np.random.default_rng(S + local)
    B, H = 2, 4
    N = B * S
    rf = precompute_frequencies(64, 512)
    cos_d, sin_d = f32(rf.cos_freq), f32(rf.sin_freq)
    qc = rounded(rng.standard_normal((N, 320)), dtype)
    kv = rounded(rng.standard_normal((N, 512)), dtype)
    dO = rounded(rng.standard_normal((N, 256)), dtype)
    seed, sid = 987654321, 5
    keep, kp = None, 1.0
    if p > 0:
        if local:
            pl, pr, nwin, _ = local_window_geometry(S)
            idx = np.arange(B * H * nwin * 16 * 16).reshape(B, H, nwin, 16, 16)
            k_, kp = keep_mask(seed, sid, idx, p)
            keep = torch.tensor(k_).permute(0, 2, 1, 3, 4)           # (B, nwin, H, 16, 16)
        else:
            idx = np.arange(B * H * S * S).reshape(B, H, S, S)
            k_, kp = keep_mask(seed, sid, idx, p)
            keep = torch.tensor(k_)
    qct = torch.tensor(qc.reshape(B, S, 320), requires_grad=True)
    kvt = torch.tensor(kv.reshape(B, S, 512), requires_grad=True)
    ref = _attn_ref(qct, kvt, torch.tensor(rf.cos_freq, dtype=torch.float64), torch.tensor(rf.sin_freq, dtype=torch.float64),
                    B, S, H, bool(local), keep, kp)
    ref.backward(torch.tensor(dO.reshape(B, S, 256)))
    qcd, kvd = dev(qc, dtype), dev(kv, dtype)
    O = torch.zeros((N, 256), device="cuda", dtype=qcd.dtype)
    LSE = torch.zeros(B * H * S, device="cuda")
    call("amt_op_attn_fwd", DT[dtype], local, P(qcd), 320, P(kvd), 512, P(O), 256, P(LSE), P(cos_d), P(sin_d), B, S, H,
         seed, sid, p, stream())
    check(host(O), ref.detach().numpy().reshape(N, 256), dtype, scale=2.0, what="attn fwd")
    dQC = torch.full((N, 320), 0.0, device="cuda", dtype=qcd.dtype)
    dKV = torch.zeros((N, 512), device="cuda", dtype=qcd.dtype)
    delta = torch.zeros(B * H * S, device="cuda")
    call("amt_op_attn_bwd", DT[dtype], local, P(qcd), 320, P(kvd), 512, P(O), 256, P(LSE), P(cos_d), P(sin_d), B, S, H,
         seed, sid, p, P(dev(dO, dtype)), 256, P(dQC), 320, P(dKV), 512, P(delta), stream())
    gq = qct.grad.numpy().reshape(N, 320)[:, :256]
    gkv = kvt.grad.numpy().reshape(N, 512)
    tol = 1e-4 if dtype == "float32" else 3e-2
    assert rel_l2(host(dQC)[:, :256], gq) < tol, ("dq", rel_l2(host(dQC)[:, :256], gq))
    assert rel_l2(host(dKV)[:, :256], gkv[:, :256]) < tol, ("dk", rel_l2(host(dKV)[:, :256], gkv[:, :256]))
    assert rel_l2(host(dKV)[:, 256:], gkv[:, 256:]) < tol, ("dv", rel_l2(host(dKV)[:, 256:], gkv[:, 256:]))
    check(host(dQC)[:, :256], gq, dtype, scale=8.0, what="attn dq")
    check(host(dKV), gkv, dtype, scale=8.0, what="attn dkv")


@pytest.mark.parametrize("S,p,sharp", [(250, 0.0, 1.0), (500, 0.0, 1.0), (500, 0.0, 2.0), (500, 0.0, 4.0), (100, 0.1, 1.0), (64, 0.0, 2.0)])
def test_attention_fp8_forward(S, p, sharp):
    """BASELINE configs[4]: global attention forward with QK^T / PV on fp8 (e4m3) MFMA, per-tensor delayed scaling.
    Not a reference path (model.py:333-386 computes in the activation dtype), so the bound is a stated fp8 one:
    e4m3 has 3 mantissa bits (relative rounding <= 2^-4), and a score error of e changes a probability by a
    factor exp(e), so the bound widens with the spread of the scores.  Against the float64 attention of the
    bf16-rounded inputs, (relative L2, max abs, |d log-sum-exp| vs the bf16 kernel):
      scores ~ N(0, 1)  (the model at init is flatter still): 7e-2, 0.12, 0.1    measured 5.3e-2, 0.065
      scores ~ N(0, 2.8^2):                                   0.11, 0.9, 0.6     measured 8.1e-2, 0.55, 0.35
      scores ~ N(0, 8^2) (near one-hot rows):                 0.16, 4.0, 2.0     measured 0.117, 2.7, 1.2
    The same bound must hold on the first call
    (no |max| history: default scale) and on the second (scales from the first call's maxima), and the maxima
    recorded by the kernel must equal those of the rotated inputs."""
    from audio_to_midi_amd.rope import precompute_frequencies
    rng = np.random.default_rng(S)
    B, H = 2, 4
    N = B * S
    b_rel, b_abs, b_lse = {1.0: (7e-2, 0.12, 0.1), 2.0: (0.11, 0.9, 0.6), 4.0: (0.16, 4.0, 2.0)}[sharp]
    rf = precompute_frequencies(64, 512)
    cos_d, sin_d = f32(rf.cos_freq), f32(rf.sin_freq)
    qc = rounded(rng.standard_normal((N, 320)) * sharp, "bfloat16")
    kv = rounded(rng.standard_normal((N, 512)) * np.sqrt(sharp), "bfloat16")
    seed, sid = 555, 3
    keep, kp = None, 1.0
    if p > 0:
        k_, kp = keep_mask(seed, sid, np.arange(B * H * S * S).reshape(B, H, S, S), p)
        keep = torch.tensor(k_)
    cos64, sin64 = torch.tensor(rf.cos_freq, dtype=torch.float64), torch.tensor(rf.sin_freq, dtype=torch.float64)
    ref = _attn_ref(torch.tensor(qc.reshape(B, S, 320)), torch.tensor(kv.reshape(B, S, 512)), cos64, sin64, B, S, H,
                    False, keep, kp).numpy().reshape(N, 256)
    qcd, kvd = dev(qc, "bfloat16"), dev(kv, "bfloat16")
    amax = torch.zeros(8, device="cuda")
    for it in range(2):
        O = torch.zeros((N, 256), device="cuda", dtype=torch.bfloat16)
        LSE = torch.zeros(B * H * S, device="cuda")
        call("amt_op_attn_global_fwd_fp8", P(qcd), 320, P(kvd), 512, P(O), 256, P(LSE), P(cos_d), P(sin_d), B, S, H,
             seed, sid, p, P(amax), stream())
        Oh = host(O)
        assert np.isfinite(Oh).all()
        err, mx = rel_l2(Oh, ref), float(np.abs(Oh - ref).max())
        assert err < b_rel and mx < b_abs, (it, err, mx)
        O16 = torch.zeros_like(O)
        LSE16 = torch.zeros_like(LSE)
        call("amt_op_attn_fwd", DT["bfloat16"], 0, P(qcd), 320, P(kvd), 512, P(O16), 256, P(LSE16), P(cos_d), P(sin_d),
             B, S, H, seed, sid, p, stream())
        assert float((LSE - LSE16).abs().max()) < b_lse
        am = host(amax)
        q_ = _rope_t(torch.tensor(qc.reshape(B, S, 320))[:, :, :256].reshape(B, S, H, 64), cos64[:S], sin64[:S]) / 8.0
        k_r = _rope_t(torch.tensor(kv.reshape(B, S, 512))[:, :, :256].reshape(B, S, H, 64), cos64[:S], sin64[:S])
        assert abs(am[4] - float(q_.abs().max())) < 2e-2 * am[4] and abs(am[5] - float(k_r.abs().max())) < 2e-2 * am[5]
        assert am[6] == float(np.abs(kv[:, 256:]).max())
        if it == 1:
            assert am[0] == am[4] and am[1] == am[5] and am[2] == am[6]     # second call scaled by the first's maxima


@pytest.mark.parametrize("dtype", DTYPES)
def test_bce_and_dropout(dtype):
    rng = np.random.default_rng(9)
    N, V, B = 300, 90, 4
    z = (rng.standard_normal((N, 96)) * 3).astype(np.float32)
    y = rng.uniform(0.005, 0.995, (N, V)).astype(np.float32)
    loss = torch.zeros(2, device="cuda")
    dl = torch.full((N, 96), float("nan"), device="cuda", dtype=dev(z, dtype).dtype)
    call("amt_op_bce_loss", DT[dtype], P(f32(z)), 96, P(f32(y)), P(loss), P(dl), 96, N, V, 4.0, B, stream())
    zt = torch.tensor(z[:, :V], dtype=torch.float64, requires_grad=True)
    l = F.binary_cross_entropy_with_logits(zt, torch.tensor(y, dtype=torch.float64), reduction="sum") / B
    (l * 4.0).backward()
    lv = host(loss)
    assert abs(lv[0] - float(l)) < 1e-4 * abs(float(l)) and abs(lv[1] - 4 * float(l)) < 4e-4 * abs(float(l))
    ref = np.zeros((N, 96)); ref[:, :V] = zt.grad.numpy()
    check(host(dl), ref, dtype, what="dlogits")
    x = rounded(rng.standard_normal(10000), dtype)
    out = torch.zeros(10000, device="cuda", dtype=dl.dtype)
    call("amt_op_dropout", DT[dtype], P(dev(x, dtype)), P(out), 10000, 42, 3, 0.1, stream())
    keep, kp = keep_mask(42, 3, np.arange(10000), 0.1)
    check(host(out), x * keep / kp, dtype, what="dropout")
