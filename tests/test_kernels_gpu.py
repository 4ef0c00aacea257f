"""Numerics of every gfx950 kernel against a plain PyTorch fp32 reference of the same op."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]


def C():
    from pytorch_distributed_training_amd.ops import native

    return native()


def tol(dtype):
    return (2e-2, 2e-2) if dtype == torch.bfloat16 else (4e-3, 4e-3)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def gelu_grad(x):
    """d gelu(x) / dx (exact-erf GELU), fp32"""
    xf = x.detach().float().requires_grad_()
    F.gelu(xf).sum().backward()
    return xf.grad


def rng_state(seed=1234, offset=0):
    return torch.tensor([seed, offset], dtype=torch.int64, device="cuda")


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 384, 192), (1024, 768, 1024), (77, 128, 128)])
def test_gemm_nt_plain_and_bias(dtype, tile, M, N, K):
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").to(dtype)
    W = torch.randn(N, K, device="cuda").to(dtype)
    b = torch.randn(N, device="cuda").to(dtype)
    out = C().op_gemm("nt", A, W, 0, None, None, None, None, 0.0, 0, 0, tile)
    ref = A.float() @ W.float().t()
    assert rel_err(out, ref) < (1e-2 if dtype == torch.bfloat16 else 2e-3)
    out = C().op_gemm("nt", A, W, 1, b, None, None, None, 0.0, 0, 0, tile)
    assert rel_err(out, ref + b.float()) < (1e-2 if dtype == torch.bfloat16 else 2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_asymmetric_identity(dtype):
    # A = I with an asymmetric B catches a transposed C write
    n = 256
    A = torch.eye(n, device="cuda").to(dtype)
    W = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).float().sub(125).div(16).to(dtype)
    for tile in (1, 2):
        out = C().op_gemm("nt", A, W, 0, None, None, None, None, 0.0, 0, 0, tile)
        assert torch.equal(out.float(), W.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5])
def test_gemm_nt_gelu_and_residual_epilogues(dtype, tile):
    torch.manual_seed(1)
    M, N, K = 520, 512, 256
    A = (torch.randn(M, K, device="cuda") * 0.5).to(dtype)
    W = (torch.randn(N, K, device="cuda") * 0.1).to(dtype)
    b = torch.randn(N, device="cuda").to(dtype)
    res = torch.randn(M, N, device="cuda").to(dtype)
    u = torch.empty(M, N, device="cuda", dtype=dtype)
    g = C().op_gemm("nt", A, W, 2, b, None, u, None, 0.0, 0, 0, tile)
    pre = A.float() @ W.float().t() + b.float()
    assert rel_err(u, gelu_grad(pre)) < 1e-2  # aux = gelu'(pre-activation), what MUL_DGELU multiplies with
    assert rel_err(g, F.gelu(pre)) < 1e-2
    z = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(), 0.0, 0, 0, tile)
    assert rel_err(z, pre + res.float()) < 1e-2
    t = C().op_gemm("nt", A, W, 7, b, None, None, None, 0.0, 0, 0, tile)
    assert rel_err(t, torch.tanh(pre)) < 1e-2


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dropout_epilogue_statistics_and_replay(dtype):
    M, N, K = 512, 1024, 64
    A = torch.zeros(M, K, device="cuda", dtype=dtype)
    W = torch.zeros(N, K, device="cuda", dtype=dtype)
    b = torch.ones(N, device="cuda", dtype=dtype)
    res = torch.zeros(M, N, device="cuda", dtype=dtype)
    st = rng_state(7, 3)
    z1 = C().op_gemm("nt", A, W, 3, b, res, None, st, 0.1, 2, 5, 1)
    for tile in (0, 2):  # other tile shapes: same mask
        z2 = C().op_gemm("nt", A, W, 3, b, res, None, st, 0.1, 2, 5, tile)
        assert torch.equal(z1, z2)
    keep = (z1 != 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01
    assert abs(z1.float().max().item() - 1 / 0.9) < 2e-2
    z3 = C().op_gemm("nt", A, W, 3, b, res, None, st, 0.1, 3, 5, 1)  # other micro-batch: other mask
    assert not torch.equal(z1, z3)
    # the standalone dropout kernel regenerates the same mask (what LN backward relies on)
    ones = torch.ones(M, N, device="cuda", dtype=dtype)
    d = C().dropout16(ones, st, 0.1, 2, 5, None)
    assert torch.equal(d != 0, z1 != 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (333, 512, 768), (1024, 1024, 3072)])
def test_gemm_nn_dgrad(dtype, tile, M, N, K):
    torch.manual_seed(2)
    dY = (torch.randn(M, K, device="cuda") * 0.5).to(dtype)
    W = (torch.randn(K, N, device="cuda") * 0.1).to(dtype)  # [out=K, in=N]
    out = C().op_gemm("nn", dY, W, 0, None, None, None, None, 0.0, 0, 0, tile)
    ref = dY.float() @ W.float()
    assert rel_err(out, ref) < 1e-2
    res = torch.randn(M, N, device="cuda").to(dtype)
    out = C().op_gemm("nn", dY, W, 5, None, res, None, None, 0.0, 0, 0, tile)
    assert rel_err(out, ref + res.float()) < 1e-2
    u = gelu_grad(torch.randn(M, N, device="cuda")).to(dtype)  # the derivative saved by the forward epilogue
    out = C().op_gemm("nn", dY, W, 4, None, None, u, None, 0.0, 0, 0, tile)
    assert rel_err(out, ref * u.float()) < 1e-2
    # fused bias gradient: column sums of the stored output, accumulated into an fp32 vector
    cs = torch.ones(N, device="cuda")
    out2 = C().op_gemm("nn", dY, W, 4, None, None, u, None, 0.0, 0, 0, tile, cs)
    # same values up to the summation order over K (without column sums the dispatcher may split K)
    rt, at = tol(dtype)
    torch.testing.assert_close(out2.float(), out.float(), rtol=rt, atol=at)
    torch.testing.assert_close(cs, 1 + out2.float().sum(0), rtol=2e-3, atol=2e-2)
    # the sums are formed without atomics: a second run is bit-identical
    cs2 = torch.ones(N, device="cuda")
    C().op_gemm("nn", dY, W, 4, None, None, u, None, 0.0, 0, 0, tile, cs2)
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nn_asymmetric_identity(dtype):
    n = 256
    A = torch.eye(n, device="cuda").to(dtype)
    W = (torch.arange(n * n, device="cuda").reshape(n, n) % 241).float().sub(120).div(16).to(dtype)
    for tile in (1, 2):
        out = C().op_gemm("nn", A, W, 0, None, None, None, None, 0.0, 0, 0, tile)
        assert torch.equal(out.float(), W.float())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("T,M,N", [(64, 256, 256), (1000, 512, 256), (4096, 768, 3072), (130, 128, 384)])
def test_gemm_tn_wgrad(dtype, tile, T, M, N):
    torch.manual_seed(3)
    dY = (torch.randn(T, M, device="cuda") * 0.5).to(dtype)
    X = (torch.randn(T, N, device="cuda") * 0.5).to(dtype)
    G = C().op_gemm_tn(dY, X, None, False, tile, 0)
    ref = dY.float().t() @ X.float()
    assert rel_err(G, ref) < 5e-3
    G2 = C().op_gemm_tn(dY, X, G.clone(), True, tile, 3)
    assert rel_err(G2, 2 * ref) < 5e-3


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,H", [(37, 128), (512, 768), (1000, 1024)])
def test_layernorm_fwd_bwd(dtype, rows, H):
    torch.manual_seed(4)
    x = torch.randn(rows, H, device="cuda").to(dtype)
    g = (1 + 0.1 * torch.randn(H, device="cuda")).to(dtype)
    b = (0.1 * torch.randn(H, device="cuda")).to(dtype)
    y, stats = C().op_layernorm_fwd(x, g, b, 1e-12)
    xf = x.float().requires_grad_()
    gf, bf = g.float().requires_grad_(), b.float().requires_grad_()
    ref = F.layer_norm(xf, (H,), gf, bf, 1e-12)
    rt, at = tol(dtype)
    torch.testing.assert_close(y.float(), ref, rtol=rt, atol=at)
    dy = torch.randn(rows, H, device="cuda").to(dtype)
    ref.backward(dy.float())
    dx, ddrop, gg, gb, gbias = C().op_layernorm_bwd(dy, x, stats, g, None, 0.0, 0, 0)
    assert rel_err(dx, xf.grad) < 2e-2
    assert rel_err(gg, gf.grad) < 1e-2
    assert rel_err(gb, bf.grad) < 1e-2
    assert rel_err(gbias, dx.float().sum(0)) < 1e-2
    # with dropout: ddrop = mask * dx / keep, and the bias gradient sums ddrop
    st = rng_state(5, 1)
    dx2, ddrop2, _, _, gbias2 = C().op_layernorm_bwd(dy, x, stats, g, st, 0.25, 1, 9)
    mask = C().dropout16(torch.ones_like(x), st, 0.25, 1, 9, None)
    assert torch.equal(dx2, dx)
    torch.testing.assert_close(ddrop2.float(), dx.float() * mask.float(), rtol=1e-2, atol=1e-3)
    assert rel_err(gbias2, ddrop2.float().sum(0)) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_split_k_forward_and_dgrad_epilogues(dtype):
    """Few output tiles + long K: the dispatcher cuts K into fp32 slabs and a second kernel sums them and applies
    the epilogue.  Every epilogue kind of both layouts against fp32 PyTorch."""
    torch.manual_seed(21)
    M, N, K = 384, 256, 4096
    rt, at = tol(dtype)
    A = (torch.randn(M, K, device="cuda") * 0.25).to(dtype)
    W = (torch.randn(N, K, device="cuda") * 0.06).to(dtype)
    b = (torch.randn(N, device="cuda") * 0.2).to(dtype)
    res = torch.randn(M, N, device="cuda").to(dtype)
    ref = A.float() @ W.float().t()
    assert C().gemm_splits("nt", A, W) > 1
    out = C().op_gemm("nt", A, W, 0, None, None, None, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(out.float(), ref, rtol=rt, atol=at)
    out = C().op_gemm("nt", A, W, 1, b, None, None, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(out.float(), ref + b.float(), rtol=rt, atol=at)
    u = torch.empty(M, N, device="cuda", dtype=dtype)
    g = C().op_gemm("nt", A, W, 2, b, None, u, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(u.float(), gelu_grad(ref + b.float()), rtol=rt, atol=at)
    torch.testing.assert_close(g.float(), torch.nn.functional.gelu(ref + b.float()), rtol=rt, atol=at)
    z = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(), 0.0, 0, 0, -1)
    torch.testing.assert_close(z.float(), ref + b.float() + res.float(), rtol=rt, atol=at)
    # dropout in the split path uses the same counters as the one-kernel path
    z1 = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(9, 1), 0.1, 2, 5, -1)
    z2 = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(9, 1), 0.1, 2, 5, 0)
    dropped1 = (z1.float() - res.float()).abs() < 1e-6
    dropped2 = (z2.float() - res.float()).abs() < 1e-6
    assert torch.equal(dropped1, dropped2) and 0.05 < dropped1.float().mean().item() < 0.15
    # dgrad layout
    Wn = (torch.randn(K, N, device="cuda") * 0.06).to(dtype)
    refn = A.float() @ Wn.float()
    out = C().op_gemm("nn", A, Wn, 0, None, None, None, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(out.float(), refn, rtol=rt, atol=at)
    out = C().op_gemm("nn", A, Wn, 5, None, res, None, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(out.float(), refn + res.float(), rtol=rt, atol=at)
    dg = gelu_grad(torch.randn(M, N, device="cuda")).to(dtype)  # what the forward epilogue saves
    out = C().op_gemm("nn", A, Wn, 4, None, None, dg, None, 0.0, 0, 0, -1)
    torch.testing.assert_close(out.float(), refn * dg.float(), rtol=rt, atol=at)


# ------------------------------------------------------------------------------------------ attention
def ref_attention(qkv, mask, B, S, A, probs_mask=None, keep=1.0):
    H = A * 64
    q, k, v = qkv.float().view(B, S, 3, A, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / 8.0
    s = s + (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    p = torch.softmax(s, -1)
    if probs_mask is not None:
        p = p * probs_mask / keep
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H), p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,S,A", [(2, 128, 2), (3, 77, 4), (2, 200, 2), (1, 512, 2), (2, 384, 1)])
def test_attention_fwd_bwd(dtype, B, S, A):
    torch.manual_seed(5)
    H = A * 64
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.7).to(dtype)
    lens = torch.randint(max(1, S // 3), S + 1, (B,), device="cuda")
    lens[0] = S
    mask = (torch.arange(S, device="cuda")[None, :] < lens[:, None]).to(torch.int32)
    kv_len = lens.to(torch.int32)
    ctx, lse = C().op_attn_fwd(qkv, mask, kv_len, B, S, A, None, 0.0, 0, 0)
    qf = qkv.float().requires_grad_()
    ref, _ = ref_attention(qf, mask, B, S, A)
    rt, at = tol(dtype)
    torch.testing.assert_close(ctx.float(), ref, rtol=rt, atol=at)
    dctx = torch.randn(B * S, H, device="cuda").to(dtype)
    # padded query rows carry no gradient in a real model
    dctx = dctx * mask.view(B * S, 1).to(dtype)
    ref.backward(dctx.float())
    fused = C().attn_bwd_fuses_dbias(S)
    dbias = torch.full((3 * H,), 0.5, device="cuda") if fused else None
    dqkv = C().op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, None, 0.0, 0, 0, dbias)
    assert rel_err(dqkv, qf.grad) < (3e-2 if dtype == torch.bfloat16 else 6e-3)
    if fused:  # the short-sequence kernel adds the QKV bias gradient (column sums of dqkv) into `dbias`
        assert rel_err(dbias - 0.5, qf.grad.sum(0)) < (2e-2 if dtype == torch.bfloat16 else 5e-3)


@pytest.mark.parametrize("S", [128, 320])
@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_general_mask_and_dropout(dtype, S):
    torch.manual_seed(6)
    B, A = 2, 2
    H = A * 64
    qkv = (torch.randn(B * S, 3 * H, device="cuda") * 0.5).to(dtype)
    mask = (torch.rand(B, S, device="cuda") > 0.3).to(torch.int32)
    mask[:, 0] = 1
    idx = torch.arange(1, S + 1, device="cuda", dtype=torch.int32)
    kv_len = (mask * idx).amax(1).to(torch.int32)
    st = rng_state(11, 2)
    p = 0.1
    # recover the probability mask from V = identity-like probes: run with V = one-hot over keys in chunks of 64
    ctx, lse = C().op_attn_fwd(qkv, mask, kv_len, B, S, A, st, p, 1, 4)
    ctx0, _ = C().op_attn_fwd(qkv, mask, kv_len, B, S, A, None, 0.0, 0, 0)
    probes = []
    for c in range(S // 64):
        q2 = qkv.clone().view(B, S, 3, A, 64)
        eye = torch.zeros(S, 64, device="cuda", dtype=dtype)
        eye[c * 64:(c + 1) * 64] = torch.eye(64, device="cuda", dtype=dtype)
        q2[:, :, 2] = eye[None, :, None, :]
        o, _ = C().op_attn_fwd(q2.view(B * S, 3 * H), mask, kv_len, B, S, A, st, p, 1, 4)
        probes.append(o.view(B, S, A, 64))
    pd = torch.cat(probes, -1).permute(0, 2, 1, 3).float()  # [B,A,S(q),S(k)] dropped probabilities
    _, pref = ref_attention(qkv, mask, B, S, A)
    keepmask = (pd != 0).float()
    valid = pref > 1e-4
    frac = keepmask[valid].mean().item()
    assert abs(frac - (1 - p)) < 0.02
    torch.testing.assert_close(pd, pref * keepmask / (1 - p), rtol=3e-2, atol=3e-3)
    # the decisions of neighbouring keys / neighbouring query rows / neighbouring heads are uncorrelated
    def corr(x, y, ok):
        x, y = x[ok] - x[ok].mean(), y[ok] - y[ok].mean()
        return (x * y).mean().item() / (x.std().item() * y.std().item() + 1e-12)
    assert abs(corr(keepmask[..., :-1], keepmask[..., 1:], valid[..., :-1] & valid[..., 1:])) < 0.02
    assert abs(corr(keepmask[:, :, :-1], keepmask[:, :, 1:], valid[:, :, :-1] & valid[:, :, 1:])) < 0.02
    assert abs(corr(keepmask[:, :-1], keepmask[:, 1:], valid[:, :-1] & valid[:, 1:])) < 0.02
    # backward with the same dropout mask
    qf = qkv.float().requires_grad_()
    ref, _ = ref_attention(qf, mask, B, S, A, keepmask, 1 - p)
    torch.testing.assert_close(ctx.float(), ref, rtol=3e-2, atol=2e-2)
    dctx = torch.randn(B * S, H, device="cuda").to(dtype)
    ref.backward(dctx.float())
    dbias = torch.zeros(3 * H, device="cuda") if C().attn_bwd_fuses_dbias(S) else None
    dqkv = C().op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, st, p, 1, 4, dbias)
    if dbias is not None:
        assert rel_err(dbias, dqkv.float().sum(0)) < 5e-3
    # keys that are masked get exactly zero gradient
    gk = dqkv.view(B, S, 3, A, 64)[:, :, 1:]
    assert gk[mask == 0].abs().max().item() == 0
    assert rel_err(dqkv, qf.grad) < (4e-2 if dtype == torch.bfloat16 else 1e-2)
    assert ctx0.shape == ctx.shape


@pytest.mark.parametrize("S", [128, 160])
def test_attention_fully_masked_sample_is_finite(S):
    """A sample whose keys are all masked gets zero context and zero gradients (no NaN from exp(-inf + inf))."""
    torch.manual_seed(8)
    B, A = 2, 2
    H = A * 64
    qkv = torch.randn(B * S, 3 * H, device="cuda").half()
    mask = torch.ones(B, S, dtype=torch.int32, device="cuda")
    mask[1] = 0
    kv_len = torch.tensor([S, S], dtype=torch.int32, device="cuda")
    ctx, lse = C().op_attn_fwd(qkv, mask, kv_len, B, S, A, None, 0.0, 0, 0)
    assert torch.isfinite(ctx.float()).all() and ctx.view(B, S, H)[1].abs().max().item() == 0
    dctx = torch.randn(B * S, H, device="cuda").half()
    dqkv = C().op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, None, 0.0, 0, 0)
    assert torch.isfinite(dqkv.float()).all() and dqkv.view(B, S, 3 * H)[1].abs().max().item() == 0


# ------------------------------------------------------------------------------------------ embeddings / head / optimizer
@pytest.mark.parametrize("dtype", DTYPES)
def test_embeddings_fwd_bwd(dtype):
    torch.manual_seed(7)
    V, P, R, H, B, S = 300, 64, 2, 128, 3, 40
    word = (torch.randn(V, H, device="cuda") * 0.5).to(dtype)
    pos = (torch.randn(P, H, device="cuda") * 0.5).to(dtype)
    typ = (torch.randn(R, H, device="cuda") * 0.5).to(dtype)
    g = (1 + 0.1 * torch.randn(H, device="cuda")).to(dtype)
    b = (0.1 * torch.randn(H, device="cuda")).to(dtype)
    ids = torch.randint(0, V, (B * S,), device="cuda")
    ids[:5] = 0  # padding rows
    pids = torch.arange(S, device="cuda").repeat(B)
    tids = torch.randint(0, R, (B * S,), device="cuda")
    st = rng_state()
    y, pre, stats = C().embed_forward(ids, pids, tids, None, word, pos, typ, g, b, st, 1e-12, 0.0, 0, 0, None, 1, None)
    wf, pf, tf = word.float().requires_grad_(), pos.float().requires_grad_(), typ.float().requires_grad_()
    gf, bf = g.float().requires_grad_(), b.float().requires_grad_()
    e = F.embedding(ids, wf, padding_idx=0) + F.embedding(pids, pf) + F.embedding(tids, tf)
    ref = F.layer_norm(e, (H,), gf, bf, 1e-12)
    torch.testing.assert_close(y.float(), ref, rtol=2e-2, atol=2e-2)
    dy = torch.randn(B * S, H, device="cuda").to(dtype)
    ref.backward(dy.float())
    f32 = dict(device="cuda", dtype=torch.float32)
    grads = [torch.zeros(V, H, **f32), torch.zeros(P, H, **f32), torch.zeros(R, H, **f32), torch.zeros(H, **f32),
             torch.zeros(H, **f32)]
    C().embed_backward(dy, pre, stats, ids, pids, tids, g, grads, st, 0.0, 0, 0, 0, False, None, 1, None)
    for got, want in zip(grads, [wf.grad, pf.grad, tf.grad, gf.grad, bf.grad]):
        assert rel_err(got, want) < 2e-2
    assert grads[0][0].abs().max().item() == 0  # padding row


def test_embedding_table_gradients_are_a_deterministic_segmented_sum():
    """Hot rows ([CLS] / [SEP] once per sample, every position row B times) span many 32-row chunks of the sorted
    index list, cold rows lie inside one; the result equals an fp32 index_add, two runs are bit-identical, and a
    second call accumulates on top of the first."""
    torch.manual_seed(3)
    dtype = torch.float16
    V, P, R, H, B, S = 1000, 128, 2, 1024, 96, 128
    T = B * S
    ids = torch.randint(3, V, (B, S), device="cuda")
    ids[:, 0] = 1          # a row with B summands
    ids[:, 5:70:3] = 2     # a row with B * 22 summands
    ids[3, 100:] = 0       # padding_idx
    ids[7, 11] = V + 9     # outside the table: skipped
    ids = ids.reshape(-1)
    pids = torch.arange(S, device="cuda").repeat(B)
    tids = torch.zeros(T, dtype=torch.long, device="cuda")
    g = torch.ones(H, device="cuda", dtype=dtype)
    pre = torch.randn(T, H, device="cuda").to(dtype)
    stats = torch.stack([pre.float().mean(1), torch.rsqrt(pre.float().var(1, unbiased=False) + 1e-12)], 1).contiguous()
    dy = torch.randn(T, H, device="cuda").to(dtype)
    st = rng_state()
    f32 = dict(device="cuda", dtype=torch.float32)

    def run(grads, acc):
        return C().embed_backward(dy, pre, stats, ids, pids, tids, g, grads, st, 0.0, 0, 0, 0, acc, None, 1, None)

    mk = lambda: [torch.zeros(V, H, **f32), torch.zeros(P, H, **f32), torch.zeros(R, H, **f32), torch.zeros(H, **f32),
                  torch.zeros(H, **f32)]
    g1, g2 = mk(), mk()
    dx = run(g1, False).float()
    run(g2, False)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    valid = (ids != 0) & (ids < V)
    want_w = torch.zeros(V, H, **f32).index_add_(0, ids[valid], dx[valid])
    want_p = torch.zeros(P, H, **f32).index_add_(0, pids, dx)
    assert rel_err(g1[0], want_w) < 1e-5 and rel_err(g1[1], want_p) < 1e-5
    assert g1[0][0].abs().max().item() == 0
    run(g1, True)  # accumulate: twice the gradient
    assert rel_err(g1[0], 2 * want_w) < 1e-5 and rel_err(g1[1], 2 * want_p) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B", [37, 64, 65, 300, 1024])
def test_head_fwd_bwd(dtype, B):
    """B <= 64: one workgroup per direction; beyond, the logits are computed in parallel over 64-sample slices, the loss
    by one workgroup from the logits, and the backward's per-slice shares of dW / dbias are summed by a finishing launch."""
    torch.manual_seed(8)
    H, Cn = 256, 2
    pooled = torch.tanh(torch.randn(B, H, device="cuda")).to(dtype)
    W = torch.randn(Cn, H, device="cuda") * 0.1
    bias = torch.randn(Cn, device="cuda") * 0.1
    labels = torch.randint(0, Cn, (B,), device="cuda")
    st = rng_state()
    logits, loss = C().head_forward(pooled, W, bias, labels, st, 0.0, 0, 0, None)
    pf = pooled.float().requires_grad_()
    Wf, bf = W.clone().requires_grad_(), bias.clone().requires_grad_()
    rl = F.linear(pf, Wf, bf)
    rloss = F.cross_entropy(rl, labels)
    torch.testing.assert_close(logits, rl, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(loss, rloss, rtol=1e-5, atol=1e-5)
    (rloss * 3.0).backward()
    gW, gb = torch.zeros_like(W), torch.zeros_like(bias)
    dp = C().head_backward(torch.tensor(3.0, device="cuda"), None, pooled, W, logits, labels, gW, gb, st, 0.0, 0, 0, False, False, None)
    assert rel_err(dp, pf.grad) < 1e-2
    assert rel_err(gW, Wf.grad) < 1e-4
    assert rel_err(gb, bf.grad) < 1e-4
    # accumulate = True adds to what is there; two runs are bit-identical (fixed summation order)
    gW2, gb2 = gW.clone(), gb.clone()
    C().head_backward(torch.tensor(3.0, device="cuda"), None, pooled, W, logits, labels, gW2, gb2, st, 0.0, 0, 0, True, False, None)
    assert rel_err(gW2, 2 * Wf.grad) < 1e-4 and rel_err(gb2, 2 * bf.grad) < 1e-4
    gW3, gb3 = torch.zeros_like(W), torch.zeros_like(bias)
    C().head_backward(torch.tensor(3.0, device="cuda"), None, pooled, W, logits, labels, gW3, gb3, st, 0.0, 0, 0, False, False, None)
    assert torch.equal(gW3, gW) and torch.equal(gb3, gb)
    # dropout: forward and backward draw the same mask (recovered from the gradient wrt the pooled input of an explicit
    # O(1) upstream gradient of the logits - nothing that could underflow in 16 bits)
    lg, _ = C().head_forward(pooled, W, bias, labels, st, 0.25, 3, 7, None)
    gW4, gb4 = torch.zeros_like(W), torch.zeros_like(bias)
    dl = torch.zeros(B, Cn, device="cuda")
    dl[:, 0] = 1.0 + torch.rand(B, device="cuda")    # one class only: dl @ W cannot cancel
    dpd = C().head_backward(None, dl, pooled, W, lg, None, gW4, gb4, st, 0.25, 3, 7, False, False, None)
    full = dl @ W                                    # gradient without the mask; times 1/(1-p) where kept
    keep = (dpd.float().abs() > 0) | (full.abs() < 1e-6)
    assert 0.70 < keep.float().mean().item() < 0.80
    xd = pooled.float() * keep / 0.75
    torch.testing.assert_close(lg, F.linear(xd, W, bias), rtol=5e-3, atol=5e-3)
    assert rel_err(gW4, dl.t() @ xd) < 2e-3 and rel_err(gb4, dl.sum(0)) < 1e-4


def test_adamw_matches_hf_formula_and_skips_on_overflow():
    torch.manual_seed(9)
    n = 4096 + 64
    p0 = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda")
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    p16 = torch.empty(n, device="cuda", dtype=torch.float16)
    lr = torch.tensor([1e-3], device="cuda")
    step = torch.zeros(1, device="cuda", dtype=torch.int32)
    scale = 128.0
    state = torch.tensor([scale, 0.0, 0.0, 0.0], device="cuda")
    rp, rm, rv = p0.clone().double(), torch.zeros(n, device="cuda").double(), torch.zeros(n, device="cuda").double()
    b1, b2, eps, wd = 0.9, 0.999, 1e-6, 0.01
    for t in range(1, 4):
        C().adamw_step(p, m, v, g * scale, p16, lr, step, state, b1, b2, eps, wd, True, True)
        C().scaler_update(state, step, 2.0, 0.5, 2000, True)
        gd = g.double()
        rm = b1 * rm + (1 - b1) * gd
        rv = b2 * rv + (1 - b2) * gd * gd
        ss = 1e-3 * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        rp = rp - ss * rm / (rv.sqrt() + eps)
        rp = rp - 1e-3 * wd * rp
    torch.testing.assert_close(p.double(), rp, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(p16.float(), p, rtol=1e-3, atol=1e-3)
    assert step.item() == 3 and state[1].item() == 3
    # overflow: nothing moves, scale halves, step does not advance
    before = p.clone()
    gbad = g * scale
    gbad[17] = float("inf")
    C().adamw_step(p, m, v, gbad, p16, lr, step, state, b1, b2, eps, wd, True, True)
    C().scaler_update(state, step, 2.0, 0.5, 2000, True)
    assert torch.equal(p, before)
    assert state[0].item() == scale / 2 and step.item() == 3 and state[3].item() == 1.0


@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
def test_gemm_pingpong_large_k_repeatable(layout):
    """The 256x256 ping-pong kernel over many K-tiles, odd tile counts and ragged edges, run repeatedly:
    every run must be bitwise identical (no LDS race) and match fp32."""
    torch.manual_seed(10)
    dtype = torch.float16
    for (M, N, K) in [(512, 512, 4096), (700, 768, 1088), (256, 1024, 64), (1000, 264, 8192 + 64)]:
        if layout == "tn":
            A = (torch.randn(K, M - M % 8, device="cuda") * 0.3).to(dtype)
            B = (torch.randn(K, N, device="cuda") * 0.3).to(dtype)
            ref = A.float().t() @ B.float()
            run = lambda: C().op_gemm_tn(A, B, None, False, 2, 0)
        elif layout == "nt":
            A = (torch.randn(M, K, device="cuda") * 0.3).to(dtype)
            B = (torch.randn(N, K, device="cuda") * 0.3).to(dtype)
            ref = A.float() @ B.float().t()
            run = lambda: C().op_gemm("nt", A, B, 0, None, None, None, None, 0.0, 0, 0, 2)
        else:
            A = (torch.randn(M, K, device="cuda") * 0.3).to(dtype)
            B = (torch.randn(K, N, device="cuda") * 0.3).to(dtype)
            ref = A.float() @ B.float()
            run = lambda: C().op_gemm("nn", A, B, 0, None, None, None, None, 0.0, 0, 0, 2)
        first = run()
        assert rel_err(first, ref) < 3e-3, (layout, M, N, K)
        for _ in range(20):
            assert torch.equal(run(), first), (layout, M, N, K)


def test_gelu_epilogue_matches_erf_gelu_closely():
    """The epilogue evaluates exact-erf GELU with a rational erf (|err| < 1.5e-7): indistinguishable after
    rounding to 16 bits, also in the tails."""
    M, N, K = 256, 256, 64
    A = torch.zeros(M, K, device="cuda", dtype=torch.float16)
    W = torch.zeros(N, K, device="cuda", dtype=torch.float16)
    vals = torch.linspace(-9, 9, N, device="cuda")
    b = vals.to(torch.float16)
    u = torch.empty(M, N, device="cuda", dtype=torch.float16)
    g = C().op_gemm("nt", A, W, 2, b, None, u, None, 0.0, 0, 0, 0)
    ref = F.gelu(b.float()).to(torch.float16)
    assert (g[0].float() - ref.float()).abs().max().item() <= 1e-3
    assert torch.equal(g[0], g[-1])
    assert (u[0].float() - gelu_grad(b)).abs().max().item() <= 1e-3  # the saved derivative, tails included
    ones = torch.ones(M, K, device="cuda", dtype=torch.float16)
    eye = torch.zeros(K, N, device="cuda", dtype=torch.float16)
    eye[0] = 1.0  # dY @ eye = 1 everywhere -> output = gelu'(u)
    d = C().op_gemm("nn", ones, eye, 4, None, None, u, None, 0.0, 0, 0, 0)
    bf = b.float().requires_grad_()
    F.gelu(bf).sum().backward()
    assert (d[0].float() - bf.grad).abs().max().item() <= 2e-3


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_gemm_persistent_launch(layout, dtype):
    """At least 1024 full output tiles and an even K-tile count: the ping-pong kernel runs as 256 persistent
    workgroups that stage the next tile's first K-tiles under the epilogue and draw their third and later
    tiles from per-XCD ticket counters.  Every epilogue kind, tile counts that do not divide evenly, K of 2, 4
    and 6 K-tiles; values checked against fp32 and bit-identical between runs (every launch re-arms the counters)."""
    torch.manual_seed(13)
    for (M, N, K) in [(8192, 8448, 256), (4096, 16640, 384), (2048, 33024, 128)]:  # 1056 / 1040 / 1032 tiles
        A = (torch.randn(M, K, device="cuda") * 0.3).to(dtype)
        rt, at = tol(dtype)
        if layout == "nt":
            W = (torch.randn(N, K, device="cuda") * 0.1).to(dtype)
            b = torch.randn(N, device="cuda").to(dtype)
            res = torch.randn(M, N, device="cuda").to(dtype)
            pre = A.float() @ W.float().t()
            out = C().op_gemm("nt", A, W, 0, None, None, None, None, 0.0, 0, 0, 2)
            assert rel_err(out, pre) < 5e-3
            assert torch.equal(out, C().op_gemm("nt", A, W, 0, None, None, None, None, 0.0, 0, 0, 2))
            aux = torch.empty(M, N, device="cuda", dtype=dtype)
            act = C().op_gemm("nt", A, W, 2, b, None, aux, None, 0.0, 0, 0, 2)
            assert rel_err(act, F.gelu(pre + b.float())) < 1e-2
            assert rel_err(aux, gelu_grad(pre + b.float())) < 1e-2
            z = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(), 0.0, 0, 0, 2)
            assert rel_err(z, pre + b.float() + res.float()) < 1e-2
            # dropout: the same mask as the one-tile-per-workgroup 128x128 kernel (counters depend on (m, n) only)
            d2 = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(3, 1), 0.1, 2, 5, 2)
            d0 = C().op_gemm("nt", A, W, 3, b, res, None, rng_state(3, 1), 0.1, 2, 5, 0)
            torch.testing.assert_close(d2.float(), d0.float(), rtol=rt, atol=at)
        else:
            W = (torch.randn(K, N, device="cuda") * 0.1).to(dtype)
            ref = A.float() @ W.float()
            out = C().op_gemm("nn", A, W, 0, None, None, None, None, 0.0, 0, 0, 2)
            assert rel_err(out, ref) < 5e-3
            dg = gelu_grad(torch.randn(M, N, device="cuda")).to(dtype)
            cs = torch.zeros(N, device="cuda")
            du = C().op_gemm("nn", A, W, 4, None, None, dg, None, 0.0, 0, 0, 2, cs)
            assert rel_err(du, ref * dg.float()) < 1e-2
            torch.testing.assert_close(cs, du.float().sum(0), rtol=5e-3, atol=0.5)
            res = torch.randn(M, N, device="cuda").to(dtype)
            dx = C().op_gemm("nn", A, W, 5, None, res, None, None, 0.0, 0, 0, 2)
            assert rel_err(dx, ref + res.float()) < 1e-2


# ------------------------------------------------------------------------------ large token counts
def test_kernels_at_131072_tokens():
    """Per-GPU batch 1024 x 128 tokens (what the 288 GB of HBM are sized for): every index computation of the
    hot kernels at offsets beyond 2^30 bytes - checked on the last rows / samples against fp32 PyTorch."""
    torch.manual_seed(12)
    dtype = torch.float16
    T, H, F4, S, A = 131072, 1024, 4096, 128, 16
    B = T // S
    tail = slice(T - 300, T)
    mid = slice(T // 2 + 5, T // 2 + 133)
    x = (torch.randn(T, H, device="cuda") * 0.5).to(dtype)
    w1 = (torch.randn(F4, H, device="cuda") * 0.05).to(dtype)
    b1 = torch.randn(F4, device="cuda").to(dtype)
    aux = torch.empty(T, F4, device="cuda", dtype=dtype)
    act = C().op_gemm("nt", x, w1, 2, b1, None, aux, None, 0.0, 0, 0, -1)
    for rows in (tail, mid):
        pre = x[rows].float() @ w1.float().t() + b1.float()
        assert rel_err(act[rows], F.gelu(pre)) < 5e-3
        assert rel_err(aux[rows], gelu_grad(pre)) < 5e-3
    # data gradient with the saved derivative (NN, 4096-wide output) and with a residual (NN, 1024-wide output)
    w2 = (torch.randn(H, F4, device="cuda") * 0.05).to(dtype)  # [out=H, in=4H]
    du = C().op_gemm("nn", x, w2, 4, None, None, aux, None, 0.0, 0, 0, -1)
    for rows in (tail, mid):
        assert rel_err(du[rows], (x[rows].float() @ w2.float()) * aux[rows].float()) < 5e-3
    dx = C().op_gemm("nn", act, w1, 5, None, x, None, None, 0.0, 0, 0, -1)
    for rows in (tail, mid):
        assert rel_err(dx[rows], act[rows].float() @ w1.float() + x[rows].float()) < 5e-3
    # weight gradient: K = 131072 rows
    G = C().op_gemm_tn(act, x, None, False, -1, 0)
    ref = torch.zeros(F4, H, device="cuda")
    for c in range(0, T, 16384):
        ref += act[c:c + 16384].float().t() @ x[c:c + 16384].float()
    assert rel_err(G, ref) < 3e-3
    # LayerNorm forward / backward on the last rows
    g, b = torch.randn(H, device="cuda").to(dtype), torch.randn(H, device="cuda").to(dtype)
    y, stats = C().op_layernorm_fwd(x, g, b, 1e-12)
    xr = x[tail].float().requires_grad_()
    yr = F.layer_norm(xr, (H,), g.float(), b.float(), 1e-12)
    assert rel_err(y[tail], yr) < 5e-3
    dy = (torch.randn(T, H, device="cuda") * 0.5).to(dtype)
    dxl = C().op_layernorm_bwd(dy, x, stats, g, None, 0.0, 0, 0)[0]
    yr.backward(dy[tail].float())
    assert rel_err(dxl[tail], xr.grad) < 5e-3
    del act, aux, du, dx, y, dxl
    # attention on the last samples of 1024
    qkv = (torch.randn(T, 3 * H, device="cuda") * 0.5).to(dtype)
    mask = torch.ones(B, S, device="cuda", dtype=torch.int32)
    mask[-1, 100:] = 0
    kv_len = mask.sum(1).to(torch.int32)
    ctx, lse = C().op_attn_fwd(qkv, mask, kv_len, B, S, A, None, 0.0, 0, 0)
    nb = 3
    qf = qkv[-nb * S:].float().requires_grad_()
    cref, _ = ref_attention(qf, mask[-nb:], nb, S, A)
    vm = mask[-nb:].reshape(nb * S, 1).float()
    assert rel_err(ctx[-nb * S:].float() * vm, cref * vm) < 5e-3
    dctx = (torch.randn(T, H, device="cuda") * 0.5).to(dtype) * mask.view(T, 1).to(dtype)
    (cref * dctx[-nb * S:].float()).sum().backward()
    dbias = torch.zeros(3 * H, device="cuda") if C().attn_bwd_fuses_dbias(S) else None
    dqkv = C().op_attn_bwd(qkv, mask, kv_len, ctx, dctx, lse, B, S, A, None, 0.0, 0, 0, dbias)
    assert rel_err(dqkv[-nb * S:], qf.grad) < 6e-3
    if dbias is not None:
        torch.testing.assert_close(dbias, dqkv.float().sum(0), rtol=5e-3, atol=0.5)


def test_embedding_indices_outside_their_tables_are_clamped_not_dereferenced():
    """Token / position / token-type ids come from user data.  An id outside its table (a tokenizer with a larger
    vocabulary, a corrupted batch) must not become a wild address: the forward lookup clamps it into the table, the
    gradient scatter skips it."""
    torch.manual_seed(14)
    dtype = torch.float16
    V, P, R, H, T = 300, 64, 2, 128, 96
    word = (torch.randn(V, H, device="cuda") * 0.5).to(dtype)
    pos = (torch.randn(P, H, device="cuda") * 0.5).to(dtype)
    typ = (torch.randn(R, H, device="cuda") * 0.5).to(dtype)
    g = torch.ones(H, device="cuda", dtype=dtype)
    b = torch.zeros(H, device="cuda", dtype=dtype)
    ids = torch.randint(1, V, (T,), device="cuda")
    pids = torch.arange(T, device="cuda") % P
    tids = torch.randint(0, R, (T,), device="cuda")
    bad_ids, bad_p, bad_t = ids.clone(), pids.clone(), tids.clone()
    bad_ids[3], bad_ids[10], bad_p[5], bad_t[7], bad_t[8] = V + 5, -4, P + 100, 28995, -1
    st = rng_state()
    y_bad, pre_bad, stats_bad = C().embed_forward(bad_ids, bad_p, bad_t, None, word, pos, typ, g, b, st, 1e-12, 0.0, 0, 0, None, 1, None)
    y_ref, _, _ = C().embed_forward(bad_ids.clamp(0, V - 1), bad_p.clamp(0, P - 1), bad_t.clamp(0, R - 1), None, word,
                                    pos, typ, g, b, st, 1e-12, 0.0, 0, 0, None, 1, None)
    assert torch.equal(y_bad, y_ref)
    f32 = dict(device="cuda", dtype=torch.float32)
    grads = [torch.zeros(V, H, **f32), torch.zeros(P, H, **f32), torch.zeros(R, H, **f32), torch.zeros(H, **f32),
             torch.zeros(H, **f32)]
    dy = torch.randn(T, H, device="cuda").to(dtype)
    C().embed_backward(dy, pre_bad, stats_bad, bad_ids, bad_p, bad_t, g, grads, st, 0.0, 0, 0, 0, False, None, 1, None)
    torch.cuda.synchronize()
    assert all(torch.isfinite(x).all() for x in grads)
    # rows of valid tokens received their gradient; nothing was written for the out-of-table word / position rows
    assert grads[0][ids[0]].abs().sum() > 0 and grads[1][pids[0]].abs().sum() > 0


@pytest.mark.parametrize("smax", [96, 320])
def test_attention_unpadded_batch_equals_padded(smax):
    """cu_seqlens: the samples' rows are packed back to back; forward output, dQ/dK/dV of the real tokens equal the
    padded [B, S] run with key lengths (both the fused S <= 128 backward and the two-kernel path), dropout on."""
    torch.manual_seed(5)
    dtype = torch.float16
    A, B = 4, 5
    H = A * 64
    lens = [smax, 17, smax // 2, 64, 33]
    S = max(lens)
    qkv_pad = torch.zeros(B, S, 3 * H, device="cuda", dtype=dtype)
    dctx_pad = torch.zeros(B, S, H, device="cuda", dtype=dtype)
    mask = torch.zeros(B, S, dtype=torch.int32, device="cuda")
    for b, n in enumerate(lens):
        qkv_pad[b, :n] = (torch.randn(n, 3 * H, device="cuda") * 0.7).to(dtype)
        dctx_pad[b, :n] = torch.randn(n, H, device="cuda").to(dtype)
        mask[b, :n] = 1
    kv_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    rows = torch.cat([torch.arange(n) + b * S for b, n in enumerate(lens)]).cuda()
    qkv = qkv_pad.reshape(B * S, -1)[rows].contiguous()
    dctx = dctx_pad.reshape(B * S, -1)[rows].contiguous()
    sids = torch.arange(70, 70 + B, dtype=torch.int32, device="cuda")
    st = rng_state()
    dummy = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx_p, lse_p = C().op_attn_fwd(qkv_pad.reshape(B * S, -1), mask, kv_len, B, S, A, st, 0.2, 3, 9, sids)
    ctx_u, lse_u = C().op_attn_fwd(qkv, dummy, dummy, B, S, A, st, 0.2, 3, 9, sids, cu)
    torch.testing.assert_close(ctx_u, ctx_p[rows], rtol=1e-3, atol=1e-3)
    dq_p = C().op_attn_bwd(qkv_pad.reshape(B * S, -1), mask, kv_len, ctx_p, dctx_pad.reshape(B * S, -1), lse_p, B, S, A, st, 0.2, 3, 9,
                           None, sids)
    dq_u = C().op_attn_bwd(qkv, dummy, dummy, ctx_u, dctx, lse_u, B, S, A, st, 0.2, 3, 9, None, sids, cu)
    assert rel_err(dq_u.float(), dq_p[rows].float()) < 2e-3


def test_gemm_tn_wgrad_over_more_tokens_than_one_operand_window():
    """A weight gradient over 278528 tokens x 4096 columns: the 16-bit operand is 2.3 GB, beyond the 32-bit byte
    offsets the kernels address a window with - the binding walks the token range in chunks that accumulate."""
    torch.manual_seed(0)
    T, M, N = 278528, 4096, 256
    dY = (torch.randn(T, M, device="cuda") * 0.05).to(torch.float16)
    X = (torch.randn(T, N, device="cuda") * 0.05).to(torch.float16)
    G = C().op_gemm_tn(dY, X, None, False, -1, 0)
    ref = torch.zeros(M, N, device="cuda")
    for t0 in range(0, T, 32768):  # fp32 reference in slices (a full fp32 copy of dY would be 4.6 GB)
        ref += dY[t0:t0 + 32768].float().t() @ X[t0:t0 + 32768].float()
    assert rel_err(G, ref) < 2e-3
    G2 = C().op_gemm_tn(dY, X, G.clone(), True, -1, 0)
    assert rel_err(G2, 2 * ref) < 2e-3
