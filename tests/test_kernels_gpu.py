"""GPU: each HIP kernel (through the C ABI) against a plain PyTorch fp32 reference
of the same op on the same (bf16-rounded) inputs.  Asymmetric random data so that
operand / output transposes cannot pass."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (1060 * 2, 768, 768), (1025, 3072, 768), (333, 256, 3072)])
def test_gemm_nt_epilogue(M, N, K):
    from ifseg_amd import hip
    dev = _dev()
    x, w = _rand((M, K), dev, 1, 0.5), _rand((N, K), dev, 2, 0.5)
    bias, res = _rand((N,), dev, 3), _rand((M, N), dev, 4)
    out = hip.linear_fwd(x, w, bias, alpha=0.37, alpha_ncols=N // 2 if N >= 256 else -1, resid=res)
    ref = x.float() @ w.float().t() + bias.float()
    nc = N // 2 if N >= 256 else N
    ref[:, :nc] *= 0.37
    ref = ref + res.float()
    torch.cuda.synchronize()
    assert _rel(out, ref) < 6e-3, _rel(out, ref)
    # plain
    out2 = hip.linear_fwd(x, w)
    assert _rel(out2, x.float() @ w.float().t()) < 6e-3


@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (1060 * 2, 768, 3072), (1025, 2304, 768)])
def test_gemm_nn_dx(M, N, K):
    from ifseg_amd import hip
    dev = _dev()
    dy, w, res = _rand((M, N), dev, 5, 0.5), _rand((N, K), dev, 6, 0.5), _rand((M, K), dev, 7)
    out = hip.linear_dx(dy, w, resid=res)
    ref = dy.float() @ w.float() + res.float()
    assert _rel(out, ref) < 6e-3, _rel(out, ref)


@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (1060 * 2, 768, 3072), (1025 * 3, 3072, 768), (8200, 768, 768)])
def test_gemm_tn_dw(M, N, K):
    from ifseg_amd import hip
    dev = _dev()
    dy, x = _rand((M, N), dev, 8, 0.5), _rand((M, K), dev, 9, 0.5)
    ref = dy.float().t() @ x.float()
    out = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    hip.linear_dw(dy, x, out)
    assert _rel(out, ref) < 6e-3, _rel(out, ref)
    out32 = torch.ones(N, K, dtype=torch.float32, device=dev)
    hip.linear_dw(dy, x, out32, accumulate=True)
    assert _rel(out32, ref + 1.0) < 1e-4, _rel(out32, ref + 1.0)


@pytest.mark.parametrize("Cin,Cout,KH,stride,H", [(64, 64, 1, 1, 16), (64, 64, 3, 1, 16), (128, 128, 3, 2, 16),
                                                  (256, 512, 1, 2, 16), (256, 1024, 1, 1, 8)])
def test_conv_nhwc(Cin, Cout, KH, stride, H):
    from ifseg_amd import hip
    import torch.nn.functional as F
    dev = _dev()
    B, W = 2, H + 4
    pad = KH // 2
    x = _rand((B, H, W, Cin), dev, 10)
    w = _rand((Cout, KH, KH, Cin), dev, 11, 1.0 / math.sqrt(Cin * KH * KH))
    shift = _rand((Cout,), dev, 12)
    OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KH) // stride + 1
    res = _rand((B, OH, OW, Cout), dev, 13)
    out = torch.empty(B, OH, OW, Cout, dtype=torch.bfloat16, device=dev)
    hip.conv2d_nhwc(x, w, shift, res, out, B, H, W, Cin, Cout, KH, KH, stride, pad, True)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), shift.float(), stride, pad)
    ref = torch.relu(ref + res.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert _rel(out, ref) < 6e-3, _rel(out, ref)


# ----------------------------------------------------------------------------- attention
def _dense_rel(H, T, S, P, gcode, code_bias, rel2d, rel1d, relx):
    """independent (index-arithmetic free) construction of the rel-pos bias [H,T,S]"""
    bias = torch.zeros(H, T, S)
    Lt = T - P
    gc = gcode.cpu()
    idx = gc[:, None] - gc[None, :] + code_bias
    bias[:, :P, :P] = rel2d.cpu()[:, idx]
    if Lt > 0:
        t = torch.arange(Lt)
        bias[:, P:, P:] = rel1d.cpu()[:, t[:, None] - t[None, :] + Lt - 1]
        bias[:, :P, P:] = relx.cpu()[:, 0][:, None, None]
        bias[:, P:, :P] = relx.cpu()[:, 1][:, None, None]
    return bias


def _causal_mask(T, S, P):
    i = torch.arange(T)[:, None]
    j = torch.arange(S)[None, :]
    grid_key = j < P
    masked = (grid_key & ((i >= P) | (j > i))) | (~grid_key & (i >= P) & (j > i))
    return masked


def _attn_ref(q, k, v, pq, pk, bias, mask):
    B, T, C = q.shape
    H = C // 64
    S = k.shape[1]
    qh = q.float().view(B, T, H, 64).transpose(1, 2)
    kh = k.float().view(B, S, H, 64).transpose(1, 2)
    vh = v.float().view(B, S, H, 64).transpose(1, 2)
    s = qh @ kh.transpose(2, 3)
    if pq is not None:
        s = s + (pq.float().view(T, H, 64).transpose(0, 1) @ pk.float().view(S, H, 64).transpose(0, 1).transpose(1, 2))
    if bias is not None:
        s = s + bias.to(s.device)
    if mask is not None:
        s = s.masked_fill(mask.to(s.device), float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ vh).transpose(1, 2).reshape(B, T, C)
    return o, torch.logsumexp(s, -1)


def _grid_codes(gh, gw):
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    code = (ys * (2 * gw - 1) + xs).reshape(-1).int()
    return code, (gh - 1) * (2 * gw - 1) + (gw - 1), (2 * gh - 1) * (2 * gw - 1)


@pytest.mark.parametrize("case", ["cross", "enc_rel", "dec_causal", "dec_full", "dense_nopos", "big_enc"])
def test_attn_fwd(case):
    from ifseg_amd import hip
    dev = _dev()
    H = 2
    B = 2
    rel = None
    causal = False
    dense = None
    P = None
    use_pos = True
    if case == "cross":
        T, S = 193, 292
    elif case == "enc_rel":
        gh, gw = 8, 16
        P, Lt = 128, 36
        T = S = P + Lt
    elif case in ("dec_causal", "dec_full"):
        gh, gw = 8, 8
        P, Lt = 64, 1
        T = S = P + Lt
        causal = case == "dec_causal"
    elif case == "dense_nopos":
        T, S = 100, 130
        use_pos = False
    elif case == "big_enc":
        gh, gw = 32, 32
        P, Lt = 1024, 36
        T = S = P + Lt
        H = 12
        B = 1
    C = H * 64
    q, k, v = _rand((B, T, C), dev, 20, 0.35), _rand((B, S, C), dev, 21), _rand((B, S, C), dev, 22)
    pq, pk = (_rand((T, C), dev, 23, 0.35), _rand((S, C), dev, 24)) if use_pos else (None, None)
    bias = None
    if P is not None:
        gcode, code_bias, n2d = _grid_codes(gh, gw)
        g = torch.Generator().manual_seed(30)
        rel2d = torch.randn(H, n2d, generator=g)
        rel1d = torch.randn(H, 2 * Lt - 1, generator=g)
        relx = torch.randn(H, 2, generator=g)
        rel = hip.RelBias(P, gcode.to(dev), code_bias, rel2d.to(dev), rel1d.to(dev), relx.to(dev))
        bias = _dense_rel(H, T, S, P, gcode.long(), code_bias, rel2d, rel1d, relx)
    if case == "dense_nopos":
        g = torch.Generator().manual_seed(31)
        bias = torch.randn(H, T, S, generator=g)
        dense = bias.to(dev).contiguous()
    mask = _causal_mask(T, S, P) if causal else None
    out = torch.zeros(B, T, C, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, H, T, dtype=torch.float32, device=dev)
    hip.attn_fwd(q, k, v, pq, pk, out, lse, B, H, T, S, rel=rel, causal=causal, dense_bias=dense)
    torch.cuda.synchronize()
    ref_o, ref_lse = _attn_ref(q, k, v, pq, pk, bias, mask)
    e_o, e_l = _rel(out, ref_o), (lse - ref_lse).abs().max().item()
    print(case, "attn fwd rel err", e_o, "lse max abs", e_l)
    assert e_o < 1e-2, e_o
    assert e_l < 2e-3, e_l
