"""Training on a feature grid other than the trained one: the parameter-side maps and their adjoints.

When the ResNet grid (h, w) differs from the grid the position tables were trained on, the reference
  * bilinear-resizes the image rows of `embed_image_positions` (encoder_module.py:356-372, only when P > orig) and of
    `embed_seg_positions` (decoder_module.py:541-550) to the new grid, and
  * builds every layer's relative-position bias on the ORIGINAL grid -- a gather of the bucket table into [H, P0, P0] -- and
    resizes it twice, over the key grid and over the query grid (encoder_module.py:798-809; decoder_module.py:603-627 with
    the bos row / column passed through),
all under autograd.  These are linear maps of PARAMETER-sized tensors ([H, P, P] per layer), evaluated once per step and
layer.  Here they are written with device torch ops and differentiated with torch.autograd: the hot path (grid == trained
grid, every shipped script) never comes here, and this path only has to be correct.  No activation
is touched: the engine hands the resulting dense bias to the same batch-inner attention kernels as the fast path and brings
`sum_b dS` back for the adjoint.  Nothing runs on the CPU."""
import torch
import torch.nn.functional as F


def resize_hw(t, src_hw, dst_hw):
    """bilinear resize (align_corners=False) of the trailing, flattened grid dimension: [..., h*w] -> [..., h'*w']"""
    lead = t.shape[:-1]
    t4 = t.reshape(1, -1, src_hw[0], src_hw[1])
    t4 = F.interpolate(t4, size=tuple(dst_hw), mode="bilinear")
    return t4.reshape(*lead, dst_hw[0] * dst_hw[1])


def grid_ids(h, w, bucket, device):
    """position id of grid cell (y, x): x + y * bucket + 1 (encoder_module.py:339-342, decoder_module.py:541-542)"""
    return (torch.arange(w, device=device)[None, :] + torch.arange(h, device=device)[:, None] * bucket + 1).reshape(-1)


def encoder_rel_bias(tok_table, img_table, token_rp_bucket, image_rp_bucket, hw, orig, bucket, L):
    """[H, T, T] (T = h*w + L, image tokens first): the image block is the bucket-table gather on the orig x orig grid resized
    over keys, then over queries (encoder_module.py:798-809); the text block the token table's gather (:790-797); zero
    elsewhere.  tok_table [Nt, H], img_table [Ni, H] fp32 (differentiable)."""
    h, w = hw
    P = h * w
    H = img_table.shape[1]
    ids0 = grid_ids(orig, orig, bucket, img_table.device)
    rp = image_rp_bucket[ids0][:, ids0]                                   # [P0, P0]
    img = img_table[rp].permute(2, 0, 1)                                  # [H, P0, P0]
    if (h, w) != (orig, orig):
        img = resize_hw(img, (orig, orig), (h, w))                                          # keys
        img = resize_hw(img.transpose(1, 2), (orig, orig), (h, w)).transpose(1, 2)          # queries
    tok = tok_table[token_rp_bucket[:L, :L]].permute(2, 0, 1)             # [H, L, L]
    out = img.new_zeros(H, P + L, P + L)
    out[:, :P, :P] = img
    out[:, P:, P:] = tok
    return out


def decoder_rel_bias(seg_table, seg_rp_bucket, hw, sb):
    """[H, P + 1, P + 1] in the ENGINE's token order [patches ..., bos] (the reference's is [bos, patches ...]): the gather on
    the sb x sb grid with the bos slot, resized over rows then columns, bos row / column passed through
    (decoder_module.py:327-333, 603-627)."""
    h, w = hw
    P = h * w
    rel = seg_table[seg_rp_bucket].permute(2, 0, 1)                       # [H, N0 + 1, N0 + 1], index 0 = bos
    if (h, w) != (sb, sb):
        t = rel.transpose(1, 2)
        t = torch.cat([t[..., :1], resize_hw(t[..., 1:], (sb, sb), (h, w))], dim=-1)
        t = t.transpose(1, 2)
        rel = torch.cat([t[..., :1], resize_hw(t[..., 1:], (sb, sb), (h, w))], dim=-1)      # [H, P + 1, P + 1]
    perm = torch.cat([torch.arange(1, P + 1, device=rel.device), torch.zeros(1, dtype=torch.long, device=rel.device)])
    return rel[:, perm][:, :, perm]


def causal_mask(P, device):
    """True where the engine's order [patches ..., bos] hides (query i, key j): a patch sees the patches up to itself and the
    bos slot, the bos slot only itself (decoder_module.py:592-600: buffered_future_mask in the reference's order)"""
    i = torch.arange(P + 1, device=device)
    ref = torch.cat([i[:P] + 1, i[P:] * 0])          # reference position of an engine index: patch k -> k + 1, bos -> 0
    return ref[None, :] > ref[:, None]


def abs_bias(pq, pk, H):
    """[H, T, S] = pos_q[i] . pos_k[j] per head (encoder_module.py:757-771): bf16 operands, fp32 products"""
    T, S = pq.shape[0], pk.shape[0]
    return pq.float().view(T, H, -1).transpose(0, 1) @ pk.float().view(S, H, -1).permute(1, 2, 0)


def rows_resize_adjoint(d_rows, src_hw, dst_hw):
    """adjoint of rows -> resize_hw(rows^T, src, dst)^T: d_rows [P, C] -> [P0, C] fp32"""
    P0 = src_hw[0] * src_hw[1]
    with torch.enable_grad():
        old = torch.zeros(P0, d_rows.shape[1], dtype=torch.float32, device=d_rows.device, requires_grad=True)
        y = resize_hw(old.t(), src_hw, dst_hw).t()
        (g,) = torch.autograd.grad(y, old, d_rows.float())
    return g


def table_grads(fn, tables, d_bias):
    """vjp of a (linear) bias map: fn(*tables) -> [H, T, T]; d_bias the gradient of its output -> gradients of the tables"""
    with torch.enable_grad():
        ts = [t.detach().float().requires_grad_(True) for t in tables]
        out = fn(*ts)
        return torch.autograd.grad(out, ts, d_bias)
