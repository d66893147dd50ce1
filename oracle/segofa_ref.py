"""CPU oracle for the SegOFA hot path (fp32, plain PyTorch, functional).

TEST INFRASTRUCTURE -- NOT THE PRODUCT.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product
(``ifseg_amd``) never imports it and fails loudly when its HIP library is
missing.

This is a from-scratch *restatement* of the reference algorithm (alinlab/ifseg,
``/root/reference``) operating directly on a ``state_dict`` with the reference's
own key names, batch-first ``[B, T, C]`` layout, batch-invariant position biases
computed once.  Every function cites the reference lines it follows.

Pinning: ``oracle/gen_golden.py`` imports the reference's own
``models/segofa/*.py`` in the build container (through ``oracle/_refshim.py``),
fills both with the same procedural weights and checks logits / loss / grads of
this restatement against it (max-abs <= 1e-5 fp32) on the small fixture config
and on a P != orig (bilinear-resize) eval case, then writes the golden vectors
under ``tests/golden/`` that ``tests/test_oracle_golden.py`` re-checks anywhere.
The reference itself ships no tests / golden vectors for this path (SURVEY.md
section 4), so these reference-generated vectors are the pin.
"""
import math
import zlib
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

PAD, BOS, EOS = 1, 0, 2  # fairseq Dictionary specials (data/dictionary.py)


@dataclass
class SegOFAConfig:
    """Subset of the reference's args that the hot path reads.

    Defaults = ``segofa_base`` + run_scripts/IFSeg/coco_unseen.sh
    (models/segofa/segofa.py:351-431)."""
    embed_dim: int = 768
    ffn_dim: int = 3072
    heads: int = 12
    enc_layers: int = 6
    dec_layers: int = 6
    resnet_layers: Tuple[int, ...] = (3, 4, 23)
    num_seg_tokens: int = 15
    vocab_size: int = 59458            # len(dict) - num_seg_tokens (unify_transformer.py:402)
    patch_image_size: int = 512
    orig_patch_image_size: int = 512
    image_bucket_size: int = 42
    token_bucket_size: int = 256
    attn_scale_factor: float = 2.0
    max_source_positions: int = 1024
    max_target_positions: int = 1024
    code_image_size: int = 128

    @property
    def head_dim(self):
        return self.embed_dim // self.heads

    @property
    def seg_bucket_size(self):          # decoder_module.py:191
        return self.patch_image_size // 16

    @property
    def seg_id_offset(self):            # index of <seg_0> == vocab_size - 1
        return self.vocab_size - 1


def base_config(**kw):
    return SegOFAConfig(**kw)


def large_config(**kw):
    d = dict(embed_dim=1024, ffn_dim=4096, heads=16, enc_layers=12, dec_layers=12,
             resnet_layers=(3, 8, 36))
    d.update(kw)
    return SegOFAConfig(**d)


def fixture_config(**kw):
    """Small configuration used by the golden fixtures (head_dim stays 64 and the
    8x8 feature grid gives P = 64 tokens, so the HIP kernels run the very same case)."""
    d = dict(embed_dim=128, ffn_dim=256, heads=2, enc_layers=2, dec_layers=2,
             resnet_layers=(3, 4, 6), num_seg_tokens=5, vocab_size=101,
             patch_image_size=128, orig_patch_image_size=128)
    d.update(kw)
    return SegOFAConfig(**d)


# --------------------------------------------------------------------------- #
# bucket tables (unify_transformer.py:55-88; duplicated in encoder/decoder)    #
# --------------------------------------------------------------------------- #
def make_token_bucket_position(bucket_size, max_position=1024):
    """unify_transformer.py:55-68."""
    ctx = torch.arange(max_position, dtype=torch.long)[:, None]
    mem = torch.arange(max_position, dtype=torch.long)[None, :]
    rel = ctx - mem
    sign = torch.sign(rel)
    mid = bucket_size // 2
    abs_pos = torch.where((rel < mid) & (rel > -mid), mid - 1, torch.abs(rel))
    log_pos = torch.ceil(torch.log(abs_pos / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    log_pos = log_pos.int()
    bucket = torch.where(abs_pos.le(mid), rel, log_pos * sign).long()
    return bucket + bucket_size - 1


def make_image_bucket_position(bucket_size, num_relative_distance):
    """unify_transformer.py:71-88."""
    ch = torch.arange(bucket_size)
    cw = torch.arange(bucket_size)
    coords = torch.stack(torch.meshgrid([ch, cw], indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = flat[:, :, None] - flat[:, None, :]
    rel = rel.permute(1, 2, 0).contiguous()
    rel[:, :, 0] += bucket_size - 1
    rel[:, :, 1] += bucket_size - 1
    rel[:, :, 0] *= 2 * bucket_size - 1
    idx = torch.zeros(size=(bucket_size * bucket_size + 1,) * 2, dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = num_relative_distance - 3
    idx[0:, 0] = num_relative_distance - 2
    idx[0, 0] = num_relative_distance - 1
    return idx


# --------------------------------------------------------------------------- #
# state-dict specification + procedural weights                               #
# --------------------------------------------------------------------------- #
def state_dict_spec(cfg: SegOFAConfig) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """name -> (shape, kind).  Floating tensors only (the int64 bucket buffers and
    ``version`` are derived, see ``derived_buffers``).  Mirrors the reference's
    module tree: encoder_module.py:117-295, decoder_module.py:114-274,
    unify_transformer_layer.py:125-172,312-387, unify_multihead_attention.py:44-90,
    resnet.py:140-213, frozen_bn.py:27-34."""
    C, Fd, H = cfg.embed_dim, cfg.ffn_dim, cfg.heads
    s = OrderedDict()

    def lin(p, o, i, bias=True):
        s[p + ".weight"] = ((o, i), "linear_w")
        if bias:
            s[p + ".bias"] = ((o,), "bias")

    def ln(p, n):
        s[p + ".weight"] = ((n,), "ln_w")
        s[p + ".bias"] = ((n,), "ln_b")

    def bn(p, n, last=False):
        s[p + ".weight"] = ((n,), "bn_w_last" if last else "bn_w")
        s[p + ".bias"] = ((n,), "bn_b")
        s[p + ".running_mean"] = ((n,), "bn_mean")
        s[p + ".running_var"] = ((n,), "bn_var")

    def mha(p):
        s[p + ".c_attn"] = ((H,), "gain")
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "." + n, C, C)

    # ---- encoder ----
    e = "encoder."
    s[e + "embed_tokens.weight"] = ((cfg.vocab_size, C), "embed")
    s[e + "seg_embed_tokens.weight"] = ((cfg.num_seg_tokens, C), "seg_embed")
    s[e + "embed_tokens_bag.weight"] = ((cfg.vocab_size, C), "alias:encoder.embed_tokens.weight")
    ln(e + "layernorm_embedding", C)
    s[e + "type_embedding.weight"] = ((2, C), "embed")
    r = e + "embed_images."
    s[r + "conv1.weight"] = ((64, 3, 7, 7), "conv")
    bn(r + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256), cfg.resnet_layers), start=1):
        for b in range(blocks):
            p = "%slayer%d.%d." % (r, li, b)
            s[p + "conv1.weight"] = ((planes, inplanes, 1, 1), "conv")
            bn(p + "bn1", planes)
            s[p + "conv2.weight"] = ((planes, planes, 3, 3), "conv")
            bn(p + "bn2", planes)
            s[p + "conv3.weight"] = ((planes * 4, planes, 1, 1), "conv")
            bn(p + "bn3", planes * 4, last=True)
            if b == 0:
                s[p + "downsample.0.weight"] = ((planes * 4, inplanes, 1, 1), "conv")
                bn(p + "downsample.1", planes * 4)
            inplanes = planes * 4
    lin(e + "image_proj", C, 1024)
    ln(e + "patch_layernorm_embedding", C)
    s[e + "embed_positions.weight"] = ((cfg.max_source_positions + 2, C), "embed")
    s[e + "embed_image_positions.weight"] = ((cfg.image_bucket_size ** 2 + 1, C), "embed")
    ln(e + "pos_ln", C)
    ln(e + "image_pos_ln", C)
    lin(e + "pos_q_linear", C, C)
    lin(e + "pos_k_linear", C, C)
    for i in range(cfg.enc_layers):
        p = "%slayers.%d." % (e, i)
        mha(p + "self_attn")
        ln(p + "self_attn_layer_norm", C)
        lin(p + "fc1", Fd, C)
        lin(p + "fc2", C, Fd)
        ln(p + "attn_ln", C)
        ln(p + "ffn_layernorm", Fd)
        ln(p + "final_layer_norm", C)
    ln(e + "layer_norm", C)
    n_tok = 2 * cfg.token_bucket_size - 1
    n_img = (2 * cfg.image_bucket_size - 1) ** 2 + 3
    for i in range(cfg.enc_layers):
        s["%stoken_rel_pos_table_list.%d.weight" % (e, i)] = ((n_tok, H), "rel")
    for i in range(cfg.enc_layers):
        s["%simage_rel_pos_table_list.%d.weight" % (e, i)] = ((n_img, H), "rel")

    # ---- decoder ----
    d = "decoder."
    s[d + "seg_embed_tokens.weight"] = ((cfg.num_seg_tokens, C), "alias:encoder.seg_embed_tokens.weight")
    s[d + "seg_projection.weight"] = ((cfg.num_seg_tokens, C), "alias:encoder.seg_embed_tokens.weight")
    s[d + "embed_tokens.weight"] = ((cfg.vocab_size, C), "alias:encoder.embed_tokens.weight")
    ln(d + "layernorm_embedding", C)
    s[d + "embed_positions.weight"] = ((cfg.max_target_positions + 2, C), "embed")
    s[d + "embed_image_positions.weight"] = ((cfg.image_bucket_size ** 2 + 1, C), "embed")
    s[d + "embed_seg_positions.weight"] = ((cfg.seg_bucket_size ** 2 + 1, C), "embed")
    ln(d + "pos_ln", C)
    ln(d + "image_pos_ln", C)
    ln(d + "seg_pos_ln", C)
    for n in ("self_pos_q_linear", "self_pos_k_linear", "cross_pos_q_linear", "cross_pos_k_linear"):
        lin(d + n, C, C)
    ln(d + "code_layernorm_embedding", C)
    for i in range(cfg.dec_layers):
        p = "%slayers.%d." % (d, i)
        mha(p + "self_attn")
        ln(p + "self_attn_ln", C)
        ln(p + "cross_attn_ln", C)
        ln(p + "self_attn_layer_norm", C)
        mha(p + "encoder_attn")
        ln(p + "encoder_attn_layer_norm", C)
        ln(p + "ffn_layernorm", Fd)
        lin(p + "fc1", Fd, C)
        lin(p + "fc2", C, Fd)
        ln(p + "final_layer_norm", C)
    ln(d + "layer_norm", C)
    n_seg = (2 * cfg.seg_bucket_size - 1) ** 2 + 3
    for i in range(cfg.dec_layers):
        s["%stoken_rel_pos_table_list.%d.weight" % (d, i)] = ((n_tok, H), "rel")
    for i in range(cfg.dec_layers):
        s["%simage_rel_pos_table_list.%d.weight" % (d, i)] = ((n_img, H), "rel")
    for i in range(cfg.dec_layers):
        s["%sseg_rel_pos_table_list.%d.weight" % (d, i)] = ((n_seg, H), "rel")
    return s


def _fill(name, shape, kind, seed):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    n = lambda std: torch.randn(shape, generator=g) * std  # noqa: E731
    if kind == "linear_w":
        return n(1.0 / math.sqrt(shape[1]) * 0.9)
    if kind == "bias":
        return n(0.02)
    if kind == "ln_w":
        return 1.0 + n(0.1)
    if kind == "ln_b":
        return n(0.05)
    if kind == "gain":
        return 1.0 + n(0.1)
    if kind == "embed":
        return n(0.5)
    if kind == "seg_embed":
        return n(0.06)
    if kind == "rel":
        return n(0.5)
    if kind == "conv":
        fan_in = shape[1] * shape[2] * shape[3]
        return n(math.sqrt(2.0 / fan_in))
    if kind == "bn_w":
        return 1.0 + n(0.05)
    if kind == "bn_w_last":
        return 0.3 + n(0.03)
    if kind == "bn_b":
        return n(0.05)
    if kind == "bn_mean":
        return n(0.05)
    if kind == "bn_var":
        return 0.8 + 0.4 * torch.rand(shape, generator=g)
    raise KeyError(kind)


def procedural_state_dict(cfg: SegOFAConfig, seed: int = 0, include_aliases=True) -> Dict[str, torch.Tensor]:
    """Deterministic weights keyed by parameter name -- no blobs are committed;
    both the reference (in gen_golden.py) and every consumer regenerate them."""
    sd = OrderedDict()
    spec = state_dict_spec(cfg)
    for name, (shape, kind) in spec.items():
        if kind.startswith("alias:"):
            continue
        sd[name] = _fill(name, shape, kind, seed)
    if include_aliases:
        for name, (shape, kind) in spec.items():
            if kind.startswith("alias:"):
                sd[name] = sd[kind[len("alias:"):]]
    return sd


# --------------------------------------------------------------------------- #
# building blocks                                                             #
# --------------------------------------------------------------------------- #
def _ln(sd, p, x, eps=1e-5):
    """fairseq/modules/layer_norm.py:30-35 -> torch LayerNorm, eps 1e-5."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gelu(x):
    """fairseq/modules/gelu.py:24-25: erf GELU evaluated in fp32."""
    return F.gelu(x.float()).type_as(x)


def _frozen_bn(sd, p, x, eps=1e-5):
    """frozen_bn.py:36-57 (no-grad branch: F.batch_norm(training=False))."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def resnet_trunk(sd, p, x, layers):
    """resnet.py:215-229 (_forward_impl) + Bottleneck.forward :117-137."""
    x = F.conv2d(x, sd[p + "conv1.weight"], None, 2, 3)
    x = F.relu(_frozen_bn(sd, p + "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, blocks in enumerate(layers, start=1):
        for b in range(blocks):
            q = "%slayer%d.%d." % (p, li, b)
            stride = 2 if (b == 0 and li > 1) else 1
            idt = x
            o = F.relu(_frozen_bn(sd, q + "bn1", F.conv2d(x, sd[q + "conv1.weight"])))
            o = F.relu(_frozen_bn(sd, q + "bn2", F.conv2d(o, sd[q + "conv2.weight"], None, stride, 1)))
            o = _frozen_bn(sd, q + "bn3", F.conv2d(o, sd[q + "conv3.weight"]))
            if b == 0:
                idt = _frozen_bn(sd, q + "downsample.1",
                                 F.conv2d(x, sd[q + "downsample.0.weight"], None, stride))
            x = F.relu(idt + o)
    return x


# Attention dropout (unify_multihead_attention.py:498: attn_probs = self.dropout_module(attn_weights), 0.0 in every shipped
# script): a test that wants it installs ATTN_PROB_HOOK(prefix, probs) -> probs, e.g. probs * keep / (1 - p) with the keep mask
# of the implementation under test (F.dropout with a known mask); None = the recipe's identity.
ATTN_PROB_HOOK = None
# Activation dropout (unify_transformer_layer.py:280,556: between the activation and ffn_layernorm; 0 in every shipped script):
# ACT_HOOK(prefix, a) -> a, as above.
ACT_HOOK = None


def mha(sd, p, cfg, xq, xkv, bias, causal=False, key_padding_mask=None):
    """unify_multihead_attention.py:327-513, batch-first.

    q is scaled by (head_dim*scale_factor)^-0.5 *before* QK^T (:58,:346); the
    bias is added unscaled (:464-465); causal mask = -inf strictly above the
    diagonal (decoder_module.py:878-890); softmax in fp32 (fairseq/utils.py:510-514);
    per-head gain c_attn (:509-512); out_proj (:513).
    ``bias``: [H, T, S] (batch-invariant) or [B, H, T, S]."""
    B, T, C = xq.shape
    S = xkv.shape[1]
    H, d = cfg.heads, cfg.head_dim
    scaling = float(d * cfg.attn_scale_factor) ** -0.5
    q = (_lin(sd, p + ".q_proj", xq) * scaling).view(B, T, H, d).transpose(1, 2)
    k = _lin(sd, p + ".k_proj", xkv).view(B, S, H, d).transpose(1, 2)
    v = _lin(sd, p + ".v_proj", xkv).view(B, S, H, d).transpose(1, 2)
    s = torch.matmul(q, k.transpose(2, 3))
    if bias is not None:
        s = s + (bias if bias.dim() == 4 else bias.unsqueeze(0))
    if causal:
        s = s + torch.triu(torch.full((T, S), float("-inf"), dtype=s.dtype, device=s.device), 1)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    pr = F.softmax(s.float(), dim=-1).type_as(s)
    if ATTN_PROB_HOOK is not None:
        pr = ATTN_PROB_HOOK(p, pr)
    o = torch.matmul(pr, v)                       # [B,H,T,d]
    o = o * sd[p + ".c_attn"].view(1, H, 1, 1)
    o = o.transpose(1, 2).reshape(B, T, C)
    return _lin(sd, p + ".out_proj", o)


def _ffn(sd, p, cfg, x):
    """unify_transformer_layer.py:276-289 / :552-566 (pre-LN, scale_fc)."""
    y = _ln(sd, p + "final_layer_norm", x)
    y = _gelu(_lin(sd, p + "fc1", y))
    if ACT_HOOK is not None:
        y = ACT_HOOK(p, y)
    y = _ln(sd, p + "ffn_layernorm", y)
    y = _lin(sd, p + "fc2", y)
    return x + y


def encoder_layer(sd, p, cfg, x, bias, key_padding_mask=None):
    """unify_transformer_layer.py:222-292 (normalize_before, scale_attn)."""
    y = _ln(sd, p + "self_attn_layer_norm", x)
    y = mha(sd, p + "self_attn", cfg, y, y, bias, False, key_padding_mask)
    y = _ln(sd, p + "attn_ln", y)
    x = x + y
    return _ffn(sd, p, cfg, x)


def decoder_layer(sd, p, cfg, x, enc, self_bias, cross_bias, causal, enc_padding_mask=None):
    """unify_transformer_layer.py:431-581."""
    y = _ln(sd, p + "self_attn_layer_norm", x)
    y = mha(sd, p + "self_attn", cfg, y, y, self_bias, causal)
    y = _ln(sd, p + "self_attn_ln", y)
    x = x + y
    y = _ln(sd, p + "encoder_attn_layer_norm", x)
    y = mha(sd, p + "encoder_attn", cfg, y, enc, cross_bias, False, enc_padding_mask)
    y = _ln(sd, p + "cross_attn_ln", y)
    x = x + y
    return _ffn(sd, p, cfg, x)


def _heads(x, H):
    """[T, C] -> [H, T, d]"""
    T, C = x.shape
    return x.view(T, H, C // H).transpose(0, 1)


def _resize_hw(t, src_hw, dst_hw):
    """bilinear resize of the trailing (flattened) grid dim: [..., h*w] -> [..., h'*w']
    (F.interpolate(mode='bilinear'), align_corners=False)."""
    lead = t.shape[:-1]
    t4 = t.reshape(1, -1, src_hw[0], src_hw[1])
    t4 = F.interpolate(t4, size=tuple(dst_hw), mode="bilinear")
    return t4.reshape(*lead, dst_hw[0] * dst_hw[1])


def image_grid_ids(h, w, bucket):
    """encoder_module.py:339-341 / decoder_module.py:541-542: x + y*bucket + 1."""
    return (torch.arange(w)[None, :] + torch.arange(h)[:, None] * bucket + 1).reshape(-1)


def encoder_rel_bias(sd, cfg, idx, L, hw, image_rp_bucket, token_rp_bucket):
    """Per-layer rel-pos bias pieces of encode() (encoder_module.py:313-331,790-809):
    token block [H, L, L] and image block [H, P, P] (double-bilinear-resized from
    the orig grid when the feature grid differs; identity otherwise)."""
    H = cfg.heads
    tok = sd["encoder.token_rel_pos_table_list.%d.weight" % idx][token_rp_bucket[:L, :L]]
    tok = tok.permute(2, 0, 1)
    oh = cfg.orig_patch_image_size // 16
    ids0 = image_grid_ids(oh, oh, cfg.image_bucket_size)
    rp = image_rp_bucket[ids0][:, ids0]                                  # [P0, P0]
    img = sd["encoder.image_rel_pos_table_list.%d.weight" % idx][rp].permute(2, 0, 1)  # [H,P0,P0]
    if tuple(hw) != (oh, oh):
        # :803-807: resize over the key grid, then over the query grid
        img = _resize_hw(img, (oh, oh), hw)                              # [H, P0, P]
        img = _resize_hw(img.transpose(1, 2), (oh, oh), hw).transpose(1, 2)  # [H, P, P]
    return tok, img


def embed_bag_offsets(offsets_cat, B):
    """encode_with_artificial_image (encoder_module.py:531-536): the collated per-sample cumulative bag
    ends ``offsets_cat`` [B*P] (segmentation_dataset.py:327-328,100) -> EmbeddingBag start offsets [B*P]
    into the pad-stripped, concatenated id stream."""
    o = offsets_cat.view(B, -1)
    o = torch.cat([o.new_zeros(B, 1), o], dim=1)
    base = torch.cat([o.new_zeros(1), o[:-1, -1]]).cumsum(0)
    return (o + base[:, None])[:, :-1].reshape(-1)


def embed_bag_mean(weight, ids, starts):
    """nn.EmbeddingBag(mode='mean') over ``weight`` (encoder_module.py:147-148,538): bag i averages rows
    ids[starts[i]:starts[i+1]] (last bag runs to the end); an empty bag is zero."""
    n = starts.numel()
    ends = torch.cat([starts[1:], torch.tensor([ids.numel()])])
    seg = torch.repeat_interleave(torch.arange(n), ends - starts)
    out = torch.zeros(n, weight.shape[1], dtype=weight.dtype)
    out = out.index_add(0, seg, weight[ids])
    cnt = (ends - starts).clamp(min=1).to(weight.dtype)
    return out / cnt[:, None]


def encode(sd, cfg, src_tokens, patch_images, patch_masks=None, image_feat=None, bag=None):
    """TransformerEncoder.encode (encoder_module.py:677-851), real-image path; with ``bag`` =
    (padded ids [B, maxlen], collated bag ends [B*P]) the image-free entry
    encode_with_artificial_image (encoder_module.py:499-675): the patch embeddings are the mean
    token embedding of a class name per patch -- no ResNet, no image_proj.

    Returns dict with batch-first tensors: encoder_out [B,T,C],
    position_embeddings [T,C] (LN'd, batch-invariant), image_embed_shape,
    image_embed_before_proj [B,P,1024], encoder_padding_mask [B,T] | None."""
    B, L = src_tokens.shape
    H = cfg.heads
    if bag is not None:
        ids, ends = bag
        h = w = cfg.patch_image_size // 16                                # :540
        P = h * w
        flat = ids[ids != PAD]                                            # :529
        image_embed = embed_bag_mean(sd["encoder.embed_tokens.weight"], flat, embed_bag_offsets(ends, B)).view(B, P, -1)
    else:
        if image_feat is None:
            image_feat = resnet_trunk(sd, "encoder.embed_images.", patch_images, cfg.resnet_layers)
        h, w = image_feat.shape[-2:]
        P = h * w
        image_embed = image_feat.flatten(2).transpose(1, 2)               # [B,P,1024]
    # ---- get_patch_images_info :333-372 ----
    ids = image_grid_ids(h, w, cfg.image_bucket_size)
    oh = cfg.orig_patch_image_size // 16
    if P > oh * oh:
        old = sd["encoder.embed_image_positions.weight"][image_grid_ids(oh, oh, cfg.image_bucket_size)]
        image_pos = _resize_hw(old.t(), (oh, oh), (h, w)).t()             # [P,C]
    else:
        image_pos = sd["encoder.embed_image_positions.weight"][ids]
    pos = sd["encoder.embed_positions.weight"][torch.arange(L)]           # :744
    # ---- forward_embedding :388-446 (embed_scale 1.0, type embedding, LN) ----
    tok = sd["encoder.embed_tokens.weight"][src_tokens] + sd["encoder.type_embedding.weight"][0]
    tok = _ln(sd, "encoder.layernorm_embedding", tok)
    if bag is not None:
        img = image_embed + sd["encoder.type_embedding.weight"][1]        # :589-593 (embed_scale 1.0)
    else:
        img = _lin(sd, "encoder.image_proj", image_embed) + sd["encoder.type_embedding.weight"][1]
    img = _ln(sd, "encoder.patch_layernorm_embedding", img)
    x = torch.cat([img, tok], dim=1)                                      # [B,T,C]
    # ---- padding :730-752 ----
    pad_mask = src_tokens.eq(PAD)
    img_pad = torch.zeros(B, P, dtype=torch.bool)
    if patch_masks is not None and bag is None:
        img_pad[~patch_masks] = True
    pad_mask = torch.cat([img_pad, pad_mask], dim=1)
    has_pads = bool(pad_mask.any())
    if has_pads:
        x = x * (1 - pad_mask.unsqueeze(-1).type_as(x))
    # ---- abs-pos bias :757-771 ----
    pos_all = torch.cat([_ln(sd, "encoder.image_pos_ln", image_pos), _ln(sd, "encoder.pos_ln", pos)], dim=0)
    pos_scaling = float(cfg.head_dim * cfg.attn_scale_factor) ** -0.5
    pq = _heads(_lin(sd, "encoder.pos_q_linear", pos_all), H) * pos_scaling
    pk = _heads(_lin(sd, "encoder.pos_k_linear", pos_all), H)
    abs_bias = torch.matmul(pq, pk.transpose(1, 2))                       # [H,T,T]
    n_img = (2 * cfg.image_bucket_size - 1) ** 2 + 3
    image_rp_bucket = make_image_bucket_position(cfg.image_bucket_size, n_img)
    token_rp_bucket = make_token_bucket_position(cfg.token_bucket_size)
    for idx in range(cfg.enc_layers):
        tokb, imgb = encoder_rel_bias(sd, cfg, idx, L, (h, w), image_rp_bucket, token_rp_bucket)
        bias = abs_bias.clone()
        bias[:, P:, P:] += tokb
        bias[:, :P, :P] += imgb
        x = encoder_layer(sd, "encoder.layers.%d." % idx, cfg, x, bias, pad_mask if has_pads else None)
    x = _ln(sd, "encoder.layer_norm", x)
    return {
        "encoder_out": x,
        "position_embeddings": pos_all,
        "image_embed_shape": (h, w),
        "image_embed_before_proj": image_embed if bag is None else None,
        "encoder_padding_mask": pad_mask if has_pads else None,
    }


def decoder_seg_rel_bias(sd, cfg, idx, hw, seg_rp_bucket):
    """decoder_module.py:327-333 + :603-627: gather [H, N0+1, N0+1]; the [1:,1:]
    block is bilinear-resized over rows then columns with the bos row/col passed
    through; identity when the grid equals seg_bucket_size^2."""
    sb = cfg.seg_bucket_size
    rel = sd["decoder.seg_rel_pos_table_list.%d.weight" % idx][seg_rp_bucket].permute(2, 0, 1)  # [H,hw1,hw2]
    if tuple(hw) == (sb, sb):
        return rel
    # first pass: resize along hw1 (rows) for every column hw2
    t = rel.transpose(1, 2)                                   # [H, hw2, hw1]
    t = torch.cat([t[..., :1], _resize_hw(t[..., 1:], (sb, sb), hw)], dim=-1)   # [H, hw2, P+1]
    # second pass: resize along hw2 (columns) for every (new) row
    t = t.transpose(1, 2)                                     # [H, P+1, hw2]
    t = torch.cat([t[..., :1], _resize_hw(t[..., 1:], (sb, sb), hw)], dim=-1)   # [H, P+1, P+1]
    return t


def decode(sd, cfg, enc, bos_tokens, full_context_alignment=False):
    """TransformerDecoder.extract_features_scriptable_surrogate + output_layer
    (decoder_module.py:486-677, 290-294)."""
    enc_out = enc["encoder_out"]
    B = enc_out.shape[0]
    H = cfg.heads
    h, w = enc["image_embed_shape"]
    P = h * w
    sb = cfg.seg_bucket_size
    x = torch.cat([sd["decoder.embed_tokens.weight"][bos_tokens[:, :1]], enc_out[:, :P]], dim=1)  # :530-538
    old = sd["decoder.embed_seg_positions.weight"][image_grid_ids(sb, sb, sb)]                  # :541-544
    img_pos = old if (h, w) == (sb, sb) else _resize_hw(old.t(), (sb, sb), (h, w)).t()
    tgt_pos = torch.cat([sd["decoder.embed_seg_positions.weight"][:1], img_pos], dim=0)          # :550
    tp = _ln(sd, "decoder.seg_pos_ln", tgt_pos)
    pos_scaling = float(cfg.head_dim * cfg.attn_scale_factor) ** -0.5
    self_abs = torch.matmul(_heads(_lin(sd, "decoder.self_pos_q_linear", tp), H) * pos_scaling,
                            _heads(_lin(sd, "decoder.self_pos_k_linear", tp), H).transpose(1, 2))
    src_pos = enc["position_embeddings"]
    cross_abs = torch.matmul(_heads(_lin(sd, "decoder.cross_pos_q_linear", tp), H) * pos_scaling,
                             _heads(_lin(sd, "decoder.cross_pos_k_linear", src_pos), H).transpose(1, 2))
    x = _ln(sd, "decoder.layernorm_embedding", x)            # disable_entangle: no pos add (:572)
    n_seg = (2 * sb - 1) ** 2 + 3
    seg_rp_bucket = make_image_bucket_position(sb, n_seg)
    for idx in range(cfg.dec_layers):
        rel = decoder_seg_rel_bias(sd, cfg, idx, (h, w), seg_rp_bucket)
        x = decoder_layer(sd, "decoder.layers.%d." % idx, cfg, x, enc_out, self_abs + rel, cross_abs,
                          causal=not full_context_alignment,
                          enc_padding_mask=enc["encoder_padding_mask"])
    x = _ln(sd, "decoder.layer_norm", x)
    logits = F.linear(x, sd["decoder.seg_projection.weight"])
    return logits, x


def segofa_forward(sd, cfg, src_tokens, patch_images, prev_output_tokens=None, patch_masks=None,
                   full_context_alignment=False, image_feat=None):
    """SegOFAModel.forward (segofa.py:69-153), real-image branch.
    -> (logits [B, P+1, nseg], extra)."""
    B = src_tokens.shape[0]
    if prev_output_tokens is None:
        prev_output_tokens = torch.zeros(B, 1, dtype=torch.long)
    enc = encode(sd, cfg, src_tokens, patch_images, patch_masks, image_feat=image_feat)
    logits, feat = decode(sd, cfg, enc, prev_output_tokens, full_context_alignment)
    return logits, {"encoder_returns": enc, "penultimate": feat}


def segofa_forward_imfree(sd, cfg, aux_input):
    """SegOFAModel.forward, ``aux_input`` branch (segofa.py:136-151): encoder on the artificial image,
    decoder with its defaults (causal).  aux_input: src_tokens [B,L], patch_images = padded bag ids
    [B,maxlen], patch_masks = collated bag ends [B*P], prev_output_tokens [B,>=1] (only bos is read)."""
    enc = encode(sd, cfg, aux_input["src_tokens"], None, None,
                 bag=(aux_input["patch_images"], aux_input["patch_masks"]))
    logits, feat = decode(sd, cfg, enc, aux_input["prev_output_tokens"], False)
    return logits, {"encoder_returns": enc, "penultimate": feat}


def imfree_loss(cfg, logits, text2seg_target, label_smoothing=0.0):
    """compute_imfree_loss (seg_criterion.py:246-267): upsample_logits with its defaults (32x32 -> 512x512),
    drop eos / pad / ignore, mean CE."""
    P = logits.shape[1] - 1
    hp = wp = int(round(P ** 0.5))
    h = w = int(round((text2seg_target.shape[1] - 1) ** 0.5))
    scores = upsample_logits(logits.float(), hp, wp, h, w)[:, :-1]
    tgt = text2seg_target[:, :-1]
    scores = scores.reshape(-1, scores.shape[-1])
    tgt = tgt.reshape(-1)
    mask = (tgt != PAD) & (tgt != cfg.seg_id_offset + cfg.num_seg_tokens)
    return F.cross_entropy(scores[mask], tgt[mask] - cfg.seg_id_offset, label_smoothing=label_smoothing)


# --------------------------------------------------------------------------- #
# criterion math (criterions/seg_criterion.py)                                #
# --------------------------------------------------------------------------- #
def upsample_logits(logits, hp, wp, h, w):
    """seg_criterion.py:237-244."""
    lo = logits[:, :-1]
    B, _, n = lo.shape
    lo = lo.transpose(1, 2).reshape(B, n, hp, wp)
    lo = F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
    lo = lo.reshape(B, n, h * w).transpose(1, 2)
    return torch.cat([lo, logits[:, -1:]], dim=1)


def seg_loss(cfg, logits, target, hp, wp, h, w, label_smoothing=0.0):
    """compute_loss, training branch with upscale_lprobs (seg_criterion.py:269-347):
    bilinear upsample -> drop pad/eos/ignore -> mean CE.  ``target`` is
    [B, h*w+1] of dictionary ids (<seg_i> = seg_id_offset + i, last = eos)."""
    scores = upsample_logits(logits.float(), hp, wp, h, w)
    mask = (target == PAD) | (target == cfg.seg_id_offset + cfg.num_seg_tokens) | (target == EOS)
    t = target[~mask] - cfg.seg_id_offset
    s = scores[~mask]
    loss = F.cross_entropy(s, t, label_smoothing=label_smoothing)
    return loss, s, t


def neighbour_smoothing(logits, resnet_feature, iters=25, topk=3, temperature=1.0):
    """Eval-time post-processing of seg_criterion.py:197-213: every patch's class probabilities are replaced,
    ``iters`` times, by the mean over its ``topk`` nearest patches in cosine similarity of the trunk features
    (the patch itself is its own nearest neighbour).  logits [B,P+1,n], resnet_feature [B,P,D]
    -> probabilities [B,P+1,n] with a zero row in the eos slot."""
    f = F.normalize(resnet_feature, dim=-1)
    sim = f @ f.transpose(-1, -2)
    _, ind = torch.topk(sim, k=topk, dim=-1)                       # [B,P,k]
    B, P = ind.shape[:2]
    bi = torch.arange(B).view(B, 1, 1).expand(B, P, topk)
    prob = (logits / temperature).softmax(-1)
    for _ in range(iters):
        prob = prob[bi, ind].mean(dim=-2)
    return torch.cat([prob, prob.new_zeros(B, 1, prob.size(-1))], dim=1)


def round_weights_bf16(sd):
    """Test helper: every floating-point tensor outside the (BN-folded) trunk is replaced by its bf16 rounding, held in
    fp32 -- the fp32 reference / oracle and the bf16 HIP path then run on IDENTICAL weight values (BASELINE.md section 5:
    "identical inputs/weights"), so what a parity test measures is the arithmetic, not the one-off rounding of procedural
    fp32 weights.  Aliased tensors stay aliased.  Returns a new state dict."""
    out, done = {}, {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point or "embed_images" in k:
            out[k] = v
            continue
        key = v.data_ptr()
        if key not in done:
            done[key] = v.detach().to(torch.bfloat16).to(torch.float32)
        out[k] = done[key]
    return out


def diversify_seg_projection(sd, cfg, batch, gain=6.0):
    """Test helper for the eval fixtures: with the procedural weights every patch predicts the same class (the
    class offsets W x_mean dominate the logits), which would make argmax / histogram checks vacuous.  Project the
    mean penultimate feature out of the (tied) seg projection rows and scale them, so predictions vary across
    patches.  Deterministic given (sd, batch); returns a new state dict."""
    with torch.no_grad():
        _, extra = segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        xm = extra["penultimate"].reshape(-1, cfg.embed_dim).mean(0)
        w = sd["encoder.seg_embed_tokens.weight"]
        w = (w - torch.outer(w @ xm / xm.dot(xm), xm)) * gain
    out = dict(sd)
    for k in ("encoder.seg_embed_tokens.weight", "decoder.seg_embed_tokens.weight", "decoder.seg_projection.weight"):
        out[k] = w
    return out


def seg_eval(cfg, scores_low, ori_seg, hp, wp):
    """compute_loss, eval branch (seg_criterion.py:289-347): per-patch scores (logits, or the smoothed
    probabilities) are bilinearly resized to the ORIGINAL image shape, the eos slot dropped, and compared with
    the original label map.  scores_low [1,P+1,n]; ori_seg int [h,w] class ids (num_seg = ignore).
    -> (display CE loss, (intersect, pred, label, union) histograms)."""
    h, w = ori_seg.shape
    target = ori_seg.reshape(1, -1).long() + cfg.seg_id_offset
    scores = upsample_logits(scores_low.float(), hp, wp, h, w)[:, :-1]
    mask = (target == PAD) | (target == cfg.seg_id_offset + cfg.num_seg_tokens) | (target == EOS)
    t = target[~mask] - cfg.seg_id_offset
    s = scores[~mask]
    return F.cross_entropy(s, t), seg_metric(s, t, cfg.num_seg_tokens)


def seg_metric(scores, target, num_classes):
    """compute_metric (seg_criterion.py:349-362): per-class area histograms."""
    pred = scores.argmax(-1)
    inter = pred[pred == target]
    a_i = torch.histc(inter.float(), bins=num_classes, min=0, max=num_classes - 1)
    a_p = torch.histc(pred.float(), bins=num_classes, min=0, max=num_classes - 1)
    a_l = torch.histc(target.float(), bins=num_classes, min=0, max=num_classes - 1)
    return a_i, a_p, a_l, a_p + a_l - a_i


def miou(area_intersect, area_union):
    """reduce_metrics (seg_criterion.py:558-562): nanmean(intersect / union)."""
    iou = area_intersect / area_union
    return torch.nanmean(iou)


# --------------------------------------------------------------------------- #
# synthetic inputs                                      #
# --------------------------------------------------------------------------- #
def synthetic_batch(cfg, batch, src_len, image_size=None, seed=1234, image_hw=None):
    g = torch.Generator().manual_seed(seed)
    S = image_size or cfg.patch_image_size
    hw = image_hw or (S, S)
    src = torch.randint(4, min(50000, cfg.vocab_size - 1), (1, src_len), generator=g).repeat(batch, 1)
    src[:, 0] = BOS
    src[:, -1] = EOS
    img = torch.randn(batch, 3, hw[0], hw[1], generator=g)
    tgt = torch.randint(0, cfg.num_seg_tokens, (batch, hw[0] * hw[1]), generator=g) + cfg.seg_id_offset
    tgt = torch.cat([tgt, torch.full((batch, 1), EOS, dtype=torch.long)], dim=1)
    return {"src_tokens": src, "patch_images": img, "target": tgt,
            "prev_output_tokens": torch.zeros(batch, 1, dtype=torch.long),
            "patch_masks": torch.ones(batch, dtype=torch.bool)}


def synthetic_aux_batch(cfg, batch, src_len, seed=4321, image_size=None):
    """Image-free sample in the layout of segmentation_dataset.py:303-345 + collate :85-107:
    every class "name" is 1-3 random BPE ids; a random low-res class map (``rand_k``) is nearest-resized
    to the patch grid (bag ids) and to the image (targets)."""
    g = torch.Generator().manual_seed(seed)
    S = image_size or cfg.patch_image_size
    hp = S // 16
    n = cfg.num_seg_tokens
    name_len = torch.randint(1, 4, (n,), generator=g)
    names = [torch.randint(4, min(50000, cfg.vocab_size - 1), (int(k),), generator=g) for k in name_len]
    src = torch.randint(4, min(50000, cfg.vocab_size - 1), (1, src_len), generator=g).repeat(batch, 1)
    src[:, 0] = BOS
    src[:, -1] = EOS
    ids, ends, tgts = [], [], []
    for _ in range(batch):
        sh, sw = (int(v) for v in torch.randint(1, hp + 1, (2,), generator=g))
        coarse = torch.randint(0, n, (1, 1, sh, sw), generator=g).float()
        low = F.interpolate(coarse, size=(hp, hp), mode="nearest").long().reshape(-1)
        high = F.interpolate(coarse, size=(S, S), mode="nearest").long().reshape(-1)
        ids.append(torch.cat([names[int(c)] for c in low]))
        ends.append(torch.tensor([int(name_len[int(c)]) for c in low]).cumsum(0))
        tgts.append(torch.cat([high + cfg.seg_id_offset, torch.tensor([EOS])]))
    maxlen = max(t.numel() for t in ids)
    padded = torch.full((batch, maxlen), PAD, dtype=torch.long)
    for b, t in enumerate(ids):
        padded[b, : t.numel()] = t
    return {"aux_input": {"src_tokens": src, "src_lengths": torch.full((batch,), src_len),
                          "patch_images": padded, "patch_masks": torch.cat(ends),
                          "prev_output_tokens": torch.zeros(batch, 1, dtype=torch.long)},
            "text2seg_target": torch.stack(tgts)}
