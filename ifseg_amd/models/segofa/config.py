"""Architecture presets and the parameter layout of SegOFA.

Mirrors the reference's ``register_model_architecture("segofa", ...)`` presets
(models/segofa/segofa.py:351-467) with the flags the shipped run scripts set
(run_scripts/IFSeg/coco_unseen.sh:76,89-96,114-121,128-134): pre-LN, scale_attn /
scale_fc / scale_heads, add_type_embedding, disable_entangle, tie_seg_projection,
decoder_type=surrogate, decoder_input_type=encoder_output, frozen ResNet /
embeddings.  ``param_spec`` lists every floating state_dict entry with the
reference's key names (SURVEY.md section 8a "state-dict contract").
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class SegOFAConfig:
    embed_dim: int = 768
    ffn_dim: int = 3072
    heads: int = 12
    enc_layers: int = 6
    dec_layers: int = 6
    resnet_layers: Tuple[int, ...] = (3, 4, 23)
    num_seg_tokens: int = 15
    vocab_size: int = 59458           # len(dictionary) - num_seg_tokens (unify_transformer.py:402)
    patch_image_size: int = 512
    orig_patch_image_size: int = 512
    image_bucket_size: int = 42
    token_bucket_size: int = 256
    attn_scale_factor: float = 2.0
    max_source_positions: int = 1024
    max_target_positions: int = 1024
    code_image_size: int = 128
    # stochastic regularisers (coco_unseen.sh:19-22: dropout 0.1, encoder/decoder drop-path 0.1,
    # attention_dropout 0.0).  Parity is defined at 0 (the RNG stream cannot match torch's).
    dropout: float = 0.0
    activation_dropout: float = 0.0     # between GELU and ffn_layernorm (unify_transformer_layer.py:142-147,280; alias --relu-dropout)
    attention_dropout: float = 0.0      # on the softmax probabilities (unify_multihead_attention.py:498); inside the attention kernels
    encoder_drop_path_rate: float = 0.0
    decoder_drop_path_rate: float = 0.0
    # freezes of the shipped recipe (coco_unseen.sh:31-33,76)
    freeze_resnet: bool = True
    freeze_embeddings: bool = True                 # the shared token table (encoder / decoder embed_tokens)
    freeze_seg_embedding: Optional[bool] = None    # the seg embeddings = tied seg projection; None: as freeze_embeddings
    # prompts of different lengths in one batch (right-padded with <pad>: encoder_padding_mask, encoder_module.py:730-752, masks
    # those keys in the encoder self- and the decoder cross-attention, unify_multihead_attention.py:477-489).  Off: a padded
    # batch is refused (the IFSeg recipe gives every sample the same prompt) and a step carries no per-batch length tensor;
    # on (or IFSEG_PADDED_PROMPTS=1): every batch carries its valid key counts to the attention kernels.
    padded_prompts: bool = False

    @property
    def head_dim(self):
        return self.embed_dim // self.heads

    @property
    def seg_bucket_size(self):        # decoder_module.py:191
        return self.patch_image_size // 16

    @property
    def seg_id_offset(self):          # index of <seg_0> in the task dictionary
        return self.vocab_size - 1


ARCHS = {
    # models/segofa/segofa.py:422-431
    "segofa_base": dict(embed_dim=768, ffn_dim=3072, heads=12, enc_layers=6, dec_layers=6, resnet_layers=(3, 4, 23)),
    # :351-419
    "segofa_large": dict(embed_dim=1024, ffn_dim=4096, heads=16, enc_layers=12, dec_layers=12, resnet_layers=(3, 8, 36)),
    # :434-443
    "segofa_huge": dict(embed_dim=1280, ffn_dim=5120, heads=16, enc_layers=24, dec_layers=12, resnet_layers=(3, 8, 36)),
    # :446-455
    "segofa_medium": dict(embed_dim=512, ffn_dim=2048, heads=8, enc_layers=4, dec_layers=4, resnet_layers=(3, 4, 23)),
    # :458-467
    "segofa_tiny": dict(embed_dim=256, ffn_dim=1024, heads=4, enc_layers=4, dec_layers=4, resnet_layers=(3, 4, 6)),
}


def make_config(arch="segofa_base", **overrides):
    d = dict(ARCHS[arch])
    d.update(overrides)
    return SegOFAConfig(**d)


# ---- bucket index generators (closed forms of unify_transformer.py:55-88) ---------------
def token_bucket_of_delta(bucket_size=256, max_position=1024):
    """bucket id for every relative offset d = i - j in [-(max-1), max-1]
    (make_token_bucket_position, unify_transformer.py:55-68): exact for |d| <= mid,
    log-spaced beyond.  Returned tensor is indexed by d + max_position - 1."""
    d = torch.arange(-(max_position - 1), max_position, dtype=torch.long)
    sign = torch.sign(d)
    mid = bucket_size // 2
    a = torch.where((d < mid) & (d > -mid), torch.full_like(d, mid - 1), d.abs())
    lp = torch.ceil(torch.log(a.float() / mid) / math.log((max_position - 1) / mid) * (mid - 1)) + mid
    lp = lp.int().long()
    b = torch.where(a.le(mid), d, lp * sign)
    return b + bucket_size - 1


def token_rp_bucket(bucket_size=256, max_position=1024):
    t = token_bucket_of_delta(bucket_size, max_position)
    i = torch.arange(max_position)
    return t[(i[:, None] - i[None, :]) + max_position - 1]


def image_rp_bucket(bucket_size, num_rel):
    """make_image_bucket_position (unify_transformer.py:71-88): index 0 is the
    special (bos / cls) slot, grid cell (y, x) has position id x + y*bucket + 1."""
    n = bucket_size * bucket_size
    ys = torch.arange(bucket_size).repeat_interleave(bucket_size)
    xs = torch.arange(bucket_size).repeat(bucket_size)
    rel = (ys[:, None] - ys[None, :] + bucket_size - 1) * (2 * bucket_size - 1) + (xs[:, None] - xs[None, :] + bucket_size - 1)
    idx = torch.zeros(n + 1, n + 1, dtype=torch.long)
    idx[1:, 1:] = rel
    idx[0, :] = num_rel - 3
    idx[:, 0] = num_rel - 2
    idx[0, 0] = num_rel - 1
    return idx


def param_spec(cfg: SegOFAConfig):
    """OrderedDict name -> (shape, kind, trainable).  kind in
    {linear_w, bias, ln_w, ln_b, gain, embed, rel, conv, bn_*} or 'alias:<name>'."""
    C, Fd, H = cfg.embed_dim, cfg.ffn_dim, cfg.heads
    s = OrderedDict()
    fe = not cfg.freeze_embeddings
    fs = not (cfg.freeze_embeddings if cfg.freeze_seg_embedding is None else cfg.freeze_seg_embedding)
    fr = not cfg.freeze_resnet

    def lin(p, o, i, tr=True):
        s[p + ".weight"] = ((o, i), "linear_w", tr)
        s[p + ".bias"] = ((o,), "bias", tr)

    def ln(p, n):
        s[p + ".weight"] = ((n,), "ln_w", True)
        s[p + ".bias"] = ((n,), "ln_b", True)

    def bn(p, n, last=False):
        s[p + ".weight"] = ((n,), "bn_w_last" if last else "bn_w", False)
        s[p + ".bias"] = ((n,), "bn_b", False)
        s[p + ".running_mean"] = ((n,), "bn_mean", False)
        s[p + ".running_var"] = ((n,), "bn_var", False)

    def mha(p):
        s[p + ".c_attn"] = ((H,), "gain", True)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            lin(p + "." + n, C, C)

    e = "encoder."
    s[e + "embed_tokens.weight"] = ((cfg.vocab_size, C), "embed", fe)
    s[e + "seg_embed_tokens.weight"] = ((cfg.num_seg_tokens, C), "seg_embed", fs)
    s[e + "embed_tokens_bag.weight"] = ((cfg.vocab_size, C), "alias:encoder.embed_tokens.weight", fe)
    ln(e + "layernorm_embedding", C)
    s[e + "type_embedding.weight"] = ((2, C), "embed", True)
    r = e + "embed_images."
    s[r + "conv1.weight"] = ((64, 3, 7, 7), "conv", fr)
    bn(r + "bn1", 64)
    inpl = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256), cfg.resnet_layers), start=1):
        for b in range(blocks):
            p = "%slayer%d.%d." % (r, li, b)
            s[p + "conv1.weight"] = ((planes, inpl, 1, 1), "conv", fr)
            bn(p + "bn1", planes)
            s[p + "conv2.weight"] = ((planes, planes, 3, 3), "conv", fr)
            bn(p + "bn2", planes)
            s[p + "conv3.weight"] = ((planes * 4, planes, 1, 1), "conv", fr)
            bn(p + "bn3", planes * 4, last=True)
            if b == 0:
                s[p + "downsample.0.weight"] = ((planes * 4, inpl, 1, 1), "conv", fr)
                bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    lin(e + "image_proj", C, 1024, tr=fr)
    ln(e + "patch_layernorm_embedding", C)
    s[e + "embed_positions.weight"] = ((cfg.max_source_positions + 2, C), "embed", True)
    s[e + "embed_image_positions.weight"] = ((cfg.image_bucket_size ** 2 + 1, C), "embed", True)
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
        s["%stoken_rel_pos_table_list.%d.weight" % (e, i)] = ((n_tok, H), "rel", True)
    for i in range(cfg.enc_layers):
        s["%simage_rel_pos_table_list.%d.weight" % (e, i)] = ((n_img, H), "rel", True)

    d = "decoder."
    s[d + "seg_embed_tokens.weight"] = ((cfg.num_seg_tokens, C), "alias:encoder.seg_embed_tokens.weight", fs)
    s[d + "seg_projection.weight"] = ((cfg.num_seg_tokens, C), "alias:encoder.seg_embed_tokens.weight", fs)
    s[d + "embed_tokens.weight"] = ((cfg.vocab_size, C), "alias:encoder.embed_tokens.weight", fe)
    ln(d + "layernorm_embedding", C)
    s[d + "embed_positions.weight"] = ((cfg.max_target_positions + 2, C), "embed", True)
    s[d + "embed_image_positions.weight"] = ((cfg.image_bucket_size ** 2 + 1, C), "embed", True)
    s[d + "embed_seg_positions.weight"] = ((cfg.seg_bucket_size ** 2 + 1, C), "embed", True)
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
        s["%stoken_rel_pos_table_list.%d.weight" % (d, i)] = ((n_tok, H), "rel", True)
    for i in range(cfg.dec_layers):
        s["%simage_rel_pos_table_list.%d.weight" % (d, i)] = ((n_img, H), "rel", True)
    for i in range(cfg.dec_layers):
        s["%sseg_rel_pos_table_list.%d.weight" % (d, i)] = ((n_seg, H), "rel", True)
    return s


def init_tensor(name, shape, kind, gen):
    """Random init following the reference: init_bert_params N(0, 0.02) for every
    Linear / Embedding incl. the rel-pos tables (segofa.py:33,
    transformer_sentence_encoder.py:42-49), LayerNorm (1, 0), c_attn ones, ResNet
    kaiming fan_out (resnet.py:201-206), FrozenBN identity (frozen_bn.py:30-34)."""
    if kind in ("linear_w", "embed", "seg_embed", "rel"):
        return torch.randn(shape, generator=gen) * 0.02
    if kind in ("bias", "ln_b", "bn_b", "bn_mean"):
        return torch.zeros(shape)
    if kind in ("ln_w", "gain", "bn_w", "bn_w_last"):
        return torch.ones(shape)
    if kind == "bn_var":
        return torch.ones(shape) - 1e-5
    if kind == "conv":
        fan_out = shape[0] * shape[2] * shape[3]
        return torch.randn(shape, generator=gen) * math.sqrt(2.0 / fan_out)
    raise KeyError(kind)
