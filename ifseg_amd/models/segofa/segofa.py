"""SegOFAModel -- the reference's plugin surface over the MI355X HIP engine.

Drop-in boundary: same registration names
(``@register_model("segofa")``, architectures ``segofa_{tiny,medium,base,large,huge}``
-- models/segofa/segofa.py:25,351-467), same ``state_dict`` keys / shapes (889
entries for Base), same ``forward(**net_input, full_context_alignment=...) ->
(logits [B, P+1, nseg], extra)`` contract, same attributes the criterion reaches
into (``encoder.embed_tokens_bag``, ``encoder/decoder.seg_embed_tokens``,
``decoder.seg_projection``, ``decoder.tie_seg_projection`` --
criterions/seg_criterion.py:393-405).  The body is not PyTorch: it is the launch
sequence in ``engine.HipEngine`` and it FAILS LOUDLY without a GPU / the HIP library.
"""
import argparse

import torch
import torch.nn as nn

from ...registry import ModelBase, register_model, register_model_architecture, str_bool
from . import config as C
from .engine import HipEngine


class _Node(nn.Module):
    """Plain container; children / parameters are attached by dotted name."""


def _attach(root, name, tensor, is_param, trainable, shared):
    parts = name.split(".")
    mod = root
    for i, p in enumerate(parts[:-1]):
        nxt = getattr(mod, p, None) if not p.isdigit() else (mod[int(p)] if int(p) < len(mod) else None)
        if nxt is None:
            nxt = nn.ModuleList() if (i + 1 < len(parts) - 1 and parts[i + 1].isdigit()) else _Node()
            if p.isdigit():
                mod.append(nxt)
            else:
                setattr(mod, p, nxt)
        mod = nxt
    leaf = parts[-1]
    if is_param:
        prm = shared if shared is not None else nn.Parameter(tensor, requires_grad=trainable)
        mod.register_parameter(leaf, prm)
        return prm
    mod.register_buffer(leaf, tensor)
    return tensor


# ---- the reference's model flags (unify_transformer.py:114-313), as data: (flag, kind[, extra]) ----------------------
# kind: "flag" = store_true, int / float / str = typed option.  Defaults are NOT set here: like fairseq's model group
# (options.py:140-147, argument_default=SUPPRESS) absent flags stay absent and the architecture function fills them.
_MODEL_FLAGS = (
    ("--activation-fn", str), ("--dropout", float), ("--attention-dropout", float),
    ("--activation-dropout", float, "--relu-dropout"), ("--encoder-embed-path", str), ("--encoder-embed-dim", int),
    ("--encoder-ffn-embed-dim", int), ("--encoder-layers", int), ("--encoder-attention-heads", int),
    ("--encoder-normalize-before", "flag"), ("--encoder-learned-pos", "flag"), ("--bitfit", "flag"),
    ("--adapter", "flag"), ("--adapter-dim", int), ("--encoder-prompt", "flag"), ("--encoder-prompt-type", str),
    ("--encoder-prompt-projection", "flag"), ("--encoder-prompt-length", int), ("--encoder-prompt-dim", int),
    ("--decoder-embed-path", str), ("--decoder-embed-dim", int), ("--decoder-ffn-embed-dim", int),
    ("--decoder-layers", int), ("--decoder-attention-heads", int), ("--decoder-learned-pos", "flag"),
    ("--decoder-normalize-before", "flag"), ("--decoder-output-dim", int), ("--freeze-decoder", "flag"),
    ("--decoder-prompt", "flag"), ("--decoder-prompt-type", str), ("--decoder-prompt-length", int),
    ("--decoder-prompt-projection", "flag"), ("--decoder-prompt-dim", int),
    ("--share-decoder-input-output-embed", "flag"), ("--share-all-embeddings", "flag"),
    ("--no-token-positional-embeddings", "flag"), ("--adaptive-softmax-cutoff", str),
    ("--adaptive-softmax-dropout", float), ("--layernorm-embedding", "flag"), ("--no-scale-embedding", "flag"),
    ("--checkpoint-activations", "flag"), ("--offload-activations", "flag"), ("--no-cross-attention", "flag"),
    ("--cross-self-attention", "flag"), ("--encoder-layerdrop", float), ("--decoder-layerdrop", float),
    ("--encoder-layers-to-keep", str), ("--decoder-layers-to-keep", str), ("--quant-noise-pq", float),
    ("--quant-noise-pq-block-size", int), ("--quant-noise-scalar", float), ("--min-params-to-wrap", int),
    ("--resnet-drop-path-rate", float), ("--encoder-drop-path-rate", float), ("--decoder-drop-path-rate", float),
    ("--token-bucket-size", int), ("--image-bucket-size", int), ("--attn-scale-factor", float),
    ("--freeze-resnet", str), ("--freeze-entire-resnet", str), ("--freeze-encoder-embedding", str),
    ("--freeze-decoder-embedding", str), ("--freeze-seg-embedding", str), ("--freeze-encoder-transformer", str),
    ("--freeze-encoder-transformer-layers", int), ("--add-type-embedding", "flag"), ("--interpolate-position", "flag"),
    ("--resnet-type", str), ("--resnet-model-path", str), ("--code-image-size", int),
    ("--patch-layernorm-embedding", "flag"), ("--code-layernorm-embedding", "flag"),
    ("--entangle-position-embedding", "flag"), ("--disable-entangle", "flag"), ("--sync-bn", "flag"),
    ("--scale-attn", "flag"), ("--scale-fc", "flag"), ("--scale-heads", "flag"), ("--scale-resids", "flag"),
    ("--num-seg-tokens", int), ("--decoder-type", str), ("--tie-seg-projection", str), ("--decoder-input-type", str),
    ("--patch-image-size", int), ("--orig-patch-image-size", int),
)
# options whose reference declaration carries an explicit default (unify_transformer.py:154,209-213,218-241,262-313)
_FLAG_DEFAULTS = {
    "bitfit": False, "no_token_positional_embeddings": False, "no_cross_attention": False, "cross_self_attention": False,
    "encoder_layerdrop": 0.0, "decoder_layerdrop": 0.0, "encoder_layers_to_keep": None, "decoder_layers_to_keep": None,
    "quant_noise_pq": 0.0, "quant_noise_pq_block_size": 8, "quant_noise_scalar": 0.0, "min_params_to_wrap": int(1e8),
    "freeze_resnet": "false", "freeze_entire_resnet": "false", "freeze_encoder_embedding": "false",
    "freeze_decoder_embedding": "false", "freeze_seg_embedding": "false", "freeze_encoder_transformer": "false",
    "freeze_encoder_transformer_layers": 0, "num_seg_tokens": 150, "decoder_type": "surrogate",
    "tie_seg_projection": "false", "decoder_input_type": "encoder_input", "patch_image_size": 512,
    "orig_patch_image_size": 512,
}
_RESNET_LAYERS = {"resnet50": (3, 4, 6), "resnet101": (3, 4, 23), "resnet152": (3, 8, 36)}   # resnet.py via encoder_module.py:166-176

# what `segofa_large_architecture` fills in when a flag is absent (models/segofa/segofa.py:352-419); the per-arch
# widths/depths come first (ARCH_DEFAULTS below, :422-467)
_COMMON_ARCH_DEFAULTS = (
    ("encoder_embed_path", None), ("encoder_embed_dim", 1024), ("encoder_ffn_embed_dim", 4096), ("encoder_layers", 12),
    ("encoder_attention_heads", 16), ("encoder_normalize_before", True), ("encoder_learned_pos", True),
    ("decoder_embed_path", None), ("decoder_embed_dim", "=encoder_embed_dim"),
    ("decoder_ffn_embed_dim", "=encoder_ffn_embed_dim"), ("decoder_layers", 12), ("decoder_attention_heads", 16),
    ("decoder_normalize_before", True), ("decoder_learned_pos", True), ("attention_dropout", 0.0), ("relu_dropout", 0.0),
    ("dropout", 0.0), ("max_target_positions", 1024), ("max_source_positions", 1024), ("adaptive_softmax_cutoff", None),
    ("adaptive_softmax_dropout", 0), ("share_decoder_input_output_embed", True), ("share_all_embeddings", True),
    ("decoder_output_dim", "=decoder_embed_dim"), ("decoder_input_dim", "=decoder_embed_dim"), ("no_scale_embedding", True),
    ("layernorm_embedding", True), ("activation_fn", "gelu"), ("pooler_activation_fn", "tanh"), ("pooler_dropout", 0.0),
    ("pooler_classifier", "mlp"), ("resnet_drop_path_rate", 0.0), ("encoder_drop_path_rate", 0.0),
    ("decoder_drop_path_rate", 0.0), ("resnet_type", "resnet152"), ("token_bucket_size", 256), ("image_bucket_size", 42),
    ("freeze_encoder_embedding", False), ("freeze_decoder_embedding", False), ("add_type_embedding", True),
    ("attn_scale_factor", 2), ("code_image_size", 128), ("patch_layernorm_embedding", True),
    ("code_layernorm_embedding", True), ("entangle_position_embedding", False), ("disable_entangle", False),
    ("sync_bn", False), ("scale_attn", False), ("scale_fc", False), ("scale_heads", False), ("scale_resids", False),
    ("orig_patch_image_size", 256),
)
_RESNET_OF_ARCH = {"segofa_base": "resnet101", "segofa_large": "resnet152", "segofa_huge": "resnet152",
                   "segofa_medium": "resnet101", "segofa_tiny": "resnet50"}


def recipe_args(arch="segofa_base", **over):
    """Namespace carrying the model flags of the shipped recipe (run_scripts/IFSeg/coco_unseen.sh:76,89-96,99-103,
    114-121,128-134) -- what `build_model` receives under train.py; used by the bundled harness (no argparse there)."""
    a = argparse.Namespace(
        arch=arch, encoder_normalize_before=True, decoder_normalize_before=True, share_decoder_input_output_embed=True,
        share_all_embeddings=True, layernorm_embedding=True, patch_layernorm_embedding=True, code_layernorm_embedding=True,
        resnet_drop_path_rate=0.0, encoder_drop_path_rate=0.1, decoder_drop_path_rate=0.1, dropout=0.1,
        attention_dropout=0.0, add_type_embedding=True, scale_attn=True, scale_fc=True, scale_heads=True,
        disable_entangle=True, patch_image_size=512, orig_patch_image_size=512, freeze_encoder_embedding="true",
        freeze_decoder_embedding="true", freeze_seg_embedding="true", freeze_entire_resnet="true",
        tie_seg_projection="true", decoder_type="surrogate", decoder_input_type="encoder_output", num_seg_tokens=15)
    for k, v in over.items():
        setattr(a, k, v)
    return a


@register_model("segofa")
class SegOFAModel(ModelBase):
    def __init__(self, cfg: C.SegOFAConfig, seed=None, args=None):
        super().__init__()
        self.cfg = cfg
        self.args = args
        gen = torch.Generator().manual_seed(0 if seed is None else seed)
        spec = C.param_spec(cfg)
        made = {}
        # FrozenBatchNorm2d keeps its affine + stats as *buffers* (frozen_bn.py:30-34)
        for name, (shape, kind, trainable) in spec.items():
            if kind.startswith("alias:"):
                continue
            t = C.init_tensor(name, shape, kind, gen)
            is_param = not kind.startswith("bn_")
            made[name] = _attach(self, name, t, is_param, trainable, None)
        for name, (shape, kind, trainable) in spec.items():
            if kind.startswith("alias:"):
                _attach(self, name, None, True, trainable, made[kind[6:]])
        # derived integer buffers, kept for state_dict parity with the reference
        n_img = (2 * cfg.image_bucket_size - 1) ** 2 + 3
        sb = cfg.seg_bucket_size
        tok = C.token_rp_bucket(cfg.token_bucket_size, cfg.max_source_positions)
        img = C.image_rp_bucket(cfg.image_bucket_size, n_img)
        for side in ("encoder", "decoder"):
            _attach(self, side + ".version", torch.Tensor([3]), False, False, None)
            _attach(self, side + ".token_rp_bucket", tok, False, False, None)
            _attach(self, side + ".image_rp_bucket", img, False, False, None)
        _attach(self, "decoder.seg_rp_bucket", C.image_rp_bucket(sb, (2 * sb - 1) ** 2 + 3), False, False, None)
        ws = cfg.code_image_size // 8
        ipi = (torch.arange(ws)[None, :] + torch.arange(ws)[:, None] * cfg.image_bucket_size + 1).reshape(-1)
        ipi = torch.cat([torch.tensor([0]), ipi, torch.tensor([1024] * 769)])            # decoder_module.py:249-252
        _attach(self, "decoder.image_position_idx", ipi, False, False, None)
        _attach(self, "decoder.bin_id_offset", torch.tensor([max(4, cfg.vocab_size - 1 - 1000)]), False, False, None)
        _attach(self, "decoder.seg_id_offset", torch.tensor([cfg.seg_id_offset]), False, False, None)
        _attach(self, "decoder.region_prefix", torch.tensor([976, 35]), False, False, None)
        self.decoder.tie_seg_projection = True
        self.encoder.padding_idx = 1
        # EmbeddingBag view over the token table, used by the criterion's lazy seg-token init
        bag = nn.EmbeddingBag(cfg.vocab_size, cfg.embed_dim, mode="mean")
        bag.weight = self.encoder.embed_tokens.weight
        self.encoder.embed_tokens_bag = bag
        object.__setattr__(self, "engine", HipEngine(self))
        self.autograd_mode = "inputs"     # "inputs": grads flow through autograd (DDP-compatible);
        #                                   "arena": grads stay in engine.g16 (bundled trainer)

    # -- fairseq model API ---------------------------------------------------------------
    @staticmethod
    def add_args(parser):
        """Every model flag of the reference (unify_transformer.py:114-313), same names, types and explicit defaults,
        so the command line of run_scripts/IFSeg/coco_unseen.sh:73-137 parses unchanged.  `build_model` maps the ones
        that select behaviour of this path onto SegOFAConfig and REFUSES values the HIP path does not implement."""
        for spec in _MODEL_FLAGS:
            flag, kind = spec[0], spec[1]
            names = (flag,) + tuple(spec[2:])
            dest = flag[2:].replace("-", "_")
            kw = {}
            if dest in _FLAG_DEFAULTS:
                kw["default"] = _FLAG_DEFAULTS[dest]
            if kind == "flag":
                parser.add_argument(*names, action="store_true", **kw)
            else:
                parser.add_argument(*names, type=kind, **kw)

    @classmethod
    def build_model(cls, args, task):
        """unify_transformer.py:316-398: vocabulary = len(dictionary) - num_seg_tokens (:402); the architecture
        function fills absent flags; freezes follow `--freeze-*` (:362-374, encoder_module.py:191-197)."""
        arch = getattr(args, "arch", None) or "segofa_base"
        ARCH_FNS[arch](args)
        for k, v in _FLAG_DEFAULTS.items():          # explicit argparse defaults (present even when the flag is not)
            if not hasattr(args, k):
                setattr(args, k, v)
        g = lambda k, d=None: getattr(args, k, d)

        def refuse(what):
            raise NotImplementedError("ifseg_amd.SegOFAModel: %s is not implemented by the MI355X path (the shipped "
                                      "IFSeg recipe, run_scripts/IFSeg/coco_unseen.sh:73-137, never asks for it)" % what)

        must_true = ("encoder_normalize_before", "decoder_normalize_before", "layernorm_embedding",
                     "patch_layernorm_embedding", "add_type_embedding", "share_all_embeddings", "no_scale_embedding",
                     "scale_attn", "scale_fc", "scale_heads", "disable_entangle")
        for k in must_true:
            if not g(k, False):
                refuse("--%s off" % k.replace("_", "-"))
        must_false = ("scale_resids", "entangle_position_embedding", "bitfit", "adapter", "encoder_prompt", "decoder_prompt",
                      "sync_bn", "interpolate_position", "no_cross_attention", "cross_self_attention",
                      "no_token_positional_embeddings", "freeze_decoder")
        for k in must_false:
            if g(k, False):
                refuse("--%s" % k.replace("_", "-"))
        for k in ("resnet_drop_path_rate", "encoder_layerdrop", "decoder_layerdrop", "quant_noise_pq", "quant_noise_scalar"):
            if float(g(k, 0.0) or 0.0) != 0.0:
                refuse("--%s > 0" % k.replace("_", "-"))
        if g("activation_fn", "gelu") != "gelu":
            refuse("--activation-fn %s" % g("activation_fn"))
        if g("decoder_type", "surrogate") != "surrogate":
            refuse("--decoder-type %s (only 'surrogate' exists in the reference: decoder_module.py:465-468)" % g("decoder_type"))
        if g("decoder_input_type") != "encoder_output":
            refuse("--decoder-input-type %s" % g("decoder_input_type"))
        if not str_bool(g("tie_seg_projection")):
            refuse("--tie-seg-projection=false")
        if not str_bool(g("freeze_entire_resnet")):
            refuse("--freeze-entire-resnet=false (gradients of the ResNet trunk / image_proj)")
        # --freeze-encoder-embedding / --freeze-decoder-embedding / --freeze-seg-embedding false: built (the token table is ONE
        # tensor under share_all_embeddings -- unify_transformer.py:340-372 -- so either flag freezes it)
        if str_bool(g("freeze_encoder_transformer")) or int(g("freeze_encoder_transformer_layers", 0) or 0):
            refuse("--freeze-encoder-transformer")
        if g("encoder_layers_to_keep") or g("decoder_layers_to_keep"):
            refuse("--{en,de}coder-layers-to-keep")
        if (g("decoder_embed_dim") != g("encoder_embed_dim") or g("decoder_ffn_embed_dim") != g("encoder_ffn_embed_dim")
                or g("decoder_attention_heads") != g("encoder_attention_heads")):
            refuse("decoder width / heads different from the encoder's")
        if g("resnet_type") not in _RESNET_LAYERS:
            refuse("--resnet-type %s" % g("resnet_type"))
        src_dict, tgt_dict = task.source_dictionary, task.target_dictionary
        if src_dict != tgt_dict:
            raise ValueError("--share-all-embeddings requires a joined dictionary")      # unify_transformer.py:337-338
        nseg = int(g("num_seg_tokens"))
        cfg = C.SegOFAConfig(
            embed_dim=int(g("encoder_embed_dim")), ffn_dim=int(g("encoder_ffn_embed_dim")),
            heads=int(g("encoder_attention_heads")), enc_layers=int(g("encoder_layers")), dec_layers=int(g("decoder_layers")),
            resnet_layers=_RESNET_LAYERS[g("resnet_type")], num_seg_tokens=nseg, vocab_size=len(src_dict) - nseg,
            patch_image_size=int(g("patch_image_size")), orig_patch_image_size=int(g("orig_patch_image_size")),
            image_bucket_size=int(g("image_bucket_size")), token_bucket_size=int(g("token_bucket_size")),
            attn_scale_factor=float(g("attn_scale_factor")), max_source_positions=int(g("max_source_positions", 1024) or 1024),
            max_target_positions=int(g("max_target_positions", 1024) or 1024), code_image_size=int(g("code_image_size")),
            dropout=float(g("dropout", 0.0) or 0.0), attention_dropout=float(g("attention_dropout", 0.0) or 0.0),
            activation_dropout=float(g("activation_dropout", 0.0) or 0.0) or float(g("relu_dropout", 0.0) or 0.0),
            encoder_drop_path_rate=float(g("encoder_drop_path_rate", 0.0) or 0.0),
            decoder_drop_path_rate=float(g("decoder_drop_path_rate", 0.0) or 0.0),
            freeze_embeddings=str_bool(g("freeze_encoder_embedding")) or str_bool(g("freeze_decoder_embedding")),
            freeze_seg_embedding=str_bool(g("freeze_seg_embedding")))
        model = cls(cfg, args=args)
        model.encoder.dictionary = src_dict
        model.decoder.dictionary = tgt_dict
        return model

    def max_positions(self):
        return (self.cfg.max_source_positions, self.cfg.max_target_positions)

    def set_num_updates(self, n):
        self.num_updates = n

    def upgrade_state_dict_named(self, state_dict, name):
        """Checkpoint up-conversion of the reference, in its order (fairseq_model.py:120-140 walks the children first):
          encoder (encoder_module.py:943-987) / decoder (decoder_module.py:892-940): drop `decoder.output_projection`,
          fill keys the checkpoint lacks with the current values, grow `embed_image_positions` with N(0, C^-0.5) rows,
          delete seg-token tables whose class count differs;
          model (segofa.py:255-287): drop the <mask> row when the dictionary has none, append N(0, C^-0.5) rows to
          the shared token table when the checkpoint's vocabulary is smaller (ofa_base.pt: 59457 rows, model: 59458).
        Random rows come from torch's global generator in the reference's order (encoder, decoder, model)."""
        prefix = name + "." if name != "" else ""
        sd = state_dict
        mine = self.state_dict()
        sd.pop(prefix + "decoder.output_projection.weight", None)
        for side in ("encoder", "decoder"):
            sp = prefix + side + "."
            if side == "decoder":
                sd[sp + "image_position_idx"] = mine["decoder.image_position_idx"]
            for k, v in mine.items():
                if k.startswith(side + ".") and (prefix + k) not in sd:
                    sd[prefix + k] = v
            key = sp + "embed_image_positions.weight"
            have, want = sd[key].shape[0], mine[side + ".embed_image_positions.weight"].shape[0]
            if have < want:
                add = torch.zeros(want - have, sd[key].shape[1])
                nn.init.normal_(add, mean=0, std=sd[key].shape[1] ** -0.5)
                sd[key] = torch.cat([sd[key], add.to(dtype=sd[key].dtype, device=sd[key].device)])
            for leaf in (("seg_embed_tokens.weight",) if side == "encoder" else ("seg_embed_tokens.weight", "seg_projection.weight")):
                if sp + leaf in sd and sd[sp + leaf].shape[0] != mine[side + "." + leaf].shape[0]:
                    del sd[sp + leaf]
        ek, dk = prefix + "encoder.embed_tokens.weight", prefix + "decoder.embed_tokens.weight"
        loaded = sd[ek].shape[0]
        want = self.cfg.vocab_size
        dictionary = getattr(self.encoder, "dictionary", None)
        if loaded == want + 1 and (dictionary is None or "<mask>" not in dictionary):
            for k in (ek, dk, prefix + "encoder.output_projection.weight", prefix + "decoder.output_projection.weight"):
                if k in sd:
                    sd[k] = sd[k][:-1, :]
        if loaded < want:
            add = torch.zeros(want - loaded, sd[ek].shape[1])
            nn.init.normal_(add, mean=0, std=sd[ek].shape[1] ** -0.5)
            add = add.to(dtype=sd[ek].dtype, device=sd[ek].device)
            sd[ek] = torch.cat([sd[ek], add])
            sd[dk] = torch.cat([sd[dk], add])
        # the EmbeddingBag view shares the token table (encoder_module.py:147-148): keep it consistent
        bk = prefix + "encoder.embed_tokens_bag.weight"
        if bk in sd and sd[bk].shape[0] != sd[ek].shape[0]:
            sd[bk] = sd[ek]
        return state_dict

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        """fairseq_model.py:103-118: up-convert, then the plain nn.Module load."""
        self.upgrade_state_dict_named(state_dict, "")
        out = nn.Module.load_state_dict(self, state_dict, strict=strict)
        eng = self.engine
        if eng.packed:
            # the parameters are views of the bf16 arena (copied into in place above); the fp32 master copy takes the
            # checkpoint's own values, the folded ResNet / padded seg projection are re-derived
            for n in eng.trainable_names():
                if n in state_dict:
                    eng.Wf(n).copy_(state_dict[n].to(eng.device, torch.float32).view(eng.shapes[n]))
            eng._master_stale = False
            eng._wver += 1                  # cached resized rel-pos biases belong to the old weights
            eng._pack_resnet()
            eng.refresh_frozen()
        return out

    def seg_tokens_from_text(self, token_ids):
        """mean token embedding per category name (criterions/seg_criterion.py:387-393: `embed_tokens_bag(ids, offsets)`)
        on the device: token_ids = list of 1-D int64 tensors -> bf16 [len(token_ids), C]."""
        eng = self.engine
        dev = eng.device if eng.packed else next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("ifseg_amd.SegOFAModel.seg_tokens_from_text runs on the MI355X: move the model to the device first")
        if not eng.packed or eng.device != dev:
            eng.pack(dev)
        from ... import hip
        ids = torch.cat([t.reshape(-1) for t in token_ids]).to(dev).view(1, -1).contiguous()
        ends = torch.tensor([t.numel() for t in token_ids], dtype=torch.long).cumsum(0).to(dev)
        out = torch.empty(len(token_ids), self.cfg.embed_dim, dtype=torch.bfloat16, device=dev)
        prev = hip.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            hip.embed_bag_mean(eng.W("encoder.embed_tokens.weight"), ids, ends, None, out)
        finally:
            hip.set_stream(prev)
        return out

    def _apply(self, fn, *a, **k):
        """.cuda()/.half()/.to(bf16) of the trainer (trainer.py:95-111) would re-allocate
        every parameter; the arena is re-packed lazily on the next forward instead."""
        r = super()._apply(fn, *a, **k)
        self.engine.packed = False
        return r

    # -- forward -------------------------------------------------------------------------
    def forward(self, src_tokens=None, src_lengths=None, prev_output_tokens=None, patch_images=None,
                patch_images_2=None, patch_masks=None, code_masks=None, sample_patch_num=None, features_only=False,
                full_context_alignment=False, classification_head_name=None, token_embeddings=None,
                return_all_hiddens=False, alignment_layer=None, alignment_heads=None, encoder_only=False,
                aux_input=None):
        eng = self.engine
        x, extra = None, {}
        if src_tokens is not None:
            if patch_images is None or not patch_images.is_cuda:
                raise RuntimeError("ifseg_amd.SegOFAModel runs only on an MI355X: patch_images must be a device tensor "
                                   "(there is no CPU / PyTorch fallback)")
            if patch_masks is not None:
                # validated without draining the queue (see HipEngine.deferred_check)
                from ... import hip as _hip
                eng.deferred_check(patch_masks, lambda t: t.all().logical_not(), "masked-out patch images are not supported",
                                   native=(_hip.CHECK_ANY_ZERO_BYTE, 0, 1) if patch_masks.dtype == torch.bool else None)
            x, extra = self._run(src_tokens, patch_images, prev_output_tokens, bool(full_context_alignment), None)
        if aux_input is not None:
            # image-free branch (segofa.py:136-151): encoder on the artificial image, decoder with its defaults
            # (causal).  The engine keeps the activations of ONE forward: when both branches are requested the
            # second forward (this one) is the one a following backward differentiates.
            a_src = aux_input.get("src_tokens")
            if a_src is None or not a_src.is_cuda:
                raise RuntimeError("ifseg_amd.SegOFAModel runs only on an MI355X: aux_input tensors must be on the device")
            if x is not None and torch.is_grad_enabled() and self.training:
                raise NotImplementedError("gradients through both the image and the image-free branch of one call "
                                          "(the reference's criterion never asks for it: seg_criterion.py:179-186)")
            bag = (aux_input.get("patch_images"), aux_input.get("patch_masks"))
            extra["aux_output"] = self._run(a_src, None, aux_input.get("prev_output_tokens"), False, bag)
        return x, extra

    def _run(self, src_tokens, patch_images, prev_output_tokens, full, bag):
        eng = self.engine
        dev = src_tokens.device
        if not eng.packed or eng.device != dev:
            eng.pack(dev)
        eng.refresh_frozen()
        # (dropout / DropPath / attention and activation dropout follow `need_grad`, which requires model.training: an eval()
        # forward -- with or without grad mode -- applies none of them, as the reference's dropout modules in eval)
        need_grad = torch.is_grad_enabled() and self.training
        params = eng.trainable_params() if (need_grad and self.autograd_mode == "inputs") else ()
        anchor = torch.zeros(1, device=dev, requires_grad=need_grad)
        logits = _SegOFAFn.apply(eng, src_tokens, patch_images, prev_output_tokens, full, self.autograd_mode, anchor,
                                 bag, *params)
        ctx = eng.ctx
        B, T, Cc = ctx["enc_out"].shape
        extra = {
            "encoder_returns": {
                "encoder_out": [ctx["enc_out"].transpose(0, 1)],
                "image_embed_shape": [(ctx["h"], ctx["w"])],
                "image_embed_before_proj": [ctx["feat"]],
                "position_embeddings": [eng.ws["e_pos_all"]],
                "resized_grid": bool(ctx.get("resized", False)),
                # (encoder_module.py:730-752,838: True at <pad> source tokens -- the P patch positions are never padding)
                "encoder_padding_mask": [torch.cat([torch.zeros(B, ctx["P"], dtype=torch.bool, device=logits.device), ~ctx["nonpad"]], 1)
                                         if ctx.get("nonpad") is not None else torch.zeros(B, T, dtype=torch.bool, device=logits.device)],
            },
            "attn": [None],
            "logits_padded": eng.ws["logits_pad"],
        }
        return logits, extra


class _SegOFAFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, eng, src_tokens, patch_images, prev, full, mode, anchor, bag, *params):
        logits_pad, _ = eng.forward(src_tokens, patch_images, prev, full, need_grad=bool(anchor.requires_grad), bag=bag)
        ctx.eng, ctx.mode, ctx.nparams = eng, mode, len(params)
        return logits_pad[:, :, : eng.cfg.num_seg_tokens]

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.eng
        eng.backward(dlogits)
        grads = ()
        if ctx.mode == "inputs":
            grads = tuple(eng.G(n) for n in eng.trainable_names())
        return (None,) * 8 + grads


ARCH_FNS = {}


def _make_arch(arch_name):
    def fn(args):
        """register_model_architecture default filler (models/segofa/segofa.py:351-467): a flag that is ABSENT gets the
        architecture's value, a present one is kept (the reference's `getattr(args, k, default)`)."""
        a = C.ARCHS[arch_name]
        first = (("encoder_embed_dim", a["embed_dim"]), ("encoder_ffn_embed_dim", a["ffn_dim"]),
                 ("encoder_layers", a["enc_layers"]), ("encoder_attention_heads", a["heads"]),
                 ("decoder_layers", a["dec_layers"]), ("decoder_attention_heads", a["heads"]),
                 ("resnet_type", _RESNET_OF_ARCH[arch_name]))
        for k, v in first + _COMMON_ARCH_DEFAULTS:
            if not hasattr(args, k):
                setattr(args, k, getattr(args, v[1:]) if isinstance(v, str) and v.startswith("=") else v)
    fn.__name__ = arch_name + "_architecture"
    ARCH_FNS[arch_name] = fn
    return register_model_architecture("segofa", arch_name)(fn)


segofa_large_architecture = _make_arch("segofa_large")
segofa_base_architecture = _make_arch("segofa_base")
segofa_huge_architecture = _make_arch("segofa_huge")
segofa_medium_architecture = _make_arch("segofa_medium")
segofa_tiny_architecture = _make_arch("segofa_tiny")
