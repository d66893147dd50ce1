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

from ...registry import register_model, register_model_architecture
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


@register_model("segofa")
class SegOFAModel(nn.Module):
    def __init__(self, cfg: C.SegOFAConfig, seed=None):
        super().__init__()
        self.cfg = cfg
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
        """Model flags of the reference that select behaviour of this path
        (unify_transformer.py:114-313); others are accepted and ignored."""
        for flag in ("--patch-image-size", "--orig-patch-image-size", "--num-seg-tokens", "--image-bucket-size",
                     "--token-bucket-size", "--attn-scale-factor"):
            parser.add_argument(flag, type=int)
        for flag in ("--freeze-encoder-embedding", "--freeze-decoder-embedding", "--freeze-seg-embedding",
                     "--freeze-entire-resnet", "--tie-seg-projection", "--decoder-type", "--decoder-input-type"):
            parser.add_argument(flag, type=str)

    @classmethod
    def build_model(cls, args, task):
        """unify_transformer.py:316-398: vocabulary = len(dictionary) - num_seg_tokens."""
        arch = getattr(args, "arch", "segofa_base")
        nseg = int(getattr(args, "num_seg_tokens", 15))
        over = dict(num_seg_tokens=nseg, vocab_size=len(task.source_dictionary) - nseg,
                    patch_image_size=int(getattr(args, "patch_image_size", 512)),
                    orig_patch_image_size=int(getattr(args, "orig_patch_image_size", 512)))
        if getattr(args, "decoder_type", "surrogate") != "surrogate":
            raise NotImplementedError("only decoder_type=surrogate exists in the reference (decoder_module.py:465-468)")
        return cls(C.make_config(arch, **over))

    def max_positions(self):
        return (self.cfg.max_source_positions, self.cfg.max_target_positions)

    def set_num_updates(self, n):
        self.num_updates = n

    def upgrade_state_dict_named(self, state_dict, name):
        """Grow an OFA checkpoint's vocabulary by the seg tokens (segofa.py:265-287) and
        fill keys the checkpoint lacks with the current values (strict=False loading)."""
        mine = self.state_dict()
        for k, v in mine.items():
            if k not in state_dict:
                state_dict[k] = v
        return state_dict

    def load_state_dict(self, state_dict, strict=True, model_cfg=None, args=None):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.engine.packed:
            self.engine._pack_resnet()
            self.engine.refresh_frozen()
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
                eng.deferred_check(patch_masks, lambda t: t.all().logical_not(), "masked-out patch images are not supported")
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
                "encoder_padding_mask": [torch.zeros(B, T, dtype=torch.bool, device=logits.device)],
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


def _make_arch(arch_name):
    def fn(args):
        """register_model_architecture default filler (models/segofa/segofa.py:351-467)."""
        a = C.ARCHS[arch_name]
        defaults = dict(encoder_embed_dim=a["embed_dim"], encoder_ffn_embed_dim=a["ffn_dim"],
                        encoder_layers=a["enc_layers"], decoder_layers=a["dec_layers"],
                        encoder_attention_heads=a["heads"], decoder_attention_heads=a["heads"],
                        token_bucket_size=256, image_bucket_size=42, attn_scale_factor=2,
                        encoder_normalize_before=True, decoder_normalize_before=True, no_scale_embedding=True,
                        layernorm_embedding=True, patch_layernorm_embedding=True, add_type_embedding=True,
                        share_all_embeddings=True, activation_fn="gelu")
        for k, v in defaults.items():
            if getattr(args, k, None) is None:
                setattr(args, k, v)
        if getattr(args, "arch", None) is None:
            args.arch = arch_name
    fn.__name__ = arch_name + "_architecture"
    return register_model_architecture("segofa", arch_name)(fn)


segofa_large_architecture = _make_arch("segofa_large")
segofa_base_architecture = _make_arch("segofa_base")
segofa_huge_architecture = _make_arch("segofa_huge")
segofa_medium_architecture = _make_arch("segofa_medium")
segofa_tiny_architecture = _make_arch("segofa_tiny")
