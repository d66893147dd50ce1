"""HIP execution engine for SegOFA: explicit forward + backward on MI355X.

This is the host side of the hot path.  It owns
  * one flat bf16 parameter arena and one flat bf16 gradient arena (the
    nn.Parameters of the module tree are views into the arena, so the reference's
    state_dict keys / load_state_dict keep working; q|k|v weights are laid out
    contiguously so one GEMM serves the three projections),
  * a static activation workspace (allocated once per input shape, reused every
    step: no allocator traffic inside the step, hipGraph-capturable),
  * the launch sequence of the hand-written kernels in csrc/ through the C ABI
    (ifseg_amd.hip).  No torch compute op touches an activation; torch is used for
    memory, streams and a handful of <=H-element scalar fix-ups.

Reference call graph reproduced:
  SegOFAModel.forward            models/segofa/segofa.py:69-153
  TransformerEncoder.encode      models/segofa/encoder_module.py:677-851
  TransformerEncoderLayer        models/segofa/unify_transformer_layer.py:222-292
  extract_features_..surrogate   models/segofa/decoder_module.py:486-677
  TransformerDecoderLayer        models/segofa/unify_transformer_layer.py:431-581
  MultiheadAttention             models/segofa/unify_multihead_attention.py:327-513
  ResNet / FrozenBatchNorm2d     models/segofa/resnet.py:215-229, frozen_bn.py:36-57
and their autograd.  Internal token order of the decoder is [patches..., bos]
(bos moved to the end so the seg grid is tile aligned); logits are produced in the
reference order [bos, patches...].
"""
import contextlib
import math
import os
import time

import torch

from ... import hip, lab
from . import resized as R
from .config import image_rp_bucket, token_bucket_of_delta

BF = torch.bfloat16
# TIMING EXPERIMENTS ONLY (wrong results): kernels left out of the step to bound what optimising them could return,
# e.g. IFSEG_EXP_SKIP=lnwide,dq (tools/skip_bound.sh; DESIGN section 4 "Round 3")
_POISON = lab.get("POISON_WS", "") == "1"
_EXP_SKIP = set(filter(None, lab.get("EXP_SKIP", "").split(",")))


def _pad8(n):
    return (n + 7) // 8 * 8


class HipEngine:
    def __init__(self, model):
        self.model = model
        self.cfg = model.cfg
        self.packed = False
        self.device = None
        # activation workspaces: `_ws_grad` belongs to the forward a backward will differentiate; a no-grad
        # forward issued while that backward is still pending (the image-free criterion evaluates the real
        # images for its metrics between the two, seg_criterion.py:179-186) runs in `_ws_aux` so that it does
        # not overwrite the saved activations.  `ws` / `saved` / `ctx` always name the current one.
        self._ws_grad, self._ws_aux = {}, {}
        self._saved_grad, self._saved_aux = {}, {}
        self.ws, self.saved = self._ws_grad, self._saved_grad
        self._gctx = None                 # ctx of the forward awaiting its backward
        self._popt = None                 # events of a still-running optimizer (params_pending)
        self.ctx_building = None          # ctx of the forward being enqueued
        self._views = {}
        self.geo = {}
        self.ctx = None
        self.grad_ready_hook = None      # callable(prefix) once every gradient under `prefix` is final
        self.drop_on = False
        self.step_seed = 0               # set by the trainer (seed + num_updates, trainer.py:1297)
        # the per-update part of every dropout / DropPath seed lives in DEVICE memory (`step_dev[0]`, added to the site's
        # seed by the kernels): a HIP-graph-captured step then draws new masks on every replay
        self.step_dev, self._step_pin, self._step_pin_i, self._step_dev_val = None, None, 0, None
        # backward: everything that only produces PARAMETER gradients (dW GEMMs, bias / LayerNorm / table
        # reductions) is enqueued on a second HIP stream and overlaps the dX chain on the main stream
        self._pending_checks, self._checked_kinds = [], set()
        self.master_owned = False        # True once the bundled Trainer updates `master` and `p16` together
        self.overlap = lab.get("NO_OVERLAP") is None
        self.delta_fused = lab.get("NO_DELTA_FUSE") is None     # delta from the out_proj dX GEMM's epilogue
        self._side = None
        self._trunk_stream, self._pf, self._pf_slot = None, None, 0
        self._pf_more = []               # features of the batches AFTER the next one (one trunk pass over several batches)
        self.trunk_lookahead = max(1, int(lab.get("TRUNK_LOOKAHEAD", "2")))
        self._pf_request = None          # images of the next batch (set by the trainer; consumed by the next forward)
        self._pending, self._in_flush = [], False
        # weight-gradient GEMMs of a layer are collected and launched as ONE grouped GEMM at the end of the layer's
        # backward (no split-K slabs / reduction launches: hip.linear_dw_group); IFSEG_NO_DW_GROUP=1: one GEMM each
        self._dw_tasks = []
        self._kfix_tasks = []       # (k_proj dW [C, C], k_proj db [C], projection input x [rows, C], tag of a shared column-sum workspace or None)
        self._ln_red_tasks, self._ln_red_acc = [], []    # (partials, [2, C] gradient view, ...) of LayerNorm backward launches
        self.dw_grouped = lab.get("NO_DW_GROUP") is None
        # IFSEG_DRAIN_TIMING=1: event pair around the end-of-backward join (bench.py reports `end_of_backward_wait_ms`).
        # Measured 0.49 ms per C2 step.  Moving the attentions' bias-gradient reductions to the third stream shortened it to
        # 0.41 ms and made the step 0.45 ms SLOWER (a third busy queue under the dX chain); for the last layer only: 0.42 ms,
        # step unchanged.  Not adopted.
        self.drain_timing = [] if lab.get("DRAIN_TIMING") else None
        # IFSEG_PHASE_TIMING=1: a few timing events per step on the main stream (`mark`, with the host's clock next to each);
        # bench.py reports the phases' GPU and host durations over the timed region (`phase_ms`, `phase_host_ms`)
        self.marks = [] if lab.get("PHASE_TIMING") else None
        self.attn_bwd_timing = None      # {"stride": n, "seen": 0, "pairs": []} while bench.py times the attention backward
        self._rb_cache, self._wver = {}, 0   # dense resized rel-pos biases (eval on other aspect ratios), weights version
        # ffn_layernorm(gelu(fc1)) backward folded into the fc2 dX GEMM's epilogue (csrc/rowops.hip "FFN's ffn_layernorm",
        # csrc/gemm.hip EPI_GLN): no 3072-wide LayerNorm-backward pass.  IFSEG_NO_FFN_LN_FUSE=1: the stand-alone kernel.
        self.ffn_ln_fused = lab.get("NO_FFN_LN_FUSE") is None
        # attention backward with the batch inside the workgroup (csrc/attention_bi.hip): the bias is a dense batch-invariant
        # operand built once per layer from parameters, sum_b dS leaves the dQ kernel once per tile.  IFSEG_ATTN_BI=0: the
        # round-3 kernels (one workgroup per (batch, head, tile), bias regenerated per batch element).
        # Default "auto" = "1": all three attentions of a training step take this path whenever the grid width is a multiple of
        # 8 and <= 64 (measured in the step, DESIGN round 4, same box: Base C2 17.44 vs 18.03 ms with the round-3 kernels;
        # SegOFA-Large at 640^2 83.4 vs 101.7 ms).  Until the dense bias lost its transposed copy the causal decoder
        # self-attention was 0.1 ms better off on the round-3 kernels' 32-wide fast path; now it is 0.06 ms worse.  "0": nowhere.
        self.attn_bi = lab.get("ATTN_BI", "auto")
        # k_proj.weight gradients without the product of dK's spurious column sum and the token mean of the projection's input
        # (an exact identity: sum_j dK_j = 0; csrc/rowops.hip ifseg_kproj_common_mode).  IFSEG_NO_KPROJ_FIX=1: as computed.
        self.kproj_fix = lab.get("NO_KPROJ_FIX") is None
        # the forward through the batch-inner kernel wherever the layer's dense bias exists (IFSEG_ATTN_BI_FWD=0: round-3 forward)
        self.bi_fwd = lab.get("ATTN_BI_FWD", "1") != "0"
        self._ffn_pg_tasks = []
        self._train_fwd = False
        # where the NEXT batch's frozen-trunk pass is launched: "fwd" = at the start of this step's forward, "e<k>" = when
        # the backward reaches encoder layer k, "end" = after the last backward kernel of the main stream
        self.trunk_at = lab.get("TRUNK_AT", "fwd")
        self._bt = ""                    # tag of the backward block being processed (unique gradient buffers)

    # ------------------------------------------------------------------ packing
    def _arena_order(self):
        cfg = self.cfg
        names = dict(self.model.named_parameters())   # unique tensors (ties appear once)
        order, fused = [], {}

        def add(n):
            if n in names and n not in order:
                order.append(n)

        def mha(p, cross=False):
            # every (fused) Linear is laid out weight-then-bias so that dW and db are one contiguous range
            add(p + ".c_attn")
            if cross:       # q projects the decoder stream, k|v (one GEMM) the encoder output
                add(p + ".q_proj.weight"); add(p + ".q_proj.bias")
                for suf in (".weight", ".bias"):
                    for pr in ("k_proj", "v_proj"):
                        add("%s.%s%s" % (p, pr, suf))
            else:
                for suf in (".weight", ".bias"):
                    for pr in ("q_proj", "k_proj", "v_proj"):
                        add("%s.%s%s" % (p, pr, suf))
            add(p + ".out_proj.weight"); add(p + ".out_proj.bias")

        def lnp(p):
            add(p + ".weight"); add(p + ".bias")

        e = "encoder."
        for p in ("layernorm_embedding", "patch_layernorm_embedding", "pos_ln", "image_pos_ln", "layer_norm"):
            lnp(e + p)
        add(e + "type_embedding.weight"); add(e + "embed_positions.weight"); add(e + "embed_image_positions.weight")
        for suf in (".weight", ".bias"):
            add(e + "pos_q_linear" + suf); add(e + "pos_k_linear" + suf)
        for i in range(cfg.enc_layers):
            p = "%slayers.%d." % (e, i)
            mha(p + "self_attn")
            for q in ("self_attn_layer_norm", "attn_ln", "final_layer_norm", "ffn_layernorm"):
                lnp(p + q)
            for q in ("fc1", "fc2"):
                add(p + q + ".weight"); add(p + q + ".bias")
            add("%stoken_rel_pos_table_list.%d.weight" % (e, i)); add("%simage_rel_pos_table_list.%d.weight" % (e, i))
        d = "decoder."
        for p in ("layernorm_embedding", "seg_pos_ln", "layer_norm"):
            lnp(d + p)
        add(d + "embed_seg_positions.weight")
        for suf in (".weight", ".bias"):
            add(d + "self_pos_q_linear" + suf); add(d + "self_pos_k_linear" + suf)
        for n in ("cross_pos_q_linear", "cross_pos_k_linear"):
            add(d + n + ".weight"); add(d + n + ".bias")
        for i in range(cfg.dec_layers):
            p = "%slayers.%d." % (d, i)
            mha(p + "self_attn"); mha(p + "encoder_attn", cross=True)
            for q in ("self_attn_layer_norm", "self_attn_ln", "encoder_attn_layer_norm", "cross_attn_ln",
                      "final_layer_norm", "ffn_layernorm"):
                lnp(p + q)
            for q in ("fc1", "fc2"):
                add(p + q + ".weight"); add(p + q + ".bias")
            add("%sseg_rel_pos_table_list.%d.weight" % (d, i))
        # trainable parameters the path never touches (no grad in the reference either)
        rest_train = [n for n, p in names.items() if n not in order and p.requires_grad and "embed_images" not in n]
        order += rest_train
        n_train_names = len(order)
        frozen = [n for n in names if n not in order and "embed_images" not in n]
        order += frozen
        return order, n_train_names

    def pack(self, device):
        """(Re)build the arenas on `device` from the current parameter values."""
        m = self.model
        names = dict(m.named_parameters())
        order, n_train_names = self._arena_order()
        unsupported = [n for n in order[n_train_names:] if names[n].requires_grad]
        unsupported += [n for n, p in names.items() if ("embed_images" in n or "image_proj" in n) and p.requires_grad]
        # token table / seg embeddings: frozen by every shipped script; trainable ones (--freeze-*-embedding false,
        # unify_transformer.py:362-373) get their gradients from `_embed_grads` / the seg-projection dW
        self.train_tok = bool(names["encoder.embed_tokens.weight"].requires_grad)
        self.train_seg = bool(names["encoder.seg_embed_tokens.weight"].requires_grad)
        if unsupported:
            raise NotImplementedError(
                "ifseg_amd HIP engine: gradients for %s are not implemented (the shipped IFSeg recipe freezes "
                "them: coco_unseen.sh:31-33,76)" % unsupported[:4])
        offs, off = {}, 0
        for n in order:
            offs[n] = off
            off += _pad8(names[n].numel())
        total = off
        n_train = offs[order[n_train_names]] if n_train_names < len(order) else total
        self._views = {}
        p16 = torch.zeros(total, dtype=BF, device=device)
        g16 = torch.zeros(n_train, dtype=BF, device=device)
        master = torch.zeros(n_train, dtype=torch.float32, device=device)
        for n in order:
            p = names[n]
            v = p.data.detach().to(device=device, dtype=torch.float32).reshape(-1)
            o = offs[n]
            if o < n_train:
                master[o:o + v.numel()] = v
            p16[o:o + v.numel()] = v.to(BF)
            p.data = p16[o:o + v.numel()].view(p.shape)
        self.p16, self.g16, self.master_init = p16, g16, master
        # fp32 copy of the trainable part of the arena.  The bundled Trainer runs Adam on it (`master_owned`); the kernels
        # read the SMALL parameters whose bf16 rounding is coherent per channel -- LayerNorm gains / biases and the
        # per-head c_attn gains -- from here (`Wf`): their rounding alone costs 1.0e-2 of logits rel-L2 on SegOFA-Base
        # (tools/err_budget2.py).  With an external optimizer (fairseq's, writing the bf16 parameters) the copy is
        # re-synchronised after every backward: entries whose bf16 value changed take it, the others stay exact.
        self.master = master
        self._master_stale = False
        self.offs, self.n_train, self.order = offs, n_train, order
        self.shapes = {n: tuple(names[n].shape) for n in order}
        self.names = names
        # everything else (ResNet trunk, buffers) just moves to the device
        for n, p in names.items():
            if "embed_images" in n:
                p.data = p.data.to(device)
        for b_name, b in m.named_buffers():
            if b.device != device:
                b.data = b.data.to(device)
        self.device = device
        self._pack_resnet()
        self._pack_misc()
        self.packed = True
        self._wver += 1
        self._rb_cache.clear()
        self._ws_grad.clear(); self._ws_aux.clear()
        self._saved_grad.clear(); self._saved_aux.clear()
        self._gctx = None
        self.geo.clear()

    def trainable_names(self):
        return [n for n in self.order if self.offs[n] < self.n_train and self.names[n].requires_grad]

    def trainable_params(self):
        return tuple(self.names[n] for n in self.trainable_names())

    # views into the arenas are cached: the arenas are allocated once per pack()
    def W(self, n):
        v = self._views.get(n)
        if v is None:
            o, sh = self.offs[n], self.shapes[n]
            v = self._views[n] = self.p16[o:o + math.prod(sh)].view(sh)
        return v

    def Wf(self, n):
        """fp32 view of a trainable parameter in the master copy (LayerNorm gains / biases, c_attn)"""
        v = self._views.get(("f", n))
        if v is None:
            o, sh = self.offs[n], self.shapes[n]
            if o >= self.n_train:
                raise KeyError("%s is not in the fp32 master copy (frozen parameter)" % n)
            v = self._views[("f", n)] = self.master[o:o + math.prod(sh)].view(sh)
        return v

    def G(self, n):
        v = self._views.get(("g", n))
        if v is None:
            o, sh = self.offs[n], self.shapes[n]
            v = self._views[("g", n)] = self.g16[o:o + math.prod(sh)].view(sh)
        return v

    def _fused(self, arena, first, rows, cols=None):
        """view of `rows` x cols starting at parameter `first` (parameters laid out contiguously)"""
        key = (arena is self.g16, first, rows, cols)
        v = self._views.get(key)
        if v is None:
            o = self.offs[first]
            n = rows * (cols or 1)
            v = arena[o:o + n]
            v = self._views[key] = v.view(rows, cols) if cols else v
        return v

    def _pack_resnet(self):
        """Fold FrozenBN into the conv weights (frozen_bn.py:39-40: scale = w*rsqrt(var+eps),
        bias = b - mean*scale) and lay the kernels out [Cout][KH][KW][Cin] bf16."""
        sd = {k: v for k, v in self.model.state_dict().items() if k.startswith("encoder.embed_images.")}
        pre = "encoder.embed_images."
        dev = self.device

        def fold(conv, bn):
            w = sd[pre + conv + ".weight"].float()
            scale = sd[pre + bn + ".weight"].float() * torch.rsqrt(sd[pre + bn + ".running_var"].float() + 1e-5)
            shift = sd[pre + bn + ".bias"].float() - sd[pre + bn + ".running_mean"].float() * scale
            return w * scale.view(-1, 1, 1, 1), shift

        w, sh = fold("conv1", "bn1")
        self.stem_w = w.permute(2, 3, 1, 0).contiguous().to(dev)          # [7][7][3][64] fp32
        # (the matrix-core stem kernel takes bf16 weights, like every other convolution of the trunk; IFSEG_STEM_DIRECT=1: the
        # round-1 direct fp32 kernel)
        if lab.get("STEM_DIRECT") is None:
            self.stem_w = hip.stem_weights_mfma(self.stem_w)
        self.stem_shift = sh.contiguous().to(dev)
        self.rn_blocks = []
        for li, nb in enumerate(self.cfg.resnet_layers, start=1):
            for b in range(nb):
                p = "layer%d.%d." % (li, b)
                blk = {"stride": 2 if (b == 0 and li > 1) else 1}
                for cn, bnn in (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3")):
                    w, sh = fold(p + cn, p + bnn)
                    blk[cn] = (w.permute(0, 2, 3, 1).contiguous().to(dev, BF), sh.to(dev, BF))
                if b == 0:
                    w, sh = fold(p + "downsample.0", p + "downsample.1")
                    blk["down"] = (w.permute(0, 2, 3, 1).contiguous().to(dev, BF), sh.to(dev, BF))
                self.rn_blocks.append(blk)

    def _pack_misc(self):
        cfg, dev = self.cfg, self.device
        C = cfg.embed_dim
        npad = _pad8(cfg.num_seg_tokens)
        self.npad = npad
        self.wseg_pad = torch.zeros(npad, C, dtype=BF, device=dev)
        self._wseg_key = None
        self.refresh_frozen()

    def refresh_frozen(self):
        """Re-derive packed copies of frozen tensors (call after they are modified in place,
        e.g. seg_criterion._lazy_initialization writes seg_embed_tokens)."""
        w = self.model.decoder.seg_projection.weight.data
        key = (w.data_ptr(), w._version, tuple(w.shape))
        if getattr(self, "_wseg_key", None) == key:
            return                      # unchanged since the last forward (the weight is frozen in the recipe)
        self._wseg_key = key
        self.wseg_pad.zero_()
        self.wseg_pad[: w.shape[0]] = w.to(self.wseg_pad.device, BF)

    # --------------------------------------------------------------- workspace
    def buf(self, name, shape, dtype=BF):
        t = self.ws.get(name)
        if t is not None and t.shape == shape and t.dtype == dtype:
            return t
        shape = tuple(int(s) for s in shape)
        if t is None or tuple(t.shape) != shape or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            if _POISON:          # (debug) a read of a workspace element nobody wrote shows up as NaN / a huge index
                t.view(torch.uint8).fill_(0xFF) if t.numel() else None
            self.ws[name] = t
        return t

    def mark(self, name):
        if self.marks is not None and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev, time.perf_counter()))

    def gbuf(self, name, shape, dtype=BF):
        """backward buffer that weight-gradient work on the side stream may still be reading while the main
        stream has moved on: one instance per block so it is never overwritten within a step"""
        return self.buf(name + "@" + self._bt if self.overlap else name, shape, dtype)

    # ---- second stream for weight-gradient work ------------------------------------------------
    def _ev(self):
        e = self._evs[self._evi]
        self._evi = (self._evi + 1) % len(self._evs)
        return e

    def _new_stream(self, which, priority=0):
        """a side stream.  (Round 5 measured CU-masked streams for the weight-gradient / dQ / trunk work: a masked stream is one
        more hardware queue, 25-31 ms per step -- profiles/round5_cumask_ab.txt; removed.)"""
        return torch.cuda.Stream(device=self.device, priority=priority)

    def _wgrad_init(self):
        if self._side is None:
            self._side = self._new_stream("SIDE")
            self._evs = [torch.cuda.Event() for _ in range(128)]
            self._evi = 0

    @contextlib.contextmanager
    def _wgrad(self):
        """Everything enqueued so far on the current stream happens-before the body; the body runs on the
        side stream (in order with earlier bodies).  `_join_side` makes the main stream wait for it."""
        if not self.overlap:
            yield
            return
        self._wgrad_init()
        e = self._ev()
        e.record(torch.cuda.current_stream())
        self._side.wait_event(e)
        with torch.cuda.stream(self._side):
            prev = hip.set_stream(self._side.cuda_stream)
            try:
                yield
            finally:
                hip.set_stream(prev)

    def _side_do(self, fn):
        """Queue parameter-gradient work for the side stream.  The queue is flushed once per backward block
        (`_side_flush`): one event on the main stream per block instead of one per call site.  Everything a queued
        body reads lives in per-block buffers (`gbuf`), so running it later is safe."""
        if not self.overlap or self._in_flush:
            fn()
        else:
            self._pending.append(fn)

    def _side_flush(self):
        if not self._pending:
            return
        todo, self._pending = self._pending, []
        with self._wgrad():
            self._in_flush = True
            try:
                for fn in todo:
                    fn()
            finally:
                self._in_flush = False

    def _dq_stream_get(self):
        if getattr(self, "_dqs", None) is None:
            self._wgrad_init()
            self._dqs = self._new_stream("DQ")
        return self._dqs

    @contextlib.contextmanager
    def _fork(self, stream):
        """body runs on `stream`, after everything enqueued so far on the current stream"""
        e = self._ev()
        e.record(torch.cuda.current_stream())
        stream.wait_event(e)
        with torch.cuda.stream(stream):
            prev = hip.set_stream(stream.cuda_stream)
            try:
                yield
            finally:
                hip.set_stream(prev)

    def _join_side(self):
        self._side_flush()
        if self.overlap and self._side is not None:
            e = self._ev()
            e.record(self._side)
            torch.cuda.current_stream().wait_event(e)

    def _dense_bias(self, tag, H, T, S, pq, pk, rel, causal, P):
        """dense batch-invariant bias operands of one attention (hip.DenseBias), built on the CURRENT stream"""
        key = "dense_" + tag
        d = self.ws.get(key)
        if d is None or (d.H, d.T, d.S) != (H, T, S):
            d = hip.DenseBias(H, T, S, self.device)
            self.ws[key] = d
        hip.attn_dense_bias(d, pq, pk, rel=rel, causal=causal, P=P)
        return d

    def _dense_from(self, tag, H, T, S, bias):
        """a dense bias operand from a ready fp32 [H, T, S] tensor (training on a resized grid: models/segofa/resized.py)"""
        key = "dense_" + tag
        d = self.ws.get(key)
        if d is None or (d.H, d.T, d.S) != (H, T, S):
            d = hip.DenseBias(H, T, S, self.device)
            d.D.fill_(float("-inf"))              # rows / columns of the padding stay masked
            self.ws[key] = d
        d.D[:, :T, :S].copy_(bias)
        return d

    def _dense_built(self, ctx, tag):
        """(side stream) one event per dense bias: the forward attention of that layer waits for ITS operand only"""
        if not self.overlap:
            return
        evs = self.__dict__.setdefault("_dense_evs", {})
        ev = evs.get(tag)
        if ev is None:
            ev = evs[tag] = torch.cuda.Event()
        ev.record(self._side)
        ctx.setdefault("dense_ev", {})[tag] = ev

    def _dense_wait(self, tag):
        ev = self.ctx_building.get("dense_ev", {}).pop(tag, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _geometry(self, h, w, L):
        """index tables for a (h, w) feature grid and L text tokens (device tensors, cached)."""
        key = (h, w, L)
        g = self.geo.get(key)
        if g is not None:
            return g
        cfg, dev = self.cfg, self.device
        P = h * w
        g = {"P": P}
        ys = torch.arange(h).repeat_interleave(w)
        xs = torch.arange(w).repeat(h)
        g["gcode"] = (ys * (2 * w - 1) + xs).int().to(dev)
        g["code_bias"] = (h - 1) * (2 * w - 1) + (w - 1)
        g["n2d"] = (2 * h - 1) * (2 * w - 1)
        dy = torch.arange(-(h - 1), h).repeat_interleave(2 * w - 1)
        dx = torch.arange(-(w - 1), w).repeat(2 * h - 1)
        b = cfg.image_bucket_size
        g["enc_idx2d"] = ((dy + b - 1) * (2 * b - 1) + (dx + b - 1)).int().to(dev)      # encoder_module.py:87-104
        tb = token_bucket_of_delta(cfg.token_bucket_size, cfg.max_source_positions)
        dl = torch.arange(-(L - 1), L)
        g["enc_idx1d"] = tb[dl + cfg.max_source_positions - 1].int().to(dev)
        g["enc_idxx"] = torch.tensor([-1, -1], dtype=torch.int32, device=dev)
        sb = cfg.seg_bucket_size
        nseg_rel = (2 * sb - 1) ** 2 + 3
        g["dec_idx2d"] = ((dy + sb - 1) * (2 * sb - 1) + (dx + sb - 1)).int().to(dev)
        g["dec_idx1d"] = torch.tensor([nseg_rel - 1], dtype=torch.int32, device=dev)        # [0,0] corner
        # relx[0]: grid query x bos key = column 0 -> N-2 ; relx[1]: bos query x grid key = row 0 -> N-3
        g["dec_idxx"] = torch.tensor([nseg_rel - 2, nseg_rel - 3], dtype=torch.int32, device=dev)
        self.geo[key] = g
        return g

    # ------------------------------------------------------------------ ResNet
    def prefetch_trunk(self, patch_images, append=False):
        """Start the frozen ResNet-101 trunk of FUTURE batches on its own stream.

        The trunk has no trainable parameter (resnet.py + frozen_bn.py, `freeze_resnet`), so its output for
        batch n+1 does not depend on the update of step n: its ~90 small convolutions (each too small to fill
        256 CUs) run underneath step n instead of in front of step n+1.  `forward` picks the features up when
        it is handed the same tensor; anything else falls back to running the trunk in line.

        A LIST of image tensors (the next calls' batches, in order; same shape) goes through the trunk in ONE pass: the
        convolutions sit at a fixed ~15 us floor at B = 8 (tools/conv_bench.py: 1.97 ms per trunk at B = 8, 2.87 ms at 16,
        4.53 ms at 32), so two batches per pass cost 27 % less trunk time per image.  Every image's result is bit-identical
        to the single-batch pass (one output pixel = one k-loop in the same order)."""
        many = isinstance(patch_images, (list, tuple))
        batches = list(patch_images) if many else [patch_images]
        if not batches or not self.packed or any(self.device != t.device for t in batches):
            return
        if any(t.shape != batches[0].shape or t.dtype != batches[0].dtype for t in batches):
            batches = batches[:1]
        if self._trunk_stream is None:
            # high priority: the ~100 small convolutions must finish within the step they run under -- at normal priority
            # they were starved by the main / weight-gradient queues and the NEXT forward waited 2.6 ms for its features
            self._trunk_stream = self._new_stream("TRUNK", -1)
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)                                   # the images were produced on the caller's stream
        self._trunk_stream.wait_event(ready)
        slot = self._pf_slot = self._pf_slot ^ 1            # two feature buffers: the running step keeps its own
        n, B = len(batches), batches[0].shape[0]
        with torch.cuda.stream(self._trunk_stream):
            prev = hip.set_stream(self._trunk_stream.cuda_stream)
            try:
                if n == 1:
                    feat, h, w = self._resnet(batches[0], "@pf%d" % slot)
                else:
                    allimg = self.buf("rn_in@pf%dx%d" % (slot, n), (n * B,) + tuple(batches[0].shape[1:]), batches[0].dtype)
                    for i, t in enumerate(batches):
                        allimg[i * B:(i + 1) * B].copy_(t)
                    feat, h, w = self._resnet(allimg, "@pf%dx%d" % (slot, n))
            finally:
                hip.set_stream(prev)
            done = torch.cuda.Event()
            done.record(self._trunk_stream)
        ents = [{"key": self._tkey(t), "images": t, "feat": feat[i * B:(i + 1) * B], "h": h, "w": w, "done": done}
                for i, t in enumerate(batches)]
        if append and self._pf is not None:                  # behind the features that are still waiting to be used
            self._pf_more = self._pf_more + ents
        else:
            self._pf, self._pf_more = ents[0], ents[1:]

    @staticmethod
    def _tkey(t):
        return (t.data_ptr(), t._version, tuple(t.shape), t.dtype)

    def _trunk(self, patch_images):
        pf, self._pf = self._pf, None
        more, self._pf_more = self._pf_more, []
        if pf is not None and pf["key"] == self._tkey(patch_images):
            if more:                                        # the same pass also covered the following batches
                self._pf, self._pf_more = more[0], more[1:]
            # (a captured step joins the trunk stream at its own end -- Trainer._step_body -- so the features of the
            # previous replay are complete in stream order; an event of another capture must not be waited on)
            if pf.get("done") is not None and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream().wait_event(pf["done"])
            return pf["feat"], pf["h"], pf["w"]
        return self._resnet(patch_images)

    def _resnet(self, images, tag=""):
        B, _, Hh, Ww = images.shape
        _buf = self.buf
        buf = lambda name, shape: _buf(name + tag, shape)
        x4 = buf("rn_x4", (B, Hh, Ww, 4))
        hip.nchw_to_nhwc(images.contiguous(), x4, 4)
        H1, W1 = (Hh + 6 - 7) // 2 + 1, (Ww + 6 - 7) // 2 + 1
        s = buf("rn_stem", (B, H1, W1, 64))
        hip.stem_conv(x4, self.stem_w, self.stem_shift, s, B, Hh, Ww)
        H2, W2 = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
        maxel = B * H2 * W2 * 256
        pool = [buf("rn_p%d" % i, (maxel,)) for i in range(4)]
        cur = pool[0][: B * H2 * W2 * 64].view(B, H2, W2, 64)
        hip.maxpool(s, cur, B, H1, W1, 64)
        curi, Hc, Wc, Cc = 0, H2, W2, 64

        def take(busy):
            for i in range(4):
                if i not in busy:
                    return i
            raise RuntimeError("resnet buffer pool exhausted")

        for blk in self.rn_blocks:
            st = blk["stride"]
            w1, s1 = blk["conv1"]; w2, s2 = blk["conv2"]; w3, s3 = blk["conv3"]
            mid, cout = w1.shape[0], w3.shape[0]
            Ho, Wo = (Hc + 2 - 3) // st + 1, (Wc + 2 - 3) // st + 1
            i1 = take({curi})
            o1 = pool[i1][: B * Hc * Wc * mid].view(B, Hc, Wc, mid)
            hip.conv2d_nhwc(cur, w1, s1, None, o1, B, Hc, Wc, Cc, mid, 1, 1, 1, 0, True)
            i2 = take({curi, i1})
            o2 = pool[i2][: B * Ho * Wo * mid].view(B, Ho, Wo, mid)
            hip.conv2d_nhwc(o1, w2, s2, None, o2, B, Hc, Wc, mid, mid, 3, 3, st, 1, True)
            if "down" in blk:
                wd, sd_ = blk["down"]
                i3 = take({curi, i1, i2})
                idt = pool[i3][: B * Ho * Wo * cout].view(B, Ho, Wo, cout)
                hip.conv2d_nhwc(cur, wd, sd_, None, idt, B, Hc, Wc, Cc, cout, 1, 1, st, 0, False)
                io = i1
            else:
                i3, idt = curi, cur
                io = i1
            out = pool[io][: B * Ho * Wo * cout].view(B, Ho, Wo, cout)
            hip.conv2d_nhwc(o2, w3, s3, idt, out, B, Ho, Wo, mid, cout, 1, 1, 1, 0, True)
            cur, curi, Hc, Wc, Cc = out, io, Ho, Wo, cout
        feat = buf("rn_feat", (B, Hc * Wc, Cc))
        feat.copy_(cur.view(B, Hc * Wc, Cc))
        return feat, Hc, Wc

    # ------------------------------------------------------------ small helpers
    def _rel_tables(self, tag, tables_idx):
        """gather (table, idx) pairs into fp32 [H, n] delta tables"""
        out = []
        H = self.cfg.heads
        for k, (tab, idx) in enumerate(tables_idx):
            o = self.buf("%s_rel%d" % (tag, k), (H, idx.numel()), torch.float32)
            if tab is None:
                o.zero_()
            else:
                hip.rel_gather(tab, idx, o)
            out.append(o)
        return out

    def _rel_tables_all(self, tag, names, idxs):
        """the delta tables of every layer in one launch per index set: names[l] = parameter of layer l (or None for
        an all-zero table), idxs = index sets -> list over index sets of fp32 [L, H, n] (layer l = [l])"""
        H, L = self.cfg.heads, len(names)
        out = []
        for k, (use, idx) in enumerate(idxs):
            name = "%s_relall%d" % (tag, k)
            fresh = name not in self.ws
            o = self.buf(name, (L, H, idx.numel()), torch.float32)
            if not use:
                if fresh:
                    o.zero_()             # constant: zeroed once per allocation
            else:
                hip.rel_gather_multi([self.W(n) for n in names], idx, o)
            out.append(o)
        return out

    # ---- dropout / DropPath sites (a12) -------------------------------------------------------
    def _drop_setup(self, B, train):
        cfg = self.cfg
        self.drop_on = bool(train and (cfg.dropout > 0 or cfg.encoder_drop_path_rate > 0 or cfg.decoder_drop_path_rate > 0))
        if not self.drop_on:
            return
        # per-sample DropPath keep masks for every residual branch, one draw per forward
        # (rates: torch.linspace(0, rate, n_layers), encoder_module.py:232 / decoder_module.py:219)
        key = (cfg.encoder_drop_path_rate, cfg.decoder_drop_path_rate, cfg.enc_layers, cfg.dec_layers, self.device)
        if getattr(self, "_dp_key", None) != key:
            rates = []
            for l in range(cfg.enc_layers):
                r = cfg.encoder_drop_path_rate * l / max(1, cfg.enc_layers - 1)
                rates += [r, r]
            for l in range(cfg.dec_layers):
                r = cfg.decoder_drop_path_rate * l / max(1, cfg.dec_layers - 1)
                rates += [r, r, r]
            self._dp_keep = (1.0 - torch.tensor(rates, dtype=torch.float32, device=self.device)).contiguous()
            self._dp_key = key
        keep = self._dp_keep
        self.dp_scale = self.buf("dp_scale", (keep.shape[0], B), torch.float32)
        hip.droppath_scale(self.dp_scale, keep, self._site_seed(7))
        self._dp_rows = {}

    def _dp(self, kind, layer, k):
        i = 2 * layer + k if kind == "e" else 2 * self.cfg.enc_layers + 3 * layer + k
        v = self._dp_rows.get(i)
        if v is None:
            v = self._dp_rows[i] = self.dp_scale[i]
        return v

    def _attn_drop(self, tag, p):
        """(p, seed) of the attention dropout of block `tag` (e<l> | d<l> | d<l>c), or None"""
        if not p:
            return None
        kind = 3000 if tag.endswith("c") else (1000 if tag[0] == "e" else 2000)
        return (p, self._site_seed(kind + int(tag[1:].rstrip("c"))))

    def _site_seed(self, site):
        # full seed = (step_seed * 1000003 + site) * 0x100000001B3 + 0x9E3779B97F4A7C15 (mod 2^64); the step part is
        # `step_dev[0]` on the device (upload_step_seed)
        return site * 0x100000001B3 + 0x9E3779B97F4A7C15

    def upload_step_seed(self):
        """step_dev[0] <- step_seed * 1000003 * 0x100000001B3 (mod 2^64), by an async copy from a ring of pinned words"""
        if self._step_dev_val == self.step_seed and self.step_dev is not None:
            return
        if self.step_dev is None or self.step_dev.device != self.device:
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
            self._step_pin = torch.zeros(64, dtype=torch.int64).pin_memory()
        v = (self.step_seed * 1000003 * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        i = self._step_pin_i = (self._step_pin_i + 1) % 64
        self._step_pin[i] = v - (1 << 64) if v >= (1 << 63) else v
        self.step_dev.copy_(self._step_pin[i:i + 1], non_blocking=True)
        self._step_dev_val = self.step_seed

    def _drop(self, x, resid, out, site, dp=None, rows_per_batch=None):
        hip.dropout(x, resid, out, self.cfg.dropout, self._site_seed(site), dp, rows_per_batch)
        return out

    def _dropargs(self, site, dp=None, rows_per_batch=None):
        """operand of the dropout fused into ln_fwd / ln_bwd (same mask as `_drop` at that site), or None"""
        if not self.drop_on:
            return None
        return (self.cfg.dropout, self._site_seed(site), dp, rows_per_batch)

    def _ln_stats(self, tag, rows):
        return self.buf(tag + "_mu", (rows,), torch.float32), self.buf(tag + "_rs", (rows,), torch.float32)

    def _gain32(self, tag, name):
        return self.Wf(name)       # fp32 [H] from the master copy

    def weights_changed(self):
        """Tell the engine that the bf16 parameters were edited behind its back in a way no version counter sees (an EMA swap or
        a checkpoint restore through `p.data.copy_`): the next forward re-derives the fp32 copy of the LayerNorm / c_attn
        operands and the cached resized rel-pos biases are retired.  (Optimizer steps after a backward, in-place edits of the
        parameters themselves and load_state_dict are detected without this call.)"""
        self._master_stale = True
        self._wver += 1

    # ----------------------------------------------------------------- a still-running optimizer (Trainer, defer_optimizer)
    def params_pending(self, events):
        """`events`: {slice key: event} recorded on the optimizer's stream as the slices of the parameter arena become final, in
        the order the forward first reads them ("g0": everything outside the layers and the token table, "emb": the token
        table, "e<l>": encoder layer l, "all": the last launch).  Whatever reads parameters waits for what it needs
        (`_params_wait`); the optimizer of step n then runs underneath the start of the forward of step n + 1."""
        self._popt = dict(events) if events else None

    def _params_wait(self, keys=None, stream=None):
        po = self._popt
        if not po:
            return
        st = stream if stream is not None else torch.cuda.current_stream()
        if keys is None:
            st.wait_event(po["all"])
            if stream is None:
                self._popt = None          # the main stream is behind the whole update: nothing left to wait for
            return
        for k in keys:
            ev = po.get(k)
            if ev is not None:
                st.wait_event(ev)

    # ----------------------------------------------------------------- forward
    def forward(self, *args, **kw):
        prev = hip.set_stream(torch.cuda.current_stream().cuda_stream)    # one stream lookup per pass, not per launch
        need_grad = kw.get("need_grad", True)
        if self._popt and not need_grad:
            self._params_wait(None)        # (only the training forward waits slice by slice)
        aux = (not need_grad) and self._gctx is not None
        self.ws, self.saved = (self._ws_aux, self._saved_aux) if aux else (self._ws_grad, self._saved_grad)
        prev_sa = None
        try:
            if not torch.cuda.is_current_stream_capturing():
                self.upload_step_seed()
            prev_sa = hip.set_seed_add(self.step_dev)
            if self.packed and not self.master_owned and not torch.cuda.is_current_stream_capturing():
                # an external owner of the bf16 parameters (fairseq's optimizer, an EMA swap, a manual p.data.copy_) may
                # have edited them since the last forward: the fp32 copy the LayerNorm / c_attn operands are read from
                # follows (one pass over the arena; entries whose bf16 rounding still matches keep their fp32 value) -- on
                # every TRAINING forward, and at evaluation only when a parameter's version counter has moved (ADVICE r3:
                # a validation pass does not pay a full-arena sweep per batch).  A detected edit also retires the cached
                # resized rel-pos biases, which were built from the old tables.
                ver = sum(p._version for p in self.trainable_params())
                changed = ver != getattr(self, "_pver_seen", None)
                # (`.data` edits bump no version counter -- an EMA swap by p.data.copy_ ahead of a validation pass.
                # The first evaluation forward after a training forward always re-syncs; edits BETWEEN two evaluation
                # forwards need `weights_changed()`, which the model's load_state_dict / swap hooks call.)
                entering_eval = (not need_grad) and getattr(self, "_last_fwd_train", True)
                self._last_fwd_train = bool(need_grad)
                if entering_eval:
                    self._wver += 1
                if need_grad or self._master_stale or changed or entering_eval:
                    hip.sync_master(self.master, self.p16[: self.n_train])
                    self._master_stale = False
                if changed:
                    self._pver_seen = ver
                    self._wver += 1
            out = self._forward(*args, **kw)
            if need_grad:
                self._gctx = self.ctx
                self.ctx["drop_state"] = (self.drop_on, getattr(self, "dp_scale", None), getattr(self, "_dp_rows", None))
                self.ctx["attn_drop_p"] = self.attn_drop_p
            return out
        finally:
            hip.set_seed_add(prev_sa)
            hip.set_stream(prev)

    def backward(self, dlogits):
        if self._gctx is None:
            raise RuntimeError("ifseg_amd HIP engine: backward without a pending training forward")
        prev = hip.set_stream(torch.cuda.current_stream().cuda_stream)
        self.ws, self.saved, self.ctx = self._ws_grad, self._saved_grad, self._gctx
        self.drop_on, self.dp_scale, self._dp_rows = self.ctx["drop_state"]
        prev_sa = hip.set_seed_add(self.step_dev)
        try:
            return self._backward(dlogits)
        finally:
            hip.set_seed_add(prev_sa)
            self._gctx = None
            self._master_stale = not self.master_owned
            self._wver += 1                 # an optimizer step follows: cached resized biases are stale
            hip.set_stream(prev)

    def deferred_check(self, tensor, bad, message, exc=NotImplementedError, native=None):
        """Input validation without a device sync: `bad(tensor)` (a 0-d bool tensor) is evaluated on the stream and read
        back once its event has completed -- at the latest by the next call, i.e. the error surfaces one step late instead
        of draining the queue on every step (each batch is a new tensor, so a cache keyed on the tensor never hits with a
        real data iterator).  The first call of a process checks synchronously.
        `native` = (hip.CHECK_* mode, value, row_len): the same predicate as ONE launch that writes its verdict straight into
        the pinned word (csrc/rowops.hip check_inputs_kernel) instead of compare + reduce + cast + copy torch kernels on the
        main queue -- three checks per step, ~12 tiny kernels between the optimizer and the first kernel of the forward."""
        if torch.cuda.is_current_stream_capturing():
            return                                 # a captured forward replays validated inputs (checked at warm-up)
        if lab.get("SYNC_CHECKS"):
            if bool(bad(tensor)):
                raise exc(message)
            return
        # flags travel to PINNED host memory by an async copy in stream order; the host reads the pinned word once the
        # event behind the copy has completed.  (Reading the device flag with bool()/.item() is a blocking copy queued
        # behind everything already enqueued: it drained the whole step and cost the host 3 x 3 ms per step.)
        pend = self._pending_checks
        keep, err = [], None
        for slot, ev, msg in pend:
            if ev.query():
                self._check_free.append(slot)
                if int(self._check_pin[slot]) and err is None:
                    err = msg
            else:
                keep.append((slot, ev, msg))
        if err is not None:
            self._pending_checks = keep
            raise err[1](err[0])
        if getattr(self, "_check_pin", None) is None:
            self._check_pin = torch.zeros(64, dtype=torch.uint8).pin_memory()
            self._check_free = list(range(64))
        if not self._check_free:               # never let the list grow: settle the oldest
            slot, ev, msg = keep.pop(0)
            ev.synchronize()
            self._check_free.append(slot)
            if int(self._check_pin[slot]):
                self._pending_checks = keep
                raise msg[1](msg[0])
        use_native = native is not None and tensor.is_cuda and tensor.is_contiguous()
        first = message not in self._checked_kinds
        if use_native:
            slot = self._check_free.pop()
            hip.check_inputs(tensor, native[0], self._check_pin, slot, native[1], native[2])
            ev = torch.cuda.Event()
            ev.record()
            if first:                           # the first call of every kind of check is synchronous
                self._checked_kinds.add(message)
                ev.synchronize()
                self._check_free.append(slot)
                if int(self._check_pin[slot]):
                    self._pending_checks = keep
                    raise exc(message)
            else:
                keep.append((slot, ev, (message, exc)))
            self._pending_checks = keep
            return
        flag = bad(tensor)
        if first:
            self._checked_kinds.add(message)
            if bool(flag):
                raise exc(message)
        else:
            slot = self._check_free.pop()
            self._check_pin[slot:slot + 1].copy_(flag.reshape(1).to(torch.uint8), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            keep.append((slot, ev, (message, exc)))
        self._pending_checks = keep

    def _forward(self, src_tokens, patch_images, prev_output_tokens=None, full_context_alignment=False,
                 need_grad=True, bag=None):
        """bag = (ids int64 [B, maxlen], ends int64 [B*P]): the image-free entry (encode_with_artificial_image,
        encoder_module.py:499-675) -- patch embeddings are EmbeddingBag means of class-name tokens, the ResNet
        trunk and image_proj are not run; `patch_images` is ignored."""
        cfg = self.cfg
        self.ctx_building = None
        self.attn_drop_p = float(getattr(cfg, "attention_dropout", 0.0)) if need_grad else 0.0
        self.act_drop_p = float(getattr(cfg, "activation_dropout", 0.0)) if need_grad else 0.0
        dev = src_tokens.device if bag is not None else patch_images.device
        if not self.packed or self.device != dev:
            self.pack(dev)
        pads = bool(getattr(cfg, "padded_prompts", False))
        if pads:
            # (the image-free entry shares the encoder from the embeddings on: the same key counts -- fixture_imfree_padded.npz)
            # the padding of a sample must be a suffix of its prompt (collate pads on the right): a valid key COUNT per sample
            self.deferred_check(src_tokens, lambda t: (t[:, :-1].eq(1) & t[:, 1:].ne(1)).any() | t[:, 0].eq(1).any(),
                                "ifseg_amd HIP engine: <pad> tokens must form a suffix of every prompt (right padding)",
                                exc=ValueError, native=(hip.CHECK_SUFFIX, 1, src_tokens.shape[1]))
        else:
            self.deferred_check(src_tokens, lambda t: t.eq(1).any(),
                                "ifseg_amd HIP engine: padded source tokens need cfg.padded_prompts = True: every IFSeg sample carries the same unpadded prompt, so by "
                                "default a step carries no per-sample key counts", native=(hip.CHECK_ANY_EQ, 1, 1))
        B, L = src_tokens.shape
        C, Fd, H = cfg.embed_dim, cfg.ffn_dim, cfg.heads
        scaling = float(cfg.head_dim * cfg.attn_scale_factor) ** -0.5
        W, Wf, buf = self.W, self.Wf, self.buf
        if bag is not None:
            feat, h, w = None, cfg.patch_image_size // 16, cfg.patch_image_size // 16     # encoder_module.py:540
            if bag[1].numel() != B * h * w:
                raise RuntimeError("aux_input: %d bags for a batch of %d x %dx%d patches" % (bag[1].numel(), B, h, w))
        else:
            feat, h, w = self._trunk(patch_images)
            self.mark("trunk_ready")
            # the next batch's trunk starts once this batch's features are taken -- or (IFSEG_TRUNK_AT=e<k> / end) later,
            # inside this step's backward, see `_trunk_launch_point`
            if self._pf_request is not None and (self.trunk_at in ("fwd", "fwd1") or not need_grad):
                req = self._pf_request
                keep = self._prefetch_request(req)
                if not need_grad or not keep:
                    self._pf_request = None
                # (training, next batch already cached: the request stays for the end of the backward -- `_backward`)
        P = h * w
        oh = cfg.orig_patch_image_size // 16
        slow = (h, w) != (oh, oh) or (h, w) != (cfg.seg_bucket_size,) * 2 or P % 64 != 0
        if slow:
            # images whose feature grid differs from the trained one (seg_criterion.py:194-217 at evaluation: batch 1, native
            # aspect ratio; `--patch-image-size 640` on a 512-grid checkpoint in training): position tables / rel-pos biases are
            # bilinear-resized exactly like the reference.  Evaluation: the dense fp32 bias of csrc/resize.hip, cached per shape.
            # Training (round 6): the standard step below with dense biases from models/segofa/resized.py and their adjoints.
            # Padded prompts (round 6): the key counts go to the batch-inner kernels of the standard step, so an EVALUATION of a
            # padded batch takes the standard step too (dense biases from resized.py) instead of the cached-bias slow path.
            if bag is not None:
                raise NotImplementedError("ifseg_amd HIP engine: the image-free entry on a resized feature grid is not supported")
            if not need_grad and not pads:
                return self._forward_resized(src_tokens, feat, h, w, prev_output_tokens, full_context_alignment)
        resized = bool(slow)
        g = self._geometry(h, w, L)
        T, Td = P + L, P + 1
        self._drop_setup(B, need_grad)
        self._train_fwd = bool(need_grad)
        ctx = {"B": B, "L": L, "P": P, "T": T, "Td": Td, "h": h, "w": w, "full": bool(full_context_alignment),
               "src_tokens": src_tokens, "feat": feat}
        if resized:
            ctx["resized"] = True
        if pads:
            # encoder_padding_mask as valid key counts: the P patch tokens + the prompt tokens in front of the padding
            nonpad = src_tokens.ne(1)
            ctx["nonpad"] = nonpad
            ctx["klen"] = (nonpad.sum(1) + P).to(torch.int32).contiguous()
        self.ctx_building = ctx
        e = "encoder."
        if need_grad and getattr(self, "train_tok", False):
            if bag is not None:
                # (the reference steps encoder.embed_tokens_bag.weight -- EmbeddingBag.from_pretrained: a SECOND Parameter over the
                # token table's storage, encoder_module.py:147 -- with an optimizer state of its own; not reproduced)
                raise NotImplementedError("ifseg_amd HIP engine: the image-free entry with a trainable token table "
                                          "(--freeze-encoder-embedding false) is not supported")
            ctx["src_ids"] = src_tokens.reshape(-1).contiguous()
        self._params_wait(["g0"])
        # ---- embeddings (forward_embedding, encoder_module.py:388-446)
        img_pre = buf("img_pre", (B * P, C))
        if bag is not None:
            self._params_wait(["emb"])
            hip.embed_bag_mean(W(e + "embed_tokens.weight"), bag[0].contiguous(), bag[1].contiguous(),
                               W(e + "type_embedding.weight")[1], img_pre)
        else:
            bias_img = buf("bias_img", (C,))
            hip.add_bf16(W(e + "image_proj.bias"), W(e + "type_embedding.weight")[1], bias_img)
            hip.linear_fwd(feat.view(B * P, 1024), W(e + "image_proj.weight"), bias_img, out=img_pre)
        x = buf("e_x_in", (B, T, C))
        mu, rs = self._ln_stats("img_ln", B * P)
        hip.ln_fwd(img_pre.view(B, P, C), Wf(e + "patch_layernorm_embedding.weight"),
                   Wf(e + "patch_layernorm_embedding.bias"), x[:, :P], mu, rs, drop=self._dropargs(1))
        tok_pre = buf("tok_pre", (B * L, C))
        self._params_wait(["emb"])
        hip.embed_rows(W(e + "embed_tokens.weight"), src_tokens.reshape(-1).contiguous(),
                       W(e + "type_embedding.weight")[0], tok_pre)
        mu, rs = self._ln_stats("tok_ln", B * L)
        hip.ln_fwd(tok_pre.view(B, L, C), Wf(e + "layernorm_embedding.weight"), Wf(e + "layernorm_embedding.bias"),
                   x[:, P:], mu, rs, drop=self._dropargs(2))
        if pads:
            x[:, P:].mul_(ctx["nonpad"].unsqueeze(-1).to(x.dtype))      # x = x * (1 - encoder_padding_mask), encoder_module.py:738-742
        # ---- abs-pos operands (encoder_module.py:757-771): LN over the table rows in place
        bsz = cfg.image_bucket_size
        pos_all = buf("e_pos_all", (T, C))
        mu, rs = self._ln_stats("ipos_ln", P)
        if resized and P > oh * oh:
            # more patches than the trained grid: the trained grid's rows resized (get_patch_images_info, encoder_module.py:358-370)
            ipos = buf("e_ipos_resized", (P, C))
            hip.resize_rows_bilinear(W(e + "embed_image_positions.weight"), ipos, h, w, oh, oh, bsz, 1)
            hip.ln_fwd(ipos, Wf(e + "image_pos_ln.weight"), Wf(e + "image_pos_ln.bias"), pos_all[:P], mu, rs)
        else:
            img_pos_view = W(e + "embed_image_positions.weight")[1:1 + bsz * h].view(h, bsz, C)[:, :w]
            hip.ln_fwd(img_pos_view, Wf(e + "image_pos_ln.weight"), Wf(e + "image_pos_ln.bias"), pos_all[:P].view(h, w, C), mu, rs)
        mu, rs = self._ln_stats("tpos_ln", L)
        hip.ln_fwd(W(e + "embed_positions.weight")[:L], Wf(e + "pos_ln.weight"), Wf(e + "pos_ln.bias"), pos_all[P:], mu, rs)
        pqk = buf("e_pqk", (T, 2 * C))
        hip.linear_fwd(pos_all, self._fused(self.p16, e + "pos_q_linear.weight", 2 * C, C),
                       self._fused(self.p16, e + "pos_q_linear.bias", 2 * C), out=pqk, alpha=scaling, alpha_ncols=C)
        ctx["e_pq"], ctx["e_pk"] = pqk[:, :C], pqk[:, C:]
        # ---- encoder layers (rel-pos delta tables of all layers: two launches)
        (e_r2,) = self._rel_tables_all("e_img", ["%simage_rel_pos_table_list.%d.weight" % (e, l) for l in range(cfg.enc_layers)],
                                       [(True, g["enc_idx2d"])])
        (e_r1,) = self._rel_tables_all("e_tok", ["%stoken_rel_pos_table_list.%d.weight" % (e, l) for l in range(cfg.enc_layers)],
                                       [(True, g["enc_idx1d"])])
        (e_rx,) = self._rel_tables_all("e_x", [None] * cfg.enc_layers, [(False, g["enc_idxx"])])
        self.mark("enc_layers_start")
        x_pre = None
        bi = resized or ((need_grad or pads) and w <= 64 and w % 8 == 0 and self.attn_bi in ("1", "auto"))
        # which attentions: e(ncoder self), d(ecoder self), c(ross); IFSEG_ATTN_BI_WHICH overrides
        bi_which = lab.get("ATTN_BI_WHICH", "e+d+c").split("+") if bi else []
        if pads and not (bi and "e" in bi_which and "c" in bi_which and self.bi_fwd):
            raise NotImplementedError("ifseg_amd HIP engine: key padding is implemented by the batch-inner attention kernels "
                                      "(grids up to 64 wide, a multiple of 8)")
        ctx["dense"] = {}
        if resized:
            bi_which = ["e", "d", "c"]
            with self._wgrad():          # parameters only: torch ops on [H, T, T] tensors (models/segofa/resized.py), side stream
                bufs = dict(self.model.named_buffers())
                absb = R.abs_bias(ctx["e_pq"], ctx["e_pk"], H)
                for l in range(cfg.enc_layers):
                    relb = R.encoder_rel_bias(W("%stoken_rel_pos_table_list.%d.weight" % (e, l)).float(),
                                              W("%simage_rel_pos_table_list.%d.weight" % (e, l)).float(),
                                              bufs["encoder.token_rp_bucket"], bufs["encoder.image_rp_bucket"], (h, w), oh, bsz, L)
                    ctx["dense"]["e%d" % l] = self._dense_from(("e%d" % l), H, T, T, absb + relb)
                    self._dense_built(ctx, "e%d" % l)
        elif bi and "e" in bi_which:
            # parameters only: every layer's dense bias on the side stream, under the first blocks of the forward
            with self._wgrad():
                for l in range(cfg.enc_layers):
                    rel = hip.RelBias(P, g["gcode"], g["code_bias"], e_r2[l], e_r1[l], e_rx[l], grid_w=w)
                    ctx["dense"]["e%d" % l] = self._dense_bias("e%d" % l, H, T, T, ctx["e_pq"], ctx["e_pk"], rel, False, P)
                    self._dense_built(ctx, "e%d" % l)
        # (profiles/round6_early_decoder_dense_ab.txt: -0.11 ms per step; a pending deferred optimizer keeps the late place)
        if need_grad and bi and self.overlap and not self._popt:
            self._dec_pos_dense(ctx, g, pos_all, scaling, bi, bi_which, resized)
        if need_grad and self.overlap and not torch.cuda.is_current_stream_capturing() and "g16fwd" not in _EXP_SKIP:
            with self._wgrad():          # (behind the encoder's dense biases, which the forward waits for; the previous step's Adam read g16)
                self._params_wait(None, stream=self._side)      # (a deferred optimizer is still reading the gradients)
                self.g16.zero_()
                if getattr(self, "_g16_ev", None) is None:
                    self._g16_ev = torch.cuda.Event()
                self._g16_ev.record(self._side)
                ctx["g16_zeroed"] = self._g16_ev
        for l in range(cfg.enc_layers):
            p = "%slayers.%d." % (e, l)
            tg = "e%d" % l
            self._params_wait([tg])
            rel = None if resized else hip.RelBias(P, g["gcode"], g["code_bias"], e_r2[l], e_r1[l], e_rx[l], grid_w=w)
            x, xn = self._self_block_fwd(tg, p, "self_attn", "self_attn_layer_norm", "attn_ln", x, B, T,
                                         ctx["e_pq"], ctx["e_pk"], rel, False, scaling, site=("e", l, 0), xn_pre=x_pre,
                                         next_ln=(p + "final_layer_norm", tg + "_fln1", buf(tg + "_fxn", (B * T, C))))
            if l + 1 < cfg.enc_layers:      # the next layer's pre-LN, or the encoder's final LayerNorm
                nl = ("%slayers.%d.self_attn_layer_norm" % (e, l + 1), "e%d_ln1" % (l + 1), buf("e%d_xn" % (l + 1), (B * T, C)))
            else:
                nl = (e + "layer_norm", "e_final_ln", buf("enc_out", (B, T, C)).view(B * T, C))
            if l + 1 < cfg.enc_layers:
                # the FFN's tail launch also computes layer l+1's pre-LN: it reads that layer's self_attn_layer_norm gain / bias
                # (fp32 master) -- a deferred optimizer must be done with slice e<l+1> first
                self._params_wait(["e%d" % (l + 1)])
            x, x_pre = self._ffn_fwd(tg, p, x, B * T, site=("e", l, 1), rpb=T, xn_pre=xn, next_ln=nl)
        enc_out = buf("enc_out", (B, T, C))
        if x_pre is None:
            mu, rs = self._ln_stats("e_final_ln", B * T)
            hip.ln_fwd(x.view(B * T, C), Wf(e + "layer_norm.weight"), Wf(e + "layer_norm.bias"), enc_out.view(B * T, C), mu, rs)
        ctx["e_x_final"] = x
        ctx["enc_out"] = enc_out
        self.mark("enc_fwd_end")
        self._params_wait(None)           # everything from here on (decoder, all layers' fc2 for the FFN coefficients)

        # ---- decoder (extract_features_scriptable_surrogate, decoder_module.py:486-677)
        d = "decoder."
        # the K|V projections of every decoder layer's cross-attention only read the encoder output: they run on
        # the side stream underneath the first decoder blocks instead of inside each layer's dependent chain
        ctx["ckv_ready"] = None
        if need_grad and self.ffn_ln_fused:
            self._ffn_ln_coefs()        # one small launch for all layers, main stream (read again by the backward)
        if self.overlap and need_grad:
            with self._wgrad():
                ctx["ckv_ready"] = []
                for l in range(cfg.dec_layers):
                    a_ = "%slayers.%d.encoder_attn" % (d, l)
                    kv = buf("d%d_ckv" % l, (B, T, 2 * C))
                    hip.linear_fwd(enc_out.view(B * T, C), self._fused(self.p16, a_ + ".k_proj.weight", 2 * C, C),
                                   self._fused(self.p16, a_ + ".k_proj.bias", 2 * C), out=kv.view(B * T, 2 * C))
                    ev = self._ev()              # one event per layer: a cross attention waits for ITS projection only
                    ev.record(self._side)
                    ctx["ckv_ready"].append(ev)
        y0b = buf("d_bos", (B, 1, C))
        bos = (prev_output_tokens[:, :1] if prev_output_tokens is not None
               else torch.zeros(B, 1, dtype=torch.long, device=dev))
        hip.embed_rows(W(e + "embed_tokens.weight"), bos.reshape(-1).contiguous(), None, y0b)
        ctx["bos_ids"] = bos.reshape(-1).contiguous()
        y = buf("d_y_in", (B, Td, C))
        mu, rs = self._ln_stats("d_emb_ln_p", B * P)
        hip.ln_fwd(enc_out[:, :P], Wf(d + "layernorm_embedding.weight"), Wf(d + "layernorm_embedding.bias"), y[:, :P],
                   mu, rs, drop=self._dropargs(3))
        mu, rs = self._ln_stats("d_emb_ln_b", B)
        hip.ln_fwd(y0b, Wf(d + "layernorm_embedding.weight"), Wf(d + "layernorm_embedding.bias"), y[:, P:], mu, rs,
                   drop=self._dropargs(4))
        if "d_cpq" not in ctx:       # (not already done ahead of the encoder layers)
            self._dec_pos_dense(ctx, g, pos_all, scaling, bi, bi_which, resized)
        cpq, cpk, causal = ctx["d_cpq"], ctx["d_cpk"], ctx["d_causal"]
        d_r2, d_r1, d_rx = ctx["d_rel"]
        y_pre = None
        for l in range(cfg.dec_layers):
            p = "%slayers.%d." % (d, l)
            tg = "d%d" % l
            # (resized grid: the causal mask is inside the dense operand, the kernels walk every block)
            rel = None if resized else hip.RelBias(P, g["gcode"], g["code_bias"], d_r2[l], d_r1[l], d_rx[l], grid_w=w)
            y, yn = self._self_block_fwd(tg, p, "self_attn", "self_attn_layer_norm", "self_attn_ln", y, B, Td,
                                         ctx["d_spq"], ctx["d_spk"], rel, causal and not resized, scaling, site=("d", l, 0), xn_pre=y_pre,
                                         next_ln=(p + "encoder_attn_layer_norm", tg + "_cln1", buf(tg + "_cyn", (B * Td, C))))
            y, yn = self._cross_block_fwd(tg, p, y, enc_out, B, Td, T, cpq, cpk, scaling, site=("d", l, 1), yn_pre=yn,
                                          next_ln=(p + "final_layer_norm", tg + "_fln1", buf(tg + "_fxn", (B * Td, C))))
            nl = (("%slayers.%d.self_attn_layer_norm" % (d, l + 1), "d%d_ln1" % (l + 1), buf("d%d_xn" % (l + 1), (B * Td, C)))
                  if l + 1 < cfg.dec_layers else None)
            y, y_pre = self._ffn_fwd(tg, p, y, B * Td, site=("d", l, 2), rpb=Td, xn_pre=yn, next_ln=nl)
        ctx["d_y_final"] = y
        # final LN written in reference order [bos, patches] (decoder_module.py:668-675)
        featb = buf("d_feat", (B, Td, C))
        mu, rs = self._ln_stats("d_final_ln_p", B * P)
        hip.ln_fwd(y[:, :P], Wf(d + "layer_norm.weight"), Wf(d + "layer_norm.bias"), featb[:, 1:], mu, rs)
        mu, rs = self._ln_stats("d_final_ln_b", B)
        hip.ln_fwd(y[:, P:], Wf(d + "layer_norm.weight"), Wf(d + "layer_norm.bias"), featb[:, :1], mu, rs)
        logits = buf("logits_pad", (B, Td, self.npad))
        if self.train_seg:      # the tied projection follows the (trainable) seg embeddings: the padded copy is rebuilt per forward
            self.wseg_pad[: cfg.num_seg_tokens].copy_(W("encoder.seg_embed_tokens.weight"))
        hip.linear_fwd(featb.view(B * Td, C), self.wseg_pad, out=logits.view(B * Td, self.npad))   # :290-294
        self.ctx = ctx
        return logits, ctx

    def _dec_pos_dense(self, ctx, g, pos_all, scaling, bi, bi_which, resized):
        """The decoder's position operands (seg positions -> seg_pos_ln -> self / cross abs-pos projections, decoder_module.py:
        541-558), its rel-pos delta tables and every decoder layer's dense bias operand (side stream).  All of it depends on
        PARAMETERS only (and on the encoder's position rows): a training forward runs it ahead of the encoder layers, so the
        decoder's dense biases are built under the encoder instead of at the encoder -> decoder hand-over, where the main queue
        waited for them behind the six cross-attention K|V projections (round 6)."""
        cfg = self.cfg
        B, L, P, T, Td, h, w = (ctx[k] for k in ("B", "L", "P", "T", "Td", "h", "w"))
        C, H = cfg.embed_dim, cfg.heads
        W, Wf, buf = self.W, self.Wf, self.buf
        d = "decoder."
        dev = self.device
        # positions: internal order [grid cells 1..P | slot 0]
        sb = cfg.seg_bucket_size
        segtab = W(d + "embed_seg_positions.weight")
        tp = buf("d_tp", (Td, C))
        seg_rows, seg_bos = segtab[1:1 + P], segtab[:1]
        if resized:      # the sb x sb grid's rows resized to (h, w) (decoder_module.py:541-550); the bos slot as it is
            tgt = buf("d_spos_resized", (Td, C))
            hip.resize_rows_bilinear(segtab, tgt[:P], h, w, sb, sb, sb, 1)
            tgt[P:].copy_(segtab[:1])
            seg_rows, seg_bos = tgt[:P], tgt[P:]
        mu, rs = self._ln_stats("d_tp_ln_p", P)
        hip.ln_fwd(seg_rows, Wf(d + "seg_pos_ln.weight"), Wf(d + "seg_pos_ln.bias"), tp[:P], mu, rs)
        mu, rs = self._ln_stats("d_tp_ln_b", 1)
        hip.ln_fwd(seg_bos, Wf(d + "seg_pos_ln.weight"), Wf(d + "seg_pos_ln.bias"), tp[P:], mu, rs)
        spqk = buf("d_spqk", (Td, 2 * C))
        hip.linear_fwd(tp, self._fused(self.p16, d + "self_pos_q_linear.weight", 2 * C, C),
                       self._fused(self.p16, d + "self_pos_q_linear.bias", 2 * C), out=spqk, alpha=scaling, alpha_ncols=C)
        cpq = buf("d_cpq", (Td, C))
        hip.linear_fwd(tp, W(d + "cross_pos_q_linear.weight"), W(d + "cross_pos_q_linear.bias"), out=cpq, alpha=scaling)
        cpk = buf("d_cpk", (T, C))
        hip.linear_fwd(pos_all, W(d + "cross_pos_k_linear.weight"), W(d + "cross_pos_k_linear.bias"), out=cpk)
        ctx.update(d_spq=spqk[:, :C], d_spk=spqk[:, C:], d_cpq=cpq, d_cpk=cpk)
        causal = not ctx["full"]
        d_r2, d_r1, d_rx = self._rel_tables_all(
            "d_seg", ["%sseg_rel_pos_table_list.%d.weight" % (d, l) for l in range(cfg.dec_layers)],
            [(True, g["dec_idx2d"]), (True, g["dec_idx1d"]), (True, g["dec_idxx"])])
        if resized:
            with self._wgrad():
                bufs = dict(self.model.named_buffers())
                absb = R.abs_bias(ctx["d_spq"], ctx["d_spk"], H)
                if causal:
                    absb = absb.masked_fill(R.causal_mask(P, dev)[None], float("-inf"))
                for l in range(cfg.dec_layers):
                    relb = R.decoder_rel_bias(W("%sseg_rel_pos_table_list.%d.weight" % (d, l)).float(), bufs["decoder.seg_rp_bucket"], (h, w), sb)
                    ctx["dense"]["d%d" % l] = self._dense_from("d%d" % l, H, Td, Td, absb + relb)
                    self._dense_built(ctx, "d%d" % l)
                    if l == 0:
                        ctx["dense"]["dc"] = self._dense_bias("dc", H, Td, T, cpq, cpk, None, False, None)
                        self._dense_built(ctx, "dc")
                if self.overlap:
                    if getattr(self, "_dense_ev", None) is None:
                        self._dense_ev = torch.cuda.Event()
                    self._dense_ev.record(self._side)
                    ctx["dense_ready"] = self._dense_ev
        elif bi:
            with self._wgrad():
                for l in range(cfg.dec_layers if "d" in bi_which else 0):
                    rel = hip.RelBias(P, g["gcode"], g["code_bias"], d_r2[l], d_r1[l], d_rx[l], grid_w=w)
                    ctx["dense"]["d%d" % l] = self._dense_bias("d%d" % l, H, Td, Td, ctx["d_spq"], ctx["d_spk"], rel, causal, P)
                    self._dense_built(ctx, "d%d" % l)
                    if l == 0 and "c" in bi_which:      # (the first decoder layer's cross attention follows its self attention)
                        ctx["dense"]["dc"] = self._dense_bias("dc", H, Td, T, cpq, cpk, None, False, None)
                        self._dense_built(ctx, "dc")
                # the cross-attention bias has no per-layer part (decoder_module.py:556-558): one operand for all layers
                if "c" in bi_which and "dc" not in ctx["dense"]:
                    ctx["dense"]["dc"] = self._dense_bias("dc", H, Td, T, cpq, cpk, None, False, None)
                    self._dense_built(ctx, "dc")
                if self.overlap:        # (a dedicated event: the ring of `_ev` wraps around long before the backward waits for it)
                    if getattr(self, "_dense_ev", None) is None:
                        self._dense_ev = torch.cuda.Event()
                    self._dense_ev.record(self._side)
                    ctx["dense_ready"] = self._dense_ev
        ctx["d_causal"], ctx["d_rel"] = causal, (d_r2, d_r1, d_rx)

    # ---------------------------------------------------------- resized-grid slow path (eval)
    def _resized_biases(self, kind, h, w, Lt, causal):
        """fp32 [layers, H, T, T] dense relative-position biases of a (h, w) feature grid, T = h*w + Lt: the reference's
        doubly bilinear-resized [H, P0, P0] bias (encoder_module.py:802-808, decoder_module.py:603-627), one HIP kernel per
        layer (csrc/resize.hip) from the delta tables of the TRAINED grid.  Cached per (grid, prompt length, mask) for as
        long as the weights do not change (`_wver`): at evaluation the weights are fixed and a validation set repeats a
        handful of aspect ratios (ADE20K: 512x683, 683x512, ...) -- ~1.1 GB per shape for SegOFA-Base, 16 shapes kept."""
        cfg = self.cfg
        key = (kind, h, w, Lt, bool(causal), self._wver)
        cache = self._rb_cache
        if key in cache:
            cache[key] = cache.pop(key)                   # most recently used last
            return cache[key]
        H = cfg.heads
        if kind == "e":
            oh = cfg.orig_patch_image_size // 16
            g0 = self._geometry(oh, oh, Lt)
            e = "encoder."
            (r2,) = self._rel_tables_all("rs_e_img", ["%simage_rel_pos_table_list.%d.weight" % (e, l) for l in range(cfg.enc_layers)],
                                         [(True, g0["enc_idx2d"])])
            (r1,) = self._rel_tables_all("rs_e_tok", ["%stoken_rel_pos_table_list.%d.weight" % (e, l) for l in range(cfg.enc_layers)],
                                         [(True, g0["enc_idx1d"])])
            rx, nl = None, cfg.enc_layers
        else:
            oh = cfg.seg_bucket_size
            g0 = self._geometry(oh, oh, 1)
            d = "decoder."
            r2, r1, rx = self._rel_tables_all("rs_d_seg", ["%sseg_rel_pos_table_list.%d.weight" % (d, l) for l in range(cfg.dec_layers)],
                                              [(True, g0["dec_idx2d"]), (True, g0["dec_idx1d"]), (True, g0["dec_idxx"])])
            nl = cfg.dec_layers
        T = h * w + Lt
        # rows padded to a multiple of 4 floats: the attention kernel then seeds its accumulators with 16-byte loads (the pad
        # columns are never read as keys: j < S)
        out = torch.zeros(nl, H, T, (T + 3) // 4 * 4, dtype=torch.float32, device=self.device)[..., :T]
        for l in range(nl):
            hip.resized_rel_bias(out[l], r2[l], r1[l], rx[l] if rx is not None else None, h, w, oh, oh, Lt, causal=causal)
        while len(cache) >= 2 * 16:
            cache.pop(next(iter(cache)))                  # least recently used
        for k in [k for k in cache if k[-1] != self._wver]:      # entries of older weights can never be hit again (~0.5 GB each)
            del cache[k]
        cache[key] = out
        return out

    def _forward_resized(self, src_tokens, feat, h, w, prev_output_tokens, full_context_alignment):
        """encoder_module.py:360-368,802-808 and decoder_module.py:541-548,603-627 when (h, w) differs from
        the trained grid.  The dense biases are built with a few PyTorch ops on PARAMETER-sized tensors
        (no activation is touched); every activation op is still a HIP kernel."""
        cfg, dev = self.cfg, self.device
        W, Wf, buf = self.W, self.Wf, self.buf
        self.drop_on = False
        B, L = src_tokens.shape
        C, H = cfg.embed_dim, cfg.heads
        P, T, Td = h * w, h * w + L, h * w + 1
        scaling = float(cfg.head_dim * cfg.attn_scale_factor) ** -0.5
        oh = cfg.orig_patch_image_size // 16
        bsz, sb = cfg.image_bucket_size, cfg.seg_bucket_size
        e, d = "encoder.", "decoder."
        ids = lambda hh, ww, b: (torch.arange(ww, device=dev)[None, :] + torch.arange(hh, device=dev)[:, None] * b + 1).reshape(-1)
        ctx = {"B": B, "L": L, "P": P, "T": T, "Td": Td, "h": h, "w": w, "full": bool(full_context_alignment),
               "src_tokens": src_tokens, "feat": feat, "resized": True}
        # ---- embeddings (same kernels as the fast path)
        bias_img = buf("bias_img", (C,))
        hip.add_bf16(W(e + "image_proj.bias"), W(e + "type_embedding.weight")[1], bias_img)
        img_pre = buf("img_pre", (B * P, C))
        hip.linear_fwd(feat.view(B * P, 1024), W(e + "image_proj.weight"), bias_img, out=img_pre)
        x = buf("e_x_in", (B, T, C))
        hip.ln_fwd(img_pre.view(B, P, C), Wf(e + "patch_layernorm_embedding.weight"),
                   Wf(e + "patch_layernorm_embedding.bias"), x[:, :P])
        tok_pre = buf("tok_pre", (B * L, C))
        hip.embed_rows(W(e + "embed_tokens.weight"), src_tokens.reshape(-1).contiguous(),
                       W(e + "type_embedding.weight")[0], tok_pre)
        hip.ln_fwd(tok_pre.view(B, L, C), Wf(e + "layernorm_embedding.weight"), Wf(e + "layernorm_embedding.bias"), x[:, P:])
        # ---- position embeddings (get_patch_images_info :358-370)
        itab = W(e + "embed_image_positions.weight")
        if P > oh * oh:
            ipos = buf("e_ipos_resized", (P, C))
            hip.resize_rows_bilinear(itab, ipos, h, w, oh, oh, bsz, 1)          # rows y * bsz + x + 1 of the table
        else:
            ipos = itab[ids(h, w, bsz)].contiguous()
        pos_all = buf("e_pos_all", (T, C))
        hip.ln_fwd(ipos, Wf(e + "image_pos_ln.weight"), Wf(e + "image_pos_ln.bias"), pos_all[:P])
        hip.ln_fwd(W(e + "embed_positions.weight")[:L], Wf(e + "pos_ln.weight"), Wf(e + "pos_ln.bias"), pos_all[P:])
        pqk = buf("e_pqk", (T, 2 * C))
        hip.linear_fwd(pos_all, self._fused(self.p16, e + "pos_q_linear.weight", 2 * C, C),
                       self._fused(self.p16, e + "pos_q_linear.bias", 2 * C), out=pqk, alpha=scaling, alpha_ncols=C)
        e_dense = self._resized_biases("e", h, w, L, False)
        for l in range(cfg.enc_layers):
            p = "%slayers.%d." % (e, l)
            tg = "e%d" % l
            x, _ = self._self_block_fwd(tg, p, "self_attn", "self_attn_layer_norm", "attn_ln", x, B, T, pqk[:, :C],
                                        pqk[:, C:], None, False, scaling, dense=e_dense[l])
            x, _ = self._ffn_fwd(tg, p, x, B * T)
        enc_out = buf("enc_out", (B, T, C))
        hip.ln_fwd(x.view(B * T, C), Wf(e + "layer_norm.weight"), Wf(e + "layer_norm.bias"), enc_out.view(B * T, C))
        ctx["enc_out"] = enc_out
        # ---- decoder
        y0b = buf("d_bos", (B, 1, C))
        bos = (prev_output_tokens[:, :1] if prev_output_tokens is not None
               else torch.zeros(B, 1, dtype=torch.long, device=dev))
        hip.embed_rows(W(e + "embed_tokens.weight"), bos.reshape(-1).contiguous(), None, y0b)
        y = buf("d_y_in", (B, Td, C))
        hip.ln_fwd(enc_out[:, :P], Wf(d + "layernorm_embedding.weight"), Wf(d + "layernorm_embedding.bias"), y[:, :P])
        hip.ln_fwd(y0b, Wf(d + "layernorm_embedding.weight"), Wf(d + "layernorm_embedding.bias"), y[:, P:])
        segtab = W(d + "embed_seg_positions.weight")
        tgt = buf("d_spos_resized", (Td, C))                                               # internal order: bos last
        hip.resize_rows_bilinear(segtab, tgt[:P], h, w, sb, sb, sb, 1)     # (h, w) == (sb, sb): weights 1 / 0, a plain gather
        tgt[P:].copy_(segtab[:1])
        tp = buf("d_tp", (Td, C))
        hip.ln_fwd(tgt, Wf(d + "seg_pos_ln.weight"), Wf(d + "seg_pos_ln.bias"), tp)
        spqk = buf("d_spqk", (Td, 2 * C))
        hip.linear_fwd(tp, self._fused(self.p16, d + "self_pos_q_linear.weight", 2 * C, C),
                       self._fused(self.p16, d + "self_pos_q_linear.bias", 2 * C), out=spqk, alpha=scaling, alpha_ncols=C)
        cpq = buf("d_cpq", (Td, C))
        hip.linear_fwd(tp, W(d + "cross_pos_q_linear.weight"), W(d + "cross_pos_q_linear.bias"), out=cpq, alpha=scaling)
        cpk = buf("d_cpk", (T, C))
        hip.linear_fwd(pos_all, W(d + "cross_pos_k_linear.weight"), W(d + "cross_pos_k_linear.bias"), out=cpk)
        d_dense = self._resized_biases("d", h, w, 1, not full_context_alignment)
        for l in range(cfg.dec_layers):
            p = "%slayers.%d." % (d, l)
            tg = "d%d" % l
            y, _ = self._self_block_fwd(tg, p, "self_attn", "self_attn_layer_norm", "self_attn_ln", y, B, Td,
                                        spqk[:, :C], spqk[:, C:], None, False, scaling, dense=d_dense[l])
            y, _ = self._cross_block_fwd(tg, p, y, enc_out, B, Td, T, cpq, cpk, scaling)
            y, _ = self._ffn_fwd(tg, p, y, B * Td)
        featb = buf("d_feat", (B, Td, C))
        hip.ln_fwd(y[:, :P], Wf(d + "layer_norm.weight"), Wf(d + "layer_norm.bias"), featb[:, 1:])
        hip.ln_fwd(y[:, P:], Wf(d + "layer_norm.weight"), Wf(d + "layer_norm.bias"), featb[:, :1])
        logits = buf("logits_pad", (B, Td, self.npad))
        if getattr(self, "train_seg", False):
            self.wseg_pad[: cfg.num_seg_tokens].copy_(W("encoder.seg_embed_tokens.weight"))
        hip.linear_fwd(featb.view(B * Td, C), self.wseg_pad, out=logits.view(B * Td, self.npad))
        self.ctx = ctx
        return logits, ctx

    def _site_id(self, site):
        kind, l, k = site
        return 16 + (l * 4 + k) * 2 + (0 if kind == "e" else 1)

    def _self_block_fwd(self, tg, p, attn, ln1, ln2, x, B, T, pq, pk, rel, causal, scaling, dense=None, site=None,
                        next_ln=None, xn_pre=None):
        """xn_pre: ln1(x) if the previous layer already produced it.  -> (block output, pre-LN of the next block or None)"""
        C, H = self.cfg.embed_dim, self.cfg.heads
        W, Wf, buf = self.W, self.Wf, self.buf
        a_ = p + attn
        xn = xn_pre
        if xn is None:
            xn = buf(tg + "_xn", (B * T, C))
            mu, rs = self._ln_stats(tg + "_ln1", B * T)
            hip.ln_fwd(x.view(B * T, C), Wf(p + ln1 + ".weight"), Wf(p + ln1 + ".bias"), xn, mu, rs)
        qkv = buf(tg + "_qkv", (B, T, 3 * C))
        hip.linear_fwd(xn, self._fused(self.p16, a_ + ".q_proj.weight", 3 * C, C),
                       self._fused(self.p16, a_ + ".q_proj.bias", 3 * C), out=qkv.view(B * T, 3 * C),
                       alpha=scaling, alpha_ncols=C)
        o = buf(tg + "_o", (B, T, C))
        lse = buf(tg + "_lse", (B, H, T), torch.float32)
        gain = self._gain32(tg + "_sa", a_ + ".c_attn")
        dd = self.ctx_building.get("dense", {}).get(tg) if (self.bi_fwd and self.ctx_building is not None) else None
        if dd is not None:
            # the forward of the batch-inner formulation: dense bias tile shared by four batch elements, no abs-pos columns in
            # the contraction, no table look-ups, one path for every grid width (csrc/attention_bi.hip: attn_bi_fwd_kernel)
            self._dense_wait(tg)
            # (measured and dropped: the round-3 kernel seeded from the dense bias by global loads, 94.2 vs 91.3 ms on C4)
            hip.attn_fwd_bi(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], dd, o, lse, B, H, T, T, causal=causal,
                            P=rel.P if rel is not None else None, gain=gain, kv_len=self.ctx_building.get("klen") if tg[0] == "e" else None,
                            drop=self._attn_drop(tg, self.attn_drop_p))
        elif self.attn_drop_p:
            raise NotImplementedError("ifseg_amd HIP engine: attention dropout needs the batch-inner attention kernels")
        else:
            hip.attn_fwd(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], pq, pk, o, lse, B, H, T, T, rel=rel,
                         causal=causal, gain=gain, dense_bias=dense)
        a = buf(tg + "_a", (B * T, C))
        hip.linear_fwd(o.view(B * T, C), W(a_ + ".out_proj.weight"), W(a_ + ".out_proj.bias"), out=a)
        x1 = buf(tg + "_x1", (B, T, C))
        drop = self._dropargs(self._site_id(site), self._dp(*site), T) if (self.drop_on and site is not None) else None
        nxt = self._post_ln(a, p + ln2, tg + "_ln2", x.view(B * T, C), x1.view(B * T, C), drop, B * T, next_ln)
        self._save(tg + "_sa", x=x, xn=xn, qkv=qkv, o=o, lse=lse, a=a, rel=rel, gain=gain, causal=causal, site=site)
        return x1, nxt

    def _cross_block_fwd(self, tg, p, y1, enc_out, B, Td, Te, cpq, cpk, scaling, site=None, yn_pre=None, next_ln=None):
        """yn_pre: encoder_attn_layer_norm(y1) if the previous block already produced it.  -> (y2, next pre-LN or None)"""
        C, H = self.cfg.embed_dim, self.cfg.heads
        W, Wf, buf = self.W, self.Wf, self.buf
        a_ = p + "encoder_attn"
        yn = yn_pre
        if yn is None:
            yn = buf(tg + "_cyn", (B * Td, C))
            mu, rs = self._ln_stats(tg + "_cln1", B * Td)
            hip.ln_fwd(y1.view(B * Td, C), Wf(p + "encoder_attn_layer_norm.weight"), Wf(p + "encoder_attn_layer_norm.bias"), yn, mu, rs)
        q = buf(tg + "_cq", (B, Td, C))
        hip.linear_fwd(yn, W(a_ + ".q_proj.weight"), W(a_ + ".q_proj.bias"), out=q.view(B * Td, C), alpha=scaling)
        kv = buf(tg + "_ckv", (B, Te, 2 * C))
        ready = self.ctx_building.get("ckv_ready") if self.ctx_building is not None else None
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready[int(tg[1:])])      # this layer's K|V were projected on the side stream
        else:
            hip.linear_fwd(enc_out.view(B * Te, C), self._fused(self.p16, a_ + ".k_proj.weight", 2 * C, C),
                           self._fused(self.p16, a_ + ".k_proj.bias", 2 * C), out=kv.view(B * Te, 2 * C))
        o = buf(tg + "_co", (B, Td, C))
        lse = buf(tg + "_clse", (B, H, Td), torch.float32)
        gain = self._gain32(tg + "_ca", a_ + ".c_attn")
        dd = self.ctx_building.get("dense", {}).get("dc") if (self.bi_fwd and self.ctx_building is not None) else None
        if dd is not None:
            self._dense_wait("dc")
            hip.attn_fwd_bi(q, kv[:, :, :C], kv[:, :, C:], dd, o, lse, B, H, Td, Te, gain=gain, kv_len=self.ctx_building.get("klen"),
                            drop=self._attn_drop(tg + "c", self.attn_drop_p))
        elif self.attn_drop_p:
            raise NotImplementedError("ifseg_amd HIP engine: attention dropout needs the batch-inner attention kernels")
        else:
            hip.attn_fwd(q, kv[:, :, :C], kv[:, :, C:], cpq, cpk, o, lse, B, H, Td, Te, gain=gain)
        a = buf(tg + "_ca_a", (B * Td, C))
        hip.linear_fwd(o.view(B * Td, C), W(a_ + ".out_proj.weight"), W(a_ + ".out_proj.bias"), out=a)
        y2 = buf(tg + "_y2", (B, Td, C))
        drop = self._dropargs(self._site_id(site), self._dp(*site), Td) if (self.drop_on and site is not None) else None
        nxt = self._post_ln(a, p + "cross_attn_ln", tg + "_cln2", y1.view(B * Td, C), y2.view(B * Td, C), drop, B * Td, next_ln)
        self._save(tg + "_ca", x=y1, xn=yn, q=q, kv=kv, o=o, lse=lse, a=a, gain=gain, site=site)
        return y2, nxt

    def _ffn_fwd(self, tg, p, x1, rows, site=None, rpb=None, xn_pre=None, next_ln=None):
        """xn_pre: final_layer_norm(x1) if the previous block already produced it; next_ln as in _post_ln (used when the
        block output goes through the dropout kernel anyway).  -> (x2, pre-LN of the next layer or None)"""
        C, Fd = self.cfg.embed_dim, self.cfg.ffn_dim
        W, Wf, buf = self.W, self.Wf, self.buf
        xn = xn_pre
        if xn is None:
            xn = buf(tg + "_fxn", (rows, C))
            mu, rs = self._ln_stats(tg + "_fln1", rows)
            hip.ln_fwd(x1.view(rows, C), Wf(p + "final_layer_norm.weight"), Wf(p + "final_layer_norm.bias"), xn, mu, rs)
        u = buf(tg + "_u", (rows, Fd))
        hip.linear_fwd(xn, W(p + "fc1.weight"), W(p + "fc1.bias"), out=u)
        z = buf(tg + "_z", (rows, Fd))
        mu, rs = self._ln_stats(tg + "_fln2", rows)
        eps = 1e-5
        if self.act_drop_p:
            # activation dropout (unify_transformer_layer.py:280,556: between GELU and ffn_layernorm): dropped pre-activations
            # become -30 in place (gelu = gelu' = 0 exactly: every kernel that recomputes gelu(u), forward and backward, then
            # sees the mask); the 1 / (1 - p) cancels in the LayerNorm that follows, exactly, with eps (1 - p)^2 (csrc/rowops.hip)
            hip.dropout_fill(u, u, self.act_drop_p, self._site_seed((4000 if tg[0] == "e" else 5000) + int(tg[1:])))
            eps = 1e-5 * (1.0 - self.act_drop_p) ** 2
        hip.ln_fwd(u, Wf(p + "ffn_layernorm.weight"), Wf(p + "ffn_layernorm.bias"), z, mu, rs, gelu=True, eps=eps)
        x2 = buf(tg + "_x2", x1.shape)
        nxt = None
        dropping = self.drop_on and site is not None
        keep_t = self.ffn_ln_fused and self._train_fwd and site is not None    # the backward reads the fc2 output (row means)
        t = None
        if dropping or keep_t:
            t = buf(tg + "_t", (rows, C)) if keep_t else buf("drop_tmp_%d" % rows, (rows, C))
            hip.linear_fwd(z, W(p + "fc2.weight"), W(p + "fc2.bias"), out=t)
            drop = (self.cfg.dropout, self._site_seed(self._site_id(site)), self._dp(*site), rpb) if dropping else None
            if next_ln is not None:
                # dropout + DropPath + residual of this block and the pre-LN of the next layer in one launch
                npname, ntag, nxt = next_ln
                mu2, rs2 = self._ln_stats(ntag, rows)
                hip.ln_fwd_pair(t, None, None, x2.view(rows, C), None, None, Wf(npname + ".weight"), Wf(npname + ".bias"), nxt,
                                mu2, rs2, resid=x1.view(rows, C), drop=drop)
            elif dropping:
                self._drop(t, x1.view(rows, C), x2.view(rows, C), self._site_id(site), self._dp(*site), rpb)
            else:
                hip.dropout(t, x1.view(rows, C), x2.view(rows, C), 0.0, 0, None, None)       # x2 = x1 + t
        else:
            hip.linear_fwd(z, W(p + "fc2.weight"), W(p + "fc2.bias"), out=x2.view(rows, C), resid=x1.view(rows, C))
        self._save(tg + "_ffn", x1=x1, xn=xn, u=u, z=z, site=site, rpb=rpb, t=t if keep_t else None)
        return x2, nxt

    def _save(self, key, **kw):
        self.saved[key] = kw

    def _post_ln(self, a, pname, stats_tag, resid, out, drop, rows, next_ln):
        """out = resid + drop(LN(a)); with next_ln = (param prefix, stats tag, output [rows, C]) the pre-LN of the next
        block is computed in the same launch and returned (else None)."""
        C = self.cfg.embed_dim
        W, Wf = self.W, self.Wf
        mu, rs = self._ln_stats(stats_tag, rows)
        if next_ln is None:
            hip.ln_fwd(a, Wf(pname + ".weight"), Wf(pname + ".bias"), out, mu, rs, resid=resid, drop=drop)
            return None
        npname, ntag, xn = next_ln
        mu2, rs2 = self._ln_stats(ntag, rows)
        hip.ln_fwd_pair(a, Wf(pname + ".weight"), Wf(pname + ".bias"), out, mu, rs, Wf(npname + ".weight"), Wf(npname + ".bias"),
                        xn, mu2, rs2, resid=resid, drop=drop)
        return xn

    # ---------------------------------------------------------------- backward
    def _ln_bwd(self, dy, x, pname, stats_tag, dx, dx_add=None, gelu=False, accumulate=False, drop=None):
        C = x.shape[-1]
        rows = x.numel() // C
        mu, rs = self._ln_stats(stats_tag, rows)
        # per-block partial sums of d(gamma), d(beta); one buffer per LayerNorm site (the reduction below
        # runs on the side stream)
        part = self._ln_part(C, stats_tag)
        hip.ln_bwd(dy, x, self.Wf(pname + ".weight"), mu, rs, dx, part[0], part[1], dx_add=dx_add, gelu=gelu, drop=drop)
        # weight and bias of a LayerNorm are adjacent in the arena: one [2, C] reduction
        # (an accumulating reduction adds to what an earlier LayerNorm launch of the same parameters produced: it must not
        # share a launch with it)
        (self._ln_red_acc if accumulate else self._ln_red_tasks).append(
            (part, self._fused(self.g16, pname + ".weight", 2, C), 2, hip.LN_BWD_BLOCKS, C, accumulate))
        return dx

    def _ln_part(self, C, stats_tag):
        # one buffer per LayerNorm site: the reductions of a layer block are batched into one later launch
        return self.buf("ln_dgbp_%d@%s" % (C, stats_tag), (2, hip.LN_BWD_BLOCKS, C), torch.float32)

    def _ln_bwd_fused(self, dy, x, pname, stats_tag, dx, dx_add, nxt):
        """`_ln_bwd` of a block's pre-LN plus, in the same launch, the fc2-dropout adjoint that opens the NEXT block of the
        backward (`nxt` from `_next_drop`; None: plain `_ln_bwd`)."""
        C = x.shape[-1]
        rows = x.numel() // C
        if nxt is None or C > 1024:
            self._ln_bwd(dy, x, pname, stats_tag, dx, dx_add=dx_add)
            if nxt is not None:          # wide models: two launches
                p_, seed, dp, rpb = nxt["drop"]
                hip.dropout(dx, None, nxt["out"], p_, seed, dp, rpb)
            return dx
        mu, rs = self._ln_stats(stats_tag, rows)
        part = self._ln_part(C, stats_tag)
        hip.ln_bwd_drop(dy, x, self.Wf(pname + ".weight"), mu, rs, dx, part[0], part[1], nxt["out"], dx_add=dx_add,
                        drop2=nxt["drop"])
        self._ln_red_tasks.append((part, self._fused(self.g16, pname + ".weight", 2, C), 2, hip.LN_BWD_BLOCKS, C, False))
        return dx

    def _next_drop(self, tg, rows):
        """descriptor of the fc2 dropout adjoint that opens the FFN block of layer `tg` in the backward (None if inactive)"""
        s = self.saved[tg + "_ffn"]
        if not (self.drop_on and s["site"] is not None):
            return None
        bt, self._bt = self._bt, tg + "f"
        out = self.gbuf("g_drop_%d" % rows, (rows, self.cfg.embed_dim))
        self._bt = bt
        return dict(kind="drop", out=out, drop=(self.cfg.dropout, self._site_seed(self._site_id(s["site"])), self._dp(*s["site"]), s["rpb"]))

    def _bias_grad(self, dy2d, gout, accumulate=False):
        N = dy2d.shape[-1]
        # one partial buffer per STREAM: at the end of the backward the main stream (embedding half of the tail) and the side
        # stream (last weight gradients; bias gradients of projections too small for the grouped dW launch) run this side by
        # side -- with one shared buffer the two column sums raced (seen as a rare 1-ulp difference in a bias gradient of
        # encoder layer 0 on small models; deterministic under IFSEG_POISON_WS=1)
        part = self.buf("colsum_%d@%d" % (N, torch.cuda.current_stream().stream_id), (hip.COLSUM_BLOCKS, N), torch.float32)
        hip.colsum(dy2d, part)
        hip.reduce_parts(part, gout, 1, hip.COLSUM_BLOCKS, N, accumulate=accumulate)

    def _linear_bwd(self, dy, x, wname_or_view, gw, gb, dx_out=None, dx_resid=None, dx_accumulate=False, need_dx=True):
        """dy [M,N], x [M,K]: writes dW -> gw, db -> gb, returns dx [M,K]"""
        def wgrad():
            if not hip.linear_dw(dy, x, gw, bias_out=gb) and gb is not None:
                self._bias_grad(dy, gb)
        if self.overlap and self.dw_grouped and dy.shape[0] >= 4096 and hip.dw_groupable(dy, x, gw, gb):
            # operands live in per-block buffers (gbuf) / saved activations: still intact at the end of the layer
            self._dw_tasks.append((dy, x, gw, gb))
        else:
            self._side_do(wgrad)
        if need_dx:
            return hip.linear_dx(dy, wname_or_view, out=dx_out, resid=dx_resid, accumulate=dx_accumulate)
        return None

    def _ffn_bwd(self, tg, p, dx2, rows, dbr_pre=None, nxt=None):
        """dx2: grad of the block output [rows, C]; returns grad of x1 (block input).  dbr_pre: the fc2 dropout adjoint of
        dx2 if the previous block of the backward already produced it; nxt: see `_ln_bwd_fused`."""
        C, Fd = self.cfg.embed_dim, self.cfg.ffn_dim
        s = self.saved[tg + "_ffn"]
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        self._bt = tg + "f"
        gbuf = self.gbuf
        dbr = dx2
        if dbr_pre is not None:
            dbr = dbr_pre
        elif self.drop_on and s["site"] is not None:      # adjoint of dropout + DropPath on the branch
            dbr = self._drop(dx2, None, gbuf("g_drop_%d" % rows, (rows, C)), self._site_id(s["site"]), self._dp(*s["site"]), s["rpb"])
        du = gbuf("g_du_%d" % rows, (rows, Fd))
        if self.ffn_ln_fused and s.get("t") is not None:
            # fc2 dW / db as usual; dX with the ffn_layernorm + GELU backward in its epilogue: the two row means come from the
            # 768-wide dbr and the saved fc2 output, dz is never written (no wide LayerNorm-backward kernel)
            self._linear_bwd(dbr, s["z"], W(p + "fc2.weight"), G(p + "fc2.weight"), G(p + "fc2.bias"), need_dx=False)
            mu, rs = self._ln_stats(tg + "_fln2", rows)
            cst = buf(tg + "_fc12", (rows, 2), torch.float32)
            hip.ffn_ln_rowstats(dbr, s["t"].view(rows, C), self.ws[tg + "_fcoef"], cst, Fd)
            hip.linear_dx_gelu_ln_bwd(dbr, W(p + "fc2.weight"), du, s["u"], Wf(p + "ffn_layernorm.weight"), mu, rs, cst)
            self._ffn_pg_tasks.append((W(p + "fc2.weight"), G(p + "fc2.weight"), G(p + "fc2.bias"), Wf(p + "ffn_layernorm.weight"),
                                       Wf(p + "ffn_layernorm.bias"), G(p + "ffn_layernorm.weight"), G(p + "ffn_layernorm.bias"),
                                       dbr, s["u"], mu, rs))
        else:
            dz = buf("g_dz_%d" % rows, (rows, Fd))
            self._linear_bwd(dbr, s["z"], W(p + "fc2.weight"), G(p + "fc2.weight"), G(p + "fc2.bias"), dx_out=dz)
            if "lnwide" not in _EXP_SKIP:
                self._ln_bwd(dz, s["u"], p + "ffn_layernorm", tg + "_fln2", du, gelu=True)
        dxn = buf("g_dxn_%d" % rows, (rows, C))
        self._linear_bwd(du, s["xn"], W(p + "fc1.weight"), G(p + "fc1.weight"), G(p + "fc1.bias"), dx_out=dxn)
        dx1 = gbuf("g_dx1_%d" % rows, (rows, C))
        self._ln_bwd_fused(dxn, s["x1"].view(rows, C), p + "final_layer_norm", tg + "_fln1", dx1, dx2, nxt)
        self._side_flush()
        return dx1

    def _out_proj_bwd(self, da, o, a_, do, B, T):
        """dO = da . W_out (+ dW, db on the side stream) and delta = rowsum(dO * O) per head from the same GEMM epilogue
        -> delta [B, H, T] fp32 (None: the attention backward computes it in its own launch)"""
        C, H = self.cfg.embed_dim, self.cfg.heads
        rows = B * T
        W, G = self.W, self.G
        if not self.delta_fused:
            self._linear_bwd(da, o.view(rows, C), W(a_ + ".out_proj.weight"), G(a_ + ".out_proj.weight"),
                             G(a_ + ".out_proj.bias"), dx_out=do.view(rows, C))
            return None
        self._linear_bwd(da, o.view(rows, C), W(a_ + ".out_proj.weight"), G(a_ + ".out_proj.weight"),
                         G(a_ + ".out_proj.bias"), need_dx=False)
        delta = self.gbuf("g_delta_%d" % T, (B, H, T), torch.float32)
        hip.linear_dx_rowdot(da, W(a_ + ".out_proj.weight"), do.view(rows, C), o.view(rows, C), delta, T)
        return delta

    def _attn_core_bwd(self, tag, q, k, v, pq, pk, o, lse, do, dq, dk, dv, B, T, S, rel, causal, gain, gain_name,
                       scaling, dpq_acc, dpk_acc, first_pos, rel_grads, delta=None, dense=None):
        cfg = self.cfg
        C, H = cfg.embed_dim, cfg.heads
        buf = self.buf
        gbuf = self.gbuf
        have_delta = delta is not None
        if not have_delta:
            delta = gbuf("g_delta_%d" % T, (B, H, T), torch.float32)
        # key padding: the keys of the encoder's self-attention (tags e<l>) and of the decoder's cross-attention (d<l>c) are the
        # encoder positions
        kv_len = self.ctx.get("klen") if (tag[0] == "e" or tag.endswith("c")) else None
        adrop = self._attn_drop(tag, self.ctx.get("attn_drop_p", 0.0))
        if dense is not None:
            return self._attn_core_bwd_bi(q, k, v, pq, pk, o, lse, do, dq, dk, dv, B, T, S, rel, causal, gain, gain_name,
                                          scaling, dpq_acc, dpk_acc, first_pos, rel_grads, delta, have_delta, dense, kv_len, adrop)
        if adrop is not None:
            raise NotImplementedError("ifseg_amd HIP engine: attention dropout needs the batch-inner attention backward")
        if kv_len is not None:
            raise NotImplementedError("ifseg_amd HIP engine: key padding needs the batch-inner attention backward")
        dpq_part = gbuf("g_dpq_part_%d" % T, (B, T, C))        # bf16 per-batch partials, summed by attn_bwd_reduce
        dpk_part = gbuf("g_dpk_part_%d" % S, (B, S, C))
        nparts = B * ((S + 127) // 128)
        parts = [None, None, None]
        if rel is not None:
            parts = [gbuf("g_relp%d_%d" % (i, t.shape[1]), (H, nparts, t.shape[1]), torch.float32)
                     for i, t in enumerate((rel.rel2d, rel.rel1d, rel.relx))]
        # per-row terms of d c_attn (sum_j P dP = dO . O_pre) from the dQ kernel: no division by c_attn anywhere
        dgr = gbuf("g_dgain_rows_%d" % T, (B, H, T), torch.float32)
        ones = self.ws.get("ones_h")
        if ones is None or ones.numel() != H:
            ones = self.ws["ones_h"] = torch.ones(H, dtype=torch.float32, device=self.device)
        kw = dict(rel=rel, causal=causal, gain=gain, dq_scale=scaling, dpq_scale=scaling, drel2d_part=parts[0],
                  drel1d_part=parts[1], drelx_part=parts[2], nparts=nparts, dgain_rows=dgr)
        args = (q, k, v, pq, pk, o, do, lse, delta, dq, dk, dv, dpq_part, dpk_part, B, H, T, S)
        timing = self.attn_bwd_timing
        if timing is not None:
            timing["seen"] += 1
            if (timing["seen"] - 1) % timing["stride"]:
                timing = None
            else:       # one event pair on the main stream around delta + dK/dV + dQ (bench.py's roofline object)
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
        if self.overlap and not lab.get("DQ_SERIAL"):
            # dK/dV and dQ are independent once delta exists; each leaves its last round of workgroups partly
            # empty (864 workgroups on 512 slots), so they run on two streams and fill each other's holes
            if not have_delta:
                hip.attn_bwd(*args, phases=hip.ATTN_BWD_DELTA, **kw)
            with self._fork(self._dq_stream_get()):
                if "dq" not in _EXP_SKIP:
                    hip.attn_bwd(*args, phases=hip.ATTN_BWD_DQ, **kw)
                dq_done = self._ev()
                dq_done.record(self._dqs)
            if "dkv" not in _EXP_SKIP:
                hip.attn_bwd(*args, phases=hip.ATTN_BWD_DKV, **kw)
            torch.cuda.current_stream().wait_event(dq_done)
        else:
            hip.attn_bwd(*args, phases=(hip.ATTN_BWD_DKV | hip.ATTN_BWD_DQ) if have_delta else 0, **kw)
        if timing is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            timing["pairs"].append((t0, t1, 8.0 * 64 * T * S * B * H))   # dV, dP, dK, dQ of the reference at d = 64
        def reductions():
            # one launch: batch sums of the abs-pos operand gradients, d c_attn[h] = sum_{b,t} delta / c_attn[h], and for
            # every rel-pos table the sum over the workgroup partials scattered into its bucket accumulator
            tables = []
            if rel is not None:
                for (tabname, idx), part in zip(rel_grads, parts):
                    if tabname is not None:
                        tables.append((part, idx, self._table_acc(tabname)))
            # (d c_attn[h] = sum of the dQ kernel's row terms: `ones` stands in for the gain the round-3 formula divided delta by)
            hip.attn_bwd_reduce(B, H, T, S, C, dpq_part, dpk_part, dpq_acc, dpk_acc, not first_pos, dgr, ones,
                                self.G(gain_name), nparts if rel is not None else 1, tables)
        if "reduce" not in _EXP_SKIP:
            self._side_do(reductions)

    def _attn_core_bwd_bi(self, q, k, v, pq, pk, o, lse, do, dq, dk, dv, B, T, S, rel, causal, gain, gain_name, scaling,
                          dpq_acc, dpk_acc, first_pos, rel_grads, delta, have_delta, dense, kv_len=None, adrop=None):
        """csrc/attention_bi.hip: the bias is the dense operand `dense` (built once per layer by the forward's side stream),
        a workgroup holds four batch elements, sum_b dS leaves the dQ kernel once per tile; everything behind that sum --
        abs-pos operand gradients, rel-pos tables, c_attn -- is two launches on the weight-gradient stream."""
        C, H = self.cfg.embed_dim, self.cfg.heads
        gbuf = self.gbuf
        ng = (B + 3) // 4
        # (causal and full attentions never share a buffer -- single-stream execution has no per-block instances, and the
        # decoder's self- and cross-attention can have the same padded shape: the causal one relies on blocks staying zero)
        # (the causal buffer's name also carries P: the blocks a causal launch skips depend on it, and they must never have been
        # written under this name -- ADVICE r4)
        dbias = gbuf("g_dbias%s_%dx%d" % ("c%d" % (rel.P if rel is not None else 0) if causal else "", T, dense.Sp), (ng, H, T, dense.Sp))
        if not getattr(dbias, "_ifseg_zeroed", False):
            # causal launches never write the blocks above the diagonal (the same blocks every step), and on grids that are not
            # 32 wide the table kernel reads masked pairs of a grid row that lie in such blocks: zero once per ALLOCATION (a
            # smaller last batch reallocates the buffer under the same name)
            dbias.zero_()
            dbias._ifseg_zeroed = True
        if not have_delta:
            hip.attn_bwd(q, k, v, None, None, o, do, lse, delta, dq, dk, dv, None, None, B, H, T, S, phases=hip.ATTN_BWD_DELTA)
        timing = self.attn_bwd_timing
        if timing is not None:
            timing["seen"] += 1
            if (timing["seen"] - 1) % timing["stride"]:
                timing = None
            else:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record()
        P = rel.P if rel is not None else None
        # per-row terms of d c_attn (sum_j P dP = dO . O_pre) from the dQ kernel: no division by c_attn anywhere
        dgr = gbuf("g_dgain_rows_%d" % T, (B, H, T), torch.float32)
        ones = self.ws.get("ones_h")
        if ones is None or ones.numel() != H:
            ones = self.ws["ones_h"] = torch.ones(H, dtype=torch.float32, device=self.device)
        ph = 0
        if "dq" in _EXP_SKIP:
            ph = hip.ATTN_BWD_DKV
        if "dkv" in _EXP_SKIP:
            ph = hip.ATTN_BWD_DQ if not ph else -1
        if ph >= 0:
            hip.attn_bwd_bi(q, k, v, do, lse, delta, dense, dq, dk, dv, dbias, B, H, T, S, causal=causal, P=P, gain=gain,
                            dq_scale=scaling, phases=ph, dgain_rows=dgr, kv_len=kv_len, drop=adrop)
        if timing is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            timing["pairs"].append((t0, t1, 8.0 * 64 * T * S * B * H))
        parts = [None, None, None]
        if rel is not None:
            parts = [gbuf("g_relg%d_%d" % (i, t.shape[1]), (H, hip.dbias_nparts(), t.shape[1]), torch.float32)
                     for i, t in enumerate((rel.rel2d, rel.rel1d, rel.relx))]

        def reductions():
            kw = {}
            if rel is not None:
                kw = dict(P=rel.P, grid_h=rel.P // rel.grid_w, grid_w=rel.grid_w, drel2d=parts[0], drel1d=parts[1], drelx=parts[2])
            hip.attn_dbias_grads(dbias, S, pos_q=pq, pos_k=pk, dpq_acc=dpq_acc, dpk_acc=dpk_acc, accumulate_pos=not first_pos,
                                 dpq_scale=scaling, causal=causal, **kw)
            tables = []
            if rel is not None:
                for (tabname, idx), part in zip(rel_grads, parts):
                    if tabname is not None:
                        tables.append((part, idx, self._table_acc(tabname)))
            # bucket scatter of the delta-table gradients and d c_attn (no abs-pos partials: nothing to sum over the batch)
            # (d c_attn[h] = sum of the dQ kernel's row terms: `ones` stands in for the gain the round-3 formula divides by)
            hip.attn_bwd_reduce(B, H, T, S, C, None, None, dpq_acc, dpk_acc, True, dgr, ones, self.G(gain_name),
                                hip.dbias_nparts(), tables)
            if self.ctx.get("resized") and rel_grads is not None:
                self._resized_rel_grads(rel_grads, dbias, T)
        if "reduce" not in _EXP_SKIP:
            self._side_do(reductions)

    def _resized_rel_grads(self, rel_grads, dbias, T):
        """(side stream) training on a resized grid: the rel-pos bucket tables' gradients = the adjoint of
        models/segofa/resized.py's bias map applied to sum_b dS [ng, H, T, Sp] (bf16, from the dQ kernel)"""
        cfg, ctx = self.cfg, self.ctx
        h, w, L = ctx["h"], ctx["w"], ctx["L"]
        names = []
        for n, _ in rel_grads:
            if n is not None and n not in names:
                names.append(n)
        dB = dbias[:, :, :, :T].float().sum(0)
        bufs = dict(self.model.named_buffers())
        if names[0].startswith("encoder."):
            img_n = [n for n in names if "image_rel" in n][0]
            tok_n = [n for n in names if "token_rel" in n][0]
            oh = cfg.orig_patch_image_size // 16
            fn = lambda tt, it: R.encoder_rel_bias(tt, it, bufs["encoder.token_rp_bucket"], bufs["encoder.image_rp_bucket"], (h, w), oh,
                                                   cfg.image_bucket_size, L)
            order = [tok_n, img_n]
        else:
            fn = lambda st: R.decoder_rel_bias(st, bufs["decoder.seg_rp_bucket"], (h, w), cfg.seg_bucket_size)
            order = names[:1]
        for n, gv in zip(order, R.table_grads(fn, [self.W(n) for n in order], dB)):
            self.G(n).copy_(gv)

    def _table_acc(self, tabname):
        """fp32 accumulator of one rel-pos bucket table's gradient.  All of them are views of ONE flat buffer that the backward
        clears with a single fill (`_tabacc_clear`) instead of one fill per table and step."""
        key = "g_tabacc_" + tabname
        if key not in self.ws:
            names = sorted(n for n in self.shapes if "rel_pos_table_list" in n and self.offs[n] < self.n_train)
            assert tabname in names, tabname
            sizes = [(math.prod(self.shapes[n]) + 3) // 4 * 4 for n in names]
            flat = self.ws["g_tabacc_flat"] = torch.zeros(sum(sizes), dtype=torch.float32, device=self.device)
            o = 0
            for n, sz in zip(names, sizes):
                self.ws["g_tabacc_" + n] = flat[o:o + math.prod(self.shapes[n])].view(self.shapes[n])
                o += sz
        self._tab_touched.setdefault(key, tabname)
        return self.ws[key]

    def _tabacc_clear(self):
        """(side stream, ahead of the step's first attention reduction)"""
        flat = self.ws.get("g_tabacc_flat")        # (None in the very first backward: created zero-filled further down the queue)
        if flat is not None:
            flat.zero_()

    def _self_block_bwd(self, tg, p, attn, ln1, ln2, dx1, B, T, pq, pk, scaling, dpq_acc, dpk_acc, first_pos, rel_grads,
                        da_pre=None, nxt=None):
        """da_pre: the post-LN backward of dx1 if the previous block of the backward already ran it; nxt: `_ln_bwd_fused`"""
        C = self.cfg.embed_dim
        s = self.saved[tg + "_sa"]
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        a_ = p + attn
        rows = B * T
        self._bt = tg + "s"
        gbuf = self.gbuf
        da = da_pre
        if da is None:
            da = gbuf("g_da_%d" % rows, (rows, C))
            drop = None
            if self.drop_on and s["site"] is not None:       # adjoint of the dropout fused into the forward LN
                drop = self._dropargs(self._site_id(s["site"]), self._dp(*s["site"]), T)
            self._ln_bwd(dx1, s["a"], p + ln2, tg + "_ln2", da, drop=drop)
        do = buf("g_do_%d" % rows, (B, T, C))
        delta = self._out_proj_bwd(da, s["o"], a_, do, B, T)
        dqkv = gbuf("g_dqkv_%d" % rows, (B, T, 3 * C))
        qkv = s["qkv"]
        self._attn_core_bwd(tg, qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], pq, pk, s["o"], s["lse"], do,
                            dqkv[:, :, :C], dqkv[:, :, C:2 * C], dqkv[:, :, 2 * C:], B, T, T, s["rel"], s["causal"],
                            s["gain"], a_ + ".c_attn", scaling, dpq_acc, dpk_acc, first_pos, rel_grads, delta=delta,
                            dense=self.ctx.get("dense", {}).get(tg))
        dxn = buf("g_dxn_%d" % rows, (rows, C))
        self._linear_bwd(dqkv.view(rows, 3 * C), s["xn"], self._fused(self.p16, a_ + ".q_proj.weight", 3 * C, C),
                         self._fused(self.g16, a_ + ".q_proj.weight", 3 * C, C),
                         self._fused(self.g16, a_ + ".q_proj.bias", 3 * C), dx_out=dxn)
        if self.kproj_fix:
            self._kfix_tasks.append((G(a_ + ".k_proj.weight"), G(a_ + ".k_proj.bias"), s["xn"], tg))
        dx = gbuf("g_dx0_%d" % rows, (rows, C))
        self._ln_bwd_fused(dxn, s["x"].view(rows, C), p + ln1, tg + "_ln1", dx, dx1, nxt)
        return dx                # the caller flushes the side queue together with the layer's hook

    def _cross_block_bwd(self, tg, p, dy2, B, Td, Te, cpq, cpk, scaling, d_enc_out, first_cross, dcpq_acc, dcpk_acc,
                         da_pre=None, nxt=None):
        C = self.cfg.embed_dim
        s = self.saved[tg + "_ca"]
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        a_ = p + "encoder_attn"
        rows = B * Td
        self._bt = tg + "c"
        gbuf = self.gbuf
        da = da_pre
        if da is None:
            da = gbuf("g_da_%d" % rows, (rows, C))
            drop = None
            if self.drop_on and s["site"] is not None:
                drop = self._dropargs(self._site_id(s["site"]), self._dp(*s["site"]), Td)
            self._ln_bwd(dy2, s["a"], p + "cross_attn_ln", tg + "_cln2", da, drop=drop)
        do = buf("g_do_%d" % rows, (B, Td, C))
        delta = self._out_proj_bwd(da, s["o"], a_, do, B, Td)
        dq = gbuf("g_cdq", (B, Td, C))
        dkv = gbuf("g_cdkv", (B, Te, 2 * C))
        kv = s["kv"]
        self._attn_core_bwd(tg + "c", s["q"], kv[:, :, :C], kv[:, :, C:], cpq, cpk, s["o"], s["lse"], do, dq,
                            dkv[:, :, :C], dkv[:, :, C:], B, Td, Te, None, False, s["gain"], a_ + ".c_attn", scaling,
                            dcpq_acc, dcpk_acc, first_cross, None, delta=delta, dense=self.ctx.get("dense", {}).get("dc"))
        dyn = buf("g_dxn_%d" % rows, (rows, C))
        self._linear_bwd(dq.view(rows, C), s["xn"], W(a_ + ".q_proj.weight"), G(a_ + ".q_proj.weight"),
                         G(a_ + ".q_proj.bias"), dx_out=dyn)
        # K/V projections read the encoder output: accumulate into d_enc_out
        enc2d = self.ctx["enc_out"].view(B * Te, C)
        # the K|V projections read the encoder output: their dX accumulates into d_enc_out, which nothing needs before
        # the encoder backward starts -- the whole linear backward (dW, db, dX) goes to the side stream, in layer order
        wkv = self._fused(self.p16, a_ + ".k_proj.weight", 2 * C, C)
        if self.overlap:
            # dW / db with the rest of the weight-gradient work; the dX that accumulates into d_enc_out goes to the dQ
            # stream (idle between attentions, in layer order): the decoder->encoder hand-over then waits for a stream
            # that is a few microseconds behind, not for the whole weight-gradient queue
            self._side_do(lambda: self._linear_bwd(
                dkv.view(B * Te, 2 * C), enc2d, wkv, self._fused(self.g16, a_ + ".k_proj.weight", 2 * C, C),
                self._fused(self.g16, a_ + ".k_proj.bias", 2 * C), need_dx=False))
            with self._fork(self._dq_stream_get()):
                hip.linear_dx(dkv.view(B * Te, 2 * C), wkv, out=d_enc_out.view(B * Te, C), accumulate=not first_cross)
        else:
            self._linear_bwd(dkv.view(B * Te, 2 * C), enc2d, wkv, self._fused(self.g16, a_ + ".k_proj.weight", 2 * C, C),
                             self._fused(self.g16, a_ + ".k_proj.bias", 2 * C), dx_out=d_enc_out.view(B * Te, C),
                             dx_accumulate=not first_cross)
        if self.kproj_fix:       # (every layer's K|V projection reads the same encoder output: one column sum per step)
            self._kfix_tasks.append((G(a_ + ".k_proj.weight"), G(a_ + ".k_proj.bias"), enc2d, "enc_out"))
        dy1 = gbuf("g_dy1c_%d" % rows, (rows, C))
        self._ln_bwd_fused(dyn, s["x"].view(rows, C), p + "encoder_attn_layer_norm", tg + "_cln1", dy1, dy2, nxt)
        self._side_flush()
        return dy1

    def _backward(self, dlogits):
        """dlogits: [B, P+1, nseg] (any float dtype) in reference order.  Fills the gradient arena."""
        cfg, ctx = self.cfg, self.ctx
        B, L, P, T, Td, h, w = (ctx[k] for k in ("B", "L", "P", "T", "Td", "h", "w"))
        C, H = cfg.embed_dim, cfg.heads
        scaling = float(cfg.head_dim * cfg.attn_scale_factor) ** -0.5
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        g = self._geometry(h, w, L)
        self.mark("bwd_start")
        if ctx.get("g16_zeroed") is not None:
            # the gradient arena (218 MB for SegOFA-Base) was cleared on the side stream during the forward: the main stream only
            # orders itself behind that fill instead of running it (50 us) at the head of the backward
            torch.cuda.current_stream().wait_event(ctx.pop("g16_zeroed"))
        else:
            self.g16.zero_()
        self._tab_touched = {}
        self._side_do(self._tabacc_clear)      # flushed with the first block's side work, ahead of every table reduction
        self._xsum_done = set()
        self._bt = "top"
        if ctx.get("dense_ready") is not None:      # the dense biases were built on the side stream during the forward
            torch.cuda.current_stream().wait_event(ctx["dense_ready"])
        e, d = "encoder.", "decoder."
        # ---- seg projection (frozen, tied to seg_embed_tokens: no weight grad)
        fresh = "g_dlogits" not in self.ws or self.ws["g_dlogits"].shape != (B * Td, self.npad)
        dl = buf("g_dlogits", (B * Td, self.npad))
        if fresh or _POISON:
            dl.zero_()               # the padding columns: written once per allocation (nothing else writes them)
        dl.view(B, Td, self.npad)[:, :, : cfg.num_seg_tokens].copy_(dlogits)
        dfeat = buf("g_dfeat", (B, Td, C))
        hip.linear_dx(dl, self.wseg_pad, out=dfeat.view(B * Td, C))
        if self.train_seg:
            # tied seg projection (decoder_module.py:133-141,290-294): d seg_embed_tokens = dlogits^T feat
            dwseg = buf("g_dwseg", (self.npad, C))
            hip.linear_dw(dl, self.ws["d_feat"].view(B * Td, C), dwseg)
            G("encoder.seg_embed_tokens.weight").copy_(dwseg[: cfg.num_seg_tokens])
        y = ctx["d_y_final"]
        dy = buf("g_dy_final", (B, Td, C))
        self._ln_bwd(dfeat[:, 1:], y[:, :P], d + "layer_norm", "d_final_ln_p", dy[:, :P])
        self._ln_bwd(dfeat[:, :1], y[:, P:], d + "layer_norm", "d_final_ln_b", dy[:, P:], accumulate=True)
        dy = dy.view(B * Td, C)
        d_enc_out = buf("g_d_enc_out", (B, T, C))
        dspq, dspk = buf("g_dspq", (Td, C), torch.float32), buf("g_dspk", (Td, C), torch.float32)
        dcpq, dcpk = buf("g_dcpq", (Td, C), torch.float32), buf("g_dcpk", (T, C), torch.float32)
        fuse = lab.get("NO_LN_BWD_DROP") is None
        dbr = None
        for l in reversed(range(cfg.dec_layers)):
            p = "%slayers.%d." % (d, l)
            tg = "d%d" % l
            first = l == cfg.dec_layers - 1
            # the self block's closing pre-LN backward also produces the fc2-dropout adjoint that opens layer l-1's FFN block
            nx_f = self._next_drop("d%d" % (l - 1), B * Td) if fuse and l > 0 else None
            # (measured and dropped in round 6: each block's closing pre-LN backward together with the post-LN backward that opens
            # the next block in one launch -- +0.11 ms per step, profiles/round6_ln_bwd_pair_ab.txt)
            dy = self._ffn_bwd(tg, p, dy, B * Td, dbr_pre=dbr)
            dy = self._cross_block_bwd(tg, p, dy, B, Td, T, ctx["d_cpq"], ctx["d_cpk"], scaling, d_enc_out, first, dcpq, dcpk)
            tabn = "%sseg_rel_pos_table_list.%d.weight" % (d, l)
            dy = self._self_block_bwd(tg, p, "self_attn", "self_attn_layer_norm", "self_attn_ln", dy, B, Td,
                                      ctx["d_spq"], ctx["d_spk"], scaling, dspq, dspk, first,
                                      [(tabn, g["dec_idx2d"]), (tabn, g["dec_idx1d"]), (tabn, g["dec_idxx"])], nxt=nx_f)
            dbr = nx_f["out"] if nx_f else None
            self._layer_end(p)                       # final in side-stream order
        # ---- decoder embedding LN (input = [enc_out[:, :P] | embed(bos)])
        self.mark("dec_bwd_end")
        self._bt = "dtop"
        if self.overlap and getattr(self, "_dqs", None) is not None:      # d_enc_out was accumulated on the dQ stream
            ev = self._ev()
            ev.record(self._dqs)
            torch.cuda.current_stream().wait_event(ev)
        dy3 = dy.view(B, Td, C)
        enc_out = ctx["enc_out"]
        dyp, dyb = dy3[:, :P], dy3[:, P:]
        self._ln_bwd(dyp, enc_out[:, :P], d + "layernorm_embedding", "d_emb_ln_p", d_enc_out[:, :P],
                     dx_add=d_enc_out[:, :P], drop=self._dropargs(3))
        scratch = buf("g_bos_scratch", (B, 1, C))
        self._ln_bwd(dyb, self.ws["d_bos"], d + "layernorm_embedding", "d_emb_ln_b", scratch, accumulate=True,
                     drop=self._dropargs(4))
        if self.train_tok:      # d embed_tokens[bos] (decoder.embed_tokens is the encoder's table: share_all_embeddings)
            self._embed_grads(scratch.view(B, C), ctx["bos_ids"])
        # ---- decoder position operands: parameter gradients only, from accumulators the side stream filled -> the whole
        # section runs there, in order (no join of the main stream at the decoder -> encoder hand-over)
        pos_all = self.ws["e_pos_all"]
        dpos_all = buf("g_dpos_all", (T, C))
        self._side_do(lambda: self._dec_pos_bwd(B, P, T, Td, dspq, dspk, dcpq, dcpk, pos_all, dpos_all))
        self._side_flush()
        # ---- encoder
        dx = buf("g_dx_enc", (B * T, C))
        self._ln_bwd(d_enc_out.view(B * T, C), ctx["e_x_final"].view(B * T, C), e + "layer_norm", "e_final_ln", dx)
        depq, depk = buf("g_depq", (T, C), torch.float32), buf("g_depk", (T, C), torch.float32)
        dbr = None
        for l in reversed(range(cfg.enc_layers)):
            p = "%slayers.%d." % (e, l)
            tg = "e%d" % l
            first = l == cfg.enc_layers - 1
            if self.trunk_at == "e%d" % l:
                self._trunk_launch_point()
            nx_f = self._next_drop("e%d" % (l - 1), B * T) if fuse and l > 0 else None
            dx = self._ffn_bwd(tg, p, dx, B * T, dbr_pre=dbr)
            dx = self._self_block_bwd(tg, p, "self_attn", "self_attn_layer_norm", "attn_ln", dx, B, T, ctx["e_pq"],
                                      ctx["e_pk"], scaling, depq, depk, first,
                                      [("%simage_rel_pos_table_list.%d.weight" % (e, l), g["enc_idx2d"]),
                                       ("%stoken_rel_pos_table_list.%d.weight" % (e, l), g["enc_idx1d"]),
                                       (None, None)], nxt=nx_f)
            dbr = nx_f["out"] if nx_f else None
            self._layer_end(p)
        # ---- encoder abs-pos operands
        self._bt = "etop"
        # the main stream has run out of work: the side stream gets the whole GPU for the last weight gradients (no workgroup
        # cap) and the abs-pos half of the tail, the main stream takes the embedding half, which only reads its own dx.
        # (Round 6 measured the last layer's fc1 / fc2 weight gradients started under its attention backward and its bias-gradient
        # kernels on the main stream at the end: the wait for the side stream shrank from 0.44 to 0.17 ms and the step grew by
        # 0.04 ms -- profiles/round6_tail_ab.txt: the exposed end is resource-time, not slack.)
        if "tailsplit" in _EXP_SKIP:     # (measurement: the whole tail on the side stream, capped grid)
            self._side_do(lambda: (self._dw_flush(), self._enc_tail_pos_bwd(B, L, P, T, h, w, depq, depk, pos_all, dpos_all),
                                   self._enc_tail_emb_bwd(B, L, P, T, dx)))
        else:
            self._side_do(lambda: (self._dw_flush(wgs=0), self._enc_tail_pos_bwd(B, L, P, T, h, w, depq, depk, pos_all, dpos_all)))
            self._side_flush()
            self._enc_tail_emb_bwd(B, L, P, T, dx)
        if self.trunk_at == "end":
            self._trunk_launch_point()
        elif self.trunk_at == "fwd" and self._pf_request is not None and not torch.cuda.is_current_stream_capturing():
            req, self._pf_request = self._pf_request, None           # ("fwd1": passes start at a forward only -- measurement)
            self._prefetch_request(req, at_end=True)
        if self.drain_timing is not None:       # (measurement: how long the main stream really waits for the side queue here)
            t0 = torch.cuda.Event(enable_timing=True); t0.record()
        self.mark("bwd_main_end")
        self._join_side()            # the optimizer (main stream) reads the whole gradient arena next
        self.mark("joined")
        if self.drain_timing is not None:
            t1 = torch.cuda.Event(enable_timing=True); t1.record()
            self.drain_timing.append((t0, t1))
        self._notify(e)              # both halves of the tail are in: the last gradient slice may be reduced (main stream)
        return self.g16

    def _ffn_ln_coefs(self):
        """per layer: row sums of fc2.weight against ffn_layernorm's gamma / beta (+ fc2.bias) -> ws[tag + "_fcoef"] fp32 [2, C]
        (what `ifseg_ffn_ln_rowstats` dots the fc2 output gradient with): all layers in one launch"""
        cfg = self.cfg
        w2, gam, bet, b2, coef = [], [], [], [], []
        for kind, n in (("e", cfg.enc_layers), ("d", cfg.dec_layers)):
            for l in range(n):
                p = "%s.layers.%d." % ("encoder" if kind == "e" else "decoder", l)
                w2.append(self.W(p + "fc2.weight")); gam.append(self.Wf(p + "ffn_layernorm.weight"))
                bet.append(self.Wf(p + "ffn_layernorm.bias")); b2.append(self.W(p + "fc2.bias"))
                coef.append(self.buf("%s%d_fcoef" % (kind, l), (2, cfg.embed_dim), torch.float32))
        for i in range(0, len(w2), 32):
            hip.ffn_ln_coef(w2[i:i + 32], gam[i:i + 32], bet[i:i + 32], b2[i:i + 32], coef[i:i + 32])

    def _prefetch_request(self, req, at_end=False):
        """`req`: the image tensors of the following forward calls, in order (or one tensor).  At the start of a forward
        (`at_end` False) a pass over the first `trunk_lookahead` of them is started unless the next batch's features are
        already there; returns True if the request should be looked at again at the end of this step's backward.  There
        (`at_end`), with at most ONE batch left in the cache, the pass over the batches behind it starts: its convolutions
        then run under the final join, the gradient norm and the HBM-bound Adam, where nothing else needs the MFMAs, and
        under the next forward."""
        if not isinstance(req, (list, tuple)):
            if not at_end:
                self.prefetch_trunk(req)
            return False
        cache = ([self._pf] if self._pf is not None else []) + (self._pf_more if self._pf is not None else [])
        covered = 0
        while covered < len(cache) and covered < len(req) and cache[covered]["key"] == self._tkey(req[covered]):
            covered += 1
        if covered < len(cache):                            # the cache is not a prefix of the request: start over
            covered = 0
            self._pf, self._pf_more = None, []
        if not at_end:
            if covered == 0:
                if req:
                    self.prefetch_trunk(list(req[: self.trunk_lookahead]))
                return False
            return self.trunk_lookahead > 1
        if covered <= 1 and len(req) > covered and self.trunk_lookahead > 1:
            self.prefetch_trunk(list(req[covered: covered + self.trunk_lookahead]), append=covered > 0)
        return False

    def _trunk_launch_point(self):
        """the frozen trunk of the NEXT batch, launched from inside this step's backward: its ~90 convolutions then run
        under the tail of the backward, the final join and the HBM-bound clip + Adam instead of under the next forward"""
        if self._pf_request is not None and not torch.cuda.is_current_stream_capturing():
            req, self._pf_request = self._pf_request, None
            self._prefetch_request(req)

    def _enc_tail_pos_bwd(self, B, L, P, T, h, w, depq, depk, pos_all, dpos_all):
        """encoder abs-pos operands (parameter gradients only; inputs accumulated on the side stream) -- side stream"""
        cfg = self.cfg
        C = cfg.embed_dim
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        e = "encoder."
        depqk = buf("g_depqk", (T, 2 * C))
        tmp = buf("g_tmp_ec", (T, C))
        hip.cast_f32_bf16(depq, tmp); depqk[:, :C].copy_(tmp)
        hip.cast_f32_bf16(depk, tmp); depqk[:, C:].copy_(tmp)
        self._linear_bwd(depqk, pos_all, self._fused(self.p16, e + "pos_q_linear.weight", 2 * C, C),
                         self._fused(self.g16, e + "pos_q_linear.weight", 2 * C, C),
                         self._fused(self.g16, e + "pos_q_linear.bias", 2 * C), dx_out=dpos_all, dx_accumulate=True)
        bsz = cfg.image_bucket_size
        oh = cfg.orig_patch_image_size // 16
        if self.ctx.get("resized") and P > oh * oh:
            dipos = buf("g_dipos", (P, C))
            self._ln_bwd(dpos_all[:P], self.ws["e_ipos_resized"], e + "image_pos_ln", "ipos_ln", dipos)
            G(e + "embed_image_positions.weight")[1:1 + bsz * oh].view(oh, bsz, C)[:, :oh].copy_(
                R.rows_resize_adjoint(dipos, (oh, oh), (h, w)).view(oh, oh, C))
        else:
            ipos_view = W(e + "embed_image_positions.weight")[1:1 + bsz * h].view(h, bsz, C)[:, :w]
            ipos_grad = G(e + "embed_image_positions.weight")[1:1 + bsz * h].view(h, bsz, C)[:, :w]
            self._ln_bwd(dpos_all[:P].view(h, w, C), ipos_view, e + "image_pos_ln", "ipos_ln", ipos_grad)
        self._ln_bwd(dpos_all[P:], W(e + "embed_positions.weight")[:L], e + "pos_ln", "tpos_ln",
                     G(e + "embed_positions.weight")[:L])
        self._dw_flush()                 # the LayerNorm partials of this half
        assert not self._ln_red_tasks and not self._ln_red_acc and not self._dw_tasks

    def _enc_tail_emb_bwd(self, B, L, P, T, dx):
        """encoder embedding LayerNorms / type embedding (embed_tokens / image_proj / ResNet frozen -> stop here): reads only
        the main stream's dx -- main stream, next to the side stream's last weight gradients"""
        cfg = self.cfg
        C = cfg.embed_dim
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        e = "encoder."
        dx3 = dx.view(B, T, C)
        dxi, dxt = dx3[:, :P], dx3[:, P:]
        if self.ctx.get("nonpad") is not None:
            # adjoint of x * (1 - encoder_padding_mask): padded rows pass no gradient into the embeddings (they hold exact
            # zeros already -- nothing downstream reads a padded position -- this keeps it so by construction)
            dxt.mul_(self.ctx["nonpad"].unsqueeze(-1).to(dxt.dtype))
        dimg = buf("g_dimg_pre", (B, P, C))
        self._ln_bwd(dxi, self.ws["img_pre"].view(B, P, C), e + "patch_layernorm_embedding", "img_ln", dimg,
                     drop=self._dropargs(1))
        dtok = buf("g_dtok_pre", (B, L, C))
        self._ln_bwd(dxt, self.ws["tok_pre"].view(B, L, C), e + "layernorm_embedding", "tok_ln", dtok,
                     drop=self._dropargs(2))
        gt = G(e + "type_embedding.weight")
        if self.train_tok:
            self._embed_grads(dtok.view(B * L, C), self.ctx["src_ids"])
        self._bias_grad(dtok.view(B * L, C), gt[0])
        self._bias_grad(dimg.view(B * P, C), gt[1])
        self._dw_flush()                 # the LayerNorm partials of this half
        assert not self._ln_red_tasks and not self._ln_red_acc and not self._dw_tasks

    def _embed_grads(self, drows, ids):
        """nn.Embedding backward into the token table's gradient (padding_idx row excluded, as F.embedding does): rows sorted
        by token id (torch.sort, stable, no host sync), runs of equal ids summed in order by one workgroup each"""
        sid, perm = torch.sort(ids.reshape(-1), stable=True)
        hip.rows_segment_sum(drows, perm, None, sid, self.G("encoder.embed_tokens.weight"), skip_id=1)

    def _dec_pos_bwd(self, B, P, T, Td, dspq, dspk, dcpq, dcpk, pos_all, dpos_all):
        """gradients of the decoder's position operands (self / cross abs-pos projections, seg positions) -- side stream"""
        cfg = self.cfg
        C = cfg.embed_dim
        W, Wf, G, buf = self.W, self.Wf, self.G, self.buf
        d = "decoder."
        dspqk = buf("g_dspqk", (Td, 2 * C))
        hip.cast_f32_bf16(dspq, buf("g_tmp_tc", (Td, C)))
        dspqk[:, :C].copy_(self.ws["g_tmp_tc"])
        hip.cast_f32_bf16(dspk, self.ws["g_tmp_tc"])
        dspqk[:, C:].copy_(self.ws["g_tmp_tc"])
        tp = self.ws["d_tp"]
        dtp = buf("g_dtp", (Td, C))
        self._linear_bwd(dspqk, tp, self._fused(self.p16, d + "self_pos_q_linear.weight", 2 * C, C),
                         self._fused(self.g16, d + "self_pos_q_linear.weight", 2 * C, C),
                         self._fused(self.g16, d + "self_pos_q_linear.bias", 2 * C), dx_out=dtp)
        dcpq16 = buf("g_dcpq16", (Td, C))
        hip.cast_f32_bf16(dcpq, dcpq16)
        self._linear_bwd(dcpq16, tp, W(d + "cross_pos_q_linear.weight"), G(d + "cross_pos_q_linear.weight"),
                         G(d + "cross_pos_q_linear.bias"), dx_out=dtp, dx_accumulate=True)
        segG = G(d + "embed_seg_positions.weight")
        segtab = W(d + "embed_seg_positions.weight")
        if self.ctx.get("resized"):
            # the LayerNorm saw the resized rows: its backward, then the adjoint of the bilinear resize into the sb x sb grid's rows
            sb = cfg.seg_bucket_size
            tgt, dtgt = self.ws["d_spos_resized"], buf("g_dtgt", (Td, C))
            self._ln_bwd(dtp[:P], tgt[:P], d + "seg_pos_ln", "d_tp_ln_p", dtgt[:P])
            self._ln_bwd(dtp[P:], tgt[P:], d + "seg_pos_ln", "d_tp_ln_b", dtgt[P:], accumulate=True)
            segG[1:1 + sb * sb].copy_(R.rows_resize_adjoint(dtgt[:P], (sb, sb), (self.ctx["h"], self.ctx["w"])))
            segG[:1].copy_(dtgt[P:])
        else:
            self._ln_bwd(dtp[:P], segtab[1:1 + P], d + "seg_pos_ln", "d_tp_ln_p", segG[1:1 + P])
            self._ln_bwd(dtp[P:], segtab[:1], d + "seg_pos_ln", "d_tp_ln_b", segG[:1], accumulate=True)
        dcpk16 = buf("g_dcpk16", (T, C))
        hip.cast_f32_bf16(dcpk, dcpk16)
        self._linear_bwd(dcpk16, pos_all, W(d + "cross_pos_k_linear.weight"), G(d + "cross_pos_k_linear.weight"),
                         G(d + "cross_pos_k_linear.bias"), dx_out=dpos_all)
        self._dw_flush()                 # the LayerNorm partials queued since the last layer
        self._side_do(lambda: self._notify(d))

    def _dw_flush(self, wgs=None):
        tasks, self._dw_tasks = self._dw_tasks, []
        if tasks and "dw" not in _EXP_SKIP:
            hip.linear_dw_group(tasks, wgs)
        # ffn_layernorm's dgamma / dbeta from fc2's (now final) weight / bias gradient
        # key projections: the weight gradient without (spurious column sum of dK) x (token-common component of the input)
        kf, self._kfix_tasks = self._kfix_tasks, []
        for gw, gb, x, tag in kf:
            key = "g_xmean_" + tag
            if key not in self._xsum_done:       # (the decoder's cross-attention key projections all read the encoder output)
                # (sum_j dK_j = 0 makes dK^T (x - 1 c^T) = dK^T x for EVERY c: the mean over every 8th row removes the
                # token-common component just as well as the mean over all of them, at an eighth of the traffic)
                hip.col_mean(x, self.buf(key, (hip.COLSUM_BLOCKS + 1, x.shape[-1]), torch.float32), every=8)
                self._xsum_done.add(key)
            hip.kproj_common_mode(gw, gb, self.ws[key][hip.COLSUM_BLOCKS])
        pg, self._ffn_pg_tasks = self._ffn_pg_tasks, []
        for w2, dw2, db2, gam, bet, dgam, dbet, dy_, u_, mu_, rs_ in pg:
            hip.ffn_ln_param_grads(w2, dw2, db2, gam, bet, dgam, dbet, dy=dy_, u=u_, mean=mu_, rstd=rs_)
        # the LayerNorm dgamma / dbeta partials collected since the last flush: one reduction launch
        for attr in ("_ln_red_tasks", "_ln_red_acc"):
            red = getattr(self, attr)
            setattr(self, attr, [])
            if red:
                hip.reduce_parts_multi(red)

    def _layer_end(self, p):
        """(end of a layer's backward) the layer's weight-gradient group, its followers, the table casts and the gradient-ready
        notification, in side-stream order.  (Round 6 measured holding the group back until a point inside the NEXT layer's
        backward -- beside the attention backward instead of beside the LayerNorm pair: the LayerNorms got 15 us faster each and
        the dK|dV kernel 70 us slower, +0.3 ms per step; profiles/round6_dw_release_ab.txt.  And the group on the MAIN queue,
        alone on the chip: +0.8 ms; profiles/round6_dw_small_experiments.txt.)"""
        self._side_do(lambda: (self._flush_tables(), self._notify(p)))
        self._side_flush()

    def _flush_tables(self):
        self._dw_flush()
        for key, tabname in self._tab_touched.items():
            hip.cast_f32_bf16(self.ws[key].view(-1), self.G(tabname).view(-1))
        self._tab_touched = {}

    def _notify(self, prefix):
        if self.grad_ready_hook is not None:
            self.grad_ready_hook(prefix)
