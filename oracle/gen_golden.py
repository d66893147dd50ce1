"""Pin the oracle against the reference and write the golden vectors.

TEST INFRASTRUCTURE -- BUILD CONTAINER ONLY (needs /root/reference).

    python oracle/gen_golden.py [--skip-base]

For each case the *reference's own* ``models/segofa`` code (imported through
``oracle/_refshim.py``) and the restatement ``oracle/segofa_ref.py`` are given
the same procedural weights (``procedural_state_dict``) and the same seeded
inputs (``synthetic_batch``); the script asserts they agree to fp32 round-off
and then stores the REFERENCE's outputs as the golden vectors:

  tests/golden/fixture_train.npz   small config, B=2, causal + full-context
                                   logits, bilinear-upsampled CE loss (computed by
                                   the reference's SegCriterion.compute_loss),
                                   area histograms, selected parameter grads
  tests/golden/fixture_resize.npz  small config, 128x192 image (P=96 > orig 64):
                                   the double-bilinear rel-pos resize path, eval
  tests/golden/fixture_imfree.npz  small widths on the full 32x32 patch grid: the image-free branch
                                   (aux_input, EmbeddingBag patches, causal decoder) + the
                                   reference's compute_imfree_loss, selected grads
  tests/golden/base_c1.npz         SegOFA-Base, B=1, 512x512, nseg 15, L=36
                                   (BASELINE config 1 shapes): logits, loss,
                                   grad norms

Only data (inputs are regenerated from seeds; outputs are arrays) is written.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refshim  # noqa: E402
import segofa_ref as O  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

GRAD_KEYS = [
    "encoder.layers.0.self_attn.q_proj.weight",
    "encoder.layers.0.self_attn.c_attn",
    "encoder.layers.1.fc1.bias",
    "encoder.image_rel_pos_table_list.0.weight",
    "encoder.token_rel_pos_table_list.1.weight",
    "encoder.pos_q_linear.weight",
    "encoder.type_embedding.weight",
    "encoder.layers.0.ffn_layernorm.weight",
    "decoder.seg_rel_pos_table_list.0.weight",
    "decoder.cross_pos_k_linear.weight",
    "decoder.layers.1.encoder_attn.v_proj.weight",
    "decoder.layers.0.self_attn_ln.bias",
    "decoder.layer_norm.weight",
    "decoder.embed_seg_positions.weight",
]
# the tensors whose gradients pass through the resizes of a non-trained grid (case resize_train)
RESIZE_KEYS = ["encoder.embed_image_positions.weight", "encoder.image_rel_pos_table_list.1.weight",
               "decoder.seg_rel_pos_table_list.1.weight", "encoder.embed_positions.weight"]



def build_reference(cfg, arch, overrides=None):
    model, args = _refshim.build_reference_model(
        arch, cfg.num_seg_tokens, cfg.vocab_size - 1, cfg.patch_image_size,
        cfg.orig_patch_image_size, overrides)
    sd = O.procedural_state_dict(cfg, seed=0)
    ref_keys = set(model.state_dict().keys())
    derived = {k for k in ref_keys if k.endswith(("_rp_bucket", ".version", "image_position_idx",
                                                  "bin_id_offset", "seg_id_offset", "region_prefix"))}
    assert ref_keys - derived == set(sd.keys()), (
        sorted(ref_keys - derived - set(sd.keys()))[:10], sorted(set(sd.keys()) - (ref_keys - derived))[:10])
    for k, v in model.state_dict().items():
        if k in sd:
            assert tuple(v.shape) == tuple(sd[k].shape), (k, v.shape, sd[k].shape)
    missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
    assert not unexpected and set(missing) <= derived, (missing, unexpected)
    # bucket tables: the restatement's generators must reproduce the reference's buffers
    rsd = model.state_dict()
    n_img = (2 * cfg.image_bucket_size - 1) ** 2 + 3
    assert torch.equal(rsd["encoder.image_rp_bucket"], O.make_image_bucket_position(cfg.image_bucket_size, n_img))
    assert torch.equal(rsd["encoder.token_rp_bucket"], O.make_token_bucket_position(cfg.token_bucket_size))
    sb = cfg.seg_bucket_size
    assert torch.equal(rsd["decoder.seg_rp_bucket"], O.make_image_bucket_position(sb, (2 * sb - 1) ** 2 + 3))
    return model, sd


def build_criterion(cfg):
    from criterions.seg_criterion import SegCriterion

    class _Cfg:
        num_seg_tokens = cfg.num_seg_tokens
        category_list = ",".join("c%d" % i for i in range(cfg.num_seg_tokens))

    d = _refshim.FakeDictionary(cfg.vocab_size - 1, cfg.num_seg_tokens)

    class _Task:
        cfg = _Cfg()
        target_dictionary = d
        tgt_dict = d

    return SegCriterion(_Task(), sentence_avg=False, label_smoothing=0.0,
                        unsupervised_segmentation="false", init_seg_with_text="false")


def ref_forward(model, batch, full_ctx):
    B, L = batch["src_tokens"].shape
    return model(src_tokens=batch["src_tokens"], src_lengths=torch.full((B,), L),
                 prev_output_tokens=batch["prev_output_tokens"], patch_images=batch["patch_images"],
                 patch_masks=batch["patch_masks"], full_context_alignment=full_ctx)


def oracle_train(cfg, sd, batch, grad_keys):
    sdg = {k: v.clone() for k, v in sd.items()}
    # restore aliases
    for name, (_, kind) in O.state_dict_spec(cfg).items():
        if kind.startswith("alias:"):
            sdg[name] = sdg[kind[6:]]
    for k in grad_keys:
        sdg[k].requires_grad_(True)
    logits, extra = O.segofa_forward(sdg, cfg, batch["src_tokens"], batch["patch_images"],
                                     batch["prev_output_tokens"], batch["patch_masks"], False)
    hp, wp = extra["encoder_returns"]["image_embed_shape"]
    h, w = batch["patch_images"].shape[-2:]
    loss, s, t = O.seg_loss(cfg, logits, batch["target"], hp, wp, h, w)
    loss.backward()
    return logits.detach(), loss.detach(), {k: sdg[k].grad for k in grad_keys}, O.seg_metric(s.detach(), t, cfg.num_seg_tokens)


SUB_N = 4096


def sub_index(name, numel, n=SUB_N):
    """positions of the seeded subsample of a gradient tensor stored as `gsub:<name>` (all of it when numel <= n);
    tests/test_configs_gpu.py rebuilds the same indices from the name"""
    if numel <= n:
        return torch.arange(numel)
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return torch.randperm(numel, generator=g)[:n].sort().values


def case_train(cfg, arch, overrides, batch_size, src_len, out_name, grad_keys, full_grads=True, diversify=False, sub_all=False,
               bf16_weights=False, pad_tail=None, image_hw=None):
    """bf16_weights: the procedural weights are rounded to bf16-representable values on both sides (identical weights);
    diversify: the (frozen, tied) seg projection is replaced by O.diversify_seg_projection(...) (measured in round 3: it
    removes the dominant common component of the logits, so the relative error of the HIP path grows by the same factor
    as the margins -- not used for the committed goldens); sub_all: every parameter the reference gives a gradient
    contributes a seeded SUB_N-element subsample of it (`gsub:<name>`)."""
    t0 = time.time()
    model, sd = build_reference(cfg, arch, overrides)
    crit = build_criterion(cfg)
    # image_hw: an image whose feature grid differs from the trained one -- TRAINING through the bilinear resizes of the position
    # rows and of every layer's relative-position bias (encoder_module.py:356-372,798-809; decoder_module.py:541-550,603-627)
    batch = O.synthetic_batch(cfg, batch_size, src_len, image_hw=image_hw) if image_hw else O.synthetic_batch(cfg, batch_size, src_len)
    if pad_tail is not None:
        # padded prompts (collate pads shorter prompts on the right, data/mm_data/segmentation_dataset.py collate ->
        # data_utils.collate_tokens): sample b ends with pad_tail[b] pad tokens, eos in front of them; the reference then
        # builds encoder_padding_mask (encoder_module.py:730-752) and masks those keys in the encoder self-attention and the
        # decoder cross-attention (unify_multihead_attention.py:477-489)
        for b, n in enumerate(pad_tail):
            if n:
                batch["src_tokens"][b, -n:] = O.PAD
                batch["src_tokens"][b, -n - 1] = O.EOS
    if bf16_weights:      # identical weight values on both sides of the parity test (O.round_weights_bf16)
        sd = O.round_weights_bf16(sd)
    if diversify:
        sd = O.diversify_seg_projection(sd, cfg, batch)
    if bf16_weights or diversify:
        missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
        assert not unexpected
    model.train()
    params = dict(model.named_parameters())
    for k in grad_keys:
        params[k].requires_grad_(True)
    out = ref_forward(model, batch, False)
    sample = {"target": batch["target"], "net_input": {"patch_images": batch["patch_images"]}}
    loss, metrics, _ = crit.compute_loss(model, out, sample, 0)
    loss.backward()
    ref_logits = out[0].detach()
    with torch.no_grad():
        ref_full = ref_forward(model, batch, True)[0]
    print("[%s] reference done %.1fs loss=%.9f" % (out_name, time.time() - t0, loss.item()))

    o_logits, o_loss, o_grads, o_metric = oracle_train(cfg, sd, batch, grad_keys)
    with torch.no_grad():
        o_full = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None,
                                  batch["patch_masks"], True)[0]
    err = (o_logits - ref_logits).abs().max().item()
    errf = (o_full - ref_full).abs().max().item()
    print("  oracle vs reference: logits max-abs %.3e (full-ctx %.3e)  loss diff %.3e"
          % (err, errf, abs(o_loss.item() - loss.item())))
    assert err <= 2e-5 and errf <= 2e-5, "oracle != reference"
    assert abs(o_loss.item() - loss.item()) <= 2e-6
    save = {"logits_causal": ref_logits.numpy(), "logits_full": ref_full.numpy(),
            "loss": np.float64(loss.item()),
            "batch_size": batch_size, "src_len": src_len, "diversified": int(diversify), "bf16_weights": int(bf16_weights)}
    if image_hw:
        save["image_hw"] = np.array(image_hw)
    if pad_tail is not None:
        save["src_tokens"] = batch["src_tokens"].numpy()
    if sub_all:
        nsub = 0
        for k, p_ in params.items():
            if p_.grad is None or "embed_images" in k or float(p_.grad.abs().sum()) == 0.0:
                continue
            save["gsub:" + k] = p_.grad.reshape(-1)[sub_index(k, p_.grad.numel())].numpy().astype(np.float32)
            nsub += 1
        print("  gradient subsamples of %d tensors" % nsub)
    for name in ("area_intersect", "area_pred_label", "area_label", "area_union"):
        save[name] = metrics[name].numpy()
    for a, b in zip(o_metric, (metrics["area_intersect"], metrics["area_pred_label"],
                               metrics["area_label"], metrics["area_union"])):
        assert torch.equal(a, b), "metric histograms differ"
    for k in grad_keys:
        g_ref = params[k].grad
        g_o = o_grads[k]
        rel = ((g_o - g_ref).norm() / (g_ref.norm() + 1e-30)).item()
        print("  grad %-52s rel-L2 %.2e  |g|=%.4e" % (k, rel, g_ref.norm().item()))
        assert rel <= 2e-5, k
        save["gradnorm:" + k] = np.float64(g_ref.norm().item())
        if full_grads or g_ref.numel() <= 16384:
            save["grad:" + k] = g_ref.numpy()
    np.savez_compressed(os.path.join(GOLDEN, out_name), **save)
    print("  wrote", out_name, "%.1fs" % (time.time() - t0))


def case_resize(cfg, arch, overrides, out_name):
    model, sd = build_reference(cfg, arch, overrides)
    model.eval()
    batch = O.synthetic_batch(cfg, 1, 12, image_hw=(128, 192), seed=4321)
    with torch.no_grad():
        ref = ref_forward(model, batch, False)[0]
        ora = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], False)[0]
    err = (ora - ref).abs().max().item()
    print("[%s] oracle vs reference logits max-abs %.3e  shape %s" % (out_name, err, tuple(ref.shape)))
    assert err <= 2e-5
    np.savez_compressed(os.path.join(GOLDEN, out_name), logits_causal=ref.numpy(), image_hw=np.array([128, 192]),
                        seed=4321, src_len=12)


def case_imfree(cfg, arch, overrides, batch_size, src_len, out_name, grad_keys, pad_tail=None):
    """Image-free branch: SegOFAModel.forward(aux_input=...) + SegCriterion.compute_imfree_loss
    of the reference vs the restatement.  The reference's loss hard-codes a 32x32 -> 512x512 upsample
    (seg_criterion.py:236,249), so this case runs the small-width fixture on the full 32x32 patch grid."""
    t0 = time.time()
    model, sd = build_reference(cfg, arch, overrides)
    crit = build_criterion(cfg)
    batch = O.synthetic_aux_batch(cfg, batch_size, src_len)
    if pad_tail is not None:       # prompts of different lengths (right-padded) in the image-free step: the same encoder_padding_mask path
        for b, n in enumerate(pad_tail):
            if n:
                batch["aux_input"]["src_tokens"][b, -n:] = O.PAD
                batch["aux_input"]["src_tokens"][b, -n - 1] = O.EOS
    model.train()
    params = dict(model.named_parameters())
    for k in grad_keys:
        params[k].requires_grad_(True)
    x, extra = model(aux_input=batch["aux_input"])
    assert x is None
    aux_out = extra["aux_output"]
    loss = crit.compute_imfree_loss(model, aux_out, {"text2seg_target": batch["text2seg_target"]}, 0)
    loss.backward()
    ref_logits = aux_out[0].detach()
    print("[%s] reference done %.1fs loss=%.9f" % (out_name, time.time() - t0, loss.item()))
    sdg = {k: v.clone() for k, v in sd.items()}
    for name, (_, kind) in O.state_dict_spec(cfg).items():
        if kind.startswith("alias:"):
            sdg[name] = sdg[kind[6:]]
    for k in grad_keys:
        sdg[k].requires_grad_(True)
    o_logits, _ = O.segofa_forward_imfree(sdg, cfg, batch["aux_input"])
    o_loss = O.imfree_loss(cfg, o_logits, batch["text2seg_target"])
    o_loss.backward()
    err = (o_logits.detach() - ref_logits).abs().max().item()
    print("  oracle vs reference: logits max-abs %.3e  loss diff %.3e" % (err, abs(o_loss.item() - loss.item())))
    assert err <= 2e-5 and abs(o_loss.item() - loss.item()) <= 2e-6
    save = {"logits": ref_logits.numpy(), "loss": np.float64(loss.item()), "batch_size": batch_size, "src_len": src_len}
    if pad_tail is not None:
        save["src_tokens"] = batch["aux_input"]["src_tokens"].numpy()
    for k in grad_keys:
        g_ref, g_o = params[k].grad, sdg[k].grad
        rel = ((g_o - g_ref).norm() / (g_ref.norm() + 1e-30)).item()
        print("  grad %-52s rel-L2 %.2e  |g|=%.4e" % (k, rel, g_ref.norm().item()))
        assert rel <= 2e-5, k
        save["gradnorm:" + k] = np.float64(g_ref.norm().item())
        if g_ref.numel() <= 65536:
            save["grad:" + k] = g_ref.numpy()
    np.savez_compressed(os.path.join(GOLDEN, out_name), **save)
    print("  wrote", out_name, "%.1fs" % (time.time() - t0))


def case_eval(cfg, arch, overrides, out_name, ori_hw=(150, 200), iters=25, topk=3):
    """Eval branch of the criterion (BASELINE config 5 / SURVEY 8f row 4): top-k neighbour smoothing of the class
    probabilities on the trunk features (seg_criterion.py:197-213) and metrics at the ORIGINAL image resolution
    (:289-347), reference vs restatement."""
    model, sd = build_reference(cfg, arch, overrides)
    model.eval()
    crit = build_criterion(cfg)
    crit.resnet_iters, crit.resnet_topk, crit.resnet_prob_temperature = iters, topk, 1.0
    batch = O.synthetic_batch(cfg, 1, 12, seed=777)
    sd = O.diversify_seg_projection(sd, cfg, batch)          # predictions that differ from patch to patch
    missing, unexpected = torch.nn.Module.load_state_dict(model, sd, strict=False)
    assert not unexpected
    g = torch.Generator().manual_seed(778)
    ori = torch.randint(0, cfg.num_seg_tokens + 1, ori_hw, generator=g)      # num_seg = ignore label
    sample = {"net_input": {"src_tokens": batch["src_tokens"], "src_lengths": torch.full((1,), 12),
                            "patch_images": batch["patch_images"], "patch_masks": batch["patch_masks"],
                            "prev_output_tokens": batch["prev_output_tokens"]},
              "target": batch["target"], "ori_semantic_seg": [ori.numpy()], "ori_shape": [(ori_hw[0], ori_hw[1], 3)],
              "ntokens": 1, "nsentences": 1}
    import unittest.mock as mock
    with torch.no_grad(), mock.patch("torch.zeros", wraps=torch.zeros) as z:
        # the reference builds `imfree_loss = torch.zeros(..., device='cuda')` in eval (:215): run it on the CPU
        z.side_effect = lambda *a, **k: torch.ones(()).new_zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
        loss, ss, logs = crit(model, sample)
        out = ref_forward(model, batch, False)
    logits, extra = out
    feat = extra["encoder_returns"]["image_embed_before_proj"][0]
    hp, wp = extra["encoder_returns"]["image_embed_shape"][0]
    with torch.no_grad():
        o_logits, o_extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        o_prob = O.neighbour_smoothing(o_logits, o_extra["encoder_returns"]["image_embed_before_proj"], iters, topk)
        o_loss, o_hist = O.seg_eval(cfg, o_logits, ori, hp, wp)
        _, o_hist_pp = O.seg_eval(cfg, o_prob, ori, hp, wp)
    assert (o_logits - logits).abs().max().item() <= 2e-5
    assert abs(o_loss.item() - loss.item()) <= 2e-6, (o_loss.item(), loss.item())
    names = ("area_intersect", "area_pred_label", "area_label", "area_union")
    save = {"loss": np.float64(loss.item()), "ori": ori.numpy(), "logits": logits.numpy(), "prob_smooth": o_prob.numpy(),
            "iters": iters, "topk": topk}
    for nme, a in zip(names, o_hist):
        assert torch.equal(a, logs[nme]), nme
        save[nme] = logs[nme].numpy()
    for nme, a in zip(names, o_hist_pp):
        assert torch.equal(a, logs[nme + "_resnet_postprocess"]), nme
        save[nme + "_pp"] = logs[nme + "_resnet_postprocess"].numpy()
    print("[%s] eval loss %.6f; hist and smoothed-hist equal to the reference" % (out_name, loss.item()))
    np.savez_compressed(os.path.join(GOLDEN, out_name), **save)


def case_optim(cfg, arch, overrides, out_name, n_updates=3, lr=5e-5, total_updates=2000):
    """Optimizer + harness golden: the REFERENCE's FairseqAdam (fairseq/optim/adam.py:
    45-240, decoupled weight decay 0.1), `multiply_grads`, `clip_grad_norm(1.0)` (fairseq_optimizer.py:105-113) and
    its cosine schedule driven exactly like train.py / trainer.py drive them: `reinit(total, 0)` (train.py:184), then
    `begin_epoch` (train.py:304 -> trainer.py:717-721,1126-1134) steps the scheduler once with num_updates = 0 BEFORE
    the first update -- period is already `total`, so update 1 runs at cosine(0) = the peak lr and update k at
    cosine(k-1) -- then `step_update(k)` after every update (trainer.py:962), for `n_updates` updates on one fixed batch.  Stores losses, grad norms, the lr of every update and the
    post-update parameters."""
    _refshim.install()
    from fairseq.optim.adam import FairseqAdam, FairseqAdamConfig
    from fairseq.optim.lr_scheduler.cosine_lr_scheduler import CosineLRSchedule, CosineLRScheduleConfig
    model, sd = build_reference(cfg, arch, overrides)
    crit = build_criterion(cfg)
    batch = O.synthetic_batch(cfg, 2, 12)
    params = [p for p in model.parameters() if p.requires_grad]
    ocfg = FairseqAdamConfig(adam_betas="(0.9,0.999)", adam_eps=1e-8, weight_decay=0.1)
    ocfg.lr, ocfg.tpu = [lr], False
    opt = FairseqAdam(ocfg, params)
    scfg = CosineLRScheduleConfig()
    scfg.lr, scfg.max_update = [lr], total_updates
    sched = CosineLRSchedule(scfg, opt)
    sched.reinit(total_updates, 0)
    sched.step_begin_epoch(1)
    sched.step_update(0)                                          # trainer.begin_epoch -> lr_step_begin_epoch -> lr_step_update
    model.train()
    save = {"n_updates": n_updates, "lr0": lr, "total_updates": total_updates}
    losses, gnorms, lrs = [], [], []
    sample = {"target": batch["target"], "net_input": {"patch_images": batch["patch_images"]}}
    for k in range(n_updates):
        opt.zero_grad()
        out = ref_forward(model, batch, False)
        loss, metrics, ntok = crit.compute_loss(model, out, sample, k)
        opt.backward(loss)
        opt.multiply_grads(1.0 / float(ntok))                     # world 1, sample_size = ntokens = 1
        gn = opt.clip_grad_norm(1.0)
        lrs.append(opt.get_lr())
        opt.step()
        sched.step_update(k + 1)
        losses.append(loss.item()); gnorms.append(float(gn))
        print("[%s] update %d: loss %.6f |g| %.5f lr %.4e" % (out_name, k + 1, losses[-1], gnorms[-1], lrs[-1]))
    named = dict(model.named_parameters())
    for k in GRAD_KEYS + ["encoder.layers.1.fc2.weight", "decoder.layers.0.encoder_attn.out_proj.bias"]:
        if named[k].requires_grad:
            save["param:" + k] = named[k].detach().numpy().copy()
            save["init:" + k] = sd[k].numpy()
    save.update(losses=np.array(losses), gnorms=np.array(gnorms), lrs=np.array(lrs))
    np.savez_compressed(os.path.join(GOLDEN, out_name), **save)
    print("  wrote", out_name)


def case_upgrade(cfg, arch, overrides, out_name):
    """Checkpoint up-conversion (models/segofa/segofa.py:197-299, encoder_module.py:943-987, decoder_module.py:
    892-940): a procedurally filled "OFA" checkpoint (token table one row short, no seg-token tables, a stale
    `decoder.output_projection`, image-position tables 40 rows short) goes through the reference's
    `upgrade_state_dict`; the keys, shapes and the rows it appended (torch.manual_seed(7)) are the golden."""
    model, sd = build_reference(cfg, arch, overrides)
    ck = {k: v.clone() for k, v in sd.items() if "seg_embed_tokens" not in k and "seg_projection" not in k
          and "embed_tokens_bag" not in k}
    for k in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight"):
        ck[k] = ck[k][:-1].clone()
    ck["decoder.output_projection.weight"] = ck["decoder.embed_tokens.weight"].clone()
    for k in ("encoder.embed_image_positions.weight", "decoder.embed_image_positions.weight"):
        ck[k] = ck[k][:-40].clone()
    torch.manual_seed(7)
    model.upgrade_state_dict(ck)
    missing, unexpected = torch.nn.Module.load_state_dict(model, ck, strict=True)
    save = {"keys": np.array(sorted(ck.keys())), "shapes": np.array([str(tuple(ck[k].shape)) for k in sorted(ck.keys())])}
    save["enc_tok_tail"] = ck["encoder.embed_tokens.weight"][-2:].numpy()
    save["dec_tok_tail"] = ck["decoder.embed_tokens.weight"][-2:].numpy()
    save["enc_ipos_tail"] = ck["encoder.embed_image_positions.weight"][-41:].numpy()
    save["dec_ipos_tail"] = ck["decoder.embed_image_positions.weight"][-41:].numpy()
    save["seg_embed"] = ck["encoder.seg_embed_tokens.weight"].numpy()
    np.savez_compressed(os.path.join(GOLDEN, out_name), **save)
    print("[%s] %d keys after up-conversion; token table %s" % (out_name, len(ck), tuple(ck["encoder.embed_tokens.weight"].shape)))


COCO_UNSEEN = ("frisbee, skateboard, cardboard, carrot, scissors, suitcase, giraffe, cow, road, concrete wall, tree, grass, "
               "river, clouds, playingfield")          # run_scripts/IFSeg/coco_unseen.sh:16


def case_lazy_init(cfg, arch, overrides, out_name):
    """`_lazy_initialization` (criterions/seg_criterion.py:373-407): the category names of the shipped recipe are
    tokenised by the reference's own GPT-2 BPE + dict.txt (utils/BPE), and the reference criterion writes the mean
    token embedding of every name into the seg-token tables of the fixture model (token ids folded into the fixture
    vocabulary).  Golden: the real token ids and the resulting table."""
    import unittest.mock as mock
    _refshim.install()
    from fairseq.data import Dictionary
    from fairseq.data.encoders.gpt2_bpe import GPT2BPE, GPT2BPEConfig
    bdir = os.path.join(_refshim.REFERENCE_ROOT, "utils", "BPE")
    bcfg = GPT2BPEConfig()
    bcfg.gpt2_encoder_json, bcfg.gpt2_vocab_bpe = os.path.join(bdir, "encoder.json"), os.path.join(bdir, "vocab.bpe")
    bpe = GPT2BPE(bcfg)
    d = Dictionary.load(os.path.join(bdir, "dict.txt"))
    names = [x.strip() for x in COCO_UNSEEN.split(",")]

    def encode_text(text):          # the reference's closure, :375-385
        line = " ".join(bpe.encode(" {}".format(w.strip())) for w in text.strip().split())
        return d.encode_line(line=line, add_if_not_exist=False, append_eos=False).long()

    real_ids = [encode_text(" %s" % x) for x in names]
    prompt = encode_text(" what is the segmentation map of the image? object:")
    print("[%s] prompt ids %s" % (out_name, prompt.tolist()))
    cfg15 = O.fixture_config(num_seg_tokens=len(names), vocab_size=cfg.vocab_size)
    model, sd = build_reference(cfg15, arch, overrides)
    crit = build_criterion(cfg15)
    crit.init_seg_with_text = True
    crit.id2rawtext = names
    nvocab = cfg15.vocab_size - 1           # fold real ids into the fixture table (keeps specials out)
    fold = lambda t: 4 + (t % (nvocab - 4))

    class _Bpe:
        def encode(self, w):
            return " ".join(str(int(i)) for i in fold(encode_text(w)))

    class _Dict:
        def encode_line(self, line, add_if_not_exist, append_eos):
            return torch.tensor([int(x) for x in line.split()], dtype=torch.long)

    crit.task.bpe, crit.task.tgt_dict = _Bpe(), _Dict()
    with mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        crit._lazy_initialization(None, model, None)
    out = model.encoder.seg_embed_tokens.weight.data
    assert out.data_ptr() == model.decoder.seg_embed_tokens.weight.data.data_ptr()
    o = torch.stack([sd["encoder.embed_tokens.weight"][fold(t)].mean(0) for t in real_ids])
    assert (o - out).abs().max().item() <= 1e-6
    maxlen = max(t.numel() for t in real_ids)
    ids = np.full((len(names), maxlen), -1, dtype=np.int64)
    for i, t in enumerate(real_ids):
        ids[i, : t.numel()] = t.numpy()
    np.savez_compressed(os.path.join(GOLDEN, out_name), category_ids=ids, prompt_ids=prompt.numpy(), seg_table=out.numpy(),
                        names=np.array(names), fold_mod=nvocab - 4)
    print("  lengths %s; wrote %s" % ([t.numel() for t in real_ids], out_name))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-base", action="store_true")
    ap.add_argument("--only-imfree", action="store_true")
    ap.add_argument("--only-eval", action="store_true")
    ap.add_argument("--only", default="", help="comma list of: train, optim, upgrade, lazy, base_b2, base_c3, padded, base_padded, resize_train, resize_padded, imfree_padded")
    a = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    fx = O.fixture_config()
    ov = dict(encoder_embed_dim=fx.embed_dim, encoder_ffn_embed_dim=fx.ffn_dim, encoder_layers=fx.enc_layers,
              decoder_layers=fx.dec_layers, encoder_attention_heads=fx.heads, decoder_attention_heads=fx.heads)
    only = [x for x in a.only.split(",") if x]
    if only:
        if "optim" in only:
            case_optim(fx, "tiny", ov, "fixture_optim.npz")
        if "upgrade" in only:
            case_upgrade(fx, "tiny", ov, "fixture_upgrade.npz")
        if "lazy" in only:
            case_lazy_init(fx, "tiny", ov, "fixture_lazy_init.npz")
        if "padded" in only:       # prompts of different lengths in one batch: key padding in encoder self- and decoder cross-attention
            case_train(fx, "tiny", ov, 3, 12, "fixture_padded.npz", GRAD_KEYS, pad_tail=[0, 3, 5])
        if "base_padded" in only:  # configs[0] geometry with prompts of different lengths (sample 1 ends in 9 <pad>): key padding at Base size
            case_train(O.base_config(), "base", None, 2, 36, "base_c1_padded.npz", GRAD_KEYS, full_grads=False, sub_all=True, bf16_weights=True,
                       pad_tail=[0, 9])
        if "train" in only:        # the fixture's training case (regenerated in round 6 so that the file carries every key the generator writes)
            case_train(fx, "tiny", ov, 2, 12, "fixture_train.npz", GRAD_KEYS)
        if "resize_train" in only:  # training on a 128 x 192 image (grid 8 x 12, P = 96 > the trained 64): VERDICT r5 item 5
            case_train(fx, "tiny", ov, 2, 12, "fixture_resize_train.npz", GRAD_KEYS + RESIZE_KEYS, sub_all=True, image_hw=(128, 192))
        if "resize_padded" in only:  # the two together: prompts of different lengths on a resized grid (round 6)
            case_train(fx, "tiny", ov, 3, 12, "fixture_resize_padded.npz", GRAD_KEYS + RESIZE_KEYS, sub_all=True, image_hw=(128, 192),
                       pad_tail=[0, 3, 5])
        if "imfree_padded" in only:  # the image-free step with prompts of different lengths (round 6)
            fx512 = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
            case_imfree(fx512, "tiny", ov, 2, 12, "fixture_imfree_padded.npz",
                        [k for k in GRAD_KEYS if "token_rel_pos" not in k] + ["encoder.patch_layernorm_embedding.weight"], pad_tail=[0, 4])
        if "base_b2" in only:      # BASELINE config 1 as written: B = 2
            case_train(O.base_config(), "base", None, 2, 36, "base_c1_b2.npz", GRAD_KEYS, full_grads=False, sub_all=True, bf16_weights=True)
        if "base_c3" in only:      # BASELINE config 3 geometry on one device: Base width, 150 classes, L = 215 (T_enc 1239)
            case_train(O.base_config(num_seg_tokens=150, vocab_size=59457 + 151 - 150), "base", None, 1, 215, "base_c3.npz",
                       GRAD_KEYS, full_grads=False, sub_all=True, bf16_weights=True)
        return
    if a.only_eval:
        case_eval(fx, "tiny", ov, "fixture_eval.npz")
        return
    case_eval(fx, "tiny", ov, "fixture_eval.npz")
    fx512 = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
    imfree_keys = [k for k in GRAD_KEYS if "token_rel_pos" not in k] + ["encoder.patch_layernorm_embedding.weight"]
    case_imfree(fx512, "tiny", ov, 2, 12, "fixture_imfree.npz", imfree_keys)
    if a.only_imfree:
        return
    case_train(fx, "tiny", ov, 2, 12, "fixture_train.npz", GRAD_KEYS)
    case_resize(fx, "tiny", ov, "fixture_resize.npz")
    if not a.skip_base:
        case_train(O.base_config(), "base", None, 1, 36, "base_c1.npz", GRAD_KEYS, full_grads=False)


if __name__ == "__main__":
    main()
