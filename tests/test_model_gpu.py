"""GPU parity: the HIP SegOFA path (through the plugin surface and the C ABI) against
the CPU oracle on the same seeded inputs and procedural weights, and against the
reference-generated golden vectors.  Tolerances (BASELINE.md section 5, bf16 vs fp32):
logits rel-L2 <= 2e-2, loss |d| <= 1e-2, argmax agreement >= 99 %, grads rel-L2 <= 6e-2."""
import math
import os

import numpy as np
import pytest
import torch

import segofa_ref as O

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _product_cfg(ocfg, **over):
    from ifseg_amd.models.segofa import make_config
    return make_config("segofa_tiny", **over, embed_dim=ocfg.embed_dim, ffn_dim=ocfg.ffn_dim, heads=ocfg.heads,
                       enc_layers=ocfg.enc_layers, dec_layers=ocfg.dec_layers, resnet_layers=ocfg.resnet_layers,
                       num_seg_tokens=ocfg.num_seg_tokens, vocab_size=ocfg.vocab_size,
                       patch_image_size=ocfg.patch_image_size, orig_patch_image_size=ocfg.orig_patch_image_size)


def _build(ocfg, sd, dev, **over):
    from ifseg_amd.models.segofa import SegOFAModel
    m = SegOFAModel(_product_cfg(ocfg, **over))
    missing, unexpected = torch.nn.Module.load_state_dict(m, sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith(("_rp_bucket", "version", "image_position_idx", "_offset", "region_prefix")) for k in missing), missing
    return m.to(dev)


def _oracle_all_grads(ocfg, sd, batch, hw):
    sdg = {}
    spec = O.state_dict_spec(ocfg)
    for k, v in sd.items():
        if spec[k][1].startswith("alias:"):
            continue
        sdg[k] = v.clone().requires_grad_(v.dtype.is_floating_point and "embed_images" not in k)
    for k, (_, kind) in spec.items():
        if kind.startswith("alias:"):
            sdg[k] = sdg[kind[6:]]
    logits, extra = O.segofa_forward(sdg, ocfg, batch["src_tokens"], batch["patch_images"])
    hp, wp = extra["encoder_returns"]["image_embed_shape"]
    loss, s, t = O.seg_loss(ocfg, logits, batch["target"], hp, wp, hw[0], hw[1])
    loss.backward()
    return logits.detach(), loss.detach(), {k: v.grad for k, v in sdg.items() if v.grad is not None}, (s.detach(), t)


@pytest.mark.parametrize("streams", ["overlapped", "single_stream"])
def test_fixture_forward_backward_vs_oracle(golden_dir, streams):
    """(single_stream: the engine without its side streams -- no per-block instances of the backward buffers; on this fixture
    the decoder's causal self-attention and its cross-attention have the same padded sum_b dS shape and once shared a buffer)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 2, 12)
    o_logits, o_loss, o_grads, (o_s, o_t) = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    g = np.load(os.path.join(golden_dir, "fixture_train.npz"))
    assert np.abs(o_logits.numpy() - g["logits_causal"]).max() <= 1e-5      # oracle == reference

    m = _build(ocfg, sd, dev)
    m.train()
    m.engine.overlap = streams == "overlapped"
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {"src_tokens": batch["src_tokens"].to(dev), "src_lengths": torch.full((2,), 12).to(dev),
                            "patch_images": batch["patch_images"].to(dev), "patch_masks": batch["patch_masks"].to(dev),
                            "prev_output_tokens": batch["prev_output_tokens"].to(dev)},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": 2}
    loss, sample_size, logs = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    e_logits = _rel(logits, o_logits)
    e_loss = abs(loss.item() - o_loss.item())
    agree = (logits.argmax(-1) == o_logits.argmax(-1)).float().mean().item()
    print("fixture: logits rel-L2 %.4f  loss %.5f vs %.5f (d=%.5f)  argmax agree %.4f" % (e_logits, loss.item(), o_loss.item(), e_loss, agree))
    assert e_logits <= 2e-2 and e_loss <= 1e-2 and agree >= 0.99
    assert abs(loss.item() - float(g["loss"])) <= 1e-2                      # HIP vs reference golden
    assert _rel(logits, torch.from_numpy(g["logits_causal"])) <= 2e-2
    loss.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    worst = []
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0:
            continue
        if k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            # key biases shift every score of a query equally: their gradient is zero in exact
            # arithmetic (fp32 oracle: ~1e-9 noise; bf16 path: ~1e-4 noise) -- only check it is tiny
            assert named[k].grad.float().norm().item() < 2e-2, k
            continue
        hg = named[k].grad
        assert hg is not None, k
        if k.endswith("c_attn"):
            # d c_attn[h] = sum_{b,t,d} dO*O is a heavily cancelling sum of bf16-rounded terms:
            # judge it on the scale of the largest head-gain gradient in the model
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        worst.append((_rel(hg, og), k))
    worst.sort(reverse=True)
    print("worst grads:", [(round(e, 4), k) for e, k in worst[:8]])
    print("checked %d parameter grads; median rel err %.4f" % (len(worst), sorted(e for e, _ in worst)[len(worst) // 2]))
    assert len(worst) > 100
    for e_, k_ in worst[:4]:
        print(k_, "hip", named[k_].grad.float().cpu().flatten()[:6].tolist(), "oracle", o_grads[k_].flatten()[:6].tolist())
    bad = [(e, k) for e, k in worst if e > 6e-2]
    assert not bad, bad[:10]
    # never-used trainable parameters get an exactly zero grad
    assert named["decoder.embed_positions.weight"].grad.abs().sum().item() == 0
    # full-context forward (no causal mask)
    m.eval()
    with torch.no_grad():
        lf, _ = m(**sample["net_input"], full_context_alignment=True)
    assert _rel(lf, torch.from_numpy(g["logits_full"])) <= 2e-2
    # histograms from the HIP logits match the reference's where the argmax agrees
    ai = logs["area_intersect"].cpu().numpy()
    assert np.abs(ai - g["area_intersect"]).sum() <= 0.02 * g["area_label"].sum()


def test_base_config1_vs_reference_golden(golden_dir):
    """SegOFA-Base at BASELINE config-1 shapes (512x512, nseg 15, L=36; B=1): the HIP path against
    the REFERENCE's own outputs stored in tests/golden/base_c1.npz (logits, loss, grad norms)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "base_c1.npz"))
    ocfg = O.base_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 1, int(g["src_len"]))
    m = SegOFAModel(make_config("segofa_base"))
    missing, unexpected = torch.nn.Module.load_state_dict(m, sd, strict=False)
    assert not unexpected
    m.to(dev).train()
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=15, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": 1}
    loss, _, logs = crit(m, sample)
    loss.backward()
    torch.cuda.synchronize()
    logits = m.engine.ws["logits_pad"][:, :, :15].float().cpu()
    ref = torch.from_numpy(g["logits_causal"])
    e = _rel(logits, ref)
    agree = (logits[:, :-1].argmax(-1) == ref[:, :-1].argmax(-1)).float().mean().item()
    print("base c1: logits rel-L2 %.4f, loss %.5f vs reference %.5f, patch argmax agreement %.4f" % (e, loss.item(), float(g["loss"]), agree))
    # BASELINE.md section 5 tolerances as stated (round 1 measured 2.08e-2: rounding LayerNorm gains / biases and the head
    # gains to bf16 alone cost 1.0e-2 -- tools/err_budget2.py; they are read from the fp32 master copy now)
    assert e <= 2e-2 and abs(loss.item() - float(g["loss"])) <= 1e-2 and agree >= 0.99
    named = dict(m.named_parameters())
    for k in g.files:
        if not k.startswith("gradnorm:"):
            continue
        name = k[9:]
        if name.endswith("c_attn"):
            continue
        hn, rn = named[name].grad.float().norm().item(), float(g[k])
        print("   |grad| %-50s hip %.4e  reference %.4e" % (name, hn, rn))
        assert abs(hn - rn) <= 0.06 * rn + 1e-6, (name, hn, rn)
    # mIoU machinery: per-class areas from the fused kernel vs the reference's histograms
    ai, al = logs["area_intersect"].cpu().numpy(), logs["area_label"].cpu().numpy()
    assert np.array_equal(al, g["area_label"])
    assert np.abs(ai - g["area_intersect"]).sum() <= 0.01 * al.sum()


def test_resized_grid_eval_vs_reference_golden(golden_dir):
    """eval on a 128x192 image with a model trained at 128x128 (P = 96 > 64): position-table and
    rel-pos bias bilinear resizes (encoder_module.py:360-368,802-808; decoder_module.py:541-548,603-627)
    through the dense-bias slow path of the HIP attention kernel, against the reference's logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(golden_dir, "fixture_resize.npz"))
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    hw = tuple(int(v) for v in g["image_hw"])
    batch = O.synthetic_batch(ocfg, 1, int(g["src_len"]), image_hw=hw, seed=int(g["seed"]))
    m = _build(ocfg, sd, dev)
    m.eval()
    with torch.no_grad():
        logits, extra = m(src_tokens=batch["src_tokens"].to(dev), patch_images=batch["patch_images"].to(dev),
                          prev_output_tokens=batch["prev_output_tokens"].to(dev), patch_masks=batch["patch_masks"].to(dev))
    assert extra["encoder_returns"]["image_embed_shape"][0] == (hw[0] // 16, hw[1] // 16)
    ref = torch.from_numpy(g["logits_causal"])
    e = _rel(logits, ref)
    print("resized grid: logits rel-L2 %.4f" % e)
    assert logits.shape == ref.shape and e <= 2e-2
    # (training on a resized grid: test_training_on_a_resized_grid_vs_reference_golden)


def test_dropout_path_wiring_and_training_mode(golden_dir):
    """a12: (i) with keep-probability ~1 the dropout/DropPath code path (unfused LN -> dropout kernel ->
    residual, and its adjoint) must reproduce the fused p=0 path; (ii) with the shipped rates (0.1 / 0.1)
    the step is finite, stochastic across steps and reproducible for a fixed seed."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 2, 12)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": 2}
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)

    def run(dropout, dpr, seed):
        m = _build(ocfg, sd, dev)
        m.cfg.dropout, m.cfg.encoder_drop_path_rate, m.cfg.decoder_drop_path_rate = dropout, dpr, dpr
        m.autograd_mode = "arena"
        m.train()
        m.engine.step_seed = seed
        torch.manual_seed(seed)
        loss, _, _ = crit(m, sample)
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), m.engine.g16.float().clone(), m.engine.drop_on

    l0, g0, on0 = run(0.0, 0.0, 1)
    l1, g1, on1 = run(1e-9, 1e-9, 1)
    assert not on0 and on1
    assert abs(l0 - l1) < 2e-3, (l0, l1)
    assert _rel(g1, g0) < 2e-2, _rel(g1, g0)
    la, ga, _ = run(0.1, 0.1, 7)
    lb, gb, _ = run(0.1, 0.1, 7)
    lc, gc, _ = run(0.1, 0.1, 8)
    assert la == lb and torch.equal(ga, gb)          # same seed -> same masks
    assert la != lc and abs(la - l0) > 1e-4          # different seed / vs no dropout
    assert torch.isfinite(ga).all() and ga.norm() > 0


def _parity_case(ocfg, batch_size, src_len, image_hw, tol_logits=2e-2, check_grads=True, tol_grads=8e-2, skip_gain=True):
    """HIP vs CPU oracle (run here, fp32) on a derived configuration."""
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, batch_size, src_len, image_hw=image_hw)
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, image_hw)
    m = _build(ocfg, sd, dev)
    m.train()
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": batch_size}
    loss, _, _ = crit(m, sample)
    loss.backward()
    torch.cuda.synchronize()
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    e = _rel(logits, o_logits)
    print("  logits rel-L2 %.4f  loss %.5f vs %.5f" % (e, loss.item(), o_loss.item()))
    assert e <= tol_logits and abs(loss.item() - o_loss.item()) <= 1e-2
    if check_grads:
        named = dict(m.named_parameters())
        bad = []
        for k, og in o_grads.items():
            if k not in named or not named[k].requires_grad or og.norm() == 0:
                continue
            if k.endswith(("k_proj.bias", "pos_k_linear.bias") + (("c_attn",) if skip_gain else ())):
                continue
            r = _rel(named[k].grad, og)
            if r > tol_grads:
                bad.append((round(r, 4), k))
        assert not bad, sorted(bad)[-8:]


def test_many_classes_long_prompt_vs_oracle():
    """BASELINE config-3 flavour at fixture width: 150 classes, L = 215 text tokens -> the log-spaced
    token buckets (|i-j| > 128, unify_transformer.py:55-68) and the 150-wide seg projection / loss."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ocfg = O.fixture_config(num_seg_tokens=150, vocab_size=400)
    _parity_case(ocfg, 2, 215, (128, 128))


def test_large_geometry_vs_oracle():
    """BASELINE config-4 geometry (SegOFA-Large widths: C=1024, 16 heads, FFN 4096, 640x640 -> 40x40 grid,
    P=1600, 171 classes, L=239) at depth 1+1 with a shallow trunk so the CPU oracle stays in seconds:
    exercises the non-32-wide grid (LDS-histogram bias gradient), NCH=8 LayerNorm, H=16."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ocfg = O.SegOFAConfig(embed_dim=1024, ffn_dim=4096, heads=16, enc_layers=1, dec_layers=1, resnet_layers=(1, 1, 1),
                          num_seg_tokens=171, vocab_size=600, patch_image_size=640, orig_patch_image_size=640)
    _parity_case(ocfg, 1, 239, (640, 640))


def test_large_geometry_depth4_gradient_values_vs_oracle():
    """The same Large geometry at depth 4 + 4 (B = 1: three of a batch-inner workgroup's four batch waves are idle): logits,
    loss and EVERY gradient tensor the oracle produces -- head gains included -- at the stated 6e-2, so an error that only
    shows when gradients have passed through several 1024-wide layers on the 40-wide grid cannot hide at depth 1 + 1."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ocfg = O.SegOFAConfig(embed_dim=1024, ffn_dim=4096, heads=16, enc_layers=4, dec_layers=4, resnet_layers=(1, 1, 1),
                          num_seg_tokens=171, vocab_size=600, patch_image_size=640, orig_patch_image_size=640)
    _parity_case(ocfg, 1, 239, (640, 640), tol_grads=6e-2, skip_gain=False)


def test_smaller_last_batch_on_a_grid_that_is_not_32_wide(monkeypatch):
    """A step with B = 5 followed by one with B = 1 (the last batch of an epoch) on a 40-wide grid: the causal decoder's sum_b dS
    buffer (one slab per four batch elements) is reallocated under the same name and must be cleared again -- the table kernel reads masked pairs of a grid row
    that lie in blocks no causal launch writes.  With fresh workspaces poisoned (0xFF bytes = NaN) the second step's gradients
    must equal, bit for bit, those of a model that only ever saw the B = 1 batch."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.models.segofa import engine as eng_mod
    monkeypatch.setattr(eng_mod, "_POISON", True)
    dev = torch.device("cuda:0")
    ocfg = O.SegOFAConfig(embed_dim=256, ffn_dim=512, heads=4, enc_layers=1, dec_layers=1, resnet_layers=(1, 1, 1),
                          num_seg_tokens=5, vocab_size=400, patch_image_size=640, orig_patch_image_size=640)
    sd = O.procedural_state_dict(ocfg)
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)

    def step(m, bs, seed):
        torch.manual_seed(seed)
        batch = O.synthetic_batch(ocfg, bs, 9, image_hw=(640, 640), seed=seed)
        sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
                  "target": batch["target"].to(dev), "ntokens": 1, "nsentences": bs}
        m.engine.step_seed = 5
        loss, _, _ = crit(m, sample)
        loss.backward()
        torch.cuda.synchronize()
        return m.engine.g16.clone()

    def model():
        m = _build(ocfg, sd, dev)
        m.autograd_mode = "arena"
        m.cfg.dropout, m.cfg.encoder_drop_path_rate, m.cfg.decoder_drop_path_rate = 0.0, 0.0, 0.0
        return m.train()

    a = model()
    step(a, 5, 1)
    g1 = step(a, 1, 2)
    g2 = step(model(), 1, 2)
    assert torch.isfinite(g1.float()).all()
    assert torch.equal(g1, g2)


def test_trunk_prefetch_matches_inline():
    """HipEngine.prefetch_trunk: features computed one batch ahead on the trunk stream give the same logits
    as the in-line trunk, and a different tensor falls back to the in-line path."""
    import torch
    from ifseg_amd.tasks.mm_tasks.segmentation import SegmentationTask
    dev = torch.device("cuda:0")
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")
    model = task.build_model().to(dev).eval()
    s1 = task.synthetic_sample(2, dev, seed=11)
    s2 = task.synthetic_sample(2, dev, seed=12)
    with torch.no_grad():
        ref1 = model(**s1["net_input"])[0].float().clone()
        ref2 = model(**s2["net_input"])[0].float().clone()
        eng = model.engine
        eng.prefetch_trunk(s2["net_input"]["patch_images"])
        assert eng._pf is not None
        out1 = model(**s1["net_input"])[0].float().clone()      # other tensor: in-line trunk, prefetch dropped
        assert eng._pf is None
        eng.prefetch_trunk(s2["net_input"]["patch_images"])
        out2 = model(**s2["net_input"])[0].float().clone()      # same tensor: prefetched features
        assert eng._pf is None
    torch.cuda.synchronize()
    assert torch.equal(out1, ref1)
    assert torch.equal(out2, ref2)
    assert not torch.equal(ref1, ref2)


def test_trunk_lookahead_pass_is_bit_identical_and_every_batch_passes_once():
    """One trunk pass over the NEXT TWO batches (IFSEG_TRUNK_LOOKAHEAD): per-batch features bit-equal to the single-batch
    pass, consumed in order, nothing recomputed while the next batch is cached; a training run with the look-ahead gives
    the same losses and parameters as the run without it (on the 32-wide grid: narrower grids take the LDS-atomic
    bias-gradient path, whose summation order depends on what runs next to it)."""
    import torch
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.tasks.mm_tasks.segmentation import SegmentationTask
    from ifseg_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=512, arch="segofa_tiny")
    model = task.build_model().to(dev).eval()
    ss = [task.synthetic_sample(2, dev, seed=21 + j) for j in range(3)]
    imgs = [q["net_input"]["patch_images"] for q in ss]
    eng = model.engine
    assert eng.trunk_lookahead == 2
    with torch.no_grad():
        refs = [model(**q["net_input"])[0].float().clone() for q in ss]
        calls = []
        orig = eng._resnet
        eng._resnet = lambda images, tag="": (calls.append(images.shape[0]), orig(images, tag))[1]
        eng._prefetch_request(imgs)                          # three on offer, two taken
        assert calls == [4] and eng._pf is not None and len(eng._pf_more) == 1
        o0 = model(**ss[0]["net_input"])[0].float().clone()
        assert len(eng._pf_more) == 0 and eng._pf is not None
        eng._prefetch_request(imgs[1:])                      # the next batch is cached: nothing to do
        assert calls == [4]
        o1 = model(**ss[1]["net_input"])[0].float().clone()
        assert eng._pf is None
        o2 = model(**ss[2]["net_input"])[0].float().clone()  # not covered: in line
        assert calls == [4, 2]
        eng._resnet = orig
    torch.cuda.synchronize()
    for o, r_ in zip((o0, o1, o2), refs):
        assert torch.equal(o, r_)

    def run(ahead):
        torch.manual_seed(0)
        m = task.build_model()
        tr = Trainer(m, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
        losses = []
        for i in range(5):
            nxt = [ss[(i + k) % 3] for k in range(1, ahead + 1)] if ahead else None
            losses.append(float(tr.train_step([ss[i % 3]], prefetch=nxt)[0]["loss"]))
        torch.cuda.synchronize()
        return losses, tr.p32.clone()
    l0, p0 = run(0)
    l3, p3 = run(3)
    assert l0 == l3 and torch.equal(p0, p3)

    # update_freq 2: the second micro-batch of an update is look-ahead for the first, the next call's samples for both
    def run2(ahead):
        torch.manual_seed(0)
        m = task.build_model()
        tr = Trainer(m, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
        calls = []
        orig = tr.eng._resnet
        tr.eng._resnet = lambda images, tag="": (calls.append(images.shape[0]), orig(images, tag))[1]
        losses = []
        for i in range(3):
            cur = [ss[(2 * i) % 3], ss[(2 * i + 1) % 3]]
            nxt = [ss[(2 * i + 2) % 3], ss[(2 * i + 3) % 3]] if ahead else None
            losses += [float(lg["loss"]) for lg in tr.train_step(cur, prefetch=nxt)]
        torch.cuda.synchronize()
        return losses, tr.p32.clone(), calls
    la, pa, ca = run2(False)
    lb, pb, cb = run2(True)
    assert la == lb and torch.equal(pa, pb)
    assert ca == [2] * 6 and sum(cb[1:]) >= 2 * 5 and cb[0] == 2 and 4 in cb, (ca, cb)   # in line once, then passes of two batches


def test_image_free_branch_vs_reference_golden(golden_dir):
    """model(aux_input=...) (EmbeddingBag patches, no trunk, causal decoder) + the criterion's
    image-free branch (loss on the artificial image, no-grad evaluation of the real images in between) against
    the reference golden and the oracle's gradients."""
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_imfree.npz"))
    aux = O.synthetic_aux_batch(ocfg, 2, 12)
    real = O.synthetic_batch(ocfg, 2, 12)
    # oracle gradients of every parameter
    sdg = {}
    spec = O.state_dict_spec(ocfg)
    for k, v in sd.items():
        if not spec[k][1].startswith("alias:"):
            sdg[k] = v.clone().requires_grad_(v.dtype.is_floating_point and "embed_images" not in k)
    for k, (_, kind) in spec.items():
        if kind.startswith("alias:"):
            sdg[k] = sdg[kind[6:]]
    o_logits, _ = O.segofa_forward_imfree(sdg, ocfg, aux["aux_input"])
    o_loss = O.imfree_loss(ocfg, o_logits, aux["text2seg_target"])
    o_loss.backward()
    assert np.abs(o_logits.detach().numpy() - g["logits"]).max() <= 1e-5

    m = _build(ocfg, sd, dev)
    m.train()
    crit = SegCriterion(init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset, unsupervised_segmentation=True)
    to = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
    sample = {"net_input": to({"src_tokens": real["src_tokens"], "src_lengths": torch.full((2,), 12),
                               "patch_images": real["patch_images"], "patch_masks": real["patch_masks"],
                               "prev_output_tokens": real["prev_output_tokens"]}),
              "target": real["target"].to(dev), "aux_input": to(aux["aux_input"]),
              "text2seg_target": aux["text2seg_target"].to(dev), "ntokens": 1, "nsentences": 2}
    loss, sample_size, logs = crit(m, sample)          # aux forward (grad) -> real-image forward (no grad) -> ...
    assert float(logs["imfree_loss"]) == loss.item() and float(logs["seg_loss"]) != loss.item()
    e_loss = abs(loss.item() - float(g["loss"]))
    loss.backward()                                   # ... -> backward of the aux forward
    torch.cuda.synchronize()
    logits = m.engine._ws_grad["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    e_logits = _rel(logits, torch.from_numpy(g["logits"]))
    print("image-free: logits rel-L2 %.4f  loss %.5f vs %.5f" % (e_logits, loss.item(), float(g["loss"])))
    assert e_logits <= 2e-2 and e_loss <= 1e-2
    named = dict(m.named_parameters())
    worst = []
    gain_scale = max(v.grad.abs().max().item() for k, v in sdg.items() if k.endswith("c_attn") and v.grad is not None)
    for k, v in sorted(sdg.items()):
        og = v.grad
        if og is None or k not in named or not named[k].requires_grad or og.norm() == 0:
            continue
        if k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            assert named[k].grad.float().norm().item() < 2e-2, k
            continue
        hg = named[k].grad
        assert hg is not None, k
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        worst.append((_rel(hg, og), k))
    worst.sort(reverse=True)
    print("worst grads:", [(round(e, 4), k) for e, k in worst[:6]], "checked", len(worst))
    assert len(worst) > 100
    # the 12 text tokens are ~1 % of this 1036-token sequence: the text rel-pos tables collect a tiny, heavily
    # cancelling gradient (|g| ~ 1e-4) and sit at the bf16 noise floor -- 8e-2 for them, 6e-2 for everything else
    bad = [(e, k) for e, k in worst if e > (8e-2 if "token_rel_pos_table" in k else 6e-2)]
    assert not bad, bad[:10]
    for k in [f[9:] for f in g.files if f.startswith("gradnorm:")]:
        if k.endswith("c_attn"):
            continue
        assert abs(named[k].grad.float().norm().item() / float(g["gradnorm:" + k]) - 1) <= 5e-2, k
    # the no-grad forward on the real images in between must equal a stand-alone evaluation
    m.eval()
    with torch.no_grad():
        alone = m(**sample["net_input"])[0].float().cpu()
    o_real = O.segofa_forward(sd, ocfg, real["src_tokens"], real["patch_images"])[0]
    assert _rel(alone, o_real) <= 2e-2


def test_image_free_branch_with_padded_prompts_vs_reference_golden(golden_dir):
    """The image-free step with prompts of different lengths (refused until round 6): the artificial image's encoder pass builds the
    same encoder_padding_mask (encoder_module.py:566-606) -- here the same per-sample key counts as the supervised step.
    tests/golden/fixture_imfree_padded.npz is the REFERENCE's output (oracle/gen_golden.py --only imfree_padded: sample 1 ends in
    4 <pad> tokens)."""
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_imfree_padded.npz"))
    aux = O.synthetic_aux_batch(ocfg, 2, 12)
    aux["aux_input"]["src_tokens"] = torch.from_numpy(g["src_tokens"])
    assert (aux["aux_input"]["src_tokens"] == O.PAD).sum(1).tolist() == [0, 4]
    real = O.synthetic_batch(ocfg, 2, 12)
    sdg = {}
    spec = O.state_dict_spec(ocfg)
    for k, v in sd.items():
        if not spec[k][1].startswith("alias:"):
            sdg[k] = v.clone().requires_grad_(v.dtype.is_floating_point and "embed_images" not in k)
    for k, (_, kind) in spec.items():
        if kind.startswith("alias:"):
            sdg[k] = sdg[kind[6:]]
    o_logits, _ = O.segofa_forward_imfree(sdg, ocfg, aux["aux_input"])
    o_loss = O.imfree_loss(ocfg, o_logits, aux["text2seg_target"])
    o_loss.backward()
    assert np.abs(o_logits.detach().numpy() - g["logits"]).max() <= 1e-5 and abs(o_loss.item() - float(g["loss"])) <= 1e-5
    m = _build(ocfg, sd, dev)
    m.cfg.padded_prompts = True
    m.train()
    crit = SegCriterion(init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset, unsupervised_segmentation=True)
    to = lambda d: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}
    sample = {"net_input": to({"src_tokens": real["src_tokens"], "src_lengths": torch.full((2,), 12),
                               "patch_images": real["patch_images"], "patch_masks": real["patch_masks"],
                               "prev_output_tokens": real["prev_output_tokens"]}),
              "target": real["target"].to(dev), "aux_input": to(aux["aux_input"]),
              "text2seg_target": aux["text2seg_target"].to(dev), "ntokens": 1, "nsentences": 2}
    loss, sample_size, logs = crit(m, sample)
    loss.backward()
    torch.cuda.synchronize()
    logits = m.engine._ws_grad["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    ref = torch.from_numpy(g["logits"])
    print("image-free, padded prompts: logits rel-L2 %.4f (padded sample %.4f)  loss %.5f vs %.5f"
          % (_rel(logits, ref), _rel(logits[1], ref[1]), loss.item(), float(g["loss"])))
    assert _rel(logits, ref) <= 2e-2 and abs(loss.item() - float(g["loss"])) <= 1e-2
    named = dict(m.named_parameters())
    gain_scale = max(v.grad.abs().max().item() for k, v in sdg.items() if k.endswith("c_attn") and v.grad is not None)
    worst = []
    for k, v in sorted(sdg.items()):
        og = v.grad
        if og is None or k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        assert hg is not None, k
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        worst.append((_rel(hg, og), k))
    worst.sort(reverse=True)
    print("worst grads:", [(round(e, 4), k) for e, k in worst[:4]], "checked", len(worst))
    assert len(worst) > 100
    bad = [(e, k) for e, k in worst if e > (8e-2 if "token_rel_pos_table" in k else 6e-2)]
    assert not bad, bad[:10]
    for k in [f[5:] for f in g.files if f.startswith("grad:")]:          # the reference's own gradients
        if not k.endswith("c_attn"):
            assert _rel(named[k].grad, torch.from_numpy(g["grad:" + k])) <= 6e-2, k


def test_eval_branch_vs_reference_golden(golden_dir):
    """BASELINE config 5 / SURVEY 8f row 4: SegCriterion eval branch on the device -- logits, metrics at the original
    150x200 resolution and the top-k neighbour smoothing (25 iterations, k=3) against the reference golden."""
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_eval.npz"))
    batch = O.synthetic_batch(ocfg, 1, 12, seed=777)
    sd = O.diversify_seg_projection(sd, ocfg, batch)
    m = _build(ocfg, sd, dev)
    m.eval()
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset, resnet_iters=int(g["iters"]),
                        resnet_topk=int(g["topk"]))
    sample = {"net_input": {"src_tokens": batch["src_tokens"].to(dev), "src_lengths": torch.full((1,), 12).to(dev),
                            "patch_images": batch["patch_images"].to(dev), "patch_masks": batch["patch_masks"].to(dev),
                            "prev_output_tokens": batch["prev_output_tokens"].to(dev)},
              "target": batch["target"].to(dev), "ori_semantic_seg": [g["ori"]], "ori_shape": [g["ori"].shape + (3,)],
              "ntokens": 1, "nsentences": 1}
    loss, ss, logs = crit(m, sample)
    n = ocfg.num_seg_tokens
    valid = int((g["ori"] < n).sum())
    print("eval loss %.5f vs %.5f" % (loss.item(), float(g["loss"])))
    assert abs(loss.item() - float(g["loss"])) <= 1e-2
    assert np.array_equal(logs["area_label"].cpu().numpy(), g["area_label"])                 # label counts are exact
    assert logs["area_pred_label"].sum().item() == valid
    for k in ("area_intersect", "area_pred_label"):
        d = np.abs(logs[k].cpu().numpy() - g[k]).sum()
        dpp = np.abs(logs[k + "_resnet_postprocess"].cpu().numpy() - g[k + "_pp"]).sum()
        print(k, "L1 diff", d, "smoothed", dpp, "of", valid)
        # 5 near-tied synthetic classes: ~3 % of the pixels flip their argmax between bf16 and fp32 logits
        # (each flip moves two histogram bins)
        assert d <= 0.08 * valid and dpp <= 0.05 * valid
    # the resize / argmax / histogram kernel itself, fed the HIP logits, against the restatement on the same logits
    hl = m.engine.ws["logits_pad"][:, :, :n].float().cpu()
    o_loss, o_hist = O.seg_eval(ocfg, hl, torch.from_numpy(g["ori"]), 8, 8)
    assert abs(o_loss.item() - loss.item()) <= 1e-4
    for k, a in zip(("area_intersect", "area_pred_label", "area_label", "area_union"), o_hist):
        assert np.abs(logs[k].cpu().numpy() - a.numpy()).sum() <= 0.002 * valid, k
    # the smoothing itself, on identical inputs (the oracle's logits / features), is exact up to fp32 round-off
    from ifseg_amd import hip
    with torch.no_grad():
        ol, oe = O.segofa_forward(sd, ocfg, batch["src_tokens"], batch["patch_images"])
    feat = oe["encoder_returns"]["image_embed_before_proj"].to(torch.bfloat16)
    lp = torch.zeros(1, ol.shape[1], 8, dtype=torch.bfloat16, device=dev)
    lp[:, :, :n] = ol.to(torch.bfloat16).to(dev)
    prob = hip.neighbour_smoothing(lp, n, feat.to(dev), int(g["iters"]), int(g["topk"]))
    ref = O.neighbour_smoothing(ol.to(torch.bfloat16).float(), feat.float(), int(g["iters"]), int(g["topk"]))[:, :-1]
    assert _rel(prob, ref) <= 2e-2


@pytest.mark.parametrize("arch,size", [("segofa_base", 512), ("segofa_tiny", 128)])
def test_training_step_is_deterministic_across_streams(arch, size, monkeypatch):
    """Two trainers stepped on the same batches (B=2, dropout on; weight-gradient, dQ and trunk-prefetch streams
    active) produce bit-identical losses, gradients and updated parameters, and so does the single-stream execution:
    no race between the streams, no order-dependent reduction anywhere in the step.  Base takes the grouped weight-gradient
    launches; the tiny model the per-projection ones (bias gradients by separate column sums -- where round 4 found two streams
    sharing a partial buffer).  Fresh workspaces are poisoned, so a read of memory nobody wrote shows up as NaN."""
    from ifseg_amd.models.segofa import engine as eng_mod
    monkeypatch.setattr(eng_mod, "_POISON", True)
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer
    dev = torch.device("cuda:0")

    def run(overlap=True):
        torch.manual_seed(0)
        task = SegmentationTask(num_seg_tokens=15, patch_image_size=size, arch=arch)
        model = task.build_model()
        model.cfg.dropout, model.cfg.encoder_drop_path_rate, model.cfg.decoder_drop_path_rate = 0.1, 0.1, 0.1
        tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
        tr.eng.overlap = overlap
        ring = []
        for j in range(2):
            sm = task.synthetic_sample(2, dev, seed=100 + j)
            sm["net_input"]["patch_images"] = sm["net_input"]["patch_images"].to(torch.bfloat16)
            ring.append(sm)
        losses = []
        for i in range(3):
            logs = tr.train_step([ring[i % 2]], prefetch=[ring[(i + 1) % 2]] if overlap else None)
            losses.append(float(logs[-1]["loss"]))
        torch.cuda.synchronize()
        return losses, tr.eng.g16.clone(), tr.eng.p16.clone()

    l1, g1, p1 = run()
    l2, g2, p2 = run()
    assert l1 == l2, (l1, l2)
    assert torch.equal(g1, g2) and torch.equal(p1, p2)
    # ... and identical to the single-stream execution of the same step (wgrad / dQ / trunk streams off)
    l3, g3, p3 = run(overlap=False)
    assert l1 == l3 and torch.equal(g1, g3) and torch.equal(p1, p3)
    assert g1.float().abs().sum().item() > 0 and all(x == x for x in l1)


def test_inference_forward_is_graph_capturable():
    """The eval forward touches only its static workspace, launches every kernel on the current stream and never
    syncs after the first call on a given input tensor, so it can be captured into a HIP graph and replayed on new
    data copied into the static inputs (serving path; tools/infer_time.py measures it on Base)."""
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    m = _build(ocfg, sd, dev)
    m.eval()
    b1, b2 = O.synthetic_batch(ocfg, 2, 12, seed=5), O.synthetic_batch(ocfg, 2, 12, seed=6)
    inp = lambda b: {"src_tokens": b["src_tokens"].to(dev), "src_lengths": torch.full((2,), 12).to(dev),
                     "patch_images": b["patch_images"].to(dev), "patch_masks": b["patch_masks"].to(dev),
                     "prev_output_tokens": b["prev_output_tokens"].to(dev)}
    static, other = inp(b1), inp(b2)
    with torch.no_grad():
        ref_other = m(**other)[0].clone()
        m(**static)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(**static)[0]
        for k in ("src_tokens", "patch_images"):
            static[k].copy_(other[k])
        g.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, ref_other)


def test_update_freq_accumulates_micro_batches():
    """train_step([s1, s2]) (update_freq 2): the gradient handed to Adam is the fp32 sum of the two micro-batch
    gradients, scaled by 1 / (number of micro-batches) inside the fused optimizer step."""
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")

    def make():
        torch.manual_seed(0)
        model = task.build_model()
        model.cfg.dropout = model.cfg.encoder_drop_path_rate = model.cfg.decoder_drop_path_rate = 0.0   # masks depend on the step seed
        return Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)

    s1, s2 = task.synthetic_sample(2, dev, seed=1), task.synthetic_sample(2, dev, seed=2)
    ta = make()
    grads = []
    for s in (s1, s2):                       # the two micro-batch gradients, each from the same initial weights
        t = make()
        t.task.train_step(s, t.model, t.criterion, None, 0)
        torch.cuda.synchronize()
        grads.append(t.eng.g16.float().clone())
    ta.train_step([s1, s2])
    torch.cuda.synchronize()
    want = (grads[0] + grads[1]).to(torch.bfloat16).float()
    got = ta.eng.g16.float()
    if not torch.equal(got, want):          # say WHICH tensors differ (a flaky mismatch was seen once in a full-suite run)
        eng, bad = ta.eng, []
        for n in eng.trainable_names():
            o, k = eng.offs[n], math.prod(eng.shapes[n])
            d = (got[o:o + k] - want[o:o + k]).abs()
            if d.max().item() > 0 or not torch.isfinite(got[o:o + k]).all():
                bad.append((n, int((d > 0).sum()), float(d.max()), float(want[o:o + k].abs().max())))
        raise AssertionError("accumulated gradient differs in %d tensors: %s" % (len(bad), bad[:12]))
    assert ta.num_updates == 1


def test_unsupported_inputs_are_refused_without_a_per_step_sync():
    """Padded prompts / masked-out images are refused loudly: synchronously on a process's first forward, afterwards
    through HipEngine.deferred_check (the flag is read back once its event has completed, at the latest one call later)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 2, 12)
    good = dict(src_tokens=batch["src_tokens"].to(dev), patch_images=batch["patch_images"].to(dev),
                prev_output_tokens=batch["prev_output_tokens"].to(dev))
    padded = dict(good, src_tokens=good["src_tokens"].clone())
    padded["src_tokens"][0, -1] = 1
    m = _build(ocfg, sd, dev)
    m.eval()
    with torch.no_grad():
        with pytest.raises(NotImplementedError):       # first check of the process: synchronous
            m(**padded)
        m(**good)
        with pytest.raises(NotImplementedError):       # later: this call or, once the flag has landed, the next one
            m(**padded)
            torch.cuda.synchronize()
            m(**good)
        torch.cuda.synchronize()
        m(**good)                                       # the engine keeps working after a refusal
        masks = torch.ones(2, dtype=torch.bool, device=dev)
        masks[1] = False
        with pytest.raises(NotImplementedError):
            m(patch_masks=masks, **good)
            torch.cuda.synchronize()
            m(**good)


def test_parameters_stepped_from_outside_are_seen_by_the_next_forward(golden_dir):
    """`master_owned=False` (the plugin under the reference's own trainer: fairseq's FP16Optimizer owns fp32 masters and writes
    the half parameters back with `p.data.copy_(p32)`, custom_fairseq/fairseq/optim/fp16_optimizer.py:96-222 -- no version
    counter moves): a stand-in optimizer steps every trainable tensor from OUTSIDE the engine; the next training forward /
    backward and the first evaluation forward after a further `.data` edit of the LayerNorm gains (which the engine reads from
    its own fp32 copy) must match the oracle on the stepped values."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 2, 12)
    m = _build(ocfg, sd, dev)
    m.train()
    assert not m.engine.master_owned
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {"src_tokens": batch["src_tokens"].to(dev), "src_lengths": torch.full((2,), 12).to(dev),
                            "patch_images": batch["patch_images"].to(dev), "patch_masks": batch["patch_masks"].to(dev),
                            "prev_output_tokens": batch["prev_output_tokens"].to(dev)},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": 2}
    loss, _, _ = crit(m, sample)
    logits0 = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    loss.backward()
    named = dict(m.named_parameters())
    # the stand-in optimizer: fp32 masters, a normalised-gradient step of up to 5 % of a tensor's mean magnitude, written back
    # through .data (exactly what FP16Optimizer._sync_fp32_params_to_fp16 does)
    with torch.no_grad():
        for k, p in named.items():
            if not p.requires_grad or p.grad is None:
                continue
            p32 = p.data.float()
            g = p.grad.float()
            p32 -= 0.05 * p32.abs().mean() * g / (g.abs().max() + 1e-12)
            p.data.copy_(p32)
            p.grad = None
    def oracle_on_current(edit=None):
        sd2 = {k: v.clone() for k, v in sd.items()}
        for k, p in named.items():
            if k in sd2 and p.requires_grad:
                sd2[k] = p.data.float().cpu()
        return sd2
    sd2 = oracle_on_current()
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd2, batch, (128, 128))
    loss, _, _ = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    moved = _rel(logits, logits0)
    print("external step: logits moved by %.4f, vs oracle on the stepped values %.4f, loss d %.5f" % (moved, _rel(logits, o_logits), abs(loss.item() - o_loss.item())))
    assert moved > 5e-2                                   # the step is visible at all (the test has power)
    assert _rel(logits, o_logits) <= 2e-2 and abs(loss.item() - o_loss.item()) <= 1e-2
    loss.backward()
    torch.cuda.synchronize()
    bad = []
    for k, og in o_grads.items():
        if k in named and named[k].requires_grad and og.norm() > 0 and not k.endswith(("k_proj.bias", "pos_k_linear.bias", "c_attn")):
            e = _rel(named[k].grad, og)
            if e > 6e-2:
                bad.append((round(e, 4), k))
    assert not bad, bad[:10]
    # an EMA-like swap of the LayerNorm gains through .data, then evaluation: the engine's fp32 copy must follow
    with torch.no_grad():
        for k, p in named.items():
            if p.requires_grad and "layer_norm" in k and k.endswith("weight"):
                p.data.copy_(p.data.float() * 1.25)
    sd3 = oracle_on_current()
    with torch.no_grad():
        o_eval, _ = O.segofa_forward(sd3, ocfg, batch["src_tokens"], batch["patch_images"])
    m.eval()
    with torch.no_grad():
        le, _ = m(**sample["net_input"])
    le = le[..., : ocfg.num_seg_tokens].float().cpu() if le.shape[-1] != o_eval.shape[-1] else le.float().cpu()
    print("eval after a .data edit of the LayerNorm gains: rel-L2 %.4f (against the pre-edit logits %.4f)" % (_rel(le, o_eval), _rel(le, o_logits)))
    assert _rel(o_eval, o_logits) > 5e-2                  # the edit is visible in the oracle
    assert _rel(le, o_eval) <= 2e-2


@pytest.mark.parametrize("delay", [False, True])
def test_deferred_optimizer_is_bit_equal_and_the_next_forward_waits_for_its_slices(delay, monkeypatch):
    """Trainer.train_step(defer_optimizer=True): clip + Adam of update n run per parameter slice on their own stream while the
    forward of update n + 1 starts (trainer.py:865-907 / optim/adam.py:45-110 semantics unchanged).  Six updates with dropout
    (the masks depend on the step seed only) against the same six updates with the single launch on the main stream:
    parameters, fp32 masters, both Adam moments and the losses bit-equal; an evaluation right behind a deferred update sees
    the updated weights (valid_step waits).
    delay: every Adam slice is preceded by a ~1.5 ms sleep kernel on the stream it runs on, so the optimizer is
    BEHIND the next forward instead of well ahead of it: a forward that reads a slice it did not wait for -- the FFN tail of
    encoder layer l computes layer l + 1's pre-LN and reads that layer's gains -- would see stale parameters and the bits would
    differ."""
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer
    from ifseg_amd import hip as hip_mod
    dev = torch.device("cuda:0")
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")
    if delay:
        real = hip_mod.adam_step

        def slow_adam(*a, **k):
            torch.cuda._sleep(3_500_000)          # on torch's current stream = the stream the slice is launched on
            return real(*a, **k)
        monkeypatch.setattr(hip_mod, "adam_step", slow_adam)

    def make():
        torch.manual_seed(0)
        model = task.build_model()
        return Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, lr=1e-3, device=dev)

    samples = [task.synthetic_sample(2, dev, seed=s) for s in range(3)]
    runs = []
    for defer in (False, True):
        tr = make()
        losses = []
        for k in range(6):
            logs = tr.train_step([samples[k % 3]], prefetch=[samples[(k + 1) % 3]], defer_optimizer=defer)
            losses.append(logs[0]["loss"])
        vl = tr.valid_step(samples[0])                       # right behind the last (possibly still running) update
        torch.cuda.synchronize()
        assert tr.eng._popt is None
        runs.append((tr.eng.p16.clone(), tr.p32.clone(), tr.m.clone(), tr.v.clone(), [float(x) for x in losses],
                     float(vl[0] if isinstance(vl, (tuple, list)) else vl)))
        tr.close()
    a, b = runs
    for x, y, name in zip(a[:4], b[:4], ("p16", "p32", "m", "v")):
        assert torch.equal(x, y), name
    assert a[4] == b[4] and a[5] == b[5], (a[4], b[4], a[5], b[5])


def test_padded_prompts_vs_reference_golden(golden_dir):
    """Prompts of different lengths in one batch (right-padded with <pad>): the reference builds encoder_padding_mask
    (encoder_module.py:730-752), zeroes those embedding rows and masks those keys in the encoder self-attention and the decoder
    cross-attention (unify_multihead_attention.py:477-489).  Here: `cfg.padded_prompts` -> valid key counts per sample into the
    batch-inner attention kernels (ifseg_attn_bi_args.kv_len).  tests/golden/fixture_padded.npz is the REFERENCE's output for
    B = 3 with 0 / 3 / 5 padded tokens (oracle/gen_golden.py --only padded); logits, loss and every gradient the oracle
    produces are compared, training and evaluation; without the switch the batch is refused."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_padded.npz"))
    B = int(g["batch_size"])
    batch = O.synthetic_batch(ocfg, B, int(g["src_len"]))
    batch["src_tokens"] = torch.from_numpy(g["src_tokens"])
    assert (batch["src_tokens"] == O.PAD).sum(1).tolist() == [0, 3, 5]
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    assert np.abs(o_logits.numpy() - g["logits_causal"]).max() <= 1e-5 and abs(o_loss.item() - float(g["loss"])) <= 1e-5
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {"src_tokens": batch["src_tokens"].to(dev), "src_lengths": torch.full((B,), 12).to(dev),
                            "patch_images": batch["patch_images"].to(dev), "patch_masks": batch["patch_masks"].to(dev),
                            "prev_output_tokens": batch["prev_output_tokens"].to(dev)},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    m = _build(ocfg, sd, dev)
    m.train()
    with pytest.raises(NotImplementedError):          # default: a padded batch is refused (first check of its kind on an engine: synchronous)
        crit(m, sample)
    m.cfg.padded_prompts = True
    loss, _, logs = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    ref = torch.from_numpy(g["logits_causal"])
    print("padded prompts: logits rel-L2 %.4f (unpadded sample %.4f, 5 pads %.4f), loss %.5f vs %.5f"
          % (_rel(logits, ref), _rel(logits[0], ref[0]), _rel(logits[2], ref[2]), loss.item(), float(g["loss"])))
    assert _rel(logits, ref) <= 2e-2 and abs(loss.item() - float(g["loss"])) <= 1e-2
    assert (logits.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.99
    loss.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    bad, n = [], 0
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        n += 1
        if _rel(hg, og) > 6e-2:
            bad.append((round(_rel(hg, og), 4), k))
    assert n > 100 and not bad, bad[:10]
    for k in g.files:                                   # the reference's own gradients of the golden's keys
        if k.startswith("grad:") and not k.endswith("c_attn"):
            assert _rel(named[k[5:]].grad, torch.from_numpy(g[k])) <= 6e-2, k
    m.eval()
    with torch.no_grad():
        lf, ex = m(**sample["net_input"], full_context_alignment=True)
    assert _rel(lf, torch.from_numpy(g["logits_full"])) <= 2e-2
    # the mask the reference returns (encoder_module.py:730-752,838): True at <pad> source tokens, never at a patch position
    pm = ex["encoder_returns"]["encoder_padding_mask"][0].cpu()
    P_ = pm.shape[1] - batch["src_tokens"].shape[1]
    assert not pm[:, :P_].any() and torch.equal(pm[:, P_:], batch["src_tokens"] == O.PAD)


def test_training_on_a_resized_grid_vs_reference_golden(golden_dir):
    """Training on an image whose feature grid (8 x 12, P = 96) is not the trained one (8 x 8): the reference resizes the image
    rows of both position tables (encoder_module.py:356-372, decoder_module.py:541-550) and every layer's relative-position
    bias (encoder_module.py:798-809, decoder_module.py:603-627) under autograd.  tests/golden/fixture_resize_train.npz is the
    REFERENCE's output (oracle/gen_golden.py --only resize_train: logits, loss, and its gradients of the position tables, the
    rel-pos bucket tables and the rest); the engine takes the standard training step with dense biases from
    models/segofa/resized.py and the adjoints of the resizes (this configuration used to be refused)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_resize_train.npz"))
    B, hw = int(g["batch_size"]), tuple(int(v) for v in g["image_hw"])
    batch = O.synthetic_batch(ocfg, B, int(g["src_len"]), image_hw=hw)
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, hw)
    assert np.abs(o_logits.numpy() - g["logits_causal"]).max() <= 1e-5 and abs(o_loss.item() - float(g["loss"])) <= 1e-5
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    m = _build(ocfg, sd, dev)
    m.train()
    loss, _, logs = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    ref = torch.from_numpy(g["logits_causal"])
    print("resized-grid training: logits rel-L2 %.4f, loss %.5f vs %.5f" % (_rel(logits, ref), loss.item(), float(g["loss"])))
    assert logits.shape == ref.shape and _rel(logits, ref) <= 2e-2 and abs(loss.item() - float(g["loss"])) <= 1e-2
    assert (logits.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.99
    loss.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    bad, n = [], 0
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        n += 1
        if _rel(hg, og) > 6e-2:
            bad.append((round(_rel(hg, og), 4), k))
    assert n > 100 and not bad, bad[:10]
    # the reference's own gradients of the tensors behind the resizes
    shown = {}
    for k in g.files:
        if k.startswith("grad:") and not k.endswith("c_attn"):
            shown[k[5:]] = _rel(named[k[5:]].grad, torch.from_numpy(g[k]))
            assert shown[k[5:]] <= 6e-2, (k, shown[k[5:]])
    for k in ("encoder.embed_image_positions.weight", "decoder.embed_seg_positions.weight", "encoder.image_rel_pos_table_list.0.weight",
              "encoder.image_rel_pos_table_list.1.weight", "decoder.seg_rel_pos_table_list.0.weight", "decoder.seg_rel_pos_table_list.1.weight",
              "encoder.token_rel_pos_table_list.1.weight"):
        assert k in shown, k
    print("  gradients behind the resizes:", {k.split(".")[1] + "." + k.split(".")[-2]: round(v, 4) for k, v in shown.items() if "pos" in k})
    # a second step on the trained grid afterwards (the engine switches paths per batch), then evaluation on the resized one
    sq = O.synthetic_batch(ocfg, B, int(g["src_len"]))
    loss2, _, _ = crit(m, {"net_input": {k: sq[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
                           "target": sq["target"].to(dev), "ntokens": 1, "nsentences": B})
    loss2.backward()
    m.eval()
    with torch.no_grad():
        lf, _ = m(**sample["net_input"], full_context_alignment=True)
    assert _rel(lf, torch.from_numpy(g["logits_full"])) <= 2e-2


def test_padded_prompts_on_a_resized_grid_vs_reference_golden(golden_dir):
    """The two together (refused until round 6): prompts of different lengths (0 / 3 / 5 <pad> tokens) on a 128 x 192 image whose
    feature grid (8 x 12) is not the trained one.  Training takes the standard step with the dense biases of
    models/segofa/resized.py and per-sample key counts (ifseg_attn_bi_args.kv_len); the evaluation of such a batch takes the same
    step instead of the cached-bias slow path.  tests/golden/fixture_resize_padded.npz is the REFERENCE's output
    (oracle/gen_golden.py --only resize_padded): logits, loss, gradients, evaluation logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    g = np.load(os.path.join(golden_dir, "fixture_resize_padded.npz"))
    B, hw = int(g["batch_size"]), tuple(int(v) for v in g["image_hw"])
    batch = O.synthetic_batch(ocfg, B, int(g["src_len"]), image_hw=hw)
    batch["src_tokens"] = torch.from_numpy(g["src_tokens"])
    assert (batch["src_tokens"] == O.PAD).sum(1).tolist() == [0, 3, 5]
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, hw)
    assert np.abs(o_logits.numpy() - g["logits_causal"]).max() <= 1e-5 and abs(o_loss.item() - float(g["loss"])) <= 1e-5
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    m = _build(ocfg, sd, dev)
    m.cfg.padded_prompts = True
    m.train()
    loss, _, logs = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    ref = torch.from_numpy(g["logits_causal"])
    print("padded prompts on a resized grid: logits rel-L2 %.4f (5 pads %.4f), loss %.5f vs %.5f"
          % (_rel(logits, ref), _rel(logits[2], ref[2]), loss.item(), float(g["loss"])))
    assert logits.shape == ref.shape and _rel(logits, ref) <= 2e-2 and abs(loss.item() - float(g["loss"])) <= 1e-2
    assert (logits.argmax(-1) == ref.argmax(-1)).float().mean().item() >= 0.99
    loss.backward()
    torch.cuda.synchronize()
    named = dict(m.named_parameters())
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    bad, n = [], 0
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        n += 1
        if _rel(hg, og) > 6e-2:
            bad.append((round(_rel(hg, og), 4), k))
    assert n > 100 and not bad, bad[:10]
    for k in g.files:                                   # the reference's own gradients of the golden's keys
        if k.startswith("grad:") and not k.endswith("c_attn"):
            assert _rel(named[k[5:]].grad, torch.from_numpy(g[k])) <= 6e-2, k
    m.eval()
    with torch.no_grad():
        lf, ex = m(**sample["net_input"], full_context_alignment=True)
    assert _rel(lf, torch.from_numpy(g["logits_full"])) <= 2e-2
    pm = ex["encoder_returns"]["encoder_padding_mask"][0].cpu()
    P_ = pm.shape[1] - batch["src_tokens"].shape[1]
    assert not pm[:, :P_].any() and torch.equal(pm[:, P_:], batch["src_tokens"] == O.PAD)


def test_label_smoothing_runs_in_the_fused_criterion():
    """--label-smoothing > 0 (seg_criterion.py:142,265: F.cross_entropy(label_smoothing=eps)) stays on the fused loss kernel:
    loss, metrics and the gradient handed to the decoder equal the torch composition of the reference ops on the same logits."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    batch = O.synthetic_batch(ocfg, 2, 12)
    m = _build(ocfg, sd, dev)
    m.train()
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": 2}
    sample["net_input"]["src_lengths"] = torch.full((2,), 12).to(dev)
    crit = SegCriterion(label_smoothing=0.1, unsupervised_segmentation=False, init_seg_with_text=False,
                        num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    out = m(**sample["net_input"])
    loss, metrics, _ = crit.compute_loss(m, out, sample, 0)
    assert "dl" in crit._bufs, "the fused kernel did not run"
    fused_grad = crit._bufs["dl"][:, :, : ocfg.num_seg_tokens].float().clone()
    lg = out[0].detach().float().requires_grad_(True)
    tl, tm, _ = crit.compute_loss_torch(m, (lg, out[1]), sample, 0)
    tl.backward()
    plain, _, _ = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens,
                               seg_id_offset=ocfg.seg_id_offset).compute_loss(m, out, sample, 0)
    print("label smoothing 0.1: fused %.6f torch %.6f (plain CE %.6f)" % (loss.item(), tl.item(), plain.item()))
    assert abs(loss.item() - tl.item()) <= 2e-4 * max(1.0, abs(tl.item())) and abs(loss.item() - plain.item()) > 1e-4   # (near-uniform logits: smoothing moves the loss little)
    assert _rel(fused_grad, lg.grad) <= 6e-3
    assert torch.equal(metrics["area_label"].cpu(), tm["area_label"].cpu())


def test_attention_dropout_against_the_oracle_with_the_same_masks():
    """--attention-dropout > 0 (unify_multihead_attention.py:498; 0.0 in every shipped script): the mask lives inside the
    attention kernels (counter-based, regenerated by the backward).  ifseg_attn_dropout_mask writes out the masks of one step;
    the oracle applies exactly those after its softmax (oracle.ATTN_PROB_HOOK) -- logits, loss and every gradient must then
    agree as they do without dropout.  Also: the masks change with the update number, and evaluation applies none."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd import hip
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    B, p = 3, 0.25
    batch = O.synthetic_batch(ocfg, B, 12)
    m = _build(ocfg, sd, dev)
    m.cfg.attention_dropout = p
    m.train()
    eng = m.engine
    eng.step_seed = 17
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    sample["net_input"]["src_lengths"] = torch.full((B,), 12).to(dev)
    loss, _, _ = crit(m, sample)
    logits = eng.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    loss.backward()
    torch.cuda.synchronize()
    H, P = ocfg.heads, (ocfg.patch_image_size // 16) ** 2
    Te, Td = P + 12, P + 1
    masks = {}
    prev = hip.set_seed_add(eng.step_dev)
    try:
        for l in range(ocfg.enc_layers):
            masks["encoder.layers.%d.self_attn" % l] = hip.attn_dropout_mask(B, H, Te, Te, p, eng._attn_drop("e%d" % l, p)[1], dev)
        for l in range(ocfg.dec_layers):
            masks["decoder.layers.%d.self_attn" % l] = hip.attn_dropout_mask(B, H, Td, Td, p, eng._attn_drop("d%d" % l, p)[1], dev)
            masks["decoder.layers.%d.encoder_attn" % l] = hip.attn_dropout_mask(B, H, Td, Te, p, eng._attn_drop("d%dc" % l, p)[1], dev)
    finally:
        hip.set_seed_add(prev)
    torch.cuda.synchronize()
    masks = {k: v.float().cpu() / (1 - p) for k, v in masks.items()}
    # the engine keeps the decoder's rows as [patches..., bos] (its causal schedule's "tail"), the reference as [bos, patches...]
    perm = torch.tensor([P] + list(range(P)))
    for l in range(ocfg.dec_layers):
        ks, kc = "decoder.layers.%d.self_attn" % l, "decoder.layers.%d.encoder_attn" % l
        masks[ks] = masks[ks][:, :, perm][:, :, :, perm]
        masks[kc] = masks[kc][:, :, perm]
    assert all(abs(v.mean().item() - 1.0) < 0.05 for v in masks.values())
    assert not torch.equal(masks["encoder.layers.0.self_attn"], masks["encoder.layers.1.self_attn"])
    seen = []

    def hook(prefix, pr):
        seen.append(prefix)
        return pr * masks[prefix].to(pr.dtype)
    O.ATTN_PROB_HOOK = hook
    try:
        o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    finally:
        O.ATTN_PROB_HOOK = None
    assert len(seen) == ocfg.enc_layers + 2 * ocfg.dec_layers
    plain_logits, plain_loss, _, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    print("attention dropout %.2f: logits rel-L2 %.4f (the masks move the oracle's logits by %.4f), loss %.5f vs %.5f"
          % (p, _rel(logits, o_logits), _rel(o_logits, plain_logits), loss.item(), o_loss.item()))
    assert _rel(o_logits, plain_logits) > 5 * _rel(logits, o_logits)
    assert _rel(logits, o_logits) <= 2e-2 and abs(loss.item() - o_loss.item()) <= 1e-2
    named = dict(m.named_parameters())
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    bad, n = [], 0
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        n += 1
        if _rel(hg, og) > 6e-2:
            bad.append((round(_rel(hg, og), 4), k))
    assert n > 100 and not bad, bad[:10]
    # another update number: other masks; evaluation: none
    eng.step_seed = 18
    crit(m, sample)
    l2 = eng.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    assert _rel(l2, logits) > 1e-3
    m.eval()
    with torch.no_grad():
        le, _ = m(**sample["net_input"])
    assert _rel(le, plain_logits) <= 2e-2


def test_activation_dropout_against_the_oracle_with_the_same_masks():
    """--activation-dropout > 0 (unify_transformer_layer.py:142-147,280,556; 0 in every shipped script): dropped FFN
    pre-activations are set to -30 in place (gelu = gelu' = 0 exactly), the 1 / (1 - p) cancels in ffn_layernorm via
    eps (1 - p)^2 -- no kernel of the FFN forward / backward knows about the mask.  The oracle applies the same masks
    (oracle.ACT_HOOK): logits, loss and every gradient agree as without dropout."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd import hip
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    B, p = 3, 0.3
    batch = O.synthetic_batch(ocfg, B, 12)
    m = _build(ocfg, sd, dev)
    m.cfg.activation_dropout = p
    m.train()
    eng = m.engine
    eng.step_seed = 5
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    sample["net_input"]["src_lengths"] = torch.full((B,), 12).to(dev)
    loss, _, _ = crit(m, sample)
    logits = eng.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    loss.backward()
    torch.cuda.synchronize()
    P, Fd = (ocfg.patch_image_size // 16) ** 2, ocfg.ffn_dim
    Te, Td = P + 12, P + 1
    masks = {}
    prev = hip.set_seed_add(eng.step_dev)
    try:
        for kind, n, T, base in (("encoder", ocfg.enc_layers, Te, 4000), ("decoder", ocfg.dec_layers, Td, 5000)):
            for l in range(n):
                ones = torch.ones(B * T, Fd, dtype=torch.bfloat16, device=dev)
                masks["%s.layers.%d." % (kind, l)] = hip.dropout_fill(ones, torch.empty_like(ones), p, eng._site_seed(base + l), fill=0.0).view(B, T, Fd)
    finally:
        hip.set_seed_add(prev)
    torch.cuda.synchronize()
    masks = {k: v.float().cpu() / (1 - p) for k, v in masks.items()}
    perm = torch.tensor([P] + list(range(P)))          # engine rows [patches..., bos] -> reference rows [bos, patches...]
    for l in range(ocfg.dec_layers):
        masks["decoder.layers.%d." % l] = masks["decoder.layers.%d." % l][:, perm]
    assert all(abs(v.mean().item() - 1.0) < 0.02 for v in masks.values())
    seen = []

    def hook(prefix, a):
        seen.append(prefix)
        return a * masks[prefix].to(a.dtype)
    O.ACT_HOOK = hook
    try:
        o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    finally:
        O.ACT_HOOK = None
    assert len(seen) == ocfg.enc_layers + ocfg.dec_layers
    plain_logits, _, _, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    print("activation dropout %.2f: logits rel-L2 %.4f (the masks move the oracle's logits by %.4f), loss %.5f vs %.5f"
          % (p, _rel(logits, o_logits), _rel(o_logits, plain_logits), loss.item(), o_loss.item()))
    assert _rel(o_logits, plain_logits) > 5 * _rel(logits, o_logits)
    assert _rel(logits, o_logits) <= 2e-2 and abs(loss.item() - o_loss.item()) <= 1e-2
    named = dict(m.named_parameters())
    gain_scale = max(v.abs().max().item() for k, v in o_grads.items() if k.endswith("c_attn"))
    bad, n = [], 0
    for k, og in sorted(o_grads.items()):
        if k not in named or not named[k].requires_grad or og.norm() == 0 or k.endswith(("k_proj.bias", "pos_k_linear.bias")):
            continue
        hg = named[k].grad
        if k.endswith("c_attn"):
            assert (hg.float().cpu() - og).abs().max().item() <= 5e-2 * gain_scale, k
            continue
        n += 1
        if _rel(hg, og) > 6e-2:
            bad.append((round(_rel(hg, og), 4), k))
    assert n > 100 and not bad, bad[:10]
    m.eval()
    with torch.no_grad():
        le, _ = m(**sample["net_input"])
    assert _rel(le, plain_logits) <= 2e-2



def test_trainable_token_and_seg_embeddings_vs_the_oracle():
    """--freeze-encoder-embedding / --freeze-decoder-embedding / --freeze-seg-embedding false (unify_transformer.py:362-373;
    every shipped script freezes all three): the token table's gradient is the nn.Embedding backward of the prompt tokens and
    of the decoder's bos row (one shared tensor; padding_idx row excluded), the seg embeddings' gradient is the tied projection's
    dlogits^T feat.  Against the oracle's autograd on the same (here: padded, repeated-token) batch; every other gradient
    unchanged; an optimizer step moves both tensors and the next forward -- training and evaluation -- projects with the
    stepped seg embeddings; the image-free entry with a trainable token table is refused by name."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ifseg_amd.criterions import SegCriterion
    dev = torch.device("cuda:0")
    ocfg = O.fixture_config()
    sd = O.procedural_state_dict(ocfg)
    B = 3
    batch = O.synthetic_batch(ocfg, B, 12)
    batch["src_tokens"][1, -3:] = O.PAD; batch["src_tokens"][1, -4] = O.EOS            # one padded prompt
    batch["src_tokens"][2, 3] = batch["src_tokens"][0, 5]                              # a token shared between samples
    m = _build(ocfg, sd, dev, freeze_embeddings=False, padded_prompts=True)
    assert m.encoder.embed_tokens.weight.requires_grad and m.decoder.seg_projection.weight.requires_grad
    m.train()
    crit = SegCriterion(unsupervised_segmentation=False, init_seg_with_text=False, num_seg_tokens=ocfg.num_seg_tokens, seg_id_offset=ocfg.seg_id_offset)
    sample = {"net_input": {k: batch[k].to(dev) for k in ("src_tokens", "patch_images", "patch_masks", "prev_output_tokens")},
              "target": batch["target"].to(dev), "ntokens": 1, "nsentences": B}
    sample["net_input"]["src_lengths"] = torch.full((B,), 12).to(dev)
    loss, _, _ = crit(m, sample)
    logits = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    loss.backward()
    torch.cuda.synchronize()
    assert m.engine.train_tok and m.engine.train_seg
    o_logits, o_loss, o_grads, _ = _oracle_all_grads(ocfg, sd, batch, (128, 128))
    assert _rel(logits, o_logits) <= 2e-2 and abs(loss.item() - o_loss.item()) <= 1e-2
    named = dict(m.named_parameters())
    gt, ot = named["encoder.embed_tokens.weight"].grad.float().cpu(), o_grads["encoder.embed_tokens.weight"]
    gs, os_ = named["encoder.seg_embed_tokens.weight"].grad.float().cpu(), o_grads["encoder.seg_embed_tokens.weight"]
    used = ot.abs().sum(1) > 0
    print("token table: %d rows with a gradient, rel-L2 %.4f; seg embeddings rel-L2 %.4f" % (int(used.sum()), _rel(gt, ot), _rel(gs, os_)))
    assert int(used.sum()) >= 12 and gt[~used].abs().max().item() == 0.0 and gt[O.PAD].abs().max().item() == 0.0
    assert _rel(gt, ot) <= 6e-2 and _rel(gs, os_) <= 6e-2
    for r in torch.nonzero(used).flatten().tolist():                     # row by row (bos: encoder + decoder contributions)
        assert _rel(gt[r], ot[r]) <= 8e-2, r
    bad = []
    for k, og in sorted(o_grads.items()):
        if k in named and named[k].requires_grad and og.norm() > 0 and not k.endswith(("k_proj.bias", "pos_k_linear.bias", "c_attn")):
            if _rel(named[k].grad, og) > 6e-2:
                bad.append((round(_rel(named[k].grad, og), 4), k))
    assert not bad, bad[:10]
    # one plain SGD step from outside on the two tensors: the tied projection follows in training and evaluation
    with torch.no_grad():
        for k in ("encoder.embed_tokens.weight", "encoder.seg_embed_tokens.weight"):
            named[k].data.add_(named[k].grad, alpha=-0.5)
    sd2 = {k: v.clone() for k, v in sd.items()}
    spec = O.state_dict_spec(ocfg)
    for k in ("encoder.embed_tokens.weight", "encoder.seg_embed_tokens.weight"):
        sd2[k] = named[k].data.float().cpu()
    for k, (_, kind) in spec.items():
        if kind.startswith("alias:") and kind[6:] in ("encoder.embed_tokens.weight", "encoder.seg_embed_tokens.weight"):
            sd2[k] = sd2[kind[6:]]
    loss2, _, _ = crit(m, sample)
    l2 = m.engine.ws["logits_pad"][:, :, : ocfg.num_seg_tokens].float().cpu()
    o2, _, _, _ = _oracle_all_grads(ocfg, sd2, batch, (128, 128))
    assert _rel(o2, o_logits) > 5e-2 and _rel(l2, o2) <= 2e-2, (_rel(o2, o_logits), _rel(l2, o2))
    m.eval()
    with torch.no_grad():
        le, _ = m(**sample["net_input"])
    assert _rel(le, o2) <= 2e-2
    m.train()
    aux = O.synthetic_aux_batch(ocfg, B, 12)
    with pytest.raises(NotImplementedError):
        m(**sample["net_input"], aux_input={k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in aux["aux_input"].items()})


def test_trainer_steps_trainable_embeddings_through_the_reference_flags():
    """The fairseq surface: --freeze-encoder-embedding=false --freeze-decoder-embedding=false --freeze-seg-embedding=false
    (segofa.py build_model -> SegOFAConfig.freeze_embeddings / freeze_seg_embedding).  The bundled Trainer's arena then holds
    the token table ("emb" slice of optimizer_plan) and the seg embeddings; three updates move both, only in the rows that
    received a gradient (weight decay 0), and the loss of the fixed batch goes down."""
    from ifseg_amd.criterions import SegCriterion
    from ifseg_amd.models.segofa.segofa import SegOFAModel, recipe_args
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    from ifseg_amd.trainer import Trainer, optimizer_plan
    dev = torch.device("cuda:0")
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")
    torch.manual_seed(0)
    args = recipe_args("segofa_tiny", num_seg_tokens=5, patch_image_size=128, orig_patch_image_size=128, dropout=0.0,
                       encoder_drop_path_rate=0.0, decoder_drop_path_rate=0.0, freeze_encoder_embedding="false",
                       freeze_decoder_embedding="false", freeze_seg_embedding="false")
    model = SegOFAModel.build_model(args, task)
    assert not model.cfg.freeze_embeddings and model.cfg.freeze_seg_embedding is False
    tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, lr=3e-4, weight_decay=0.0, device=dev)
    s = task.synthetic_sample(2, dev, seed=0)
    tok0 = model.encoder.embed_tokens.weight.detach().float().clone()
    seg0 = model.decoder.seg_projection.weight.detach().float().clone()
    losses = [float(tr.train_step([s])[0]["loss"]) for _ in range(8)]
    torch.cuda.synchronize()
    plan = dict(optimizer_plan(tr.eng))
    assert "emb" in plan
    dt = (model.encoder.embed_tokens.weight.detach().float() - tok0).abs().sum(1)
    ds = (model.decoder.seg_projection.weight.detach().float() - seg0).abs().sum(1)
    used = torch.zeros(tok0.shape[0], dtype=torch.bool, device=dev)
    used[s["net_input"]["src_tokens"].reshape(-1)] = True
    used[s["net_input"]["prev_output_tokens"][:, 0]] = True
    used[1] = False                                                        # padding_idx
    print("trainable embeddings through the trainer: losses", [round(x, 4) for x in losses], "rows moved", int((dt > 0).sum()), "of", int(used.sum()))
    assert (dt[used] > 0).all() and (dt[~used] == 0).all() and (ds > 0).all()
    assert losses[-1] < losses[0]
    tr.close()
