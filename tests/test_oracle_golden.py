"""CPU: the oracle restatement against the reference-generated golden vectors.

The vectors in tests/golden/*.npz are the REFERENCE's outputs (produced by
oracle/gen_golden.py importing /root/reference in the build container); the
oracle regenerates weights and inputs procedurally and must reproduce them.
"""
import os

import numpy as np
import pytest
import torch

import segofa_ref as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_fixture_logits_loss_and_grads(golden_dir):
    g = _load(golden_dir, "fixture_train.npz")
    cfg = O.fixture_config()
    sd = O.procedural_state_dict(cfg)
    batch = O.synthetic_batch(cfg, int(g["batch_size"]), int(g["src_len"]))
    keys = [k[5:] for k in g.files if k.startswith("grad:")]
    sd = dict(sd)
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], False)
    hp, wp = extra["encoder_returns"]["image_embed_shape"]
    loss, s, t = O.seg_loss(cfg, logits, batch["target"], hp, wp, 128, 128)
    loss.backward()
    assert np.abs(logits.detach().numpy() - g["logits_causal"]).max() <= 1e-5
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    for k in keys:
        ref = torch.from_numpy(g["grad:" + k])
        rel = (sd[k].grad - ref).norm() / ref.norm()
        assert rel <= 1e-5, (k, rel)
    ai, ap, al, au = O.seg_metric(s.detach(), t, cfg.num_seg_tokens)
    assert np.array_equal(ai.numpy(), g["area_intersect"])
    assert np.array_equal(ap.numpy(), g["area_pred_label"])
    assert np.array_equal(al.numpy(), g["area_label"])
    assert np.array_equal(au.numpy(), g["area_union"])
    with torch.no_grad():
        full = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], True)[0]
    assert np.abs(full.numpy() - g["logits_full"]).max() <= 1e-5
    # causal and full-context must genuinely differ (mask is exercised)
    assert np.abs(g["logits_full"] - g["logits_causal"]).max() > 1e-3


@pytest.mark.parametrize("name", ["fixture_padded.npz", "fixture_resize_train.npz", "fixture_resize_padded.npz"])
def test_fixture_padded_and_resized_training_cases(golden_dir, name):
    """The training goldens of the padded-prompt, resized-grid and padded + resized cases (the REFERENCE's logits, loss and
    gradients): the oracle reproduces them on the CPU -- the same goldens pin the HIP path in tests/test_model_gpu.py."""
    g = _load(golden_dir, name)
    cfg = O.fixture_config()
    sd = dict(O.procedural_state_dict(cfg))
    hw = tuple(int(v) for v in g["image_hw"]) if "image_hw" in g.files else None
    batch = O.synthetic_batch(cfg, int(g["batch_size"]), int(g["src_len"]), **({"image_hw": hw} if hw else {}))
    if "src_tokens" in g.files:
        batch["src_tokens"] = torch.from_numpy(g["src_tokens"])
    keys = [k[5:] for k in g.files if k.startswith("grad:")]
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], False)
    hp, wp = extra["encoder_returns"]["image_embed_shape"]
    ih, iw = hw if hw else (128, 128)
    loss, s, t = O.seg_loss(cfg, logits, batch["target"], hp, wp, ih, iw)
    loss.backward()
    assert np.abs(logits.detach().numpy() - g["logits_causal"]).max() <= 1e-5
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    assert len(keys) > 10
    for k in keys:
        ref = torch.from_numpy(g["grad:" + k])
        assert (sd[k].grad - ref).norm() / ref.norm() <= 1e-5, k
    with torch.no_grad():
        full = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], True)[0]
    assert np.abs(full.numpy() - g["logits_full"]).max() <= 1e-5


def test_fixture_resize_path(golden_dir):
    """P != orig grid: positional-embedding + double-bilinear rel-pos resize
    (encoder_module.py:360-368,802-808; decoder_module.py:541-548,603-627)."""
    g = _load(golden_dir, "fixture_resize.npz")
    cfg = O.fixture_config()
    sd = O.procedural_state_dict(cfg)
    hw = tuple(int(v) for v in g["image_hw"])
    batch = O.synthetic_batch(cfg, 1, int(g["src_len"]), image_hw=hw, seed=int(g["seed"]))
    with torch.no_grad():
        logits = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"], None, batch["patch_masks"], False)[0]
    assert logits.shape == (1, (hw[0] // 16) * (hw[1] // 16) + 1, cfg.num_seg_tokens)
    assert np.abs(logits.numpy() - g["logits_causal"]).max() <= 1e-5


def test_bucket_tables_closed_form():
    """The image bucket index is closed-form in (dy, dx) -- what the HIP kernels
    compute in-register instead of reading an index tensor."""
    for b in (4, 32, 42):
        n = (2 * b - 1) ** 2 + 3
        t = O.make_image_bucket_position(b, n)
        ys, xs = torch.meshgrid(torch.arange(b), torch.arange(b), indexing="ij")
        pid = (xs + ys * b + 1).reshape(-1)
        yi, xi = ys.reshape(-1), xs.reshape(-1)
        closed = (yi[:, None] - yi[None, :] + b - 1) * (2 * b - 1) + (xi[:, None] - xi[None, :] + b - 1)
        assert torch.equal(t[pid][:, pid], closed)
        assert (t[0, 1:] == n - 3).all() and (t[1:, 0] == n - 2).all() and t[0, 0] == n - 1
    tok = O.make_token_bucket_position(256)
    i = torch.arange(300)
    d = i[:, None] - i[None, :]
    near = d.abs() <= 128
    assert torch.equal(tok[:300, :300][near], (d + 255)[near])
    assert tok.min() >= 0 and tok.max() <= 510


def test_padding_mask_changes_only_padded_keys():
    cfg = O.fixture_config()
    sd = O.procedural_state_dict(cfg)
    batch = O.synthetic_batch(cfg, 2, 12)
    src = batch["src_tokens"].clone()
    src[1, -3:] = O.PAD
    with torch.no_grad():
        a = O.segofa_forward(sd, cfg, src, batch["patch_images"])[0]
    assert torch.isfinite(a).all()


@pytest.mark.slow
def test_base_config1_against_reference(golden_dir):
    """BASELINE config 1 shapes (B=1 here to stay in CPU-minutes): Base, 512x512, nseg 15."""
    g = _load(golden_dir, "base_c1.npz")
    cfg = O.base_config()
    sd = O.procedural_state_dict(cfg)
    batch = O.synthetic_batch(cfg, 1, int(g["src_len"]))
    with torch.no_grad():
        logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        loss, _, _ = O.seg_loss(cfg, logits, batch["target"], 32, 32, 512, 512)
    assert np.abs(logits.numpy() - g["logits_causal"]).max() <= 1e-4
    assert abs(loss.item() - float(g["loss"])) <= 1e-5


def test_fixture_imfree_branch(golden_dir):
    """Image-free branch (aux_input -> encode_with_artificial_image -> causal decoder, compute_imfree_loss):
    the restatement against the reference's logits / loss / gradients."""
    g = _load(golden_dir, "fixture_imfree.npz")
    cfg = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
    sd = dict(O.procedural_state_dict(cfg))
    batch = O.synthetic_aux_batch(cfg, int(g["batch_size"]), int(g["src_len"]))
    keys = [k[5:] for k in g.files if k.startswith("grad:")]
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    logits, _ = O.segofa_forward_imfree(sd, cfg, batch["aux_input"])
    loss = O.imfree_loss(cfg, logits, batch["text2seg_target"])
    loss.backward()
    assert np.abs(logits.detach().numpy() - g["logits"]).max() <= 1e-5
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    for k in keys:
        ref = torch.from_numpy(g["grad:" + k])
        rel = (sd[k].grad - ref).norm() / ref.norm()
        assert rel <= 1e-5, (k, rel)
    # EmbeddingBag plumbing: ragged bags, per-sample offsets rebased onto the pad-stripped stream
    ids, ends = batch["aux_input"]["patch_images"], batch["aux_input"]["patch_masks"]
    starts = O.embed_bag_offsets(ends, ids.shape[0])
    flat = ids[ids != O.PAD]
    ref = torch.nn.functional.embedding_bag(flat, sd["encoder.embed_tokens.weight"].detach(), starts, mode="mean")
    assert torch.allclose(O.embed_bag_mean(sd["encoder.embed_tokens.weight"].detach(), flat, starts), ref, atol=1e-6)


def test_fixture_imfree_branch_with_padded_prompts(golden_dir):
    """the image-free step with prompts of different lengths: the oracle reproduces the reference's logits, loss and gradients"""
    g = _load(golden_dir, "fixture_imfree_padded.npz")
    cfg = O.fixture_config(patch_image_size=512, orig_patch_image_size=512)
    sd = dict(O.procedural_state_dict(cfg))
    aux = O.synthetic_aux_batch(cfg, int(g["batch_size"]), int(g["src_len"]))
    aux["aux_input"]["src_tokens"] = torch.from_numpy(g["src_tokens"])
    keys = [k[5:] for k in g.files if k.startswith("grad:")]
    for k, (_, kind) in O.state_dict_spec(cfg).items():
        if kind.startswith("alias:"):
            sd[k] = sd[kind[6:]]
    for k in keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    logits, _ = O.segofa_forward_imfree(sd, cfg, aux["aux_input"])
    loss = O.imfree_loss(cfg, logits, aux["text2seg_target"])
    loss.backward()
    assert np.abs(logits.detach().numpy() - g["logits"]).max() <= 1e-5 and abs(loss.item() - float(g["loss"])) <= 1e-6
    assert len(keys) > 5
    for k in keys:
        ref = torch.from_numpy(g["grad:" + k])
        assert (sd[k].grad - ref).norm() / ref.norm() <= 1e-5, k


def test_fixture_eval_branch(golden_dir):
    """Eval branch of the criterion: top-k neighbour smoothing + metrics at the original image resolution
    (seg_criterion.py:197-213,289-347), restatement vs the reference's histograms and loss."""
    g = _load(golden_dir, "fixture_eval.npz")
    cfg = O.fixture_config()
    sd = O.procedural_state_dict(cfg)
    batch = O.synthetic_batch(cfg, 1, 12, seed=777)
    sd = O.diversify_seg_projection(sd, cfg, batch)
    ori = torch.from_numpy(g["ori"])
    with torch.no_grad():
        logits, extra = O.segofa_forward(sd, cfg, batch["src_tokens"], batch["patch_images"])
        hp, wp = extra["encoder_returns"]["image_embed_shape"]
        prob = O.neighbour_smoothing(logits, extra["encoder_returns"]["image_embed_before_proj"], int(g["iters"]), int(g["topk"]))
        loss, hist = O.seg_eval(cfg, logits, ori, hp, wp)
        _, hist_pp = O.seg_eval(cfg, prob, ori, hp, wp)
    assert np.abs(logits.numpy() - g["logits"]).max() <= 1e-5
    assert abs(loss.item() - float(g["loss"])) <= 1e-6
    for name, a, b in zip(("area_intersect", "area_pred_label", "area_label", "area_union"), hist, hist_pp):
        assert np.array_equal(a.numpy(), g[name]), name
        assert np.array_equal(b.numpy(), g[name + "_pp"]), name
    assert not np.array_equal(g["area_pred_label"], g["area_pred_label_pp"])     # the smoothing changes predictions
