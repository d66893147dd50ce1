"""CPU: the plugin surface -- state_dict contract, registration, C-ABI symbols -- without
launching any kernel (no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

import segofa_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_contract_matches_oracle_spec():
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    ocfg = O.fixture_config()
    m = SegOFAModel(make_config("segofa_tiny", embed_dim=128, ffn_dim=256, heads=2, enc_layers=2, dec_layers=2,
                                resnet_layers=(3, 4, 6), num_seg_tokens=5, vocab_size=101, patch_image_size=128,
                                orig_patch_image_size=128))
    sd = m.state_dict()
    spec = O.state_dict_spec(ocfg)
    derived = {k for k in sd if k.endswith(("_rp_bucket", ".version", "image_position_idx", "_id_offset", "region_prefix"))}
    assert set(sd) - derived == set(spec)
    for k, (shape, _) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    # ties (unify_transformer.py:349-361, decoder_module.py:134-137)
    assert m.encoder.embed_tokens.weight is m.decoder.embed_tokens.weight is m.encoder.embed_tokens_bag.weight
    assert m.decoder.seg_projection.weight is m.encoder.seg_embed_tokens.weight is m.decoder.seg_embed_tokens.weight
    assert torch.equal(sd["decoder.seg_rp_bucket"], O.make_image_bucket_position(8, 15 * 15 + 3))
    assert torch.equal(sd["encoder.token_rp_bucket"], O.make_token_bucket_position(256))
    assert torch.equal(sd["encoder.image_rp_bucket"], O.make_image_bucket_position(42, 83 * 83 + 3))


def test_base_param_counts():
    """183.24 M params / 109.33 M trainable, 889 state_dict entries."""
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    m = SegOFAModel(make_config("segofa_base"))
    assert len(m.state_dict()) == 889
    assert sum(p.numel() for p in m.parameters()) == 183242728
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 109327656


def test_registry_names():
    import ifseg_amd.models  # noqa: F401
    import ifseg_amd.criterions  # noqa: F401
    import ifseg_amd.tasks.mm_tasks  # noqa: F401
    from ifseg_amd import registry
    assert "segofa" in registry.MODEL_REGISTRY
    for a in ("segofa_tiny", "segofa_medium", "segofa_base", "segofa_large", "segofa_huge"):
        assert a in registry.ARCH_REGISTRY
    assert "segmentation" in registry.TASK_REGISTRY and "seg_criterion" in registry.CRITERION_REGISTRY


def test_cpu_forward_fails_loudly():
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    m = SegOFAModel(make_config("segofa_tiny", embed_dim=128, ffn_dim=256, heads=2, enc_layers=1, dec_layers=1,
                                resnet_layers=(1, 1, 1), num_seg_tokens=5, vocab_size=101, patch_image_size=128,
                                orig_patch_image_size=128))
    with pytest.raises(RuntimeError):
        m(src_tokens=torch.zeros(1, 4, dtype=torch.long), patch_images=torch.zeros(1, 3, 128, 128))


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads on CPU and exports each function include/ifseg_hip.h declares."""
    from ifseg_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        from ifseg_amd.build import build
        build(verbose=False)
    lib = hip.lib()
    hdr = open(os.path.join(ROOT, "include", "ifseg_hip.h")).read()
    names = set(re.findall(r"\bint\s+(ifseg_\w+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    lib.ifseg_abi_version.restype = ctypes.c_int
    assert lib.ifseg_abi_version() == int(re.search(r"#define IFSEG_ABI_VERSION (\d+)", hdr).group(1))


def test_library_has_no_cross_half_packed_fp32_instruction():
    """DESIGN.md "Round 3" (6): `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the low lane reading the HIGH half of its
    second source) returned wrong low results next to a running GEMM; clang's SLP vectoriser emitted it in exactly one kernel
    (ffn_ln_coef_kernel), whose source is now built without that pass.  The shipped library must not contain the form
    anywhere (tools/check_isa.py disassembles every gfx950 code object of the built .so)."""
    import importlib.util
    from ifseg_amd import build
    lib = build.build(verbose=False)
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(os.path.dirname(__file__), "..", "tools", "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("no llvm-objdump")
    nobj, hits = mod.scan(lib)
    assert nobj >= 1, "no gfx950 code object found in %s" % lib
    assert not hits, hits[:5]


def test_arena_lays_every_linear_out_weight_then_bias():
    """The gradient arena keeps each (fused) Linear's bias right behind its weight(s): dW and db are then one
    contiguous range, which lets the weight-gradient GEMM's split-K reduction write both (IFSEG_GEMM_COLSUM), and
    q|k|v (self-attention) / k|v (cross-attention) weights are adjacent so one GEMM serves them."""
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    m = SegOFAModel(make_config("segofa_tiny", embed_dim=128, ffn_dim=256, heads=2, enc_layers=2, dec_layers=2,
                                resnet_layers=(3, 4, 6), num_seg_tokens=5, vocab_size=101, patch_image_size=128,
                                orig_patch_image_size=128))
    eng = m.engine
    eng.pack(torch.device("cpu"))
    off, shp = eng.offs, eng.shapes
    numel = lambda n: int(torch.tensor(shp[n]).prod())

    def adjacent(*names):
        for a, b in zip(names[:-1], names[1:]):
            assert off[b] == off[a] + numel(a), (a, b, off[a], numel(a), off[b])

    for l in range(2):
        e, d = "encoder.layers.%d." % l, "decoder.layers.%d." % l
        for p in (e + "self_attn", d + "self_attn"):
            adjacent(p + ".q_proj.weight", p + ".k_proj.weight", p + ".v_proj.weight", p + ".q_proj.bias", p + ".k_proj.bias",
                     p + ".v_proj.bias")
            adjacent(p + ".out_proj.weight", p + ".out_proj.bias")
        c = d + "encoder_attn"
        adjacent(c + ".q_proj.weight", c + ".q_proj.bias")
        adjacent(c + ".k_proj.weight", c + ".v_proj.weight", c + ".k_proj.bias", c + ".v_proj.bias")
        adjacent(c + ".out_proj.weight", c + ".out_proj.bias")
        for p in (e, d):
            adjacent(p + "fc1.weight", p + "fc1.bias")
            adjacent(p + "fc2.weight", p + "fc2.bias")
    adjacent("encoder.pos_q_linear.weight", "encoder.pos_k_linear.weight", "encoder.pos_q_linear.bias", "encoder.pos_k_linear.bias")
    # the nn.Parameters are views into the arena (state_dict contract is unchanged by the layout)
    w = dict(m.named_parameters())["decoder.layers.1.encoder_attn.q_proj.bias"]
    assert w.data_ptr() == eng.p16.data_ptr() + 2 * off["decoder.layers.1.encoder_attn.q_proj.bias"]


def _ofa_like_checkpoint(sd):
    """what oracle/gen_golden.py:case_upgrade feeds the reference: token table one row short, no seg-token tables, a
    stale decoder.output_projection, image-position tables 40 rows short"""
    ck = {k: v.clone() for k, v in sd.items() if "seg_embed_tokens" not in k and "seg_projection" not in k
          and "embed_tokens_bag" not in k}
    for k in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight"):
        ck[k] = ck[k][:-1].clone()
    ck["decoder.output_projection.weight"] = ck["decoder.embed_tokens.weight"].clone()
    for k in ("encoder.embed_image_positions.weight", "decoder.embed_image_positions.weight"):
        ck[k] = ck[k][:-40].clone()
    return ck


def test_checkpoint_upconversion_matches_reference_golden(golden_dir):
    """models/segofa/segofa.py:197-299 (+ encoder_module.py:943-987, decoder_module.py:892-940): the same OFA-style
    checkpoint through the product's upgrade_state_dict_named gives the keys, shapes and the N(0, C^-0.5) rows the
    REFERENCE produced (tests/golden/fixture_upgrade.npz; same torch seed, same draw order), and then loads strictly."""
    import numpy as np
    from ifseg_amd.models.segofa import SegOFAModel, make_config
    g = np.load(os.path.join(golden_dir, "fixture_upgrade.npz"))
    ocfg = O.fixture_config()
    m = SegOFAModel(make_config("segofa_tiny", embed_dim=128, ffn_dim=256, heads=2, enc_layers=2, dec_layers=2,
                                resnet_layers=(3, 4, 6), num_seg_tokens=5, vocab_size=101, patch_image_size=128,
                                orig_patch_image_size=128))
    ck = _ofa_like_checkpoint(O.procedural_state_dict(ocfg))
    assert ck["encoder.embed_tokens.weight"].shape[0] == 100
    torch.manual_seed(7)
    m.upgrade_state_dict_named(ck, "")
    assert sorted(ck.keys()) == list(g["keys"])
    assert [str(tuple(ck[k].shape)) for k in sorted(ck.keys())] == list(g["shapes"])
    for key, name in (("enc_tok_tail", "encoder.embed_tokens.weight"), ("dec_tok_tail", "decoder.embed_tokens.weight")):
        assert torch.allclose(ck[name][-2:], torch.from_numpy(g[key]), atol=1e-7), name
    for key, name in (("enc_ipos_tail", "encoder.embed_image_positions.weight"), ("dec_ipos_tail", "decoder.embed_image_positions.weight")):
        assert torch.allclose(ck[name][-41:], torch.from_numpy(g[key]), atol=1e-7), name
    ck2 = _ofa_like_checkpoint(O.procedural_state_dict(ocfg))
    torch.manual_seed(7)
    missing, unexpected = m.load_state_dict(ck2, strict=True)          # load_state_dict up-converts first (fairseq_model.py:103-118)
    assert not missing and not unexpected
    assert torch.allclose(m.encoder.embed_tokens.weight[-2:], torch.from_numpy(g["enc_tok_tail"]), atol=1e-7)
    # a table with MORE rows than the dictionary + no <mask>: the extra row is dropped (segofa.py:255-265)
    ck3 = _ofa_like_checkpoint(O.procedural_state_dict(ocfg))
    for k in ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight"):
        ck3[k] = torch.cat([ck3[k], torch.ones(2, 128)])
    m.encoder.dictionary = ()                                            # "<mask>" not in dictionary
    m.upgrade_state_dict_named(ck3, "")
    assert ck3["encoder.embed_tokens.weight"].shape[0] == 101 and ck3["decoder.embed_tokens.weight"].shape[0] == 101


def test_learning_rate_schedule_matches_reference_golden(golden_dir):
    """the lr of updates 1..3 as the reference's CosineLRSchedule produced them (fixture_optim.npz): update k runs at cosine(k-1) -- the peak lr first (train.py:304 begin_epoch)"""
    import math
    import numpy as np
    from ifseg_amd.trainer import Trainer
    g = np.load(os.path.join(golden_dir, "fixture_optim.npz"))
    t = Trainer.__new__(Trainer)
    t.lr0, t.min_lr, t.max_update = float(g["lr0"]), 0.0, int(g["total_updates"])
    for k, want in enumerate(g["lrs"]):
        t.num_updates = k
        assert abs(t.get_lr() - float(want)) <= 1e-15, (k, t.get_lr(), want)
    t.num_updates = 1000
    assert abs(t.get_lr() - 0.5 * t.lr0 * (1 + math.cos(math.pi * 0.5))) < 1e-12


def test_recipe_flags_reach_the_config_without_fairseq():
    """build_model maps the namespace onto SegOFAConfig (dropout, drop-path, buckets) and refuses what the HIP path lacks"""
    from ifseg_amd.models.segofa.segofa import SegOFAModel, recipe_args
    from ifseg_amd.tasks.mm_tasks import SegmentationTask
    task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")
    m = task.build_model()
    assert (m.cfg.dropout, m.cfg.encoder_drop_path_rate, m.cfg.decoder_drop_path_rate) == (0.1, 0.1, 0.1)
    assert m.cfg.num_seg_tokens == 5 and m.cfg.resnet_layers == (3, 4, 6) and m.cfg.embed_dim == 256
    a = recipe_args("segofa_tiny", num_seg_tokens=5, patch_image_size=128, orig_patch_image_size=128, dropout=0.0,
                    image_bucket_size=20, attn_scale_factor=4.0)
    m2 = SegOFAModel.build_model(a, task)
    assert m2.cfg.dropout == 0.0 and m2.cfg.image_bucket_size == 20 and m2.cfg.attn_scale_factor == 4.0
    assert SegOFAModel.build_model(recipe_args("segofa_tiny", num_seg_tokens=5, attention_dropout=0.1), task).cfg.attention_dropout == 0.1
    assert SegOFAModel.build_model(recipe_args("segofa_tiny", num_seg_tokens=5, relu_dropout=0.2), task).cfg.activation_dropout == 0.2
    for bad in (dict(encoder_layerdrop=0.1), dict(scale_attn=False), dict(freeze_entire_resnet="false"),
                dict(decoder_input_type="encoder_input"), dict(tie_seg_projection="false")):
        with pytest.raises(NotImplementedError):
            SegOFAModel.build_model(recipe_args("segofa_tiny", num_seg_tokens=5, **bad), task)


def test_fixed_length_beam_search_is_exact_k_best():
    """ifseg_amd.sequence_generator.beam_search_independent (fairseq BeamSearch semantics on the surrogate decoder's
    per-step independent distributions): against brute-force enumeration of all V^T sequences."""
    import itertools
    from ifseg_amd.sequence_generator import beam_search_independent
    g = torch.Generator().manual_seed(3)
    B, T, V, beam = 2, 5, 4, 3
    lp = torch.log_softmax(torch.randn(B, T, V, generator=g), -1)
    tokens, scores = beam_search_independent(lp, beam)
    for b in range(B):
        allseq = sorted(((sum(lp[b, t, s[t]].item() for t in range(T)), s) for s in itertools.product(range(V), repeat=T)), reverse=True)
        for k in range(beam):
            assert tuple(tokens[b, k].tolist()) == allseq[k][1]
            assert abs(scores[b, k, -1].item() - allseq[k][0]) < 1e-5
        assert torch.equal(tokens[b, 0], lp[b].argmax(-1))               # best beam = per-position argmax


def test_torch_library_ops_are_registered_with_fake_kernels():
    """north_star / SURVEY 8b: the hot-path kernels are visible to the dispatcher as torch.ops.ifseg.* (ifseg_amd/ops.py) with
    meta implementations; there is no CPU kernel behind them (the dispatcher raises instead of falling back)."""
    import ifseg_amd.ops  # noqa: F401
    for name in ("linear", "linear_bwd", "layer_norm", "layer_norm_bwd", "bias_attention", "bias_attention_bwd"):
        assert hasattr(torch.ops.ifseg, name), name
    bf = torch.bfloat16
    x, w = torch.empty(4, 6, 128, device="meta", dtype=bf), torch.empty(256, 128, device="meta", dtype=bf)
    assert torch.ops.ifseg.linear(x, w, None).shape == (4, 6, 256)
    y, mu, rs = torch.ops.ifseg.layer_norm(x, torch.empty(128, device="meta"), torch.empty(128, device="meta"), 1e-5, False)
    assert y.shape == x.shape and mu.shape == (24,) and rs.shape == (24,)
    q = torch.empty(2, 70, 128, device="meta", dtype=bf)
    pq = torch.empty(70, 128, device="meta", dtype=bf)
    o, lse = torch.ops.ifseg.bias_attention(q, q, q, pq, pq, torch.empty(2, device="meta"), None, None, None, None, 70, 0, 0, False)
    assert o.shape == q.shape and lse.shape == (2, 2, 70)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.ifseg.linear(torch.zeros(2, 8, dtype=bf), torch.zeros(8, 8, dtype=bf), None)
