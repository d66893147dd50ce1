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
