"""Import shim for the *reference* implementation -- BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE.  Used only by ``oracle/gen_golden.py`` (and the pinning
test that is skipped when ``/root/reference`` is absent) to import the
reference's own ``models/segofa/*.py`` on CPU so that the oracle restatement
(``oracle/segofa_ref.py``) can be pinned against it and golden vectors can be
generated.  Nothing here travels to the GPU box in a usable form: the reference
source tree is not present there, and no product code imports this module.

The reference cannot be imported with a plain ``import fairseq`` in this image
(hydra / omegaconf / bitarray are not installed), so we register light stubs for
those packages and an empty ``fairseq`` package object whose ``__path__`` points
at the vendored tree; every arithmetic op executed afterwards is the
reference's own code (``models/segofa/*.py`` + real ``fairseq/utils.py``,
``fairseq/modules/{layer_norm,gelu,fairseq_dropout,quant_noise}.py``).
"""
import argparse
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("IFSEG_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "segofa"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_DONE = False


def install():
    """Make ``models.segofa.segofa`` of the reference importable."""
    global _DONE
    if _DONE:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    import numpy as np

    for alias, typ in (("float", float), ("int", int), ("bool", bool), ("object", object)):
        if not hasattr(np, alias):
            setattr(np, alias, typ)

    # ---- omegaconf / hydra / bitarray stubs (config plumbing only) ----------
    class _DictConfig(dict):
        pass

    class _OmegaConf:
        @staticmethod
        def create(*a, **k):
            return _DictConfig()

        @staticmethod
        def is_config(x):
            return False

        @staticmethod
        def set_struct(*a, **k):
            pass

        @staticmethod
        def to_container(x, **k):
            return x

    class _open_dict:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    oc = _stub("omegaconf", II=lambda x: None, MISSING="???", DictConfig=_DictConfig,
               OmegaConf=_OmegaConf, open_dict=_open_dict)
    oc._utils = _stub("omegaconf._utils", is_primitive_type=lambda x: True)
    _stub("hydra")

    class _ConfigStore:
        @staticmethod
        def instance():
            return _ConfigStore()

        def store(self, *a, **k):
            pass

    _stub("hydra.core")
    _stub("hydra.core.config_store", ConfigStore=_ConfigStore)

    class _GlobalHydra:
        @staticmethod
        def instance():
            return _GlobalHydra()

        def is_initialized(self):
            return False

    _stub("hydra.core.global_hydra", GlobalHydra=_GlobalHydra)
    _stub("hydra.experimental", compose=lambda *a, **k: None, initialize=lambda *a, **k: None)
    ba = _stub("bitarray", bitarray=object)
    ba.util = _stub("bitarray.util")

    # ---- empty ``fairseq`` package pointing at the vendored tree -------------
    fs_path = os.path.join(REFERENCE_ROOT, "custom_fairseq", "fairseq")
    fairseq = types.ModuleType("fairseq")
    fairseq.__path__ = [fs_path]
    fairseq.__version__ = "1.0.0a0"
    sys.modules["fairseq"] = fairseq
    # same alias dance as fairseq/__init__.py
    import fairseq.distributed.utils as _du  # noqa
    sys.modules["fairseq.distributed_utils"] = _du
    fairseq.distributed_utils = _du
    from fairseq.logging import meters, metrics, progress_bar  # noqa
    sys.modules["fairseq.meters"] = meters
    sys.modules["fairseq.metrics"] = metrics
    sys.modules["fairseq.progress_bar"] = progress_bar
    import fairseq.utils as _u  # noqa
    fairseq.utils = _u
    for sub in ("criterions", "distributed", "models", "modules", "optim", "tasks"):
        importlib.import_module("fairseq." + sub)

    # stubs needed by criterions/seg_criterion.py
    import torch.nn.functional as F

    def _resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
        return F.interpolate(input, size, scale_factor, mode, align_corners)

    _stub("mmseg")
    _stub("mmseg.ops", resize=_resize)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", trunc_normal_=lambda *a, **k: None)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _DONE = True


class FakeDictionary:
    """Minimal stand-in for fairseq's Dictionary: only what build_model reads."""

    def __init__(self, n_base, num_seg_tokens):
        # layout mirrors tasks/mm_tasks/segmentation.py:113-132:
        # [base symbols ... <bin_0> ... <seg_0> .. <seg_nseg>]
        self.n = n_base + num_seg_tokens + 1
        self.seg0 = n_base
        self.bin0 = max(4, n_base - 8)

    def __len__(self):
        return self.n

    def pad(self):
        return 1

    def bos(self):
        return 0

    def eos(self):
        return 2

    def unk(self):
        return 3

    def index(self, sym):
        if sym == "<seg_0>":
            return self.seg0
        if sym == "<bin_0>":
            return self.bin0
        return 3

    def __eq__(self, other):
        return self is other

    def __contains__(self, sym):
        return sym == "<mask>"


def build_reference_model(arch, num_seg_tokens, n_base_vocab, patch_image_size,
                          orig_patch_image_size=None, overrides=None):
    """Build the reference SegOFAModel the way run_scripts/IFSeg/coco_unseen.sh does.

    ``arch`` is 'base' | 'large' | 'tiny' ...; ``overrides`` may set e.g.
    encoder_layers / encoder_embed_dim for the small fixture configuration.
    """
    install()
    from models.segofa import segofa as ref_segofa
    from models.segofa.segofa import SegOFAModel

    # argparse group with SUPPRESS defaults, as fairseq/options.py:140-147 does,
    # so that store_true flags not given stay *absent* (arch fn then applies its
    # own getattr defaults, e.g. no_scale_embedding=True).
    parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS, allow_abbrev=False)
    SegOFAModel.add_args(parser)
    argv = [
        "--encoder-normalize-before", "--decoder-normalize-before",
        "--share-decoder-input-output-embed", "--share-all-embeddings",
        "--layernorm-embedding", "--patch-layernorm-embedding", "--code-layernorm-embedding",
        "--resnet-drop-path-rate=0.0", "--encoder-drop-path-rate=0.0", "--decoder-drop-path-rate=0.0",
        "--dropout=0.0", "--attention-dropout=0.0",
        "--add-type-embedding", "--scale-attn", "--scale-fc", "--scale-heads", "--disable-entangle",
        "--patch-image-size=%d" % patch_image_size,
        "--orig-patch-image-size=%d" % (orig_patch_image_size or patch_image_size),
        "--freeze-encoder-embedding=true", "--freeze-decoder-embedding=true",
        "--freeze-seg-embedding=true", "--freeze-entire-resnet=true",
        "--tie-seg-projection=true", "--decoder-type=surrogate",
        "--decoder-input-type=encoder_output", "--num-seg-tokens=%d" % num_seg_tokens,
    ]
    args, _ = parser.parse_known_args(argv)
    for k, v in (overrides or {}).items():
        setattr(args, k, v)
    args.adaptive_input = False
    args.tie_adaptive_weights = False
    arch_fn = {
        "base": ref_segofa.segofa_base_architecture,
        "large": ref_segofa.segofa_large_architecture,
        "tiny": ref_segofa.ofa_tiny_architecture,
        "medium": ref_segofa.segofa_medium_architecture,
    }[arch]
    arch_fn(args)
    d = FakeDictionary(n_base_vocab, num_seg_tokens)

    class _Task:
        source_dictionary = d
        target_dictionary = d

    model = SegOFAModel.build_model(args, _Task())
    return model, args
