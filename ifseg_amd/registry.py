"""fairseq plugin registration that degrades to a local registry.

Under the reference's launcher (`train.py --user-dir ofa_module`, ofa_module/__init__.py:1-5)
fairseq is importable and these decorators are fairseq's own
(fairseq/models/__init__.py:110-160, tasks/__init__.py, criterions/__init__.py), so
`--arch segofa_base --task segmentation --criterion seg_criterion` resolve to the
classes of this package.  On a box without fairseq (the GPU box) they fall back to a
dictionary so the same modules import and the bundled harness can look them up.
"""
MODEL_REGISTRY, ARCH_REGISTRY, TASK_REGISTRY, CRITERION_REGISTRY = {}, {}, {}, {}

try:  # pragma: no cover - fairseq is not installed in the build image
    from fairseq.models import register_model as _fs_register_model
    from fairseq.models import register_model_architecture as _fs_register_arch
    from fairseq.tasks import register_task as _fs_register_task
    from fairseq.criterions import register_criterion as _fs_register_criterion
    HAVE_FAIRSEQ = True
except Exception:  # noqa: BLE001
    HAVE_FAIRSEQ = False


def register_model(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        return _fs_register_model(name)(cls) if HAVE_FAIRSEQ else cls
    return deco


def register_model_architecture(model_name, arch_name):
    def deco(fn):
        ARCH_REGISTRY[arch_name] = (model_name, fn)
        return _fs_register_arch(model_name, arch_name)(fn) if HAVE_FAIRSEQ else fn
    return deco


def register_task(name, dataclass=None):
    def deco(cls):
        TASK_REGISTRY[name] = cls
        return _fs_register_task(name, dataclass=dataclass)(cls) if HAVE_FAIRSEQ else cls
    return deco


def register_criterion(name, dataclass=None):
    def deco(cls):
        CRITERION_REGISTRY[name] = cls
        return _fs_register_criterion(name, dataclass=dataclass)(cls) if HAVE_FAIRSEQ else cls
    return deco
