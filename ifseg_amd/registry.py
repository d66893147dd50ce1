"""fairseq plugin registration with dynamic base classes.

Under the reference's launcher (`train.py --user-dir ofa_module`, ofa_module/__init__.py:1-5)
fairseq is importable: the decorators below are then fairseq's own
(fairseq/models/__init__.py:110-160, fairseq/tasks/__init__.py:48-100, fairseq/registry.py:66-104) and
`ModelBase` / `TaskBase` / `CriterionBase` / `DataclassBase` are `BaseFairseqModel` / `FairseqTask` /
`FairseqCriterion` / `FairseqDataclass` -- the registries reject anything else
(`issubclass` checks at fairseq/models/__init__.py:131-136, tasks/__init__.py:72-75, registry.py:75-78).
On a box without fairseq (the GPU box carries only this repository) the same classes are built on
`nn.Module` / `object` and registered in the dictionaries below, so the same modules import and the
bundled harness (`ifseg_amd/trainer.py`, `bench.py`) looks them up there.

Only a *missing* fairseq selects the fallback: any other failure while importing it propagates.
"""
import importlib.util

import torch.nn as nn

MODEL_REGISTRY, ARCH_REGISTRY, TASK_REGISTRY, CRITERION_REGISTRY = {}, {}, {}, {}


def _fairseq_present():
    import sys
    if "fairseq" in sys.modules:
        return True
    try:
        return importlib.util.find_spec("fairseq") is not None
    except (ImportError, ValueError):
        return False


HAVE_FAIRSEQ = _fairseq_present()

if HAVE_FAIRSEQ:
    from fairseq.criterions import FairseqCriterion as CriterionBase
    from fairseq.criterions import register_criterion as _fs_register_criterion
    from fairseq.dataclass import FairseqDataclass as DataclassBase
    from fairseq.models import BaseFairseqModel as ModelBase
    from fairseq.models import register_model as _fs_register_model
    from fairseq.models import register_model_architecture as _fs_register_arch
    from fairseq.tasks import FairseqTask as TaskBase
    from fairseq.tasks import register_task as _fs_register_task
else:
    ModelBase = nn.Module

    class TaskBase:                      # FairseqTask stand-in: the attributes the harness touches
        def __init__(self, cfg=None, **kwargs):
            self.cfg = cfg
            self.datasets, self.dataset_to_epoch_iter, self.state = {}, {}, None

    class CriterionBase:                 # FairseqCriterion stand-in (fairseq_criterion.py:17-24)
        def __init__(self, task):
            self.task = task
            td = getattr(task, "target_dictionary", None)
            self.padding_idx = td.pad() if td is not None else -100

        def __call__(self, *a, **k):
            return self.forward(*a, **k)

    class DataclassBase:                 # FairseqDataclass stand-in (dataclass/configs.py:34-80)
        _name = None

        @classmethod
        def from_namespace(cls, args):
            import dataclasses
            if isinstance(args, cls):
                return args
            kw = {f.name: getattr(args, f.name) for f in dataclasses.fields(cls)
                  if f.init and hasattr(args, f.name)}
            return cls(**kw)


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


def str_bool(x):
    """'true' / 'false' string flags of the reference (criterions/seg_criterion.py:103-110,
    unify_transformer.py:262-267: `--freeze-entire-resnet=true`); real bools pass through."""
    if isinstance(x, bool):
        return x
    if x is None:
        return False
    s = str(x).lower()
    if s == "true":
        return True
    if s == "false":
        return False
    raise ValueError("Unable to recognize string bool input: %s" % x)
