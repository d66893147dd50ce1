"""Plugin boundary under the REFERENCE's own fairseq.

Runs in a subprocess (fairseq's registries are process-global and the rest of the CPU suite imports `ifseg_amd`
without fairseq): the reference's vendored fairseq is made importable through oracle/_refshim.py, the product is
imported through its `--user-dir` module, and fairseq's OWN `options.parse_args_and_arch` parses the exact command line
of run_scripts/IFSeg/coco_unseen.sh:73-137.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("IFSEG_REFERENCE_ROOT", "/root/reference")

CHILD = r'''
import os, subprocess, sys
ROOT, REF = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import _refshim
_refshim.install()
# ---- the command line of the shipped recipe, expanded by bash itself
sh = open(os.path.join(REF, "run_scripts/IFSeg/coco_unseen.sh")).read()
head, tail = sh.split("python3 -m torch.distributed.launch")
tail = tail.split("./train.py", 1)[1].rsplit("2>&1", 1)[0]
head = "\n".join(l for l in head.splitlines() if not l.startswith("mkdir"))
out = subprocess.run(["bash", "-c", head + "\nprintf '%s\\n' " + tail], capture_output=True, text=True, check=True).stdout
argv = [a for a in out.split("\n") if a != ""]
assert argv[0].endswith(".tsv") and "--arch=segofa_base" in argv and "--scale-heads" in argv, argv[:5]
argv = [("--user-dir=" + os.path.join(ROOT, "ofa_module_amd")) if a.startswith("--user-dir=") else a for a in argv]
from fairseq import options
parser = options.get_training_parser()
args = options.parse_args_and_arch(parser, input_args=argv)          # imports the user dir, adds model/task/criterion flags
from fairseq.models import ARCH_MODEL_REGISTRY, ARCH_CONFIG_REGISTRY, MODEL_REGISTRY, BaseFairseqModel
from fairseq.tasks import TASK_REGISTRY, FairseqTask
from fairseq.criterions import CRITERION_REGISTRY, FairseqCriterion
import ifseg_amd.registry as R
assert R.HAVE_FAIRSEQ
M = ARCH_MODEL_REGISTRY["segofa_base"]
assert M.__module__.startswith("ifseg_amd.") and issubclass(M, BaseFairseqModel) and MODEL_REGISTRY["segofa"] is M
for a in ("segofa_tiny", "segofa_medium", "segofa_base", "segofa_large", "segofa_huge"):
    assert a in ARCH_CONFIG_REGISTRY and ARCH_MODEL_REGISTRY[a] is M
T = TASK_REGISTRY["segmentation"]; Cr = CRITERION_REGISTRY["seg_criterion"]
assert T.__module__.startswith("ifseg_amd.") and issubclass(T, FairseqTask)
assert Cr.__module__.startswith("ifseg_amd.") and issubclass(Cr, FairseqCriterion)
# ---- what the recipe's flags became
assert args.arch == "segofa_base" and args.task == "segmentation" and args.criterion == "seg_criterion"
assert args.dropout == 0.1 and args.encoder_drop_path_rate == 0.1 and args.decoder_drop_path_rate == 0.1
assert args.scale_attn and args.scale_fc and args.scale_heads and args.disable_entangle and args.share_all_embeddings
assert args.freeze_entire_resnet == "true" and args.tie_seg_projection == "true" and args.decoder_input_type == "encoder_output"
assert args.num_seg_tokens == 15 and args.patch_image_size == 512 and args.unsupervised_segmentation == "true"
assert args.resnet_iters == 25 and args.resnet_topk == 3 and args.category_list.startswith("frisbee")
assert args.encoder_embed_dim == 768 and args.encoder_layers == 6 and args.resnet_type == "resnet101"   # architecture fn ran
# ---- build through the registered classes with the parsed namespace (fairseq's own build_criterion logic)
from ifseg_amd.tasks.mm_tasks.segmentation import SegmentationConfig, SizeDictionary
tcfg = SegmentationConfig.from_namespace(args)
assert tcfg.num_seg_tokens == 15 and tcfg.prompt_prefix.startswith("what is the segmentation map")
d = SizeDictionary(59457, 15)
task = T(tcfg, d, d)
model = M.build_model(args, task)
assert isinstance(model, BaseFairseqModel)
assert model.cfg.dropout == 0.1 and model.cfg.encoder_drop_path_rate == 0.1 and model.cfg.vocab_size == 59458
assert sum(p.numel() for p in model.parameters()) == 183242728
from ifseg_amd.criterions.seg_criterion import SegCriterionConfig
ccfg = SegCriterionConfig.from_namespace(args)
crit = Cr.build_criterion(ccfg, task)
assert isinstance(crit, FairseqCriterion) and crit.unsupervised_segmentation and crit.resnet_iters == 25
assert crit.num_seg == 15 and len(crit.id2rawtext) == 15 and crit.seg_id_offset == 59457
# refused, not ignored: a value the HIP path does not implement
import copy
bad = copy.copy(args); bad.encoder_layerdrop = 0.1
try:
    M.build_model(bad, task)
except NotImplementedError:
    pass
else:
    raise AssertionError("encoder_layerdrop > 0 was accepted")
ok = copy.copy(args); ok.attention_dropout = 0.1; ok.activation_dropout = 0.2
mk = M.build_model(ok, task)
assert mk.cfg.attention_dropout == 0.1 and mk.cfg.activation_dropout == 0.2        # built in round 5
print("PLUGIN-OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "custom_fairseq", "fairseq")), reason="reference tree absent")
def test_registers_under_reference_fairseq_and_parses_the_recipe():
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, REF], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "PLUGIN-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
