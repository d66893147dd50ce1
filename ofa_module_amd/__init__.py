"""`--user-dir` module for the reference's train.py (replaces ofa_module/__init__.py:1-5).

    python train.py $data --user-dir=/path/to/ofa_module_amd --task=segmentation --arch=segofa_base \
        --criterion=seg_criterion ...          (the flag list of run_scripts/IFSeg/coco_unseen.sh:73-137, unchanged)

Registers the MI355X-native model ("segofa" + segofa_{tiny,medium,base,large,huge}), task ("segmentation") and
criterion ("seg_criterion") in fairseq's own registries.  To keep the reference's data pipeline / task / criterion and
swap only the model, import `data, tasks, criterions, utils` of the reference here instead of the last two lines
 -- the two sets of names cannot be registered together.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

import ifseg_amd.models  # noqa: E402,F401
import ifseg_amd.tasks.mm_tasks  # noqa: E402,F401
import ifseg_amd.criterions  # noqa: E402,F401
