"""One rank of the 2-rank functional run on ONE GPU (tests/test_configs_gpu.py): both ranks use cuda:0, the process
group is gloo (IFSEG_DIST_BACKEND=gloo; the production backend "nccl" = RCCL needs one GPU per rank).
    python _two_rank_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys

rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
from ifseg_amd.criterions import SegCriterion  # noqa: E402
from ifseg_amd.tasks.mm_tasks import SegmentationTask  # noqa: E402
from ifseg_amd.trainer import Trainer  # noqa: E402

task = SegmentationTask(num_seg_tokens=5, patch_image_size=128, arch="segofa_tiny")
torch.manual_seed(rank)                      # DIFFERENT initial weights per rank: the trainer must broadcast rank 0's
model = task.build_model()
model.cfg.dropout = model.cfg.encoder_drop_path_rate = model.cfg.decoder_drop_path_rate = 0.0
tr = Trainer(model, SegCriterion(task, unsupervised_segmentation=False, init_seg_with_text=False), task, device=dev)
assert tr.world == world and tr.eng.grad_ready_hook is not None
sample = task.synthetic_sample(2, dev, seed=50 + rank)
out = {"p16_start": tr.eng.p16.clone().cpu()}
logs = tr.train_step([sample])
torch.cuda.synchronize()
g_first = tr.eng.g16.float().cpu()           # sum over ranks of the first step's gradients
logs2 = tr.train_step([sample])
torch.cuda.synchronize()
out.update(g16=g_first, p16=tr.eng.p16.clone().cpu(), p32=tr.p32.clone().cpu(),
           logs={k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in logs[0].items()})
torch.save(out, os.path.join(outdir, "rank%d.pt" % rank))
dist.barrier()
dist.destroy_process_group()
print("WORKER-OK", rank)
