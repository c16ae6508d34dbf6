"""Two training steps (EDM noise/loss step, score-network forward with saves, backward, AdamW) at
the benchmark shape (1024 x 29-atom template ligands): the workload for rocprofv3 --pmc passes
over the backward kernels.
Usage: rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o pmc -- python3 tools/pmc_train.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.train import ResearchAccurateTrainer
from diffusion_model_ligandmol_generation_amd.engine import template_graph

B, n = 1024, 29
torch.manual_seed(0)
dev = torch.device("cuda")
model = ResearchAccurateDiffusionModel(max_atoms=100).to(dev)
trainer = ResearchAccurateTrainer(model, [], [], [])
g, ei, _ = template_graph([n] * B, dev)


class Batch:
    pass


b = Batch()
b.x = torch.nn.functional.one_hot(torch.randint(0, 10, (B * n,), device=dev), 119).float()
b.pos = torch.randn(B * n, 3, device=dev) * 1.5
b.edge_index = ei
b.batch = torch.arange(B, device=dev).repeat_interleave(n)
for _ in range(2):
    trainer.train_step(b)
torch.cuda.synchronize()
print("ok")
