"""MI355X-native hot path of the ligand denoising-diffusion model.

Public names mirror the reference: the model (model1.py), the trainer (train.py) and the
generator (generation.py), over a C-ABI library of hand-written gfx950 kernels.
"""
from .model import ResearchAccurateDiffusionModel, ResearchValidatedDiffusionModel  # noqa: F401
