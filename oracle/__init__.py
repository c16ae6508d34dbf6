"""CPU oracle for the ligand-diffusion hot path.  TEST INFRASTRUCTURE ONLY.

A plain, single-threaded-friendly fp32 restatement (PyTorch CPU ops / numpy / pure
Python) of the reference's score network, EDM loss step, DDPM/DDIM samplers and graph
packing.  Every function cites the reference file:line it follows.  It is pinned by the
fixtures under tests/golden/, which were produced by running the reference's own
model1.py / train.py / generation.py (see tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker or the timed CPU baseline.  Nothing under
diffusion_model_ligandmol_generation_amd/ imports it: the product path is the HIP
library and fails loudly when that library is missing.
"""
from .schedule import cosine_schedule, edm_scalings, edm_sigma_from_u, edm_loss_weight  # noqa: F401
from .connectivity import template_connectivity, fully_connected_edges  # noqa: F401
from .packing import pack_csr, collate  # noqa: F401
from .score_network import score_network_forward, fourier_time_embedding  # noqa: F401
from .sampler import (ddpm_sample, ddim_sample, validity_gate, ddpm_step, chain_edges,  # noqa: F401
                      model_sample_positions, trainer_euler_positions, edm_sigma_steps)
from .loss_step import loss_step, loss_terms  # noqa: F401
from .consistency import consistency_forward  # noqa: F401,E402
