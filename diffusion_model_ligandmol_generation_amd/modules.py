"""Parameter containers with the reference's module tree.

These classes hold weights only: same submodule names, shapes, registration order and
default initialisers as model1.py, so that (a) ``state_dict`` keys match and reference
checkpoints (train.py:741-775) load unchanged, and (b) constructing the model after
``torch.manual_seed(s)`` yields the very same weights as the reference does.  None of
their ``forward`` methods is used on the hot path; the compute happens in the HIP
library, which reads the parameters through flat device pointers.
"""
import torch
import torch.nn as nn


class EDMPreconditioning(nn.Module):
    """Constants of model1.py:17-45 plus host-side scalar helpers."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, rho=7.0):
        super().__init__()
        self.sigma_min, self.sigma_max, self.sigma_data, self.rho = sigma_min, sigma_max, sigma_data, rho

    def get_scalings(self, sigma):                                   # model1.py:27-34
        sigma = sigma.view(-1, 1) if sigma.dim() == 1 else sigma
        sd = self.sigma_data
        c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
        c_out = sigma * sd / (sigma ** 2 + sd ** 2).sqrt()
        c_in = 1 / (sd ** 2 + sigma ** 2).sqrt()
        c_noise = sigma.log() / 4
        return c_skip.squeeze(), c_out.squeeze(), c_in.squeeze(), c_noise.squeeze()

    def sample_sigma(self, batch_size, device):                      # model1.py:36-41
        u = torch.rand(batch_size, device=device)
        return self.sigma_from_u(u)

    def sigma_from_u(self, u):
        a, b = self.sigma_max ** (1 / self.rho), self.sigma_min ** (1 / self.rho)
        return (a + u * (b - a)) ** self.rho

    def loss_weighting(self, sigma):                                 # model1.py:43-45
        return (sigma ** 2 + self.sigma_data ** 2) / (sigma * self.sigma_data) ** 2


class E3EquivariantMessagePassing(nn.Module):
    """Weights of model1.py:47-84."""

    def __init__(self, hidden_dim, num_radial=32):
        super().__init__()
        self.hidden_dim, self.num_radial = hidden_dim, num_radial
        self.radial_basis = nn.Sequential(nn.Linear(1, num_radial), nn.SiLU(), nn.Linear(num_radial, num_radial))
        self.edge_network = nn.Sequential(
            nn.Linear(hidden_dim * 2 + num_radial, hidden_dim), nn.SiLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim))
        self.node_network = nn.Sequential(
            nn.Linear(hidden_dim * 2, hidden_dim), nn.SiLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim))
        self.coord_network = nn.Sequential(
            nn.Linear(hidden_dim, hidden_dim // 2), nn.SiLU(), nn.Linear(hidden_dim // 2, 1), nn.Tanh())


class GraphDiTLayer(nn.Module):
    """Weights of model1.py:245-277."""

    def __init__(self, hidden_dim, num_heads, dropout=0.1):
        super().__init__()
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_dim, 6 * hidden_dim, bias=True))
        self.attention = nn.MultiheadAttention(embed_dim=hidden_dim, num_heads=num_heads,
                                               dropout=dropout, batch_first=True)
        self.ff_network = nn.Sequential(
            nn.Linear(hidden_dim, hidden_dim * 4), nn.GELU(), nn.Dropout(dropout),
            nn.Linear(hidden_dim * 4, hidden_dim), nn.Dropout(dropout))
        self.ln1 = nn.LayerNorm(hidden_dim, elementwise_affine=False, eps=1e-6)
        self.ln2 = nn.LayerNorm(hidden_dim, elementwise_affine=False, eps=1e-6)


class PILOTMultiObjectiveGuidance(nn.Module):
    """Weights of model1.py:306-333."""

    def __init__(self, property_dim, hidden_dim, num_objectives=8):
        super().__init__()
        self.property_dim, self.hidden_dim, self.num_objectives = property_dim, hidden_dim, num_objectives
        self.property_encoders = nn.ModuleList([
            nn.Sequential(nn.Linear(1, hidden_dim // 4), nn.GELU(), nn.Linear(hidden_dim // 4, hidden_dim))
            for _ in range(num_objectives)])
        self.pareto_predictor = nn.Sequential(
            nn.Linear(num_objectives * hidden_dim, hidden_dim), nn.GELU(),
            nn.Linear(hidden_dim, num_objectives), nn.Softplus())
        self.null_embedding = nn.Parameter(torch.randn(hidden_dim))


class MolDiffConsistencyModule(nn.Module):
    """Weights and buffers of model1.py:368-451.

    The reference fills ``max_valences`` / ``covalent_radii`` from RDKit's periodic
    table (model1.py:423-451).  RDKit stays a CPU-side dependency of the data pipeline
    and is not needed here: the two buffers are part of ``state_dict`` and arrive with
    any checkpoint; a freshly constructed module gets the defaults the reference uses
    when RDKit has no entry (valence 1 -> overwritten on load, radius 1.5).
    """

    def __init__(self, hidden_dim, max_atoms=100):
        super().__init__()
        self.hidden_dim, self.max_atoms = hidden_dim, max_atoms
        mv = torch.zeros(119)
        mv[1:] = 1.0
        self.register_buffer("max_valences", mv)
        self.register_buffer("covalent_radii", torch.ones(119) * 1.5)
        h = hidden_dim
        self.atom_predictor = nn.Sequential(
            nn.Linear(h, h), nn.GELU(), nn.LayerNorm(h), nn.Linear(h, h // 2), nn.GELU(),
            nn.Dropout(0.1), nn.Linear(h // 2, 119))
        self.bond_predictor = nn.Sequential(
            nn.Linear(h * 2 + 4, h), nn.GELU(), nn.LayerNorm(h), nn.Linear(h, h // 2), nn.GELU(),
            nn.Dropout(0.1), nn.Linear(h // 2, 5))
        self.valency_predictor = nn.Sequential(
            nn.Linear(h + 119, h), nn.GELU(), nn.LayerNorm(h), nn.Linear(h, h // 2), nn.GELU(),
            nn.Linear(h // 2, 1), nn.Softplus())
        self.adj_assembler = nn.Sequential(nn.Linear(h * 2, h), nn.GELU(), nn.Linear(h, 5))
