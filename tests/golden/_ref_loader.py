"""Loads the reference's model1.py / train.py / generation.py in THIS container only.

The reference imports torch_geometric and rdkit, neither of which is installed and
neither of which can be fetched.  The symbols it uses from them are few and simple, so
this file seeds ``sys.modules`` with minimal stand-ins for exactly those symbols before
importing the reference files from /root/reference, unmodified.  The stand-ins are this
project's reading of the PyG semantics (they are not PyG):

* ``MessagePassing(aggr='add', node_dim=0).propagate(edge_index, h=..., edge_msg=...)``
  calls ``self.message(h_j=h[edge_index[0]], edge_msg=edge_msg)`` and scatter-adds the
  result at ``edge_index[1]`` into ``h.size(0)`` rows (source_to_target flow).
* ``to_dense_batch(x, batch, max_num_nodes=M)`` -> ``[B, M, F]`` zero padded + bool mask.
* ``global_mean_pool(x, batch)`` -> per-graph mean over ``batch.max()+1`` graphs.
* ``Data`` -> attribute bag; an attribute passed as ``None`` is still an attribute.
* ``rdkit.Chem.GetPeriodicTable`` -> constant table (placeholder values; the buffers it
  fills are only read by the consistency module, whose outputs reach no loss).

Nothing here ships to the GPU box and nothing under the package imports it; it exists so
``make_golden.py`` can write small ``.npz`` fixtures (inputs + expected outputs).
"""
import importlib
import importlib.util
import sys
import types

import torch

REF = "/root/reference"


class _Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class _MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", node_dim=0, **kw):
        super().__init__()
        assert aggr == "add" and node_dim == 0

    def propagate(self, edge_index, h=None, edge_msg=None, **kw):
        src, dst = edge_index[0], edge_index[1]
        msg = self.message(h_j=h[src], edge_msg=edge_msg)
        out = torch.zeros(h.size(0), msg.size(-1), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        return out


def _to_dense_batch(x, batch, max_num_nodes=None, fill_value=0.0):
    B = int(batch.max().item()) + 1
    counts = torch.bincount(batch, minlength=B)
    M = int(counts.max().item()) if max_num_nodes is None else int(max_num_nodes)
    starts = torch.cumsum(counts, 0) - counts
    idx_in_mol = torch.arange(batch.numel(), device=batch.device) - starts[batch]
    keep = idx_in_mol < M
    out = x.new_full((B, M) + tuple(x.shape[1:]), fill_value)
    mask = torch.zeros(B, M, dtype=torch.bool, device=x.device)
    out[batch[keep], idx_in_mol[keep]] = x[keep]
    mask[batch[keep], idx_in_mol[keep]] = True
    return out, mask


def _global_add_pool(x, batch, size=None):
    B = int(batch.max().item()) + 1 if size is None else size
    out = torch.zeros((B,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    out.index_add_(0, batch, x)
    return out


def _global_mean_pool(x, batch, size=None):
    B = int(batch.max().item()) + 1 if size is None else size
    s = _global_add_pool(x, batch, B)
    c = torch.bincount(batch, minlength=B).clamp(min=1).to(x.dtype)
    return s / c.view(-1, *([1] * (x.dim() - 1)))


class _PeriodicTable:
    _VAL = {1: [1], 6: [4], 7: [3], 8: [2], 9: [1], 15: [3, 5, 7], 16: [2, 4, 6], 17: [1], 35: [1], 53: [1, 3, 5]}
    _RCOV = {1: 0.31, 6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 15: 1.07, 16: 1.05, 17: 1.02, 35: 1.20}

    def GetValenceList(self, z):
        return self._VAL.get(z, [1])

    def GetRcovalent(self, z):
        return self._RCOV.get(z, 1.5)


def _seed_modules():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    tg = mod("torch_geometric")
    tg_nn = mod("torch_geometric.nn")
    tg_utils = mod("torch_geometric.utils")
    tg_data = mod("torch_geometric.data")
    tg.nn, tg.utils, tg.data = tg_nn, tg_utils, tg_data
    tg_nn.MessagePassing = _MessagePassing
    tg_nn.global_add_pool = _global_add_pool
    tg_nn.global_mean_pool = _global_mean_pool
    tg_utils.to_dense_batch = _to_dense_batch
    tg_utils.dense_to_sparse = lambda adj: (_ for _ in ()).throw(NotImplementedError())
    tg_data.Data = _Data
    tg_data.Batch = _Data

    rd = mod("rdkit")
    chem = mod("rdkit.Chem")
    rd.Chem = chem
    chem.rdMolDescriptors = mod("rdkit.Chem.rdMolDescriptors")
    chem.Descriptors = mod("rdkit.Chem.Descriptors")
    chem.GetPeriodicTable = lambda: _PeriodicTable()


def _load(name, filename):
    spec = importlib.util.spec_from_file_location(name, f"{REF}/{filename}")
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def load_reference():
    """Returns (model1, train, generation) modules of the reference."""
    _seed_modules()
    model1 = _load("model1", "model1.py")
    # generation.py imports these two names from a module called `model` that the
    # reference does not contain; model1.py is the file that defines the model.
    model1.ensure_model_has_all_buffers = lambda m: None
    model1.load_model_with_missing_buffers = lambda *a, **k: None
    sys.modules["model"] = model1
    ev = types.ModuleType("evaluation")

    class MolecularGenerationBenchmark:  # evaluation.py needs seaborn/rdkit/chembl client
        pass

    ev.MolecularGenerationBenchmark = MolecularGenerationBenchmark
    sys.modules["evaluation"] = ev
    train = _load("ref_train", "train.py")
    # generation.py does `from model import _add_missing_attributes_to_model`, which is
    # defined in generation.py itself (generation.py:671); bind lazily.
    model1._add_missing_attributes_to_model = lambda m: None
    generation = _load("ref_generation", "generation.py")
    model1._add_missing_attributes_to_model = generation._add_missing_attributes_to_model
    return model1, train, generation
