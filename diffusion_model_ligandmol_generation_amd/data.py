"""Host-side batch format of the hot path: the packed ragged batch the reference's PyG DataLoader
hands to the trainer (data_utils.py:898-914 per molecule; main.py:205-212 collate).  Pure host
plumbing - no arithmetic of the model happens here."""
import torch


class MoleculeBatch:
    """Attribute bag with the fields model.forward / the trainer read: x [N,119], pos [N,3],
    edge_index [2,E] int64 (atom indices offset by the molecule's first atom), batch [N] int64,
    properties [B,15] or None."""

    def __init__(self, x, pos, edge_index, batch, properties=None):
        self.x, self.pos, self.edge_index, self.batch = x, pos, edge_index, batch
        if properties is not None:
            self.properties = properties

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


def collate_molecules(mols):
    """PyG-style collation (main.py:205-212): node arrays concatenated along dim 0, edge_index columns
    offset by the number of preceding atoms, batch[i] = molecule of atom i.  ``mols``: dicts (or
    objects) with x [n,119], pos [n,3], edge_index [2,e] and optionally properties [15]."""
    get = (lambda m, k: m[k]) if isinstance(mols[0], dict) else getattr
    has = (lambda m, k: k in m) if isinstance(mols[0], dict) else hasattr
    xs, ps, es, bs, prs = [], [], [], [], []
    off = 0
    for i, m in enumerate(mols):
        x = torch.as_tensor(get(m, "x"), dtype=torch.float32)
        n = x.shape[0]
        xs.append(x)
        ps.append(torch.as_tensor(get(m, "pos"), dtype=torch.float32))
        es.append(torch.as_tensor(get(m, "edge_index"), dtype=torch.int64).reshape(2, -1) + off)
        bs.append(torch.full((n,), i, dtype=torch.int64))
        if has(m, "properties") and get(m, "properties") is not None:
            prs.append(torch.as_tensor(get(m, "properties"), dtype=torch.float32).reshape(-1)[:15])
        off += n
    props = torch.stack(prs) if len(prs) == len(mols) else None
    return MoleculeBatch(torch.cat(xs), torch.cat(ps), torch.cat(es, 1).contiguous(), torch.cat(bs), props)


class MoleculeLoader:
    """Minimal stand-in for torch_geometric's DataLoader over a list of molecules: batches of
    ``batch_size`` collated by collate_molecules, optional shuffling (torch's global generator, as
    torch.utils.data does) and drop_last."""

    def __init__(self, molecules, batch_size=16, shuffle=False, drop_last=False):
        self.molecules, self.batch_size, self.shuffle, self.drop_last = list(molecules), batch_size, shuffle, drop_last

    def __len__(self):
        n = len(self.molecules)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = torch.randperm(len(self.molecules)).tolist() if self.shuffle else list(range(len(self.molecules)))
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            yield collate_molecules([self.molecules[j] for j in idx])
