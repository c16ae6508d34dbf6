"""Oracle (test infrastructure): ragged-batch collation and CSR edge packing, integer-exact."""
import numpy as np


def collate(mols):
    """PyG-style collation that main.py:205-212 relies on (DataLoader over Data objects
    built at data_utils.py:898-914): node arrays are concatenated along dim 0,
    edge_index columns are offset by the number of preceding atoms and concatenated
    along dim 1, and batch[i] is the molecule index of atom i.

    mols: list of dicts with 'x' [n,A] f32, 'pos' [n,3] f32, 'edge_index' [2,e] int64.
    """
    xs, ps, es, bs = [], [], [], []
    off = 0
    for m, mol in enumerate(mols):
        n = mol["x"].shape[0]
        xs.append(np.asarray(mol["x"], dtype=np.float32))
        ps.append(np.asarray(mol["pos"], dtype=np.float32))
        ei = np.asarray(mol["edge_index"], dtype=np.int64).reshape(2, -1)
        es.append(ei + off)
        bs.append(np.full(n, m, dtype=np.int64))
        off += n
    return {
        "x": np.concatenate(xs, 0),
        "pos": np.concatenate(ps, 0),
        "edge_index": np.concatenate(es, 1),
        "batch": np.concatenate(bs, 0),
    }


def pack_csr(edge_index, batch, num_nodes=None):
    """Packed form consumed by the kernels.

    The E(3) layer aggregates messages at edge_index[1] (model1.py:100, :120-121 with
    PyG's default source_to_target flow) and index_add_s coordinates at both ends
    (model1.py:110-112).  Edges are therefore grouped by destination node, keeping the
    original edge order inside each group (a stable counting sort), which also fixes the
    summation order to that of a sequential index_add_.

    Returns dict of int32 arrays:
      mol_ptr [B+1]  first atom of each molecule (batch must be sorted ascending)
      row_ptr [N+1]  CSR offsets by destination node
      perm    [E]    original edge id of each CSR slot
      src     [E]    edge_index[0][perm]
      dst     [E]    edge_index[1][perm]
    """
    ei = np.asarray(edge_index, dtype=np.int64).reshape(2, -1)
    batch = np.asarray(batch, dtype=np.int64)
    N = int(batch.shape[0]) if num_nodes is None else int(num_nodes)
    B = int(batch.max()) + 1 if N > 0 else 0
    if N > 1 and np.any(np.diff(batch) < 0):
        raise ValueError("batch vector must be sorted ascending")
    counts = np.bincount(batch, minlength=B)
    mol_ptr = np.zeros(B + 1, dtype=np.int64)
    mol_ptr[1:] = np.cumsum(counts)
    E = ei.shape[1]
    if E and (ei.min() < 0 or ei.max() >= N):
        raise ValueError("edge_index out of range")
    perm = np.argsort(ei[1], kind="stable")
    deg = np.bincount(ei[1], minlength=N)
    row_ptr = np.zeros(N + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(deg)
    return {
        "mol_ptr": mol_ptr.astype(np.int32),
        "row_ptr": row_ptr.astype(np.int32),
        "perm": perm.astype(np.int32),
        "src": ei[0][perm].astype(np.int32),
        "dst": ei[1][perm].astype(np.int32),
    }
