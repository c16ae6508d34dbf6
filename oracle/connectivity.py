"""Oracle (test infrastructure): deterministic template connectivity, integer-exact.

Follows generation.py:562-668.  Three regimes by atom count n:
  n <= 5        open chain 0-1-...-(n-1)
  6 <= n <= 15  six-ring 0..5 plus substituent k (atom 6+k) bonded to ring atom (2k) % 6
  n >= 16       six-ring 0..5, five-ring 6..10, fusion bonds 1-6 and 2-10, then
                substituent k (atom 11+k) bonded to scaffold atom k % 11
Every bond is emitted as the directed pair (a, b) immediately followed by (b, a).
Ring and fusion bonds carry feature [0,0,0,1,1]; chain and substituent bonds [1,0,0,0,0].
Returns None when more than 8n directed edges would be produced (generation.py:661).
"""

SINGLE = (1.0, 0.0, 0.0, 0.0, 0.0)
AROMATIC = (0.0, 0.0, 0.0, 1.0, 1.0)


def template_connectivity(n):
    bonds = []  # (a, b, feature)
    if n <= 5:
        bonds += [(i, i + 1, SINGLE) for i in range(n - 1)]
    elif n <= 15:
        bonds += [(i, (i + 1) % 6, AROMATIC) for i in range(6)]
        bonds += [((2 * k) % 6, 6 + k, SINGLE) for k in range(n - 6)]
    else:
        bonds += [(i, (i + 1) % 6, AROMATIC) for i in range(6)]
        bonds += [(6 + i, 6 + (i + 1) % 5, AROMATIC) for i in range(5)]
        bonds += [(1, 6, AROMATIC), (2, 10, AROMATIC)]
        bonds += [(k % 11, 11 + k, SINGLE) for k in range(n - 11)]
    edges, feats = [], []
    for a, b, f in bonds:
        edges += [[a, b], [b, a]]
        feats += [list(f), list(f)]
    if len(edges) > 8 * n:
        return None
    return edges, feats


def fully_connected_edges(n):
    """Every ordered pair (i, j != i) as the directed edge [i, j], row-major in (i, j): the
    large-ligand configuration.  The reference's model accepts any edge_index (model1.py:723 passes
    data.edge_index straight to the E(3) layers); its generator never builds this graph itself."""
    return [[i, j] for i in range(n) for j in range(n) if j != i]
