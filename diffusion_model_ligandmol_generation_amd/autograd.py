"""Glue between the reference's ``model(data, t, properties, cfg)`` call and the engine."""
import torch

from .engine import PackedGraph


def packed_graph_of(data):
    """Packs ``data`` once and caches the result on the object (a sampler calls the model
    a thousand times with the same connectivity, generation.py:424).  The cache key holds the
    tensors' addresses, shapes and version counters, so an edge_index or batch vector that was
    replaced OR modified in place is packed again."""
    g = getattr(data, "_ligdiff_graph", None)
    ei, batch = data.edge_index, data.batch
    key = (ei.data_ptr(), batch.data_ptr(), tuple(ei.shape), batch.numel(), ei._version, batch._version)
    if g is None or getattr(g, "key", None) != key:
        g = PackedGraph(ei, batch)
        g.key = key
        try:
            data._ligdiff_graph = g
        except AttributeError:
            pass
    return g


def score_network_apply(model, data, t, properties=None, classifier_free_guidance=False, keep_state=False,
                        dropout=None):
    if not hasattr(data, "batch") or data.batch is None:
        batch = torch.zeros(data.x.shape[0], dtype=torch.long, device=data.x.device)   # model1.py:702
        data.batch = batch
    graph = packed_graph_of(data)
    eng = model.engine()
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    if needs_grad:
        from .backward import ScoreNetworkFunction
        return ScoreNetworkFunction.run(model, eng, graph, data.x, data.pos, t, properties, classifier_free_guidance,
                                        keep_state, dropout)
    return eng.forward(graph, data.x, data.pos, t, properties, null_cond=classifier_free_guidance,
                       keep_state=keep_state, dropout=dropout, options=getattr(model, "library_options", None))
