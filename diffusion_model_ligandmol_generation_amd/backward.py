"""torch.autograd bridges: forward / backward of the score network and of the EDM loss step
run in the HIP library; autograd only carries the gradient tensors between them."""
import ctypes as C

import torch

from . import _lib
from ._lib import check
from .engine import _ptr, _stream


def _has_grad_slot(name):
    """Table entries that take part in the score network's backward: not the frequency table and
    not the consistency module, whose outputs reach no loss (train.py:519, :543, :559, :583)."""
    return name != "__time_freqs__" and not name.startswith("moldiff_consistency.")


def _grad_names(eng):
    """Names of the table slots this model fills that receive a gradient, in table order."""
    return [n for n in eng.live_names if _has_grad_slot(n)]


class _Ctx:
    """Non-tensor state handed to the autograd Functions."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class ScoreNetworkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st, *params):
        eng, graph = st.eng, st.graph
        lib = eng.lib
        N, B = graph.n_atoms, graph.n_mols
        flags = st.flags | _lib.FWD_SAVE_FOR_BACKWARD
        nbytes = lib.ligdiff_forward_workspace_bytes_opt(N, graph.n_edges, B, flags, eng._o(st.options))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=eng.device)   # kept alive until backward
        atom = torch.empty(N, 119, dtype=torch.float32, device=eng.device)
        posp = torch.empty(N, 3, dtype=torch.float32, device=eng.device)
        prop = torch.empty(B, 15, dtype=torch.float32, device=eng.device)
        with torch.cuda.device(eng.device):
            rc = lib.ligdiff_forward_opt(eng.table, C.byref(graph.c), _ptr(st.x), _ptr(st.pos), _ptr(st.t), None,
                                         _ptr(st.props), flags, _ptr(atom), _ptr(posp), _ptr(prop), _ptr(ws),
                                         ws.numel(), _stream(), C.byref(st.dropout) if st.dropout is not None else None,
                                         eng._o(st.options))
        check(rc, "ligdiff_forward_train(save)")
        ctx.st, ctx.ws, ctx.flags = st, ws, flags
        st.eng.last_forward = (ws, flags) if st.keep_state else None   # for consistency() right after this call
        ctx.param_shapes = [p.shape for p in params]
        ctx.set_materialize_grads(False)
        return atom, posp, prop

    @staticmethod
    def backward(ctx, d_atom, d_pos, d_prop):
        st, eng, graph = ctx.st, ctx.st.eng, ctx.st.graph
        lib = eng.lib
        N, B = graph.n_atoms, graph.n_mols
        dev = eng.device
        d_atom = torch.zeros(N, 119, device=dev) if d_atom is None else d_atom.contiguous().float()
        d_pos = torch.zeros(N, 3, device=dev) if d_pos is None else d_pos.contiguous().float()
        d_prop = None if d_prop is None else d_prop.contiguous().float()
        # one zero-filled buffer, carved into per-parameter views (one fill instead of ~150)
        sizes = [int(torch.Size(s).numel()) for s in ctx.param_shapes]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) // 64 * 64        # 256-B aligned views
        flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        grads = [flat[o:o + n].view(s) for o, n, s in zip(offs, sizes, ctx.param_shapes)]
        table = (C.c_void_p * len(eng.names))()
        gi = 0
        for i, name in enumerate(eng.names):
            if eng.present[i] and _has_grad_slot(name):
                table[i] = grads[gi].data_ptr()
                gi += 1
            else:
                table[i] = None
        nbytes = lib.ligdiff_backward_workspace_bytes_opt(N, graph.n_edges, B, eng._o(st.options))
        bws = eng.workspace(nbytes)
        with torch.cuda.device(dev):
            rc = lib.ligdiff_backward_opt(eng.table, table, C.byref(graph.c), _ptr(st.x), _ptr(st.pos), _ptr(st.props),
                                          ctx.flags, _ptr(d_atom), _ptr(d_pos), _ptr(d_prop), _ptr(ctx.ws), _ptr(bws),
                                          bws.numel(), _stream(),
                                          C.byref(st.dropout) if st.dropout is not None else None, eng._o(st.options))
        check(rc, "ligdiff_backward_train")
        ctx.ws = None
        # data parallel: the parameter gradients of one backward pass are ONE zero-padded buffer, so the
        # sum over ranks is one collective on it, with no gather / scatter copies (the views handed to
        # autograd below then already hold the global gradient)
        if st.grad_group is not None:
            from .parallel import allreduce_sum_
            allreduce_sum_(flat, st.grad_group)
        # the flat buffer and where each parameter's gradient sits in it: the fused optimiser step works on it directly
        eng.last_flat = {"flat": flat, "offsets": {n: o for n, o in zip(_grad_names(eng), offs)}}
        # parameters that did not take part get no gradient (as autograd would report them):
        # the optimiser then leaves them alone instead of applying weight decay to them
        has_props = bool(ctx.flags & _lib.FWD_PROPERTIES)
        null_cond = bool(ctx.flags & _lib.FWD_NULL_COND)
        out = []
        pnames = _grad_names(eng)
        for name, gten in zip(pnames, grads):
            used = True
            if name == "property_guidance.null_embedding":
                used = has_props and null_cond
            elif name.startswith("property_guidance."):
                used = has_props and not null_cond
            elif name.startswith("property_output."):
                used = d_prop is not None
            out.append(gten if used else None)
        # (names whose slice holds a gradient: the fused optimiser step takes its norm over the whole buffer, so it
        # checks that exactly these reached a parameter's .grad)
        eng.last_flat["live"] = {n for n, gten in zip(pnames, out) if gten is not None}
        return (None, *out)

    @staticmethod
    def run(model, eng, graph, x, pos, t, properties, null_cond, keep_state=False, dropout=None):
        """``model.grad_allreduce_group`` (set by a data-parallel trainer, else absent): process group
        over which backward sums the parameter gradients.  ``model.library_options`` (an engine Options struct, else
        absent): per-call options of the forward / backward pair, e.g. the split-operand GEMM mode."""
        N, B = graph.n_atoms, graph.n_mols
        x = x.detach().contiguous().float()
        pos = pos.detach().contiguous().float()
        t = t.detach().reshape(-1).contiguous().float()
        if t.numel() == 1 and B > 1:
            t = t.repeat(B)
        if t.numel() != B:
            raise ValueError("t needs one value per molecule")
        flags = 0
        props = None
        if properties is not None:
            flags |= _lib.FWD_PROPERTIES
            if null_cond:
                flags |= _lib.FWD_NULL_COND
            else:
                props = properties.detach().contiguous().float()
                if props.shape != (B, 15):
                    raise ValueError("properties must be [B,15]")
        sd = dict(model.state_dict(keep_vars=True))
        params = [sd[n] for n in _grad_names(eng)]
        st = _Ctx(eng=eng, graph=graph, x=x, pos=pos, t=t, props=props, flags=flags, keep_state=keep_state, dropout=dropout,
                  grad_group=getattr(model, "grad_allreduce_group", None), options=getattr(model, "library_options", None))
        return ScoreNetworkFunction.apply(st, *params)


class EDMLossFunction(torch.autograd.Function):
    """losses32 = f(atom_pred, pos_pred, prop_pred); only element 8 (total_loss) carries
    gradient.  The gradients w.r.t. the predictions are produced together with the loss
    values (one fused pass) and scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, st, atom_pred, pos_pred, prop_pred):
        eng, dev = st.eng, st.eng.device
        lib = eng.lib
        B = st.graph.n_mols
        a = _lib.LossArgs()
        ws = torch.empty(lib.ligdiff_loss_workspace_bytes(B), dtype=torch.uint8, device=dev)
        atom_pred, pos_pred = atom_pred.contiguous(), pos_pred.contiguous()
        prop_pred = prop_pred.contiguous() if prop_pred is not None else None
        a.pos, a.noise, a.noisy, a.coef = st.pos.data_ptr(), st.noise.data_ptr(), st.noisy.data_ptr(), st.coef.data_ptr()
        a.atom_pred, a.pos_pred = atom_pred.data_ptr(), pos_pred.data_ptr()
        a.prop_pred = prop_pred.data_ptr() if prop_pred is not None else None
        a.properties = st.props.data_ptr() if st.props is not None else None
        a.mol_ptr, a.n_mols, a.rot_mols = st.graph.mol_ptr.data_ptr(), B, st.rot_mols
        a.n_atoms_global, a.n_mols_global, a.workspace = st.n_atoms_global, st.n_mols_global, ws.data_ptr()
        sums = torch.empty(8, dtype=torch.float32, device=dev)
        losses = torch.zeros(32, dtype=torch.float32, device=dev)
        d_atom = torch.empty_like(atom_pred)
        d_pos = torch.empty_like(pos_pred)
        d_prop = torch.empty(B, 15, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(lib.ligdiff_loss_partial(C.byref(a), _ptr(sums), _stream()), "ligdiff_loss_partial")
            if st.group is not None:                      # data parallel: means over the global batch
                from .parallel import allreduce_sum_
                allreduce_sum_(sums, st.group)
            check(lib.ligdiff_loss_finish(C.byref(a), _ptr(sums), 1.0, _ptr(losses), _ptr(d_atom), _ptr(d_pos),
                                          _ptr(d_prop), _stream()), "ligdiff_loss_finish")
        ctx.save_for_backward(d_atom, d_pos, d_prop)
        ctx.has_prop = st.props is not None
        return losses

    @staticmethod
    def backward(ctx, g):
        d_atom, d_pos, d_prop = ctx.saved_tensors
        up = g[8]
        return None, d_atom * up, d_pos * up, (d_prop * up if ctx.has_prop else None)
