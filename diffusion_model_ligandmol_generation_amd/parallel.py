"""Sharding helpers for one-process-per-GPU runs (no counterpart in the reference, which is
single-process).  Sampling shards molecules with no collective; training shards the global
batch and sums gradients with one all-reduce."""
import torch


def shard_ranges(sizes, world, cost="atoms", align_atoms=None, slack=0.04):
    """Contiguous molecule ranges [(lo, hi)] * world, balanced by a per-molecule cost:
    'atoms' (template graphs: work ~ n) or 'pairs' (fully connected graphs: work ~ n^2).
    Contiguity keeps each rank's atoms one slice of the packed arrays and makes the global
    molecule index (rank offset + local index) well defined.

    align_atoms (e.g. 32768 = 256 compute units x 128-row workgroups): the fused kernels run a launch in
    whole rounds of one workgroup per compute unit, so a shard a few molecules above a multiple of
    align_atoms pays a whole extra round (1130 ligands of 29 atoms cost 24 % more than 1024).  A shard
    that exceeds a multiple by at most ``slack`` of align_atoms hands its last molecules to the next rank
    (the last rank takes what remains)."""
    w = [float(n) if cost == "atoms" else float(n) * float(n) for n in sizes]
    total = sum(w)
    out, lo, acc = [], 0, 0.0
    for r in range(world):
        target = total * (r + 1) / world
        hi = lo
        while hi < len(sizes) and (acc + w[hi] <= target + 1e-9 or hi == lo) and (len(sizes) - hi) > (world - r - 1):
            acc += w[hi]
            hi += 1
        if r == world - 1:
            hi = len(sizes)
        out.append((lo, hi))
        lo = hi
    if align_atoms:
        pre = [0]
        for n in sizes:
            pre.append(pre[-1] + int(n))
        for r in range(world - 1):
            lo, hi = out[r]
            while hi - lo > 1:
                over = (pre[hi] - pre[lo]) % int(align_atoms)
                if pre[hi] - pre[lo] < align_atoms or over == 0 or over > slack * align_atoms:
                    break
                hi -= 1
            out[r] = (lo, hi)
            out[r + 1] = (hi, out[r + 1][1])
    return out


def shard_batch(x, pos, edge_index, batch, lo, hi, properties=None):
    """Slice molecules [lo, hi) out of a packed batch (batch vector sorted): returns
    (x, pos, edge_index, batch, properties) re-based to local atom / molecule indices."""
    keep = (batch >= lo) & (batch < hi)
    idx = torch.nonzero(keep).flatten()
    a0 = int(idx[0]) if idx.numel() else 0
    ek = keep[edge_index[0]] & keep[edge_index[1]]
    return (x[keep], pos[keep], edge_index[:, ek] - a0, batch[keep] - lo,
            None if properties is None else properties[lo:hi])


def allreduce_sum_(t, group=None):
    """In-place sum all-reduce of one tensor.  Backend "nccl" (RCCL over xGMI) reduces the device
    buffer directly; any other backend (gloo: CPU tests and one-GPU rehearsals) goes through a host
    copy, which does not depend on that backend's handling of device memory."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        dist.all_reduce(t, group=group)
    else:
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    return t


def allreduce_flat(tensors, group=None):
    """Sum-all-reduce a list of tensors as ONE flat buffer (one collective per step: the
    6.6 M fp32 gradients are 26 MB, far below what per-tensor launches would cost)."""
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    allreduce_sum_(flat, group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
