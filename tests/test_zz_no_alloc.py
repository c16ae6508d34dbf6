"""GPU: ligdiff_pack_csr_static (counters in a device array of the library, no workspace, no allocation)
against ligdiff_pack_csr, bit for bit, on two calls in a row - the second with a smaller input, so that
what the first call left in the shared array must not leak into it."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _batch(sizes, seed, fully_connected=False):
    rng = np.random.default_rng(seed)
    mols = []
    for n in sizes:
        if fully_connected:
            i, j = np.nonzero(~np.eye(n, dtype=bool))
            ei = np.stack([i, j]).astype(np.int64)
        else:
            ei = np.asarray(oracle.template_connectivity(n)[0], dtype=np.int64).T
        ei = ei[:, rng.permutation(ei.shape[1])]            # arrival order must not matter to the grouping
        mols.append({"x": np.zeros((n, 119), np.float32), "pos": np.zeros((n, 3), np.float32), "edge_index": ei})
    c = oracle.collate(mols)
    ei = torch.from_numpy(np.ascontiguousarray(c["edge_index"])).cuda().contiguous()   # [2,E] row-major
    return ei, torch.from_numpy(c["batch"]).cuda(), len(sizes)


def _alloc(N, E, B):
    i32 = dict(dtype=torch.int32, device="cuda")
    shapes = dict(mol_ptr=B + 1, atom_mol=N, dst_ptr=N + 1, src=E, dst=E, perm=E, src_ptr=N + 1, src_slot=E)
    return {k: torch.full((v,), -7, **i32) for k, v in shapes.items()}, torch.full((2,), -7, **i32)


ORDER = ("mol_ptr", "atom_mol", "dst_ptr", "src", "dst", "perm", "src_ptr", "src_slot")


def _run(lib, static, ei, batch, B):
    N, E = batch.numel(), ei.shape[1]
    out, status = _alloc(N, E, B)
    ptrs = [C.c_void_p(out[k].data_ptr()) for k in ORDER]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        rc = lib.ligdiff_pack_csr_static(C.c_void_p(ei.data_ptr()), E, C.c_void_p(batch.data_ptr()), N, B, *ptrs,
                                         C.c_void_p(status.data_ptr()), stream)
    else:
        nbytes = lib.ligdiff_pack_workspace_bytes(N, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        rc = lib.ligdiff_pack_csr(C.c_void_p(ei.data_ptr()), E, C.c_void_p(batch.data_ptr()), N, B, *ptrs,
                                  C.c_void_p(status.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, stream)
    assert rc == 0
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, status.cpu().numpy()


def test_static_scratch_packing_matches_workspace_packing(lib):
    big = _batch([29] * 300 + list(range(2, 65)), seed=1)
    small = _batch([5, 17, 9, 128, 3], seed=2, fully_connected=True)
    for ei, batch, B in (big, small):            # two calls in a row on the same device array
        want, wst = _run(lib, False, ei, batch, B)
        got, gst = _run(lib, True, ei, batch, B)
        assert np.array_equal(gst, wst) and gst[0] == 0
        for k in ORDER:
            assert np.array_equal(got[k], want[k]), k
        # and against the CPU oracle's packing
        ref = oracle.pack_csr(ei.cpu().numpy(), batch.cpu().numpy())
        for k, rk in (("mol_ptr", "mol_ptr"), ("dst_ptr", "row_ptr"), ("src", "src"), ("dst", "dst"), ("perm", "perm")):
            assert np.array_equal(got[k], ref[rk]), k


def test_static_scratch_packing_rejects_oversize_and_bad_input(lib):
    ei, batch, B = _batch([6, 7], seed=3)
    N, E = batch.numel(), ei.shape[1]
    out, status = _alloc(N, E, B)
    ptrs = [C.c_void_p(out[k].data_ptr()) for k in ORDER]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    too_many = 262144 + 1
    rc = lib.ligdiff_pack_csr_static(C.c_void_p(ei.data_ptr()), E, C.c_void_p(batch.data_ptr()), too_many, B, *ptrs,
                                     C.c_void_p(status.data_ptr()), stream)
    assert rc == 2                                 # LIGDIFF_ERR_WORKSPACE: beyond the static array
    rc = lib.ligdiff_pack_csr_static(C.c_void_p(ei.data_ptr()), E, None, N, B, *ptrs,
                                     C.c_void_p(status.data_ptr()), stream)
    assert rc == 1
    # device-side validation still reports an unsorted batch vector
    bad = batch.flip(0).contiguous()
    rc = lib.ligdiff_pack_csr_static(C.c_void_p(ei.data_ptr()), E, C.c_void_p(bad.data_ptr()), N, B, *ptrs,
                                     C.c_void_p(status.data_ptr()), stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(status[0]) == 4
