"""Experiment: one 1024-molecule reverse loop vs two concurrent 512-molecule loops (two host
threads, two streams).  Usage: python tools/overlap_test.py [steps]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from diffusion_model_ligandmol_generation_amd import ResearchAccurateDiffusionModel
from diffusion_model_ligandmol_generation_amd.engine import template_graph, _ptr

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
model = ResearchAccurateDiffusionModel(max_atoms=100).cuda().eval()
eng = model.engine()
lib = eng.lib
dev = torch.device("cuda")


def make(B, offset):
    g, _, _ = template_graph([29] * B, dev)
    N = g.n_atoms
    x = eng.philox_normal(N * 119, 1, 0xFFFFFFFF, 0, offset * 119, 1.0).view(N, 119)
    pos = eng.philox_normal(N * 3, 1, 0xFFFFFFFF, 1, offset * 3, 1.0).view(N, 3)
    nb = lib.ligdiff_ddpm_sample_workspace_bytes(g.n_atoms, g.n_edges, g.n_mols, 0, steps)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    return g, x, pos, ws, offset, torch.cuda.Stream()


def run(job):
    g, x, pos, ws, offset, st = job
    with torch.cuda.stream(st):
        rc = lib.ligdiff_ddpm_sample(eng.table, C.byref(g.c), _ptr(x), _ptr(pos), None, 1.0, 1.0, _ptr(model.betas),
                                     _ptr(model.alphas), _ptr(model.alphas_cumprod), steps, 1, offset, 1, _ptr(ws),
                                     ws.numel(), C.c_void_p(st.cuda_stream))
    assert rc == 0, rc


def timed(jobs):
    torch.cuda.synchronize()
    t0 = time.time()
    th = [threading.Thread(target=run, args=(j,)) for j in jobs]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.time() - t0) / steps * 1e3


full = [make(1024, 0)]
halves = [make(512, 0), make(512, 512 * 29)]
quarters = [make(256, i * 256 * 29) for i in range(4)]
for name, jobs in (("1 x 1024", full), ("2 x 512", halves), ("4 x 256", quarters), ("1 x 1024", full), ("2 x 512", halves)):
    timed(jobs)
    print(f"{name}: {timed(jobs):.3f} ms/step", flush=True)
ref = torch.cat([halves[0][1], halves[1][1]])
print("halves == full:", torch.equal(ref, full[0][1]))
