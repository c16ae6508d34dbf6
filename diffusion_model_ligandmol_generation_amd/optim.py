"""Optimiser of the drop-in trainer: torch.optim.AdamW (train.py:47-53) whose step, together with the global-norm
clipping in front of it (train.py:196-197), runs as two launches of the HIP library when the gradients are the
slices of the flat buffer that the score network's backward pass produces.  State (``exp_avg``, ``exp_avg_sq``,
``step``), ``param_groups`` and ``state_dict`` are torch's own, so schedulers and checkpoints see a plain AdamW;
whenever the preconditions do not hold the stock ``clip_grad_norm_`` + ``step`` run instead."""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import check


class FusedAdamW(torch.optim.AdamW):
    CHUNK = 2048   # elements per workgroup of the update kernel (csrc/optim.hip: OPT_CHUNK)

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        self._tables = {}
        self.last_path = None      # "fused" or "torch": which path the last clip_and_step took
        self.last_norm = None      # device tensor [2]: total gradient norm, clip coefficient (fused path)

    def _table(self, key, entries, device):
        tab = self._tables.get(key)
        if tab is None:
            rows, chunk_t, chunk_s = [], [], []
            for i, (p, m, v, off) in enumerate(entries):
                rows.append([p.data_ptr(), m.data_ptr(), v.data_ptr(), off, p.numel()])
                for c in range((p.numel() + self.CHUNK - 1) // self.CHUNK):
                    chunk_t.append(i)
                    chunk_s.append(c)
            tab = (torch.tensor(rows, dtype=torch.int64).to(device), torch.tensor(chunk_t, dtype=torch.int32).to(device),
                   torch.tensor(chunk_s, dtype=torch.int32).to(device))
            if len(self._tables) > 8:
                self._tables.clear()
            self._tables[key] = tab
        return tab

    def clip_and_step(self, model, max_norm=1.0):
        """clip_grad_norm_(model.parameters(), max_norm) followed by step().  Fused when every gradient is a slice of
        the engine's flat gradient buffer of the last backward pass (and the group uses plain AdamW options)."""
        eng = model.engine()
        info = getattr(eng, "last_flat", None)
        group = self.param_groups[0] if len(self.param_groups) == 1 else None
        ok = (info is not None and group is not None and not group.get("amsgrad", False) and not group.get("maximize", False)
              and not group.get("capturable", False))
        entries_by_step = {}
        if ok:
            flat, offsets = info["flat"], info["offsets"]
            names = getattr(self, "_names", None)
            if names is None:
                names = self._names = {id(p): n for n, p in model.named_parameters()}
            base = flat.data_ptr()
            seen = set()
            for p in group["params"]:
                if p.grad is None:
                    continue
                seen.add(names.get(id(p)))
                off = offsets.get(names.get(id(p)))
                if (off is None or p.grad.data_ptr() != base + 4 * off or p.dtype != torch.float32 or not p.is_contiguous()
                        or p.grad.dtype != torch.float32):
                    ok = False
                    break
                st = self.state[p]
                if len(st) == 0:      # as torch.optim.AdamW initialises it
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                entries_by_step.setdefault(float(st["step"]), []).append((p, st["exp_avg"], st["exp_avg_sq"], off))
            # the norm is taken over the whole buffer: every slice that holds a gradient must belong to a parameter of
            # this optimiser (a frozen parameter's slice would enter the norm but not torch's clip_grad_norm_)
            if ok and seen != info.get("live", seen):
                ok = False
        if not ok or not entries_by_step:
            self.last_path = "torch"
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
            self.step()
            return
        lib, dev = eng.lib, eng.device
        beta1, beta2 = group["betas"]
        if self.last_norm is None or self.last_norm.device != dev:
            self.last_norm = torch.zeros(2, dtype=torch.float32, device=dev)
            self._ws = torch.empty(lib.ligdiff_adamw_workspace_bytes(), dtype=torch.uint8, device=dev)
        first = True
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for step0, entries in sorted(entries_by_step.items()):
                step = step0 + 1.0
                # (the rows hold parameter / moment addresses and offsets INTO the flat buffer, not its address)
                key = tuple((p.data_ptr(), m.data_ptr(), off) for p, m, _, off in entries)
                tab, chunk_t, chunk_s = self._table(key, entries, dev)
                check(lib.ligdiff_adamw_step(tab.data_ptr(), chunk_t.data_ptr(), chunk_s.data_ptr(), chunk_t.numel(),
                                             flat.data_ptr(), flat.numel(), float(group["lr"]), float(beta1), float(beta2),
                                             float(group["eps"]), float(group["weight_decay"]), 1.0 - beta1 ** step,
                                             math.sqrt(1.0 - beta2 ** step), float(max_norm), 1 if first else 0,
                                             self.last_norm.data_ptr(), self._ws.data_ptr(), self._ws.numel(), stream),
                      "ligdiff_adamw_step")
                first = False
                for p, _, _, _ in entries:
                    self.state[p]["step"] += 1
        self.last_path = "fused"
