"""bitalloc.py mirror (coder/bitalloc.py:62-121): BitAlloc on the GPU."""
import numpy as np

from . import context

DBTOBITS = 6.2


def BitAlloc(bitBudget, maxMantBits, nBands, nLines, SMR):
    import torch
    enc = context.any_encoder()
    smr = torch.as_tensor(np.ascontiguousarray(SMR, dtype=np.float64)[:nBands], device=enc.device).view(1, nBands)
    budget = torch.tensor([float(bitBudget)], dtype=torch.float64, device=enc.device)
    bits = enc.bit_alloc_generic(budget, maxMantBits, np.asarray(nLines)[:nBands], smr)
    return bits[0].cpu().numpy().astype(int)
