"""bitalloc.py mirror (coder/bitalloc.py:62-145): BitAlloc and BitAlloc_SBR on the GPU."""
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


def BitAlloc_SBR(bitBudget, maxMantBits, nBands, nLines, SMR, omittedBands):
    """coder/bitalloc.py:123-145: an SBR-omitted band is sent as ONE value, so it counts one line.  Like the
    reference this writes the 1s into the caller's nLines array (sfBands.nLines stays that way for the rest of
    the file there) before allocating."""
    for b in omittedBands:
        nLines[b] = 1
    return BitAlloc(bitBudget, maxMantBits, nBands, nLines, SMR)
