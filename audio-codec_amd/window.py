"""window.py mirror (coder/window.py): same names, window * data evaluated on
the GPU (pacx_window_batch).  On the encode hot path the window multiply is
fused into the MDCT kernel instead (csrc/k_mdct.hip)."""
import numpy as np

from . import _lib, context


def _apply(kind, data, expect):
    import torch
    data = np.ascontiguousarray(data, dtype=np.float64)
    if data.shape[-1] != expect:
        raise NotImplementedError(f"GPU window tables exist for {expect}-sample blocks here")
    enc = context.any_encoder()
    y = enc.window(kind, torch.as_tensor(data, device=enc.device).view(-1, expect))
    return y.cpu().numpy().reshape(data.shape)


def _apply_table(table, data):
    """a block length without a resident table: the window evaluated on the host with the reference's expression
    (tables.py), multiplied on the GPU (pacx_window_table_batch)"""
    import torch
    data = np.ascontiguousarray(data, dtype=np.float64)
    n = data.shape[-1]
    enc = context.any_encoder()
    y = enc.window_table(table, torch.as_tensor(data, device=enc.device).view(-1, n))
    return y.cpu().numpy().reshape(data.shape)


def SineWindow(dataSampleArray):
    """coder/window.py:14-25.  2048- and 256-sample blocks use the resident tables."""
    from . import tables
    n = np.shape(dataSampleArray)[-1]
    if n not in (2048, 256):
        return _apply_table(tables.sine(n), dataSampleArray)
    return _apply(_lib.WIN_SINE if n == 2048 else _lib.WIN_SINE_SHORT, dataSampleArray, n)


def HanningWindow(dataSampleArray):
    """coder/window.py:29-41."""
    from . import tables
    n = np.shape(dataSampleArray)[-1]
    if n not in (2048, 256):
        return _apply_table(tables.hann(n), dataSampleArray)
    return _apply(_lib.WIN_HANN if n == 2048 else _lib.WIN_HANN_SHORT, dataSampleArray, n)


def KBDWindow(dataSampleArray, alpha=4.):
    """coder/window.py:45-57.  The 2048- and 256-sample tables for alpha = 4 are resident
    (PACX_WIN_KBD*); any other length or alpha is evaluated on the host with the reference's
    expression and multiplied on the GPU (pacx_window_table_batch)."""
    import torch
    from . import tables
    n = np.shape(dataSampleArray)[-1]
    if alpha == 4. and n in (2048, 256):
        return _apply(_lib.WIN_KBD if n == 2048 else _lib.WIN_KBD_SHORT, dataSampleArray, n)
    data = np.ascontiguousarray(dataSampleArray, dtype=np.float64)
    enc = context.any_encoder()
    y = enc.window_table(tables.kbd(n, alpha), torch.as_tensor(data, device=enc.device).view(-1, n))
    return y.cpu().numpy().reshape(data.shape)


def _resident(N_long, N_short):
    return (N_long, N_short) == (2048, 256)


def StartWindow(dataSampleArray, N_long, N_short):
    """coder/window.py:61-71 (resident table for N_long=2048, N_short=256; any other pair through the table path)."""
    from . import tables
    if not _resident(N_long, N_short):
        return _apply_table(tables.start(N_long, N_short), dataSampleArray)
    return _apply(_lib.WIN_START, dataSampleArray, 2048)


def StopWindow(dataSampleArray, N_long, N_short):
    """coder/window.py:73-80."""
    from . import tables
    if not _resident(N_long, N_short):
        return _apply_table(tables.stop(N_long, N_short), dataSampleArray)
    return _apply(_lib.WIN_STOP, dataSampleArray, 2048)


def StartStopWindow(dataSampleArray, N_long, N_short):
    """coder/window.py:82-92."""
    from . import tables
    if not _resident(N_long, N_short):
        return _apply_table(tables.start_stop(N_long, N_short), dataSampleArray)
    return _apply(_lib.WIN_STARTSTOP, dataSampleArray, 2048)
