#!/usr/bin/env python3
"""Debug build only (scripts/build_variant.sh bfdbg ba_kernels.hip -DORBX_BF_DEBUG): ba_big_factor_kernel alone against numpy."""
import ctypes, os, sys
import numpy as np
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_ab", os.environ.get("BFDBG", "bfdbg") + ".so"))
dp = ctypes.POINTER(ctypes.c_double)
for n in (16, 32, 48, 64, 180, 192, 294, 319, 320):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); S = A @ A.T + n * np.eye(n); b = rng.standard_normal(n)
    L = np.linalg.cholesky(S); y = np.linalg.solve(L, b)
    Sd = S.copy(); bd = b.copy(); gi = np.zeros(n)
    rc = lib.orbx_debug_big_factor(Sd.ctypes.data_as(dp), bd.ctypes.data_as(dp), gi.ctypes.data_as(dp), n)
    Ld = np.tril(Sd)
    err = np.abs(Ld - L)
    print("n=%d rc=%d max|L-Lref|=%.3e max|y-yref|=%.3e max|ginv-1/diag|=%.3e" % (n, rc, err.max(), np.abs(bd - y).max(), np.abs(gi - 1 / np.diag(L)).max()))
    if err.max() > 1e-9:
        bad = np.argwhere(err > 1e-9)
        print("  first bad entries (row, col):", bad[:8].tolist(), " bad rows by panel:", sorted(set((bad[:, 1] // 16).tolist()))[:10], "tile rows:", sorted(set((bad[:, 0] // 16).tolist()))[:12])
