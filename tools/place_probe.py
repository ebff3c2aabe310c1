"""Development script: brk_placement_probe (today the 7-point sweep over a probe grid; when profiles/r04_g_placement_copy_matrix.txt was
made it was a linear streaming copy) from one storage into each of N freshly allocated ones, and between all pairs of the first few --
the picture behind brk_storage_alloc (bricklib_amd/csrc/brk_place.hip).
    python tools/place_probe.py [N] [GiB per storage]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bricklib_amd._lib import check, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0967
nbytes = int(gib * (1 << 30)) // 4096 * 4096
bufs = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(n)]
torch.cuda.synchronize()


def probe(a, b):
    ms = C.c_double()
    check(lib.brk_placement_probe(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), nbytes, None, C.byref(ms)))
    return ms.value * (1 << 30) / nbytes


print(f"# {n} storages of {nbytes / (1 << 30):.4f} GiB; probe time (ms, scaled to 1 GiB) row -> column")
m = min(n, 12)
print("      " + " ".join(f"{j:6d}" for j in range(m)))
for i in range(m):
    print(f"{i:4d}  " + " ".join("   -  " if i == j else f"{probe(bufs[i], bufs[j]):6.4f}" for j in range(m)), flush=True)
print("from storage 0 into every storage:")
print(" ".join(f"{probe(bufs[0], bufs[j]):6.4f}" for j in range(1, n)), flush=True)
