"""one-off: strong cycle with k-paired sweep lists against the plain tiled lists, same aliasing: max |difference|"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bricklib_amd.strong import StrongRank
dev = torch.device("cuda:0")
for name, dom, sdom in (("mpi13pt", 128, 32), ("mpi25pt", 64, 16), ("mpi13pt", 256, 128)):
    res = {}
    for kp in (True, False):
        sr = StrongRank(dom, sdom, 0, 1, dev, kpair=kp)
        gen = torch.Generator(device=dev).manual_seed(17)
        sr.storages[0].dat.copy_(torch.rand(sr.storages[0].dat.shape, dtype=torch.float64, device=dev, generator=gen))
        for _ in range(2):
            sr.exchange_begin(); sr.exchange_end(); sr.sweeps(name)
        torch.cuda.synchronize()
        res[kp] = (sr.storages[0].dat.view(sr.nsub, sr.nb, 512)[:, 1:sr.d.sep_pos[1]].clone(), sr.swept_kpaired)
    d = (res[True][0] - res[False][0]).abs()
    print(name, dom, sdom, "paired run used k-paired lists:", res[True][1], "max |diff|", float(d.max()), "nonzero", int((d > 0).sum()), "of", d.numel())
