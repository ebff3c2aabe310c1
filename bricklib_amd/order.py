"""Traversal orders for brick lists.

The stencil kernels accept any list of brick ids; the order only affects speed.  On MI355X the launch gives each
of the 8 XCDs (separate L2s) one contiguous eighth of the list, and a workgroup takes 8 consecutive entries and
shares i-faces between them through LDS.  A good order therefore keeps i fastest (runs of i-neighbours), gives
each XCD its own slab of j-rows (j-faces stay inside one L2) and walks k inside the slab with a short reuse
distance (k-faces come back from L2 / Infinity Cache instead of HBM).  Measured effect on the 512^3 7-point sweep:
0.45 ms (plain lexicographic) -> 0.40 ms (profiles/)."""
import numpy as np

NXCD = 8


LIST_SKIP = 0xFFFFFFFF   # BRK_LIST_SKIP of include/brk.h: list entries the kernels ignore


def _pad(a, n):
    return a if a.size == n else np.concatenate([a, np.full(n - a.size, LIST_SKIP, dtype=np.uint32)])


def slabs(ids3, rows_per_block=8, nxcd=NXCD, pad_rows=True):
    """Per-XCD pieces of the tiled order of `ids3` ([nk][nj][ni] brick ids): XCD x gets its own slab of j-rows, swept
    in blocks of <= rows_per_block rows over all k, i fastest.  pad_rows: rows are padded with LIST_SKIP to a multiple
    of 8 entries so that a workgroup (8 consecutive entries) never straddles two rows."""
    ids3 = np.asarray(ids3).astype(np.uint32)
    nk, nj, ni = ids3.shape
    if pad_rows and ni % 8:
        ids3 = np.concatenate([ids3, np.full((nk, nj, 8 - ni % 8), LIST_SKIP, dtype=np.uint32)], axis=2)
    bounds = [(nj * x) // nxcd for x in range(nxcd + 1)]
    out = []
    for x in range(nxcd):
        parts = [np.ascontiguousarray(ids3[:, j0:min(j0 + rows_per_block, bounds[x + 1]), :]).reshape(-1)
                 for j0 in range(bounds[x], bounds[x + 1], rows_per_block)]
        out.append(np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint32))
    return out


def assemble(*piece_lists, nxcd=NXCD):
    """Concatenate per-XCD pieces (each argument: a list of nxcd arrays) into ONE list whose x-th eighth is exactly XCD
    x's work: the kernels give XCD x the x-th eighth of the list, so the eighths are padded to equal length (a
    multiple of 8 entries) with LIST_SKIP."""
    per = [np.concatenate([pl[x] for pl in piece_lists]) for x in range(nxcd)]
    n = max(p.size for p in per)
    n = (n + 7) // 8 * 8
    return np.concatenate([_pad(p, n) for p in per]).astype(np.uint32)


def chunks(ids, nxcd=NXCD):
    """split a flat id list into nxcd contiguous pieces, each a multiple of 8 entries where possible"""
    ids = np.asarray(ids, dtype=np.uint32)
    per = ((ids.size + nxcd - 1) // nxcd + 7) // 8 * 8
    return [ids[x * per:(x + 1) * per] for x in range(nxcd)]


def tiled(ids3, rows_per_block=8, nxcd=NXCD, pad_rows=True):
    """Flat tiled list of `ids3` ([nk][nj][ni] brick ids), see slabs()/assemble()."""
    return assemble(slabs(ids3, rows_per_block, nxcd, pad_rows), nxcd=nxcd)


def decomp_lists(d, rows_per_block=8):
    """Brick lists for the weak/strong sweep schedule of a BrickDecomp `d` (see drivers.weak_cycle):
    inner   -- [1,sep0), tiled                       (never reads a ghost brick)
    shell   -- [sep0,sep2) surface + ghost, id order (thin regions)
    surface -- [sep0,sep1) id order
    all / domain -- inner + shell / inner + surface, interleaved per XCD."""
    nb = [d.dims[a] // 8 - 2 * (d.depth // 8) for a in range(3)]   # inner extent in bricks (i,j,k)
    inner = 1 + np.arange(nb[0] * nb[1] * nb[2], dtype=np.int64).reshape(nb[2], nb[1], nb[0])
    assert inner.size == d.sep_pos[0] - 1
    inner_s = slabs(inner, rows_per_block)
    shell = np.arange(d.sep_pos[0], d.sep_pos[2], dtype=np.uint32)
    surface = np.arange(d.sep_pos[0], d.sep_pos[1], dtype=np.uint32)
    return {"inner": assemble(inner_s), "shell": shell, "surface": surface,
            "all": assemble(inner_s, chunks(shell)),          # [1,sep2): subdomain + ghost ring
            "domain": assemble(inner_s, chunks(surface))}     # [1,sep1): the subdomain itself


def _box(grid, lo, hi):
    return grid[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]


def trapezoid_lists(d, sweeps, rows_per_block=8):
    """Brick lists for the deep-overlap schedule of drivers.weak_cycle_deep.  With L(b) = distance of brick b from the
    outside of the ghosted grid in bricks (ghost ring: 0, surface: 1, ...), sweep t = 1..sweeps is split into
        inner[t-1] -- bricks with L >= t+1: they depend on nothing the exchange delivers (the dependence cone of the
                      ghost ring widens by at most one brick per sweep, radius*sweeps <= 8 cells), and
        shell[t-1] -- bricks with L <= t (the last sweep leaves out the ghost ring, `skip = 1` of weak/main.cpp:198-207).
    inner + shell is exactly the brick set the reference computes in sweep t.  Returns (inner, shell): two lists of
    uint32 arrays (possibly empty)."""
    g = np.asarray(d.grid).astype(np.uint32)
    T = [g.shape[2], g.shape[1], g.shape[0]]          # bricks per axis (i,j,k) incl. ghost
    inner, shell = [], []
    for t in range(1, sweeps + 1):
        lo = [1 if t == sweeps else 0] * 3            # outer box [lo, T-lo)
        hi = [T[a] - lo[a] for a in range(3)]
        ilo = [t + 1] * 3
        ihi = [T[a] - t - 1 for a in range(3)]
        if any(ihi[a] <= ilo[a] for a in range(3)):   # the cone has closed: everything waits for the exchange
            inner.append(np.zeros(0, dtype=np.uint32))
            shell.append(tiled(_box(g, lo, hi), rows_per_block))
            continue
        inner.append(tiled(_box(g, ilo, ihi), rows_per_block))
        klo = slabs(_box(g, lo, [hi[0], hi[1], ilo[2]]), rows_per_block)
        khi = slabs(_box(g, [lo[0], lo[1], ihi[2]], hi), rows_per_block)
        rest = [_box(g, [lo[0], lo[1], ilo[2]], [hi[0], ilo[1], ihi[2]]),        # j-low / j-high
                _box(g, [lo[0], ihi[1], ilo[2]], [hi[0], hi[1], ihi[2]]),
                _box(g, [lo[0], ilo[1], ilo[2]], [ilo[0], ihi[1], ihi[2]]),      # i-low / i-high
                _box(g, [ihi[0], ilo[1], ilo[2]], [hi[0], ihi[1], ihi[2]])]
        rest = np.concatenate([np.ascontiguousarray(r).reshape(-1) for r in rest])
        shell.append(assemble(klo, khi, chunks(rest)))
    return inner, shell
