"""Traversal orders for brick lists.

The stencil kernels accept any list of brick ids; the order only affects speed.  On MI355X the launch gives each
of the 8 XCDs (separate L2s) one contiguous eighth of the list, and a workgroup takes 8 consecutive entries and
shares i-faces between them through LDS.  A good order therefore keeps i fastest (runs of i-neighbours), gives
each XCD its own slab of j-rows (j-faces stay inside one L2) and walks k inside the slab with a short reuse
distance (k-faces come back from L2 / Infinity Cache instead of HBM).  Measured effect on the 512^3 7-point sweep:
0.45 ms (plain lexicographic) -> 0.40 ms (profiles/)."""
import numpy as np

NXCD = 8


def tiled(ids3, rows_per_block=8, nxcd=NXCD, kstack=1):
    """ids3: array [nk][nj][ni] of brick ids (any subset of a grid in logical order).  Returns a flat list:
    for each XCD slab of j-rows: for each block of <= rows_per_block rows: for k: for j in block: for i.
    kstack > 1 (needs ni % 8 == 0 and nk % kstack == 0, else falls back to 1): inside a row, groups of 8 i-bricks are
    followed by the same 8 bricks of the next kstack-1 k-levels -- the layout the k-stacked kernels keep in registers."""
    ids3 = np.asarray(ids3)
    nk, nj, ni = ids3.shape
    if kstack > 1 and (ni % 8 or nk % kstack):
        kstack = 1
    bounds = [(nj * x) // nxcd for x in range(nxcd + 1)]
    parts = []
    for x in range(nxcd):
        for j0 in range(bounds[x], bounds[x + 1], rows_per_block):
            j1 = min(j0 + rows_per_block, bounds[x + 1])
            blk = np.ascontiguousarray(ids3[:, j0:j1, :])
            if kstack > 1:   # [kb][s][j][ib][w] -> [kb][j][ib][s][w]
                blk = blk.reshape(nk // kstack, kstack, j1 - j0, ni // 8, 8).transpose(0, 2, 3, 1, 4)
            parts.append(np.ascontiguousarray(blk).reshape(-1))
    return np.concatenate(parts).astype(np.uint32) if parts else np.zeros(0, dtype=np.uint32)


def decomp_lists(d, rows_per_block=8):
    """Brick lists for the weak/strong sweep schedule of a BrickDecomp `d` (see drivers.weak_cycle):
    inner   -- [1,sep0), tiled                       (never reads a ghost brick)
    shell   -- [sep0,sep2) surface + ghost, id order (thin regions)
    surface -- [sep0,sep1) id order
    all / domain -- inner + shell / inner + surface."""
    nb = [d.dims[a] // 8 - 2 * (d.depth // 8) for a in range(3)]   # inner extent in bricks (i,j,k)
    inner = 1 + np.arange(nb[0] * nb[1] * nb[2], dtype=np.int64).reshape(nb[2], nb[1], nb[0])
    assert inner.size == d.sep_pos[0] - 1
    inner_t = tiled(inner, rows_per_block)
    shell = np.arange(d.sep_pos[0], d.sep_pos[2], dtype=np.uint32)
    surface = np.arange(d.sep_pos[0], d.sep_pos[1], dtype=np.uint32)
    return {"inner": inner_t, "shell": shell, "surface": surface,
            "all": np.concatenate([inner_t, shell]),        # [1,sep2): subdomain + ghost ring
            "domain": np.concatenate([inner_t, surface])}   # [1,sep1): the subdomain itself
