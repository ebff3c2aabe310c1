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


def _pair_levels(ids3):
    """[nk][nj][ni] -> [ceil(nk/2)][nj][8 * ceil(ni/4)]: two k-levels of a row interleaved in runs of 4 i-neighbours,
    [i..i+3 at k][i..i+3 at k+1], LIST_SKIP where a run is shorter or the upper level does not exist: the k-paired order of
    k_star_kp (brk_starwg.hip), in which a wave computes entry w of the first run and entry w of the second"""
    nk, nj, ni = ids3.shape
    ni4 = (ni + 3) // 4 * 4
    nk2 = (nk + 1) // 2 * 2
    pad = np.full((nk2, nj, ni4), LIST_SKIP, dtype=np.uint32)
    pad[:nk, :, :ni] = ids3
    p = pad.reshape(nk2 // 2, 2, nj, ni4 // 4, 4)          # [pair][level][j][run][entry]
    return np.ascontiguousarray(p.transpose(0, 2, 3, 1, 4)).reshape(nk2 // 2, nj, 2 * ni4)


def slabs(ids3, rows_per_block=8, nxcd=NXCD, pad_rows=True, kpair=False, icols=0):
    """Per-XCD pieces of the tiled order of `ids3` ([nk][nj][ni] brick ids): XCD x gets its own slab of nj // nxcd
    j-rows, swept in blocks of <= rows_per_block rows over all k, i fastest; the nj % nxcd left-over rows are cut along k
    instead, one k-range per XCD, so that every XCD gets the same amount of work (66 rows: 8 each + 2 rows x 1/8 of the
    planes, instead of two XCDs with 9 rows setting the pace).  pad_rows: a row whose length is not a multiple of 8 gets ONE LIST_SKIP entry
    behind it, so that two rows never meet inside a workgroup without an idle wave between them (the kernels park
    fallback halos in an idle wave's LDS slot; two active non-neighbours side by side cost a second phase).
    icols (k-paired lists; 0 = whole rows): the rows of a block are cut along i into chunks of icols bricks (a multiple of 4), each swept
    over all k before the next -- the distance between the request of a brick's planes as the k-halo of the pair below and as the brick
    itself halves (4 MiB = the L2 of an XCD for 8 rows x 64 bricks x 2 levels) without fewer rows per block.  Pays for the radius-4 star
    (512^3: fabric reads 1.80 -> 1.32 GB, 0.4422 -> 0.4356 ms), not for the radius-2 star (brk_kernels.hip: box_list_locked)."""
    ids3 = np.asarray(ids3).astype(np.uint32)
    ni_bricks = ids3.shape[2]
    if kpair:   # every entry of the "k" axis now stands for a pair of levels; rows are multiples of 8 entries (no padding entry needed)
        ids3 = _pair_levels(ids3)
        pad_rows = False
    chunk = 2 * ((icols + 3) // 4 * 4) if (kpair and 0 < icols < ni_bricks) else 0      # entries of a paired row per chunk
    nk, nj, ni = ids3.shape
    if pad_rows and ni % 8:
        ids3 = np.concatenate([ids3, np.full((nk, nj, 1), LIST_SKIP, dtype=np.uint32)], axis=2)
    base, rem = divmod(nj, nxcd)
    out = []
    for x in range(nxcd):
        width = ids3.shape[2]                      # entries per row (with its padding entry, if any)
        parts = [np.ascontiguousarray(ids3[:, j0:min(j0 + rows_per_block, (x + 1) * base), c0:c0 + (chunk or width)]).reshape(-1)
                 for j0 in range(x * base, (x + 1) * base, rows_per_block) for c0 in range(0, width, chunk or width)]
        if rem:
            k0, k1 = (nk * x) // nxcd, (nk * (x + 1)) // nxcd
            parts.append(np.ascontiguousarray(ids3[k0:k1, nj - rem:, :]).reshape(-1))
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


def tiled(ids3, rows_per_block=8, nxcd=NXCD, pad_rows=True, kpair=False, icols=0):
    """Flat tiled list of `ids3` ([nk][nj][ni] brick ids), see slabs()/assemble().  kpair: the k-paired order for
    apply_stencil(..., kpaired=True) -- same bricks, runs of 4 i-neighbours at k followed by the same run at k+1; icols: see slabs()."""
    return assemble(slabs(ids3, rows_per_block, nxcd, pad_rows, kpair, icols), nxcd=nxcd)


def tiled_scattered(ids, gi, gj, gk, rows_per_block=8, nxcd=NXCD, kpair=False):
    """The tiled order for bricks given by coordinates instead of a box: `ids` with brick coordinates (gi, gj, gk) in any common
    frame (several subdomains of one rank in the global frame, holes and duplicates allowed).  Same traversal as slabs(): XCD x gets
    its slab of j-rows, swept in blocks of <= rows_per_block rows (None: chosen from the row length) over all k with i fastest,
    left-over rows are cut along k; runs of i-neighbours stay together and a LIST_SKIP entry follows every run whose length is not a multiple of 8.
    kpair: the k-paired order of tiled(..., kpair=True) over the bounding box of the coordinates, LIST_SKIP where the box has no
    brick (no duplicates then): groups of 8 entries = 4 i-neighbours + their +k neighbours; a group with a hole is computed one
    brick at a time by the march kernel, so this pays for brick sets that are mostly a box (a subdomain minus aliased ghost faces)."""
    ids = np.asarray(ids, dtype=np.uint32)
    if ids.size == 0:
        return np.zeros(0, dtype=np.uint32)
    gi, gj, gk = (np.asarray(c, dtype=np.int64) - int(np.min(c)) for c in (gi, gj, gk))
    J, K = int(gj.max()) + 1, int(gk.max()) + 1
    if kpair:
        box = np.full((K, J, int(gi.max()) + 1), LIST_SKIP, dtype=np.uint32)
        assert np.unique(np.stack([gk, gj, gi]), axis=1).shape[1] == ids.size, "k-paired order: one brick per coordinate"
        box[gk, gj, gi] = ids
        return tiled(box, rows_per_block or 8, nxcd, kpair=True)
    if rows_per_block is None:       # a block of rows of one k-plane stays within ~2 MiB (512 bricks), the measured optimum at 512^3
        rows_per_block = max(1, min(8, 512 // (int(gi.max()) + 1)))
    base = J // nxcd
    left = gj >= base * nxcd                                            # the J % nxcd left-over rows (all rows when J < nxcd)
    slab = np.where(left, (nxcd * (gk + 1) + K - 1) // K - 1, gj // max(base, 1))     # k in [K*x // nxcd, K*(x+1) // nxcd), as slabs()
    block = np.where(left, J, (gj - slab * base) // rows_per_block)    # left-over rows come after the blocks of the slab
    o = np.lexsort((gi, gj, gk, block, slab))
    ids, gi, gj, gk, slab = ids[o], gi[o], gj[o], gk[o], slab[o]
    cut = np.ones(ids.size, dtype=bool)                                 # entry starts a run
    cut[1:] = (gi[1:] != gi[:-1] + 1) | (gj[1:] != gj[:-1]) | (gk[1:] != gk[:-1]) | (slab[1:] != slab[:-1])
    start = np.nonzero(cut)[0]
    length = np.diff(np.append(start, ids.size))
    padded = length + (length % 8 != 0)
    run_off = np.concatenate([[0], np.cumsum(padded)])
    out = np.full(int(run_off[-1]), LIST_SKIP, dtype=np.uint32)
    run_of = np.cumsum(cut) - 1
    out[run_off[run_of] + (np.arange(ids.size) - start[run_of])] = ids
    run_slab = slab[start]
    pieces = [out[int(run_off[:-1][run_slab == x].min()):int(run_off[1:][run_slab == x].max())] if (run_slab == x).any()
              else np.zeros(0, dtype=np.uint32) for x in range(nxcd)]
    return assemble(pieces, nxcd=nxcd)


def _box(grid, lo, hi):
    return grid[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]


def shell_list(g, lo, hi, ilo, ihi, rows_per_block=8, kpair=False, icols=0):
    """Tiled list of the bricks of box [lo,hi) of the logical grid `g` that are outside box [ilo,ihi): two k-slabs in the
    per-XCD slab order, the four j/i side walls row by row (kpair: the slabs in the k-paired order; the walls stay plain
    rows, which a k-paired sweep computes as unpaired entries)."""
    if any(ihi[a] <= ilo[a] for a in range(3)):
        return tiled(_box(g, lo, hi), rows_per_block, kpair=kpair, icols=icols)
    klo = slabs(_box(g, lo, [hi[0], hi[1], ilo[2]]), rows_per_block, kpair=kpair, icols=icols)
    khi = slabs(_box(g, [lo[0], lo[1], ihi[2]], hi), rows_per_block, kpair=kpair, icols=icols)
    rest = [_box(g, [lo[0], lo[1], ilo[2]], [hi[0], ilo[1], ihi[2]]),        # j-low / j-high
            _box(g, [lo[0], ihi[1], ilo[2]], [hi[0], hi[1], ihi[2]]),
            _box(g, [lo[0], ilo[1], ilo[2]], [ilo[0], ihi[1], ihi[2]]),      # i-low / i-high
            _box(g, [ihi[0], ilo[1], ilo[2]], [hi[0], ihi[1], ihi[2]])]
    rest = np.concatenate([np.ascontiguousarray(r).reshape(-1) for r in rest])
    return assemble(klo, khi, chunks(rest))


def decomp_lists(d, rows_per_block=8, kpair=False, icols=0):
    """Brick lists for the weak/strong sweep schedule of a BrickDecomp `d` (see drivers.weak_cycle), all in the geometric
    (logical-grid) tiled order -- i fastest across ghost / surface / inner bricks alike, so runs of i-neighbours share
    halos whatever their ids are:
    inner   -- [1,sep0): bricks that never read a ghost brick
    shell   -- [sep0,sep2): surface + ghost ring            surface -- [sep0,sep1)
    all     -- [1,sep2): subdomain + ghost ring             domain  -- [1,sep1): the subdomain itself.
    kpair: every list in the k-paired order (for sweeps with apply_stencil(..., kpaired=True): the 25-point star)."""
    g = np.asarray(d.grid).astype(np.uint32)
    T = [g.shape[2], g.shape[1], g.shape[0]]
    gb = d.depth // 8                                   # ghost width in bricks
    lo0, hi0 = [0] * 3, T
    lo1, hi1 = [gb] * 3, [T[a] - gb for a in range(3)]
    lo2, hi2 = [2 * gb] * 3, [T[a] - 2 * gb for a in range(3)]
    has_inner = all(hi2[a] > lo2[a] for a in range(3))
    inner = tiled(_box(g, lo2, hi2), rows_per_block, kpair=kpair, icols=icols) if has_inner else np.zeros(0, dtype=np.uint32)
    assert (inner != LIST_SKIP).sum() == d.sep_pos[0] - 1
    return {"inner": inner, "shell": shell_list(g, lo0, hi0, lo2, hi2, rows_per_block, kpair, icols),
            "surface": shell_list(g, lo1, hi1, lo2, hi2, rows_per_block, kpair, icols),
            "all": tiled(_box(g, lo0, hi0), rows_per_block, kpair=kpair, icols=icols),
            "domain": tiled(_box(g, lo1, hi1), rows_per_block, kpair=kpair, icols=icols)}


def trapezoid_lists(d, sweeps, rows_per_block=8, kpair=False, icols=0):
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
        has = all(ihi[a] > ilo[a] for a in range(3))   # False: the cone has closed, everything waits for the exchange
        inner.append(tiled(_box(g, ilo, ihi), rows_per_block, kpair=kpair, icols=icols) if has else np.zeros(0, dtype=np.uint32))
        shell.append(shell_list(g, lo, hi, ilo, ihi, rows_per_block, kpair, icols))
    return inner, shell


def last_shell_split(d, sweeps, rows_per_block=8, kpair=False, icols=0):
    """The late pass of the LAST sweep of the deep schedule (trapezoid_lists: shell[sweeps-1] = bricks with 1 <= L <= sweeps) in two
    parts: (surface, rest) = (L == 1: the bricks the next exchange sends, L in [2, sweeps]).  drivers.weak_cycle_deep sweeps the
    surface first so that the next exchange can start under the rest."""
    g = np.asarray(d.grid).astype(np.uint32)
    T = [g.shape[2], g.shape[1], g.shape[0]]
    t = sweeps
    ilo, ihi = [t + 1] * 3, [T[a] - t - 1 for a in range(3)]
    lo1, hi1 = [1] * 3, [T[a] - 1 for a in range(3)]
    lo2, hi2 = [2] * 3, [T[a] - 2 for a in range(3)]
    surface = shell_list(g, lo1, hi1, lo2, hi2, rows_per_block, kpair, icols)
    if any(hi2[a] <= lo2[a] for a in range(3)):
        return surface, np.zeros(0, dtype=np.uint32)
    rest = shell_list(g, lo2, hi2, ilo, ihi, rows_per_block, kpair, icols)
    return surface, rest
