"""Driver-level schedules of the reference, restated on device streams.

weak_cycle == one `brick_func` call of weak/main.cpp:172-213: exchange the ghosts of A,
then ST_ITER sweeps ping-ponging A/B, the last one skipping the outer brick ring, result back in A.  Thanks to
the region ordering of BrickDecomp the three brick sets involved are plain id ranges:
    [1, sep0)  inner bricks -- never read a ghost brick, so their first sweep overlaps the exchange
    [1, sep1)  the subdomain itself (what the last sweep computes; `skip = 1` in the reference)
    [1, sep2)  subdomain + ghost ring (what every other sweep computes)
"""
import torch

from .brick import grid_to_device
from .order import last_shell_split, decomp_lists, trapezoid_lists
from .stencils import Zigzag, apply_stencil, stencil_iters
from .timing import marker


class SweepLists:
    """Device-resident brick lists of one BrickDecomp for the sweep schedule (same brick sets as the id ranges in
    the module docstring, but in the cache-friendly order of bricklib_amd.order).  kpair: the k-paired order (two bricks per
    wave for the 25-point star, brk_stencil_apply_list_kpaired); the sweeps below read it from `kpaired`."""

    def __init__(self, d, device, rows_per_block=8, kpair=False, icols=0):
        self.kpaired = bool(kpair)
        for name, ids in decomp_lists(d, rows_per_block, kpair=kpair, icols=icols).items():
            setattr(self, name, grid_to_device(ids, device))


def weak_cycle(stencil, d, A, B, exchange, coeff=None, main=None, comm=None, overlap=True, lists=None, state=None, zigzag=False):
    """A, B: Brick objects over two storages built from decomposition `d`; exchange(stream) refreshes A's ghosts.
    main/comm: torch streams (default: current stream, no overlap).  lists: SweepLists(d) (optional; without it the
    sweeps run over plain id ranges).  state: dict carried from one cycle to the next (optional): the last sweep writes
    the surface bricks first and records an event, so that the NEXT cycle's exchange can start while the inner bricks
    of the last sweep -- and then of the first sweep -- are still being computed (the exchange only reads surface bricks
    of A and writes ghost bricks of A, which the last sweep neither writes nor reads).
    zigzag: successive sweeps walk their bricks in alternating directions (BRK_SWEEP_REVERSE).  Same results; off by default: a
    ping-pong sweep reads what the one before WROTE, and the non-temporal stores do not leave that in the Infinity Cache
    (measured: 1.887 vs 1.882 ms per cycle)."""
    S = stencil_iters(stencil)
    sep0, sep1, sep2 = d.sep_pos
    main = main or torch.cuda.current_stream()
    zz = Zigzag()

    def sweep(src, dst, which):
        with marker("calc:" + which):
            _sweep(src, dst, which)

    def _sweep(src, dst, which):
        rev = zigzag and zz.turn()
        if lists is not None:
            apply_stencil(stencil, src, dst, ids=getattr(lists, which), coeff=coeff, stream=main, kpaired=lists.kpaired, reverse=rev)
        else:
            rng = {"inner": (1, sep0), "shell": (sep0, sep2), "surface": (sep0, sep1), "all": (1, sep2),
                   "domain": (1, sep1)}[which]
            apply_stencil(stencil, src, dst, brick_range=rng, coeff=coeff, stream=main, reverse=rev)

    if comm is None or not overlap:
        exchange(main)
        sweep(A, B, "all")
    else:
        ev0 = state.pop("surface_done", None) if state is not None else None
        if ev0 is None:
            ev0 = torch.cuda.Event()
            ev0.record(main)             # A as left by the previous cycle
        comm.wait_event(ev0)
        exchange(comm)                   # ghosts of A arrive on the comm stream ...
        ev1 = torch.cuda.Event()
        ev1.record(comm)
        sweep(A, B, "inner")             # ... while the inner bricks sweep
        main.wait_event(ev1)
        sweep(A, B, "shell")
    for _ in range(S // 2 - 1):
        sweep(B, A, "all")
        sweep(A, B, "all")
    if state is not None and comm is not None and overlap:
        sweep(B, A, "surface")           # what the next exchange sends ...
        ev = torch.cuda.Event()
        ev.record(main)
        state["surface_done"] = ev
        sweep(B, A, "inner")             # ... is ready before the bulk of the last sweep
    else:
        sweep(B, A, "domain")
    return S


class TrapezoidLists:
    """Device-resident lists of bricklib_amd.order.trapezoid_lists for `sweeps` sweeps per exchange."""

    def __init__(self, d, sweeps, device, rows_per_block=8, kpair=False, icols=0):
        self.kpaired = bool(kpair)
        inner, shell = trapezoid_lists(d, sweeps, rows_per_block, kpair=kpair, icols=icols)
        self.inner = [grid_to_device(x, device) if x.size else None for x in inner]
        self.shell = [grid_to_device(x, device) if x.size else None for x in shell]
        # the late pass of the LAST sweep in two parts: the surface bricks (what the next exchange sends) and the rest of its shell
        surf, rest = last_shell_split(d, sweeps, rows_per_block, kpair=kpair, icols=icols)
        self.last_surface = grid_to_device(surf, device)
        self.last_rest = grid_to_device(rest, device) if rest.size else None


def weak_cycle_deep(stencil, d, A, B, exchange, lists, coeff=None, main=None, comm=None, depth=None, full=None, zigzag=False, state=None):
    """Same result as weak_cycle, scheduled so that the exchange hides behind most of the cycle instead of one sweep
    (xGMI is a per-link-bound point-to-point fabric; a 512^3 subdomain sends 104 MB per exchange).  While the ghosts
    of A travel on `comm`, `main` runs the part of EVERY sweep that cannot depend on them yet -- sweep t over the bricks
    at least t+1 bricks inside the ghosted grid (a shrinking box, 59 % of all brick updates at 512^3 / 8 sweeps) -- and
    afterwards the remaining shells of sweeps 1..S in order.  Ping-pong safety: the early pass of sweep t+2 overwrites
    the source of sweep t+1 only at depth >= t+3, while the late pass of sweep t+1 reads depth <= t+2.
    lists: TrapezoidLists(d, stencil_iters(stencil)).
    depth (1..S, default S): only the first `depth` sweeps are split this way (2*depth launches, hiding the early passes
    of those sweeps); the remaining sweeps run whole over full.all / full.domain (full: SweepLists(d)).  A shallower
    schedule hides less exchange time and costs fewer, larger launches.
    state (round 5): dict carried from one cycle to the next.  With it the late pass of the last sweep writes the surface bricks
    first and records an event, and the NEXT cycle's exchange starts behind that event -- under the rest of that pass -- instead of
    behind the whole cycle (the exchange reads surface bricks of A and writes ghost bricks of A; the rest of the pass writes A two
    and more bricks inside and reads B).  Same field bit for bit."""
    S = stencil_iters(stencil)
    assert len(lists.inner) == S
    K = S if depth is None else max(1, min(int(depth), S))
    assert K == S or full is not None, "depth < ST_ITER needs the whole-sweep lists (full=SweepLists(d))"
    main = main or torch.cuda.current_stream()
    zz = Zigzag()
    early = state is not None and comm is not None      # hand the surface of the last sweep over early
    if comm is None:
        exchange(main)
    else:
        ev0 = state.pop("surface_done", None) if state is not None else None
        if ev0 is None:
            ev0 = torch.cuda.Event()
            ev0.record(main)             # A as left by the previous cycle
        comm.wait_event(ev0)
        exchange(comm)
        ev1 = torch.cuda.Event()
        ev1.record(comm)

    def surface_first(src, dst, surface, rest, kpaired):
        apply_stencil(stencil, src, dst, ids=surface, coeff=coeff, stream=main, kpaired=kpaired, reverse=zigzag and zz.turn())
        ev = torch.cuda.Event()
        ev.record(main)
        state["surface_done"] = ev
        if rest is not None:
            apply_stencil(stencil, src, dst, ids=rest, coeff=coeff, stream=main, kpaired=kpaired, reverse=zigzag and zz.turn())

    for ids in (lists.inner, lists.shell):
        src, dst = A, B
        for t in range(K):
            if early and ids is lists.shell and t == S - 1:
                surface_first(src, dst, lists.last_surface, lists.last_rest, lists.kpaired)
            elif ids[t] is not None:
                apply_stencil(stencil, src, dst, ids=ids[t], coeff=coeff, stream=main, kpaired=lists.kpaired, reverse=zigzag and zz.turn())
            src, dst = dst, src
        if comm is not None and ids is lists.inner:
            main.wait_event(ev1)
    for t in range(K, S):   # sweeps K+1..S whole; src/dst continue the ping-pong (sweep t+1 reads what sweep t wrote)
        if early and t == S - 1:
            surface_first(src, dst, full.surface, full.inner, full.kpaired)
        else:
            apply_stencil(stencil, src, dst, ids=full.domain if t == S - 1 else full.all, coeff=coeff, stream=main, kpaired=full.kpaired,
                          reverse=zigzag and zz.turn())
        src, dst = dst, src
    return S
