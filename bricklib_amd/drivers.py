"""Driver-level schedules of the reference, restated on device streams.

weak_cycle == one `brick_func` call of weak/main.cpp:172-213: exchange the ghosts of A,
then ST_ITER sweeps ping-ponging A/B, the last one skipping the outer brick ring, result back in A.  Thanks to
the region ordering of BrickDecomp the three brick sets involved are plain id ranges:
    [1, sep0)  inner bricks -- never read a ghost brick, so their first sweep overlaps the exchange
    [1, sep1)  the subdomain itself (what the last sweep computes; `skip = 1` in the reference)
    [1, sep2)  subdomain + ghost ring (what every other sweep computes)
"""
import torch

from .stencils import apply_stencil, stencil_iters


def weak_cycle(stencil, d, A, B, exchange, coeff=None, main=None, comm=None, overlap=True):
    """A, B: Brick objects over two storages built from decomposition `d`; exchange(stream) refreshes A's ghosts.
    main/comm: torch streams (default: current stream, no overlap)."""
    S = stencil_iters(stencil)
    sep0, sep1, sep2 = d.sep_pos
    main = main or torch.cuda.current_stream()
    if comm is None or not overlap:
        exchange(main)
        with torch.cuda.stream(main):
            apply_stencil(stencil, A, B, brick_range=(1, sep2), coeff=coeff, stream=main)
    else:
        ev0 = torch.cuda.Event()
        ev0.record(main)                 # A as left by the previous cycle
        comm.wait_event(ev0)
        exchange(comm)                   # ghosts of A arrive on the comm stream ...
        ev1 = torch.cuda.Event()
        ev1.record(comm)
        apply_stencil(stencil, A, B, brick_range=(1, sep0), coeff=coeff, stream=main)   # ... while the inner bricks sweep
        main.wait_event(ev1)
        apply_stencil(stencil, A, B, brick_range=(sep0, sep2), coeff=coeff, stream=main)
    for _ in range(S // 2 - 1):
        apply_stencil(stencil, B, A, brick_range=(1, sep2), coeff=coeff, stream=main)
        apply_stencil(stencil, A, B, brick_range=(1, sep2), coeff=coeff, stream=main)
    apply_stencil(stencil, B, A, brick_range=(1, sep1), coeff=coeff, stream=main)
    return S
