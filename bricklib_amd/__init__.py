"""bricklib_amd -- MI355X-native bricked stencil engine (hot path of benSepanski/bricklib).

Host-side mirror of the reference's Brick / BrickInfo / BrickStorage / BrickDecomp interface on top of the
C ABI in include/brk.h.  PyTorch provides device memory, streams and torch.distributed (RCCL) only; all
compute is in bricklib_amd/csrc/libbrk.so (hand-written gfx950 HIP)."""
from ._lib import (BRK_7PT_COEFF, BRK_27PT_COEFF, BRK_MPI7, BRK_MPI13, BRK_MPI25, BRK_MPI125, BrkError, lib)  # noqa: F401
from .brick import (Brick, BrickInfo, BrickStorage, compareBrick, copyFromBrick, copyToBrick, init_grid,  # noqa: F401
                    movBrickInfo, movBrickStorage)
from .stencils import STENCILS, CompiledExpr, Zigzag, apply_expr, apply_script, apply_stencil, apply_taps, finalize, last_kernel, stencil_iters, stencil_radius  # noqa: F401
from .decomp import BrickDecomp, allneighbors, dims_create, populate  # noqa: F401
