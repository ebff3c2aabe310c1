# A user-written stencil in the reference's script syntax: a 19-point "face + edge" stencil with one coefficient per
# shell (centre, 6 faces, 12 edges).  Not one of the built-in stencils: tools/brkscatter reads it as data and routes the
# brick("lap19.py", ...) line to the generic tap path (brklib::brick_apply_taps_static).
from st.expr import Index, ConstRef
from st.grid import Grid

i, j, k = Index(0), Index(1), Index(2)
bIn = Grid("bIn", 3)
bOut = Grid("bOut", 3)
centre, face, edge = ConstRef("coeff[0]"), ConstRef("coeff[1]"), ConstRef("coeff[2]")

acc = centre * bIn(i, j, k)
for s in (1, -1):
    acc = acc + face * (bIn(i + s, j, k) + bIn(i, j + s, k) + bIn(i, j, k + s))
    for t in (1, -1):
        acc = acc + edge * (bIn(i + s, j + t, k) + bIn(i + s, j, k + t) + bIn(i, j + s, k + t))
bOut(i, j, k).assign(acc)

STENCIL = [bOut]
