// vecscatter.h -- element type and the brick(...) / tile(...) stencil macros (API of the reference's
// include/vecscatter.h:12-90, written fresh for the MI355X engine).
//
// In the reference these macros become a `#pragma vecscatter` that the Python generator codegen/vecscatter replaces
// with generated code.  Here the same source line is rewritten by tools/brkscatter (same command line as the
// reference tool, so cmake/VecScatter.cmake's VSTARGET keeps working) into a call of the hand-written device code in
// brick-device.h; a `tile(...)` line (the stencil over a plain array tile, reference include/vecscatter.h:21-54) is rewritten
// into a plain loop nest over the tile with one lane per element of the contiguous axis -- arrays are not this engine's
// data layout, the form exists so that the reference's drivers (stencils/3axis.hip.cpp:70-76) build unchanged.
// A file that is compiled WITHOUT the rewrite step fails to compile (the reference silently
// compiles such a file into a kernel that does nothing, because an unknown pragma is ignored).
#ifndef BRICK_VECSCATTER_H
#define BRICK_VECSCATTER_H

#ifndef bElem
#define bElem double
#endif

#define BRK_STR_(...) #__VA_ARGS__
#define BRK_STR(...) BRK_STR_(__VA_ARGS__)

#ifdef BRK_SCATTER_PASS
// preprocessing pass of tools/brkscatter: leave a marker it can find, with file/line of the call site
#define brick(file, vec, vsdim, vsfold, ...) \
  _Pragma(BRK_STR(brkscatter Brick(__FILE__, __LINE__, file, vec, BRK_STR(__VA_ARGS__), dim = vsdim, fold = vsfold)))
#define tile(file, vec, vsdim, titer, ...) \
  _Pragma(BRK_STR(brkscatter Tile(__FILE__, __LINE__, file, vec, dim = vsdim, tile_iter = titer)))
#else
template <typename T>
struct brk_not_rewritten {
  static constexpr bool value = false;
};
#define brick(...) \
  static_assert(brk_not_rewritten<struct brk_tag>::value, "brick(...): run tools/brkscatter on this file (VSTARGET) before compiling")
#define tile(...) \
  static_assert(brk_not_rewritten<struct brk_tag>::value, "tile(...): run tools/brkscatter on this file (VSTARGET) before compiling")
#endif

#endif // BRICK_VECSCATTER_H
