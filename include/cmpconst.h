// cmpconst.h -- acceptance tolerance of compareBrick / compareArray: a and b agree when |a-b| < tol or |a-b| < (|a|+|b|) tol
// (reference include/cmpconst.h:9, used by include/brickcompare.h:36-37 and src/multiarray.cpp:57-58).
#pragma once
#ifndef BRICK_TOLERANCE
#define BRICK_TOLERANCE 1e-6
#endif
