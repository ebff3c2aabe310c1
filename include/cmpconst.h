// cmpconst.h -- comparison tolerance (reference include/cmpconst.h:9)
#ifndef BRICK_CMPCONST_H
#define BRICK_CMPCONST_H
#define BRICK_TOLERANCE 1e-6
#endif
