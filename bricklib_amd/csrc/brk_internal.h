// brk_internal.h -- shared between the translation units of libbrk.so (not installed).
#ifndef BRK_INTERNAL_H
#define BRK_INTERNAL_H
#include "brk.h"

// records the message for brk_last_error() and returns `code`
int brk_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#endif
