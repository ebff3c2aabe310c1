// zmort.h -- Z-Morton indices (reference include/zmort.h:18-105): coordinate d of a `dim`-dimensional point occupies
// bits d, d+dim, d+2*dim, ... of the index.  Same interface as the reference's ZMORT (operator[] appends a coordinate,
// operator() extracts one, set() replaces one); the bit work goes through the library (brk_zmort_get / brk_zmort_set),
// which the strong-scaling bookkeeping uses as well.
#pragma once

#include "brk.h"

struct ZMORT {
  unsigned long id = 0;  // the interleaved index
  unsigned long dim = 0; // number of coordinates interleaved so far

  ZMORT() = default;
  ZMORT(unsigned long id_, unsigned long dim_) : id(id_), dim(dim_) {}

  // a (dim+1)-dimensional index: the old coordinates keep their order, p becomes coordinate 0
  ZMORT operator[](unsigned long p) const {
    ZMORT out(0, dim + 1);
    for (unsigned long bit = 0; (id >> (bit * (dim ? dim : 1))) || (p >> bit); ++bit) {
      const unsigned long old_bits = dim ? (id >> (bit * dim)) & ((1ul << dim) - 1ul) : 0ul;
      out.id |= ((old_bits << 1) | ((p >> bit) & 1ul)) << (bit * out.dim);
      if (!dim && !(p >> (bit + 1))) break;
    }
    return out;
  }
  // coordinate d
  unsigned long operator()(unsigned long d) const {
    unsigned long pos = 0;
    for (unsigned long bit = 0, rest = id >> d; rest; ++bit, rest >>= dim) pos |= (rest & 1ul) << bit;
    return pos;
  }
  // the same point with coordinate d replaced by p
  ZMORT set(unsigned long d, unsigned long p) const {
    ZMORT out(id, dim);
    const unsigned long old = (*this)(d);
    for (unsigned long bit = 0; (old >> bit) || (p >> bit); ++bit) {
      const unsigned long where = d + bit * dim;
      out.id = (out.id & ~(1ul << where)) | (((p >> bit) & 1ul) << where);
    }
    return out;
  }
  operator unsigned long() const { return id; }
};
