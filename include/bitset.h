// bitset.h -- direction sets (reference include/bitset.h:20-126): a set of signed axis numbers +-1..+-31 packed into 64
// bits, +p in bit p, -p in bit 31+p.  Same members and operators as the reference's BitSet, written for this library.
#pragma once

#include <cstdint>
#include <initializer_list>
#include <ostream>

struct BitSet {
  static constexpr long zero = 31; // -p lives `zero` bits above +p
  uint64_t set = 0;

  static constexpr uint64_t to_set(long pos) { return uint64_t(1) << uint64_t(pos < 0 ? zero - pos : pos); }

  constexpr BitSet() = default;
  constexpr BitSet(uint64_t bits) : set(bits) {}
  BitSet(std::initializer_list<int> elems) {
    for (int e : elems) set ^= to_set(e);
  }

  BitSet &flip(long pos) {
    set ^= to_set(pos);
    return *this;
  }
  long size() const { return __builtin_popcountll(set); }
  bool get(long pos) const { return (set & to_set(pos)) != 0; }

  BitSet operator&(BitSet o) const { return BitSet(set & o.set); }
  BitSet operator|(BitSet o) const { return BitSet(set | o.set); }
  BitSet operator^(BitSet o) const { return BitSet(set ^ o.set); }
  explicit operator bool() const { return set != 0; }
  bool operator<=(BitSet o) const { return (set & o.set) == set; }     // subset
  bool operator>=(BitSet o) const { return (set & o.set) == o.set; }   // superset
  bool operator==(BitSet o) const { return set == o.set; }
  bool operator!=(BitSet o) const { return set != o.set; }
  // every direction reversed: the low and the high half trade places
  BitSet operator!() const {
    const uint64_t low = (uint64_t(1) << (zero + 1)) - 1;
    return BitSet(((set & low) << zero) | (set >> zero));
  }
};

inline std::ostream &operator<<(std::ostream &os, const BitSet &bs) {
  os << "{";
  for (long p = 1; p <= BitSet::zero; ++p) {
    if (bs.get(p)) os << p << "+";
    if (bs.get(-p)) os << p << "-";
  }
  return os << "}";
}
