// tools/shim_include/roaring64map.hh — NOT CRoaring. Declarations only of what the reference's util/bitset.h and the generated shim text
// name of CRoaring's `Roaring64Map` (cpp/roaring64map.hh); see roaring.hh next to this file for why it exists and what it does not pin.
#ifndef VIYA_SHIM_DECL_ROARING64MAP_HH_
#define VIYA_SHIM_DECL_ROARING64MAP_HH_
#include <cstdint>
#include "roaring.hh"
class Roaring64Map {
public:
  Roaring64Map();
  Roaring64Map(const Roaring64Map&);
  Roaring64Map& operator=(const Roaring64Map&);
  ~Roaring64Map();
  void add(uint64_t x);
  bool contains(uint64_t x) const;
  uint64_t cardinality() const;
  Roaring64Map& operator|=(const Roaring64Map& r);
  bool runOptimize();
  void toUint64Array(uint64_t* ans) const;
};
#endif
