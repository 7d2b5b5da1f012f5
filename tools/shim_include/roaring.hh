// tools/shim_include/roaring.hh — NOT CRoaring. DECLARATIONS ONLY of the members of CRoaring's C++ class `Roaring` that the reference's
// util/bitset.h (src/util/bitset.h:33-66: add, |=, cardinality, contains, runOptimize) and the generated shim text (toUint32Array,
// cardinality) name, spelled as CRoaring's cpp/roaring.hh spells them. The reference's third_party/CRoaring submodule is empty in this
// mount; this header exists so that tests/test_shim_compile.py can run `g++ -fsyntax-only` / `-c` on the text generated for a table
// with a bitset metric against the reference's REAL util/bitset.h. It pins SYNTAX (names, arity, constness), not behaviour: nothing is
// defined here, nothing links against it, and no product code includes it.
#ifndef VIYA_SHIM_DECL_ROARING_HH_
#define VIYA_SHIM_DECL_ROARING_HH_
#include <cstdint>
class Roaring {
public:
  Roaring();
  Roaring(const Roaring&);
  Roaring& operator=(const Roaring&);
  ~Roaring();
  void add(uint32_t x);
  bool contains(uint32_t x) const;
  uint64_t cardinality() const;
  Roaring& operator|=(const Roaring& r);
  bool runOptimize();
  void toUint32Array(uint32_t* ans) const;
};
#endif
