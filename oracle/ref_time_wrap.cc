// CPU ORACLE — TEST INFRASTRUCTURE ONLY.
// C entry points over the REFERENCE's own calendar code, compiled from the sources where they
// lie (/root/reference/src/util/time.{h,cc}); nothing of the reference is copied into this repo.
// Built only in the authoring container (oracle/build_oracle.py -> oracle/_ref/libviya_time.so)
// and used to (1) validate oracle/viya_oracle.py's gmtime/timegm/trunc/Duration restatement and
// (2) generate tests/golden/time_golden.json (oracle/make_time_golden.py).
#include "util/time.h"
#include <cstdint>

namespace vu = viya::util;

template <typename TimeT, typename V> static V trunc_any(V ts, int unit) {
  TimeT t;
  t.set_ts(ts);
  switch (unit) {
    case vu::TimeUnit::YEAR: t.template trunc<vu::TimeUnit::YEAR>(); break;
    case vu::TimeUnit::MONTH: t.template trunc<vu::TimeUnit::MONTH>(); break;
    case vu::TimeUnit::DAY: t.template trunc<vu::TimeUnit::DAY>(); break;
    case vu::TimeUnit::HOUR: t.template trunc<vu::TimeUnit::HOUR>(); break;
    case vu::TimeUnit::MINUTE: t.template trunc<vu::TimeUnit::MINUTE>(); break;
    case vu::TimeUnit::SECOND: t.template trunc<vu::TimeUnit::SECOND>(); break;
    default: break;  // no Truncator::trunc<WEEK> exists
  }
  return t.get_ts();
}

extern "C" {
uint32_t ref_trunc32(uint32_t ts, int unit) { return trunc_any<vu::Time32, uint32_t>(ts, unit); }
uint64_t ref_trunc64(uint64_t ts, int unit) { return trunc_any<vu::Time64, uint64_t>(ts, unit); }
uint32_t ref_duration_add32(int unit, uint64_t count, uint32_t ts, int sign) {
  return vu::Duration(static_cast<vu::TimeUnit>(unit), count).add_to(ts, sign);
}
uint64_t ref_duration_add64(int unit, uint64_t count, uint64_t ts, int sign) {
  return vu::Duration(static_cast<vu::TimeUnit>(unit), count).add_to(ts, sign);
}
int ref_duration_parse(const char* desc, int* unit, uint64_t* count) {
  try {
    vu::Duration d{std::string(desc)};
    *unit = d.time_unit();
    *count = d.count();
    return 0;
  } catch (...) {
    return -1;
  }
}
}
