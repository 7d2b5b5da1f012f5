// vh_time.h — Truncator::trunc<U> of the reference (src/util/time.h:57-89: gmtime_r -> zero the finer tm fields -> timegm) as integer
// civil-calendar arithmetic, proleptic Gregorian, exact for every non-negative time_t (checked against the reference's own util/time.cc:
// tests/golden/time_golden.json, tests/test_time_golden.py). Plain C++: the kernels include it (vh_kernels.h, the compiled scans through the
// embedded headers) and so does a host-compiled check of the 32-bit form against the 64-bit one (tests/test_time_golden.py).
//
// Two forms. The 64-bit one takes any time; on the GPU its ~10 divisions of 64-bit integers by constants are ~40 instructions each
// (a 64 x 64 -> 128-bit multiply out of 32-bit pieces): C5's scan spent ~250 VALU instructions per surviving row in its time rollup.
// The 32-bit one takes seconds that fit 32 bits (every `time` column; a `microtime` column's seconds until the year 2106): all days and
// years are small non-negative numbers, so the same arithmetic runs unsigned in 32 bits — a division is one v_mul_hi_u32 and a shift — and
// a MONTH needs no way back from (y, m, d) to days at all: the first of the month is the day itself minus the day of the month.
#pragma once
#include <stdint.h>
#ifndef VH_TIME_FN
#ifdef __HIPCC__
#define VH_TIME_FN __host__ __device__ __forceinline__
#else
#define VH_TIME_FN static inline
#endif
#endif
// enum vh_time_unit (include/viya_hip.h), restated so that this file stands alone
#define VHT_YEAR 0
#define VHT_MONTH 1
#define VHT_DAY 3
#define VHT_HOUR 4
#define VHT_MINUTE 5
#define VHT_SECOND 6

VH_TIME_FN int64_t vh_days_from_civil(int64_t y, int64_t m, int64_t d) {
  y -= m <= 2;
  const int64_t era = (y >= 0 ? y : y - 399) / 400;
  const int64_t yoe = y - era * 400;
  const int64_t doy = (153 * (m + (m > 2 ? -3 : 9)) + 2) / 5 + d - 1;
  const int64_t doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
  return era * 146097 + doe - 719468;
}
VH_TIME_FN void vh_civil_from_days(int64_t z, int64_t& y, int64_t& m, int64_t& d) {
  z += 719468;
  const int64_t era = (z >= 0 ? z : z - 146096) / 146097;
  const int64_t doe = z - era * 146097;
  const int64_t yoe = (doe - doe / 1460 + doe / 36524 - doe / 146096) / 365;
  y = yoe + era * 400;
  const int64_t doy = doe - (365 * yoe + yoe / 4 - yoe / 100);
  const int64_t mp = (5 * doy + 2) / 153;
  d = doy - (153 * mp + 2) / 5 + 1;
  m = mp + (mp < 10 ? 3 : -9);
  y += m <= 2;
}
// seconds since the epoch, any value. WEEK has no specialisation in the reference and is rejected when the plan is built.
VH_TIME_FN uint64_t vh_trunc_secs64(uint64_t t, int unit) {
  switch (unit) {
    case VHT_SECOND: return t;
    case VHT_MINUTE: return t - t % 60u;
    case VHT_HOUR: return t - t % 3600u;
    case VHT_DAY: return t - t % 86400u;
    default: {
      int64_t y, m, d;
      vh_civil_from_days((int64_t)(t / 86400u), y, m, d);
      if (unit == VHT_YEAR) m = 1;
      return (uint64_t)vh_days_from_civil(y, m, 1) * 86400u;
    }
  }
}
// seconds that fit 32 bits: days since the epoch < 49 711, years 1970 .. 2106
VH_TIME_FN uint32_t vh_trunc_secs32(uint32_t t, int unit) {
  switch (unit) {
    case VHT_SECOND: return t;
    case VHT_MINUTE: return t - t % 60u;
    case VHT_HOUR: return t - t % 3600u;
    case VHT_DAY: return t - t % 86400u;
    default: {
      const uint32_t days = t / 86400u;
      const uint32_t z = days + 719468u;                     // days since 0000-03-01
      const uint32_t era = z / 146097u;
      const uint32_t doe = z - era * 146097u;                // [0, 146096]
      const uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;      // [0, 399]
      const uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);                     // [0, 365], counted from March 1st
      const uint32_t mp = (5u * doy + 2u) / 153u;            // [0, 11], March = 0
      if (unit != VHT_YEAR) return (days - (doy - (153u * mp + 2u) / 5u)) * 86400u;        // MONTH: the day minus (day of the month - 1)
      // YEAR: January 1st of the civil year. January and February (mp >= 10) belong to the March-year that began the year before:
      // its January 1st is that March-year's day 306; for March .. December it lies 59 days (60 in a leap year) before this March 1st
      if (mp >= 10u) return (days - (doy - 306u)) * 86400u;
      const uint32_t y = yoe + era * 400u;                   // the civil year of March .. December
      const uint32_t leap = (y % 4u == 0u && (y % 100u != 0u || y % 400u == 0u)) ? 1u : 0u;
      return (days - doy - 59u - leap) * 86400u;
    }
  }
}
VH_TIME_FN uint64_t vh_trunc_secs(uint64_t t, int unit) {
  return t <= 0xFFFFFFFFull ? (uint64_t)vh_trunc_secs32((uint32_t)t, unit) : vh_trunc_secs64(t, unit);
}
