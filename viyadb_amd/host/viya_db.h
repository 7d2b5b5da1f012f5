// viya_db.h — host-side mirror of the reference's db layer as far as the aggregate path needs it.
//
// Same names and meaning as the reference (so the parity tests read like its own tests):
//   db::NumericType / AnyNum            src/db/column.h:34-121
//   db::Column/Dimension/Metric kinds   src/db/column.h:123-316, src/db/column.cc:253-402
//   db::DimensionDict                   src/db/dictionary.h:36-52, dictionary.cc:24-75
//   db::Table / SegmentStore / Segment  src/db/table.h:53-91, src/db/store.h:32-58,
//                                       src/codegen/db/store.cc:203-356 (SoA layout, stats)
//   input::SimpleLoader + upsert        src/input/simple.cc:29-35, src/codegen/db/upsert.cc:29-151,340-421
//   util::Duration / Time32 / Time64    src/util/time.h:27-137, src/util/time.cc:49-83
//
// What is different by design: the reference JIT-compiles a Segment class per table; here a
// segment is a set of runtime-typed column arrays (the GPU mirror needs base pointers, not a
// C++ type), and each segment carries a version counter so the HBM mirror knows what to re-sync
// (SURVEY §8(f)-1).
#pragma once
#include <cstdint>
#include <cstring>
#include <ctime>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "json.h"

namespace viya {
namespace util {

enum TimeUnit { YEAR = 0, MONTH, WEEK, DAY, HOUR, MINUTE, SECOND, _UNDEFINED };
TimeUnit time_unit_by_name(const std::string& name);

class Duration {
public:
  explicit Duration(const std::string& desc);  // "<n> <unit>s"
  Duration(TimeUnit u, size_t count) : time_unit_(u), count_(count) {}
  TimeUnit time_unit() const { return time_unit_; }
  size_t count() const { return count_; }
  uint32_t add_to(uint32_t timestamp, int sign) const;
  uint64_t add_to(uint64_t timestamp, int sign) const;
  bool operator>(const Duration& d) const { return add_to((uint32_t)0, 1) > d.add_to((uint32_t)0, 1); }

private:
  TimeUnit time_unit_;
  size_t count_;
};

// Time32 / Time64 with a run-time unit (the reference templates on it).
void trunc_tm(std::tm& tm, TimeUnit unit);  // throws for WEEK: no Truncator::trunc<WEEK> exists
class TimeAny {
public:
  explicit TimeAny(bool micro) : micro_(micro), micros_(0), tm_{} {}
  void parse(const char* format, const std::string& value);
  void set_ts(uint64_t ts);
  uint64_t get_ts();
  void trunc(TimeUnit unit);

private:
  bool micro_;
  uint32_t micros_;
  std::tm tm_;
};

}  // namespace util

namespace db {

// ---- numeric types
enum class Num { BYTE, UBYTE, SHORT, USHORT, INT, UINT, LONG, ULONG, FLOAT, DOUBLE };

struct AnyNum {  // 8 bytes, the column's own type in the low bytes (src/db/column.h:98-121)
  uint64_t bits = 0;
  template <typename T> static AnyNum of(T v) { AnyNum a; std::memcpy(&a.bits, &v, sizeof(T)); return a; }
  template <typename T> T as() const { T v; std::memcpy(&v, &bits, sizeof(T)); return v; }
};

class NumericType {
public:
  explicit NumericType(Num t) : t_(t) {}
  explicit NumericType(const std::string& name);
  Num type() const { return t_; }
  int size() const;
  bool fp() const { return t_ == Num::FLOAT || t_ == Num::DOUBLE; }
  bool is_signed() const { return t_ == Num::BYTE || t_ == Num::SHORT || t_ == Num::INT || t_ == Num::LONG; }
  std::string cpp_type() const;
  AnyNum cpp_min_value() const;  // FLT_MIN / DBL_MIN for floating types, like the reference
  AnyNum cpp_max_value() const;
  AnyNum Parse(const std::string& value) const;  // std::sto* then a C cast
  int vh_elem() const;                           // include/viya_hip.h enum vh_elem

private:
  Num t_;
};
NumericType uint_type_for_max(uint64_t max_value);  // max_value_to_uint_type

// typed helpers on raw column storage
double load_as_double(const char* p, Num t);
std::string format_num(const char* p, Num t);  // util::Format::num
int compare_typed(Num t, AnyNum a, AnyNum b);  // -1 / 0 / +1 in the column's own type (NaN: 2)

// ---- columns
struct RollupRule {
  util::TimeUnit granularity;
  util::Duration after;
};

class DimensionDict {
public:
  explicit DimensionDict(int code_size) : size_(code_size) { c2v_.push_back("__exceeded"); v2c_["__exceeded"] = 0; }
  AnyNum Decode(const std::string& value) const;  // miss -> UINTn_MAX
  std::vector<std::string>& c2v() { return c2v_; }
  std::unordered_map<std::string, uint64_t>& v2c() { return v2c_; }
  int code_size() const { return size_; }

private:
  int size_;
  std::vector<std::string> c2v_;
  std::unordered_map<std::string, uint64_t> v2c_;
};

class Column {
public:
  enum Type { METRIC, DIMENSION };
  enum SortType { STRING, INTEGER, FLOAT };
  enum DimType { DIM_STRING, DIM_NUMERIC, DIM_TIME, DIM_BOOLEAN };
  enum AggregationType { MAX, MIN, SUM, AVG, COUNT, BITSET };

  Column(Type type, size_t index, const std::string& name, NumericType nt) : type_(type), index_(index), name_(name), num_type_(nt) {}
  Type type() const { return type_; }
  size_t index() const { return index_; }
  const std::string& name() const { return name_; }
  const std::string& input_field() const { return input_field_; }
  const NumericType& num_type() const { return num_type_; }
  SortType sort_type() const;

  // dimension attributes
  DimType dim_type() const { return dim_type_; }
  uint64_t cardinality() const { return cardinality_; }
  int length() const { return length_; }
  DimensionDict* dict() const { return dict_; }
  const std::string& format() const { return format_; }
  util::TimeUnit granularity() const { return granularity_; }
  const std::vector<RollupRule>& rollup_rules() const { return rollup_rules_; }
  bool micro_precision() const { return micro_; }
  // metric attributes
  AggregationType agg_type() const { return agg_type_; }

  // storage column index inside a Segment (dims first, then metrics, then hidden count)
  size_t storage_index = 0;

private:
  friend class Table;
  Type type_;
  size_t index_;
  std::string name_;
  std::string input_field_;
  NumericType num_type_;
  DimType dim_type_ = DIM_STRING;
  uint64_t cardinality_ = UINT32_MAX;
  int length_ = -1;
  DimensionDict* dict_ = nullptr;
  std::string format_;
  util::TimeUnit granularity_ = util::_UNDEFINED;
  std::vector<RollupRule> rollup_rules_;
  bool micro_ = false;
  AggregationType agg_type_ = SUM;
};
using Dimension = Column;
using Metric = Column;

struct CardinalityGuard {
  const Column* dim;
  std::vector<const Column*> dimensions;
  size_t limit;
  std::map<std::vector<uint64_t>, std::set<uint64_t>> stats;
};

// ---- storage
struct SegmentStat { AnyNum dmin, dmax; bool tracked = false; };

class Segment {
public:
  Segment(const class Table& table);
  size_t size() const { return size_; }
  size_t capacity() const { return capacity_; }
  bool full() const { return size_ == capacity_; }
  char* column(size_t storage_index) { return cols_[storage_index].get(); }
  const char* column(size_t storage_index) const { return cols_[storage_index].get(); }
  std::vector<std::vector<uint64_t>>& bitsets(size_t metric_index) { return bitsets_[metric_index]; }
  const std::vector<std::vector<uint64_t>>& bitsets(size_t metric_index) const { return bitsets_.at(metric_index); }
  std::vector<SegmentStat> stats;  // per dimension
  uint64_t version = 0;            // bumped on every append / in-place update
  size_t dirty_lo = SIZE_MAX, dirty_hi = 0;  // rows touched since the GPU mirror last synced this segment
  void touch(size_t row) { dirty_lo = row < dirty_lo ? row : dirty_lo; dirty_hi = row + 1 > dirty_hi ? row + 1 : dirty_hi; ++version; }

private:
  friend class Table;
  size_t size_ = 0, capacity_;
  std::vector<std::unique_ptr<char[]>> cols_;
  std::map<size_t, std::vector<std::vector<uint64_t>>> bitsets_;  // metric index -> per-row sorted id sets
};

class Dictionaries {
public:
  DimensionDict* GetOrCreate(const std::string& dim_name, int code_size);

private:
  std::map<std::string, std::unique_ptr<DimensionDict>> dicts_;
};

class Table {
public:
  Table(const util::Config& config, Dictionaries& dicts);
  ~Table();
  const std::string& name() const { return name_; }
  size_t segment_size() const { return segment_size_; }
  const std::vector<const Dimension*>& dimensions() const { return dim_ptrs_; }
  const std::vector<const Metric*>& metrics() const { return metric_ptrs_; }
  const Column* column(const std::string& name) const;
  const Dimension* dimension(const std::string& name) const;
  const Dimension* dimension(size_t i) const { return dim_ptrs_[i]; }
  const Metric* metric(const std::string& name) const;
  const Metric* metric(size_t i) const { return metric_ptrs_[i]; }
  std::vector<const Column*> columns() const;
  bool has_hidden_count() const { return has_hidden_count_; }
  size_t hidden_count_storage_index() const { return dims_.size() + metrics_.size(); }
  size_t storage_columns() const { return dims_.size() + metrics_.size() + (has_hidden_count_ ? 1 : 0); }
  int storage_elem_size(size_t storage_index) const;

  std::vector<std::unique_ptr<Segment>>& segments() { return segments_; }
  const std::vector<std::unique_ptr<Segment>>& segments() const { return segments_; }

  // input::SimpleLoader (BeforeLoad -> rows -> AfterLoad); `now` overrides std::time(nullptr) the way
  // VIYA_TEST_ROLLUP_TS does in the reference (src/codegen/db/rollup.cc:47-49)
  void Load(const std::vector<std::vector<std::string>>& rows, const std::vector<std::string>* columns = nullptr,
            int64_t now = -1);

  std::mutex mu;  // one writer / readers: the reference uses RWSpinLocks per segment and store
  void* gpu_mirror = nullptr;  // owned by query::GpuAggregate (opaque here)
  void (*gpu_mirror_free)(void*) = nullptr;

private:
  void BeforeLoad(int64_t now);
  void Upsert(std::vector<std::string>& values, const std::vector<size_t>& idx_map);
  AnyNum IngestTime(const Column& d, const std::string& value);
  Segment* LastSegment();

  std::string name_;
  size_t segment_size_;
  std::vector<std::unique_ptr<Column>> dims_, metrics_;
  std::vector<const Dimension*> dim_ptrs_;
  std::vector<const Metric*> metric_ptrs_;
  bool has_hidden_count_ = false;
  std::vector<CardinalityGuard> guards_;
  std::vector<std::unique_ptr<Segment>> segments_;
  std::map<std::string, size_t> tuple_offsets_;  // packed dimension tuple -> global row index
  std::vector<std::unique_ptr<util::TimeAny>> ingest_time_;         // per dimension (nullptr if not TIME)
  std::vector<std::vector<uint64_t>> ingest_rollup_;                // per dimension: rollup boundaries
};

std::vector<uint64_t> rollup_boundaries(const Column& dim, int64_t now);  // RollupReset

}  // namespace db
}  // namespace viya
