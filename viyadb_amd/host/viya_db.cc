// viya_db.cc — see viya_db.h for the reference map.
#include "viya_db.h"

#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <sstream>

namespace viya {
namespace util {

static const char* kUnitNames[7] = {"year", "month", "week", "day", "hour", "minute", "second"};

TimeUnit time_unit_by_name(const std::string& name) {
  for (int i = 0; i < 7; ++i)
    if (name == kUnitNames[i]) return static_cast<TimeUnit>(i);
  throw std::invalid_argument("Unsupported time unit: " + name);
}

Duration::Duration(const std::string& desc) {
  std::istringstream ss(desc);
  std::string unit;
  int n = 0;
  if (!(ss >> n) || !(ss >> unit) || n <= 0) throw std::invalid_argument("Wrong duration description: " + desc);
  count_ = (size_t)n;
  unit.pop_back();  // plural 's'
  time_unit_ = time_unit_by_name(unit);
}

uint32_t Duration::add_to(uint32_t timestamp, int sign) const {
  time_t t = (time_t)timestamp;
  std::tm tm;
  gmtime_r(&t, &tm);
  const int n = sign * (int)count_;
  switch (time_unit_) {
    case YEAR: tm.tm_year += n; break;
    case MONTH: tm.tm_mon += n; break;
    case WEEK: tm.tm_mday += 7 * n; break;
    case DAY: tm.tm_mday += n; break;
    case HOUR: tm.tm_hour += n; break;
    case MINUTE: tm.tm_min += n; break;
    case SECOND: tm.tm_sec += n; break;
    default: throw std::runtime_error("Unsupported duration");
  }
  return (uint32_t)timegm(&tm);
}

uint64_t Duration::add_to(uint64_t timestamp, int sign) const {
  return (uint64_t)add_to((uint32_t)(timestamp / 1000000L), sign) * 1000000L;
}

void trunc_tm(std::tm& tm, TimeUnit unit) {
  switch (unit) {
    case YEAR: tm.tm_mon = 0;  // fallthrough
    case MONTH: tm.tm_mday = 1;
    case DAY: tm.tm_hour = 0;
    case HOUR: tm.tm_min = 0;
    case MINUTE: tm.tm_sec = 0;
    case SECOND: break;
    default: throw std::runtime_error("week granularity: the reference has no Truncator::trunc<WEEK>");
  }
}

void TimeAny::parse(const char* format, const std::string& value) {
  strptime(value.c_str(), format, &tm_);
  micros_ = 0;
}
void TimeAny::set_ts(uint64_t ts) {
  time_t t;
  if (micro_) { micros_ = (uint32_t)(ts % 1000000L); t = (time_t)(ts / 1000000L); }
  else t = (time_t)(uint32_t)ts;
  gmtime_r(&t, &tm_);
}
uint64_t TimeAny::get_ts() {
  if (micro_) return (uint64_t)timegm(&tm_) * 1000000L + micros_;
  return (uint32_t)timegm(&tm_);
}
void TimeAny::trunc(TimeUnit unit) {
  trunc_tm(tm_, unit);
  if (micro_) micros_ = 0;
}

}  // namespace util

namespace db {

// ------------------------------------------------------------------ numeric types
NumericType::NumericType(const std::string& n) {
  static const std::pair<const char*, Num> names[] = {{"byte", Num::BYTE}, {"ubyte", Num::UBYTE}, {"short", Num::SHORT},
      {"ushort", Num::USHORT}, {"int", Num::INT}, {"uint", Num::UINT}, {"long", Num::LONG}, {"ulong", Num::ULONG},
      {"float", Num::FLOAT}, {"double", Num::DOUBLE}};
  for (auto& p : names)
    if (n == p.first) { t_ = p.second; return; }
  throw std::invalid_argument("Unsupported metric type: " + n);
}

int NumericType::size() const {
  switch (t_) {
    case Num::BYTE: case Num::UBYTE: return 1;
    case Num::SHORT: case Num::USHORT: return 2;
    case Num::INT: case Num::UINT: case Num::FLOAT: return 4;
    default: return 8;
  }
}

std::string NumericType::cpp_type() const {
  static const char* n[] = {"int8_t", "uint8_t", "int16_t", "uint16_t", "int32_t", "uint32_t", "int64_t", "uint64_t", "float", "double"};
  return n[(int)t_];
}

int NumericType::vh_elem() const {
  // enum vh_elem: U8 U16 U32 U64 I8 I16 I32 I64 F32 F64
  static const int e[] = {4, 0, 5, 1, 6, 2, 7, 3, 8, 9};
  return e[(int)t_];
}

AnyNum NumericType::cpp_min_value() const {
  switch (t_) {
    case Num::BYTE: return AnyNum::of<int8_t>(INT8_MIN);
    case Num::SHORT: return AnyNum::of<int16_t>(INT16_MIN);
    case Num::INT: return AnyNum::of<int32_t>(INT32_MIN);
    case Num::LONG: return AnyNum::of<int64_t>(INT64_MIN);
    case Num::FLOAT: return AnyNum::of<float>(FLT_MIN);    // smallest positive: a reference quirk
    case Num::DOUBLE: return AnyNum::of<double>(DBL_MIN);
    default: return AnyNum();
  }
}
AnyNum NumericType::cpp_max_value() const {
  switch (t_) {
    case Num::BYTE: return AnyNum::of<int8_t>(INT8_MAX);
    case Num::UBYTE: return AnyNum::of<uint8_t>(UINT8_MAX);
    case Num::SHORT: return AnyNum::of<int16_t>(INT16_MAX);
    case Num::USHORT: return AnyNum::of<uint16_t>(UINT16_MAX);
    case Num::INT: return AnyNum::of<int32_t>(INT32_MAX);
    case Num::UINT: return AnyNum::of<uint32_t>(UINT32_MAX);
    case Num::LONG: return AnyNum::of<int64_t>(INT64_MAX);
    case Num::ULONG: return AnyNum::of<uint64_t>(UINT64_MAX);
    case Num::FLOAT: return AnyNum::of<float>(FLT_MAX);
    default: return AnyNum::of<double>(DBL_MAX);
  }
}

AnyNum NumericType::Parse(const std::string& v) const {
  switch (t_) {
    case Num::BYTE: return AnyNum::of<int8_t>((int8_t)std::stoi(v));
    case Num::UBYTE: return AnyNum::of<uint8_t>((uint8_t)std::stoul(v));
    case Num::SHORT: return AnyNum::of<int16_t>((int16_t)std::stoi(v));
    case Num::USHORT: return AnyNum::of<uint16_t>((uint16_t)std::stoul(v));
    case Num::INT: return AnyNum::of<int32_t>((int32_t)std::stoi(v));
    case Num::UINT: return AnyNum::of<uint32_t>((uint32_t)std::stoul(v));
    case Num::LONG: return AnyNum::of<int64_t>((int64_t)std::stoll(v));
    case Num::ULONG: return AnyNum::of<uint64_t>((uint64_t)std::stoull(v));
    case Num::FLOAT: return AnyNum::of<float>(std::stof(v));
    default: return AnyNum::of<double>(std::stod(v));
  }
}

NumericType uint_type_for_max(uint64_t max_value) {
  if (max_value - 1 < UINT8_MAX) return NumericType(Num::UBYTE);
  if (max_value - 1 < UINT16_MAX) return NumericType(Num::USHORT);
  if (max_value - 1 < UINT32_MAX) return NumericType(Num::UINT);
  return NumericType(Num::ULONG);
}

double load_as_double(const char* p, Num t) {
  switch (t) {
    case Num::BYTE: return *reinterpret_cast<const int8_t*>(p);
    case Num::UBYTE: return *reinterpret_cast<const uint8_t*>(p);
    case Num::SHORT: { int16_t v; memcpy(&v, p, 2); return v; }
    case Num::USHORT: { uint16_t v; memcpy(&v, p, 2); return v; }
    case Num::INT: { int32_t v; memcpy(&v, p, 4); return v; }
    case Num::UINT: { uint32_t v; memcpy(&v, p, 4); return v; }
    case Num::LONG: { int64_t v; memcpy(&v, p, 8); return (double)v; }
    case Num::ULONG: { uint64_t v; memcpy(&v, p, 8); return (double)v; }
    case Num::FLOAT: { float v; memcpy(&v, p, 4); return v; }
    default: { double v; memcpy(&v, p, 8); return v; }
  }
}

// util::Format::num (src/util/format.h:32-65): integers decimal; double "%.15g"; float through
// fmt 4.x's default formatting, which is "%g".
std::string format_num(const char* p, Num t) {
  char buf[64];
  switch (t) {
    case Num::BYTE: return std::to_string((int32_t) * reinterpret_cast<const int8_t*>(p));
    case Num::UBYTE: return std::to_string((uint32_t) * reinterpret_cast<const uint8_t*>(p));
    case Num::SHORT: { int16_t v; memcpy(&v, p, 2); return std::to_string((int32_t)v); }
    case Num::USHORT: { uint16_t v; memcpy(&v, p, 2); return std::to_string((uint32_t)v); }
    case Num::INT: { int32_t v; memcpy(&v, p, 4); return std::to_string(v); }
    case Num::UINT: { uint32_t v; memcpy(&v, p, 4); return std::to_string(v); }
    case Num::LONG: { int64_t v; memcpy(&v, p, 8); return std::to_string(v); }
    case Num::ULONG: { uint64_t v; memcpy(&v, p, 8); return std::to_string(v); }
    case Num::FLOAT: { float v; memcpy(&v, p, 4); snprintf(buf, sizeof(buf), "%g", (double)v); return buf; }
    default: { double v; memcpy(&v, p, 8); snprintf(buf, sizeof(buf), "%.15g", v); return buf; }
  }
}

int compare_typed(Num t, AnyNum a, AnyNum b) {
#define VH_CMP(T) { T x = a.as<T>(), y = b.as<T>(); return x < y ? -1 : (x > y ? 1 : (x == y ? 0 : 2)); }
  switch (t) {
    case Num::BYTE: VH_CMP(int8_t)
    case Num::UBYTE: VH_CMP(uint8_t)
    case Num::SHORT: VH_CMP(int16_t)
    case Num::USHORT: VH_CMP(uint16_t)
    case Num::INT: VH_CMP(int32_t)
    case Num::UINT: VH_CMP(uint32_t)
    case Num::LONG: VH_CMP(int64_t)
    case Num::ULONG: VH_CMP(uint64_t)
    case Num::FLOAT: VH_CMP(float)
    default: VH_CMP(double)
  }
#undef VH_CMP
}

// ------------------------------------------------------------------ columns
AnyNum DimensionDict::Decode(const std::string& value) const {
  auto it = v2c_.find(value);
  AnyNum a;
  a.bits = it != v2c_.end() ? it->second : (size_ == 8 ? UINT64_MAX : ((1ull << (size_ * 8)) - 1));
  return a;
}

Column::SortType Column::sort_type() const {
  if (type_ == DIMENSION) {
    if (dim_type_ == DIM_STRING || dim_type_ == DIM_TIME) return STRING;
    if (dim_type_ == DIM_BOOLEAN) return INTEGER;
    return num_type_.fp() ? FLOAT : INTEGER;
  }
  if (agg_type_ == BITSET) return INTEGER;
  return num_type_.fp() ? FLOAT : INTEGER;
}

DimensionDict* Dictionaries::GetOrCreate(const std::string& dim_name, int code_size) {
  auto it = dicts_.find(dim_name);
  if (it == dicts_.end()) it = dicts_.emplace(dim_name, std::make_unique<DimensionDict>(code_size)).first;
  return it->second.get();
}

static void check_legal_string(const std::string& what, const std::string& s) {
  // util::check_legal_string (src/util/sanitize.cc): names end up inside generated code there
  for (char c : s)
    if (c == '"' || c == '\\' || c == '\n' || c == '\r' || c == '\0') throw std::invalid_argument(what + " contains illegal characters");
}

static Column::AggregationType parse_agg_type(const std::string& type) {
  auto pos = type.find('_');
  if (pos != std::string::npos) {
    std::string a = type.substr(pos + 1);
    if (a == "sum") return Column::SUM;
    if (a == "max") return Column::MAX;
    if (a == "min") return Column::MIN;
    if (a == "avg") return Column::AVG;
  }
  if (type == "count") return Column::COUNT;
  if (type == "bitset") return Column::BITSET;
  throw std::invalid_argument("Unsupported metric type: " + type);
}

// ------------------------------------------------------------------ table
Table::Table(const util::Config& config, Dictionaries& dicts)
    : segment_size_((size_t)config.num("segment_size", 1000000L)) {
  name_ = config.str("name");
  check_legal_string("Table name", name_);
  size_t di = 0;
  for (const util::Config& dc : config.sublist("dimensions")) {
    const std::string type = dc.str("type", "string");
    const std::string name = dc.str("name");
    check_legal_string("Column name", name);
    std::unique_ptr<Column> c;
    if (type == "string") {
      const uint64_t card = dc.unum("cardinality", UINT32_MAX);
      c.reset(new Column(Column::DIMENSION, di, name, uint_type_for_max(card)));
      c->dim_type_ = Column::DIM_STRING;
      c->cardinality_ = card;
      c->length_ = (int)dc.num("length", -1);
      c->dict_ = dicts.GetOrCreate(name, c->num_type().size());
    } else if (type == "boolean") {
      c.reset(new Column(Column::DIMENSION, di, name, NumericType(Num::UBYTE)));
      c->dim_type_ = Column::DIM_BOOLEAN;
    } else if (type == "time" || type == "microtime") {
      const bool micro = type == "microtime";
      c.reset(new Column(Column::DIMENSION, di, name, NumericType(micro ? Num::ULONG : Num::UINT)));
      c->dim_type_ = Column::DIM_TIME;
      c->micro_ = micro;
      if (dc.exists("format")) { c->format_ = dc.str("format"); check_legal_string("Time format", c->format_); }
      if (dc.exists("granularity")) {
        c->granularity_ = util::time_unit_by_name(dc.str("granularity"));
      } else if (dc.exists("rollup_rules")) {
        for (const util::Config& rc : dc.sublist("rollup_rules"))
          c->rollup_rules_.push_back(RollupRule{util::time_unit_by_name(rc.str("granularity")), util::Duration(rc.str("after"))});
        std::stable_sort(c->rollup_rules_.begin(), c->rollup_rules_.end(),
                         [](const RollupRule& a, const RollupRule& b) { return a.after > b.after; });
      }
    } else {
      NumericType nt(Num::UINT);
      if (type == "numeric") {  // deprecated form: width from "max"
        nt = uint_type_for_max(dc.unum("max", UINT32_MAX)).size() == 8 ? NumericType(Num::ULONG) : NumericType(Num::UINT);
      } else {
        nt = NumericType(type);
      }
      c.reset(new Column(Column::DIMENSION, di, name, nt));
      c->dim_type_ = Column::DIM_NUMERIC;
    }
    if (dc.exists("field")) c->input_field_ = dc.str("field");
    c->storage_index = di;
    dim_ptrs_.push_back(c.get());
    dims_.push_back(std::move(c));
    ++di;
  }
  size_t mi = 0;
  bool has_avg = false, has_count = false;
  for (const util::Config& mc : config.sublist("metrics")) {
    const std::string type = mc.str("type");
    const std::string name = mc.str("name");
    check_legal_string("Column name", name);
    const Column::AggregationType agg = parse_agg_type(type);
    NumericType nt(Num::UINT);
    if (agg == Column::BITSET) nt = uint_type_for_max(mc.unum("max", UINT32_MAX));
    else if (agg == Column::COUNT) nt = uint_type_for_max(mc.unum("max", UINT32_MAX)).size() == 8 ? NumericType(Num::ULONG) : NumericType(Num::UINT);
    else nt = NumericType(type.substr(0, type.find('_')));
    std::unique_ptr<Column> c(new Column(Column::METRIC, mi, name, nt));
    c->agg_type_ = agg;
    if (mc.exists("field")) c->input_field_ = mc.str("field");
    c->storage_index = dims_.size() + mi;
    has_avg |= agg == Column::AVG;
    has_count |= agg == Column::COUNT;
    metric_ptrs_.push_back(c.get());
    metrics_.push_back(std::move(c));
    ++mi;
  }
  has_hidden_count_ = has_avg && !has_count;
  for (const util::Config& dc : config.sublist("dimensions")) {
    if (!dc.exists("cardinality_guard")) continue;
    const Column* d = dimension(dc.str("name"));
    if (d->dim_type() == Column::DIM_NUMERIC) throw std::invalid_argument("Can't define cardinality guard on a numeric dimension");
    util::Config gc = dc.sub("cardinality_guard");
    CardinalityGuard g{d, {}, (size_t)gc.num("limit"), {}};
    for (auto& n : gc.strlist("dimensions")) g.dimensions.push_back(dimension(n));
    guards_.push_back(std::move(g));
  }
  ingest_time_.resize(dims_.size());
  ingest_rollup_.resize(dims_.size());
  for (auto& d : dims_)
    if (d->dim_type() == Column::DIM_TIME) ingest_time_[d->index()].reset(new util::TimeAny(d->micro_precision()));
}

Table::~Table() {
  if (gpu_mirror && gpu_mirror_free) gpu_mirror_free(gpu_mirror);
}

const Column* Table::column(const std::string& name) const {
  for (auto& d : dims_) if (d->name() == name) return d.get();
  for (auto& m : metrics_) if (m->name() == name) return m.get();
  throw std::invalid_argument("No such column: " + name);
}
const Dimension* Table::dimension(const std::string& name) const {
  for (auto& d : dims_) if (d->name() == name) return d.get();
  throw std::invalid_argument("No such dimension: " + name);
}
const Metric* Table::metric(const std::string& name) const {
  for (auto& m : metrics_) if (m->name() == name) return m.get();
  throw std::invalid_argument("No such metric: " + name);
}
std::vector<const Column*> Table::columns() const {
  std::vector<const Column*> out(dim_ptrs_.begin(), dim_ptrs_.end());
  out.insert(out.end(), metric_ptrs_.begin(), metric_ptrs_.end());
  return out;
}
int Table::storage_elem_size(size_t si) const {
  if (si < dims_.size()) return dims_[si]->num_type().size();
  if (si < dims_.size() + metrics_.size()) {
    const Column& m = *metrics_[si - dims_.size()];
    return m.agg_type() == Column::BITSET ? 0 : m.num_type().size();
  }
  return 8;
}

Segment::Segment(const Table& t) : capacity_(t.segment_size()) {
  const size_t n = t.storage_columns();
  cols_.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const int es = t.storage_elem_size(i);
    if (!es) continue;
    cols_[i].reset(new char[capacity_ * es + 64]());
  }
  // Metrics(): MIN columns start at the type's max, MAX columns at cpp_min_value (store.cc:262-276)
  for (auto* m : t.metrics()) {
    if (m->agg_type() == Column::BITSET) { bitsets_[m->index()].resize(capacity_); continue; }
    if (m->agg_type() != Column::MIN && m->agg_type() != Column::MAX) continue;
    const AnyNum ident = m->agg_type() == Column::MIN ? m->num_type().cpp_max_value() : m->num_type().cpp_min_value();
    const int es = m->num_type().size();
    char* p = cols_[m->storage_index].get();
    for (size_t r = 0; r < capacity_; ++r) memcpy(p + r * es, &ident.bits, es);
  }
  stats.resize(t.dimensions().size());
  for (auto* d : t.dimensions()) {
    if (d->dim_type() != Column::DIM_NUMERIC && d->dim_type() != Column::DIM_TIME) continue;
    stats[d->index()].tracked = true;
    stats[d->index()].dmax = d->num_type().cpp_min_value();
    stats[d->index()].dmin = d->num_type().cpp_max_value();
  }
}

Segment* Table::LastSegment() {
  if (segments_.empty() || segments_.back()->full()) segments_.emplace_back(new Segment(*this));
  return segments_.back().get();
}

std::vector<uint64_t> rollup_boundaries(const Column& d, int64_t now) {
  std::vector<uint64_t> out;
  if (now < 0) now = (int64_t)std::time(nullptr);
  for (auto& r : d.rollup_rules()) {
    uint64_t b = r.after.add_to((uint32_t)now, -1);
    if (d.micro_precision()) b *= 1000000L;
    out.push_back(b);
  }
  return out;
}

void Table::BeforeLoad(int64_t now) {
  for (auto& d : dims_)
    if (d->dim_type() == Column::DIM_TIME) ingest_rollup_[d->index()] = rollup_boundaries(*d, now);
}

void Table::Load(const std::vector<std::vector<std::string>>& rows, const std::vector<std::string>* columns, int64_t now) {
  std::lock_guard<std::mutex> lk(mu);
  BeforeLoad(now);
  // LoaderDesc::InitTupleIdxMap (src/input/loader_desc.cc:54-96)
  std::vector<const Column*> input_cols(dim_ptrs_.begin(), dim_ptrs_.end());
  for (auto* m : metric_ptrs_)
    if (m->agg_type() != Column::COUNT) input_cols.push_back(m);
  std::vector<size_t> idx_map(input_cols.size());
  if (columns) {
    for (size_t i = 0; i < input_cols.size(); ++i) {
      const std::string& nm = input_cols[i]->input_field().empty() ? input_cols[i]->name() : input_cols[i]->input_field();
      auto it = std::find(columns->begin(), columns->end(), nm);
      if (it == columns->end()) throw std::runtime_error("Column name '" + nm + "' is not specified in load spec");
      idx_map[i] = (size_t)(it - columns->begin());
    }
  } else {
    for (auto* c : this->columns())
      if (!c->input_field().empty())
        throw std::runtime_error("Column names must be specified, because one or more columns define field name mapping");
    for (size_t i = 0; i < idx_map.size(); ++i) idx_map[i] = i;
  }
  for (auto& r : rows) {
    std::vector<std::string> values = r;
    if (values.size() < input_cols.size() && !columns) throw std::runtime_error("input row has too few fields");
    Upsert(values, idx_map);
  }
}

AnyNum Table::IngestTime(const Column& d, const std::string& value) {
  // ValueParser::Visit(const db::TimeDimension*) (src/codegen/db/upsert.cc:82-139)
  util::TimeAny& t = *ingest_time_[d.index()];
  const std::string& fmt = d.format();
  const bool posix = fmt == "posix", millis = fmt == "millis", micros = fmt == "micros";
  const bool is_num = fmt.empty() || posix || millis || micros;
  const bool has_rollup = !d.rollup_rules().empty();
  const bool has_gran = d.granularity() != util::_UNDEFINED;
  uint64_t tup = 0;
  auto narrow = [&](uint64_t v) { return d.micro_precision() ? v : (uint64_t)(uint32_t)v; };
  if (is_num) {
    uint64_t ts = std::stoull(value);
    if (d.micro_precision()) { if (posix) ts *= 1000000L; else if (millis) ts *= 1000L; }
    else { if (millis) ts /= 1000L; else if (micros) ts /= 1000000L; }
    tup = narrow(ts);
    if (has_rollup || has_gran) t.set_ts(tup);
  } else {
    t.parse(fmt.c_str(), value);
    if (has_rollup) tup = t.get_ts();
  }
  if (has_rollup) {
    const auto& bounds = ingest_rollup_[d.index()];
    for (size_t i = 0; i < d.rollup_rules().size(); ++i)
      if (tup < bounds[i]) { t.trunc(d.rollup_rules()[i].granularity); break; }
  } else if (has_gran) {
    t.trunc(d.granularity());
  }
  if (!is_num || has_rollup || has_gran) tup = t.get_ts();
  AnyNum a;
  a.bits = narrow(tup);
  return a;
}

void Table::Upsert(std::vector<std::string>& values, const std::vector<size_t>& idx_map) {
  size_t vi = 0;
  std::vector<AnyNum> dvals(dims_.size());
  for (auto& dp : dims_) {
    const Column& d = *dp;
    std::string& v = values[idx_map[vi]];
    switch (d.dim_type()) {
      case Column::DIM_STRING: {
        if (d.length() != -1 && (int)v.length() > d.length()) v.erase(d.length());
        DimensionDict* dict = d.dict();
        auto it = dict->v2c().find(v);
        uint64_t code;
        if (it != dict->v2c().end()) {
          code = it->second;
        } else {
          code = dict->c2v().size();
          const bool check = d.cardinality() < UINT64_MAX - 1;
          if (!check || code <= d.cardinality()) {
            dict->v2c().emplace(v, code);
            dict->c2v().emplace_back(v);
          } else {
            code = 0;  // "__exceeded"
          }
        }
        AnyNum a;
        a.bits = d.num_type().size() == 8 ? code : (code & ((1ull << (d.num_type().size() * 8)) - 1));
        dvals[d.index()] = a;
      } break;
      case Column::DIM_NUMERIC: dvals[d.index()] = d.num_type().Parse(v); break;
      case Column::DIM_BOOLEAN: dvals[d.index()] = AnyNum::of<uint8_t>(v == "true"); break;
      case Column::DIM_TIME: dvals[d.index()] = IngestTime(d, v); break;
    }
    ++vi;
  }
  std::vector<AnyNum> mvals(metrics_.size());
  for (auto& mp : metrics_) {
    const Column& m = *mp;
    if (m.agg_type() == Column::COUNT) { mvals[m.index()].bits = 1; continue; }
    mvals[m.index()] = m.num_type().Parse(values[idx_map[vi++]]);
  }
  // CardinalityProtection (upsert.cc:266-300)
  for (auto& g : guards_) {
    std::vector<uint64_t> key;
    for (auto* p : g.dimensions) key.push_back(dvals[p->index()].bits);
    const uint64_t code = dvals[g.dim->index()].bits;
    auto it = g.stats.find(key);
    if (it == g.stats.end()) {
      g.stats[key].insert(code);
    } else if (it->second.size() >= g.limit) {
      if (!it->second.count(code)) dvals[g.dim->index()].bits = 0;
    } else {
      it->second.insert(code);
    }
  }
  std::string key(dims_.size() * 8, '\0');
  for (auto& dp : dims_) {
    uint64_t b = dvals[dp->index()].bits;
    if (dp->num_type().type() == Num::FLOAT && (uint32_t)b == 0x80000000u) b = 0;          // -0.0f == 0.0f
    if (dp->num_type().type() == Num::DOUBLE && b == 0x8000000000000000ull) b = 0;
    memcpy(&key[dp->index() * 8], &b, 8);
  }
  auto off = tuple_offsets_.find(key);
  if (off != tuple_offsets_.end()) {
    Segment& seg = *segments_[off->second / segment_size_];
    const size_t ti = off->second % segment_size_;
    for (auto& mp : metrics_) {  // Segment::Metrics::Update (store.cc:311-340): in the metric's own type
      const Column& m = *mp;
      if (m.agg_type() == Column::BITSET) {
        auto& set = seg.bitsets(m.index())[ti];
        const uint64_t id = mvals[m.index()].bits;
        auto pos = std::lower_bound(set.begin(), set.end(), id);
        if (pos == set.end() || *pos != id) set.insert(pos, id);
        continue;
      }
      const int es = m.num_type().size();
      char* p = seg.column(m.storage_index) + ti * es;
      AnyNum cur;
      memcpy(&cur.bits, p, es);
      const AnyNum nv = mvals[m.index()];
      AnyNum out = cur;
      const Num t = m.num_type().type();
      if (m.agg_type() == Column::SUM || m.agg_type() == Column::AVG || m.agg_type() == Column::COUNT) {
        if (t == Num::FLOAT) out = AnyNum::of<float>(cur.as<float>() + nv.as<float>());
        else if (t == Num::DOUBLE) out = AnyNum::of<double>(cur.as<double>() + nv.as<double>());
        else out.bits = cur.bits + nv.bits;  // modular; truncated to the type's width on store
      } else {
        const int c = compare_typed(t, nv, cur);
        if (m.agg_type() == Column::MAX ? c == 1 : c == -1) out = nv;
      }
      memcpy(p, &out.bits, es);
    }
    if (has_hidden_count_) {
      uint64_t* p = reinterpret_cast<uint64_t*>(seg.column(hidden_count_storage_index())) + ti;
      *p += 1;
    }
    seg.touch(ti);
  } else {
    Segment& seg = *LastSegment();
    const size_t ti = seg.size_;
    for (auto& dp : dims_) memcpy(seg.column(dp->storage_index) + ti * dp->num_type().size(), &dvals[dp->index()].bits, dp->num_type().size());
    for (auto& mp : metrics_) {
      if (mp->agg_type() == Column::BITSET) { seg.bitsets(mp->index())[ti] = {mvals[mp->index()].bits}; continue; }
      memcpy(seg.column(mp->storage_index) + ti * mp->num_type().size(), &mvals[mp->index()].bits, mp->num_type().size());
    }
    if (has_hidden_count_) reinterpret_cast<uint64_t*>(seg.column(hidden_count_storage_index()))[ti] = 1;
    ++seg.size_;
    for (auto& dp : dims_) {  // SegmentStats::Update (store.cc:187-199)
      SegmentStat& s = seg.stats[dp->index()];
      if (!s.tracked) continue;
      const Num t = dp->num_type().type();
      if (compare_typed(t, dvals[dp->index()], s.dmax) == 1) s.dmax = dvals[dp->index()];
      if (compare_typed(t, dvals[dp->index()], s.dmin) == -1) s.dmin = dvals[dp->index()];
    }
    seg.touch(ti);
    tuple_offsets_.emplace(key, (segments_.size() - 1) * segment_size_ + ti);
  }
}

}  // namespace db
}  // namespace viya
