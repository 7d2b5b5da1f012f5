// viya_query.cc — filter factory, aggregate-query descriptor parsing, literal decoding, Database.
#include "viya_query.h"
#include "partial_state.h"

#include <algorithm>
#include <cctype>
#include <cstring>

namespace viya {
namespace query {

Filter Filter::Compare(Relation r, std::string column, std::string literal) {
  Filter f;
  f.kind_ = COMPARE; f.relation_ = r; f.column_ = std::move(column); f.literals_.push_back(std::move(literal));
  return f;
}
Filter Filter::Member(std::string column, std::vector<std::string> literals, bool inside) {
  Filter f;
  f.kind_ = MEMBER; f.inside_ = inside; f.column_ = std::move(column); f.literals_ = std::move(literals);
  return f;
}
Filter Filter::Combine(bool all, std::vector<Filter> parts) {
  Filter f;
  f.kind_ = all ? ALL_OF : ANY_OF; f.parts_ = std::move(parts);
  return f;
}

Filter Filter::FromConfig(const util::Config& config, bool negate) {
  if (!config.exists("op")) return Filter();
  const std::string op = config.str("op");
  if (op == "not") return FromConfig(config.sub("filter"), !negate);
  if (op == "and" || op == "or") {
    std::vector<Filter> parts;
    for (const util::Config& fc : config.sublist("filters")) parts.push_back(FromConfig(fc, negate));
    std::stable_sort(parts.begin(), parts.end(), [](const Filter& a, const Filter& b) { return a.rank() < b.rank(); });
    return Combine((op == "and") != negate, std::move(parts));
  }
  std::string column = config.str("column");
  if (op == "in") return Member(std::move(column), config.strlist("values"), !negate);
  static const struct { const char* name; Relation plain, negated; } table[] = {
      {"eq", EQUAL, NOT_EQUAL}, {"ne", NOT_EQUAL, EQUAL}, {"lt", LESS, GREATER_EQUAL},
      {"le", LESS_EQUAL, GREATER}, {"gt", GREATER, LESS_EQUAL}, {"ge", GREATER_EQUAL, LESS}};
  for (const auto& e : table)
    if (op == e.name) return Compare(negate ? e.negated : e.plain, std::move(column), config.str("value"));
  throw std::invalid_argument("Unsupported filter operataor: " + op);   // (the reference's spelling: its tests match on the text)
}

namespace {

// ValueDecoder (src/codegen/query/filter.cc:154-204)
db::AnyNum DecodeValue(const db::Column* col, const std::string& value) {
  using db::Column;
  if (col->type() == Column::DIMENSION) {
    switch (col->dim_type()) {
      case Column::DIM_STRING: return col->dict()->Decode(value);
      case Column::DIM_BOOLEAN: return db::AnyNum::of<uint8_t>(value == "true");
      case Column::DIM_TIME: {
        if (std::all_of(value.begin(), value.end(), ::isdigit)) return col->num_type().Parse(value);
        const uint64_t mult = col->micro_precision() ? 1000000L : 1L;
        uint64_t ts = 0;
        std::tm tm;
        memset(&tm, 0, sizeof(tm));
        const char* r = strptime(value.c_str(), "%Y-%m-%d %T", &tm);
        if (r != nullptr && *r == '\0') {
          ts = timegm(&tm) * mult;
        } else if (r != nullptr && col->micro_precision() && *r == '.') {
          ts = timegm(&tm) * mult + std::stoul(r);  // stoul(".xyz") throws, exactly as in the reference
        } else {
          memset(&tm, 0, sizeof(tm));
          r = strptime(value.c_str(), "%Y-%m-%d", &tm);
          if (r != nullptr && *r == '\0') ts = timegm(&tm) * mult;
        }
        if (ts == 0) throw std::invalid_argument("Unrecognized time format: " + value);
        db::AnyNum a;
        a.bits = col->micro_precision() ? ts : (uint64_t)(uint32_t)ts;
        return a;
      }
      default: break;
    }
  }
  const db::Num t = col->num_type().type();
  if (t == db::Num::BYTE || t == db::Num::SHORT)
    // the reference's generated code calls AnyNum::get_int8_t()/get_int16_t(), which do not exist
    // (src/db/column.h:110-117 vs src/codegen/query/filter.cc:126-132): such a query fails to compile
    throw std::runtime_error("filters on byte/short columns do not compile in the reference");
  return col->num_type().Parse(value);
}

}  // namespace

// FilterArgsPacker (src/codegen/query/filter.cc:100-152): every leaf's literals, decoded to the column's type, in evaluation order
std::vector<db::AnyNum> PackFilterArgs(const db::Table& table, const Filter* filter) {
  std::vector<db::AnyNum> args;
  if (filter)
    filter->EachLeaf([&](const Filter& leaf) {
      const db::Column* c = table.column(leaf.column());
      for (const std::string& text : leaf.literals()) args.push_back(DecodeValue(c, text));
    });
  return args;
}

DimOutputColumn::DimOutputColumn(const util::Config& config, const db::Dimension* dim, size_t index) : index_(index), dim_(dim) {
  if (dim->dim_type() == db::Column::DIM_TIME) {
    format_ = config.str("format", dim->format());
    if (config.exists("granularity")) granularity_ = util::time_unit_by_name(config.str("granularity"));
  }
}

AggregateQuery::AggregateQuery(const util::Config& config, db::Table& table, bool select_only)
    : table_(table), header_(config.boolean("header", false)), skip_((size_t)config.num("skip", 0)), limit_((size_t)config.num("limit", 0)) {
  filter_ = std::make_unique<Filter>(Filter::FromConfig(config.sub("filter", true)));
  size_t out_idx = 0;
  if (config.exists("select")) {
    for (const util::Config& sc : config.sublist("select")) {
      const std::string name = sc.str("column");
      std::vector<const db::Column*> cols;
      if (name == "*") cols = table.columns();
      else cols.push_back(table.column(name));
      for (auto* c : cols) {
        if (c->type() == db::Column::DIMENSION) dimension_cols_.emplace_back(sc, c, out_idx++);
        else metric_cols_.emplace_back(c, out_idx++);
      }
    }
  } else {
    for (auto& n : config.strlist("dimensions")) dimension_cols_.emplace_back(table.dimension(n), out_idx++);
    for (auto& n : config.strlist("metrics")) metric_cols_.emplace_back(table.metric(n), out_idx++);
  }
  if (select_only) return;
  if (config.exists("sort")) {
    for (const util::Config& sc : config.sublist("sort")) {
      const std::string name = sc.str("column");
      const db::Column* col = table.column(name);
      int idx = -1;
      for (auto& d : dimension_cols_) if (d.dim() == col) { idx = (int)d.index(); break; }
      if (idx == -1) for (auto& m : metric_cols_) if (m.metric() == col) { idx = (int)m.index(); break; }
      if (idx == -1) throw std::invalid_argument("Sort column '" + name + "' is not selected");
      sort_cols_.emplace_back(col, (size_t)idx, sc.boolean("ascending", false));
    }
  }
  if (config.exists("having")) {
    having_ = std::make_unique<Filter>(Filter::FromConfig(config.sub("having")));
    const auto names = column_names();
    having_->EachLeaf([&](const Filter& leaf) {
      if (std::find(names.begin(), names.end(), leaf.column()) == names.end())
        throw std::invalid_argument("Column '" + leaf.column() + " is not selected");
    });
  }
}

SearchQuery::SearchQuery(const util::Config& config, db::Table& table)
    : table_(table), header_(config.boolean("header", false)), dimension_(table.dimension(config.str("dimension"))),
      term_(config.str("term")), limit_((size_t)config.num("limit", 0)) {
  filter_ = std::make_unique<Filter>(Filter::FromConfig(config.sub("filter", true)));
}

std::vector<std::string> AggregateQuery::column_names() const {
  std::vector<std::string> out;
  for (auto& d : dimension_cols_) out.push_back(d.dim()->name());
  for (auto& m : metric_cols_) out.push_back(m.metric()->name());
  return out;
}

}  // namespace query

namespace db {

Database::Database(const util::Config& config, int device) {
  (void)device;
  if (config.exists("tables"))
    for (const util::Config& tc : config.sublist("tables")) CreateTable(tc);
}
Database::~Database() {}

void Database::CreateTable(const util::Config& tc) {
  const std::string name = tc.str("name");
  if (tables_.count(name)) throw std::runtime_error("Table already exists: " + name);
  tables_[name] = std::make_unique<Table>(tc, dicts_);
}

Table* Database::GetTable(const std::string& name) {
  auto it = tables_.find(name);
  if (it == tables_.end()) throw std::invalid_argument("No such table: " + name);
  return it->second.get();
}

void Database::Load(const std::string& table, const std::vector<std::vector<std::string>>& rows, int64_t now) {
  GetTable(table)->Load(rows, nullptr, now);
}

// Database::Query -> QueryFactory::Create -> QueryRunner::Visit(AggregateQuery*)
query::QueryStats Database::Query(const util::Config& conf, query::RowOutput& output, int64_t now) {
  const std::string type = conf.str("type");
  if (type != "aggregate" && type != "select" && type != "search")
    throw std::invalid_argument("unsupported query type: " + type + " (this build accelerates aggregate, select and search queries)");
  Table* table = GetTable(conf.str("table"));
  auto t0 = std::chrono::steady_clock::now();
  if (type == "select") {      // QueryRunner::Visit(SelectQuery*)  (src/query/runner.cc:29-43)
    query::SelectQuery q(conf, *table);
    query::QueryStats stats;
    std::vector<AnyNum> fargs = query::PackFilterArgs(*table, q.filter());
    stats.compile_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    query::GpuSelect(q, output, stats, fargs, q.skip(), q.limit());
    stats.whole_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return stats;
  }
  if (type == "search") {      // QueryRunner::Visit(SearchQuery*)  (src/query/runner.cc:66-80)
    query::SearchQuery q(conf, *table);
    query::QueryStats stats;
    std::vector<AnyNum> fargs = query::PackFilterArgs(*table, q.filter());
    stats.compile_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    query::GpuSearch(q, output, stats, fargs, q.term(), q.limit());
    stats.whole_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return stats;
  }
  query::AggregateQuery q(conf, *table);
  query::QueryStats stats;
  std::vector<AnyNum> fargs = query::PackFilterArgs(*table, q.filter());
  std::vector<AnyNum> hargs = query::PackFilterArgs(*table, q.having());
  stats.compile_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  query::GpuAggregate(q, output, stats, fargs, q.skip(), q.limit(), hargs, now, comm_);
  stats.whole_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return stats;
}

std::string Database::QueryPartial(const util::Config& conf, query::QueryStats& stats, int64_t now) {
  if (conf.str("type") != "aggregate") throw std::invalid_argument("partial states exist for aggregate queries only");
  Table* table = GetTable(conf.str("table"));
  auto t0 = std::chrono::steady_clock::now();
  query::AggregateQuery q(conf, *table);
  std::vector<AnyNum> fargs = query::PackFilterArgs(*table, q.filter());
  stats.compile_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::string blob = cluster::query::AggregatePartial(q, stats, fargs, now);
  stats.whole_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return blob;
}

query::QueryStats Database::QueryMerge(const util::Config& conf, const std::vector<std::string>& partials, query::RowOutput& output) {
  if (conf.str("type") != "aggregate") throw std::invalid_argument("partial states exist for aggregate queries only");
  Table* table = GetTable(conf.str("table"));
  auto t0 = std::chrono::steady_clock::now();
  query::AggregateQuery q(conf, *table);
  query::QueryStats stats;
  cluster::query::MergePartials(q, partials, output, stats);
  stats.whole_time = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return stats;
}

}  // namespace db
}  // namespace viya
