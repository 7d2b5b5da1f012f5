// vh_jit.hip — per-query compiled scan kernels (host side): shape -> HIP source -> hipRTC -> cached code object.
//
// What the reference does per query shape (src/query/runner.cc:45-64): AggQueryGenerator writes the C++ text of the whole
// aggregate function (src/codegen/query/agg_query.cc:26-75; the filter as ONE expression, ComparisonBuilder
// src/codegen/query/filter.cc:206-261), Compiler::Compile hashes it, looks for <state_dir>/codegen/<hash>.so and otherwise pipes
// the text through g++ (src/codegen/compiler.cc:36-144); literals are run-time arguments (`fargs`), so one compiled function
// serves every query of its shape. Same here: the text is a traits struct + the packed predicate loads + the filter expression
// per row slot + the survivor's gathers (vh_jit_source), the frame it includes is hand-written (vh_jit_body.h, vh_kernels.h —
// their text is embedded in this library at build time), hipRTC compiles it for gfx950, the code object is kept in memory and
// in a cache directory keyed by a hash of the text. The interpreting kernels of vh_kernels.h stay as the fallback.
#include "vh_jit.h"

#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "../build/vh_jit_embed.inc"     // vj_embedded[]: the text of viya_hip.h, vh_internal.h, vh_kernels.h, vh_jit_body.h (build.py)

// ------------------------------------------------------------------ policy
int vh_jit_policy() {
  static const int pol = [] {
    const char* e = getenv("VH_JIT");
    if (!e || !*e) return (int)VH_JIT_AUTO;
    if (!strcmp(e, "0") || !strcmp(e, "off")) return (int)VH_JIT_OFF;
    if (!strcmp(e, "force") || !strcmp(e, "2")) return (int)VH_JIT_FORCE;
    return (int)VH_JIT_AUTO;
  }();
  return pol;
}
uint64_t vh_jit_min_rows() {
  // A compile costs about a second; the interpreting kernels lose ~1 ms per 1 G rows and predicate to it. Below a few
  // million rows a query would have to repeat thousands of times to earn the compile back.
  static const uint64_t n = getenv("VH_JIT_MIN_ROWS") ? strtoull(getenv("VH_JIT_MIN_ROWS"), nullptr, 10) : (4ull << 20);
  return n;
}

// ------------------------------------------------------------------ shape key
std::string VhJitShape::key() const {
  std::string k;
  auto put = [&](long long v) { k += std::to_string(v); k += ','; };
  put(mode); put(block); put(scope); put(xcd); put(carrier); put(tw); put(key_words); put(lds_hash); put(gid32); put(ablate); put(hpart); put(hp_agg_waves); put(bitset_j); put(hp_pack); put(hp_pbits); put(hp_idbits); put(gid_bits); put(tuple4); put(bs_off32); put(hp_fan); put(part_ring);
  put(lanes); put(npred);
  for (int i = 0; i < npred; ++i) { put(pred[i].slot); put(pred[i].type); put(pred[i].width); }
  put(qpay); put(qpay_slot);
  put(pp_sliced); put(pp_slot);
  if (pp_sliced) for (int i = 0; i < npred; ++i) { put(pp_off[i]); put(pp_bits[i]); }
  put(pp_nplanes);
  for (int q = 0; q < pp_nplanes; ++q) { put(pp_plane[q].slot); put(pp_plane[q].width); put(pp_plane[q].pos); }
  if (pp_nplanes) for (int i = 0; i < npred; ++i) { put(pp_off[i]); put(pp_bits[i]); }
  put((long long)prog.size());
  for (const VhProgOp& o : prog) { put(o.w0); put(o.w1 & 0xFFFFFF00u); put(o.lit()); }    // (slot() of a leaf is implied by its pslot)
  put(nlits); put(ng); put(nm);
  auto col = [&](const VhJitCol& c) {
    put(c.slot); put(c.type); put(c.pitch); put(c.rec); put(c.off); put(c.sext); put(c.rowid); put(c.bitset); put(c.gran); put(c.nroll); put(c.micro);
    put(c.stored); put(c.bits); put(c.key_word); put(c.key_shift); for (int r = 0; r < c.nroll; ++r) put(c.roll_unit[r]); put(c.sop); put(c.tword); put(c.tshift); put(c.tbits);
  };
  for (int i = 0; i < ng; ++i) col(g[i]);
  for (int j = 0; j < nm; ++j) col(m[j]);
  return k;
}

// ------------------------------------------------------------------ source text
static const char* vj_ctype(int t) {
  switch (t) {
    case VH_U8: return "uint8_t"; case VH_U16: return "uint16_t"; case VH_U32: return "uint32_t"; case VH_U64: return "uint64_t";
    case VH_I8: return "int8_t"; case VH_I16: return "int16_t"; case VH_I32: return "int32_t"; case VH_I64: return "int64_t";
    case VH_F32: return "float"; default: return "double";
  }
}
static const char* vj_relop(int op) {
  switch (op) { case VH_OP_EQ: return "=="; case VH_OP_NE: return "!="; case VH_OP_LT: return "<"; case VH_OP_LE: return "<="; case VH_OP_GT: return ">"; default: return ">="; }
}
static std::string vj_fmt(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return buf;
}
static std::string vj_int_array(const char* name, int n, const std::vector<int>& v) {
  std::string s = vj_fmt("  static constexpr int %s[%d] = {", name, n > 0 ? n : 1);
  for (int i = 0; i < (n > 0 ? n : 1); ++i) { if (i) s += ", "; s += std::to_string(i < (int)v.size() ? v[i] : 0); }
  return s + "};\n";
}

std::string vh_jit_source(const VhJitShape& s, const char* kernel_name) {
  std::string t;
  t += "// generated by vh_jit.hip for one plan shape; the frame is vh_jit_body.h\n#define VH_HPART_KERNELS\n#include \"vh_jit_body.h\"\n#include \"vh_hpart.h\"\n";
  t += "typedef uint32_t vj_u32x2 __attribute__((ext_vector_type(2)));\n#define vj_b(x) __builtin_amdgcn_ballot_w64(x)\n";
  // ---- packed predicate registers: stream p (a predicate column, or a plane of the bit-packed predicate projection) occupies
  //      v[base[p] .. base[p] + 4 * width) of every lane
  struct Stream { int slot, width; };
  std::vector<Stream> streams;
  if (s.pp_nplanes) for (int q = 0; q < s.pp_nplanes; ++q) streams.push_back(Stream{s.pp_plane[q].slot, s.pp_plane[q].width});
  else for (int p = 0; p < s.npred; ++p) streams.push_back(Stream{s.pred[p].slot, s.pred[p].width});
  int sbase[VJ_MAX_PRED] = {}, snv = 0;                  // bit-sliced: predicate column p's planes occupy v[sbase[p] .. sbase[p] + pp_bits[p])
  if (s.pp_sliced) { streams.clear(); for (int p = 0; p < s.npred; ++p) { sbase[p] = snv; snv += s.pp_bits[p]; } }
  const int qpay_stream = s.qpay ? (int)streams.size() : -1;
  if (s.qpay) streams.push_back(Stream{s.qpay_slot, s.qpay});      // the payload records ride along with the predicate streams
  int base[VJ_MAX_PRED + 1] = {}, nv = 0;
  for (size_t p = 0; p < streams.size(); ++p) { base[p] = nv; nv += VH_SUBSTEPS * streams[p].width; }
  if (s.pp_sliced) nv = snv;
  const int nva = nv ? nv : 1;
  t += "struct VJ {\n";
  t += vj_fmt("  static constexpr int MODE = %d, BLOCK = %d, SCOPE = %d, NV = %d, NG = %d, NM = %d, TW = %d, KEY_WORDS = %d, CARRIER = %d;\n",
              s.mode, s.block, s.scope, nv, s.ng, s.nm, s.tw > 0 ? s.tw : 1, s.key_words, s.carrier);
  t += vj_fmt("  static constexpr int ABLATE = %d, BITSET_J = %d;\n  static constexpr bool HPART = %s;\n", s.ablate, s.bitset_j, s.hpart ? "true" : "false");
  t += vj_fmt("  static constexpr bool LDS_HASH = %s, XCD = %s, GID32 = %s;\n", s.lds_hash ? "true" : "false", s.xcd ? "true" : "false", s.gid32 ? "true" : "false");
  t += vj_fmt("  static constexpr bool HP_PACK = %s;\n  static constexpr int HP_PBITS = %d, HP_IDBITS = %d;\n", s.hp_pack ? "true" : "false", s.hp_pbits, s.hp_idbits);
  t += vj_fmt("  static constexpr int GID_BITS = %d;\n", s.gid_bits);
  t += vj_fmt("  static constexpr bool TUPLE4 = %s;\n", s.tuple4 ? "true" : "false");
  t += vj_fmt("  static constexpr bool BS_OFF32 = %s;\n", s.bs_off32 ? "true" : "false");
  t += vj_fmt("  static constexpr int PART_RING = %d;\n", s.part_ring);
  t += vj_fmt("  static constexpr bool LANES = %s;\n", s.lanes ? "true" : "false");
  t += vj_fmt("  static constexpr int QPAY = %d;\n", s.qpay);
  t += vj_fmt("  static constexpr bool SLICED = %s;\n", s.pp_sliced ? "true" : "false");
  {
    std::vector<int> a, b, c, d, e, f;
    for (int i = 0; i < s.ng; ++i) { a.push_back(s.g[i].type); b.push_back(s.g[i].gran); c.push_back(s.g[i].nroll); d.push_back(s.g[i].micro); e.push_back(s.g[i].key_word); f.push_back(s.g[i].key_shift); }
    t += vj_int_array("g_type", s.ng, a) + vj_int_array("g_gran", s.ng, b) + vj_int_array("g_nroll", s.ng, c) + vj_int_array("g_micro", s.ng, d) +
         vj_int_array("g_key_word", s.ng, e) + vj_int_array("g_key_shift", s.ng, f);
    t += vj_fmt("  static constexpr int g_roll_unit[%d][%d] = {", s.ng > 0 ? s.ng : 1, VH_MAX_ROLLUP);
    for (int i = 0; i < (s.ng > 0 ? s.ng : 1); ++i) {
      t += i ? ", {" : "{";
      for (int r = 0; r < VH_MAX_ROLLUP; ++r) { if (r) t += ", "; t += std::to_string(i < s.ng && r < s.g[i].nroll ? s.g[i].roll_unit[r] : 0); }
      t += "}";
    }
    t += "};\n";
    a.clear(); b.clear(); c.clear();
    d.clear();
    for (int j = 0; j < s.nm; ++j) { a.push_back(s.m[j].sop); b.push_back(s.m[j].tword); c.push_back(s.m[j].tshift); d.push_back(s.m[j].tbits); }
    t += vj_int_array("m_sop", s.nm, a) + vj_int_array("m_tword", s.nm, b) + vj_int_array("m_tshift", s.nm, c) + vj_int_array("m_tbits", s.nm, d);
    a.clear(); b.clear();         // element sizes of the OUTPUT columns (hp_aggregate_body writes them): a dimension's own type; a metric's, 8 bytes for a cardinality / row id
    for (int i = 0; i < s.ng; ++i) a.push_back(vh_elem_size(s.g[i].type));
    for (int j = 0; j < s.nm; ++j) b.push_back(vh_elem_size(s.m[j].type));
    t += vj_int_array("g_esize", s.ng, a) + vj_int_array("m_esize", s.nm, b);
  }
  // ---- the filter's literals, decoded once to the type of the column they are compared with (ValueDecoder, filter.cc:154-204)
  std::map<std::string, std::string> lit_decl;          // member name -> declaration + initialiser
  auto lit_name = [&](int idx, int type) {
    const std::string n = vj_fmt("l%d_%d", idx, type);
    if (!lit_decl.count(n)) lit_decl[n] = vj_fmt("%s %s", vj_ctype(type), n.c_str()) + "|" + vj_fmt("%s(vh_lit<%s>(P.ilits[%d]))", n.c_str(), vj_ctype(type), idx);
    return "L." + n;
  };
  // ---- the filter as one expression per row slot (ComparisonBuilder: composites are bitwise & / |, no short circuit), written
  //      twice: over the comparisons' wave ballots (the slot's pass mask, in scalar registers) and over the lane's own bools
  std::vector<std::string> sst;                             // bit-sliced: the lane's 32-row mask
  auto sliced_leaf = [&](const VhProgOp& o, int op, int lit_idx) {
    const int p = (int)o.pslot(), ty = (int)o.type();
    const std::string l = lit_name(lit_idx, ty);
    const bool sgn = ty == VH_I8 || ty == VH_I16 || ty == VH_I32 || ty == VH_I64;
    return vj_fmt("vj_bits_rel<%d, %d>(v + %d, (uint64_t)%s%s, %s)", s.pp_bits[p], op, sbase[p], sgn ? "(int64_t)" : "", l.c_str(), sgn ? ("(" + l + " < 0)").c_str() : "false");
  };
  if (s.pp_sliced)
    for (const VhProgOp& o : s.prog) {
      switch (o.kind()) {
        case VH_F_TRUE: sst.push_back("~0u"); break;
        case VH_F_AND: case VH_F_OR: {
          std::string e;
          const char* op = o.kind() == VH_F_AND ? " & " : " | ";
          for (int k = 0; k < (int)o.count() && !sst.empty(); ++k) { e = e.empty() ? sst.back() : sst.back() + op + e; sst.pop_back(); }
          sst.push_back("(" + e + ")");
        } break;
        case VH_F_REL: sst.push_back(sliced_leaf(o, (int)o.op(), (int)o.lit())); break;
        default: {         // IN: OR of ==, NOT IN: AND of != (filter.cc:223-241)
          std::string e = o.op() ? "0u" : "~0u";
          for (int k = 0; k < (int)o.count(); ++k) e += std::string(o.op() ? " | " : " & ") + sliced_leaf(o, o.op() ? VH_OP_EQ : VH_OP_NE, (int)o.lit() + k);
          sst.push_back("(" + e + ")");
        } break;
      }
    }
  std::vector<std::pair<std::string, std::string>> st;      // (mask expression, bool expression)
  for (const VhProgOp& o : s.prog) {
    switch (o.kind()) {
      case VH_F_TRUE: st.push_back({"vj_b(true)", "true"}); break;
      case VH_F_AND: case VH_F_OR: {
        std::string e, f;
        const int n = (int)o.count();
        const char* op = o.kind() == VH_F_AND ? " & " : " | ";
        for (int k = 0; k < n && !st.empty(); ++k) {
          e = e.empty() ? st.back().first : st.back().first + op + e;
          f = f.empty() ? st.back().second : st.back().second + op + f;
          st.pop_back();
        }
        st.push_back({"(" + e + ")", "(" + f + ")"});
      } break;
      default: {
        const int p = (int)o.pslot(), ty = (int)o.type();
        const std::string c = vj_fmt("c%d<I>(v)", p);
        if (o.kind() == VH_F_REL) {
          const std::string cmp = "(" + c + " " + vj_relop((int)o.op()) + " " + lit_name((int)o.lit(), ty) + ")";
          st.push_back({"vj_b" + cmp, cmp});
        } else {           // IN: OR of ==, NOT IN: AND of != (filter.cc:223-241)
          std::string e = o.op() ? "0ull" : "vj_b(true)", f = o.op() ? "false" : "true";
          for (int k = 0; k < (int)o.count(); ++k) {
            const std::string cmp = "(" + c + (o.op() ? " == " : " != ") + lit_name((int)o.lit() + k, ty) + ")";
            e += std::string(o.op() ? " | vj_b" : " & vj_b") + cmp;
            f += std::string(o.op() ? " | " : " & ") + cmp;
          }
          st.push_back({"(" + e + ")", "(" + f + ")"});
        }
      } break;
    }
  }
  const std::string filter_mask = st.empty() ? "vj_b(true)" : st.back().first, filter_bool = st.empty() ? "true" : st.back().second;
  t += "  struct Lits {\n";
  for (auto& kv : lit_decl) t += "    " + kv.second.substr(0, kv.second.find('|')) + ";\n";
  t += "    __device__ __forceinline__ Lits(const VhPlanDev& P)";
  { bool first = true; for (auto& kv : lit_decl) { t += first ? " : " : ", "; first = false; t += kv.second.substr(kv.second.find('|') + 1); } }
  t += " { (void)P; }\n  };\n";
  if (s.pp_sliced) {
    t += vj_fmt("  static __device__ __forceinline__ uint32_t mask(const Lits& L, const uint32_t (&v)[%d]) {\n    (void)L; (void)v;\n    return %s;\n  }\n", nva, sst.empty() ? "~0u" : sst.back().c_str());
    // a lane's words of the planes the filter reads: one 4-byte load per plane (256 contiguous bytes per wave and plane), non-temporal
    t += vj_fmt("  template <bool FULL> static __device__ __forceinline__ void preload(const VhPlanDev& P, uint32_t seg, uint32_t row_l, uint32_t seg_rows, uint32_t (&v)[%d]) {\n", nva);
    t += vj_fmt("    const char* base = P.colbase[%d] + (uint64_t)seg * P.colstride[%d] + (uint64_t)(row_l >> 5) * 4ull;\n    const uint64_t ps = P.colpitch[%d];\n    const bool in = FULL || row_l < seg_rows;\n    (void)base; (void)ps; (void)in;\n",
                s.pp_slot, s.pp_slot, s.pp_slot);
    for (int p = 0; p < s.npred; ++p)
      for (int b = 0; b < s.pp_bits[p]; ++b)
        t += vj_fmt("    v[%d] = in ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(base + %dull * ps)) : 0u;\n", sbase[p] + b, s.pp_off[p] + b);
    t += "  }\n";
  }
  // ---- accessors: the value of predicate column p in row slot I (I = 4 * sub-step + row of the lane's four), in the column's own type
  if (s.pp_nplanes && !s.pp_sliced) {
    // the row's word out of its planes (the compiler turns the byte / half-word picks into v_perm / SDWA selects), then each column as a bit field
    t += vj_fmt("  template <int I> static __device__ __forceinline__ uint32_t ppw(const uint32_t (&v)[%d]) {\n    return ", nva);
    for (int q = 0; q < s.pp_nplanes; ++q) {
      if (q) t += " | ";
      std::string e = s.pp_plane[q].width == 1 ? vj_fmt("((v[%d + (I >> 2)] >> (8 * (I & 3))) & 0xFFu)", base[q])
                                               : vj_fmt("((v[%d + (I >> 2) * 2 + ((I & 3) >> 1)] >> (16 * (I & 1))) & 0xFFFFu)", base[q]);
      t += s.pp_plane[q].pos ? "(" + e + vj_fmt(" << %d)", s.pp_plane[q].pos) : e;
    }
    t += ";\n  }\n";
    for (int p = 0; p < s.npred; ++p) {
      const char* T = vj_ctype(s.pred[p].type);
      t += vj_fmt("  template <int I> static __device__ __forceinline__ %s c%d(const uint32_t (&v)[%d]) { return (%s)((ppw<I>(v) >> %d) & 0x%Xu); }\n",
                  T, p, nva, T, s.pp_off[p], s.pp_bits[p] >= 32 ? 0xFFFFFFFFu : ((1u << s.pp_bits[p]) - 1u));
    }
  }
  for (int p = 0; p < s.npred && !s.pp_nplanes && !s.pp_sliced; ++p) {
    const VhJitPred& c = s.pred[p];
    const char* T = vj_ctype(c.type);
    t += vj_fmt("  template <int I> static __device__ __forceinline__ %s c%d(const uint32_t (&v)[%d]) {\n", T, p, nva);
    if (c.width == 1) t += vj_fmt("    return (%s)((v[%d + (I >> 2)] >> (8 * (I & 3))) & 0xFFu);\n", T, base[p]);
    else if (c.width == 2) t += vj_fmt("    return (%s)((v[%d + (I >> 2) * 2 + ((I & 3) >> 1)] >> (16 * (I & 1))) & 0xFFFFu);\n", T, base[p]);
    else if (c.width == 4) {
      if (c.type == VH_F32) t += vj_fmt("    return __uint_as_float(v[%d + I]);\n", base[p]);
      else t += vj_fmt("    return (%s)v[%d + I];\n", T, base[p]);
    } else {
      t += vj_fmt("    const uint64_t x = ((uint64_t)v[%d + 2 * I + 1] << 32) | v[%d + 2 * I];\n", base[p], base[p]);
      if (c.type == VH_F64) t += "    return __longlong_as_double((long long)x);\n";
      else t += vj_fmt("    return (%s)x;\n", T);
    }
    t += "  }\n";
  }
  if (!s.pp_sliced)
  t += vj_fmt("  template <int I> static __device__ __forceinline__ uint64_t pass(const Lits& L, const uint32_t (&v)[%d], bool& p) {\n    (void)L; (void)v;\n    p = ", nva) + filter_bool +
       ";\n    return " + filter_mask + ";\n  }\n";
  if (s.qpay)     // the record of row slot I, as it came in with the step's loads: what a passing row leaves in the wave's queue
    t += vj_fmt("  template <int I> static __device__ __forceinline__ uint32_t payload(const uint32_t (&v)[%d]) { return v[%d + I]; }\n", nva, base[qpay_stream]);
  // ---- the packed loads of one wave step: 4 consecutive rows per lane and sub-step, naturally aligned, non-temporal
  if (!s.pp_sliced)
  t += vj_fmt("  template <bool FULL> static __device__ __forceinline__ void preload(const VhPlanDev& P, uint32_t seg, uint32_t row_l, uint32_t seg_rows, uint32_t (&v)[%d]) {\n    (void)P; (void)seg; (void)row_l; (void)seg_rows; (void)v;\n", nva);
  for (size_t p = 0; p < streams.size() && !s.pp_sliced; ++p) {
    const Stream& c = streams[p];
    t += vj_fmt("    {\n      const char* col = P.colbase[%d] + (uint64_t)seg * P.colstride[%d];\n#pragma unroll\n      for (int k = 0; k < VH_SUBSTEPS; ++k) {\n        const uint32_t r = row_l + k * 256u;\n        if (FULL || r < seg_rows) {\n", c.slot, c.slot);
    const int b = base[p];
    if (c.width == 1) t += vj_fmt("          v[%d + k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(col + r));\n", b);
    else if (c.width == 2) t += vj_fmt("          const vj_u32x2 x = __builtin_nontemporal_load(reinterpret_cast<const vj_u32x2*>(col + 2ull * r));\n          v[%d + 2 * k] = x.x; v[%d + 2 * k + 1] = x.y;\n", b, b);
    else if (c.width == 4) t += vj_fmt("          const vh_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 4ull * r));\n          v[%d + 4 * k] = x.x; v[%d + 4 * k + 1] = x.y; v[%d + 4 * k + 2] = x.z; v[%d + 4 * k + 3] = x.w;\n", b, b, b, b);
    else t += vj_fmt("          const vh_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 8ull * r)), y = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 8ull * r + 16));\n"
                     "          v[%d + 8 * k] = x.x; v[%d + 8 * k + 1] = x.y; v[%d + 8 * k + 2] = x.z; v[%d + 8 * k + 3] = x.w; v[%d + 8 * k + 4] = y.x; v[%d + 8 * k + 5] = y.y; v[%d + 8 * k + 6] = y.z; v[%d + 8 * k + 7] = y.w;\n",
                     b, b, b, b, b, b, b, b);
    t += "        } else {\n";
    t += vj_fmt("#pragma unroll\n          for (int z = 0; z < %d; ++z) v[%d + %d * k + z] = 0u;\n", c.width, b, c.width);
    t += "        }\n      }\n    }\n";
  }
  if (!s.pp_sliced) t += "  }\n";
  // ---- a survivor's group and metric values: every load first (members of one payload projection share the record's
  //      address and are fetched with the widest aligned loads that cover them), then widths and extensions
  t += vj_fmt("  static __device__ __forceinline__ void gather(const VhPlanDev& P, uint32_t seg, uint32_t row, uint64_t (&gv)[%d], uint64_t (&mv)[%d]) {\n    (void)P; (void)seg; (void)row; (void)gv; (void)mv;\n",
              s.ng ? s.ng : 1, s.nm ? s.nm : 1);
  std::vector<const VhJitCol*> cols;
  for (int i = 0; i < s.ng; ++i) cols.push_back(&s.g[i]);
  for (int j = 0; j < s.nm; ++j) cols.push_back(&s.m[j]);
  std::map<std::pair<int, int>, std::string> dword;       // (projection, dword index inside the record) -> expression
  std::map<const VhJitCol*, std::string> raw;             // column -> expression of its raw bits (zero-extended)
  std::string loads;
  {
    std::map<int, std::vector<const VhJitCol*>> recs;
    for (const VhJitCol* c : cols) if (!c->rowid && !c->bitset && c->rec >= 0) recs[c->rec].push_back(c);
    int u = 0;
    for (auto& kv : recs) {
      const VhJitCol* b0 = kv.second[0];
      if (b0->bits) {       // bit-field record: ONE 4- or 8-byte word, every member a bit field of it (non-negative by construction: no sign to extend)
        loads += vj_fmt("    const char* r%d = P.colbase[%d] + (uint64_t)seg * P.colstride[%d] + (uint64_t)row * %uu;\n", kv.first, b0->slot, b0->slot, (unsigned)b0->pitch);
        if (b0->bits == 4) loads += vj_fmt("    const uint64_t w%d = (uint64_t)vj_gload(reinterpret_cast<const uint32_t*>(r%d));\n", kv.first, kv.first);
        else loads += vj_fmt("    const vj_u32x2 wv%d = vj_gload(reinterpret_cast<const vj_u32x2*>(r%d));\n    const uint64_t w%d = ((uint64_t)wv%d.y << 32) | wv%d.x;\n", kv.first, kv.first, kv.first, kv.first, kv.first);
        for (const VhJitCol* c : kv.second)
          raw[c] = c->stored >= 64 ? vj_fmt("w%d", kv.first) : vj_fmt("((w%d >> %d) & 0x%llXull)", kv.first, c->off, (1ull << c->stored) - 1ull);
        continue;
      }
      loads += vj_fmt("    const char* r%d = P.colbase[%d] - %d + (uint64_t)seg * P.colstride[%d] + (uint64_t)row * %uu;\n", kv.first, b0->slot, b0->off, b0->slot, (unsigned)b0->pitch);
      std::map<int, int> need;                             // 16-byte block -> mask of the dwords wanted
      for (const VhJitCol* c : kv.second) {
        const int es = c->stored ? c->stored : vh_elem_size(c->type);
        for (int by = c->off; by < c->off + es; by += 4) need[by / 16] |= 1 << ((by % 16) / 4);
      }
      for (auto& bm : need) {
        const int blk = bm.first, mask = bm.second;
        const char* comp[4] = {"x", "y", "z", "w"};
        if (__builtin_popcount(mask) == 1) {
          const int d = __builtin_ctz(mask);
          loads += vj_fmt("    const uint32_t u%d = vj_gload(reinterpret_cast<const uint32_t*>(r%d + %d));\n", u, kv.first, blk * 16 + d * 4);
          dword[{kv.first, blk * 4 + d}] = vj_fmt("u%d", u);
        } else if ((mask & 0xC) == 0 || (mask & 0x3) == 0) {
          const int h = (mask & 0x3) ? 0 : 1;
          loads += vj_fmt("    const vj_u32x2 u%d = vj_gload(reinterpret_cast<const vj_u32x2*>(r%d + %d));\n", u, kv.first, blk * 16 + h * 8);
          for (int d = 0; d < 2; ++d) dword[{kv.first, blk * 4 + h * 2 + d}] = vj_fmt("u%d.%s", u, comp[d]);
        } else {
          loads += vj_fmt("    const vh_u32x4 u%d = vj_gload(reinterpret_cast<const vh_u32x4*>(r%d + %d));\n", u, kv.first, blk * 16);
          for (int d = 0; d < 4; ++d) dword[{kv.first, blk * 4 + d}] = vj_fmt("u%d.%s", u, comp[d]);
        }
        ++u;
      }
      for (const VhJitCol* c : kv.second) {
        const int full = vh_elem_size(c->type), es = c->stored ? c->stored : full;
        const std::string lo = dword[{kv.first, c->off / 4}];
        if (es == 8) raw[c] = "(((uint64_t)" + dword[{kv.first, c->off / 4 + 1}] + " << 32) | " + lo + ")";
        else if (es == 4) raw[c] = "(uint64_t)" + lo;
        else raw[c] = vj_fmt("(uint64_t)((%s >> %d) & 0x%Xu)", lo.c_str(), 8 * (c->off % 4), es == 2 ? 0xFFFF : 0xFF);
        if (es < full && (c->type == VH_I16 || c->type == VH_I32 || c->type == VH_I64)) {
          // a compressed projection keeps the low bytes of a signed value: back to the column's own width, sign and all
          const char* st = es == 1 ? "int8_t" : es == 2 ? "int16_t" : "int32_t";
          const std::string wide = vj_fmt("(uint64_t)(int64_t)(%s)", st) + raw[c];
          raw[c] = full == 8 ? "(" + wide + ")" : vj_fmt("((%s) & 0x%llXull)", wide.c_str(), full == 4 ? 0xFFFFFFFFull : 0xFFFFull);
        }
      }
    }
    int a = 0;
    for (const VhJitCol* c : cols) {
      if (c->rowid || c->bitset || c->rec >= 0) continue;
      const int es = vh_elem_size(c->type);
      const char* lt = es == 8 ? "uint64_t" : es == 4 ? "uint32_t" : es == 2 ? "uint16_t" : "uint8_t";
      loads += vj_fmt("    const %s a%d = vj_gload(reinterpret_cast<const %s*>(P.colbase[%d] + (uint64_t)seg * P.colstride[%d] + (uint64_t)row * %uu));\n", lt, a, lt, c->slot, c->slot, (unsigned)c->pitch);
      raw[c] = vj_fmt("(uint64_t)a%d", a);
      ++a;
    }
  }
  t += loads;
  auto value = [&](const VhJitCol* c) -> std::string {
    if (c->rowid) return "(((uint64_t)seg << 32) | row)";
    if (c->bitset) return "0ull";                       // (its ids are read where they are used: vj_drain)
    const std::string& r = raw[c];
    if (!c->sext) return r;
    switch (c->type) {
      case VH_I8: return "(uint64_t)(int64_t)(int8_t)" + r;
      case VH_I16: return "(uint64_t)(int64_t)(int16_t)" + r;
      case VH_I32: return "(uint64_t)(int64_t)(int32_t)" + r;
      default: return r;
    }
  };
  for (int i = 0; i < s.ng; ++i) t += vj_fmt("    gv[%d] = %s;\n", i, value(&s.g[i]).c_str());
  for (int j = 0; j < s.nm; ++j) t += vj_fmt("    mv[%d] = %s;\n", j, value(&s.m[j]).c_str());
  t += "  }\n";
  if (s.qpay) {   // ... and the same values out of a QUEUED record (every column is a bit field of record 0's word)
    t += vj_fmt("  static __device__ __forceinline__ void unpack(uint32_t rec, uint64_t (&gv)[%d], uint64_t (&mv)[%d]) {\n    const uint64_t w0 = rec; const uint32_t seg = 0, row = 0; (void)seg; (void)row; (void)w0;\n",
                s.ng ? s.ng : 1, s.nm ? s.nm : 1);
    for (int i = 0; i < s.ng; ++i) t += vj_fmt("    gv[%d] = %s;\n", i, value(&s.g[i]).c_str());
    for (int j = 0; j < s.nm; ++j) t += vj_fmt("    mv[%d] = %s;\n", j, value(&s.m[j]).c_str());
    t += "  }\n";
  }
  if (s.lanes) {
    // ---- the no-compaction form (scan_agg_lanes_kernel's, compiled): a step's group and metric values for the lane's own 16 rows, every
    //      column with ONE aligned load per sub-step out of its arena (16 bytes for 4-byte columns, two for 8-byte ones, 8 / 4 bytes for 2- / 1-byte ones), all issued
    //      before the filter is looked at; row slot I = 4 * sub-step + row of the lane's four
    t += "  struct Payload {\n";
    for (size_t ci = 0; ci < cols.size(); ++ci) t += vj_fmt("    %s y%d[VH_LANE_ROWS];\n", vh_elem_size(cols[ci]->type) == 8 ? "uint64_t" : "uint32_t", (int)ci);
    t += "  };\n";
    t += "  template <bool FULL> static __device__ __forceinline__ void lanes_load(const VhPlanDev& P, uint32_t seg, uint32_t row_l, uint32_t seg_rows, Payload& y) {\n    (void)seg_rows;\n";
    for (size_t ci = 0; ci < cols.size(); ++ci) {
      const VhJitCol* c = cols[ci];
      const int es = vh_elem_size(c->type);
      t += vj_fmt("    {\n      const char* col = P.colbase[%d] + (uint64_t)seg * P.colstride[%d];\n#pragma unroll\n      for (int k = 0; k < VH_SUBSTEPS; ++k) {\n        const uint32_t r = row_l + k * 256u;\n        if (FULL || r + 4u <= seg_rows) {\n", c->slot, c->slot);
      if (es == 4) t += vj_fmt("          const vh_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 4ull * r));\n          y.y%d[4 * k] = x.x; y.y%d[4 * k + 1] = x.y; y.y%d[4 * k + 2] = x.z; y.y%d[4 * k + 3] = x.w;\n", (int)ci, (int)ci, (int)ci, (int)ci);
      else if (es == 2) t += vj_fmt("          const vj_u32x2 x = __builtin_nontemporal_load(reinterpret_cast<const vj_u32x2*>(col + 2ull * r));\n          y.y%d[4 * k] = x.x & 0xFFFFu; y.y%d[4 * k + 1] = x.x >> 16; y.y%d[4 * k + 2] = x.y & 0xFFFFu; y.y%d[4 * k + 3] = x.y >> 16;\n", (int)ci, (int)ci, (int)ci, (int)ci);
      else if (es == 1) t += vj_fmt("          const uint32_t x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(col + r));\n          y.y%d[4 * k] = x & 0xFFu; y.y%d[4 * k + 1] = (x >> 8) & 0xFFu; y.y%d[4 * k + 2] = (x >> 16) & 0xFFu; y.y%d[4 * k + 3] = x >> 24;\n", (int)ci, (int)ci, (int)ci, (int)ci);
      else t += vj_fmt("          const vh_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 8ull * r)), z = __builtin_nontemporal_load(reinterpret_cast<const vh_u32x4*>(col + 8ull * r + 16));\n"
                       "          y.y%d[4 * k] = ((uint64_t)x.y << 32) | x.x; y.y%d[4 * k + 1] = ((uint64_t)x.w << 32) | x.z; y.y%d[4 * k + 2] = ((uint64_t)z.y << 32) | z.x; y.y%d[4 * k + 3] = ((uint64_t)z.w << 32) | z.z;\n", (int)ci, (int)ci, (int)ci, (int)ci);
      t += "        } else {\n#pragma unroll\n          for (int z2 = 0; z2 < 4; ++z2)\n";
      t += vj_fmt("            y.y%d[4 * k + z2] = r + z2 < seg_rows ? reinterpret_cast<const %s*>(col)[r + z2] : 0;\n", (int)ci, es == 8 ? "uint64_t" : es == 4 ? "uint32_t" : es == 2 ? "uint16_t" : "uint8_t");
      t += "        }\n      }\n    }\n";
    }
    t += "  }\n";
    t += vj_fmt("  template <int I> static __device__ __forceinline__ void lanes_values(const Payload& y, uint64_t (&gv)[%d], uint64_t (&mv)[%d]) {\n    (void)y; (void)gv; (void)mv;\n", s.ng ? s.ng : 1, s.nm ? s.nm : 1);
    for (size_t ci = 0; ci < cols.size(); ++ci) raw[cols[ci]] = vj_fmt("(uint64_t)y.y%d[I]", (int)ci);
    for (int i = 0; i < s.ng; ++i) t += vj_fmt("    gv[%d] = %s;\n", i, value(&s.g[i]).c_str());
    for (int j = 0; j < s.nm; ++j) t += vj_fmt("    mv[%d] = %s;\n", j, value(&s.m[j]).c_str());
    t += "  }\n";
  }
  t += "};\n";
  t += vj_fmt("extern \"C\" __global__ __launch_bounds__(%d) void %s(const VhPlanDev P) { vj_scan<VJ>(P); }\n", s.block, kernel_name);
  // DENSE_PART: phase 2 — the ranges' aggregation in LDS — knows the same tuple layout (vj_part_agg)
  if (s.mode == VH_MODE_DENSE_PART)
    t += vj_fmt("extern \"C\" __global__ __launch_bounds__(1024) void %s_pagg(const VhPlanDev P, int blocks_per_part) { vj_part_agg<VJ, 1024>(P, blocks_per_part); }\n", kernel_name);
  // hashed partitioning: the kernel at the other end of the tuples — the ranges' aggregation in LDS (vh_hpart.h) — knows the same shape
  if (s.hpart)
  {
    if (s.hp_fan) t += "static_assert(VJ_FAN == HP_FAN && VJ_FAN_ET == HP_ET, \"the scan's level-A writer and vh_hpart.h agree on the pool's geometry\");\n";
    // (with a second argument the bound tells the compiler how many waves per SIMD the kernel is meant to run with — three 512-thread blocks
    // per CU are six: 80 registers. Without it C5's aggregation took 81, i.e. TWO blocks per CU where its 49 KB of LDS tables allow three)
    const std::string bound = s.hp_agg_waves > 0 ? vj_fmt("__launch_bounds__(%d, %d)", VH_HP_AGG_BLOCK, s.hp_agg_waves) : vj_fmt("__launch_bounds__(%d)", VH_HP_AGG_BLOCK);
    t += vj_fmt("extern \"C\" __global__ %s void %s_hpagg(const VhPlanDev P, const VhHpArgs* __restrict__ HA, int bpp, int a_first) { hp_aggregate_body<VJ, %d>(P, HA, bpp, a_first); }\n",
                bound.c_str(), kernel_name, VH_HP_AGG_BLOCK);
  }
  return t;
}

// ------------------------------------------------------------------ hipRTC (resolved at first use: the library loads where it is absent)
namespace {
struct Rtc {
  void* h = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  bool ok = false;
};
Rtc& rtc() {
  static Rtc r = [] {
    Rtc x;
    for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) { x.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (x.h) break; }
    if (!x.h) return x;
#define VJ_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.h, #name))
    VJ_SYM(create, hiprtcCreateProgram); VJ_SYM(compile, hiprtcCompileProgram); VJ_SYM(log_size, hiprtcGetProgramLogSize);
    VJ_SYM(log, hiprtcGetProgramLog); VJ_SYM(code_size, hiprtcGetCodeSize); VJ_SYM(code, hiprtcGetCode);
    VJ_SYM(destroy, hiprtcDestroyProgram); VJ_SYM(version, hiprtcVersion);
#undef VJ_SYM
    x.ok = x.create && x.compile && x.log_size && x.log && x.code_size && x.code && x.destroy;
    return x;
  }();
  return r;
}

uint64_t fnv64(const std::string& s, uint64_t h) {
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h;
}
std::string hash_hex(const std::string& text) {     // 128 bits: two FNV-1a streams with different offsets
  char b[40];
  snprintf(b, sizeof(b), "%016llx%016llx", (unsigned long long)fnv64(text, 1469598103934665603ull), (unsigned long long)fnv64(text, 0x9E3779B97F4A7C15ull));
  return b;
}
const std::string& frame_hash() {       // the embedded headers are part of every kernel: a change to them invalidates the cache
  static const std::string h = [] {
    std::string all;
    for (const auto& e : vj_embedded) { all += e.name; all.append(e.text, e.size); }
    int maj = 0, min = 0;
    if (rtc().ok && rtc().version) rtc().version(&maj, &min);
    all += "|hiprtc " + std::to_string(maj) + "." + std::to_string(min);
    return hash_hex(all);
  }();
  return h;
}
std::string arch_name() {
  static const std::string a = [] {
    if (const char* e = getenv("VH_JIT_ARCH")) return std::string(e);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.gcnArchName[0]) {
      std::string n = prop.gcnArchName;
      const size_t colon = n.find(':');
      return colon == std::string::npos ? n : n.substr(0, colon);
    }
    return std::string("gfx950");
  }();
  return a;
}
std::vector<std::string> cache_dirs() {      // [0] is written to; the rest are read-only (the in-tree directory build() warms up)
  std::vector<std::string> d;
  if (const char* e = getenv("VH_JIT_CACHE_DIR")) d.push_back(e);
  else if (const char* x = getenv("XDG_CACHE_HOME")) d.push_back(std::string(x) + "/viya_hip");
  else if (const char* h = getenv("HOME")) d.push_back(std::string(h) + "/.cache/viya_hip");
  else {
    // no home to keep a cache in: a directory under /tmp, used only if it is OURS and closed to others (a directory somebody else made
    // under that predictable name could hold planted code objects) — otherwise no disk cache at all (an empty entry: never read or written)
    const std::string p = "/tmp/viya_hip_jit_" + std::to_string((long)getuid());
    (void)mkdir(p.c_str(), 0700);
    struct stat st;
    const bool mine = lstat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & 077) == 0;
    d.push_back(mine ? p : std::string());
  }
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&vh_jit_policy), &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    d.push_back((slash == std::string::npos ? std::string(".") : p.substr(0, slash)) + "/jit_cache");
  }
  return d;
}
void mkdirs(const std::string& p) {
  for (size_t i = 1; i <= p.size(); ++i)
    if (i == p.size() || p[i] == '/') (void)mkdir(p.substr(0, i).c_str(), 0755);
}
bool read_file(const std::string& path, std::vector<char>* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(n > 0 ? (size_t)n : 0);
  const bool ok = n > 0 && fread(out->data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok;
}
void write_file_atomic(const std::string& dir, const std::string& name, const std::vector<char>& data) {
  mkdirs(dir);
  const std::string tmp = dir + "/." + name + "." + std::to_string((long)getpid()) + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
  fclose(f);
  if (!ok || rename(tmp.c_str(), (dir + "/" + name).c_str()) != 0) (void)unlink(tmp.c_str());      // (two processes compiling the same shape: the second rename wins, same bytes)
}

int rtc_compile(const std::string& src, std::vector<char>* code, std::string* log) {
  Rtc& R = rtc();
  if (!R.ok) { if (log) *log = "libhiprtc.so could not be loaded"; return -1; }
  std::vector<const char*> hsrc, hname;
  for (const auto& e : vj_embedded) { hsrc.push_back(e.text); hname.push_back(e.name); }
  hiprtcProgram prog = nullptr;
  if (R.create(&prog, src.c_str(), "viya_jit_scan.hip", (int)hsrc.size(), hsrc.data(), hname.data()) != HIPRTC_SUCCESS) { if (log) *log = "hiprtcCreateProgram failed"; return -1; }
  const std::string arch = "--offload-arch=" + arch_name();
  std::vector<const char*> opts = {arch.c_str(), "-O3", "-std=c++17", "-Wno-unused-result"};
  std::string extra = getenv("VH_JIT_FLAGS") ? getenv("VH_JIT_FLAGS") : "";      // experiments
  std::vector<std::string> extras;
  for (char& ch : extra) if (ch == ',') ch = ' ';
  for (size_t i = 0; i < extra.size();) { size_t j = extra.find(' ', i); if (j == std::string::npos) j = extra.size(); if (j > i) extras.push_back(extra.substr(i, j - i)); i = j + 1; }
  for (auto& e : extras) opts.push_back(e.c_str());
  const hiprtcResult rc = R.compile(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  if (log && R.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { log->resize(ls); R.log(prog, &(*log)[0]); }
  if (rc != HIPRTC_SUCCESS) { R.destroy(&prog); return -1; }
  size_t cs = 0;
  R.code_size(prog, &cs);
  code->resize(cs);
  R.code(prog, code->data());
  R.destroy(&prog);
  return cs ? 0 : -1;
}

struct Entry {
  std::mutex mu;
  bool done = false;
  std::unique_ptr<VhJitKernel> k;
  std::string err;
};
std::mutex g_jit_mu;
std::map<std::string, std::shared_ptr<Entry>> g_jit;
}  // namespace

static std::string kernel_name_of(const std::string& key) { return "viya_jit_scan_" + hash_hex(key).substr(0, 8); }

int vh_jit_compile_only(const VhJitShape& s, std::vector<char>* code, std::string* log) {
  const std::string name = kernel_name_of(s.key());
  return rtc_compile(vh_jit_source(s, name.c_str()), code, log);
}

VhJitKernel* vh_jit_get(const VhJitShape& s, std::string* err) {
  const std::string key = s.key();
  std::shared_ptr<Entry> e;
  {
    std::lock_guard<std::mutex> lk(g_jit_mu);
    auto& slot = g_jit[key];
    if (!slot) slot = std::make_shared<Entry>();
    e = slot;
  }
  std::lock_guard<std::mutex> lk(e->mu);          // a shape is compiled once; queries of the same shape wait for it, others do not
  if (e->done) { if (!e->k && err) *err = e->err; return e->k.get(); }
  e->done = true;
  const std::string name = kernel_name_of(key);
  const std::string src = vh_jit_source(s, name.c_str());
  const std::string file = hash_hex(src + "|" + frame_hash() + "|" + arch_name() + "|" + (getenv("VH_JIT_FLAGS") ? getenv("VH_JIT_FLAGS") : "")) + ".hsaco";
  std::vector<char> code;
  const std::vector<std::string> dirs = cache_dirs();
  bool cached = false;
  static const bool no_disk = getenv("VH_JIT_NO_DISK_CACHE") != nullptr;
  std::string cached_path;
  if (!no_disk) for (const std::string& d : dirs) if (!d.empty() && read_file(d + "/" + file, &code)) { cached = true; cached_path = d + "/" + file; break; }
  double ms = 0;
  std::unique_ptr<VhJitKernel> k(new VhJitKernel());
  if (cached) {
    // a cached code object that does not load (truncated, corrupt, built by another ROCm): forget the file and compile afresh, once —
    // a bad file must not switch the compiled kernel off for the life of the process
    hipError_t he = hipModuleLoadData(&k->mod, code.data());
    if (he == hipSuccess) he = hipModuleGetFunction(&k->fn, k->mod, name.c_str());
    if (he != hipSuccess) {
      if (getenv("VH_JIT_VERBOSE")) fprintf(stderr, "vh jit: cached %s does not load (%s): recompiling\n", cached_path.c_str(), hipGetErrorString(he));
      if (k->mod) (void)hipModuleUnload(k->mod);
      k.reset(new VhJitKernel());
      (void)unlink(cached_path.c_str());
      (void)hipGetLastError();
      cached = false;
      code.clear();
    }
  }
  if (!cached) {
    const auto t0 = std::chrono::steady_clock::now();
    std::string log;
    if (rtc_compile(src, &code, &log) != 0) {
      e->err = "hipRTC: " + log.substr(0, 1500);
      if (getenv("VH_JIT_VERBOSE")) fprintf(stderr, "vh jit: compile of %s failed:\n%s\n---- source\n%s\n", name.c_str(), log.c_str(), src.c_str());
      if (err) *err = e->err;
      return nullptr;
    }
    ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!no_disk && !dirs[0].empty()) write_file_atomic(dirs[0], file, code);
  }
  hipError_t he = hipSuccess;
  if (!k->fn) {
    he = hipModuleLoadData(&k->mod, code.data());
    if (he == hipSuccess) he = hipModuleGetFunction(&k->fn, k->mod, name.c_str());
  }
  if (he != hipSuccess) {
    e->err = std::string("loading the compiled kernel: ") + hipGetErrorString(he);
    if (err) *err = e->err;
    return nullptr;
  }
  if (s.hpart && hipModuleGetFunction(&k->fn_agg, k->mod, (name + "_hpagg").c_str()) != hipSuccess) {
    e->err = "the compiled module lacks its aggregation kernel";
    if (err) *err = e->err;
    return nullptr;
  }
  if (s.mode == VH_MODE_DENSE_PART && hipModuleGetFunction(&k->fn_pagg, k->mod, (name + "_pagg").c_str()) != hipSuccess) { (void)hipGetLastError(); k->fn_pagg = nullptr; }      // (the pre-built phase 2 answers then)
  k->name = name;
  k->compile_ms = ms;
  (void)hipFuncGetAttribute(&k->vgprs, HIP_FUNC_ATTRIBUTE_NUM_REGS, k->fn);
  if (getenv("VH_JIT_VERBOSE")) fprintf(stderr, "vh jit: %s %s (%.0f ms, %d VGPRs)\n", name.c_str(), cached ? "from the disk cache" : "compiled", ms, k->vgprs);
  e->k = std::move(k);
  return e->k.get();
}

int vh_jit_occupancy(VhJitKernel* k, int block, size_t lds) {
  int n = 0;
  if (lds > 64 * 1024 && lds > k->scan_lds_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); k->scan_lds_set = lds; }
  if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&n, k->fn, block, lds) != hipSuccess) return 0;
  return n;
}

hipError_t vh_jit_launch_hpagg(VhJitKernel* k, const VhPlanDev& P, const void* d_hpargs, int bpp, int a_first, int grid, size_t lds, hipStream_t s) {
  if (!k->fn_agg) return hipErrorInvalidDeviceFunction;
  if (lds > k->agg_lds_set) {        // (a module function may ask for more than 64 KB of dynamic LDS only after being told so)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn_agg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k->agg_lds_set = lds;
  }
  void* args[] = {const_cast<VhPlanDev*>(&P), &d_hpargs, &bpp, &a_first};
  return hipModuleLaunchKernel(k->fn_agg, (unsigned)grid, 1, 1, VH_HP_AGG_BLOCK, 1, 1, (unsigned)lds, s, args, nullptr);
}

hipError_t vh_jit_launch_pagg(VhJitKernel* k, const VhPlanDev& P, int blocks_per_part, size_t lds, hipStream_t s) {
  if (!k->fn_pagg) return hipErrorInvalidDeviceFunction;
  if (lds > k->pagg_lds_set) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn_pagg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k->pagg_lds_set = lds;
  }
  void* args[] = {const_cast<VhPlanDev*>(&P), &blocks_per_part};
  return hipModuleLaunchKernel(k->fn_pagg, (unsigned)(P.nfine * blocks_per_part), 1, 1, 1024, 1, 1, (unsigned)lds, s, args, nullptr);
}

hipError_t vh_jit_launch(VhJitKernel* k, const VhPlanDev& P, int grid, int block, size_t lds, hipStream_t s) {
  if (lds > k->scan_lds_set) {       // (the scan that partitions by itself keeps 64 KB of waiting lines next to its queues)
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k->scan_lds_set = lds;
  }
  void* args[] = {const_cast<VhPlanDev*>(&P)};
  return hipModuleLaunchKernel(k->fn, (unsigned)grid, 1, 1, (unsigned)block, 1, 1, (unsigned)lds, s, args, nullptr);
}

// ------------------------------------------------------------------ test hook (no GPU needed: hipRTC cross-compiles)
// Canonical plan shapes through the generator and hipRTC: tests/test_jit_compile.py checks here, on the CPU, that the text the
// generator writes compiles for gfx950, and looks at the code objects (registers, SDWA compares, no scratch).
static VhProgOp vj_leaf(int kind, int type, int op, int pslot, int lit, int count = 1) {
  VhProgOp o{};
  o.set_kind((uint8_t)kind); o.set_type((uint8_t)type); o.set_op((uint8_t)op); o.set_pslot((uint8_t)pslot); o.set_lit((uint16_t)lit); o.set_count((uint8_t)count);
  return o;
}
static VhProgOp vj_node(int kind, int count) { VhProgOp o{}; o.set_kind((uint8_t)kind); o.set_count((uint8_t)count); return o; }

static bool vj_canonical(int which, VhJitShape* s) {
  VhJitShape& S = *s;
  auto col = [](int slot, int type, int pitch, int rec, int off, int sext) { VhJitCol c; c.slot = slot; c.type = type; c.pitch = pitch; c.rec = rec; c.off = off; c.sext = sext; return c; };
  switch (which) {
    case 19:    // ... case 0 whose tuples leave through the block's ring writer (16 partitions' waiting lines per block, extents by position)
    case 20:    // ... case 9 with the tuple in FOUR bytes (gid 17 + 10 + 2 bits: thirty-two to a line)
    case 18:    // ... case 9 likewise (one-word tuples: sixteen to a line)
    case 14:    // ... case 10 with the predicate columns BIT-SLICED: 2 + 10 + 10 planes of one bit per row, a lane owns 32 consecutive rows per step
    case 13:    // ... case 12 with the payload records STREAMED beside the predicate planes and queued in the rows' place (no gathers)
    case 12:    // ... case 10 with the predicate columns out of a bit-packed predicate projection: d2 (2 bits) | d3 (10) | d4 (10) = a 2-byte and a 1-byte plane
    case 10:    // ... case 9 with the payload in a 4-byte BIT-FIELD record: d0 in 10 bits, d1 in 7, m0 in 10, count in 2
    case 9:     // ... case 7 with ONE-word tuples: gid in 17 bits, the SUM's value in 10, the COUNT's in 2
    case 7:     // ... case 0 with the payload from a compressed 8-byte record: m0 (i64) in 4 bytes, d0 in 2, d1 and count in 1 each
    case 0:     // C3: d2 == a & d3 < b & d4 >= c on narrow copies (1, 2, 2 bytes), payload from a 32-byte record, tuples for DENSE_PART
    case 1: {   // ... the same from the 4-byte arenas, straight into the dense HBM table (what an eighth of the table runs)
      const bool ring = which == 18 || which == 19;
      const bool four = which == 20;      // C3's one-word tuples in 4 bytes
      if (which == 18 || which == 20) which = 9;
      if (which == 19) which = 0;
      const bool part = which == 0 || which == 7 || which == 9 || which == 10 || which == 12 || which == 13 || which == 14;
      S.mode = part ? VH_MODE_DENSE_PART : VH_MODE_DENSE_GLOBAL; S.block = 256; S.scope = __HIP_MEMORY_SCOPE_AGENT; S.carrier = 1; S.tw = part ? 2 : 1; S.gid32 = 1; S.part_ring = part ? 16 : 0;
      S.npred = 3;
      S.pred[0] = VhJitPred{part ? 7 : 0, VH_U32, part ? 1 : 4}; S.pred[1] = VhJitPred{part ? 8 : 1, VH_U32, part ? 2 : 4}; S.pred[2] = VhJitPred{part ? 9 : 2, VH_U32, part ? 2 : 4};
      S.prog = {vj_leaf(VH_F_REL, VH_U32, VH_OP_EQ, 0, 0), vj_leaf(VH_F_REL, VH_U32, VH_OP_LT, 1, 1), vj_leaf(VH_F_REL, VH_U32, VH_OP_GE, 2, 2), vj_node(VH_F_AND, 3)};
      S.nlits = 3; S.ng = 2; S.nm = 2;
      (void)ring;
      if (part) {
        S.g[0] = col(10, VH_U32, 32, 0, 8, 1); S.g[1] = col(11, VH_U32, 32, 0, 12, 1);
        S.m[0] = col(12, VH_I64, 32, 0, 0, 0); S.m[0].sop = SOP_ADD64; S.m[0].tword = 1; S.m[0].tshift = 0;
        S.m[1] = col(13, VH_U32, 32, 0, 16, 0); S.m[1].sop = SOP_ADD32P; S.m[1].tword = 0; S.m[1].tshift = 32;
        if (which == 9 || which == 10 || which == 12 || which == 13 || which == 14) { S.tw = 1; S.gid_bits = 17; S.m[0].tword = 0; S.m[0].tshift = 17; S.m[0].tbits = 10; S.m[1].tword = 0; S.m[1].tshift = 27; S.m[1].tbits = 2; }
        if (four) S.tuple4 = 1;
        if (which == 13) { S.qpay = 4; S.qpay_slot = 10; }
        if (which == 14) { S.pp_sliced = 1; S.pp_slot = 7; S.pp_off[0] = 0; S.pp_bits[0] = 2; S.pp_off[1] = 2; S.pp_bits[1] = 10; S.pp_off[2] = 12; S.pp_bits[2] = 10; }
        if (which == 12 || which == 13) {
          S.pp_nplanes = 2;
          S.pp_plane[0] = {7, 2, 0}; S.pp_plane[1] = {8, 1, 16};
          S.pp_off[0] = 0; S.pp_bits[0] = 2; S.pp_off[1] = 2; S.pp_bits[1] = 10; S.pp_off[2] = 12; S.pp_bits[2] = 10;
          for (int k = 0; k < 3; ++k) { S.pred[k].slot = -1; S.pred[k].width = 0; }
        }
        if (which == 10 || which == 12 || which == 13 || which == 14) {
          for (VhJitCol* c : {&S.g[0], &S.g[1], &S.m[0], &S.m[1]}) { c->pitch = 4; c->bits = 4; }
          S.g[0].off = 0; S.g[0].stored = 10; S.g[1].off = 10; S.g[1].stored = 7; S.m[0].off = 17; S.m[0].stored = 10; S.m[1].off = 27; S.m[1].stored = 2;
        }
        if (which == 7 || which == 9) {
          S.m[0].pitch = S.m[1].pitch = S.g[0].pitch = S.g[1].pitch = 8;
          S.m[0].off = 0; S.m[0].stored = 4; S.g[0].off = 4; S.g[0].stored = 2; S.g[1].off = 6; S.g[1].stored = 1; S.m[1].off = 7; S.m[1].stored = 1;
        }
      } else {
        S.g[0] = col(3, VH_U32, 4, -1, 0, 1); S.g[1] = col(4, VH_U32, 4, -1, 0, 1);
        S.m[0] = col(5, VH_I64, 8, -1, 0, 0); S.m[0].sop = SOP_ADD64;
        S.m[1] = col(6, VH_U32, 4, -1, 0, 0); S.m[1].sop = SOP_ADD32P;
      }
      return true;
    }
    case 2: {   // C2-like: a range on an int32 column or a float, GROUP BY one u16 dictionary code, SUM(long) + MAX(double) in an LDS table
      S.mode = VH_MODE_DENSE_LDS; S.block = 512; S.scope = __HIP_MEMORY_SCOPE_WORKGROUP; S.xcd = 1; S.carrier = -1; S.gid32 = 1;
      S.npred = 2; S.pred[0] = VhJitPred{0, VH_I32, 4}; S.pred[1] = VhJitPred{1, VH_F32, 4};
      S.prog = {vj_leaf(VH_F_REL, VH_I32, VH_OP_GE, 0, 0), vj_leaf(VH_F_REL, VH_I32, VH_OP_LT, 0, 1), vj_node(VH_F_AND, 2), vj_leaf(VH_F_REL, VH_F32, VH_OP_GT, 1, 2), vj_node(VH_F_OR, 2)};
      S.nlits = 3; S.ng = 1; S.nm = 2;
      S.g[0] = col(2, VH_U16, 2, -1, 0, 1);
      S.m[0] = col(3, VH_I64, 8, -1, 0, 0); S.m[0].sop = SOP_ADD64;
      S.m[1] = col(4, VH_F64, 8, -1, 0, 0); S.m[1].sop = SOP_MAX_F64;
      return true;
    }
    case 3: {   // C5t-like: GROUP BY (time rolled up + hour granularity, u32) -> hash with the LDS front table, COUNT; filter: u8 IN (...) & i64 != x
      S.mode = VH_MODE_HASH; S.block = 512; S.scope = __HIP_MEMORY_SCOPE_AGENT; S.carrier = -1; S.key_words = 1; S.lds_hash = 1;
      S.npred = 2; S.pred[0] = VhJitPred{0, VH_U8, 1}; S.pred[1] = VhJitPred{1, VH_I64, 8};
      S.prog = {vj_leaf(VH_F_IN, VH_U8, 1, 0, 0, 3), vj_leaf(VH_F_REL, VH_I64, VH_OP_NE, 1, 3), vj_node(VH_F_AND, 2)};
      S.nlits = 4; S.ng = 2; S.nm = 1;
      S.g[0] = col(2, VH_U32, 4, -1, 0, 0); S.g[0].gran = VH_T_HOUR; S.g[0].nroll = 3; S.g[0].roll_unit[0] = VH_T_MONTH; S.g[0].roll_unit[1] = VH_T_DAY; S.g[0].roll_unit[2] = VH_T_HOUR; S.g[0].key_word = 0; S.g[0].key_shift = 0;
      S.g[1] = col(3, VH_U32, 4, -1, 0, 0); S.g[1].key_word = 0; S.g[1].key_shift = 32;
      S.m[0] = col(4, VH_U32, 4, -1, 0, 0); S.m[0].sop = SOP_ADD32;
      return true;
    }
    case 4: {   // wide hash key (double, i16, u64), MIN over the row id + MIN(i8) + SUM(float), NOT IN on a u16 column, no LDS front table
      S.mode = VH_MODE_HASH; S.block = 256; S.scope = __HIP_MEMORY_SCOPE_AGENT; S.carrier = -1; S.key_words = 3; S.lds_hash = 0;
      S.npred = 2; S.pred[0] = VhJitPred{0, VH_U16, 2}; S.pred[1] = VhJitPred{1, VH_F64, 8};
      S.prog = {vj_leaf(VH_F_IN, VH_U16, 0, 0, 0, 2), vj_leaf(VH_F_REL, VH_F64, VH_OP_LE, 1, 2), vj_node(VH_F_OR, 2)};
      S.nlits = 3; S.ng = 3; S.nm = 3;
      S.g[0] = col(1, VH_F64, 8, -1, 0, 0); S.g[0].key_word = 0;
      S.g[1] = col(2, VH_I16, 2, -1, 0, 0); S.g[1].key_word = 1;
      S.g[2] = col(3, VH_U64, 8, -1, 0, 0); S.g[2].key_word = 2;
      S.m[0].rowid = 1; S.m[0].type = VH_U64; S.m[0].sop = SOP_MIN_U64;
      S.m[1] = col(4, VH_I8, 1, -1, 0, 1); S.m[1].sop = SOP_MIN_I32;
      S.m[2] = col(5, VH_F32, 4, -1, 0, 0); S.m[2].sop = SOP_ADDF32;
      return true;
    }
    case 11: {  // C2 itself in the no-compaction form: a range on one u32 column, GROUP BY a u32 code, SUM(long) + SUM(int) in an LDS table, 256-thread blocks
      S.mode = VH_MODE_DENSE_LDS; S.block = 256; S.scope = __HIP_MEMORY_SCOPE_WORKGROUP; S.xcd = 1; S.carrier = -1; S.gid32 = 1; S.lanes = 1;
      S.npred = 1; S.pred[0] = VhJitPred{0, VH_U32, 4};
      S.prog = {vj_leaf(VH_F_REL, VH_U32, VH_OP_GE, 0, 0), vj_leaf(VH_F_REL, VH_U32, VH_OP_LT, 0, 1), vj_node(VH_F_AND, 2)};
      S.nlits = 2; S.ng = 1; S.nm = 2;
      S.g[0] = col(1, VH_U32, 4, -1, 0, 1);
      S.m[0] = col(2, VH_I64, 8, -1, 0, 0); S.m[0].sop = SOP_ADD64;
      S.m[1] = col(3, VH_I32, 4, -1, 0, 1); S.m[1].sop = SOP_ADD32;
      return true;
    }
    case 5: {   // no filter, no group columns (SELECT SUM(x)): the smallest text the generator writes
      S.mode = VH_MODE_DENSE_LDS; S.block = 1024; S.scope = __HIP_MEMORY_SCOPE_AGENT; S.carrier = -1; S.gid32 = 1;
      S.prog = {vj_node(VH_F_TRUE, 0)};
      S.ng = 0; S.nm = 1;
      S.m[0] = col(0, VH_I64, 8, -1, 0, 0); S.m[0].sop = SOP_ADD64;
      return true;
    }
    case 16:    // ... case 8 whose scan writes the level-A pool itself (vj_fan_add): 1024-thread blocks, the digits' waiting lines in LDS
    case 15:    // ... case 6 likewise (32-byte tuples: four to a line)
    case 8:     // ... case 6 with PACKED 16-byte tuples: COUNT in 2 bits, two ids of 24 bits, their count — all in the tuple's second word
    case 6: {   // C5: GROUP BY (time rolled up + hour granularity, u32), COUNT DISTINCT of a bitset metric + COUNT, hashed partitioning (32-byte tuples that carry the ids)
      S.mode = VH_MODE_HASH; S.block = 256; S.scope = __HIP_MEMORY_SCOPE_AGENT; S.carrier = -1; S.key_words = 1; S.lds_hash = 0; S.hpart = 1; S.bitset_j = 0; S.tw = 2;
      S.npred = 1; S.pred[0] = VhJitPred{1, VH_U32, 4};
      S.prog = {vj_leaf(VH_F_REL, VH_U32, VH_OP_LT, 0, 0)};
      S.nlits = 1; S.ng = 2; S.nm = 2;
      S.g[0] = col(0, VH_U32, 4, -1, 0, 0); S.g[0].gran = VH_T_HOUR; S.g[0].nroll = 3; S.g[0].roll_unit[0] = VH_T_MONTH; S.g[0].roll_unit[1] = VH_T_DAY; S.g[0].roll_unit[2] = VH_T_HOUR; S.g[0].key_word = 0; S.g[0].key_shift = 0;
      S.g[1] = col(1, VH_U32, 4, -1, 0, 0); S.g[1].key_word = 0; S.g[1].key_shift = 32;
      S.m[0].bitset = 1; S.m[0].type = VH_U64; S.m[0].sop = SOP_BITSET;
      S.m[1] = col(2, VH_U32, 4, -1, 0, 0); S.m[1].sop = SOP_ADD32; S.m[1].tword = 1; S.m[1].tshift = 0;
      if (which == 8 || which == 16) { S.hp_pack = 1; S.hp_pbits = 2; S.hp_idbits = 24; S.m[1].tbits = 2; }
      if (which >= 15) { S.hp_fan = 1; S.block = 1024; }
      return true;
    }
    default: return false;
  }
}

extern "C" VH_API int vh_jit_selftest(int32_t which, const char* hsaco_path, char* text, uint64_t text_bytes) {
  VhJitShape S;
  if (!vj_canonical(which, &S)) return VH_E_INVALID;
  const std::string src = vh_jit_source(S, "viya_jit_scan_selftest");
  std::vector<char> code;
  std::string log;
  const int rc = rtc_compile(src, &code, &log);
  if (text && text_bytes) snprintf(text, (size_t)text_bytes, "%s%s%s", rc ? log.c_str() : "", rc ? "\n---- source\n" : "", src.c_str());
  if (rc) return VH_E_UNSUPPORTED;
  if (hsaco_path && *hsaco_path) {
    FILE* f = fopen(hsaco_path, "wb");
    if (!f) return VH_E_INVALID;
    fwrite(code.data(), 1, code.size(), f);
    fclose(f);
  }
  return VH_OK;
}
