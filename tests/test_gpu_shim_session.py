"""include/viya_shim.h at work: a C++ driver does exactly what the GENERATED viya_query_agg does (tools/gen_shim_tu.py) —
column addresses of "its" segments, the size() snapshot, dictionaries bound by pointer, literals as 8-byte AnyNum words,
rows through a Send callback — against plain arrays standing for the reference's Segment objects, including the two kinds
of change upsert makes between queries (rows appended; metrics of an existing row updated in place). The rows must be the
ones the host shim's own Database returns for the same data, i.e. the reference's formatting and order."""
import os
import subprocess
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r'''
    #include <viya_shim.h>
    #include <cstdio>
    #include <cstring>
    #include <string>
    #include <vector>
    static const char kTable[] = R"j({"name":"events","segment_size":4096,"dimensions":[{"name":"country"},{"name":"event_name","cardinality":200},{"name":"install_time","type":"time"}],"metrics":[{"name":"count","type":"count"},{"name":"revenue","type":"double_sum"},{"name":"best","type":"int_max"}]})j";
    static const char kQuery[] = R"j({"type":"aggregate","table":"events","dimensions":["event_name","country"],"metrics":["revenue","count","best"],"filter":{"op":"ne","column":"country","value":"RU"},"having":{"op":"ge","column":"count","value":"3"},"sort":[{"column":"revenue","ascending":false},{"column":"country","ascending":true}],"limit":7})j";
    struct Seg { std::vector<uint32_t> country; std::vector<uint8_t> event; std::vector<uint32_t> t; std::vector<uint32_t> count; std::vector<double> revenue; std::vector<int32_t> best; size_t size = 0; };
    static void send(void* ctx, const std::vector<std::string>& row) {
      std::string line;
      for (size_t i = 0; i < row.size(); ++i) { if (i) line += "|"; line += row[i]; }
      printf("%s\n", line.c_str());
    }
    int main() {
      int table_identity = 0;                                   // &table of the reference: any stable address
      std::vector<std::string> countries = {"__exceeded", "US", "RU", "IL", "KZ", "CH"}, events = {"__exceeded", "purchase", "donate", "open", "refund"};
      std::vector<Seg> segs(2);
      uint64_t x = 88172645463325252ull;
      auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
      auto append = [&](Seg& s, size_t n) {
        for (size_t i = 0; i < n; ++i) {
          s.country.push_back(1 + rnd() % 5); s.event.push_back(1 + rnd() % 4); s.t.push_back(1496570140 - rnd() % 1000000);
          s.count.push_back(1 + rnd() % 3); s.revenue.push_back((double)(rnd() % 4000) / 8.0); s.best.push_back((int32_t)(rnd() % 2000) - 1000);
        }
        s.size += n;
      };
      for (auto& s : segs) { s.country.reserve(4096); s.event.reserve(4096); s.t.reserve(4096); s.count.reserve(4096); s.revenue.reserve(4096); s.best.reserve(4096); }
      append(segs[0], 4096); append(segs[1], 1500);
      for (int round = 0; round < 3; ++round) {
        if (round == 1) { append(segs[1], 700); countries.push_back("AZ"); segs[1].country[2100] = 6; }          // upsert appended rows, a new dictionary entry
        if (round == 2) { segs[0].revenue[17] += 1000.5; segs[0].count[17] += 4; viya::shim::Touch(&table_identity, 0, 17, 18); }   // upsert updated a row in place
        viya::shim::Session* s = viya::shim::Open(&table_identity, kTable, kQuery);
        for (uint32_t i = 0; i < segs.size(); ++i) {
          const void* cols[] = { segs[i].country.data(), segs[i].event.data(), segs[i].t.data(), segs[i].count.data(), segs[i].revenue.data(), segs[i].best.data() };
          viya::shim::Sync(s, i, segs[i].size, cols);
        }
        viya::shim::BindDict(s, 0, &countries); viya::shim::BindDict(s, 1, &events);
        uint64_t farg = 0xDEADBEEF00000002ull, harg = 0xABCDABCD00000003ull;   // AnyNum: only the low bytes are the value (code of "RU", count >= 3)
        viya::shim::Stats st{};
        printf("round %d\n", round);
        viya::shim::Run(s, &farg, 1, &harg, 1, 0, 7, send, nullptr, &st);
        printf("stats %llu %llu %llu %llu\n", (unsigned long long)st.scanned_segments, (unsigned long long)st.scanned_recs, (unsigned long long)st.aggregated_recs, (unsigned long long)st.output_recs);
        viya::shim::Release(s);
        // what the Database path says about the same rows: dump them for the Python side
        printf("data %zu %zu\n", segs[0].size, segs[1].size);
        for (auto& sg : segs) for (size_t r = 0; r < sg.size; ++r)
          printf("row %s|%s|%u|%u|%.17g|%d\n", countries[sg.country[r]].c_str(), events[sg.event[r]].c_str(), sg.t[r], sg.count[r], sg.revenue[r], sg.best[r]);
      }
      viya::shim::Close(&table_identity);
      return 0;
    }
''')


def test_session_api_matches_the_database_path(tmp_path):
    src = tmp_path / "driver.cc"
    src.write_text(DRIVER)
    exe = str(tmp_path / "driver")
    lib = os.path.join(ROOT, "viyadb_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", str(src), "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lviya_host", "-lviya_hip",
                    "-Wl,-rpath," + lib, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rounds, cur = [], None
    for line in out.stdout.splitlines():
        if line.startswith("round "):
            cur = {"rows": [], "data": []}
            rounds.append(cur)
        elif line.startswith("stats "):
            cur["stats"] = [int(v) for v in line.split()[1:]]
        elif line.startswith("row "):
            cur["data"].append(line[4:].split("|"))
        elif line.startswith("data "):
            pass
        else:
            cur["rows"].append(line.split("|"))
    assert len(rounds) == 3
    # expected rows: group, filter, having, the reference's string sort (sort.cc:24-75: FLOAT columns by value, strings by text), limit
    for k, r in enumerate(rounds):
        agg = {}
        for country, event, _t, count, revenue, best in r["data"]:
            if country == "RU":
                continue
            a = agg.setdefault((event, country), [0.0, 0, -2 ** 31])
            a[0] += float(revenue); a[1] += int(count); a[2] = max(a[2], int(best))
        rows = [(e, c, v[0], v[1], v[2]) for (e, c), v in agg.items() if v[1] >= 3]
        rows.sort(key=lambda x: (-x[2], x[1]))
        want = [[e, c, ("%.15g" % rev), str(cnt), str(b)] for e, c, rev, cnt, b in rows[:7]]
        assert r["rows"] == want, (k, r["rows"][:3], want[:3])
        assert r["stats"][0] == 2 and r["stats"][1] == len(r["data"]) and r["stats"][2] == len(agg) and r["stats"][3] == len(want)
    assert any(row[0] == "AZ" for row in rounds[1]["data"])


BITSET_DRIVER = textwrap.dedent(r'''
    #include <viya_shim.h>
    #include <cstdio>
    #include <string>
    #include <vector>
    // the reference's UserEvents shape (test/db.h:151-164): a bitset metric is a column of util::Bitset<4> OBJECTS; the generated text
    // walks every row's set into CSR (tools/gen_shim_tu.py). Here plain vectors stand for the Roaring sets.
    static const char kTable[] = R"j({"name":"user_events","segment_size":2048,"dimensions":[{"name":"country"},{"name":"event_name"},{"name":"time","type":"uint"}],"metrics":[{"name":"user_id","type":"bitset"},{"name":"count","type":"count"}]})j";
    static const char kQuery[] = R"j({"type":"aggregate","table":"user_events","dimensions":["country"],"metrics":["user_id","count"],"filter":{"op":"gt","column":"time","value":"1495475514"}})j";
    struct Seg { std::vector<uint32_t> country, event, t, count; std::vector<std::vector<uint32_t>> users; size_t size = 0; };
    static void send(void* ctx, const std::vector<std::string>& row) {
      std::string line;
      for (size_t i = 0; i < row.size(); ++i) { if (i) line += "|"; line += row[i]; }
      printf("%s\n", line.c_str());
    }
    int main() {
      int table_identity = 0;
      std::vector<std::string> countries = {"__exceeded", "US", "RU", "IL", "KZ"}, events = {"__exceeded", "open", "close"};
      std::vector<Seg> segs(2);
      uint64_t x = 0x2545F4914F6CDD1Dull;
      auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
      auto append = [&](Seg& s, size_t n) {
        for (size_t i = 0; i < n; ++i) {
          s.country.push_back(1 + rnd() % 4); s.event.push_back(1 + rnd() % 2); s.t.push_back(1495475000 + rnd() % 1000); s.count.push_back(1 + rnd() % 2);
          std::vector<uint32_t> u; const size_t k = rnd() % 4; for (size_t q = 0; q < k; ++q) { const uint32_t id = rnd() % 300; bool have = false; for (auto v : u) have |= v == id; if (!have) u.push_back(id); }
          s.users.push_back(u);
        }
        s.size += n;
      };
      for (auto& s : segs) { s.country.reserve(2048); s.event.reserve(2048); s.t.reserve(2048); s.count.reserve(2048); }
      append(segs[0], 2048); append(segs[1], 600);
      unsigned walked = 0;
      for (int round = 0; round < 4; ++round) {
        if (round == 1) append(segs[1], 300);                                                          // upsert appended rows
        if (round == 2) { segs[0].users[40].push_back(100000 + round); segs[0].count[40] += 1; viya::shim::Touch(&table_identity, 0, 40, 41); }   // `_j |= metrics._j` on an existing row
        viya::shim::Session* s = viya::shim::Open(&table_identity, kTable, kQuery);
        for (uint32_t i = 0; i < segs.size(); ++i) {
          const void* cols[] = { segs[i].country.data(), segs[i].event.data(), segs[i].t.data(), nullptr, segs[i].count.data() };
          viya::shim::Sync(s, i, segs[i].size, cols);
          if (const uint64_t walk_rows = viya::shim::BitsetStale(s, i, segs[i].size)) {
            ++walked;
            std::vector<uint64_t> offsets(walk_rows + 1, 0);
            std::vector<uint32_t> ids;
            for (size_t r = 0; r < walk_rows; ++r) { ids.insert(ids.end(), segs[i].users[r].begin(), segs[i].users[r].end()); offsets[r + 1] = ids.size(); }
            viya::shim::SyncBitset(s, i, 0, walk_rows, offsets.data(), ids.data());
          }
        }
        viya::shim::BindDict(s, 0, &countries); viya::shim::BindDict(s, 1, &events);
        uint64_t farg = 0x1111111100000000ull | 1495475514u;
        viya::shim::Stats st{};
        printf("round %d walked %u\n", round, walked);
        viya::shim::Run(s, &farg, 1, nullptr, 0, 0, 0, send, nullptr, &st);
        printf("stats %llu %llu %llu %llu\n", (unsigned long long)st.scanned_segments, (unsigned long long)st.scanned_recs, (unsigned long long)st.aggregated_recs, (unsigned long long)st.output_recs);
        viya::shim::Release(s);
        for (auto& sg : segs) for (size_t r = 0; r < sg.size; ++r) {
          printf("row %s|%u|%u|", countries[sg.country[r]].c_str(), sg.t[r], sg.count[r]);
          for (size_t q = 0; q < sg.users[r].size(); ++q) printf(q ? ",%u" : "%u", sg.users[r][q]);
          printf("\n");
        }
      }
      viya::shim::Close(&table_identity);
      return 0;
    }
''')


def test_bitset_table_through_the_session_api(tmp_path):
    """Open / Sync / BitsetStale / SyncBitset / Run for a table with a count-distinct metric — what the generated text does for the
    reference's UserEvents (tools/gen_shim_tu.py walks util::Bitset<4> objects; plain vectors stand for them here): distinct users and
    COUNT per country as the reference computes them (`_j |= metrics._j`, cardinality(): store.cc:153-155, bitset.h:26-67), after rows
    were appended and after one row's set grew in place; a segment's sets are walked again only when it changed."""
    src = tmp_path / "bitset_driver.cc"
    src.write_text(BITSET_DRIVER)
    exe = str(tmp_path / "bitset_driver")
    lib = os.path.join(ROOT, "viyadb_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", str(src), "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lviya_host", "-lviya_hip",
                    "-Wl,-rpath," + lib, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rounds, cur = [], None
    for line in out.stdout.splitlines():
        if line.startswith("round "):
            cur = {"rows": [], "data": [], "walked": int(line.split()[3])}
            rounds.append(cur)
        elif line.startswith("stats "):
            cur["stats"] = [int(v) for v in line.split()[1:]]
        elif line.startswith("row "):
            cur["data"].append(line[4:].split("|"))
        else:
            cur["rows"].append(line.split("|"))
    assert len(rounds) == 4
    assert [r["walked"] for r in rounds] == [2, 3, 4, 4]        # both segments; the one that grew; the one Touched; nothing changed
    for r in rounds:
        users, counts = {}, {}
        for country, t, count, ids in r["data"]:
            if int(t) <= 1495475514:
                continue
            users.setdefault(country, set()).update(int(v) for v in ids.split(",") if v)
            counts[country] = counts.get(country, 0) + int(count)
        want = sorted([c, str(len(users[c])), str(counts[c])] for c in counts)
        assert sorted(r["rows"]) == want, (r["rows"], want)
        assert r["stats"][0] == 2 and r["stats"][1] == len(r["data"]) and r["stats"][2] == len(counts) == r["stats"][3]


CONCURRENT_DRIVER = textwrap.dedent(r"""
    // The reference's operating mode at the swap point: `query_threads` read-pool threads inside the generated function on ONE table
    // (src/db/database.cc:28-34, src/server/http/service.cc:119) while the writer thread upserts (service.cc:103): rows appended to the
    // last segment (size published after the row is written: Segment::Insert), metrics of existing rows updated in place + Touch
    // (src/codegen/db/upsert.cc:384-411). Every reader does what the generated text does: Open, (Pin, size() snapshot, Sync) per segment
    // of segments_copy(), BindDict, Run, Release.
    #include <viya_shim.h>
    #include <atomic>
    #include <chrono>
    #include <cstdio>
    #include <mutex>
    #include <shared_mutex>
    #include <string>
    #include <thread>
    #include <vector>
    constexpr size_t CAP = 4096;
    #ifdef WITH_BITSET
    static const char kTable[] = R"j({"name":"user_events","segment_size":4096,"dimensions":[{"name":"country"},{"name":"event_name"},{"name":"time","type":"uint"}],"metrics":[{"name":"user_id","type":"bitset"},{"name":"count","type":"count"}]})j";
    static const char* kQuery[2] = {
      R"j({"type":"aggregate","table":"user_events","dimensions":["country"],"metrics":["user_id","count"],"filter":{"op":"gt","column":"time","value":"1495475514"}})j",
      R"j({"type":"aggregate","table":"user_events","dimensions":["event_name"],"metrics":["user_id"],"filter":{"op":"ne","column":"country","value":"RU"}})j"};
    struct Seg { uint32_t country[CAP], event[CAP], t[CAP], count[CAP]; std::vector<uint32_t> users[CAP]; std::atomic<size_t> size{0}; };
    static std::shared_mutex users_mu;            // the test's stand-in for "a Roaring set is not read while it is written"
    #else
    static const char kTable[] = R"j({"name":"events","segment_size":4096,"dimensions":[{"name":"country"},{"name":"event_name","cardinality":200},{"name":"install_time","type":"time"}],"metrics":[{"name":"count","type":"count"},{"name":"revenue","type":"double_sum"},{"name":"best","type":"int_max"}]})j";
    static const char* kQuery[2] = {
      R"j({"type":"aggregate","table":"events","dimensions":["event_name","country"],"metrics":["revenue","count","best"],"filter":{"op":"ne","column":"country","value":"RU"}})j",
      R"j({"type":"aggregate","table":"events","dimensions":["country"],"metrics":["count","revenue"],"filter":{"op":"eq","column":"event_name","value":"purchase"}})j"};
    struct Seg { uint32_t country[CAP]; uint8_t event[CAP]; uint32_t t[CAP]; uint32_t count[CAP]; double revenue[CAP]; int32_t best[CAP]; std::atomic<size_t> size{0}; };
    #endif
    static Seg* segs[64];
    static std::atomic<uint32_t> nseg{0};
    static std::atomic<uint64_t> version{0};      // in-place updates completed (written AND Touched)
    static int table_identity = 0;
    static std::vector<std::string> countries = {"__exceeded", "US", "RU", "IL", "KZ", "CH"}, events = {"__exceeded", "purchase", "donate", "open", "refund"};
    static std::mutex out_mu;
    struct Rows { std::string text; };
    static void send(void* ctx, const std::vector<std::string>& row) {
      std::string& t = static_cast<Rows*>(ctx)->text;
      t += "r ";
      for (size_t i = 0; i < row.size(); ++i) { if (i) t += "|"; t += row[i]; }
      t += "\n";
    }
    static uint64_t rng_state = 88172645463325252ull;
    static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
    static void append(size_t n) {                // writer thread only
      while (n) {
        Seg* s = segs[nseg.load() - 1];
        size_t at = s->size.load();
        if (at == CAP) { segs[nseg.load()] = new Seg(); nseg.fetch_add(1); continue; }
        s->country[at] = 1 + rnd() % 5; s->event[at] = 1 + rnd() % 4; s->t[at] = 1495475000 + rnd() % 1000; s->count[at] = 1 + rnd() % 3;
    #ifdef WITH_BITSET
        { std::unique_lock<std::shared_mutex> wl(users_mu); const size_t k = rnd() % 4; for (size_t q = 0; q < k; ++q) { const uint32_t id = rnd() % 300; bool have = false; for (auto v : s->users[at]) have |= v == id; if (!have) s->users[at].push_back(id); } }
    #else
        s->revenue[at] = (double)(rnd() % 4000) / 8.0; s->best[at] = (int32_t)(rnd() % 2000) - 1000;
    #endif
        s->size.store(at + 1, std::memory_order_release);
        --n;
      }
    }
    static void one_query(int tid, int qid, std::string& log) {
      const uint64_t v0 = version.load();
      viya::shim::Session* s = viya::shim::Open(&table_identity, kTable, kQuery[qid]);
      const uint32_t n = nseg.load();
      std::vector<size_t> sizes(n);
      for (uint32_t i = 0; i < n; ++i) {
        Seg* sg = segs[i];
        sizes[i] = sg->size.load(std::memory_order_acquire);
        viya::shim::Pin(s, i, sg, sizeof(Seg));
    #ifdef WITH_BITSET
        const void* cols[] = { sg->country, sg->event, sg->t, nullptr, sg->count };
    #else
        const void* cols[] = { sg->country, sg->event, sg->t, sg->count, sg->revenue, sg->best };
    #endif
        viya::shim::Sync(s, i, sizes[i], cols);
    #ifdef WITH_BITSET
        if (const uint64_t walk_rows = viya::shim::BitsetStale(s, i, sizes[i])) {
          std::vector<uint64_t> offsets(walk_rows + 1, 0);
          std::vector<uint32_t> ids;
          { std::shared_lock<std::shared_mutex> rl(users_mu);
            for (size_t r = 0; r < walk_rows; ++r) { ids.insert(ids.end(), sg->users[r].begin(), sg->users[r].end()); offsets[r + 1] = ids.size(); } }
          viya::shim::SyncBitset(s, i, 0, walk_rows, offsets.data(), ids.data());
        }
    #endif
      }
      viya::shim::BindDict(s, 0, &countries); viya::shim::BindDict(s, 1, &events);
    #ifdef WITH_BITSET
      uint64_t farg = qid == 0 ? (0x1111111100000000ull | 1495475514u) : 0xDEADBEEF00000002ull;
    #else
      uint64_t farg = qid == 0 ? 0xDEADBEEF00000002ull : 0xABCDABCD00000001ull;       // AnyNum: code of "RU" / of "purchase" in the low bytes
    #endif
      viya::shim::Stats st{};
      Rows rows;
      viya::shim::Run(s, &farg, 1, nullptr, 0, 0, 0, send, &rows, &st);
      viya::shim::Release(s);
      const uint64_t v1 = version.load();
      char head[256];
      snprintf(head, sizeof(head), "query %d %d %llu %llu %llu %llu %u", tid, qid, (unsigned long long)v0, (unsigned long long)v1, (unsigned long long)st.scanned_recs, (unsigned long long)st.aggregated_recs, n);
      log += head;
      for (size_t z : sizes) log += " " + std::to_string(z);
      log += "\n" + rows.text + "end\n";
    }
    int main() {
      for (int i = 0; i < 6; ++i) { segs[i] = new Seg(); nseg.store(i + 1); append(i < 5 ? CAP : 1500); }
      std::string warm;
      one_query(-1, 0, warm); one_query(-1, 1, warm);        // everything that exists so far is mirrored before the writer starts
      std::atomic<bool> stop{false};
      struct Op { uint32_t seg, row; uint32_t id; };
      std::vector<Op> ops;
      std::thread writer([&] {
        // in-place updates go to rows of segments 0..4 (full and mirrored), never closer than 97 rows to the previous one of the same segment;
        // appends go to the last segment
        for (uint64_t k = 0; !stop.load() && k < 4000; ++k) {
          append(20 + rnd() % 120);
    #ifdef WITH_BITSET
          const uint32_t seg = 0, row = (uint32_t)((k * 97) % CAP), id = 1000000u + (uint32_t)k;
          { std::unique_lock<std::shared_mutex> wl(users_mu); segs[seg]->users[row].push_back(id); }
    #else
          const uint32_t seg = (uint32_t)(k % 5), row = (uint32_t)(((k / 5) * 97) % CAP), id = 0;
          segs[seg]->revenue[row] += 0.5;
    #endif
          viya::shim::Touch(&table_identity, seg, row, row + 1);
          ops.push_back(Op{seg, row, id});
          version.fetch_add(1);
          std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
      });
      std::vector<std::thread> readers;
      std::vector<std::string> logs(4);
      for (int t = 0; t < 4; ++t) readers.emplace_back([&, t] { for (int it = 0; it < 12; ++it) one_query(t, (t + it) % 2, logs[t]); });
      for (auto& r : readers) r.join();
      stop.store(true);
      writer.join();
      std::string last;
      one_query(9, 0, last); one_query(9, 1, last);           // a quiet table: exactly one answer is right
      for (auto& l : logs) fputs(l.c_str(), stdout);
      fputs(last.c_str(), stdout);
      printf("ops %zu\n", ops.size());
      for (auto& o : ops) printf("op %u %u %u\n", o.seg, o.row, o.id);
      const uint32_t n = nseg.load();
      for (uint32_t i = 0; i < n; ++i) {
        const size_t sz = segs[i]->size.load();
        printf("seg %u %zu\n", i, sz);
        for (size_t r = 0; r < sz; ++r) {
    #ifdef WITH_BITSET
          printf("d %s|%s|%u|%u|", countries[segs[i]->country[r]].c_str(), events[segs[i]->event[r]].c_str(), segs[i]->t[r], segs[i]->count[r]);
          for (size_t q = 0; q < segs[i]->users[r].size(); ++q) printf(q ? ",%u" : "%u", segs[i]->users[r][q]);
          printf("\n");
    #else
          printf("d %s|%s|%u|%u|%.17g|%d\n", countries[segs[i]->country[r]].c_str(), events[segs[i]->event[r]].c_str(), segs[i]->t[r], segs[i]->count[r], segs[i]->revenue[r], segs[i]->best[r]);
    #endif
        }
      }
      viya::shim::Close(&table_identity);
      return 0;
    }
""")


def _run_concurrent(tmp_path, bitset):
    src = tmp_path / "concurrent_driver.cc"
    src.write_text(CONCURRENT_DRIVER)
    exe = str(tmp_path / ("concurrent_bitset" if bitset else "concurrent"))
    lib = os.path.join(ROOT, "viyadb_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-pthread"] + (["-DWITH_BITSET"] if bitset else []) +
                   [str(src), "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lviya_host", "-lviya_hip", "-Wl,-rpath," + lib, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    queries, ops, segs, cur = [], [], [], None
    for line in out.stdout.splitlines():
        if line.startswith("query "):
            f = line.split()
            cur = {"tid": int(f[1]), "qid": int(f[2]), "v0": int(f[3]), "v1": int(f[4]), "scanned": int(f[5]), "groups": int(f[6]),
                   "sizes": [int(x) for x in f[8:8 + int(f[7])]], "rows": []}
        elif line.startswith("r "):
            cur["rows"].append(line[2:].split("|"))
        elif line == "end":
            queries.append(cur)
        elif line.startswith("op "):
            ops.append(tuple(int(x) for x in line.split()[1:]))
        elif line.startswith("seg "):
            segs.append([])
        elif line.startswith("d "):
            segs[-1].append(line[2:].split("|"))
    return queries, ops, segs


def test_four_readers_and_a_writer_on_one_table(tmp_path):
    """Four threads inside the 'generated function' on one table, two query texts, while a writer appends rows and updates revenue of
    existing rows in place (Touch). Every answer must be the oracle's for the size() snapshot THAT call took, with every in-place update
    that was complete when the call began applied, none that began after it ended, and each of those in between applied or not — a row
    updated while it is scanned is seen old or new, the reference's own race between its readers and its writer (upsert.cc:384-396 takes
    no lock a reader would see). The last queries run on a quiet table: exactly one answer is right."""
    queries, ops, segs = _run_concurrent(tmp_path, bitset=False)
    assert len(queries) == 4 * 12 + 2
    later = {}                                    # (seg, row) -> indices of the in-place updates that hit it
    for k, (seg, row, _id) in enumerate(ops):
        later.setdefault((seg, row), []).append(k)

    def group_of(q, rec):
        country, event = rec[0], rec[1]
        if q["qid"] == 0:
            return None if country == "RU" else (event, country)
        return (country,) if event == "purchase" else None

    overlapped = 0
    for q in queries:
        assert q["scanned"] == sum(q["sizes"]), q
        lo, hi = q["v0"], min(q["v1"] + 1, len(ops))
        agg = {}                                  # group -> [revenue with the updates < lo, count, best, updates of the window that hit the group]
        for si, n in enumerate(q["sizes"]):
            for r in range(n):
                rec = segs[si][r]
                g = group_of(q, rec)
                if g is None:
                    continue
                hits = later.get((si, r), ())
                a = agg.setdefault(g, [0.0, 0, -2 ** 31, 0])
                a[0] += float(rec[4]) - 0.5 * sum(1 for k in hits if k >= lo)       # (the dump holds the FINAL values)
                a[1] += int(rec[3]); a[2] = max(a[2], int(rec[5])); a[3] += sum(1 for k in hits if lo <= k < hi)
        got = {}
        for row in q["rows"]:
            if q["qid"] == 0:
                got[(row[0], row[1])] = (float(row[2]), int(row[3]), int(row[4]))
            else:
                got[(row[0],)] = (float(row[2]), int(row[1]), None)
        assert q["groups"] == len(agg) and set(got) == set(agg), (q["tid"], q["qid"], sorted(set(got) ^ set(agg))[:5])
        for g, (rev, cnt, best, window) in agg.items():
            grev, gcnt, gbest = got[g]
            assert gcnt == cnt and (gbest is None or gbest == best), (q["tid"], q["qid"], g, got[g], agg[g])
            extra = (grev - rev) / 0.5
            assert extra == int(extra) and 0 <= extra <= window, (q["tid"], q["qid"], q["v0"], q["v1"], g, grev, rev, window)
        overlapped += q["v1"] > q["v0"]
    assert overlapped >= 8, "the writer did not run beside the readers"
    assert len({tuple(q["sizes"]) for q in queries}) >= 8, "snapshots did not move"
    assert all(q["v0"] == q["v1"] == len(ops) for q in queries[-2:])


def test_four_readers_and_a_writer_on_a_bitset_table(tmp_path):
    """The same on a table with a count-distinct metric: rows appended (with their sets) and ids added to sets of existing rows of segment 0
    in place (`_j |= metrics._j` + Touch). A call with a SMALLER snapshot than the mirror already holds must not shrink the CSR under
    another call's query (BitsetStale says how many rows to walk), and a Touch during a walk leaves the segment stale."""
    queries, ops, segs = _run_concurrent(tmp_path, bitset=True)
    assert len(queries) == 4 * 12 + 2
    added = {uid: k for k, (_seg, _row, uid) in enumerate(ops)}        # ids of in-place updates are unique

    def group_of(q, rec):
        country, event, t = rec[0], rec[1], int(rec[2])
        if q["qid"] == 0:
            return country if t > 1495475514 else None
        return None if country == "RU" else event

    overlapped = 0
    for q in queries:
        assert q["scanned"] == sum(q["sizes"]), q
        lo, hi = q["v0"], min(q["v1"] + 1, len(ops))
        users, maybe, counts = {}, {}, {}
        for si, n in enumerate(q["sizes"]):
            for r in range(n):
                rec = segs[si][r]
                g = group_of(q, rec)
                if g is None:
                    continue
                ids = [int(v) for v in rec[4].split(",") if v]
                users.setdefault(g, set()).update(v for v in ids if added.get(v, -1) < lo)
                maybe.setdefault(g, set()).update(v for v in ids if lo <= added.get(v, -1) < hi)
                counts[g] = counts.get(g, 0) + int(rec[3])
        got = {row[0]: row[1:] for row in q["rows"]}
        assert q["groups"] == len(counts) and set(got) == set(counts), (q["tid"], q["qid"], got, counts)
        for g in counts:
            card = int(got[g][0])
            assert len(users[g]) <= card <= len(users[g]) + len(maybe[g]), (q["tid"], q["qid"], q["v0"], q["v1"], g, card, len(users[g]), len(maybe[g]))
            if q["qid"] == 0:
                assert int(got[g][1]) == counts[g], (q["tid"], g, got[g], counts[g])
        overlapped += q["v1"] > q["v0"]
    assert overlapped >= 8, "the writer did not run beside the readers"
    assert all(q["v0"] == q["v1"] == len(ops) for q in queries[-2:])
