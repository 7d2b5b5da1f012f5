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
          if (viya::shim::BitsetStale(s, i, segs[i].size)) {
            ++walked;
            std::vector<uint64_t> offsets(segs[i].size + 1, 0);
            std::vector<uint32_t> ids;
            for (size_t r = 0; r < segs[i].size; ++r) { ids.insert(ids.end(), segs[i].users[r].begin(), segs[i].users[r].end()); offsets[r + 1] = ids.size(); }
            viya::shim::SyncBitset(s, i, 0, segs[i].size, offsets.data(), ids.data());
          }
        }
        viya::shim::BindDict(s, 0, &countries); viya::shim::BindDict(s, 1, &events);
        uint64_t farg = 0x1111111100000000ull | 1495475514u;
        viya::shim::Stats st{};
        printf("round %d walked %u\n", round, walked);
        viya::shim::Run(s, &farg, 1, nullptr, 0, 0, 0, send, nullptr, &st);
        printf("stats %llu %llu %llu %llu\n", (unsigned long long)st.scanned_segments, (unsigned long long)st.scanned_recs, (unsigned long long)st.aggregated_recs, (unsigned long long)st.output_recs);
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
