// Graph ingest on the host: the reader of data_loader.py:61-96 and the (subject, relation) -> tails
// index of data_loader.py:80-96 as native code, for triple files far beyond what two Python passes with dict lookups
// handle. Integer / byte work only; results are identical to the Python loader's (tests/test_host.py).
//
//   ids: first-seen order over train, valid, test; per line subject, relation, object; names lower-cased on insertion
//        (data_loader.py:64-70). The reference then looks the RAW tokens up (data_loader.py:84-86), which raises
//        KeyError for any token that lower-casing changes: reproduced as MGCN_EINVAL with the offending token.
//   lines: str.strip().split() semantics for ASCII whitespace; a line that does not hold exactly three tokens is an
//        error (the reference's tuple unpacking raises ValueError). Bytes >= 0x80 are outside this reader
//        (str.lower() is Unicode-aware): MGCN_EUNSUPPORTED, the caller falls back to the Python reader.
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "mgcn_common.h"

struct mgcn_ingest {
  std::vector<std::string> names[2];            // 0: entities, 1: relations, in id order
  std::vector<int64_t> triples[3];              // train, valid, test: flattened (s, r, o)
};

namespace {

// str.isspace() for ASCII: space, \t \n \v \f \r and the separators 0x1c-0x1f
bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }

struct Interner {
  std::unordered_map<std::string, int64_t> map;
  std::vector<std::string> *names;
  int64_t intern(const std::string &lower) {
    auto it = map.find(lower);
    if (it != map.end()) return it->second;
    const int64_t id = int64_t(names->size());
    map.emplace(lower, id);
    names->push_back(lower);
    return id;
  }
};

int read_file(const char *path, std::string *out) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return mgcn::fail(MGCN_EINVAL, "ingest: cannot open %s", path);
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out->resize(size_t(size > 0 ? size : 0));
  const size_t got = size > 0 ? std::fread(&(*out)[0], 1, size_t(size), f) : 0;
  std::fclose(f);
  if (got != out->size()) return mgcn::fail(MGCN_EINVAL, "ingest: short read on %s", path);
  return MGCN_OK;
}

// One split: tokens of every line -> ids. `raw_differs` records the first token that lower-casing changed.
int parse_split(const char *path, const std::string &text, Interner &ent, Interner &rel, std::vector<int64_t> *ids,
                std::string *raw_differs) {
  size_t pos = 0, line_no = 0;
  const size_t n = text.size();
  std::string tok[3];
  while (pos < n) {
    size_t eol = pos;   // universal newlines, as Python's text mode: \n, \r\n or a lone \r end a line
    while (eol < n && text[eol] != '\n' && text[eol] != '\r') ++eol;
    const size_t next = (eol + 1 < n && text[eol] == '\r' && text[eol + 1] == '\n') ? eol + 2 : eol + 1;
    ++line_no;
    int ntok = 0;
    size_t i = pos;
    while (i < eol) {
      while (i < eol && is_space(static_cast<unsigned char>(text[i]))) ++i;
      if (i >= eol) break;
      size_t j = i;
      while (j < eol && !is_space(static_cast<unsigned char>(text[j]))) ++j;
      if (ntok == 3) return mgcn::fail(MGCN_EINVAL, "ingest: %s line %zu: more than three tokens (ValueError)", path, line_no);
      tok[ntok].assign(text, i, j - i);
      for (char &c : tok[ntok]) {
        const unsigned char u = static_cast<unsigned char>(c);
        if (u >= 0x80) return mgcn::fail(MGCN_EUNSUPPORTED, "ingest: %s line %zu: non-ASCII token, use the Python reader", path, line_no);
        if (u >= 'A' && u <= 'Z') {
          if (raw_differs->empty()) raw_differs->assign(text, i, j - i);
          c = char(u - 'A' + 'a');
        }
      }
      ++ntok;
      i = j;
    }
    if (ntok != 3) return mgcn::fail(MGCN_EINVAL, "ingest: %s line %zu: %d tokens instead of three (ValueError)", path, line_no, ntok);
    // insertion order of data_loader.py:66-69: subject, relation, object
    const int64_t s = ent.intern(tok[0]);
    const int64_t r = rel.intern(tok[1]);
    const int64_t o = ent.intern(tok[2]);
    ids->push_back(s);
    ids->push_back(r);
    ids->push_back(o);
    pos = next;
  }
  return MGCN_OK;
}

}  // namespace

extern "C" int mgcn_ingest_open(const char *train_path, const char *valid_path, const char *test_path, mgcn_ingest **out) {
  MGCN_REQUIRE(train_path && valid_path && test_path && out, "ingest_open: null argument");
  *out = nullptr;
  mgcn_ingest *h = new mgcn_ingest();
  Interner ent{{}, &h->names[0]}, rel{{}, &h->names[1]};
  const char *paths[3] = {train_path, valid_path, test_path};
  std::string raw_differs;
  for (int split = 0; split < 3; ++split) {
    std::string text;
    int rc = read_file(paths[split], &text);
    if (rc == MGCN_OK) rc = parse_split(paths[split], text, ent, rel, &h->triples[split], &raw_differs);
    if (rc != MGCN_OK) {
      delete h;
      return rc;
    }
  }
  if (!raw_differs.empty()) {   // data_loader.py:84-86 looks the raw token up in maps keyed by the lower-cased name
    delete h;
    return mgcn::fail(MGCN_EINVAL, "ingest: KeyError: '%s' (names are stored lower-cased, looked up raw)", raw_differs.c_str());
  }
  *out = h;
  return MGCN_OK;
}

extern "C" void mgcn_ingest_close(mgcn_ingest *h) { delete h; }

extern "C" int64_t mgcn_ingest_count(const mgcn_ingest *h, int32_t what) {
  // what: 0 entities, 1 relations, 2 / 3 / 4 triples of train / valid / test
  if (!h || what < 0 || what > 4) return -1;
  return what < 2 ? int64_t(h->names[what].size()) : int64_t(h->triples[what - 2].size() / 3);
}

extern "C" int mgcn_ingest_triples(const mgcn_ingest *h, int32_t split, int64_t *triples_host) {
  MGCN_REQUIRE(h && split >= 0 && split < 3, "ingest_triples: bad handle or split");
  const std::vector<int64_t> &v = h->triples[split];
  MGCN_REQUIRE(v.empty() || triples_host, "ingest_triples: null output");
  if (!v.empty()) std::memcpy(triples_host, v.data(), v.size() * sizeof(int64_t));
  return MGCN_OK;
}

extern "C" int64_t mgcn_ingest_names_bytes(const mgcn_ingest *h, int32_t kind) {
  if (!h || kind < 0 || kind > 1) return -1;
  int64_t total = 0;
  for (const std::string &s : h->names[kind]) total += int64_t(s.size());
  return total;
}

extern "C" int mgcn_ingest_names(const mgcn_ingest *h, int32_t kind, char *bytes_host, int64_t *offsets_host) {
  MGCN_REQUIRE(h && kind >= 0 && kind <= 1 && offsets_host, "ingest_names: bad arguments");
  int64_t pos = 0;
  int64_t i = 0;
  for (const std::string &s : h->names[kind]) {
    offsets_host[i++] = pos;
    MGCN_REQUIRE(s.empty() || bytes_host, "ingest_names: null byte buffer");
    if (!s.empty()) std::memcpy(bytes_host + pos, s.data(), s.size());
    pos += int64_t(s.size());
  }
  offsets_host[i] = pos;
  return MGCN_OK;
}

// (subject, relation id) -> sorted distinct tails over the given triples in BOTH directions (the reverse query of
// (s, r, o) is (o, r + num_relations, s)): the loader's sr2o (data_loader.py:80-96) as key = s * 2R + r, CSR form.
// Call with keys_host == NULL to get the sizes (*num_keys, *num_tails), then again with buffers of those sizes.
extern "C" int mgcn_filter_index_build(int64_t num_triples, const int64_t *triples_host, int64_t num_relations,
                                       int64_t *keys_host, int64_t *ptr_host, int32_t *tails_host, int64_t *num_keys,
                                       int64_t *num_tails) {
  MGCN_REQUIRE(num_triples >= 0 && num_relations >= 0 && (num_triples == 0 || triples_host) && num_keys && num_tails,
               "filter_index_build: bad arguments");
  const int64_t r2 = 2 * num_relations;
  std::vector<std::pair<int64_t, int32_t>> items;
  items.reserve(size_t(2 * num_triples));
  for (int64_t i = 0; i < num_triples; ++i) {
    const int64_t s = triples_host[3 * i], r = triples_host[3 * i + 1], o = triples_host[3 * i + 2];
    MGCN_REQUIRE(s >= 0 && o >= 0 && s < (int64_t(1) << 31) && o < (int64_t(1) << 31) && r >= 0 && r < num_relations,
                 "filter_index_build: triple %lld out of range", (long long)i);
    items.emplace_back(s * r2 + r, int32_t(o));
    items.emplace_back(o * r2 + r + num_relations, int32_t(s));
  }
  std::sort(items.begin(), items.end());
  items.erase(std::unique(items.begin(), items.end()), items.end());
  int64_t nk = 0;
  for (size_t i = 0; i < items.size(); ++i)
    if (i == 0 || items[i].first != items[i - 1].first) ++nk;
  if (!keys_host) {
    *num_keys = nk;
    *num_tails = int64_t(items.size());
    return MGCN_OK;
  }
  MGCN_REQUIRE(ptr_host && (items.empty() || tails_host), "filter_index_build: null output");
  MGCN_REQUIRE(*num_keys == nk && *num_tails == int64_t(items.size()), "filter_index_build: buffer sizes do not match");
  int64_t k = -1;
  for (size_t i = 0; i < items.size(); ++i) {
    if (i == 0 || items[i].first != items[i - 1].first) {
      ++k;
      keys_host[k] = items[i].first;
      ptr_host[k] = int64_t(i);
    }
    tails_host[i] = items[i].second;
  }
  ptr_host[nk] = int64_t(items.size());
  return MGCN_OK;
}
