// Local-file TSV loader (see data_source.h).  A file is read whole, cut into
// line-aligned chunks that are parsed by a few threads, and appended to the type's
// storage in file order -- so edge ids (= load order, memory_edge_storage.cc:53-57)
// are deterministic, which the reference's thread-interleaved loaders
// (graph_store.cc:76-93) do not guarantee.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "graphlearn/config.h"
#include "graphlearn/data_source.h"

namespace graphlearn {
namespace io {

AttributeInfo::AttributeInfo() : ignore_invalid(GLOBAL_FLAG(IgnoreInvalid) != 0) {}

// MurmurHash64A (A. Appleby, public domain) as used by the reference with seed
// 0xDECAFCAFFE (common/base/hash.cc:99-150): 8-byte little-endian blocks, then the
// 1..7 tail bytes, then the final avalanche.
uint64_t Hash64(const char* data, size_t n) {
  const uint64_t m = 0xc6a4a7935bd1e995ULL;
  const int r = 47;
  uint64_t h = 0xDECAFCAFFEULL ^ (n * m);
  const unsigned char* p = reinterpret_cast<const unsigned char*>(data);
  for (; n >= 8; n -= 8, p += 8) {
    uint64_t k;
    std::memcpy(&k, p, 8);
    k *= m;
    k ^= k >> r;
    k *= m;
    h ^= k;
    h *= m;
  }
  if (n > 0) {
    uint64_t tail = 0;
    for (size_t i = 0; i < n; ++i) tail |= (uint64_t)p[i] << (8 * i);
    h ^= tail;
    h *= m;
  }
  h ^= h >> r;
  h *= m;
  h ^= h >> r;
  return h;
}

namespace {

bool OnlyBlanksLeft(const char* end) {
  while (isspace((unsigned char)*end)) ++end;
  return *end == '\0';
}

// numeric.cc:174-208 (FastStringTo32/64/Float): strtol/strtof + optional trailing blanks
bool ToInt64(const char* s, int64_t* v) {
  char* end = nullptr;
  const long long r = strtoll(s, &end, 10);
  if (!OnlyBlanksLeft(end)) return false;
  *v = r;
  return true;
}
bool ToInt32(const char* s, int64_t* v) {
  char* end = nullptr;
  errno = 0;
  const long long r = strtoll(s, &end, 10);
  if (!OnlyBlanksLeft(end) || errno != 0 || r > INT32_MAX || r < INT32_MIN) return false;
  *v = r;
  return true;
}
bool ToFloat(const char* s, float* v) {
  char* end = nullptr;
  const float r = strtof(s, &end);
  if (!OnlyBlanksLeft(end)) return false;
  *v = r;
  return true;
}

// Fast paths that return exactly what strtoll / strtof return, with the general routine as
// the fallback for everything else (blanks, exponents, long mantissas, overflow ...).
//  * integers: [+-]digits, at most 18 digits.
//  * floats: [+-]digits[.digits] with at most 15 significant digits.  The mantissa (an exact
//    integer below 2^53) divided by an exact power of ten (<= 10^22) is ONE correctly rounded
//    double operation; narrowing that double to float equals strtof's direct rounding unless
//    the double sits exactly on a float rounding midpoint (the 29 dropped mantissa bits equal
//    0x10000000) -- those strings, and results outside the normal float range, take the
//    general routine.
inline bool FastInt64(const char* s, size_t len, int64_t* v) {
  if (len == 0 || len > 19) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '-' || s[0] == '+') {
    neg = s[0] == '-';
    i = 1;
  }
  if (i == len || len - i > 18) return false;
  int64_t r = 0;
  for (; i < len; ++i) {
    const unsigned d = (unsigned)(s[i] - '0');
    if (d > 9) return false;
    r = r * 10 + (int64_t)d;
  }
  *v = neg ? -r : r;
  return true;
}

inline bool FastFloat(const char* s, size_t len, float* v) {
  static const double kPow10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  size_t i = 0;
  bool neg = false;
  if (len > 0 && (s[0] == '-' || s[0] == '+')) {
    neg = s[0] == '-';
    i = 1;
  }
  uint64_t m = 0;
  int digits = 0, frac = 0;
  bool seen_dot = false, any = false;
  for (; i < len; ++i) {
    const char c = s[i];
    if (c == '.') {
      if (seen_dot) return false;
      seen_dot = true;
      continue;
    }
    const unsigned d = (unsigned)(c - '0');
    if (d > 9) return false;
    any = true;
    if (m != 0 || d != 0) ++digits;  // significant digits
    if (digits > 15) return false;
    m = m * 10 + d;
    if (seen_dot) ++frac;
  }
  if (!any || frac > 22) return false;
  double d = (double)m / kPow10[frac];
  if (m != 0 && !(d > 1e-30 && d < 1e30)) return false;
  uint64_t bits;
  std::memcpy(&bits, &d, 8);
  if ((bits & 0x1FFFFFFFull) == 0x10000000ull) return false;  // on a float midpoint: let strtof decide
  const float f = (float)d;
  *v = neg ? -f : f;
  return true;
}

DataType ToDataType(const std::string& t) {  // common/io/value.cc:21-36
  if (t == "int" || t == "int32") return kInt32;
  if (t == "long" || t == "int64") return kInt64;
  if (t == "float") return kFloat;
  if (t == "double") return kDouble;
  if (t == "string") return kString;
  return kUnknown;
}

// Schemas are compared after folding int32/int64 -> int and float/double -> float
// (common/io/value.h:117-120).
char Fold(DataType t) {
  switch (t) {
    case kInt32: case kInt64: return 'i';
    case kFloat: case kDouble: return 'f';
    case kString: return 's';
    default: return '?';
  }
}
std::string Describe(const std::string& folded) {
  std::string out;
  for (char c : folded) out += c == 'i' ? "int," : c == 'f' ? "float," : c == 's' ? "string," : "unknown,";
  return out;
}

struct File {
  std::string data;
  size_t body = 0;  // offset of the first record
  std::string schema;  // folded column types of the header line
};

Status ReadFile(const std::string& path, File* f) {
  FILE* fp = std::fopen(path.c_str(), "rb");
  if (!fp) return error::NotFound("cannot open data source '" + path + "'");
  std::fseek(fp, 0, SEEK_END);
  const long size = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  f->data.resize(size > 0 ? (size_t)size : 0);
  const size_t got = size > 0 ? std::fread(&f->data[0], 1, (size_t)size, fp) : 0;
  std::fclose(fp);
  if (got != f->data.size()) return error::Internal("short read on '" + path + "'");
  size_t eol = f->data.find('\n');
  if (eol == std::string::npos) eol = f->data.size();
  std::string header = f->data.substr(0, eol);
  if (!header.empty() && header.back() == '\r') header.pop_back();
  f->body = eol < f->data.size() ? eol + 1 : f->data.size();
  size_t at = 0;
  while (at <= header.size() && !header.empty()) {
    size_t tab = header.find('\t', at);
    if (tab == std::string::npos) tab = header.size();
    const std::string col = header.substr(at, tab - at);
    const size_t colon = col.rfind(':');
    f->schema += Fold(colon == std::string::npos ? kUnknown : ToDataType(col.substr(colon + 1)));
    at = tab + 1;
  }
  return Status::OK();
}

template <class Source>
std::string ExpectedSchema(const Source& s, int id_columns) {
  std::string e(id_columns, 'i');
  if (s.IsWeighted()) e += 'f';
  if (s.IsLabeled()) e += 'i';
  if (s.IsTimestamped()) e += 'i';
  if (s.IsAttributed()) e += 's';
  return e;
}

// One record = the tab separated fields of [begin, end).
struct Fields {
  const char* ptr[8];
  size_t len[8];
  int n = 0;
};
void SplitTabs(char* begin, char* end, Fields* f) {
  f->n = 0;
  char* tok = begin;
  for (char* p = begin;; ++p) {
    if (p == end || *p == '\t') {
      if (f->n < 8) {
        f->ptr[f->n] = tok;
        f->len[f->n] = (size_t)(p - tok);
      }
      ++f->n;
      if (p == end) break;
      *p = '\0';  // numeric fields are parsed in place
      tok = p + 1;
    }
  }
}

// Cuts the records into line-aligned pieces, one per thread; parse_one(begin, end, &piece)
// appends one record to its piece (columns) or reports why not.  Returning OUT_OF_RANGE
// means "skip this record".
template <class Piece, class ParseOne>
Status ParseChunks(File* f, int threads, const ParseOne& parse_one, std::vector<Piece>* out) {
  char* base = &f->data[0];
  const size_t size = f->data.size();
  std::vector<size_t> cut;
  cut.push_back(f->body);
  for (int t = 1; t < threads; ++t) {
    size_t at = f->body + (size - f->body) * t / threads;
    while (at < size && base[at - 1] != '\n') ++at;
    if (at > cut.back() && at < size) cut.push_back(at);
  }
  cut.push_back(size);
  const int parts = (int)cut.size() - 1;
  out->assign(parts, Piece());
  std::vector<Status> st(parts);
  auto work = [&](int part) {
    char* p = base + cut[part];
    char* stop = base + cut[part + 1];
    Piece* piece = &(*out)[part];
    while (p < stop) {
      char* eol = static_cast<char*>(std::memchr(p, '\n', (size_t)(stop - p)));
      char* next = eol ? eol + 1 : stop;
      char* end = eol ? eol : stop;
      if (end > p && end[-1] == '\r') --end;
      if (end > p) {
        *end = '\0';
        const error::Code rc = parse_one(p, end, piece, &st[part]);
        if (rc != error::OK && rc != error::OUT_OF_RANGE) return;  // st[part] holds the reason
      }
      p = next;
    }
  };
  if (parts == 1) {
    work(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < parts; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  for (const Status& s : st) {
    if (!s.ok()) return s;
  }
  return Status::OK();
}

int LoaderThreads(size_t bytes) {
  if (bytes < (1u << 20)) return 1;
  unsigned hw = std::thread::hardware_concurrency();
  return (int)(hw == 0 ? 4 : hw > 16 ? 16 : hw);
}

}  // namespace

Status ParseAttribute(const char* data, size_t len, const AttributeInfo& info, std::vector<int64_t>* ints,
                      std::vector<float>* floats, std::vector<std::string>* strings) {
  // tokens in place: [tb[i], te[i]) -- the delimiter is a SET of characters, empty tokens count
  const size_t want = info.types.size();
  const char* tb[64];
  const char* te[64];
  std::vector<const char*> big_b, big_e;
  const char** b = tb;
  const char** e = te;
  if (want > 64) {
    big_b.resize(want);
    big_e.resize(want);
    b = big_b.data();
    e = big_e.data();
  }
  size_t n = 0;
  if (len > 0) {
    const bool one = info.delimiter.size() == 1;
    const char d0 = one ? info.delimiter[0] : 0;
    size_t start = 0;
    for (size_t i = 0; i <= len; ++i) {
      if (i == len || (one ? data[i] == d0 : info.delimiter.find(data[i]) != std::string::npos)) {
        if (n < want) {
          b[n] = data + start;
          e[n] = data + i;
        }
        ++n;
        start = i + 1;
      }
    }
  }
  if (n != want) return error::InvalidArgument("Unexpected attribute count");
  const size_t i0 = ints->size(), f0 = floats->size(), s0 = strings->size();
  for (size_t i = 0; i < n; ++i) {
    const DataType t = info.types[i];
    const size_t tl = (size_t)(e[i] - b[i]);
    const char* kind = nullptr;
    if (t == kInt32 || t == kInt64) {
      int64_t v = 0;
      bool ok = FastInt64(b[i], tl, &v);
      if (ok && t == kInt32) ok = v <= INT32_MAX && v >= INT32_MIN;
      if (!ok) {
        const std::string z(b[i], tl);
        ok = t == kInt32 ? ToInt32(z.c_str(), &v) : ToInt64(z.c_str(), &v);
      }
      if (ok) ints->push_back(v);
      else kind = t == kInt32 ? "int" : "int64";
    } else if (t == kFloat) {
      float v = 0.f;
      bool ok = FastFloat(b[i], tl, &v);
      if (!ok) {
        const std::string z(b[i], tl);
        ok = ToFloat(z.c_str(), &v);
      }
      if (ok) floats->push_back(v);
      else kind = "float";
    } else if (t == kDouble) {
      const std::string z(b[i], tl);
      char* end = nullptr;
      const double d = strtod(z.c_str(), &end);
      if (OnlyBlanksLeft(end)) floats->push_back(static_cast<float>(d));
      else kind = "double";
    } else if (t == kString) {
      if (!info.hash_buckets.empty() && info.hash_buckets[i] > 0) {
        ints->push_back((int64_t)(Hash64(b[i], tl) % (uint64_t)info.hash_buckets[i]));
      } else {
        strings->emplace_back(b[i], tl);
      }
    }
    if (kind) {
      ints->resize(i0);  // leave the outputs as they were: the record is rejected as a whole
      floats->resize(f0);
      strings->resize(s0);
      char msg[256];
      std::snprintf(msg, sizeof(msg), "The %dth attribute expect an %s, but got \"%.*s\".", (int)i, kind, (int)tl, b[i]);
      return error::InvalidArgument(msg);
    }
  }
  return Status::OK();
}

namespace {
inline bool FieldInt64(const Fields& f, int c, int64_t* v) {
  return FastInt64(f.ptr[c], f.len[c], v) || ToInt64(f.ptr[c], v);
}
inline bool FieldFloat(const Fields& f, int c, float* v) {
  return FastFloat(f.ptr[c], f.len[c], v) || ToFloat(f.ptr[c], v);
}
}  // namespace

namespace {
// A source path may be a directory: every regular file in it is one piece of the source
// (the reference's file system lists them, edge_loader_unittest.cpp:334-390); names are
// taken in sorted order so that load order -- hence edge ids -- is reproducible.
bool ListDirectory(const std::string& path, std::vector<std::string>* files) {
  struct stat st;
  if (::stat(path.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
  DIR* dir = ::opendir(path.c_str());
  if (!dir) return false;
  while (struct dirent* e = ::readdir(dir)) {
    const std::string name = e->d_name;
    if (name == "." || name == "..") continue;
    const std::string full = path + (path.empty() || path.back() == '/' ? "" : "/") + name;
    if (::stat(full.c_str(), &st) == 0 && S_ISREG(st.st_mode)) files->push_back(full);
  }
  ::closedir(dir);
  std::sort(files->begin(), files->end());
  return true;
}
}  // namespace

Status LoadEdges(const EdgeSource& source, GraphStore* store) {
  if (source.src_id_type.empty() || source.dst_id_type.empty() || source.edge_type.empty()) {
    return error::InvalidArgument("Node and edge types must be assigned.");
  }
  std::vector<std::string> pieces;
  if (ListDirectory(source.path, &pieces)) {
    for (const std::string& piece : pieces) {
      EdgeSource one = source;
      one.path = piece;
      Status st = LoadEdges(one, store);
      if (!st.ok()) return st;
    }
    return Status::OK();
  }
  File f;
  Status s = ReadFile(source.path, &f);
  if (!s.ok()) return s;
  const std::string expected = ExpectedSchema(source, 2);
  if (f.schema != expected) {
    return error::InvalidArgument("Invalid edge table schema, expected: " + Describe(expected) + ", but got: " +
                                  Describe(f.schema) + " (" + source.path + ")");
  }
  SideInfo info;
  ParseSideInfo(source, &info);
  info.type = source.edge_type;
  info.src_type = source.src_id_type;
  info.dst_type = source.dst_id_type;
  const int columns = (int)expected.size();
  const bool skip_bad = source.attr_info.ignore_invalid;
  const bool reversed = source.direction == kReversed;
  auto parse_one = [&](char* begin, char* end, EdgeColumns* out, Status* why) -> error::Code {
    Fields fld;
    SplitTabs(begin, end, &fld);
    int64_t src = 0, dst = 0, lab = 0, ts = 0;
    float w = 0.f;
    bool ok = fld.n == columns;
    int c = 0;
    if (ok) ok = FieldInt64(fld, c++, &src) && FieldInt64(fld, c++, &dst);
    if (ok && source.IsWeighted()) ok = FieldFloat(fld, c++, &w);
    if (ok && source.IsLabeled()) ok = FieldInt64(fld, c++, &lab);
    if (ok && source.IsTimestamped()) ok = FieldInt64(fld, c++, &ts);
    Status bad;
    if (!ok) bad = error::InvalidArgument("Invalid edge record in " + source.path);
    else if (!store->Owns(reversed ? dst : src)) {
      // Another shard's edge: read on.  With ignore_invalid off a malformed record is fatal, and it must
      // be fatal on EVERY rank (all ranks read the same files and meet in collectives afterwards): the
      // non-owners validate the attributes too, into a scratch they throw away.
      if (!skip_bad && source.IsAttributed()) {
        std::vector<int64_t> ti;
        std::vector<float> tf;
        std::vector<std::string> ts_;
        bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &ti, &tf, &ts_);
        if (!bad.ok()) {
          *why = bad;
          return bad.code();
        }
      }
      return error::OUT_OF_RANGE;
    } else if (source.IsAttributed()) bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &out->i_attrs, &out->f_attrs, &out->s_attrs);
    if (!bad.ok()) {
      if (skip_bad) return error::OUT_OF_RANGE;  // edge_loader.cc:70-78: ignore the record, read on
      *why = bad;
      return bad.code();
    }
    out->src.push_back(reversed ? dst : src);
    out->dst.push_back(reversed ? src : dst);
    if (source.IsWeighted()) out->weight.push_back(w);
    if (source.IsLabeled()) out->label.push_back((int32_t)lab);
    if (source.IsTimestamped()) out->timestamp.push_back(ts);
    return error::OK;
  };
  std::vector<EdgeColumns> parts;
  s = ParseChunks<EdgeColumns>(&f, LoaderThreads(f.data.size()), parse_one, &parts);
  if (!s.ok()) return s;
  Graph* graph = store->GetGraph(source.edge_type);
  for (EdgeColumns& part : parts) {  // file order: edge id = load order
    s = graph->AppendColumns(info, &part);
    if (!s.ok()) return s;
  }
  return Status::OK();
}

Status LoadNodes(const NodeSource& source, GraphStore* store) {
  if (source.id_type.empty()) return error::InvalidArgument("Node type must be assigned.");
  std::vector<std::string> pieces;
  if (ListDirectory(source.path, &pieces)) {
    for (const std::string& piece : pieces) {
      NodeSource one = source;
      one.path = piece;
      Status st = LoadNodes(one, store);
      if (!st.ok()) return st;
    }
    return Status::OK();
  }
  File f;
  Status s = ReadFile(source.path, &f);
  if (!s.ok()) return s;
  const std::string expected = ExpectedSchema(source, 1);
  if (f.schema != expected) {
    return error::InvalidArgument("Invalid node table schema, expected: " + Describe(expected) + ", but got: " +
                                  Describe(f.schema) + " (" + source.path + ")");
  }
  SideInfo info;
  ParseSideInfo(source, &info);
  info.type = source.id_type;
  const int columns = (int)expected.size();
  const bool skip_bad = source.attr_info.ignore_invalid;
  auto parse_one = [&](char* begin, char* end, NodeColumns* out, Status* why) -> error::Code {
    Fields fld;
    SplitTabs(begin, end, &fld);
    int64_t id = 0, lab = 0, ts = 0;
    float w = 0.f;
    bool ok = fld.n == columns;
    int c = 0;
    if (ok) ok = FieldInt64(fld, c++, &id);
    if (ok && source.IsWeighted()) ok = FieldFloat(fld, c++, &w);
    if (ok && source.IsLabeled()) ok = FieldInt64(fld, c++, &lab);
    if (ok && source.IsTimestamped()) ok = FieldInt64(fld, c++, &ts);
    Status bad;
    if (!ok) bad = error::InvalidArgument("Invalid node record in " + source.path);
    else if (!store->Owns(id)) {
      // another shard's node: read on -- after the same validation its owner performs (see LoadEdges)
      if (!skip_bad && source.IsAttributed()) {
        std::vector<int64_t> ti;
        std::vector<float> tf;
        std::vector<std::string> ts_;
        bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &ti, &tf, &ts_);
        if (!bad.ok()) {
          *why = bad;
          return bad.code();
        }
      }
      return error::OUT_OF_RANGE;
    } else if (source.IsAttributed()) bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &out->i_attrs, &out->f_attrs, &out->s_attrs);
    if (!bad.ok()) {
      if (skip_bad) return error::OUT_OF_RANGE;
      *why = bad;
      return bad.code();
    }
    out->id.push_back(id);
    if (source.IsWeighted()) out->weight.push_back(w);
    if (source.IsLabeled()) out->label.push_back((int32_t)lab);
    if (source.IsTimestamped()) out->timestamp.push_back(ts);
    return error::OK;
  };
  std::vector<NodeColumns> parts;
  s = ParseChunks<NodeColumns>(&f, LoaderThreads(f.data.size()), parse_one, &parts);
  if (!s.ok()) return s;
  Noder* noder = store->GetNoder(source.id_type);
  for (NodeColumns& part : parts) {
    s = noder->AppendColumns(info, &part);
    if (!s.ok()) return s;
  }
  return Status::OK();
}

}  // namespace io
}  // namespace graphlearn
