// Local-file TSV loader (see data_source.h).  A file is read whole, cut into
// line-aligned chunks that are parsed by a few threads, and appended to the type's
// storage in file order -- so edge ids (= load order, memory_edge_storage.cc:53-57)
// are deterministic, which the reference's thread-interleaved loaders
// (graph_store.cc:76-93) do not guarantee.
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

template <class Value, class ParseOne>
Status ParseChunks(File* f, int threads, const ParseOne& parse_one, std::vector<std::vector<Value>>* out) {
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
  out->assign(parts, {});
  std::vector<Status> st(parts);
  auto work = [&](int part) {
    char* p = base + cut[part];
    char* stop = base + cut[part + 1];
    while (p < stop) {
      char* eol = static_cast<char*>(std::memchr(p, '\n', (size_t)(stop - p)));
      char* next = eol ? eol + 1 : stop;
      char* end = eol ? eol : stop;
      if (end > p && end[-1] == '\r') --end;
      if (end > p) {
        *end = '\0';
        Value v;
        Status s = parse_one(p, end, &v);
        if (s.ok()) {
          (*out)[part].push_back(std::move(v));
        } else if (s.code() != error::OUT_OF_RANGE) {  // OUT_OF_RANGE = "skip this record"
          st[part] = s;
          return;
        }
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
  std::vector<std::string> tok;
  if (len > 0) {
    size_t start = 0;
    for (size_t i = 0; i <= len; ++i) {
      if (i == len || info.delimiter.find(data[i]) != std::string::npos) {
        tok.emplace_back(data + start, i - start);
        start = i + 1;
      }
    }
  }
  if (tok.size() != info.types.size()) return error::InvalidArgument("Unexpected attribute count");
  char msg[256];
  for (size_t i = 0; i < tok.size(); ++i) {
    const DataType t = info.types[i];
    const char* kind = nullptr;
    if (t == kInt32 || t == kInt64) {
      int64_t v = 0;
      if (t == kInt32 ? ToInt32(tok[i].c_str(), &v) : ToInt64(tok[i].c_str(), &v)) ints->push_back(v);
      else kind = t == kInt32 ? "int" : "int64";
    } else if (t == kFloat || t == kDouble) {
      float v = 0.f;
      if (t == kFloat) {
        if (ToFloat(tok[i].c_str(), &v)) floats->push_back(v);
        else kind = "float";
      } else {
        char* end = nullptr;
        const double d = strtod(tok[i].c_str(), &end);
        if (OnlyBlanksLeft(end)) floats->push_back(static_cast<float>(d));
        else kind = "double";
      }
    } else if (t == kString) {
      if (!info.hash_buckets.empty() && info.hash_buckets[i] > 0) {
        ints->push_back((int64_t)(Hash64(tok[i].data(), tok[i].size()) % (uint64_t)info.hash_buckets[i]));
      } else {
        strings->push_back(std::move(tok[i]));
      }
    }
    if (kind) {
      std::snprintf(msg, sizeof(msg), "The %dth attribute expect an %s, but got \"%s\".", (int)i, kind, tok[i].c_str());
      return error::InvalidArgument(msg);
    }
  }
  return Status::OK();
}

Status LoadEdges(const EdgeSource& source, GraphStore* store) {
  if (source.src_id_type.empty() || source.dst_id_type.empty() || source.edge_type.empty()) {
    return error::InvalidArgument("Node and edge types must be assigned.");
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
  auto parse_one = [&](char* begin, char* end, EdgeValue* v) -> Status {
    Fields fld;
    SplitTabs(begin, end, &fld);
    Status bad;
    if (fld.n != columns) {
      bad = error::InvalidArgument("Invalid edge record in " + source.path);
    } else {
      int c = 0;
      float w = 0.f;
      int64_t lab = 0;
      bool ok = ToInt64(fld.ptr[c++], &v->src_id) && ToInt64(fld.ptr[c++], &v->dst_id);
      if (ok && source.IsWeighted()) ok = ToFloat(fld.ptr[c++], &w);
      if (ok && source.IsLabeled()) ok = ToInt64(fld.ptr[c++], &lab);
      if (ok && source.IsTimestamped()) ok = ToInt64(fld.ptr[c++], &v->timestamp);
      v->weight = w;
      v->label = (int32_t)lab;
      if (!ok) bad = error::InvalidArgument("Invalid edge record in " + source.path);
      else if (source.IsAttributed()) bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &v->i_attrs, &v->f_attrs, &v->s_attrs);
    }
    if (source.direction == kReversed) std::swap(v->src_id, v->dst_id);
    if (!bad.ok() && skip_bad) return Status(error::OUT_OF_RANGE, "skipped");  // edge_loader.cc:70-78
    return bad;
  };
  std::vector<std::vector<EdgeValue>> parts;
  s = ParseChunks<EdgeValue>(&f, LoaderThreads(f.data.size()), parse_one, &parts);
  if (!s.ok()) return s;
  Graph* graph = store->GetGraph(source.edge_type);
  for (auto& part : parts) {
    UpdateEdgesRequest req(&info, (int32_t)part.size());
    for (auto& v : part) req.Append(&v);
    UpdateEdgesResponse res;
    s = graph->UpdateEdges(&req, &res);
    if (!s.ok()) return s;
  }
  return Status::OK();
}

Status LoadNodes(const NodeSource& source, GraphStore* store) {
  if (source.id_type.empty()) return error::InvalidArgument("Node type must be assigned.");
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
  auto parse_one = [&](char* begin, char* end, NodeValue* v) -> Status {
    Fields fld;
    SplitTabs(begin, end, &fld);
    Status bad;
    if (fld.n != columns) {
      bad = error::InvalidArgument("Invalid node record in " + source.path);
    } else {
      int c = 0;
      int64_t lab = 0;
      bool ok = ToInt64(fld.ptr[c++], &v->id);
      if (ok && source.IsWeighted()) ok = ToFloat(fld.ptr[c++], &v->weight);
      if (ok && source.IsLabeled()) ok = ToInt64(fld.ptr[c++], &lab);
      if (ok && source.IsTimestamped()) ok = ToInt64(fld.ptr[c++], &v->timestamp);
      v->label = (int32_t)lab;
      if (!ok) bad = error::InvalidArgument("Invalid node record in " + source.path);
      else if (source.IsAttributed()) bad = ParseAttribute(fld.ptr[c], fld.len[c], source.attr_info, &v->i_attrs, &v->attrs, &v->s_attrs);
    }
    if (!bad.ok() && skip_bad) return Status(error::OUT_OF_RANGE, "skipped");
    return bad;
  };
  std::vector<std::vector<NodeValue>> parts;
  s = ParseChunks<NodeValue>(&f, LoaderThreads(f.data.size()), parse_one, &parts);
  if (!s.ok()) return s;
  Noder* noder = store->GetNoder(source.id_type);
  for (auto& part : parts) {
    UpdateNodesRequest req(&info, (int32_t)part.size());
    for (auto& v : part) req.Append(&v);
    UpdateNodesResponse res;
    s = noder->UpdateNodes(&req, &res);
    if (!s.ok()) return s;
  }
  return Status::OK();
}

}  // namespace io
}  // namespace graphlearn
