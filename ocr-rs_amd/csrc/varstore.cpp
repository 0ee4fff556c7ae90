// Reader of tch `VarStore::save` files - the replacement of `vs.load(file)`
//   /root/reference/src/text_detection/mod.rs:41-44, /root/reference/src/char_recognition/mod.rs:46, utils.rs:55-63
// without Python or libtorch.
//
// tch 0.3.0 saves a VarStore through torch-sys' at_save_multi: one torch::serialize::OutputArchive, every variable
// written under its dotted VarStore path, save_to(file).  That is a ZIP container (entries STORED, 64-byte aligned)
// holding `<stem>/data.pkl` - a protocol-2 pickle of one object whose state is a dict name -> tensor - and the raw
// little-endian storages `<stem>/data/<key>`.  The pickle uses a small opcode subset (GLOBAL, NEWOBJ, BINPERSID for
// the storages, REDUCE of torch._utils._rebuild_tensor_v2 / _rebuild_parameter / collections.OrderedDict, ...); this
// file implements exactly that subset and fails loudly on anything else.
//
// The named tensors are re-packed into the OCRW blob the engines consume (layout: ocr-rs_amd/weights.py).  For the
// recogniser the names are the ones tch really writes: Net::new puts its four layers on ONE nn::Path
// (char_recognition/model.rs:13-24), so "weight" / "bias" collide and tch de-duplicates them as name__N; the eight
// shapes are all different, so they are mapped onto conv1 / conv2 / fc1 / fc2 by shape.
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace ocr {
namespace {

struct Bytes {
  const uint8_t* p = nullptr;
  size_t n = 0;
};

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// ---- ZIP: central directory -> name -> stored bytes -------------------------------------------------------------
struct ZipEntry {
  uint16_t method = 0;
  uint64_t comp_size = 0, size = 0, local_off = 0;
};

class Zip {
 public:
  explicit Zip(const std::vector<uint8_t>& f) : f_(f) {
    const size_t n = f.size();
    if (n < 22) fail(OCR_ERR_WEIGHTS, "varstore: file of %zu bytes is not a zip archive", n);
    size_t eocd = (size_t)-1;
    for (size_t i = n - 22;; --i) {  // the end-of-central-directory record, searched from the back (comment <= 64 KiB)
      if (rd32(&f[i]) == 0x06054b50u) {
        eocd = i;
        break;
      }
      if (i == 0 || n - i > 22 + 65536) break;
    }
    if (eocd == (size_t)-1) fail(OCR_ERR_WEIGHTS, "varstore: no zip end-of-central-directory record (not a VarStore archive?)");
    uint64_t count = rd16(&f[eocd + 10]), cd_size = rd32(&f[eocd + 12]), cd_off = rd32(&f[eocd + 16]);
    if (count == 0xffff || cd_size == 0xffffffffu || cd_off == 0xffffffffu) {  // zip64
      if (eocd < 20 || rd32(&f[eocd - 20]) != 0x07064b50u) fail(OCR_ERR_WEIGHTS, "varstore: zip64 locator missing");
      const uint64_t e64 = rd64(&f[eocd - 20 + 8]);
      if (e64 + 56 > n || rd32(&f[e64]) != 0x06064b50u) fail(OCR_ERR_WEIGHTS, "varstore: bad zip64 end record");
      count = rd64(&f[e64 + 32]);
      cd_size = rd64(&f[e64 + 40]);
      cd_off = rd64(&f[e64 + 48]);
    }
    if (cd_off + cd_size > n) fail(OCR_ERR_WEIGHTS, "varstore: central directory out of range");
    size_t p = cd_off;
    for (uint64_t k = 0; k < count; ++k) {
      if (p + 46 > n || rd32(&f[p]) != 0x02014b50u) fail(OCR_ERR_WEIGHTS, "varstore: bad central directory entry %llu", (unsigned long long)k);
      ZipEntry e;
      e.method = rd16(&f[p + 10]);
      e.comp_size = rd32(&f[p + 20]);
      e.size = rd32(&f[p + 24]);
      const uint16_t nl = rd16(&f[p + 28]), xl = rd16(&f[p + 30]), cl = rd16(&f[p + 32]);
      e.local_off = rd32(&f[p + 42]);
      if (p + 46 + nl + xl + cl > n) fail(OCR_ERR_WEIGHTS, "varstore: truncated central directory");
      const std::string name(reinterpret_cast<const char*>(&f[p + 46]), nl);
      // zip64 extended information (header id 1): the fields that overflowed, in this order
      size_t x = p + 46 + nl;
      const size_t xend = x + xl;
      while (x + 4 <= xend) {
        const uint16_t id = rd16(&f[x]), sz = rd16(&f[x + 2]);
        if (id == 1) {
          size_t q = x + 4;
          if (e.size == 0xffffffffu && q + 8 <= xend) { e.size = rd64(&f[q]); q += 8; }
          if (e.comp_size == 0xffffffffu && q + 8 <= xend) { e.comp_size = rd64(&f[q]); q += 8; }
          if (e.local_off == 0xffffffffu && q + 8 <= xend) { e.local_off = rd64(&f[q]); q += 8; }
        }
        x += 4 + sz;
      }
      entries_[name] = e;
      p += 46 + nl + xl + cl;
    }
  }

  const std::map<std::string, ZipEntry>& entries() const { return entries_; }

  Bytes stored(const std::string& name) const {
    auto it = entries_.find(name);
    if (it == entries_.end()) fail(OCR_ERR_WEIGHTS, "varstore: archive has no entry '%s'", name.c_str());
    const ZipEntry& e = it->second;
    if (e.method != 0) fail(OCR_ERR_WEIGHTS, "varstore: entry '%s' is compressed (method %u); libtorch stores tensors uncompressed", name.c_str(), e.method);
    if (e.local_off + 30 > f_.size() || rd32(&f_[e.local_off]) != 0x04034b50u) fail(OCR_ERR_WEIGHTS, "varstore: bad local header of '%s'", name.c_str());
    const size_t data = e.local_off + 30 + rd16(&f_[e.local_off + 26]) + rd16(&f_[e.local_off + 28]);
    if (data + e.size > f_.size()) fail(OCR_ERR_WEIGHTS, "varstore: entry '%s' runs past the end of the file", name.c_str());
    return {&f_[data], (size_t)e.size};
  }

 private:
  const std::vector<uint8_t>& f_;
  std::map<std::string, ZipEntry> entries_;
};

// ---- the pickle subset ------------------------------------------------------------------------------------------
struct Val;
using VP = std::shared_ptr<Val>;
struct Val {
  enum Kind { NONE, BOOL, INT, FLOAT, STR, GLOBAL, TUPLE, LIST, DICT, STORAGE, TENSOR, OBJECT, MARK } kind = NONE;
  long long i = 0;
  double f = 0.0;
  std::string s, s2;              // STR: s; GLOBAL: module s, name s2; STORAGE: dtype s, key s2
  std::vector<VP> items;          // TUPLE / LIST; DICT: key, value, key, value ...; OBJECT: [class, state]
  // TENSOR
  VP storage;
  long long offset = 0;
  std::vector<long long> size, stride;
};

VP mk(Val::Kind k) {
  auto v = std::make_shared<Val>();
  v->kind = k;
  return v;
}

class Unpickler {
 public:
  explicit Unpickler(Bytes b) : b_(b) {}

  VP run() {
    for (;;) {
      const uint8_t op = u8();
      switch (op) {
        case 0x80: u8(); break;                                   // PROTO
        case '.': return pop();                                   // STOP
        case 'N': push(mk(Val::NONE)); break;
        case 0x88: case 0x89: { auto v = mk(Val::BOOL); v->i = op == 0x88; push(v); break; }
        case 'K': { auto v = mk(Val::INT); v->i = u8(); push(v); break; }
        case 'M': { auto v = mk(Val::INT); v->i = rd16(take(2)); push(v); break; }
        case 'J': { auto v = mk(Val::INT); v->i = (int32_t)rd32(take(4)); push(v); break; }
        case 0x8a: {                                              // LONG1
          const int n = u8();
          if (n > 8) fail(OCR_ERR_WEIGHTS, "varstore: LONG1 of %d bytes", n);
          const uint8_t* p = take(n);
          long long x = 0;
          for (int k = 0; k < n; ++k) x |= (long long)p[k] << (8 * k);
          if (n > 0 && n < 8 && (p[n - 1] & 0x80)) x -= 1ll << (8 * n);
          auto v = mk(Val::INT);
          v->i = x;
          push(v);
          break;
        }
        case 'G': {                                               // BINFLOAT (big endian)
          const uint8_t* p = take(8);
          uint64_t u = 0;
          for (int k = 0; k < 8; ++k) u = (u << 8) | p[k];
          auto v = mk(Val::FLOAT);
          std::memcpy(&v->f, &u, 8);
          push(v);
          break;
        }
        case 'X': { const uint32_t n = rd32(take(4)); auto v = mk(Val::STR); v->s.assign(reinterpret_cast<const char*>(take(n)), n); push(v); break; }
        case 0x8c: { const uint32_t n = u8(); auto v = mk(Val::STR); v->s.assign(reinterpret_cast<const char*>(take(n)), n); push(v); break; }
        case 'c': { auto v = mk(Val::GLOBAL); v->s = line(); v->s2 = line(); push(v); break; }
        case 'q': memo_[u8()] = top(); break;                     // BINPUT
        case 'r': memo_[rd32(take(4))] = top(); break;            // LONG_BINPUT
        case 'h': push(get(u8())); break;                         // BINGET
        case 'j': push(get(rd32(take(4)))); break;                // LONG_BINGET
        case ')': push(mk(Val::TUPLE)); break;
        case '}': push(mk(Val::DICT)); break;
        case ']': push(mk(Val::LIST)); break;
        case '(': push(mk(Val::MARK)); break;
        case 't': { auto v = mk(Val::TUPLE); v->items = pop_to_mark(); push(v); break; }
        case 0x85: case 0x86: case 0x87: {                        // TUPLE1..3
          const int n = op - 0x84;
          auto v = mk(Val::TUPLE);
          v->items.resize(n);
          for (int k = n - 1; k >= 0; --k) v->items[k] = pop();
          push(v);
          break;
        }
        case 'a': { VP x = pop(); need(top(), Val::LIST, "APPEND")->items.push_back(x); break; }
        case 'e': { auto xs = pop_to_mark(); auto l = need(top(), Val::LIST, "APPENDS"); l->items.insert(l->items.end(), xs.begin(), xs.end()); break; }
        case 's': { VP v = pop(), k = pop(); auto d = need(top(), Val::DICT, "SETITEM"); d->items.push_back(k); d->items.push_back(v); break; }
        case 'u': { auto xs = pop_to_mark(); auto d = need(top(), Val::DICT, "SETITEMS"); d->items.insert(d->items.end(), xs.begin(), xs.end()); break; }
        case 0x81: {                                              // NEWOBJ: cls, args -> object
          pop();
          VP cls = pop();
          auto o = mk(Val::OBJECT);
          o->items = {cls, mk(Val::NONE)};
          push(o);
          break;
        }
        case 'b': {                                               // BUILD: object, state
          VP st = pop();
          VP o = top();
          if (o->kind == Val::OBJECT) o->items[1] = st;
          else if (o->kind == Val::DICT && st->kind == Val::DICT) o->items.insert(o->items.end(), st->items.begin(), st->items.end());
          break;
        }
        case 'Q': push(persistent(pop())); break;                 // BINPERSID
        case 'R': { VP args = pop(), fn = pop(); push(reduce(fn, args)); break; }
        default: fail(OCR_ERR_WEIGHTS, "varstore: pickle opcode 0x%02x at byte %zu is outside the subset libtorch archives use", op, pos_ - 1);
      }
    }
  }

 private:
  Bytes b_;
  size_t pos_ = 0;
  std::vector<VP> stack_;
  std::map<uint32_t, VP> memo_;

  const uint8_t* take(size_t n) {
    if (pos_ + n > b_.n) fail(OCR_ERR_WEIGHTS, "varstore: truncated pickle");
    const uint8_t* p = b_.p + pos_;
    pos_ += n;
    return p;
  }
  uint8_t u8() { return *take(1); }
  std::string line() {
    std::string s;
    for (;;) {
      const char c = (char)u8();
      if (c == '\n') return s;
      s.push_back(c);
    }
  }
  void push(VP v) { stack_.push_back(std::move(v)); }
  VP pop() {
    if (stack_.empty()) fail(OCR_ERR_WEIGHTS, "varstore: pickle stack underflow");
    VP v = stack_.back();
    stack_.pop_back();
    return v;
  }
  VP top() {
    if (stack_.empty()) fail(OCR_ERR_WEIGHTS, "varstore: pickle stack underflow");
    return stack_.back();
  }
  VP get(uint32_t k) {
    auto it = memo_.find(k);
    if (it == memo_.end()) fail(OCR_ERR_WEIGHTS, "varstore: pickle memo %u missing", k);
    return it->second;
  }
  Val* need(const VP& v, Val::Kind k, const char* what) {
    if (v->kind != k) fail(OCR_ERR_WEIGHTS, "varstore: pickle %s on the wrong kind of object", what);
    return v.get();
  }
  std::vector<VP> pop_to_mark() {
    size_t m = stack_.size();
    while (m > 0 && stack_[m - 1]->kind != Val::MARK) --m;
    if (m == 0) fail(OCR_ERR_WEIGHTS, "varstore: pickle MARK missing");
    std::vector<VP> xs(stack_.begin() + m, stack_.end());
    stack_.resize(m - 1);
    return xs;
  }
  static std::vector<long long> ints(const VP& t) {
    std::vector<long long> v;
    if (t->kind != Val::TUPLE && t->kind != Val::LIST) fail(OCR_ERR_WEIGHTS, "varstore: tensor size / stride is not a tuple");
    for (const VP& x : t->items) {
      if (x->kind != Val::INT) fail(OCR_ERR_WEIGHTS, "varstore: non-integer tensor extent");
      v.push_back(x->i);
    }
    return v;
  }
  // ('storage', torch.FloatStorage, key, device, numel)
  VP persistent(const VP& id) {
    if (id->kind != Val::TUPLE || id->items.size() < 5 || id->items[0]->kind != Val::STR || id->items[0]->s != "storage" ||
        id->items[1]->kind != Val::GLOBAL || id->items[2]->kind != Val::STR || id->items[4]->kind != Val::INT)
      fail(OCR_ERR_WEIGHTS, "varstore: unexpected persistent id in the pickle");
    auto s = mk(Val::STORAGE);
    s->s = id->items[1]->s2;  // FloatStorage, DoubleStorage, ...
    s->s2 = id->items[2]->s;
    s->i = id->items[4]->i;
    return s;
  }
  VP reduce(const VP& fn, const VP& args) {
    if (fn->kind != Val::GLOBAL || args->kind != Val::TUPLE) fail(OCR_ERR_WEIGHTS, "varstore: REDUCE of a non-global");
    const std::string name = fn->s + "." + fn->s2;
    if (name == "collections.OrderedDict") return mk(Val::DICT);
    if (name == "torch._utils._rebuild_tensor_v2" || name == "torch._utils._rebuild_tensor") {
      if (args->items.size() < 4 || args->items[0]->kind != Val::STORAGE || args->items[1]->kind != Val::INT)
        fail(OCR_ERR_WEIGHTS, "varstore: malformed _rebuild_tensor arguments");
      auto t = mk(Val::TENSOR);
      t->storage = args->items[0];
      t->offset = args->items[1]->i;
      t->size = ints(args->items[2]);
      t->stride = ints(args->items[3]);
      if (t->size.size() != t->stride.size()) fail(OCR_ERR_WEIGHTS, "varstore: tensor size / stride rank mismatch");
      return t;
    }
    if (name == "torch._utils._rebuild_parameter") {
      if (args->items.empty() || args->items[0]->kind != Val::TENSOR) fail(OCR_ERR_WEIGHTS, "varstore: malformed _rebuild_parameter arguments");
      return args->items[0];
    }
    auto o = mk(Val::OBJECT);  // anything else stays opaque (it cannot be a tensor)
    o->items = {fn, args};
    return o;
  }
};

float half_to_float(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
  uint32_t u;
  if (exp == 0) {
    if (man == 0) u = sign;
    else {  // subnormal
      int e = -1;
      uint32_t m = man;
      do { ++e; m <<= 1; } while (!(m & 0x400u));
      u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
    }
  } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
  else u = sign | ((exp + 112) << 23) | (man << 13);
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

void collect(const VP& v, const std::string& prefix, std::vector<std::pair<std::string, VP>>& out) {
  if (!v) return;
  if (v->kind == Val::OBJECT) {
    collect(v->items[1], prefix, out);
  } else if (v->kind == Val::DICT) {
    for (size_t k = 0; k + 1 < v->items.size(); k += 2) {
      if (v->items[k]->kind != Val::STR) continue;
      const std::string name = prefix.empty() ? v->items[k]->s : prefix + "." + v->items[k]->s;
      const VP& x = v->items[k + 1];
      if (x->kind == Val::TENSOR) out.emplace_back(name, x);
      else if (x->kind == Val::OBJECT || x->kind == Val::DICT) collect(x, name, out);  // nested modules (not written by tch, harmless)
    }
  }
}

std::vector<uint8_t> read_file(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) fail(OCR_ERR_WEIGHTS, "varstore: cannot open '%s'", path);
  std::vector<uint8_t> buf;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  if (n < 0) {
    std::fclose(f);
    fail(OCR_ERR_WEIGHTS, "varstore: cannot size '%s'", path);
  }
  buf.resize((size_t)n);
  const size_t got = n ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  if (got != (size_t)n) fail(OCR_ERR_WEIGHTS, "varstore: short read of '%s'", path);
  return buf;
}

struct Spec {
  const char* name;
  int ndim;
  int dims[4];
};
const Spec kRecSpecs[8] = {{"conv1.weight", 4, {32, 1, 5, 5}}, {"conv1.bias", 1, {32, 1, 1, 1}}, {"conv2.weight", 4, {64, 32, 5, 5}},
                           {"conv2.bias", 1, {64, 1, 1, 1}},   {"fc1.weight", 2, {512, 1024, 1, 1}}, {"fc1.bias", 1, {512, 1, 1, 1}},
                           {"fc2.weight", 2, {62, 512, 1, 1}}, {"fc2.bias", 1, {62, 1, 1, 1}}};

}  // namespace

// Reads every tensor of a VarStore archive as f32, row-major (strides resolved).
std::vector<NamedTensor> read_varstore(const char* path) {
  if (!path) fail(OCR_ERR_INVALID, "varstore: null path");
  const std::vector<uint8_t> file = read_file(path);
  const Zip zip(file);
  std::string pkl_name;
  for (const auto& kv : zip.entries()) {
    const std::string& n = kv.first;
    if (n.size() >= 8 && n.compare(n.size() - 8, 8, "data.pkl") == 0 && (n.size() == 8 || n[n.size() - 9] == '/') &&
        (pkl_name.empty() || n.size() < pkl_name.size()))
      pkl_name = n;
  }
  if (pkl_name.empty()) fail(OCR_ERR_WEIGHTS, "varstore: '%s' holds no data.pkl (not a libtorch archive)", path);
  const std::string dir = pkl_name.substr(0, pkl_name.size() - 8);  // "<stem>/"
  const VP root = Unpickler(zip.stored(pkl_name)).run();
  std::vector<std::pair<std::string, VP>> found;
  collect(root, "", found);
  if (found.empty()) fail(OCR_ERR_WEIGHTS, "varstore: '%s' holds no named tensors", path);
  std::vector<NamedTensor> out;
  for (auto& kv : found) {
    const Val& t = *kv.second;
    const Val& st = *t.storage;
    int esize = 0;
    if (st.s == "FloatStorage") esize = 4;
    else if (st.s == "DoubleStorage") esize = 8;
    else if (st.s == "HalfStorage" || st.s == "BFloat16Storage") esize = 2;
    else fail(OCR_ERR_WEIGHTS, "varstore: tensor '%s' has storage type %s (only floating point weights are expected)", kv.first.c_str(), st.s.c_str());
    std::string entry = dir + "data/" + st.s2;
    if (!zip.entries().count(entry)) entry = dir + "tensors/" + st.s2;  // libtorch <= 1.5 layout
    const Bytes raw = zip.stored(entry);
    if (t.size.size() > 4) fail(OCR_ERR_WEIGHTS, "varstore: tensor '%s' has rank %zu", kv.first.c_str(), t.size.size());
    NamedTensor nt;
    nt.name = kv.first;
    size_t count = 1;
    for (long long d : t.size) {
      if (d < 0 || d > (1ll << 31)) fail(OCR_ERR_WEIGHTS, "varstore: tensor '%s' has a bad extent", kv.first.c_str());
      nt.dims.push_back((int)d);
      count *= (size_t)d;
    }
    nt.data.resize(count);
    const int r = (int)t.size.size();
    long long idx[4] = {0, 0, 0, 0};
    for (size_t k = 0; k < count; ++k) {
      long long e = t.offset;
      for (int d = 0; d < r; ++d) e += idx[d] * t.stride[d];
      if (e < 0 || (size_t)(e + 1) * esize > raw.n) fail(OCR_ERR_WEIGHTS, "varstore: tensor '%s' reads past its storage", kv.first.c_str());
      const uint8_t* p = raw.p + (size_t)e * esize;
      float v;
      if (esize == 4) std::memcpy(&v, p, 4);
      else if (esize == 8) {
        double dv;
        std::memcpy(&dv, p, 8);
        v = (float)dv;
      } else if (st.s == "HalfStorage") v = half_to_float(rd16(p));
      else {
        const uint32_t u = (uint32_t)rd16(p) << 16;
        std::memcpy(&v, &u, 4);
      }
      nt.data[k] = v;
      for (int d = r - 1; d >= 0; --d) {
        if (++idx[d] < t.size[d]) break;
        idx[d] = 0;
      }
    }
    out.push_back(std::move(nt));
  }
  return out;
}

// tch's de-duplicated names of the recogniser (weight, bias, weight__2, ...) -> conv1 / conv2 / fc1 / fc2, by shape
void rename_tch_rec(std::vector<NamedTensor>& ts) {
  bool dotted = true;
  for (const Spec& s : kRecSpecs) {
    bool have = false;
    for (const NamedTensor& t : ts) have = have || t.name == s.name;
    dotted = dotted && have;
  }
  if (dotted) return;
  bool used[8] = {};
  for (NamedTensor& t : ts) {
    const std::string base = t.name.substr(0, t.name.find("__"));
    if (base != "weight" && base != "bias") continue;
    for (int k = 0; k < 8; ++k) {
      const Spec& s = kRecSpecs[k];
      const std::string leaf = std::string(s.name).substr(std::string(s.name).find('.') + 1);
      if (used[k] || leaf != base || (int)t.dims.size() != s.ndim) continue;
      bool same = true;
      for (int d = 0; d < s.ndim; ++d) same = same && t.dims[d] == s.dims[d];
      if (same) {
        t.name = s.name;
        used[k] = true;
        break;
      }
    }
  }
}

// OCRW v1 blob (ocr-rs_amd/weights.py): header, directory of 100-byte records, 64-byte aligned f32 payloads
std::vector<uint8_t> pack_ocrw(const std::vector<NamedTensor>& ts) {
  const size_t rec = 64 + 4 + 16 + 8 + 8;
  size_t off = (16 + rec * ts.size() + 63) / 64 * 64;
  std::vector<uint64_t> offs;
  for (const NamedTensor& t : ts) {
    offs.push_back(off);
    off = (off + t.data.size() * 4 + 63) / 64 * 64;
  }
  std::vector<uint8_t> b(off, 0);
  std::memcpy(&b[0], "OCRW", 4);
  const uint32_t ver = 1, n = (uint32_t)ts.size();
  std::memcpy(&b[4], &ver, 4);
  std::memcpy(&b[8], &n, 4);
  for (size_t i = 0; i < ts.size(); ++i) {
    uint8_t* r = &b[16 + i * rec];
    if (ts[i].name.size() > 63) fail(OCR_ERR_WEIGHTS, "varstore: tensor name '%s' is longer than 63 bytes", ts[i].name.c_str());
    std::memcpy(r, ts[i].name.data(), ts[i].name.size());
    const uint32_t nd = (uint32_t)ts[i].dims.size();
    uint32_t dims[4] = {1, 1, 1, 1};
    for (uint32_t d = 0; d < nd; ++d) dims[d] = (uint32_t)ts[i].dims[d];
    const uint64_t cnt = ts[i].data.size();
    std::memcpy(r + 64, &nd, 4);
    std::memcpy(r + 68, dims, 16);
    std::memcpy(r + 84, &offs[i], 8);
    std::memcpy(r + 92, &cnt, 8);
    if (cnt) std::memcpy(&b[offs[i]], ts[i].data.data(), cnt * 4);
  }
  return b;
}

std::vector<uint8_t> varstore_to_blob(const char* path, int kind) {
  std::vector<NamedTensor> ts = read_varstore(path);
  if (kind == 2) rename_tch_rec(ts);
  return pack_ocrw(ts);
}

}  // namespace ocr
