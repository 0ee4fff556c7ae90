// The one exchange step of the sharded path behind the C ABI: an all-gather of the variable-length result blocks
// (PolygonScores per frame, labels per crop) over RCCL - xGMI inside a node.  The reference has no distributed code
// (SURVEY.md 2.3); this is the multi-GPU layer north_star asks for, for a host that is not Python: one process (or
// thread) per GPU creates an ocr_comm_t from a 128-byte id that rank 0 generated and handed to the others through
// whatever channel the host already has (file, environment, its own launcher).
//
// Frames and crops are independent, so there is NO data-path collective; the payload here is KB-scale and the
// collective is latency-bound: a fixed-size header all-gather (bytes per rank), then one payload all-gather padded to
// the largest rank.  RCCL is loaded lazily (dlopen of librccl.so.1) by the first ocr_comm_* call, so the library
// carries no link-time dependency on it.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>

#include "api_internal.hpp"

namespace ocr {
namespace {

// the slice of rccl.h this file needs (types are ABI-stable across NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclInt8 = 0 };

struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int*) = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) {
      err = std::string("cannot load RCCL: ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.so, n);
      if (!p && err.empty()) err = std::string("RCCL lacks ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
  });
  if (!err.empty()) fail(OCR_ERR_INTERNAL, "%s", err.c_str());
  return r;
}

#define OCR_RCCL(call)                                                                                   \
  do {                                                                                                   \
    const int e__ = (call);                                                                              \
    if (e__ != ncclSuccess) ::ocr::fail(OCR_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(e__)); \
  } while (0)

}  // namespace

// ---- wire format of one rank's shard (host side; also used by the CPU tests through ocr_test_comm_assemble) --------
//   int32 n_images, n_polygons, n_vertices, 0 | int32 polygons_per_image[n_images] | int32 vertices_per_polygon[n_polygons]
//   | (pad to 8) f64 scores[n_polygons] | u32 xy[2 n_vertices]
std::vector<uint8_t> pack_shard(const ocr_polygons_t& p) {
  if (p.n_images < 0 || p.n_polygons < 0 || p.n_vertices < 0) fail(OCR_ERR_INVALID, "comm: negative counts in the local block");
  const size_t ints = 4 + (size_t)p.n_images + p.n_polygons;
  const size_t o_sc = (ints * 4 + 7) / 8 * 8, o_xy = o_sc + (size_t)p.n_polygons * 8;
  std::vector<uint8_t> b(o_xy + (size_t)p.n_vertices * 8, 0);
  int32_t* h = reinterpret_cast<int32_t*>(b.data());
  h[0] = p.n_images;
  h[1] = p.n_polygons;
  h[2] = p.n_vertices;
  for (int i = 0; i < p.n_images; ++i) h[4 + i] = p.img_offsets[i + 1] - p.img_offsets[i];
  for (int k = 0; k < p.n_polygons; ++k) h[4 + p.n_images + k] = p.poly_offsets[k + 1] - p.poly_offsets[k];
  if (p.n_polygons) std::memcpy(b.data() + o_sc, p.scores, (size_t)p.n_polygons * 8);
  if (p.n_vertices) std::memcpy(b.data() + o_xy, p.xy, (size_t)p.n_vertices * 8);
  return b;
}

// shards in rank order -> one CSR block (images of rank 0, then rank 1, ...)
void assemble_shards(const uint8_t* const* shards, const size_t* bytes, int world, PolygonsOwned& out) {
  out.img_offsets.assign(1, 0);
  out.poly_offsets.assign(1, 0);
  out.xy.clear();
  out.scores.clear();
  for (int r = 0; r < world; ++r) {
    if (bytes[r] < 16) fail(OCR_ERR_INTERNAL, "comm: shard of rank %d is truncated", r);
    const int32_t* h = reinterpret_cast<const int32_t*>(shards[r]);
    const int ni = h[0], np = h[1], nv = h[2];
    const size_t ints = 4 + (size_t)ni + np, o_sc = (ints * 4 + 7) / 8 * 8, o_xy = o_sc + (size_t)np * 8;
    if (ni < 0 || np < 0 || nv < 0 || o_xy + (size_t)nv * 8 > bytes[r]) fail(OCR_ERR_INTERNAL, "comm: shard of rank %d is malformed", r);
    long long polys = 0, verts = 0;
    for (int i = 0; i < ni; ++i) {
      polys += h[4 + i];
      out.img_offsets.push_back(out.img_offsets.back() + h[4 + i]);
    }
    for (int k = 0; k < np; ++k) {
      verts += h[4 + ni + k];
      out.poly_offsets.push_back(out.poly_offsets.back() + h[4 + ni + k]);
    }
    if (polys != np || verts != nv) fail(OCR_ERR_INTERNAL, "comm: shard of rank %d has inconsistent counts", r);
    const double* sc = reinterpret_cast<const double*>(shards[r] + o_sc);
    const uint32_t* xy = reinterpret_cast<const uint32_t*>(shards[r] + o_xy);
    out.scores.insert(out.scores.end(), sc, sc + np);
    out.xy.insert(out.xy.end(), xy, xy + 2 * (size_t)nv);
  }
  out.finish();
}

class Comm {
 public:
  Comm(const uint8_t* id, int world, int rank, int device) : world_(world), rank_(rank), device_(device) {
    if (!id || world < 1 || rank < 0 || rank >= world) fail(OCR_ERR_INVALID, "comm: world %d rank %d", world, rank);
    check_device(device);
    Rccl& r = rccl();
    OCR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, 128);
    const int e = r.CommInitRank(&comm_, world, uid, rank);
    if (e != ncclSuccess) {   // a constructor that throws runs no destructor: give the stream back first
      (void)hipStreamDestroy(stream_);
      stream_ = nullptr;
      fail(OCR_ERR_HIP, "ncclCommInitRank failed: %s", r.GetErrorString(e));
    }
  }
  ~Comm() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (comm_) (void)rccl().CommDestroy(comm_);
    if (dev_) (void)hipFree(dev_);
    if (stream_) (void)hipStreamDestroy(stream_);
  }
  int world() const { return world_; }
  int rank() const { return rank_; }

  // every rank contributes `bytes` bytes (any size, also 0); returns all contributions in rank order
  void all_gather_bytes(const void* mine, size_t bytes, std::vector<std::vector<uint8_t>>& all) {
    OCR_HIP(hipSetDevice(device_));
    Rccl& r = rccl();
    // header: payload length of every rank
    reserve(16 * (size_t)(world_ + 1));
    unsigned long long len = bytes;
    OCR_HIP(hipMemcpyAsync(dev_, &len, 8, hipMemcpyHostToDevice, stream_));
    OCR_RCCL(r.AllGather(dev_, dev_ + 16, 8, ncclInt8, comm_, stream_));
    std::vector<unsigned long long> lens(world_);
    OCR_HIP(hipMemcpyAsync(lens.data(), dev_ + 16, 8 * (size_t)world_, hipMemcpyDeviceToHost, stream_));
    OCR_HIP(hipStreamSynchronize(stream_));
    size_t mx = 0;
    for (auto l : lens) mx = std::max<size_t>(mx, (size_t)l);
    mx = (mx + 15) / 16 * 16;
    all.assign(world_, {});
    if (mx == 0) return;
    // payload, padded to the largest rank
    reserve(mx * (size_t)(world_ + 1));
    if (bytes) OCR_HIP(hipMemcpyAsync(dev_, mine, bytes, hipMemcpyHostToDevice, stream_));
    OCR_RCCL(r.AllGather(dev_, dev_ + mx, mx, ncclInt8, comm_, stream_));
    std::vector<uint8_t> host(mx * (size_t)world_);
    OCR_HIP(hipMemcpyAsync(host.data(), dev_ + mx, host.size(), hipMemcpyDeviceToHost, stream_));
    OCR_HIP(hipStreamSynchronize(stream_));
    for (int k = 0; k < world_; ++k) all[k].assign(host.begin() + k * mx, host.begin() + k * mx + (size_t)lens[k]);
  }

 private:
  void reserve(size_t bytes) {
    if (bytes <= cap_) return;
    OCR_HIP(hipStreamSynchronize(stream_));
    if (dev_) OCR_HIP(hipFree(dev_));
    dev_ = nullptr;
    cap_ = 0;
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&dev_), bytes + bytes / 2));
    cap_ = bytes + bytes / 2;
  }
  int world_, rank_, device_;
  hipStream_t stream_ = nullptr;
  ncclComm_t comm_ = nullptr;
  uint8_t* dev_ = nullptr;
  size_t cap_ = 0;
};

void comm_unique_id(uint8_t* id128) {
  if (!id128) fail(OCR_ERR_INVALID, "comm: null id buffer");
  ncclUniqueId uid;
  OCR_RCCL(rccl().GetUniqueId(&uid));
  std::memcpy(id128, uid.internal, 128);
}

int comm_rccl_version() {
  int v = 0;
  OCR_RCCL(rccl().GetVersion(&v));
  return v;
}

Comm* comm_create(const uint8_t* id, int world, int rank, int device) { return new Comm(id, world, rank, device); }
void comm_destroy(Comm* c) { delete c; }
int comm_world(const Comm* c) { return c->world(); }
int comm_rank(const Comm* c) { return c->rank(); }

void comm_all_gather_polygons(Comm* c, const ocr_polygons_t& local, PolygonsOwned& out) {
  const std::vector<uint8_t> mine = pack_shard(local);
  std::vector<std::vector<uint8_t>> all;
  c->all_gather_bytes(mine.data(), mine.size(), all);
  std::vector<const uint8_t*> ptr(all.size());
  std::vector<size_t> len(all.size());
  for (size_t k = 0; k < all.size(); ++k) {
    ptr[k] = all[k].data();
    len[k] = all[k].size();
  }
  assemble_shards(ptr.data(), len.data(), (int)all.size(), out);
}

void comm_all_gather_labels(Comm* c, const int32_t* labels, int n_local, std::vector<int32_t>& all, std::vector<int32_t>& counts) {
  if (n_local < 0 || (n_local > 0 && !labels)) fail(OCR_ERR_INVALID, "comm: bad label block");
  std::vector<std::vector<uint8_t>> parts;
  c->all_gather_bytes(labels, (size_t)n_local * 4, parts);
  all.clear();
  counts.clear();
  for (const auto& p : parts) {
    counts.push_back((int32_t)(p.size() / 4));
    const int32_t* v = reinterpret_cast<const int32_t*>(p.data());
    all.insert(all.end(), v, v + p.size() / 4);
  }
}

}  // namespace ocr
