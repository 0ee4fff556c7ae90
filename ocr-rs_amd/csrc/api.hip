// extern "C" boundary (include/ocr_amd.h).  Nothing throws across it.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>


#include "api_internal.hpp"
#include "thread_pool.hpp"

namespace ocr {
thread_local std::string g_last_error;
}

namespace {
using ocr::guard;
using ocr::PolygonsOwned;
using ocr::align256;

// device contours (contours.hip): where the pieces live inside one scratch slot, and the launches that fill them
struct ContourBuffers {
  static constexpr int CAP = 1 << 15, MAXC = 4096;   // points / contours per image (a dense page: 12 k / 60)
  char* base = nullptr;
  size_t o_bits = 0, o_pts = 0, o_st = 0, o_hdr = 0, o_pk = 0, o_ln = 0, o_sp = 0, total = 0, wpi = 0;
  ContourBuffers(int n, size_t hw) {
    wpi = ocr::binarize_pack_words(hw);
    o_pts = o_bits + align256((size_t)n * wpi * 4);
    o_st = o_pts + align256((size_t)n * CAP * 4);
    o_hdr = o_st + align256((size_t)n * (MAXC + 1) * 4);
    o_pk = o_hdr + align256((size_t)n * 16);
    o_ln = o_pk + align256((size_t)n * CAP * 4);
    o_sp = o_ln + align256((size_t)n * MAXC * 4);
    total = o_sp + align256(ocr::contour_spec_bytes(n));
  }
  uint32_t* bits() const { return reinterpret_cast<uint32_t*>(base + o_bits); }
};
// binarize + pack + trace of a batch whose map is (or will be, stream order) on the device: everything on `s`, nothing waits
static ContourBuffers enqueue_contours(ocr::Detector& det, int slot, const float* prob_dev, int n, int h, int w, float thresh, hipStream_t s) {
  using namespace ocr;
  ContourBuffers cb(n, (size_t)h * w);
  cb.base = static_cast<char*>(det.scratch(slot, cb.total));
  launch_binarize_pack(prob_dev, cb.bits(), thresh, n, (size_t)h * w, s);
  launch_contour_trace(cb.bits(), cb.wpi, n, h, w, reinterpret_cast<uint32_t*>(cb.base + cb.o_pts), ContourBuffers::CAP, reinterpret_cast<int*>(cb.base + cb.o_st),
                       ContourBuffers::MAXC, reinterpret_cast<int*>(cb.base + cb.o_hdr), reinterpret_cast<uint32_t*>(cb.base + cb.o_pk),
                       reinterpret_cast<int*>(cb.base + cb.o_ln), cb.base + cb.o_sp, det.device_contours() == 2, s);
  return cb;
}

// the polygon chain behind the device tracer (candidates.hip, box_score.hip, unclip.hip): where its pieces live inside one scratch slot
struct ChainBuffers {
  char* sc = nullptr;
  int max_jobs = 0, max_pts = 0;
  size_t o_jobs = 0, o_pts = 0, o_sum = 0, o_cnt = 0, o_adj = 0, o_st = 0, o_len = 0, o_oxy = 0, o_work = 0, o_tot = 0, o_hd = 0, o_cs = 0, total = 0;
  explicit ChainBuffers(int n) {
    max_jobs = n * 1024;   // a dense page: 60 - 130 candidates of 4 - 16 points; a batch that needs more takes the host path
    max_pts = n * 8192;
    o_pts = o_jobs + align256((size_t)max_jobs * sizeof(ocr::BoxScoreJob));
    o_sum = o_pts + align256((size_t)max_pts * 8);
    o_cnt = o_sum + align256((size_t)max_jobs * 8);
    o_adj = o_cnt + align256((size_t)max_jobs * 8);
    o_st = o_adj + align256((size_t)n * 16);
    o_len = o_st + align256((size_t)max_jobs * 4);
    o_oxy = o_len + align256((size_t)max_jobs * 4);
    o_work = o_oxy + align256(3 * (size_t)max_pts * 8);
    o_tot = o_work + align256(ocr::unclip_work_bytes((size_t)max_pts, max_jobs));
    o_hd = o_tot + align256((size_t)n * 8);
    o_cs = o_hd + 256;
    total = o_cs + ocr::candidates_scratch_bytes(n, ContourBuffers::CAP, ContourBuffers::MAXC);
  }
  ocr::BoxScoreJob* jobs() const { return reinterpret_cast<ocr::BoxScoreJob*>(sc + o_jobs); }
  int32_t* pts() const { return reinterpret_cast<int32_t*>(sc + o_pts); }
  double* sums() const { return reinterpret_cast<double*>(sc + o_sum); }
  double* counts() const { return reinterpret_cast<double*>(sc + o_cnt); }
  int* totals() const { return reinterpret_cast<int*>(sc + o_hd); }
};
// Douglas-Peucker + job list, box scores, unclip of a batch whose contours are (or will be, stream order) in `cb`: everything on `s`
// adj: host adjust values, uploaded here on `s` - or nullptr when the caller has already put them at o_adj of the slot (upload_adj below)
static ChainBuffers enqueue_chain(ocr::Detector& det, int slot, const ContourBuffers& cb, const float* prob_dev, int n, int h, int w, const double* adj,
                                  const ocr_postproc_params_t& prm, hipStream_t s) {
  using namespace ocr;
  ChainBuffers ch(n);
  ch.sc = static_cast<char*>(det.scratch(slot, ch.total));
  const UnclipParams up{prm.box_thresh, prm.unclip_ratio, prm.min_size};
  if (adj) OCR_HIP(hipMemcpyAsync(ch.sc + ch.o_adj, adj, (size_t)n * 16, hipMemcpyHostToDevice, s));
  launch_candidates(reinterpret_cast<const int*>(cb.base + cb.o_hdr), reinterpret_cast<const uint32_t*>(cb.base + cb.o_pts), ContourBuffers::CAP,
                    reinterpret_cast<const int*>(cb.base + cb.o_st), ContourBuffers::MAXC, n, h, w, ch.sc + ch.o_cs, ch.jobs(), ch.max_jobs, ch.pts(), ch.max_pts,
                    reinterpret_cast<int*>(ch.sc + ch.o_tot), ch.totals(), s);
  launch_box_scores_counted(prob_dev, h, w, ch.jobs(), ch.pts(), ch.totals(), std::min(ch.max_jobs, 4096), ch.sums(), ch.counts(), s);
  launch_unclip(ch.jobs(), ch.pts(), ch.totals(), ch.max_jobs, (size_t)ch.max_pts, ch.sums(), ch.counts(), reinterpret_cast<const double*>(ch.sc + ch.o_adj), up,
                ch.sc + ch.o_work, reinterpret_cast<uint32_t*>(ch.sc + ch.o_oxy), reinterpret_cast<int32_t*>(ch.sc + ch.o_len),
                reinterpret_cast<int32_t*>(ch.sc + ch.o_st), s);
  return ch;
}

// device_contours in the pipelined calls: the contours of the batch that was just queued are requested right away - behind its
// forward, on a stream of their own - so that the call which brings its polygons back finds them done instead of waiting
static void pretrace_pending(ocr::Detector& d) {
  using namespace ocr;
  if (!d.has_pending() || !d.device_contours()) return;
  Detector::Pending& p = d.pending();
  if (!contour_trace_fits(p.h, p.w)) return;
  hipStream_t ts = d.trace_stream();   // not the post-processing stream: crops of the batch that just came back must not queue behind this forward
  const bool chain = d.device_polygons() && d.device_unclip() && p.h == p.w;
  if (chain) {
    // The adjust values do not depend on the forward: they go up FIRST, while the trace stream is idle (the batch that used scratch
    // slot 4 before has been collected), from a pinned block.  Queued behind the wait for the forward from pageable memory the copy is
    // staged and awaited on the host - this call would not return before forward k and its trace had finished, and the caller could
    // not queue forward k + 1 behind forward k.
    ChainBuffers ch(p.n);
    ch.sc = static_cast<char*>(d.scratch(4, ch.total));
    double* pin = static_cast<double*>(d.host_adj((size_t)p.n * 16));
    std::memcpy(pin, p.adj.data(), (size_t)p.n * 16);
    OCR_HIP(hipMemcpyAsync(ch.sc + ch.o_adj, pin, (size_t)p.n * 16, hipMemcpyHostToDevice, ts));
  }
  OCR_HIP(hipStreamWaitEvent(ts, p.event, 0));
  const ContourBuffers cb = enqueue_contours(d, 3, p.prob, p.n, p.h, p.w, (float)p.params.thresh, ts);
  if (chain) {   // ... and the rest of the chain behind them: the call that comes back only collects
    enqueue_chain(d, 4, cb, p.prob, p.n, p.h, p.w, nullptr, p.params, ts);
    p.prechained = true;
  }
  OCR_HIP(hipEventRecord(d.trace_done_event(), ts));
  p.pretraced = true;
}

// get_boxes_and_box_scores (metrics.rs:37-56) over the whole batch.  Dense, regular work on the GPU (binarisation into
// a packed bit image, box scores, unclip), irregular work on the detector's host thread pool, one image per task.
// With device contours AND device polygons (options device_contours, device_polygons) a square map's whole chain runs on the
// device - trace, Douglas-Peucker, job list, box scores, unclip - and the host only collects results; it still finishes the polygons
// the unclip kernel hands back (UNCLIP_HOST) and takes, from the bit image on, the images the tracer gave up.
// pretraced: the batch's contours were requested on `s` earlier (enqueue_contours into scratch slot 3): only read them
void postprocess(ocr::Detector& det, const float* prob, int n, int h, int w, int mem_kind, const double* adj,
                 const ocr_postproc_params_t& prm, ocr_polygons_t** out, hipStream_t s, bool pretraced = false, bool prechained = false) {
  using namespace ocr;
#ifdef POSTPROC_TIMING
  auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double T0 = tnow();
  double T1 = 0, T2 = 0, T3 = 0;
#endif
  if (!prob || !adj || !out) fail(OCR_ERR_INVALID, "det_postprocess: null argument");
  if (n <= 0 || h <= 0 || w <= 0) fail(OCR_ERR_INVALID, "det_postprocess: bad shape");
  OCR_HIP(hipSetDevice(det.device()));
  const size_t hw = (size_t)h * w, px = (size_t)n * hw;
  const size_t wpi = binarize_pack_words(hw);  // 32-bit words per packed image
  const bool dev_trace = pretraced || (det.device_contours() && contour_trace_fits(h, w));
  const bool dev_chain = prechained || (dev_trace && det.device_polygons() && det.device_unclip() && h == w);
  // scratch: [prob copy if host] [packed bitmaps]
  const size_t off_bits = mem_kind == OCR_MEM_HOST ? align256(px * 4) : 0;
  char* scratch = static_cast<char*>(det.scratch(0, off_bits + align256((size_t)n * wpi * 4)));
  const float* prob_dev = prob;
  if (mem_kind == OCR_MEM_HOST) {
    if (pretraced) fail(OCR_ERR_INTERNAL, "postprocess: a pretraced batch lives on the device");
    OCR_HIP(hipMemcpyAsync(scratch, prob, px * 4, hipMemcpyHostToDevice, s));
    prob_dev = reinterpret_cast<const float*>(scratch);
  }
  uint32_t* bits_dev = reinterpret_cast<uint32_t*>(scratch + off_bits);
  ThreadPool& pool = det.pool();
  struct PerImage {
    std::vector<uint32_t> xy;
    std::vector<int32_t> lens;
    std::vector<double> scores;
  };
  std::vector<PerImage> per(n);
  std::vector<std::vector<std::vector<geom::Pt>>> cands(n);
  std::vector<int> todo;            // images whose candidates the host has (or must make): box scores + unclip in the second part
  std::vector<uint32_t> bits;
  bool dev_unclip = det.device_unclip();   // (the host-built job list below decides per call: a handful of polygons per pool thread is faster on the host)
  const UnclipParams up{prm.box_thresh, prm.unclip_ratio, prm.min_size};
  // a device-settled or host-finished candidate into its image's lists
  auto take = [&](PerImage& r, int st, const uint32_t* o, int olen, const std::vector<geom::Pt>& c, double score, int b) {
    if (st == UNCLIP_KEEP) {
      r.xy.insert(r.xy.end(), o, o + 2 * (size_t)olen);
      r.lens.push_back(olen);
      r.scores.push_back(score);
    } else if (st == UNCLIP_HOST) {
      const size_t before = r.xy.size();
      if (geom::finish_polygon(c, score, adj[2 * b], adj[2 * b + 1], prm, r.xy)) {
        r.lens.push_back((int32_t)((r.xy.size() - before) / 2));
        r.scores.push_back(score);
      }
    }
  };

  if (dev_chain) {
    // ---- everything on the device (contours.hip, candidates.hip, box_score.hip, unclip.hip); ONE round trip of small headers,
    // one of results
    ContourBuffers cb(n, hw);
    if (pretraced) {   // filled when the batch was queued
      cb.base = static_cast<char*>(det.scratch(3, cb.total));
      OCR_HIP(hipStreamWaitEvent(s, det.trace_done_event(), 0));
    } else {
      cb = enqueue_contours(det, 2, prob_dev, n, h, w, (float)prm.thresh, s);
    }
    bits_dev = cb.bits();
    ChainBuffers ch(n);
    if (prechained) ch.sc = static_cast<char*>(det.scratch(4, ch.total));   // queued with the contours (pretrace_pending)
    else ch = enqueue_chain(det, 1, cb, prob_dev, n, h, w, adj, prm, s);
    char* sc = ch.sc;
    const size_t o_tot = ch.o_tot, o_st = ch.o_st, o_len = ch.o_len, o_oxy = ch.o_oxy;
    BoxScoreJob* d_jobs = ch.jobs();
    int32_t* d_pts = ch.pts();
    double *d_sum = ch.sums(), *d_cnt = ch.counts();
    int* d_totals = ch.totals();
    // both round trips land in the handle's pinned buffer: the copies are queued back to back and really asynchronous (into pageable
    // memory each of them is a blocking staged copy)
    const size_t h_tot = 0, h_totals = align256((size_t)n * 8);
    char* hb = static_cast<char*>(det.host_scratch(h_totals + 256));
    OCR_HIP(hipMemcpyAsync(hb + h_tot, sc + o_tot, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipMemcpyAsync(hb + h_totals, d_totals, 12, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
    const std::vector<int32_t> tot(reinterpret_cast<const int32_t*>(hb + h_tot), reinterpret_cast<const int32_t*>(hb + h_tot) + 2 * (size_t)n);
    int32_t totals[4] = {0, 0, 0, 0};
    std::memcpy(totals, hb + h_totals, 12);
#ifdef POSTPROC_TIMING
    T1 = T2 = tnow();
#endif
    if (totals[2] != 0) {
      for (int b = 0; b < n; ++b) todo.push_back(b);   // the lists overflowed: the host path takes the batch
    } else {
      const int tj = totals[0];
      const size_t tp = (size_t)totals[1];
      const size_t h_jobs = 0, h_sum = h_jobs + align256((size_t)tj * sizeof(BoxScoreJob)), h_cnt = h_sum + align256((size_t)tj * 8),
                   h_st = h_cnt + align256((size_t)tj * 8), h_len = h_st + align256((size_t)tj * 4), h_pts = h_len + align256((size_t)tj * 4),
                   h_oxy = h_pts + align256(tp * 8), h_end = h_oxy + align256(3 * tp * 8);
      hb = static_cast<char*>(det.host_scratch(h_end + 256));
      const BoxScoreJob* jobs = reinterpret_cast<const BoxScoreJob*>(hb + h_jobs);
      const double *sums = reinterpret_cast<const double*>(hb + h_sum), *counts = reinterpret_cast<const double*>(hb + h_cnt);
      const int32_t *ustatus = reinterpret_cast<const int32_t*>(hb + h_st), *ulen = reinterpret_cast<const int32_t*>(hb + h_len),
                    *pts = reinterpret_cast<const int32_t*>(hb + h_pts);
      const uint32_t* uxy = reinterpret_cast<const uint32_t*>(hb + h_oxy);
      if (tj > 0) {
        OCR_HIP(hipMemcpyAsync(hb + h_jobs, d_jobs, (size_t)tj * sizeof(BoxScoreJob), hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_sum, d_sum, (size_t)tj * 8, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_cnt, d_cnt, (size_t)tj * 8, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_st, sc + o_st, (size_t)tj * 4, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_len, sc + o_len, (size_t)tj * 4, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_pts, d_pts, tp * 8, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipMemcpyAsync(hb + h_oxy, sc + o_oxy, 3 * tp * 8, hipMemcpyDeviceToHost, s));
        OCR_HIP(hipStreamSynchronize(s));
      }
      std::vector<int> first_job(n + 1, 0);
      {
        int at = 0;
        for (int b = 0; b < n; ++b) {
          first_job[b] = at;
          if (tot[2 * b] > 0) at += tot[2 * b];
          if (tot[2 * b] < 0) todo.push_back(b);
        }
        first_job[n] = at;
        if (at != tj) fail(OCR_ERR_INTERNAL, "postprocess: device job list holds %d jobs, its image table %d", tj, at);
      }
      for (int b = 0; b < n; ++b)
        if (tot[2 * b] >= 0) {
          ++det.post_stats[0];
          ++det.post_stats[4];
        }
      for (int j = 0; j < tj; ++j) ++det.post_stats[ustatus[j] == UNCLIP_HOST ? 3 : 2];
      pool.parallel_for(n, [&](int b) {
        std::vector<geom::Pt> c;
        for (int j = first_job[b]; j < first_job[b + 1]; ++j) {
          const BoxScoreJob& jb = jobs[j];
          const double score = sums[j] / counts[j];
          if (ustatus[j] == UNCLIP_HOST) {
            c.resize((size_t)jb.n_pts);
            for (int i = 0; i < jb.n_pts; ++i) c[i] = {pts[2 * ((size_t)jb.pt_offset + i)], pts[2 * ((size_t)jb.pt_offset + i) + 1]};
          }
          take(per[b], ustatus[j], uxy + 6 * (size_t)jb.pt_offset, ulen[j], c, score, b);
        }
      });
    }
    if (!todo.empty()) {   // the images the device gave up: bit image to the host, host tracer + Douglas-Peucker
      bits.resize((size_t)n * wpi);
      for (int b : todo) OCR_HIP(hipMemcpyAsync(bits.data() + (size_t)b * wpi, bits_dev + (size_t)b * wpi, wpi * 4, hipMemcpyDeviceToHost, s));
      OCR_HIP(hipStreamSynchronize(s));
      pool.parallel_for((int)todo.size(), [&](int k) { geom::contour_candidates_bits(bits.data() + (size_t)todo[k] * wpi, h, w, cands[todo[k]]); });
      det.post_stats[1] += (long long)todo.size();
    }
  } else if (dev_trace) {
    // contour tracing on the device (contours.hip), Douglas-Peucker on the pool.  An image the device gives up on (buffers too
    // small - noise: thousands of contours - or a start outside the parallel form's list) takes the host tracer below.
    ContourBuffers cb(n, hw);
    if (pretraced) {   // filled when the batch was queued
      cb.base = static_cast<char*>(det.scratch(3, cb.total));
      OCR_HIP(hipStreamWaitEvent(s, det.trace_done_event(), 0));
    }
    else cb = enqueue_contours(det, 2, prob_dev, n, h, w, (float)prm.thresh, s);
    bits_dev = cb.bits();
    char* cs = cb.base;
    const size_t o_hdr = cb.o_hdr, o_pk = cb.o_pk, o_ln = cb.o_ln;
    std::vector<int32_t> hdr((size_t)n * 4);
    OCR_HIP(hipMemcpyAsync(hdr.data(), cs + o_hdr, hdr.size() * 4, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
    std::vector<size_t> p_at(n + 1, 0), c_at(n + 1, 0);
    int failed = 0;
    for (int b = 0; b < n; ++b) {
      const bool ok = hdr[4 * b + 2] == 0;
      failed += !ok;
      c_at[b + 1] = c_at[b] + (ok ? (size_t)hdr[4 * b] : 0);
      p_at[b + 1] = p_at[b] + (ok ? (size_t)hdr[4 * b + 1] : 0);
    }
    std::vector<uint32_t> cpts(p_at[n]);
    std::vector<int32_t> clens(c_at[n]);
    if (!cpts.empty()) OCR_HIP(hipMemcpyAsync(cpts.data(), cs + o_pk, cpts.size() * 4, hipMemcpyDeviceToHost, s));
    if (!clens.empty()) OCR_HIP(hipMemcpyAsync(clens.data(), cs + o_ln, clens.size() * 4, hipMemcpyDeviceToHost, s));
    if (failed) {
      bits.resize((size_t)n * wpi);
      for (int b = 0; b < n; ++b)
        if (hdr[4 * b + 2] != 0) OCR_HIP(hipMemcpyAsync(bits.data() + (size_t)b * wpi, bits_dev + (size_t)b * wpi, wpi * 4, hipMemcpyDeviceToHost, s));
    }
    OCR_HIP(hipStreamSynchronize(s));
#ifdef POSTPROC_TIMING
    T1 = tnow();
#endif
    pool.parallel_for(n, [&](int b) {
      if (hdr[4 * b + 2] == 0) geom::contour_candidates_packed(cpts.data() + p_at[b], clens.data() + c_at[b], hdr[4 * b], cands[b]);
      else geom::contour_candidates_bits(bits.data() + (size_t)b * wpi, h, w, cands[b]);
    });
    det.post_stats[0] += n - failed;
    det.post_stats[1] += failed;
    for (int b = 0; b < n; ++b) todo.push_back(b);
  } else {
    launch_binarize_pack(prob_dev, bits_dev, (float)prm.thresh, n, hw, s);  // metrics.rs:41,129
    bits.resize((size_t)n * wpi);
    OCR_HIP(hipMemcpyAsync(bits.data(), bits_dev, bits.size() * 4, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
#ifdef POSTPROC_TIMING
    T1 = tnow();
#endif
    // contour tracing + Douglas-Peucker (metrics.rs:78-98)
    pool.parallel_for(n, [&](int b) { geom::contour_candidates_bits(bits.data() + (size_t)b * wpi, h, w, cands[b]); });
    for (int b = 0; b < n; ++b) todo.push_back(b);
    det.post_stats[1] += n;
  }
#ifdef POSTPROC_TIMING
  if (!dev_chain) T2 = tnow();
#endif

  // ---- the images whose candidates are on the host: box scores on the GPU (metrics.rs:99 -> :150-184), unclip behind them
  std::vector<BoxScoreJob> jobs;
  std::vector<int32_t> pts;
  std::vector<int> first_job(todo.size() + 1, 0);
  for (size_t k = 0; k < todo.size(); ++k) {
    const int b = todo[k];
    first_job[k] = (int)jobs.size();
    for (const auto& c : cands[b]) {
      if ((int)c.size() > kBoxScoreMaxPts) fail(OCR_ERR_INVALID, "polygon with %zu vertices exceeds %d", c.size(), kBoxScoreMaxPts);
      int mnx = INT32_MAX, mxx = 0, mny = INT32_MAX, mxy = 0;
      for (const auto& p : c) {
        mnx = std::min(mnx, p.x);
        mxx = std::max(mxx, p.x);
        mny = std::min(mny, p.y);
        mxy = std::max(mxy, p.y);
      }
      // the reference clamps x by size[-2] (=H) and y by size[-1] (=W): metrics.rs:151-166
      const int cw = h, ch = w;
      mnx = std::clamp(mnx, 0, cw - 1);
      mxx = std::clamp(mxx, 0, cw - 1);
      mny = std::clamp(mny, 0, ch - 1);
      mxy = std::clamp(mxy, 0, ch - 1);
      if (mxx >= w || mxy >= h) fail(OCR_ERR_INVALID, "non-square map: box (%d,%d) leaves the %dx%d map (the reference would fail in narrow())", mxx, mxy, w, h);
      BoxScoreJob j{b, (int)(pts.size() / 2), (int)c.size(), mnx, mny, mxx - mnx + 1, mxy - mny + 1};
      jobs.push_back(j);
      for (const auto& p : c) {
        pts.push_back(p.x);
        pts.push_back(p.y);
      }
    }
  }
  first_job[todo.size()] = (int)jobs.size();
  const int nj = (int)jobs.size();
  // The unclip kernel is lane-serial: about 0.2 ms however few polygons it gets, against 6 us per polygon and pool thread on the host -
  // it takes the list when there are more than 40 polygons per thread (32 text maps of three polygons with 16 threads: host; with one: device)
  if (dev_unclip && !det.device_unclip_always() && nj <= 40 * det.post_threads()) dev_unclip = false;
  // unclip on the device behind the box score (unclip.hip): per candidate a status, and for the ones it settles the adjusted polygon.
  // Job list up and results down through the handle's pinned buffer (asynchronous copies, one wait)
  const double *sums = nullptr, *counts = nullptr;
  const int32_t *ustatus = nullptr, *ulen = nullptr;
  const uint32_t* uxy = nullptr;
  if (nj > 0) {
    const size_t npts = pts.size() / 2;
    const size_t o_jobs = 0;
    const size_t o_pts = o_jobs + align256(jobs.size() * sizeof(BoxScoreJob));
    const size_t o_adj = o_pts + align256(pts.size() * 4);
    const size_t o_sum = o_adj + align256((size_t)n * 16);            // from here on: results (one block on either side)
    const size_t o_cnt = o_sum + align256((size_t)nj * 8);
    const size_t o_st = o_cnt + align256((size_t)nj * 8);
    const size_t o_len = o_st + align256((size_t)nj * 4);
    const size_t o_oxy = o_len + align256((size_t)nj * 4);
    const size_t o_work = o_oxy + align256(dev_unclip ? 3 * npts * 8 : 0);
    const size_t total = o_work + align256(dev_unclip ? unclip_work_bytes(npts, nj) : 0);
    scratch = static_cast<char*>(det.scratch(1, total));  // slot 0 (map copy) stays valid
    char* hb = static_cast<char*>(det.host_scratch(o_work));
    std::memcpy(hb + o_jobs, jobs.data(), jobs.size() * sizeof(BoxScoreJob));
    std::memcpy(hb + o_pts, pts.data(), pts.size() * 4);
    std::memcpy(hb + o_adj, adj, (size_t)n * 16);
    OCR_HIP(hipMemcpyAsync(scratch + o_jobs, hb + o_jobs, o_sum - o_jobs, hipMemcpyHostToDevice, s));   // jobs, points, adjust values: one copy
    launch_box_scores(prob_dev, h, w, reinterpret_cast<const BoxScoreJob*>(scratch + o_jobs),
                      reinterpret_cast<const int32_t*>(scratch + o_pts), nj, reinterpret_cast<double*>(scratch + o_sum),
                      reinterpret_cast<double*>(scratch + o_cnt), s);
    if (dev_unclip)
      launch_unclip(reinterpret_cast<const BoxScoreJob*>(scratch + o_jobs), reinterpret_cast<const int32_t*>(scratch + o_pts), nullptr, nj, npts,
                    reinterpret_cast<const double*>(scratch + o_sum), reinterpret_cast<const double*>(scratch + o_cnt),
                    reinterpret_cast<const double*>(scratch + o_adj), up, scratch + o_work, reinterpret_cast<uint32_t*>(scratch + o_oxy),
                    reinterpret_cast<int32_t*>(scratch + o_len), reinterpret_cast<int32_t*>(scratch + o_st), s);
    OCR_HIP(hipMemcpyAsync(hb + o_sum, scratch + o_sum, (dev_unclip ? o_work : o_st) - o_sum, hipMemcpyDeviceToHost, s));   // sums, counts [, status, lengths, polygons]
    OCR_HIP(hipStreamSynchronize(s));
    sums = reinterpret_cast<const double*>(hb + o_sum);
    counts = reinterpret_cast<const double*>(hb + o_cnt);
    ustatus = reinterpret_cast<const int32_t*>(hb + o_st);
    ulen = reinterpret_cast<const int32_t*>(hb + o_len);
    uxy = reinterpret_cast<const uint32_t*>(hb + o_oxy);
  }
#ifdef POSTPROC_TIMING
  T3 = tnow();
#endif

  ++det.post_stats[5];
  for (int j = 0; j < nj; ++j) ++det.post_stats[(dev_unclip && ustatus[j] != UNCLIP_HOST) ? 2 : 3];
  // what the device did not settle - filters + unclip + coordinate adjustment (metrics.rs:100-123) - per image on the pool
  pool.parallel_for((int)todo.size(), [&](int k) {
    const int b = todo[k];
    PerImage& r = per[b];
    int j = first_job[k];
    for (const auto& c : cands[b]) {
      const double score = sums[j] / counts[j];
      take(r, dev_unclip ? ustatus[j] : (int)UNCLIP_HOST, dev_unclip ? uxy + 6 * (size_t)jobs[j].pt_offset : nullptr, dev_unclip ? ulen[j] : 0, c,
           score, b);
      ++j;
    }
  });
  // the CSR block in image order
  auto res = std::make_unique<PolygonsOwned>();
  res->img_offsets.push_back(0);
  res->poly_offsets.push_back(0);
  for (int b = 0; b < n; ++b) {
    const PerImage& r = per[b];
    res->xy.insert(res->xy.end(), r.xy.begin(), r.xy.end());
    for (int32_t L : r.lens) res->poly_offsets.push_back(res->poly_offsets.back() + L);
    res->scores.insert(res->scores.end(), r.scores.begin(), r.scores.end());
    res->img_offsets.push_back((int32_t)res->scores.size());
  }
  res->finish();
  *out = &res.release()->view;
#ifdef POSTPROC_TIMING
  fprintf(stderr, "postprocess n=%d: binarize+copy %.3f ms, contours %.3f ms, box scores (%d) %.3f ms, finish %.3f ms\n", n, T1 - T0, T2 - T1,
          nj, T3 - T2, tnow() - T3);
#endif
}
}  // namespace

extern "C" {

const char* ocr_last_error(void) { return ocr::g_last_error.c_str(); }
const char* ocr_version(void) { return "ocr_amd 0.1 gfx950"; }
int ocr_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int ocr_det_create(const void* weights, size_t bytes, int device, ocr_det_t** out) {
  return ocr_det_create_with_options(weights, bytes, device, nullptr, out);
}
int ocr_det_create_with_options(const void* weights, size_t bytes, int device, const char* options, ocr_det_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_det_create: out is null");
    *out = nullptr;
    *out = new ocr_det(weights, bytes, device, options);
  });
}
int ocr_det_create_from_varstore(const char* path, int device, ocr_det_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_det_create_from_varstore: out is null");
    *out = nullptr;
    const std::vector<uint8_t> blob = ocr::varstore_to_blob(path, 1);
    *out = new ocr_det(blob.data(), blob.size(), device);
  });
}
void ocr_det_destroy(ocr_det_t* det) { delete det; }

int ocr_varstore_to_blob(const char* path, int kind, void** blob, size_t* blob_bytes) {
  return guard([&] {
    if (!blob || !blob_bytes) ocr::fail(OCR_ERR_INVALID, "ocr_varstore_to_blob: null output");
    *blob = nullptr;
    *blob_bytes = 0;
    if (kind < 0 || kind > 2) ocr::fail(OCR_ERR_INVALID, "ocr_varstore_to_blob: kind %d", kind);
    const std::vector<uint8_t> b = ocr::varstore_to_blob(path, kind);
    void* p = std::malloc(b.size() ? b.size() : 1);
    if (!p) ocr::fail(OCR_ERR_INTERNAL, "out of memory");
    std::memcpy(p, b.data(), b.size());
    *blob = p;
    *blob_bytes = b.size();
  });
}
void ocr_blob_free(void* blob) { std::free(blob); }

int ocr_det_set_stream(ocr_det_t* det, void* s) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.set_stream(static_cast<hipStream_t>(s));
  });
}

int ocr_det_set_precision(ocr_det_t* det, int precision) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.set_precision(precision);
  });
}

int ocr_det_forward(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, int mem_kind) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    if (mem_kind == OCR_MEM_HOST) {
      if (!x || !prob) ocr::fail(OCR_ERR_INVALID, "det_forward: null tensor");
      if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) ocr::fail(OCR_ERR_INVALID, "det_forward: N=%d H=%d W=%d (H and W must be positive multiples of 32)", n, h, w);
      det->impl.forward_host(x, 0, n, h, w, prob);
    } else if (mem_kind == OCR_MEM_DEVICE) {
      det->impl.forward(x, n, h, w, prob, nullptr, 0.f, nullptr);
      det->impl.synchronize();
    } else {
      ocr::fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    }
  });
}

int ocr_det_forward_u8(ocr_det_t* det, const uint8_t* x, int n, int h, int w, float* prob, int mem_kind) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    if (!x || !prob) ocr::fail(OCR_ERR_INVALID, "det_forward_u8: null tensor");
    if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) ocr::fail(OCR_ERR_INVALID, "det_forward_u8: N=%d H=%d W=%d (H and W must be positive multiples of 32)", n, h, w);
    if (mem_kind == OCR_MEM_HOST) {
      det->impl.forward_host(x, 1, n, h, w, prob);
    } else if (mem_kind == OCR_MEM_DEVICE) {
      det->impl.forward(x, n, h, w, prob, nullptr, 0.f, nullptr, 1);
      det->impl.synchronize();
    } else {
      ocr::fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    }
  });
}

int ocr_host_alloc(size_t bytes, void** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_host_alloc: out is null");
    *out = nullptr;
    OCR_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  });
}
void ocr_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int ocr_det_forward_async(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, uint8_t* bitmap, float thresh) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.forward(x, n, h, w, prob, bitmap, thresh, nullptr);
  });
}

int ocr_det_synchronize(ocr_det_t* det) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "null handle");
    det->impl.synchronize();
  });
}

int ocr_det_forward_profile(ocr_det_t* det, const float* x, int n, int h, int w, float* prob, int max_entries,
                            const char** names, float* ms, double* flops, double* bytes, int* n_entries) {
  return guard([&] {
    if (!det || !n_entries) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<ocr::ProfileEntry> prof;
    det->impl.forward(x, n, h, w, prob, nullptr, 0.f, &prof);
    const int k = std::min<int>(max_entries, (int)prof.size());
    for (int i = 0; i < k; ++i) {
      if (names) names[i] = prof[i].name;
      if (ms) ms[i] = prof[i].ms;
      if (flops) flops[i] = prof[i].flops;
      if (bytes) bytes[i] = prof[i].bytes;
    }
    *n_entries = k;
  });
}

int ocr_preprocess_image(ocr_det_t* det, const uint8_t* rgba, int w, int h, int target_w, int target_h, uint8_t* gray,
                         float* gray_f32, double* adj_xy, int mem_kind) {
  return guard([&] {
    using namespace ocr;
    if (!det || !rgba || (!gray && !gray_f32)) fail(OCR_ERR_INVALID, "preprocess_image: null argument");
    if (w < 1 || h < 1 || target_w < 1 || target_h < 1) fail(OCR_ERR_INVALID, "preprocess_image: bad dimensions");
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    OCR_HIP(hipSetDevice(det->impl.device()));
    hipStream_t s = det->impl.stream();
    const size_t px_in = (size_t)w * h * 4, px_out = (size_t)target_w * target_h;
    const size_t need = preprocess_scratch_bytes(w, h, target_w, target_h);
    if (mem_kind == OCR_MEM_DEVICE) {
      // gray may be null when only the f32 frame is wanted: stage it in scratch
      char* sc = static_cast<char*>(det->impl.scratch(1, need + align256(px_out)));
      uint8_t* g = gray ? gray : reinterpret_cast<uint8_t*>(sc + align256(need));
      launch_preprocess(rgba, w, h, target_w, target_h, g, gray_f32, sc, need, adj_xy, s);
      OCR_HIP(hipStreamSynchronize(s));
    } else {
      const size_t o_in = align256(need), o_g = o_in + align256(px_in), o_f = o_g + align256(px_out);
      char* sc = static_cast<char*>(det->impl.scratch(1, o_f + align256(px_out * 4)));
      OCR_HIP(hipMemcpyAsync(sc + o_in, rgba, px_in, hipMemcpyHostToDevice, s));
      launch_preprocess(reinterpret_cast<const unsigned char*>(sc + o_in), w, h, target_w, target_h,
                        reinterpret_cast<unsigned char*>(sc + o_g), gray_f32 ? reinterpret_cast<float*>(sc + o_f) : nullptr, sc, need,
                        adj_xy, s);
      if (gray) OCR_HIP(hipMemcpyAsync(gray, sc + o_g, px_out, hipMemcpyDeviceToHost, s));
      if (gray_f32) OCR_HIP(hipMemcpyAsync(gray_f32, sc + o_f, px_out * 4, hipMemcpyDeviceToHost, s));
      OCR_HIP(hipStreamSynchronize(s));
    }
  });
}

int ocr_extract_crops(ocr_det_t* det, const float* frames, int n, int h, int w, int mem_kind, const ocr_polygons_t* polys,
                      const double* adj_xy, float* crops) {
  return guard([&] {
    using namespace ocr;
    if (!det || !frames || !polys || !adj_xy || (!crops && polys->n_polygons > 0)) fail(OCR_ERR_INVALID, "extract_crops: null argument");
    if (polys->n_images != n) fail(OCR_ERR_INVALID, "extract_crops: polygon block holds %d images, frames %d", polys->n_images, n);
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    const int np = polys->n_polygons;
    if (np == 0) return;
    std::vector<CropBox> boxes;
    boxes.reserve(np);
    for (int b = 0; b < n; ++b) {
      const double ax = adj_xy[2 * b], ay = adj_xy[2 * b + 1];
      for (int k = polys->img_offsets[b]; k < polys->img_offsets[b + 1]; ++k) {
        double mnx = 1e300, mxx = -1e300, mny = 1e300, mxy = -1e300;
        for (int v = polys->poly_offsets[k]; v < polys->poly_offsets[k + 1]; ++v) {
          const double x = polys->xy[2 * v] * ax, y = polys->xy[2 * v + 1] * ay;  // back to frame coordinates
          mnx = std::min(mnx, x); mxx = std::max(mxx, x);
          mny = std::min(mny, y); mxy = std::max(mxy, y);
        }
        const double x0 = std::min(std::max(mnx, 0.0), w - 1.0), x1 = std::min(std::max(mxx + 1.0, x0 + 1.0), (double)w);
        const double y0 = std::min(std::max(mny, 0.0), h - 1.0), y1 = std::min(std::max(mxy + 1.0, y0 + 1.0), (double)h);
        boxes.push_back({b, (float)x0, (float)y0, (float)x1, (float)y1});
      }
    }
    OCR_HIP(hipSetDevice(det->impl.device()));
    // while a pipelined forward is in flight on the handle's stream the crops of the batch that just came back are cut on
    // the post-processing stream, beside it: ordered behind everything that was queued on the handle's stream BEFORE that
    // forward (whatever produced these frames), not behind the forward itself
    hipStream_t s = det->impl.stream();
    if (det->impl.has_pending()) {
      s = det->impl.post_stream();
      if (det->impl.before_forward_event()) OCR_HIP(hipStreamWaitEvent(s, det->impl.before_forward_event(), 0));
    }
    const size_t fr_bytes = (size_t)n * h * w * 4, bx_bytes = boxes.size() * sizeof(CropBox), cr_bytes = (size_t)np * 784 * 4;
    if (mem_kind == OCR_MEM_DEVICE) {
      char* sc = static_cast<char*>(det->impl.scratch(1, align256(bx_bytes)));
      OCR_HIP(hipMemcpyAsync(sc, boxes.data(), bx_bytes, hipMemcpyHostToDevice, s));
      launch_crops(frames, h, w, reinterpret_cast<const CropBox*>(sc), np, crops, s);
      OCR_HIP(hipStreamSynchronize(s));
    } else {
      const size_t o_fr = align256(bx_bytes), o_cr = o_fr + align256(fr_bytes);
      char* sc = static_cast<char*>(det->impl.scratch(0, o_cr + align256(cr_bytes)));
      OCR_HIP(hipMemcpyAsync(sc, boxes.data(), bx_bytes, hipMemcpyHostToDevice, s));
      OCR_HIP(hipMemcpyAsync(sc + o_fr, frames, fr_bytes, hipMemcpyHostToDevice, s));
      launch_crops(reinterpret_cast<const float*>(sc + o_fr), h, w, reinterpret_cast<const CropBox*>(sc), np,
                   reinterpret_cast<float*>(sc + o_cr), s);
      OCR_HIP(hipMemcpyAsync(crops, sc + o_cr, cr_bytes, hipMemcpyDeviceToHost, s));
      OCR_HIP(hipStreamSynchronize(s));
    }
  });
}

static std::vector<std::vector<ocr::geom::Pt>> csr_polys(const uint32_t* xy, const int32_t* offsets, int n) {
  std::vector<std::vector<ocr::geom::Pt>> out(n);
  for (int k = 0; k < n; ++k)
    for (int v = offsets[k]; v < offsets[k + 1]; ++v) out[k].push_back({(int)xy[2 * v], (int)xy[2 * v + 1]});
  return out;
}

int ocr_evaluate_image(const uint32_t* gt_xy, const int32_t* gt_offsets, int n_gt, const uint8_t* ignore_flags,
                       const uint32_t* pred_xy, const int32_t* pred_offsets, int n_pred, ocr_metrics_item_t* out) {
  return guard([&] {
    if (!out || n_gt < 0 || n_pred < 0 || (n_gt > 0 && (!gt_xy || !gt_offsets || !ignore_flags)) || (n_pred > 0 && (!pred_xy || !pred_offsets)))
      ocr::fail(OCR_ERR_INVALID, "evaluate_image: null argument");
    std::vector<bool> ign(n_gt);
    for (int i = 0; i < n_gt; ++i) ign[i] = ignore_flags[i] != 0;
    const ocr::geom::MetricsItem m = ocr::geom::evaluate_image(csr_polys(gt_xy, gt_offsets, n_gt), ign, csr_polys(pred_xy, pred_offsets, n_pred));
    out->precision = m.precision;
    out->recall = m.recall;
    out->hmean = m.hmean;
    out->gt_care = m.gt_care;
    out->det_care = m.det_care;
    out->det_matched = m.det_matched;
  });
}

int ocr_combine_results(const ocr_metrics_item_t* items, int n, double* precision, double* recall, double* hmean) {
  return guard([&] {
    if ((n > 0 && !items) || !precision || !recall || !hmean) ocr::fail(OCR_ERR_INVALID, "combine_results: null argument");
    std::vector<ocr::geom::MetricsItem> v(n);
    for (int i = 0; i < n; ++i) v[i] = {items[i].precision, items[i].recall, items[i].hmean, items[i].gt_care, items[i].det_care, items[i].det_matched};
    ocr::geom::combine_results(v.data(), n, precision, recall, hmean);
  });
}

void ocr_postproc_default_params(ocr_postproc_params_t* p) {
  if (!p) return;
  p->thresh = 0.6;        // metrics.rs:38
  p->box_thresh = 0.7;    // metrics.rs:64
  p->min_size = 5.0;      // metrics.rs:66
  p->unclip_ratio = 2.0;  // metrics.rs:103
  p->skip_degenerate = 0; // faithful: the reference aborts on such a candidate
  p->reserved = 0;
}

int ocr_det_postprocess(ocr_det_t* det, const float* prob, int n, int h, int w, int mem_kind, const double* adj,
                        const ocr_postproc_params_t* params, ocr_polygons_t** out) {
  return guard([&] {
    if (!det) ocr::fail(OCR_ERR_INVALID, "det_postprocess needs a detector handle (GPU + stream)");
    if (mem_kind != OCR_MEM_HOST && mem_kind != OCR_MEM_DEVICE) ocr::fail(OCR_ERR_INVALID, "mem_kind %d", mem_kind);
    ocr_postproc_params_t prm;
    ocr_postproc_default_params(&prm);
    if (params) prm = *params;
    if (out) *out = nullptr;
    postprocess(det->impl, prob, n, h, w, mem_kind, adj, prm, out, det->impl.stream());
  });
}

int ocr_det_post_stats(ocr_det_t* det, int64_t out[6]) {
  return guard([&] {
    if (!det || !out) ocr::fail(OCR_ERR_INVALID, "det_post_stats: null argument");
    for (int i = 0; i < 6; ++i) out[i] = (int64_t)det->impl.post_stats[i];
  });
}

int ocr_det_detect_pipelined(ocr_det_t* det, const float* x_dev, int n, int h, int w, float* prob_dev, const double* adj_xy,
                             const ocr_postproc_params_t* params, ocr_polygons_t** prev_out) {
  return guard([&] {
    using namespace ocr;
    if (!det || !prev_out) fail(OCR_ERR_INVALID, "detect_pipelined: null argument");
    *prev_out = nullptr;
    Detector& d = det->impl;
    OCR_HIP(hipSetDevice(d.device()));
    Detector::Pending next;
    if (x_dev) {
      if (!prob_dev || !adj_xy) fail(OCR_ERR_INVALID, "detect_pipelined: null tensor");
      d.mark_before_forward();
      d.forward(x_dev, n, h, w, prob_dev, nullptr, 0.f, nullptr);   // enqueue: runs while the previous batch is post-processed
      next.prob = prob_dev;
      next.n = n;
      next.h = h;
      next.w = w;
      next.adj.assign(adj_xy, adj_xy + 2 * (size_t)n);
      ocr_postproc_default_params(&next.params);
      if (params) next.params = *params;
      next.event = d.pipeline_event();
      OCR_HIP(hipEventRecord(next.event, d.stream()));
      next.valid = true;
    }
    Detector::Pending prev = d.swap_pending(next);
    if (prev.valid) {
      hipStream_t ps = d.post_stream();
      OCR_HIP(hipStreamWaitEvent(ps, prev.event, 0));   // the forward that produced prev.prob
      postprocess(d, prev.prob, prev.n, prev.h, prev.w, OCR_MEM_DEVICE, prev.adj.data(), prev.params, prev_out, ps, prev.pretraced, prev.prechained);
    }
    pretrace_pending(d);
  });
}

int ocr_det_detect_pipelined_host(ocr_det_t* det, const void* x_host, int x_elem, int n, int h, int w, float* prob_host,
                                  const double* adj_xy, const ocr_postproc_params_t* params, ocr_polygons_t** prev_out) {
  return guard([&] {
    using namespace ocr;
    if (!det || !prev_out) fail(OCR_ERR_INVALID, "detect_pipelined_host: null argument");
    *prev_out = nullptr;
    Detector& d = det->impl;
    OCR_HIP(hipSetDevice(d.device()));
    // finishing a batch: wait for its forward, send the map home if it was asked for, polygons out
    auto finish = [&](Detector::Pending& prev) {
      hipStream_t ps = d.post_stream();
      OCR_HIP(hipStreamWaitEvent(ps, prev.event, 0));   // the forward that produced prev.prob
      if (prev.prob_host)   // the caller asked for the map too: it leaves on the same stream, ahead of the bit image
        OCR_HIP(hipMemcpyAsync(prev.prob_host, prev.prob, (size_t)prev.n * prev.h * prev.w * 4, hipMemcpyDeviceToHost, ps));
      postprocess(d, prev.prob, prev.n, prev.h, prev.w, OCR_MEM_DEVICE, prev.adj.data(), prev.params, prev_out, ps, prev.pretraced, prev.prechained);
    };
    Detector::Pending next;
    if (x_host) {
      if (!adj_xy) fail(OCR_ERR_INVALID, "detect_pipelined_host: null adjust values");
      if (x_elem != OCR_ELEM_F32 && x_elem != OCR_ELEM_U8) fail(OCR_ERR_INVALID, "detect_pipelined_host: element kind %d", x_elem);
      if (n <= 0 || h <= 0 || w <= 0 || h % 32 || w % 32) fail(OCR_ERR_INVALID, "detect_pipelined_host: N=%d H=%d W=%d (H and W must be positive multiples of 32)", n, h, w);
      const size_t es = x_elem == OCR_ELEM_U8 ? 1 : 4, px = (size_t)n * h * w;
      constexpr int SET = Detector::STAGE_PIPELINED;
      // a batch that needs larger staging slots than the pending one (more frames, or f32 after u8) frees the slot the pending
      // batch's map lives in: that batch is finished FIRST (this one call loses its overlap), then the slots grow
      if (d.staging_would_grow(SET, px * es, px) && d.has_pending()) {
        Detector::Pending none;
        Detector::Pending prev = d.swap_pending(none);
        finish(prev);
      }
      d.ensure_staging(SET, px * es, px);
      // this batch's frames into the free input slot (the slot's previous forward was awaited when ITS polygons came back),
      // the forward behind the copy; the map stays on the device
      const int slot = d.next_stage_slot(SET);
      hipEvent_t arrived;
      const void* xd = d.stage_input(SET, slot, x_host, px * es, &arrived);
      d.mark_before_forward();
      d.forward(xd, n, h, w, d.stage_prob(SET, slot), nullptr, 0.f, nullptr, x_elem == OCR_ELEM_U8 ? 1 : 0, arrived);
      OCR_HIP(hipEventRecord(d.forward_done_event(SET, slot), d.stream()));
      d.stage_used(SET);
      next.prob = d.stage_prob(SET, slot);
      next.prob_host = prob_host;
      next.n = n;
      next.h = h;
      next.w = w;
      next.adj.assign(adj_xy, adj_xy + 2 * (size_t)n);
      ocr_postproc_default_params(&next.params);
      if (params) next.params = *params;
      next.event = d.forward_done_event(SET, slot);
      next.valid = true;
    }
    Detector::Pending prev = d.swap_pending(next);
    if (prev.valid) finish(prev);
    pretrace_pending(d);
  });
}

void ocr_polygons_free(ocr_polygons_t* p) {
  if (!p) return;
  delete reinterpret_cast<PolygonsOwned*>(reinterpret_cast<char*>(p) - offsetof(PolygonsOwned, view));
}

int ocr_rec_create(const void* weights, size_t bytes, int device, ocr_rec_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_rec_create: out is null");
    *out = nullptr;
    *out = new ocr_rec(weights, bytes, device);
  });
}
int ocr_rec_create_from_varstore(const char* path, int device, ocr_rec_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_rec_create_from_varstore: out is null");
    *out = nullptr;
    const std::vector<uint8_t> blob = ocr::varstore_to_blob(path, 2);
    *out = new ocr_rec(blob.data(), blob.size(), device);
  });
}
void ocr_rec_destroy(ocr_rec_t* rec) { delete rec; }
int ocr_rec_set_stream(ocr_rec_t* rec, void* s) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.set_stream(static_cast<hipStream_t>(s));
  });
}
int ocr_rec_set_options(ocr_rec_t* rec, const char* options) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.set_options(options);
  });
}
int ocr_rec_synchronize(ocr_rec_t* rec) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.synchronize();
  });
}
int ocr_rec_forward(ocr_rec_t* rec, const float* crops, int n, float* logits, int mem_kind) {
  return guard([&] {
    if (!rec || !crops || !logits) ocr::fail(OCR_ERR_INVALID, "null argument");
    if (n < 0) ocr::fail(OCR_ERR_INVALID, "negative crop count");
    if (mem_kind == OCR_MEM_HOST) rec->impl.forward_host(crops, n, logits, nullptr, nullptr);
    else {
      rec->impl.classify(crops, n, logits, nullptr, nullptr);
      rec->impl.synchronize();
    }
  });
}
int ocr_rec_classify_async(ocr_rec_t* rec, const float* crops, int n, float* logits, int32_t* labels, double* probs) {
  return guard([&] {
    if (!rec) ocr::fail(OCR_ERR_INVALID, "null handle");
    rec->impl.classify(crops, n, logits, labels, probs);
  });
}
int ocr_rec_classify_profile(ocr_rec_t* rec, const float* crops, int n, int32_t* labels, double* probs, int max_entries,
                             const char** names, float* ms, double* flops, double* bytes, int* n_entries) {
  return guard([&] {
    if (!rec || !n_entries) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<ocr::ProfileEntry> prof;
    rec->impl.classify(crops, n, nullptr, labels, probs, &prof);
    const int k = std::min<int>(max_entries, (int)prof.size());
    for (int i = 0; i < k; ++i) {
      if (names) names[i] = prof[i].name;
      if (ms) ms[i] = prof[i].ms;
      if (flops) flops[i] = prof[i].flops;
      if (bytes) bytes[i] = prof[i].bytes;
    }
    *n_entries = k;
  });
}
int ocr_rec_classify(ocr_rec_t* rec, const float* crops, int n, int32_t* labels, double* probs, int mem_kind) {
  return guard([&] {
    if (!rec || !crops) ocr::fail(OCR_ERR_INVALID, "null argument");
    if (n < 0) ocr::fail(OCR_ERR_INVALID, "negative crop count");
    if (mem_kind == OCR_MEM_HOST) rec->impl.forward_host(crops, n, nullptr, labels, probs);
    else {
      rec->impl.classify(crops, n, nullptr, labels, probs);
      rec->impl.synchronize();
    }
  });
}
int ocr_ctc_greedy_decode(ocr_rec_t* rec, const float* logits, int n, int t, int c, int blank, int mem_kind, int32_t* labels, int32_t* lengths) {
  return guard([&] {
    using namespace ocr;
    if (!rec || !logits || !labels || !lengths) fail(OCR_ERR_INVALID, "ctc_greedy_decode: null argument");
    if (n < 0 || t <= 0 || c <= 0 || blank < 0 || blank >= c) fail(OCR_ERR_INVALID, "ctc_greedy_decode: N=%d T=%d C=%d blank=%d", n, t, c, blank);
    if (n == 0) return;
    OCR_HIP(hipSetDevice(rec->impl.device()));
    hipStream_t s = rec->impl.stream();
    if (mem_kind == OCR_MEM_DEVICE) {   // enqueued on the handle's stream and awaited (the lengths are usually read right away)
      launch_ctc_greedy(logits, n, t, c, blank, labels, lengths, s);
      OCR_HIP(hipStreamSynchronize(s));
      return;
    }
    const size_t in_b = (size_t)n * t * c * 4, lab_b = (size_t)n * t * 4, len_b = (size_t)n * 4;
    char* d = nullptr;
    OCR_HIP(hipMalloc(reinterpret_cast<void**>(&d), align256(in_b) + align256(lab_b) + align256(len_b)));
    struct Free { char* p; ~Free() { (void)hipFree(p); } } free_d{d};
    int32_t* d_lab = reinterpret_cast<int32_t*>(d + align256(in_b));
    int32_t* d_len = reinterpret_cast<int32_t*>(d + align256(in_b) + align256(lab_b));
    OCR_HIP(hipMemcpyAsync(d, logits, in_b, hipMemcpyHostToDevice, s));
    launch_ctc_greedy(reinterpret_cast<const float*>(d), n, t, c, blank, d_lab, d_len, s);
    OCR_HIP(hipMemcpyAsync(labels, d_lab, lab_b, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipMemcpyAsync(lengths, d_len, len_b, hipMemcpyDeviceToHost, s));
    OCR_HIP(hipStreamSynchronize(s));
  });
}
const char* ocr_rec_alphabet(void) { return "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"; }

/* ---- multi-GPU exchange (comm.hip) */
struct ocr_comm {
  ocr::Comm* impl = nullptr;
  ~ocr_comm() { ocr::comm_destroy(impl); }
};

int ocr_comm_unique_id(uint8_t* id) {
  return guard([&] { ocr::comm_unique_id(id); });
}
int ocr_comm_rccl_version(int* version) {
  return guard([&] {
    if (!version) ocr::fail(OCR_ERR_INVALID, "null argument");
    *version = ocr::comm_rccl_version();
  });
}
int ocr_comm_create(const uint8_t* id, int world, int rank, int device, ocr_comm_t** out) {
  return guard([&] {
    if (!out) ocr::fail(OCR_ERR_INVALID, "ocr_comm_create: out is null");
    *out = nullptr;
    auto c = std::make_unique<ocr_comm>();
    c->impl = ocr::comm_create(id, world, rank, device);
    *out = c.release();
  });
}
void ocr_comm_destroy(ocr_comm_t* comm) { delete comm; }
int ocr_comm_all_gather_polygons(ocr_comm_t* comm, const ocr_polygons_t* local, ocr_polygons_t** all) {
  return guard([&] {
    if (!comm || !local || !all) ocr::fail(OCR_ERR_INVALID, "null argument");
    *all = nullptr;
    auto res = std::make_unique<PolygonsOwned>();
    ocr::comm_all_gather_polygons(comm->impl, *local, *res);
    *all = &res.release()->view;
  });
}
int ocr_comm_all_gather_labels(ocr_comm_t* comm, const int32_t* labels, int n_local, int32_t* all, int capacity,
                               int32_t* counts, int* n_all) {
  return guard([&] {
    if (!comm || !n_all) ocr::fail(OCR_ERR_INVALID, "null argument");
    std::vector<int32_t> a, c;
    ocr::comm_all_gather_labels(comm->impl, labels, n_local, a, c);
    *n_all = (int)a.size();
    if ((int)a.size() > capacity) ocr::fail(OCR_ERR_INVALID, "all_gather_labels: %zu labels, capacity %d", a.size(), capacity);
    if (!a.empty()) {
      if (!all) ocr::fail(OCR_ERR_INVALID, "null label buffer");
      std::memcpy(all, a.data(), a.size() * 4);
    }
    if (counts) std::memcpy(counts, c.data(), c.size() * 4);
  });
}

}  // extern "C"
