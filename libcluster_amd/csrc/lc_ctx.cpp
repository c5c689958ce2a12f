#include "lc_ctx.hpp"
#include "lc_engine.hpp"  // the host worker pool (parallel_chunks)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <iostream>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>

namespace lcc {

#define LC_HIP(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      throw HipFailure(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr " (" __FILE__ \
                                                                            ":" +                      \
                       std::to_string(__LINE__) + ")");                                                \
  } while (0)

// Freed device / page-locked blocks are kept for re-use: the split search builds and tears down a sub-context per
// attempt, and hipFree / hipHostFree (50-340 us each, the former also a device-wide synchronisation) were 6 % of a
// model-selection run.  Blocks are matched best-fit within 2x; the device cache is capped at 16 GiB per process
// (lc_trim_cache() empties it).  Nothing is returned to the driver at process exit (the runtime may be gone).
namespace {
struct BlockCache {
  struct Block {
    void* p;
    size_t bytes;
    int device;
    size_t owner;  // host thread that released it (0: anybody).  A thread works on ONE stream, so handing a block back
                   // to the thread that freed it is ordered by that stream; another thread (LIBCLUSTER_GPUS: another
                   // stream) may only take it once the owner has let go of it (cache_release_thread, after a sync)
  };
  std::mutex m;
  std::vector<Block> dev, pinned;
  size_t dev_bytes = 0;
  // A tag per host thread: a counter, never re-used (the OS recycles thread ids, and a hash of one could hand a block
  // with work in flight to an unrelated thread).  The thread's last act is to let go of its blocks: a short-lived API
  // thread, or a shard thread that threw, does not leave blocks behind that nobody else may take.
  struct ThreadTag {
    size_t id;
    ThreadTag() {
      static std::atomic<size_t> next{1};
      id = next.fetch_add(1);
    }
    ~ThreadTag();  // cache_release_thread() for this tag
  };
  static size_t thread_tag() {
    static thread_local ThreadTag t;
    return t.id;
  }
  static BlockCache& get() {
    static BlockCache* c = new BlockCache();  // intentionally leaked
    return *c;
  }
  static int& relaxed_fit() {
    static thread_local int depth = 0;
    return depth;
  }
  void* take(std::vector<Block>& v, size_t need, int device, size_t* got) {
    std::lock_guard<std::mutex> g(m);
    const size_t me = thread_tag();
    int best = -1;
    // (a block up to twice the request -- up to 16 times for short-lived sub-problem buffers, RelaxedFit: the split
    // search's sub-problems halve from round to round, and mapping a fresh block of tens of gigabytes costs more than
    // lending a big one for a few milliseconds)
    const size_t slack = relaxed_fit() > 0 ? 16 : 2;
    for (int i = 0; i < (int)v.size(); ++i)
      if (v[i].device == device && v[i].bytes >= need && v[i].bytes <= slack * need + 4096 &&
          (v[i].owner == 0 || v[i].owner == me) && (best < 0 || v[i].bytes < v[best].bytes))
        best = i;
    if (best < 0) return nullptr;
    void* p = v[best].p;
    *got = v[best].bytes;
    if (&v == &dev) dev_bytes -= v[best].bytes;
    v[best] = v.back();
    v.pop_back();
    return p;
  }
};
// Released device blocks are kept for re-use up to this much (LC_BLOCK_CACHE_GB, default 64: a fresh block of tens of
// gigabytes costs seconds to map, and at tens of millions of rows every sub-problem of the split search takes and
// returns such blocks); a failed allocation trims the cache and retries (DevBuf::reserve).
static size_t dev_cache_limit() {
  static const size_t v = [] {
    const char* e = std::getenv("LC_BLOCK_CACHE_GB");
    const double gb = e ? std::atof(e) : 64.0;
    return (size_t)(std::max(0.0, gb) * (double)((size_t)1 << 30));
  }();
  return v;
}
#define DEV_CACHE_LIMIT dev_cache_limit()
#define DEV_BLOCK_LIMIT (dev_cache_limit() / 4 * 3)
int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}
}  // namespace

RelaxedFit::RelaxedFit() { ++BlockCache::relaxed_fit(); }
RelaxedFit::~RelaxedFit() { --BlockCache::relaxed_fit(); }

static void cache_release_tag(size_t me) {
  BlockCache& c = BlockCache::get();
  std::lock_guard<std::mutex> g(c.m);
  for (auto& b : c.dev)
    if (b.owner == me) b.owner = 0;
  for (auto& b : c.pinned)
    if (b.owner == me) b.owner = 0;
}
void cache_release_thread() { cache_release_tag(BlockCache::thread_tag()); }
namespace {
BlockCache::ThreadTag::~ThreadTag() { cache_release_tag(id); }
}  // namespace

void trim_cache() {
  BlockCache& c = BlockCache::get();
  std::lock_guard<std::mutex> g(c.m);
  for (auto& b : c.dev) (void)hipFree(b.p);
  for (auto& b : c.pinned) (void)hipHostFree(b.p);
  c.dev.clear();
  c.pinned.clear();
  c.dev_bytes = 0;
}

template <typename T>
void DevBuf<T>::release() {
  if (p) {
    BlockCache& c = BlockCache::get();
    const size_t bytes = cap * sizeof(T);
    bool kept = false;
    if (bytes <= DEV_BLOCK_LIMIT) {
      std::lock_guard<std::mutex> g(c.m);
      if (c.dev_bytes + bytes <= DEV_CACHE_LIMIT && c.dev.size() < 4096) {
        c.dev.push_back({p, bytes, device >= 0 ? device : current_device(), BlockCache::thread_tag()});
        c.dev_bytes += bytes;
        kept = true;
      }
    }
    if (!kept) (void)hipFree(p);
  }
  p = nullptr;
  cap = 0;
  device = -1;
}

template <typename T>
void DevBuf<T>::reserve(size_t n) {
  if (n <= cap) return;
  release();
  BlockCache& c = BlockCache::get();
  size_t got = 0;
  const int dev = current_device();  // (Context methods select their own device first: use_device)
  void* q = c.take(c.dev, n * sizeof(T), dev, &got);
  if (!q) {
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) {  // make room and retry once
      trim_cache();
      e = hipMalloc(&q, n * sizeof(T));
    }
    if (e != hipSuccess)
      throw AllocFailure(std::string("HIP error: hipMalloc of ") + std::to_string(n * sizeof(T)) +
                       " bytes failed: " + hipGetErrorString(e));
    got = n * sizeof(T);
  }
  p = static_cast<T*>(q);
  cap = got / sizeof(T);
  device = dev;
}

PinnedBuf::~PinnedBuf() {
  if (!p) return;
  BlockCache& c = BlockCache::get();
  std::lock_guard<std::mutex> g(c.m);
  if (c.pinned.size() < 256) c.pinned.push_back({p, cap * sizeof(double), 0, BlockCache::thread_tag()});
  else (void)hipHostFree(p);
}

void PinnedBuf::resize(size_t count) {
  if (count > cap) {
    BlockCache& c = BlockCache::get();
    if (p) {
      std::lock_guard<std::mutex> g(c.m);
      c.pinned.push_back({p, cap * sizeof(double), 0, BlockCache::thread_tag()});
    }
    p = nullptr;
    cap = 0;
    const size_t want = count < 1024 ? 1024 : count + count / 2;
    size_t got = 0;
    void* q = c.take(c.pinned, want * sizeof(double), 0, &got);
    if (!q) {
      hipError_t e = hipHostMalloc(&q, want * sizeof(double), hipHostMallocDefault);
      if (e != hipSuccess)
        throw HipFailure(std::string("HIP error: hipHostMalloc of ") + std::to_string(want * sizeof(double)) +
                         " bytes failed: " + hipGetErrorString(e));
      got = want * sizeof(double);
    }
    p = static_cast<double*>(q);
    cap = got / sizeof(double);
  }
  n = count;
}

template struct DevBuf<double>;
template struct DevBuf<int>;
template struct DevBuf<int64_t>;
template struct DevBuf<unsigned char>;

Context::Context(int device, hipStream_t stream) : device_(device), stream_(stream) {
  // hipGetDeviceCount costs ~2.5 ms per call; the split search builds a context per attempt
  static int n = 0;
  static hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw HipFailure("libcluster_amd: no HIP device available (the E-step has no CPU fallback): " +
                     std::string(e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
  if (device < 0 || device >= n) throw std::invalid_argument("device index out of range");
  LC_HIP(hipSetDevice(device_));
}

void Context::use_device() const { LC_HIP(hipSetDevice(device_)); }

Context::~Context() {
  (void)hipSetDevice(device_);  // the blocks below are released to this device's cache bin
  // the buffers go back to the block cache (no implicit device synchronisation like hipFree): make sure nothing
  // enqueued by this context still uses them
  if (X_.p || qz_[0].buf.p || qz_[1].buf.p) (void)hipStreamSynchronize(stream_);
  for (auto& p : pending_) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (hipEvent_t e : evpool_) (void)hipEventDestroy(e);
}

void Context::synchronize() const {
  use_device();
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::require_gw_width() const {
  if (DP_ > lck::GW_MAX_DP)
    throw std::invalid_argument("D > 1024 is not supported by the Gauss-Wishart kernels (the diagonal and exponential "
                                "families take any D)");
}

void Context::build_layout(int J, const int64_t* Nj, int D) {
  if (J < 1) throw std::invalid_argument("need at least one group of observations");
  if (D < 1) throw std::invalid_argument("observations must have at least one dimension");
  const int DP = lck::padded_dim_wide(D);  // D > 128: diagonal / exponential families only (checked where it matters)
  dcache_release();  // new observations: no cached distance survives, and the buffers are sized by the row count
  J_ = J;
  D_ = D;
  DP_ = DP;
  DC_ = lck::estep_active_width(D, DP);
  if (lck::test_switch("LC_FULL_WIDTH")) DC_ = DP;  // (tests, libcluster_hip_testhooks.so only: every kernel walks the padded width)
  Nj_.assign(Nj, Nj + J);
  goff_.assign(J + 1, 0);
  Ntot_ = 0;
  for (int j = 0; j < J; ++j) {
    if (Nj[j] < 0) throw std::invalid_argument("negative group size");
    goff_[j + 1] = goff_[j] + (Nj[j] + lck::RG - 1) / lck::RG * lck::RG;
    Ntot_ += Nj[j];
  }
  NP_ = goff_[J];
  // (the responsibility buffers are sized by rows x columns: a new row count means new buffers)
  qz_[0].K = qz_[1].K = 0;
  qz_[0].cap = qz_[1].cap = 0;
  LC_HIP(hipSetDevice(device_));
  X_.reserve((size_t)std::max<int64_t>(NP_, 1) * DP_);
  goff_d_.reserve(J + 1);
  LC_HIP(hipMemcpyAsync(goff_d_.p, goff_.data(), (J + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  if (J > 1) {
    std::vector<int> info((size_t)(NP_ / lck::RG));
    for (int j = 0; j < J; ++j) {
      const int64_t g0 = goff_[j] / lck::RG, g1 = goff_[j + 1] / lck::RG;
      for (int64_t g = g0; g < g1; ++g) {
        const int64_t rem = Nj[j] - (g - g0) * lck::RG;
        info[(size_t)g] = lck::rginfo_pack(j, (int)std::min<int64_t>(rem, lck::RG));
      }
    }
    rginfo_.reserve(std::max<size_t>(info.size(), 1));
    if (!info.empty())
      LC_HIP(hipMemcpyAsync(rginfo_.p, info.data(), info.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
  } else {
    rginfo_.release();
  }
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::set_data(int J, const double* const* Xj, const int64_t* Nj, int D, int64_t rs, int64_t cs) {
  build_layout(J, Nj, D);
  if (NP_ == 0) return;
  // The padded row-major image of X is contiguous over groups: pack it in 32 MB pieces that may span many (small)
  // groups, into two page-locked buffers that alternate, so packing piece i+1 overlaps the transfer of piece i and
  // thousands of small documents do not cost a transfer and a synchronisation each.
  const int64_t chunk = std::max<int64_t>(lck::RG, ((int64_t)32 << 20) / 8 / DP_ / lck::RG * lck::RG);
  PinnedBuf stage[2];
  int which = 0;
  const unsigned pack_threads = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
  for (int64_t p0 = 0; p0 < NP_; p0 += chunk, which ^= 1) {
    const int64_t nr = std::min(chunk, NP_ - p0);
    PinnedBuf& st = stage[which];
    if (st.size() == 0) st.resize((size_t)chunk * DP_);
    else LC_HIP(hipStreamSynchronize(stream_));  // the transfer that last used this buffer (two pieces ago) is done
    double* base = st.data();
    // 1024-row blocks of the piece on the worker pool: one host thread packs at about the rate of the link, and
    // column-major callers (Eigen's default) need a blocked transpose rather than a strided gather
    const int DP = DP_;
    const int nblk = (int)((nr + 1023) / 1024);
    auto pack = [&](int blk) {
      const int64_t pa = p0 + (int64_t)blk * 1024, pb = std::min<int64_t>(pa + 1024, p0 + nr);
      int jj = (int)(std::upper_bound(goff_.begin(), goff_.end(), pa) - goff_.begin()) - 1;
      if (jj < 0) jj = 0;
      int64_t pr = pa;
      while (pr < pb) {
        while (jj + 1 < J && goff_[(size_t)jj + 1] <= pr) ++jj;
        const int64_t g0 = goff_[(size_t)jj], vend = g0 + Nj[jj], gend = jj + 1 <= J ? goff_[(size_t)jj + 1] : pb;
        const int64_t v1 = std::min(pb, vend);
        if (pr < v1) {  // valid rows [pr, v1) of group jj
          const double* src = Xj[jj] + (pr - g0) * rs;
          double* dst = base + (size_t)(pr - p0) * DP;
          const int64_t n = v1 - pr;
          if (cs == 1) {
            for (int64_t r = 0; r < n; ++r) {
              std::memcpy(dst + r * DP, src + r * rs, (size_t)D * sizeof(double));
              for (int d = D; d < DP; ++d) dst[r * DP + d] = 0.0;
            }
          } else {  // blocked transpose: contiguous (or short-stride) reads along the rows, writes inside a 64-row tile
            for (int64_t r0 = 0; r0 < n; r0 += 64) {
              const int64_t r1 = std::min<int64_t>(r0 + 64, n);
              for (int d = 0; d < D; ++d) {
                const double* sc = src + d * cs;
                for (int64_t r = r0; r < r1; ++r) dst[r * DP + d] = sc[r * rs];
              }
              for (int64_t r = r0; r < r1; ++r)
                for (int d = D; d < DP; ++d) dst[r * DP + d] = 0.0;
            }
          }
          pr = v1;
        }
        const int64_t z1 = std::min(pb, gend);  // pad rows of the group
        if (pr < z1) {
          std::memset(base + (size_t)(pr - p0) * DP, 0, (size_t)(z1 - pr) * DP * sizeof(double));
          pr = z1;
        }
        if (pr < pb && jj + 1 >= J) {  // (cannot happen: the image ends with the last group's padding)
          std::memset(base + (size_t)(pr - p0) * DP, 0, (size_t)(pb - pr) * DP * sizeof(double));
          pr = pb;
        }
      }
    };
    lce::parallel_chunks(nblk, pack_threads, 1024.0 * DP * 8.0, pack);
    LC_HIP(hipMemcpyAsync(X_.p + (size_t)p0 * DP_, base, (size_t)nr * DP_ * sizeof(double), hipMemcpyHostToDevice,
                          stream_));
  }
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::synth(int64_t N, int D, int K, const double* mu, const double* L, uint64_t seed, int64_t row_offset,
                    double hard) {
  synth_groups(1, &N, D, K, mu, L, nullptr, seed, nullptr, row_offset, hard);
}

void Context::synth_groups(int J, const int64_t* Nj, int D, int K, const double* mu, const double* L,
                           const double* cdf, uint64_t seed, const int64_t* group_ids, int64_t row_offset,
                           double hard) {
  use_device();
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  build_layout(J, Nj, D);
  if (DP_ > 128) throw std::invalid_argument("the synthetic-data generator stops at D = 128");
  DevBuf<double> dmu, dL, dcdf;
  DevBuf<int64_t> dgid;
  if (group_ids) {
    dgid.reserve((size_t)J);
    LC_HIP(hipMemcpyAsync(dgid.p, group_ids, (size_t)J * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  }
  dmu.reserve((size_t)K * D);
  dL.reserve((size_t)K * D * D);
  LC_HIP(hipMemcpyAsync(dmu.p, mu, (size_t)K * D * sizeof(double), hipMemcpyHostToDevice, stream_));
  LC_HIP(hipMemcpyAsync(dL.p, L, (size_t)K * D * D * sizeof(double), hipMemcpyHostToDevice, stream_));
  if (cdf) {
    dcdf.reserve((size_t)J * K);
    LC_HIP(hipMemcpyAsync(dcdf.p, cdf, (size_t)J * K * sizeof(double), hipMemcpyHostToDevice, stream_));
  }
  ensure_qz(qz_[cur_], K, false);
  qz_[cur_].K = K;
  lck::SynthLaunch a;
  a.DP = DP_;
  a.D = D;
  a.K = K;
  a.X = X_.p;
  a.qZ = qz_[cur_].buf.p;
  a.ldq = NP_;
  a.nrows = Nj[0];
  a.NP = NP_;
  a.row_offset = row_offset;
  a.seed = seed;
  a.mu = dmu.p;
  a.L = dL.p;
  a.hard = hard;
  a.rginfo = J > 1 ? rginfo_.p : nullptr;
  a.goff = J > 1 ? goff_d_.p : nullptr;
  a.cdf = cdf ? dcdf.p : nullptr;
  a.gids = group_ids ? dgid.p : nullptr;
  a.group_base = 0;
  LC_HIP(lck::launch_synth(a, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::get_rows(int j, int64_t row0, int64_t n, double* out) const {
  use_device();
  if (j < 0 || j >= J_ || row0 < 0 || n < 0 || row0 + n > Nj_[j]) throw std::invalid_argument("row range out of bounds");
  if (n == 0) return;
  std::vector<double> tmp((size_t)n * DP_);
  LC_HIP(hipMemcpyAsync(tmp.data(), X_.p + (size_t)(goff_[j] + row0) * DP_, tmp.size() * sizeof(double),
                        hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
  for (int64_t r = 0; r < n; ++r) std::memcpy(out + r * D_, tmp.data() + (size_t)r * DP_, D_ * sizeof(double));
}

// ---------------------------------------------------------------------------
// qZ
// ---------------------------------------------------------------------------
#ifdef LC_TEST_HOOKS
// tests (LC_TEST_QHASH_KEEP_STALE): the writers of qZ "forget" what they owe the row fingerprints -- ensure_qz and
// qz_keep_columns leave hash_ok standing, qz_split_from does not mark the rows it rewrites.  The fingerprint check of
// LC_TEST_VERIFY_QHASH (estep_cache) has to notice.
static bool test_keep_stale() {
  static const bool on = std::getenv("LC_TEST_QHASH_KEEP_STALE") != nullptr;
  return on;
}
#endif
void Context::ensure_qz(QZ& q, int K, bool preserve) {
  use_device();
  q.hash_ok = false;  // (whoever asks for the buffer is about to write it; estep_cache and the split helpers re-validate)
#ifdef LC_TEST_HOOKS
  if (test_keep_stale() && q.hash.p) q.hash_ok = true;
#endif
  if (K <= q.cap && q.buf.p) return;
  int newcap = std::max(K, q.cap > 0 ? q.cap + std::max(4, q.cap / 2) : K);
  DevBuf<double> nb;
  nb.reserve((size_t)std::max<int64_t>(NP_, 1) * newcap);
  LC_HIP(hipMemsetAsync(nb.p, 0, (size_t)std::max<int64_t>(NP_, 1) * newcap * sizeof(double), stream_));
  if (preserve && q.buf.p && q.K > 0)
    LC_HIP(hipMemcpyAsync(nb.p, q.buf.p, (size_t)NP_ * q.K * sizeof(double), hipMemcpyDeviceToDevice, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
  std::swap(q.buf.p, nb.p);
  std::swap(q.buf.cap, nb.cap);
  std::swap(q.buf.device, nb.device);
  q.cap = newcap;
}

void Context::qz_fill(int K, double value) {
  use_device();
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  ensure_qz(qz_[cur_], K, false);
  qz_[cur_].K = K;
  LC_HIP(lck::launch_fill_qz(qz_[cur_].buf.p, NP_, K, rginfo_.p, Nj_[0], NP_ / lck::RG, value, stream_));
}

void Context::qz_set(int j, const double* q, int K, int64_t rs, int64_t cs) {
  use_device();
  if (j < 0 || j >= J_) throw std::invalid_argument("group index out of range");
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  qz_[cur_].hash_ok = false;
  if (K != qz_[cur_].K) {
    ensure_qz(qz_[cur_], K, false);
    // a new K invalidates every group: start from zeros
    LC_HIP(hipMemsetAsync(qz_[cur_].buf.p, 0, (size_t)NP_ * K * sizeof(double), stream_));
    qz_[cur_].K = K;
  }
  const int64_t n = Nj_[j];
  if (n == 0) return;
  std::vector<double> col((size_t)n);
  for (int k = 0; k < K; ++k) {
    for (int64_t r = 0; r < n; ++r) col[(size_t)r] = q[r * rs + k * cs];
    LC_HIP(hipMemcpyAsync(qz_[cur_].buf.p + (size_t)k * NP_ + goff_[j], col.data(), n * sizeof(double),
                          hipMemcpyHostToDevice, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
  }
}

void Context::qz_get_column(int j, int k, double* out) const {
  use_device();
  if (j < 0 || j >= J_ || k < 0 || k >= qz_[cur_].K) throw std::invalid_argument("qZ column out of range");
  if (Nj_[j] == 0) return;
  LC_HIP(hipMemcpyAsync(out, qz_[cur_].buf.p + (size_t)k * NP_ + goff_[j], Nj_[j] * sizeof(double),
                        hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::qz_get_rows(int j, int64_t row0, int64_t n, double* q, int64_t rs, int64_t cs) const {
  use_device();
  if (j < 0 || j >= J_ || row0 < 0 || n < 0 || row0 + n > Nj_[j]) throw std::invalid_argument("row range out of bounds");
  if (n == 0) return;
  std::vector<double> col((size_t)n);
  for (int k = 0; k < qz_[cur_].K; ++k) {
    LC_HIP(hipMemcpyAsync(col.data(), qz_[cur_].buf.p + (size_t)k * NP_ + goff_[j] + row0, n * sizeof(double),
                          hipMemcpyDeviceToHost, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
    for (int64_t r = 0; r < n; ++r) q[r * rs + k * cs] = col[(size_t)r];
  }
}

void Context::qz_get(int j, double* q, int64_t rs, int64_t cs) const {
  if (j < 0 || j >= J_) throw std::invalid_argument("group index out of range");
  const int64_t n = Nj_[j];
  std::vector<double> col((size_t)n);
  for (int k = 0; k < qz_[cur_].K; ++k) {
    qz_get_column(j, k, col.data());
    for (int64_t r = 0; r < n; ++r) q[r * rs + k * cs] = col[(size_t)r];
  }
}

// all groups at once: out is [Ntotal x K] row-major, the groups' rows concatenated (one device-to-host copy of the
// whole K x NP buffer instead of K copies per group: 20 000 small documents took 4.3 s the other way)
void Context::qz_get_all(double* out) const {
  const int K = qz_[cur_].K;
  if (K < 1 || NP_ == 0) return;
  LC_HIP(hipSetDevice(device_));
  {
    // transpose on the device (row-major [NP x K], pad rows included), bring it over in 32 MB pieces through two
    // alternating page-locked buffers, and copy the valid row ranges of every group out on the worker pool
    DevBuf<double> qT;
    qT.reserve((size_t)NP_ * K);
    LC_HIP(lck::launch_transpose_qz(qz_[cur_].buf.p, NP_, K, NP_, qT.p, stream_));
    std::vector<int64_t> ooff((size_t)J_ + 1, 0);  // first output row of every group
    for (int j = 0; j < J_; ++j) ooff[(size_t)j + 1] = ooff[(size_t)j] + Nj_[(size_t)j];
    const int64_t chunk = std::max<int64_t>(lck::RG, ((int64_t)32 << 20) / 8 / K / lck::RG * lck::RG);
    const unsigned nthr = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    PinnedBuf stage[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    auto unpack = [&](const double* base, int64_t p0, int64_t nr) {
      const int nblk = (int)((nr + 4095) / 4096);
      lce::parallel_chunks(nblk, nthr, 4096.0 * K * 8.0, [&](int blk) {
        const int64_t pa = p0 + (int64_t)blk * 4096, pb = std::min<int64_t>(pa + 4096, p0 + nr);
        int jj = (int)(std::upper_bound(goff_.begin(), goff_.end(), pa) - goff_.begin()) - 1;
        if (jj < 0) jj = 0;
        for (; jj < J_ && goff_[(size_t)jj] < pb; ++jj) {
          const int64_t g0 = goff_[(size_t)jj];
          const int64_t lo = std::max(pa, g0), hi = std::min(pb, g0 + Nj_[(size_t)jj]);
          if (lo < hi)
            std::memcpy(out + (size_t)(ooff[(size_t)jj] + lo - g0) * K, base + (size_t)(lo - p0) * K,
                        (size_t)(hi - lo) * K * sizeof(double));
        }
      });
    };
    int64_t prev0 = -1, prevn = 0;
    int which = 0;
    for (int64_t p0 = 0; p0 < NP_; p0 += chunk, which ^= 1) {
      const int64_t nr = std::min(chunk, NP_ - p0);
      PinnedBuf& st = stage[which];
      if (st.size() == 0) st.resize((size_t)chunk * K);
      if (!done[which]) LC_HIP(hipEventCreateWithFlags(&done[which], hipEventDisableTiming));
      LC_HIP(hipMemcpyAsync(st.data(), qT.p + (size_t)p0 * K, (size_t)nr * K * sizeof(double), hipMemcpyDeviceToHost,
                            stream_));
      LC_HIP(hipEventRecord(done[which], stream_));
      if (prev0 >= 0) {  // unpack the previous piece while this one is in flight
        LC_HIP(hipEventSynchronize(done[which ^ 1]));
        unpack(stage[which ^ 1].data(), prev0, prevn);
      }
      prev0 = p0;
      prevn = nr;
    }
    which ^= 1;  // the piece issued last
    LC_HIP(hipEventSynchronize(done[which]));
    unpack(stage[which].data(), prev0, prevn);
    for (auto& e : done)
      if (e) (void)hipEventDestroy(e);
  }
}

// Every group at once in the layout Eigen callers hold (one column-major N_j x K matrix per group): the device buffer
// is column-major already, so no transpose anywhere -- the flat [K x NP] array crosses in 32 MB pieces through two
// alternating page-locked buffers and pool threads copy the runs (column k, valid rows of group j) to their places
// while the next piece is in flight.
void Context::qz_get_all_colmajor(double* const* out, const int64_t* ld) const {
  const int K = qz_[cur_].K;
  if (K < 1 || NP_ == 0) return;
  LC_HIP(hipSetDevice(device_));
  const int64_t total = (int64_t)K * NP_, chunk = ((int64_t)32 << 20) / 8;
  const unsigned nthr = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
  PinnedBuf stage[2];
  hipEvent_t done[2] = {nullptr, nullptr};
  auto unpack = [&](const double* base, int64_t f0, int64_t n) {
    const int64_t blk = 262144;
    lce::parallel_chunks((int)((n + blk - 1) / blk), nthr, (double)blk * 8.0, [&](int b) {
      int64_t f = f0 + (int64_t)b * blk;
      const int64_t fe = std::min(f + blk, f0 + n);
      while (f < fe) {
        const int64_t k = f / NP_, p = f - k * NP_;              // column k, padded row p
        const int64_t pe = std::min<int64_t>(NP_, p + (fe - f));  // this column's rows inside the block
        int jj = (int)(std::upper_bound(goff_.begin(), goff_.end(), p) - goff_.begin()) - 1;
        if (jj < 0) jj = 0;
        for (; jj < J_ && goff_[(size_t)jj] < pe; ++jj) {
          const int64_t g0 = goff_[(size_t)jj], nj = Nj_[(size_t)jj];
          const int64_t lo = std::max(p, g0), hi = std::min(pe, g0 + nj);
          if (lo < hi)
            std::memcpy(out[jj] + (size_t)k * (ld ? ld[jj] : nj) + (lo - g0), base + (f - f0) + (lo - p),
                        (size_t)(hi - lo) * sizeof(double));
        }
        f += pe - p;
      }
    });
  };
  int64_t prev0 = -1, prevn = 0;
  int which = 0;
  for (int64_t f0 = 0; f0 < total; f0 += chunk, which ^= 1) {
    const int64_t n = std::min(chunk, total - f0);
    PinnedBuf& st = stage[which];
    if (st.size() == 0) st.resize((size_t)std::min(chunk, total));
    if (!done[which]) LC_HIP(hipEventCreateWithFlags(&done[which], hipEventDisableTiming));
    LC_HIP(hipMemcpyAsync(st.data(), qz_[cur_].buf.p + f0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    LC_HIP(hipEventRecord(done[which], stream_));
    if (prev0 >= 0) {
      LC_HIP(hipEventSynchronize(done[which ^ 1]));
      unpack(stage[which ^ 1].data(), prev0, prevn);
    }
    prev0 = f0;
    prevn = n;
  }
  which ^= 1;
  LC_HIP(hipEventSynchronize(done[which]));
  unpack(stage[which].data(), prev0, prevn);
  for (auto& e : done)
    if (e) (void)hipEventDestroy(e);
}

void Context::qz_keep_columns(const std::vector<int>& keep) {
  use_device();
  QZ& q = qz_[cur_];
  for (size_t i = 0; i < keep.size(); ++i) {
    const int src = keep[i];
    if (src < (int)i || src >= q.K) throw std::invalid_argument("bad column list");
    if (src != (int)i)
      LC_HIP(hipMemcpyAsync(q.buf.p + i * (size_t)NP_, q.buf.p + (size_t)src * NP_, NP_ * sizeof(double),
                            hipMemcpyDeviceToDevice, stream_));
  }
  q.K = (int)keep.size();
  q.hash_ok = false;  // (columns moved: the rows' fingerprints weigh every value by its column)
#ifdef LC_TEST_HOOKS
  if (test_keep_stale() && q.hash.p) q.hash_ok = true;
#endif
}

void Context::qz_clone_to_alt() {
  use_device();
  QZ& a = qz_[cur_ ^ 1];
  const int Ksave = qz_[cur_].K;
  // The copy a rejected candidate left behind differs from the original in that candidate's rows only: with fingerprints
  // on both sides (the moved-row sweeps keep them) only those rows are copied (qz_resync_kernel) -- 0.6 ms instead of
  // the 1.2-1.7 ms a clone of 2.6 GB takes, once per candidate (N = 10M, K = 32: the candidate's 312 k rows lie scattered).  LC_SPLIT_NO_QHASH / LC_SPLIT_FULL_CLONE (tests): always the
  // full copy; the two must agree in every bit (tests/test_gpu_splitsearch.py).
  static const bool full_only = lck::test_switch("LC_SPLIT_FULL_CLONE") != nullptr;
  if (!full_only && NP_ > 0 && Ksave >= 1 && qz_[cur_].hash_ok && a.hash_ok && a.buf.p && a.hash.p && qz_[cur_].hash.p &&
      a.cap >= Ksave + 1 && (a.K == Ksave || a.K == Ksave + 1)) {
    LC_HIP(lck::launch_qz_resync(a.buf.p, qz_[cur_].buf.p, NP_, Ksave, Ksave + 1, NP_, a.hash.p, qz_[cur_].hash.p, stream_));
    a.K = Ksave;
    clone_resyncs_ += 1;
    return;
  }
  clone_fulls_ += 1;
  static const bool trace = std::getenv("LC_TRACE_PHASES") != nullptr;
  if (trace) std::cerr << "[clone] full copy at K " << Ksave << " (row-wise so far: " << clone_resyncs_ << ")" << std::endl;
  ensure_qz(a, Ksave + 1, false);
  a.K = Ksave;
  LC_HIP(hipMemcpyAsync(a.buf.p, qz_[cur_].buf.p, (size_t)NP_ * Ksave * sizeof(double), hipMemcpyDeviceToDevice,
                        stream_));
  LC_HIP(hipMemsetAsync(a.buf.p + (size_t)NP_ * Ksave, 0, (size_t)NP_ * sizeof(double), stream_));
  if (qz_[cur_].hash_ok && NP_ > 0) {  // the copy is what the original is: so are its rows' fingerprints
    a.hash.reserve((size_t)NP_);
    LC_HIP(hipMemcpyAsync(a.hash.p, qz_[cur_].hash.p, (size_t)NP_ * sizeof(int64_t), hipMemcpyDeviceToDevice, stream_));
    a.hash_ok = true;
  }
}

void Context::qz_swap_alt() { cur_ ^= 1; }

// ---------------------------------------------------------------------------
// split search on the device
// ---------------------------------------------------------------------------
void Context::select_rows(int k, double thresh, RowSelection& sel) {
  const QZ& q = qz_[cur_];
  if (k < 0 || k >= q.K) throw std::invalid_argument("qZ column out of range");
  select_rows_col(q.buf.p + (size_t)k * NP_, thresh, sel);
}

void Context::select_rows_col(const double* col, double thresh, RowSelection& sel) {
  LC_HIP(hipSetDevice(device_));
  sel.M = 0;
  sel.starts.assign((size_t)J_ + 1, 0);
  sel.starts_d.reserve((size_t)J_ + 1);
  if (NP_ == 0) {
    LC_HIP(hipMemsetAsync(sel.starts_d.p, 0, (size_t)(J_ + 1) * sizeof(int64_t), stream_));
    return;
  }
  const int nb = lck::select_blocks(NP_);
  selcnt_.reserve((size_t)nb);
  seloff_.reserve((size_t)nb);
  LC_HIP(lck::launch_select_count(col, NP_, thresh, selcnt_.p, stream_));
  std::vector<int> cnt((size_t)nb);
  LC_HIP(hipMemcpyAsync(cnt.data(), selcnt_.p, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
  std::vector<int64_t> off((size_t)nb);
  int64_t tot = 0;
  for (int b = 0; b < nb; ++b) {
    off[(size_t)b] = tot;
    tot += cnt[(size_t)b];
  }
  sel.M = tot;
  sel.idx.reserve((size_t)std::max<int64_t>(tot, 1));
  LC_HIP(hipMemcpyAsync(seloff_.p, off.data(), (size_t)nb * sizeof(int64_t), hipMemcpyHostToDevice, stream_));
  LC_HIP(lck::launch_select_compact(col, NP_, thresh, seloff_.p, sel.idx.p, stream_));
  LC_HIP(lck::launch_group_starts(sel.idx.p, tot, goff_d_.p, J_, sel.starts_d.p, stream_));
  LC_HIP(hipMemcpyAsync(sel.starts.data(), sel.starts_d.p, (size_t)(J_ + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                        stream_));
  LC_HIP(hipStreamSynchronize(stream_));  // `off` goes out of scope; starts are needed by the caller
}

void Context::set_data_gather(const Context& src, const RowSelection& sel) {
  use_device();
  if (src.device_ != device_) throw std::invalid_argument("contexts live on different devices");
  std::vector<int64_t> mj((size_t)src.J_);
  for (int j = 0; j < src.J_; ++j) mj[(size_t)j] = sel.starts[(size_t)j + 1] - sel.starts[(size_t)j];
  {
    RelaxedFit lend;  // (a sub-problem: short-lived buffers)
    build_layout(src.J_, mj.data(), src.D_);
  }
  if (NP_ == 0) return;
  LC_HIP(hipMemsetAsync(X_.p, 0, (size_t)NP_ * DP_ * sizeof(double), stream_));
  LC_HIP(lck::launch_gather_rows(src.X_.p, DP_, sel.idx.p, sel.M, sel.starts_d.p, goff_d_.p, J_, X_.p, stream_));
}

void Context::qz_gather_column(const Context& src, const RowSelection& sel, int col) {
  use_device();
  if (src.device_ != device_ || src.J_ != J_) throw std::invalid_argument("not a sub-problem of that context");
  const QZ& sq = src.qz_[src.cur_];
  if (col < 0 || col >= sq.K) throw std::invalid_argument("qZ column out of range");
  QZ& q = qz_[cur_];
  ensure_qz(q, 1, false);
  q.K = 1;
  if (NP_ == 0) return;
  LC_HIP(hipMemsetAsync(q.buf.p, 0, (size_t)NP_ * sizeof(double), stream_));  // padding rows carry nothing
  LC_HIP(lck::launch_gather_cols(sq.buf.p + (size_t)col * src.NP_, src.NP_, 1, sel.idx.p, sel.M, sel.starts_d.p, goff_d_.p,
                                 J_, q.buf.p, NP_, stream_));
}

void Context::qz_init_split(const double* m, const double* v) {
  use_device();
  ensure_qz(qz_[cur_], 2, false);
  qz_[cur_].K = 2;
  if (NP_ == 0) return;
  std::vector<double> mv((size_t)2 * DP_, 0.0);
  std::copy(m, m + D_, mv.begin());
  std::copy(v, v + D_, mv.begin() + DP_);
  mv_.reserve(mv.size());
  LC_HIP(hipMemcpyAsync(mv_.p, mv.data(), mv.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  LC_HIP(lck::launch_split_init(X_.p, DP_, D_, NP_, J_ > 1 ? rginfo_.p : nullptr, Nj_[0], mv_.p, qz_[cur_].buf.p, NP_,
                                0, nullptr, stream_));
  LC_HIP(hipStreamSynchronize(stream_));  // mv is a local
}

void Context::qz_init_split_mean(const double* v) {
  use_device();
  ensure_qz(qz_[cur_], 2, false);
  qz_[cur_].K = 2;
  // rows of one group may be spread over ranks (row sharding): the group means are global quantities, so every
  // rank takes part in the exchange, also one that holds no selected row
  const bool global_mean = distributed() && !group_sharded_;
  std::vector<double> mv((size_t)2 * DP_, 0.0);
  std::copy(v, v + D_, mv.begin() + DP_);
  std::vector<double> sc((size_t)J_ * 2, 0.0);  // [sum of projections | row count] per group
  const int* rgi = J_ > 1 ? rginfo_.p : nullptr;
  if (NP_ > 0) {
    mv_.reserve(mv.size() + (size_t)J_);
    LC_HIP(hipMemcpyAsync(mv_.p, mv.data(), mv.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    // pass 1: the projections x.v land in column 0; their per-group sums give the means
    LC_HIP(lck::launch_split_init(X_.p, DP_, D_, NP_, rgi, Nj_[0], mv_.p, qz_[cur_].buf.p, NP_, 1, nullptr, stream_));
    std::vector<double> sums((size_t)J_ * 2);
    red_.reserve((size_t)J_ * 2);
    LC_HIP(lck::launch_group_colsum(qz_[cur_].buf.p, NP_, 2, goff_d_.p, J_, red_.p, stream_));
    LC_HIP(hipMemcpyAsync(sums.data(), red_.p, sums.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
    for (int j = 0; j < J_; ++j) sc[(size_t)j] = sums[(size_t)j * 2];
  }
  for (int j = 0; j < J_; ++j) sc[(size_t)J_ + j] = (double)Nj_[j];
  if (global_mean) allreduce_values(sc.data(), 2 * J_);
  if (NP_ == 0) return;
  std::vector<double> thr((size_t)J_);
  for (int j = 0; j < J_; ++j) thr[(size_t)j] = sc[(size_t)J_ + j] > 0 ? sc[(size_t)j] / sc[(size_t)J_ + j] : 0.0;  // XdotL.sum()/size
  double* thr_d = mv_.p + mv.size();
  LC_HIP(hipMemcpyAsync(thr_d, thr.data(), thr.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  LC_HIP(lck::launch_split_init(X_.p, DP_, D_, NP_, rgi, Nj_[0], mv_.p, qz_[cur_].buf.p, NP_, 2, thr_d, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::qz_split_from(const Context& sub, const RowSelection& sel, int k) {
  use_device();
  QZ& q = qz_[cur_];
  if (k < 0 || k >= q.K) throw std::invalid_argument("split column out of range");
  if (sub.qz_[sub.cur_].K < 2) throw std::invalid_argument("sub-problem has no second column");
  const bool hashed = q.hash_ok;  // (a new zero column leaves a row's fingerprint as it is; rewritten rows are marked)
  if (q.K + 1 > q.cap) ensure_qz(q, q.K + 1, true);
  q.hash_ok = hashed;
  bool mark = hashed;
#ifdef LC_TEST_HOOKS
  if (test_keep_stale()) mark = false;
#endif
  LC_HIP(hipMemsetAsync(q.buf.p + (size_t)NP_ * q.K, 0, (size_t)NP_ * sizeof(double), stream_));
  LC_HIP(lck::launch_aug_from_sub(q.buf.p, NP_, k, q.K, sel.idx.p, sel.M, sel.starts_d.p, sub.goff_d_.p, J_,
                                  sub.qz_[sub.cur_].buf.p + (size_t)sub.NP_, stream_, mark ? q.hash.p : nullptr));
  q.K += 1;
}

// ---------------------------------------------------------------------------
// hot path
// ---------------------------------------------------------------------------
void Context::allreduce(double* dbuf, int64_t count) {
  if (!comm_ && !ar_fn_) return;
  EvPair ev{};
  if (timing_) {  // events around the exchange step: the sum plus the wait for the slowest rank
    ev.a = timing_event();
    ev.b = timing_event();
    ev.kind = 3;
    LC_HIP(hipEventRecord(ev.a, stream_));
  }
  if (comm_) {  // RCCL / host-staged sum on this context's stream
    comm_->allreduce_sum(dbuf, count, stream_);
  } else {
    const int rc = ar_fn_(ar_user_, dbuf, count, (void*)stream_);
    if (rc != 0) throw std::runtime_error("all-reduce hook failed with status " + std::to_string(rc));
  }
  if (timing_) {
    LC_HIP(hipEventRecord(ev.b, stream_));
    pending_.push_back(ev);
  }
}

double Context::allreduce_value(double v) {
  allreduce_values(&v, 1);
  return v;
}

void Context::allreduce_values(double* v, int n) {
  if (!distributed() || n <= 0) return;
  LC_HIP(hipSetDevice(device_));
  red_.reserve((size_t)n);
  LC_HIP(hipMemcpyAsync(red_.p, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream_));
  allreduce(red_.p, n);
  LC_HIP(hipMemcpyAsync(v, red_.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::pack_estep_params(int K, const double* A, const double* m, const double* c) {
  const int D = D_, DP = DP_, DC = DC_, NT = DC / 4;  // (narrow layouts: the tile rows of the active width)
  const bool wide = DP > 128;
  const int NTILES = wide ? 0 : lck::ntiles(DC);
  const int64_t PS = lck::estep_pstride(DP, DC);
  hpack_.assign((size_t)K * PS + (size_t)J_ * K, 0.0);
  std::vector<double> bneg((size_t)DP);
  for (int k = 0; k < K; ++k) {
    const double* Ak = A + (size_t)k * D * D;
    const double* mk = m + (size_t)k * D;
    double* P = hpack_.data() + (size_t)k * PS;
    std::fill(bneg.begin(), bneg.end(), 0.0);
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
      for (int j = 0; j <= i; ++j) s += Ak[(size_t)i * D + j] * mk[j];
      bneg[(size_t)i] = -s;  // the kernel's accumulators start at -b so that y = A x - b
    }
    auto fill_tile = [&](double* T, int i0, int j0) {  // element (lo, hi) of a 4x4 tile = A[i0 + lo][j0 + hi]
      for (int hi = 0; hi < 4; ++hi)
        for (int lo = 0; lo < 4; ++lo) {
          const int i = i0 + lo, j = j0 + hi;
          T[lo + 4 * hi] = (i < D && j <= i) ? Ak[(size_t)i * D + j] : 0.0;
        }
    };
    if (!wide) {
      // ---- tiles of A_k in consumption order, then -b_k = -A_k m_k ------
      for (int it = 0; it < NT; ++it)
        for (int jt = 0; jt <= it; ++jt) fill_tile(P + (size_t)(it * (it + 1) / 2 + jt) * 16, 4 * it, 4 * jt);
      std::copy(bneg.begin(), bneg.begin() + DC, P + (size_t)NTILES * 16);
    } else {
      // ---- 64 x 64 blocks (I, J <= I), row-major; per chunk 16 x 16 tiles, then -b_I (estep_wide_kernel) ------
      double* C = P;
      for (int I = 0; I < DP / 64; ++I)
        for (int Jb = 0; Jb <= I; ++Jb, C += lck::WIDE_CHUNK) {
          // tile column jt = 4 q + jr holds the columns {16 q + jr + 4 h}: estep_wide_kernel's column groups (wide_col)
          for (int it = 0; it < 16; ++it)
            for (int jt = 0; jt < 16; ++jt) {
              double* T = C + (size_t)(it * 16 + jt) * 16;
              for (int h = 0; h < 4; ++h)
                for (int lo = 0; lo < 4; ++lo) {
                  const int i = 64 * I + 4 * it + lo, j = 64 * Jb + 16 * (jt / 4) + 4 * h + (jt % 4);
                  T[lo + 4 * h] = (i < D && j <= i) ? Ak[(size_t)i * D + j] : 0.0;
                }
            }
          std::copy(bneg.begin() + 64 * I, bneg.begin() + 64 * (I + 1), C + 4096);
        }
    }
  }
  std::memcpy(hpack_.data() + (size_t)K * PS, c, (size_t)J_ * K * sizeof(double));
}

void Context::estep(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk, bool raw,
                    double* target) {
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  if (target && !raw) throw std::invalid_argument("a target buffer needs raw mode");
  require_gw_width();  // (an upper limit only: wide observations stream through estep_wide_kernel)
  if (NP_ == 0 && !distributed()) {
    if (Fz) *Fz = -0.0;
    if (LLk) std::fill(LLk, LLk + K, 0.0);
    if (!target) qz_[cur_].K = K;
    return;
  }
  LC_HIP(hipSetDevice(device_));
  const int DP = DP_;
  const int64_t PS = lck::estep_pstride(DP, DC_);
  pack_estep_params(K, A, m, c);
  params_.reserve(hpack_.size());
  LC_HIP(hipMemcpyAsync(params_.p, hpack_.data(), hpack_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));

  if (!target) {
    ensure_qz(qz_[cur_], K, false);  // E-step overwrites every column
    qz_[cur_].K = K;
  }
  const int64_t nrg = NP_ / lck::RG;
  lck::EstepLaunch a;
  a.DP = DP;
  a.DC = DC_;
  a.X = X_.p;
  a.nrg = nrg;
  a.rginfo = J_ > 1 ? rginfo_.p : nullptr;
  a.nrows = Nj_[0];
  a.params = params_.p;
  a.ctab = params_.p + (size_t)K * PS;
  a.K = K;
  a.qZ = target ? target : qz_[cur_].buf.p;
  a.ldq = NP_;
  a.raw = raw ? 1 : 0;
  for (size_t t = 0; t < (size_t)J_ * K && !a.sparse; ++t)
    if (c[t] == -std::numeric_limits<double>::infinity()) a.sparse = 1;
  const int64_t grid = lck::estep_grid(a);  // (a function of the whole launch: shape, raw, sparse)
  fzpart_.reserve((size_t)std::max<int64_t>(grid, 1));
  llpart_.reserve((size_t)std::max<int64_t>(grid, 1) * K);
  red_.reserve((size_t)1 + K);
  a.fz_part = fzpart_.p;
  a.ll_part = LLk ? llpart_.p : nullptr;
  EvPair ev{};
  if (timing_) {
    ev.a = timing_event();
    ev.b = timing_event();
    ev.kind = 0;
    LC_HIP(hipEventRecord(ev.a, stream_));
  }
  LC_HIP(lck::launch_estep(a, stream_));
  if (timing_) {
    LC_HIP(hipEventRecord(ev.b, stream_));
    pending_.push_back(ev);
  }
  if (raw) {
    LC_HIP(hipStreamSynchronize(stream_));
    return;
  }
  hred_.resize((size_t)1 + K);
  constexpr bool direct_env = true;  // (the copy-back command instead: measured slower)
  if (direct_env && !distributed() && grid > 0) {
    // nothing to sum over ranks: the folds write F_z (and LL_k) straight into the page-locked host buffer
    redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
    LC_HIP(lck::launch_reduce_partials(fzpart_.p, (int)grid, 1, hred_.data(), stream_, redtmp_.p));
    if (LLk) LC_HIP(lck::launch_reduce_partials(llpart_.p, (int)grid, K, hred_.data() + 1, stream_, redtmp_.p));
    run_overlap();
    LC_HIP(hipStreamSynchronize(stream_));
    if (Fz) *Fz = hred_[0];
    if (LLk) std::copy(hred_.begin() + 1, hred_.end(), LLk);
    return;
  }
  if (grid > 0) {
    redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
    LC_HIP(lck::launch_reduce_partials(fzpart_.p, (int)grid, 1, red_.p, stream_, redtmp_.p));
    if (LLk) LC_HIP(lck::launch_reduce_partials(llpart_.p, (int)grid, K, red_.p + 1, stream_, redtmp_.p));
    else LC_HIP(hipMemsetAsync(red_.p + 1, 0, (size_t)K * sizeof(double), stream_));
  } else {
    LC_HIP(hipMemsetAsync(red_.p, 0, (size_t)(1 + K) * sizeof(double), stream_));
  }
  allreduce(red_.p, 1 + K);
  LC_HIP(hipMemcpyAsync(hred_.data(), red_.p, (size_t)(1 + K) * sizeof(double), hipMemcpyDeviceToHost, stream_));
  run_overlap();
  LC_HIP(hipStreamSynchronize(stream_));
  if (Fz) *Fz = hred_[0];
  if (LLk) std::copy(hred_.begin() + 1, hred_.end(), LLk);
}

bool Context::estep_suffstat_fused(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk,
                                   double* Nk, double* xs, double* xxs, double* Njk) {
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  const int D = D_, DP = DP_;
  const int64_t nrg = NP_ / lck::RG;
  if (!lck::fused_eligible(DP, K)) return false;
  const int grid = NP_ > 0 ? lck::fused_plan(DP, nrg, K) : 0;  // (0: this rank holds no rows; it still joins the sums)
  LC_HIP(hipSetDevice(device_));
  const int64_t PS = lck::pstride(DP), SS = lck::stat_stride(DP);
  pack_estep_params(K, A, m, c);
  params_.reserve(hpack_.size());
  LC_HIP(hipMemcpyAsync(params_.p, hpack_.data(), hpack_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  ensure_qz(qz_[cur_], K, false);  // the E-step overwrites every column
  qz_[cur_].K = K;
  const int64_t W = lck::fused_record(DP, K);
  sspart_reserve((size_t)std::max(grid, 1) * W);
  sspart_clean_ = false;  // (another record layout)
  // one buffer for everything that is summed over ranks and copied back: [K records | Fz | LL_k | J x K counts]
  // (the first three are the fold of the kernel's per-block records: one reduction launch)
  // (single group without group sharding: the counts ARE the N_k of the records -- no count block at all)
  const bool own_counts = J_ > 1 || group_sharded();
  const size_t nrec = (size_t)W, nout = nrec + (own_counts ? (size_t)J_ * K : 0);
  const size_t ofz = (size_t)K * SS, oll = ofz + 1;
  ssout_.reserve(nout);
  double* njk_d = ssout_.p + nrec;
  lck::FusedLaunch a;
  a.DP = DP;
  a.D = D;
  a.X = X_.p;
  a.nrg = nrg;
  a.rginfo = J_ > 1 ? rginfo_.p : nullptr;
  a.nrows = Nj_[0];
  a.params = params_.p;
  a.ctab = params_.p + (size_t)K * PS;
  a.K = K;
  a.qZ = qz_[cur_].buf.p;
  a.ldq = NP_;
  a.partial = sspart_.p;
  a.want_ll = LLk != nullptr;
  a.grid = grid;
  a.ngroups = J_;
  EvPair ev{};
  if (timing_) {
    ev.a = timing_event();
    ev.b = timing_event();
    ev.kind = 2;
    LC_HIP(hipEventRecord(ev.a, stream_));
  }
  if (grid > 0) LC_HIP(lck::launch_fused(a, stream_));
  if (timing_) {
    LC_HIP(hipEventRecord(ev.b, stream_));
    pending_.push_back(ev);
  }
  hss_.resize(nout);
  // nothing to sum over ranks: the fold writes straight into the pinned host buffer (coherent, device-visible
  // memory) and the copy-back command drops off the iteration's critical path
  constexpr bool direct_env = true;  // (the copy-back command instead: 0.258 against 0.247 ms per iteration in round 2)
  const bool direct = direct_env && !distributed() && grid > 0;
  double* dst = direct ? hss_.data() : ssout_.p;
  if (grid > 0) {
    LC_HIP(lck::launch_reduce_partials(sspart_.p, grid, W, dst, stream_));
    if (own_counts) {
      redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
      LC_HIP(lck::launch_group_colsum(qz_[cur_].buf.p, NP_, K, goff_d_.p, J_, direct ? dst + nrec : njk_d, stream_,
                                      redtmp_.p, NP_));
    }
  } else {
    LC_HIP(hipMemsetAsync(ssout_.p, 0, nout * sizeof(double), stream_));
  }
  if (!direct) {
    // whole groups per rank: the per-group counts stay local; rows sharded: they are summed with the rest
    allreduce(ssout_.p, (int64_t)(group_sharded_ ? nrec : nout));
    LC_HIP(hipMemcpyAsync(hss_.data(), ssout_.p, nout * sizeof(double), hipMemcpyDeviceToHost, stream_));
  }
  run_overlap();
  LC_HIP(hipStreamSynchronize(stream_));
  for (int k = 0; k < K; ++k) {
    const double* rec = hss_.data() + (size_t)k * SS;
    if (Nk) Nk[k] = rec[0];
    if (xs)
      for (int d = 0; d < D; ++d) xs[(size_t)k * D + d] = rec[1 + d];
    if (xxs) {
      const double* S = rec + 1 + DP;
      double* o = xxs + (size_t)k * D * D;
      for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {  // the lower triangle is authoritative
          o[(size_t)i * D + j] = S[(size_t)i * DP + j];
          o[(size_t)j * D + i] = S[(size_t)i * DP + j];
        }
    }
  }
  if (Njk) {
    if (!own_counts)
      for (int k = 0; k < K; ++k) Njk[k] = hss_[(size_t)k * SS];
    else
      std::copy(hss_.begin() + nrec, hss_.begin() + nout, Njk);
  }
  if (Fz) *Fz = hss_[ofz];
  if (LLk) std::copy(hss_.begin() + oll, hss_.begin() + oll + K, LLk);
  return true;
}

// Sparse statistics (cluster.cpp:67-79): turn the J x K activity mask into a work list of
// (32-row-aligned row range inside ONE group) x (slice of that group's ACTIVE clusters), upload it with the active
// cluster lists and the per-cluster record lists, and point the launch at it.  Returns the number of partial records
// (one per listed (row range, cluster) pair).
int Context::build_sparse_worklist(const unsigned char* smask, int K, int64_t SS, lck::SuffstatLaunch& a) {
  const int DP = DP_;
  int nrec = 0;
  const int cpb = lck::suffstat_clusters_per_block(DP, K);
  std::vector<int> klist, kofs((size_t)J_ + 1, 0);
  for (int j2 = 0; j2 < J_; ++j2) {
    for (int k2 = 0; k2 < K; ++k2)
      if (smask[(size_t)j2 * K + k2]) klist.push_back(k2);
    kofs[(size_t)j2 + 1] = (int)klist.size();
  }
  // row chunks: whole 32-row batches inside one group, about 2048 blocks in total (units = rows x slices)
  double units = 0.0;
  for (int j2 = 0; j2 < J_; ++j2)
    units += (double)(goff_[(size_t)j2 + 1] - goff_[(size_t)j2]) *
                      (double)((kofs[(size_t)j2 + 1] - kofs[(size_t)j2] + cpb - 1) / cpb);
  int64_t rows = (int64_t)(units / 2048.0);
  rows = std::max<int64_t>(256, (rows + 31) / 32 * 32);
  std::vector<lck::SSItem> items;
  std::vector<std::vector<int>> recs((size_t)K);
  for (int j2 = 0; j2 < J_; ++j2) {
    const int na = kofs[(size_t)j2 + 1] - kofs[(size_t)j2];
    if (na == 0) continue;
    for (int64_t b0 = goff_[(size_t)j2]; b0 < goff_[(size_t)j2 + 1]; b0 += rows) {
      const int64_t b1 = std::min<int64_t>(b0 + rows, goff_[(size_t)j2 + 1]);
      for (int s0 = 0; s0 < na; s0 += cpb) {
        lck::SSItem it;
        it.r0 = b0;
        it.r1 = b1;
        it.kofs = kofs[(size_t)j2] + s0;
        it.kcnt = std::min(cpb, na - s0);
        it.rec0 = nrec;
        for (int t = 0; t < it.kcnt; ++t) recs[(size_t)klist[(size_t)it.kofs + t]].push_back(nrec + t);
        nrec += it.kcnt;
        items.push_back(it);
      }
    }
  }
  std::vector<int> kptr((size_t)K + 1, 0), krec;
  krec.reserve((size_t)nrec);
  for (int k2 = 0; k2 < K; ++k2) {
    krec.insert(krec.end(), recs[(size_t)k2].begin(), recs[(size_t)k2].end());
    kptr[(size_t)k2 + 1] = (int)krec.size();
  }
  ssitems_.reserve(items.size() * sizeof(lck::SSItem));
  ssints_.reserve(klist.size() + kptr.size() + krec.size() + 1);
  int* klist_d = ssints_.p;
  int* kptr_d = klist_d + klist.size();
  int* krec_d = kptr_d + kptr.size();
  if (!items.empty()) {
    LC_HIP(hipMemcpyAsync(ssitems_.p, items.data(), items.size() * sizeof(lck::SSItem), hipMemcpyHostToDevice,
                          stream_));
    LC_HIP(hipMemcpyAsync(klist_d, klist.data(), klist.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    LC_HIP(hipMemcpyAsync(krec_d, krec.data(), krec.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  }
  LC_HIP(hipMemcpyAsync(kptr_d, kptr.data(), kptr.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
  LC_HIP(hipStreamSynchronize(stream_));  // the host vectors go out of scope
  a.items = reinterpret_cast<const lck::SSItem*>(ssitems_.p);
  a.klist = klist_d;
  a.nitems = (int)items.size();
  a.smask = nullptr;  // only active clusters are listed
  a.rginfo = nullptr;
  sskptr_ = kptr_d;
  sskrec_ = krec_d;
  sspart_reserve((size_t)std::max(nrec, 1) * SS);
  return nrec;
}

void Context::suffstat(const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk) {
  const int K = qz_[cur_].K, D = D_, DP = DP_;
  if (K < 1) throw std::invalid_argument("qZ has not been set");
  require_gw_width();
  LC_HIP(hipSetDevice(device_));
  const int64_t SS = lck::stat_stride(DP);
  const size_t nout = (size_t)K * SS + (size_t)J_ * K;
  ssout_.reserve(nout);
  double* njk_d = ssout_.p + (size_t)K * SS;
  const bool own_counts = J_ > 1 || group_sharded() || smask != nullptr;  // N_jk from a column-sum pass of their own
  // Nothing to sum over ranks, one group, no mask: the fold of the per-chunk records writes straight into the page-locked
  // host buffer (device-visible; complete when the stream synchronises) -- no fill and no copy-back command, which
  // count where a pass lasts 0.1 ms (the sub-problems of the split search run thousands of them)
  constexpr bool direct_env = true;  // (the copy-back command instead: measured slower)
  const bool direct = direct_env && !distributed() && !own_counts && NP_ > 0;
  hss_.resize(nout);
  if (NP_ > 0) {
    int64_t chunk_rows = 0;
    const int nchunks = lck::suffstat_plan(DP, NP_, K, &chunk_rows, DC_);
    sspart_reserve((size_t)nchunks * K * SS);
    lck::SuffstatLaunch a;
    a.DP = DP;
    a.DC = DC_;
    a.X = X_.p;
    a.NP = NP_;
    a.qZ = qz_[cur_].buf.p;
    a.ldq = NP_;
    a.K = K;
    a.rginfo = nullptr;
    a.smask = nullptr;
    if (smask && J_ > 1) {
      smask_.reserve((size_t)J_ * K);
      LC_HIP(hipMemcpyAsync(smask_.p, smask, (size_t)J_ * K, hipMemcpyHostToDevice, stream_));
      a.rginfo = rginfo_.p;
      a.smask = smask_.p;
    } else if (smask) {
      // single group: a masked cluster simply receives nothing (handled on the host below)
    }
    a.skip_zero = skip_zero_ ? 1 : 0;
    bool listed = false;  // sparse work list instead of the dense (chunk, slice) grid
    if (a.smask) {
      double off = 0.0, tot = 0.0;
      for (int j2 = 0; j2 < J_; ++j2)
        for (int k2 = 0; k2 < K; ++k2) {
          tot += (double)Nj_[j2];
          if (!smask[(size_t)j2 * K + k2]) off += (double)Nj_[j2];
        }
      // Sparse mode (cluster.cpp:67-79).  When the mask removes a fair share of the (row, cluster) pairs the pass
      // runs over a work list -- (row range of one group) x (slice of that group's ACTIVE clusters) -- so the
      // work is proportional to sum_j N_j * K_active(j); otherwise the dense grid with masked q staged as zeros.
      if (off > 0.3 * tot) {
        listed = true;
        build_sparse_worklist(smask, K, SS, a);  // (sizes the partial buffer for its records)
      } else if (!skip_zero_) {
        a.skip_zero = -1;  // dense variant; masked q are staged as zeros
      }
    }
    // ragged K (dense grid only): extra row-split records of the last cluster slice, folded in after the reduction
    int klast0 = K;
    const bool skipping = a.skip_zero > 0 || (a.skip_zero == 0 && a.smask);
    const int extra = listed ? 0 : lck::suffstat_extra_records(DP, K, skipping, &klast0, DC_);
    const int KR = K + extra;
    if (extra > 0) {
      sspart_reserve((size_t)nchunks * KR * SS);
      ssext_.reserve((size_t)KR * SS);
    }
    a.KR = KR;
    // active width below the padded one: the feature-GEMM kernel never writes the record entries of the idle columns --
    // they have to BE zero (every other Gauss-Wishart writer of this buffer stores zeros there; the separable families and
    // the fused pass, which keep other layouts in it, mark it dirty)
    if (DC_ < DP_ && !sspart_clean_) {
      LC_HIP(hipMemsetAsync(sspart_.p, 0, sspart_.cap * sizeof(double), stream_));
      sspart_clean_ = true;
    }
    a.partial = sspart_.p;
    a.nchunks = nchunks;
    a.chunk_rows = chunk_rows;
    EvPair ev{};
    if (timing_) {
      ev.a = timing_event();
      ev.b = timing_event();
      ev.kind = 1;
      LC_HIP(hipEventRecord(ev.a, stream_));
    }
    LC_HIP(lck::launch_suffstat(a, stream_));
    if (timing_) {
      LC_HIP(hipEventRecord(ev.b, stream_));
      pending_.push_back(ev);
    }
    if (listed)
      LC_HIP(lck::launch_reduce_records(sspart_.p, SS, K, sskptr_, sskrec_, ssout_.p, stream_));
    else if (extra > 0) {
      LC_HIP(lck::launch_reduce_partials(sspart_.p, nchunks, (int64_t)KR * SS, ssext_.p, stream_));
      LC_HIP(lck::launch_fold_extra(ssext_.p, SS, K, klast0, extra, stream_));
      LC_HIP(hipMemcpyAsync(direct ? hss_.data() : ssout_.p, ssext_.p, (size_t)K * SS * sizeof(double),
                            direct ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, stream_));
    } else
      LC_HIP(lck::launch_reduce_partials(sspart_.p, nchunks, (int64_t)K * SS, direct ? hss_.data() : ssout_.p, stream_));
    // per-group counts N_jk: with one group they are the N_k just reduced (filled in on the host below) -- unless the
    // records are about to be summed over ranks that hold OTHER groups, or a mask removes clusters from them
    if (own_counts) {
      redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
      LC_HIP(lck::launch_group_colsum(qz_[cur_].buf.p, NP_, K, goff_d_.p, J_, njk_d, stream_, redtmp_.p, NP_));
    }
    else if (!direct) LC_HIP(hipMemsetAsync(njk_d, 0, (size_t)K * sizeof(double), stream_));
    if (smask && J_ == 1)  // single group: a masked cluster receives nothing from it (cluster.cpp:67-70)
      for (int k = 0; k < K; ++k)
        if (!smask[k]) LC_HIP(hipMemsetAsync(ssout_.p + (size_t)k * SS, 0, (size_t)SS * sizeof(double), stream_));
  } else {
    LC_HIP(hipMemsetAsync(ssout_.p, 0, nout * sizeof(double), stream_));
  }
  // rows of the same groups on every rank: everything is summed.  Whole groups per rank: the K
  // cluster records are summed, the per-group counts are local by construction.
  if (!direct) {
    allreduce(ssout_.p, group_sharded_ ? (int64_t)K * SS : (int64_t)nout);
    LC_HIP(hipMemcpyAsync(hss_.data(), ssout_.p, nout * sizeof(double), hipMemcpyDeviceToHost, stream_));
  }
  LC_HIP(hipStreamSynchronize(stream_));
  for (int k = 0; k < K; ++k) {
    const double* rec = hss_.data() + (size_t)k * SS;
    const bool off = false;  // (a single group's masked clusters were zeroed on the device, before the sum over ranks)
    if (Nk) Nk[k] = off ? 0.0 : rec[0];
    if (xs)
      for (int d = 0; d < D; ++d) xs[(size_t)k * D + d] = off ? 0.0 : rec[1 + d];
    if (xxs) {
      const double* S = rec + 1 + DP;
      double* o = xxs + (size_t)k * D * D;
      // lower triangle is authoritative; mirror it so the result is exactly symmetric
      for (int i = 0; i < D; ++i)
        for (int j = 0; j <= i; ++j) {
          const double v = off ? 0.0 : S[(size_t)i * DP + j];
          o[(size_t)i * D + j] = v;
          o[(size_t)j * D + i] = v;
        }
    }
  }
  if (Njk) {
    if (!own_counts)  // qZ.colwise().sum() of the only group == the N_k record
      for (int k = 0; k < K; ++k) Njk[k] = hss_[(size_t)k * SS];
    else
      std::copy(hss_.begin() + (size_t)K * SS, hss_.end(), Njk);
  }
}

bool Context::dcache_eligible(int K) const {
  static const bool off = lck::test_switch("LC_SPLIT_NO_DCACHE") != nullptr || lck::test_switch("LC_SPLIT_NO_DELTA") != nullptr;
  // (the normalisation sweep keeps a row's K values in registers)
  return !off && DP_ <= lck::GW_MAX_DP && K >= 1 && K <= lck::softmax_cached_max_k() && !lck::fused_eligible(DP_, K);
}

void Context::dcache_invalidate() {
  dc_K_ = 0;
  rm_valid_ = false;
  dc_ver_.clear();
  dc_slot_.clear();
  std::fill(dc_used_.begin(), dc_used_.end(), (unsigned char)0);
  dc_saved_.clear();
  dc_journal_ = false;
  dq_K_ = 0;
}

void Context::dcache_release() {
  dcache_invalidate();
  dc_cap_ = 0;
  dc_room_K_ = 0;
  dc_used_.clear();
  dc_slab_.release();
  dfresh_.release();
  rm_max_.release();
  rm_arg_.release();
  bs_need_.release();
  bs_x_.release();
  bs_out_.release();
  dq_.release();
  amax_.release();
  dc_tagA_.clear();
  dc_tagm_.clear();
}

int Context::dc_find_run(int n) const {
  int run = 0;
  for (int s = 0; s < dc_cap_; ++s) {
    run = dc_used_[(size_t)s] ? 0 : run + 1;
    if (run == n) return s - n + 1;
  }
  return -1;
}

void Context::dcache_journal_begin() {
  dcache_journal_end();  // (a journal left open: its old columns are no longer needed)
  dc_journal_ = true;
  dc_jK0_ = dc_K_;
}

void Context::dcache_journal_end() {
  for (auto& sv : dc_saved_) dc_used_[(size_t)sv->slot] = 0;  // commit: the journaled contents go
  dc_saved_.clear();
  dc_journal_ = false;
}

void Context::dcache_rollback() {
  if (!dc_journal_) return;
  // back to the map of dcache_journal_begin: the trial's columns give their slots back, the journaled ones return
  for (int k = dc_jK0_; k < dc_K_; ++k) dc_used_[(size_t)dc_slot_[(size_t)k]] = 0;
  if ((int)dc_slot_.size() < dc_jK0_) dc_slot_.resize((size_t)dc_jK0_, -1);
  for (auto& sv : dc_saved_) {
    const int cur = sv->col < dc_K_ ? dc_slot_[(size_t)sv->col] : -1;
    if (cur >= 0 && cur != sv->slot) dc_used_[(size_t)cur] = 0;
    dc_slot_[(size_t)sv->col] = sv->slot;
    dc_used_[(size_t)sv->slot] = 1;
    if ((int)dc_ver_.size() <= sv->col) dc_ver_.resize((size_t)sv->col + 1, 0);
    dc_ver_[(size_t)sv->col] = sv->ver;
    dc_tagA_[(size_t)sv->col].swap(sv->A);
    dc_tagm_[(size_t)sv->col].swap(sv->m);
  }
  dc_K_ = dc_jK0_;
  dc_slot_.resize((size_t)dc_K_);
  dc_saved_.clear();
  dc_journal_ = false;
}

// Recompute the changed columns only for the rows they can reach (lck::BoundSelectLaunch has the argument).  Returns false
// when the preconditions do not hold -- the caller then runs the ordinary raw pass over all rows, which overwrites whatever
// this function wrote.  One group, every changed column with a reference column in the slab, few columns, many rows.
bool Context::bound_static_ok() const {
  // (below ~ 200 k rows the selection's two round trips cost what the pass does; tests lower the limit to walk the path on
  //  small problems)
  static const int64_t min_rows = lck::test_switch("LC_SPLIT_BOUND_MIN_ROWS") ? std::atoll(lck::test_switch("LC_SPLIT_BOUND_MIN_ROWS")) : 200000;
  return J_ == 1 && !distributed() && NP_ >= min_rows;
}

bool Context::recompute_bounded(int K, const std::vector<int>& changed, const std::vector<int>& oldslot,
                                const std::vector<std::vector<double>>& oldA, const std::vector<std::vector<double>>& oldm,
                                const std::vector<int>& dest, const double* A, const double* m, const double* c, bool delta) {
  static const bool off = lck::test_switch("LC_SPLIT_NO_BOUND") != nullptr;  // (tests: every recomputation over all rows)
  const int nch = (int)changed.size(), D = D_;
  const size_t AA = (size_t)D * D;
  static const bool trace = std::getenv("LC_TRACE_PHASES") != nullptr;
  auto no = [&](const char* why) {
    if (trace) std::cerr << "[cache] no bounded recomputation at K " << K << " (" << nch << " columns): " << why << std::endl;
    return false;
  };
  if (off || !bound_static_ok()) return false;
  if (!rm_valid_) return no("no row maxima");
  if (nch < 1 || nch > lck::BOUND_MAX_COLS || K > lck::BOUND_MAX_K || nch * 3 > K) return no("too many columns");
  // a reference for every column: its own previous version, else the first previous version any changed column has
  int fallback = -1;
  for (int t = 0; t < nch && fallback < 0; ++t)
    if (oldslot[(size_t)t] >= 0) fallback = t;
  if (fallback < 0) return no("no reference column");
  lck::BoundSelectLaunch b;
  b.ncol = nch;
  b.K = K;
  b.NP = NP_;
  std::vector<double> M(AA), v((size_t)D), w((size_t)D), cert;
  for (int t = 0; t < nch; ++t) {
    const int rt = oldslot[(size_t)t] >= 0 ? t : fallback;
    const double* Ar = oldA[(size_t)rt].data();
    const double* mr = oldm[(size_t)rt].data();
    const double* An = A + (size_t)changed[(size_t)t] * AA;
    const double* mn = m + (size_t)changed[(size_t)t] * D;
    // M = A_ref A_new^-1 (both lower triangular): row i of M from M A_new = A_ref, right to left
    std::fill(M.begin(), M.end(), 0.0);
    for (int i = 0; i < D; ++i)
      for (int j = i; j >= 0; --j) {
        double s = Ar[(size_t)i * D + j];
        for (int l = j + 1; l <= i; ++l) s -= M[(size_t)i * D + l] * An[(size_t)l * D + j];
        const double d = An[(size_t)j * D + j];
        if (!(d > 0.0)) return no("singular whitener");
        M[(size_t)i * D + j] = s / d;
      }
    // |M|_2 = 1 / sigma_min(B).  The power method on M^T M approaches |M|_2 from BELOW (and can sit on the second singular
    // vector for a while: ADVICE r5), so its estimate only PROPOSES the bound tau = (est / 0.97)^2; the bound is then
    // PROVED: tau I - M^T M has an LDL^T factorisation with positive pivots iff |M|_2^2 < tau (norm_certified, D^3 / 3
    // operations -- what the power method's iterations cost).  Not proved: the ordinary pass runs.  Beyond D = 128 the
    // proof costs more than the column: there sqrt(|M|_1 |M|_inf) >= |M|_2 is taken as it is (looser, never wrong).
    double upper = 0.0;
    if (D > 128) {
      std::fill(v.begin(), v.end(), 0.0);  // column sums
      double rinf = 0.0;
      for (int i = 0; i < D; ++i) {
        double rs = 0.0;
        for (int j = 0; j <= i; ++j) {
          const double a = std::fabs(M[(size_t)i * D + j]);
          rs += a;
          v[(size_t)j] += a;
        }
        rinf = std::max(rinf, rs);
      }
      double r1 = 0.0;
      for (int j = 0; j < D; ++j) r1 = std::max(r1, v[(size_t)j]);
      upper = std::sqrt(r1 * rinf) * (1.0 + 1e-12);
      if (!(upper > 0.0) || !std::isfinite(upper)) return no("norm bound broke down");
    } else {
      for (int i = 0; i < D; ++i) v[(size_t)i] = 1.0 + 0.01 * i;
      double est = 0.0, prev = -1.0;
      bool settled = false;
      for (int it = 0; it < 200 && !settled; ++it) {
        for (int i = 0; i < D; ++i) {  // w = M v
          double s = 0.0;
          for (int j = 0; j <= i; ++j) s += M[(size_t)i * D + j] * v[(size_t)j];
          w[(size_t)i] = s;
        }
        double nv = 0.0;
        for (int j = 0; j < D; ++j) {  // v = M^T w
          double s = 0.0;
          for (int i = j; i < D; ++i) s += M[(size_t)i * D + j] * w[(size_t)i];
          v[(size_t)j] = s;
          nv += s * s;
        }
        nv = std::sqrt(nv);
        if (!(nv > 0.0) || !std::isfinite(nv)) return no("power method broke down");
        for (int j = 0; j < D; ++j) v[(size_t)j] /= nv;
        est = std::sqrt(nv);  // |M^T M v| -> lambda_max = |M|_2^2 for unit v
        settled = it >= 4 && std::fabs(est - prev) <= 1e-3 * est;
        prev = est;
      }
      if (!settled) return no("power method not settled");
      upper = est / 0.97;
      if (!lch::norm_certified(M.data(), D, upper * upper, cert)) return no("norm estimate not proved");
    }
    double bn = 0.0;
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
      for (int j = 0; j <= i; ++j) s += An[(size_t)i * D + j] * (mr[j] - mn[j]);
      bn += s * s;
    }
    b.ref[t] = dc_slab_.p + (size_t)oldslot[(size_t)rt] * NP_;
    b.dest[t] = dc_slab_.p + (size_t)dest[(size_t)t] * NP_;
    b.sigma[t] = 1.0 / upper;          // (a lower bound of sigma_min(B): `upper` is a proved upper bound of |M|_2)
    b.bnorm[t] = 1.03 * std::sqrt(bn) + 1e-9;
    b.cnew[t] = c[changed[(size_t)t]];
  }
  std::fill(b.usable, b.usable + lck::BOUND_MAX_K, (unsigned char)0);
  std::fill(b.dcj, b.dcj + lck::BOUND_MAX_K, 0.0);
  int nusable = 0;
  for (int j = 0; j < K && j < rm_K_; ++j) {
    bool ch = false;
    for (int k : changed) ch = ch || k == j;
    if (ch || j >= (int)dc_ver_.size() || dc_ver_[(size_t)j] != rm_ver_[(size_t)j]) continue;
    b.usable[j] = 1;
    b.dcj[j] = c[j] - rm_c_[(size_t)j];
    ++nusable;
  }
  if (nusable == 0) return no("no unchanged column");
  b.rmax = rm_max_.p;
  b.ramax = rm_arg_.p;
  b.T = (delta ? 208.0 : 746.0) + 32.0;
  // The path's own buffers (up to a third of X, gathered) are not part of the room question estep_cache asked: when the
  // device says no, the ordinary pass runs (it overwrites whatever the selection wrote) -- never an error out of learn*()
  try {
    bs_need_.reserve((size_t)NP_);
    b.need = bs_need_.p;
    LC_HIP(lck::launch_bound_select(b, stream_));
    RowSelection sel;
    select_rows_col(bs_need_.p, 0.5, sel);
    if (trace) {
      std::cerr << "[cache] bounded recomputation: " << sel.M << " of " << NP_ << " rows for " << nch << " columns; usable " << nusable
                << ", T " << b.T;
      for (int t = 0; t < nch; ++t) std::cerr << " | col " << changed[(size_t)t] << " sigma " << b.sigma[t] << " |b| " << b.bnorm[t] << " c " << b.cnew[t];
      std::cerr << std::endl;
    }
    if (sel.M * 3 > NP_) return false;  // (most rows: the ordinary pass is the cheaper one, and overwrites the bounds just written)
    bound_rows_ += sel.M;
    bound_passes_ += 1;
    if (sel.M == 0) return true;
    // the selected rows side by side, the ordinary raw E-step on them, the results back to their rows
    const int64_t Mp = (sel.M + lck::RG - 1) / lck::RG * lck::RG;
    bs_x_.reserve((size_t)Mp * DP_);
    bs_out_.reserve((size_t)Mp * nch);
    if (Mp > sel.M) LC_HIP(hipMemsetAsync(bs_x_.p + (size_t)sel.M * DP_, 0, (size_t)(Mp - sel.M) * DP_ * sizeof(double), stream_));
    LC_HIP(lck::launch_gather_rows_plain(X_.p, DP_, sel.idx.p, sel.M, bs_x_.p, stream_));
    std::vector<double> A2((size_t)nch * AA), m2((size_t)nch * D);
    for (int t = 0; t < nch; ++t) {
      std::copy(A + (size_t)changed[(size_t)t] * AA, A + (size_t)(changed[(size_t)t] + 1) * AA, A2.begin() + (size_t)t * AA);
      std::copy(m + (size_t)changed[(size_t)t] * D, m + (size_t)(changed[(size_t)t] + 1) * D, m2.begin() + (size_t)t * D);
    }
    const std::vector<double> zero((size_t)nch, 0.0);
    const int64_t PS = lck::estep_pstride(DP_, DC_);  // (the wide layout beyond DP = 128: ADVICE r5, high)
    pack_estep_params(nch, A2.data(), m2.data(), zero.data());
    params_.reserve(hpack_.size());
    LC_HIP(hipMemcpyAsync(params_.p, hpack_.data(), hpack_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    const int64_t nrg = Mp / lck::RG;
    lck::EstepLaunch a;
    a.DP = DP_;
    a.DC = DC_;
    a.X = bs_x_.p;
    a.nrg = nrg;
    a.rginfo = nullptr;
    a.nrows = sel.M;
    a.params = params_.p;
    a.ctab = params_.p + (size_t)nch * PS;
    a.K = nch;
    a.qZ = bs_out_.p;
    a.ldq = Mp;
    a.ll_part = nullptr;
    a.raw = 1;
    const int64_t grid = lck::estep_grid(a);
    fzpart_.reserve((size_t)std::max<int64_t>(grid, 1));
    a.fz_part = fzpart_.p;
    EvPair ev{};
    if (timing_) {
      ev.a = timing_event();
      ev.b = timing_event();
      ev.kind = 0;
      LC_HIP(hipEventRecord(ev.a, stream_));
    }
    LC_HIP(lck::launch_estep(a, stream_));
    if (timing_) {
      LC_HIP(hipEventRecord(ev.b, stream_));
      pending_.push_back(ev);
    }
    double* dp[lck::BOUND_MAX_COLS];
    for (int t = 0; t < nch; ++t) dp[t] = b.dest[t];
    LC_HIP(lck::launch_scatter_cols(bs_out_.p, Mp, nch, dp, sel.idx.p, sel.M, stream_));
    LC_HIP(hipStreamSynchronize(stream_));  // (the packed parameters and `sel` are about to go)
    return true;
  } catch (const AllocFailure&) {
    (void)hipGetLastError();
    LC_HIP(hipStreamSynchronize(stream_));
    return no("no room for the gathered rows");
  }
}

int Context::estep_cache(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk, double delta_tol,
                         int* stale_out) {
  use_device();
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  require_gw_width();
  const int D = D_;
  const size_t AA = (size_t)D * D;
  dq_K_ = 0;
  // the columns whose tag is not, bit for bit, this cluster's whitener and mean
  std::vector<int> changed;
  int stale = 0;
  for (int k = 0; k < K; ++k) {
    const bool have = k < dc_K_ && dc_tagA_[(size_t)k].size() == AA && dc_tagm_[(size_t)k].size() == (size_t)D;
    if (have && std::memcmp(dc_tagA_[(size_t)k].data(), A + (size_t)k * AA, AA * sizeof(double)) == 0 &&
        std::memcmp(dc_tagm_[(size_t)k].data(), m + (size_t)k * D, (size_t)D * sizeof(double)) == 0)
      continue;
    changed.push_back(k);
    stale += have ? 1 : 0;
  }
  if (stale_out) *stale_out = stale;
  const int nch = (int)changed.size();
  const size_t NPs = (size_t)std::max<int64_t>(NP_, 1);
  // ---- which slots the changed columns need (nothing is touched yet)
  auto is_saved = [&](int k) {
    for (auto& sv : dc_saved_)
      if (sv->col == k) return true;
    return false;
  };
  // a valid column of the journal's base that has not been journaled yet keeps its slot as it is (a rollback returns to
  // it): the recomputed column needs a new one; so does a column that does not exist yet.  Everything else may be
  // overwritten where it is.
  std::vector<int> writable((size_t)nch, -1);
  int new_needed = 0;
  for (int t = 0; t < nch; ++t) {
    const int k = changed[(size_t)t];
    const bool exists = k < dc_K_ && k < (int)dc_slot_.size() && dc_slot_[(size_t)k] >= 0;
    if (exists && !(dc_journal_ && k < dc_jK0_ && !is_saved(k))) writable[(size_t)t] = dc_slot_[(size_t)k];
    else ++new_needed;
  }
  int used = 0;
  for (unsigned char u : dc_used_) used += u ? 1 : 0;
  // Is there room?  Asked before anything is touched, and answered by ALL ranks together (`changed` and the slot map are
  // functions of the replicated M-step: every rank arrives at the same numbers): a grown slab lives next to the old one
  // for a moment, the move of the responsibilities takes K more columns.
  // (capacities grow geometrically: at tens of millions of rows every one of these blocks is past the block cache's
  // limit, and a hipMalloc / hipFree of ten gigabytes costs seconds)
  // (the recomputed columns are written side by side: when they cannot stay where they are, a RUN of free slots has to
  //  exist -- a slab that is nearly full has free slots but no run, and every recomputation would go through the scratch
  //  buffer and a copy per column: grow it instead, once)
  bool need_run = nch > 0 && writable[0] < 0;
  for (int t = 1; t < nch && !need_run; ++t) need_run = writable[(size_t)t] != writable[0] + t;
  bool have_run = !need_run;
  if (need_run && dc_cap_ > 0) {
    std::vector<unsigned char> u(dc_used_);
    for (int t = 0; t < nch; ++t)
      if (writable[(size_t)t] >= 0) u[(size_t)writable[(size_t)t]] = 0;
    int runlen = 0;
    for (int sl = 0; sl < dc_cap_ && !have_run; ++sl) {
      runlen = u[(size_t)sl] ? 0 : runlen + 1;
      have_run = runlen >= nch;
    }
  }
  const bool grow = K > dc_cap_ || used + new_needed > dc_cap_ || !have_run;
  const int grown_cap = std::max(std::max(K, used + new_needed) + 8, 2 * dc_cap_);
  bool ask = K > dc_room_K_ || grow;
#ifdef LC_TEST_HOOKS  // (libcluster_hip_testhooks.so only: the shipped library has no fault hooks)
  // tests: the named rank "has no room" for the columns of a split trial (every rank reads the same environment, so every
  // rank asks the question at the same E-step)
  static const char* fail_rank = std::getenv("LC_TEST_JOURNAL_FAIL_RANK");
  const bool trial = fail_rank && dc_journal_ && new_needed > 0;
  ask = ask || trial;
#endif
  if (ask) {
    const int newcap = grow ? grown_cap : dc_cap_;
    size_t need = 0;
    if (grow) need += NPs * (size_t)newcap * sizeof(double);
    if (delta_tol >= 0.0 && dq_.cap < NPs * (size_t)newcap) need += NPs * (size_t)(newcap + 1) * sizeof(double);
    bool ok = true;
    std::string why;
    size_t free_b = 0, total_b = 0;
    if (need > 0) {
      need += NPs * 4 * sizeof(double);
      LC_HIP(hipMemGetInfo(&free_b, &total_b));
      if (need > free_b) {
        trim_cache();
        LC_HIP(hipMemGetInfo(&free_b, &total_b));
      }
      ok = need <= free_b;
      if (!ok) why = std::to_string(need) + " bytes needed, " + std::to_string(free_b) + " free on this rank";
    }
#ifdef LC_TEST_HOOKS
    static const char* fake = std::getenv("LC_TEST_CACHE_NO_ROOM");  // tests: pretend the device is full from this K on
    if (fake && K >= std::atoi(fake)) ok = false, why = "LC_TEST_CACHE_NO_ROOM";
    if (trial) {
      const char* er = std::getenv("RANK");  // (hook-based runs have no communicator to ask)
      if (std::atoi(fail_rank) == (comm_ ? comm_->rank() : er ? std::atoi(er) : 0)) ok = false, why = "LC_TEST_JOURNAL_FAIL_RANK";
    }
#endif
    if (allreduce_value(ok ? 0.0 : 1.0) > 0.0)
      throw CacheNoRoom("no room for the distance cache at K = " + std::to_string(K) + (ok ? std::string(" on another rank") : ": " + why));
    dc_room_K_ = std::max(dc_room_K_, K);
  }
  // room for K columns and the few a trial holds on to (the slab moves as a whole when it grows: slots keep their numbers)
  if (grow) {
    const int newcap = grown_cap;
    static const bool trace = std::getenv("LC_TRACE_PHASES") != nullptr;
    if (trace) std::cerr << "[cache] slab " << dc_cap_ << " -> " << newcap << " columns" << std::endl;
    DevBuf<double> nb;
    nb.reserve(NPs * newcap);
    if (dc_cap_ > 0 && NP_ > 0 && used > 0)
      LC_HIP(hipMemcpyAsync(nb.p, dc_slab_.p, (size_t)NP_ * dc_cap_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
    std::swap(dc_slab_.p, nb.p);
    std::swap(dc_slab_.cap, nb.cap);
    std::swap(dc_slab_.device, nb.device);
    dc_cap_ = newcap;
    dc_used_.resize((size_t)newcap, 0);
  }
  if ((int)dc_tagA_.size() < K) {
    dc_tagA_.resize((size_t)K);
    dc_tagm_.resize((size_t)K);
  }
  auto journal_column = [&](int k) {  // by reference: the slot stays as it is, whoever changes the column moves it on
    auto sv = std::make_unique<SavedColumn>();
    sv->col = k;
    sv->slot = dc_slot_[(size_t)k];
    sv->ver = k < (int)dc_ver_.size() ? dc_ver_[(size_t)k] : 0;
    sv->A = dc_tagA_[(size_t)k];
    sv->m = dc_tagm_[(size_t)k];
    dc_saved_.push_back(std::move(sv));
  };
  // columns past K (a narrower model than the last call's): their slots are free again -- unless a journal returns to them
  for (int k = K; k < dc_K_; ++k) {
    if (dc_journal_ && k < dc_jK0_) {
      if (!is_saved(k)) journal_column(k);
    } else {
      dc_used_[(size_t)dc_slot_[(size_t)k]] = 0;
    }
  }
  if ((int)dc_slot_.size() < K) dc_slot_.resize((size_t)K, -1);
  // ---- where the recomputed columns go.  The raw E-step writes its columns side by side, so the changed columns need a
  // RUN of slots: their own when those are writable and adjacent, else a free run (the writable old slots are given up
  // first), else the scratch buffer and one copy per column.
  if (nch > 0) {
    for (int t = 0; t < nch; ++t)
      if (writable[(size_t)t] < 0 && changed[(size_t)t] < dc_K_ && changed[(size_t)t] < dc_jK0_ && dc_journal_ &&
          dc_slot_[(size_t)changed[(size_t)t]] >= 0 && !is_saved(changed[(size_t)t]))
        journal_column(changed[(size_t)t]);
    bool inplace = writable[0] >= 0;
    for (int t = 1; t < nch && inplace; ++t) inplace = writable[(size_t)t] == writable[0] + t;
    int run = inplace ? writable[0] : -1;
    if (!inplace) {
      for (int t = 0; t < nch; ++t)
        if (writable[(size_t)t] >= 0) dc_used_[(size_t)writable[(size_t)t]] = 0;
      run = dc_find_run(nch);
    }
    std::vector<double> A2, m2;
    const double *Ap = A + (size_t)changed.front() * AA, *mp = m + (size_t)changed.front() * D;
    if (changed.back() - changed.front() + 1 != nch) {  // (the parameter records of the changed clusters, side by side)
      A2.resize((size_t)nch * AA);
      m2.resize((size_t)nch * D);
      for (int t = 0; t < nch; ++t) {
        std::copy(A + (size_t)changed[t] * AA, A + (size_t)(changed[t] + 1) * AA, A2.begin() + (size_t)t * AA);
        std::copy(m + (size_t)changed[t] * D, m + (size_t)(changed[t] + 1) * D, m2.begin() + (size_t)t * D);
      }
      Ap = A2.data();
      mp = m2.data();
    }
    // the changed columns' previous versions (slot and tags), where they have one: the references of the distance bound
    std::vector<int> oldslot((size_t)nch, -1);
    std::vector<std::vector<double>> oldA((size_t)nch), oldm((size_t)nch);
    for (int t = 0; t < nch; ++t) {
      const int k = changed[(size_t)t];
      if (k < dc_K_ && k < (int)dc_slot_.size() && dc_slot_[(size_t)k] >= 0 && dc_tagA_[(size_t)k].size() == AA) {
        oldslot[(size_t)t] = dc_slot_[(size_t)k];
        oldA[(size_t)t] = dc_tagA_[(size_t)k];
        oldm[(size_t)t] = dc_tagm_[(size_t)k];
      }
    }
    std::vector<int> dest((size_t)nch, -1);
    if (run >= 0) {
      for (int t = 0; t < nch; ++t) dest[(size_t)t] = run + t;
    } else {  // no run of nch free slots: one slot each (there are enough: `grow` above), filled from the scratch buffer
      for (int t = 0; t < nch; ++t) {
        const int sl = dc_find_run(1);
        if (sl < 0) throw std::logic_error("distance cache: no free column slot");
        dest[(size_t)t] = sl;
        dc_used_[(size_t)sl] = 1;
      }
      dfresh_.reserve(NPs * nch);
    }
    for (int t = 0; t < nch; ++t) {
      dc_used_[(size_t)dest[(size_t)t]] = 1;
      dc_slot_[(size_t)changed[(size_t)t]] = dest[(size_t)t];
    }
    const std::vector<double> zero((size_t)J_ * nch, 0.0);
    double fz0 = 0.0;
    double* target = run >= 0 ? dc_slab_.p + (size_t)run * NP_ : dfresh_.p;
    if ((int)dc_ver_.size() < K) dc_ver_.resize((size_t)K, 0);
    for (int k : changed) dc_ver_[(size_t)k] = ++dc_vernext_;
    if (NP_ > 0 && run >= 0 && recompute_bounded(K, changed, oldslot, oldA, oldm, dest, A, m, c, delta_tol >= 0.0 && qz_[cur_].K == K)) {
      // (only the rows the new columns can reach were recomputed; the others hold an upper bound that the sweep turns
      //  into exactly 0.0; the margin is the one of the sweep's REAL mode: it flushes only with old responsibilities)
    } else if (NP_ > 0) {
      estep(nch, Ap, mp, zero.data(), &fz0, nullptr, true, target);
      if (run < 0)
        for (int t = 0; t < nch; ++t)
          LC_HIP(hipMemcpyAsync(dc_slab_.p + (size_t)dest[(size_t)t] * NP_, dfresh_.p + (size_t)t * NP_,
                                (size_t)NP_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    }
    for (int k : changed) {
      dc_tagA_[(size_t)k].assign(A + (size_t)k * AA, A + (size_t)(k + 1) * AA);
      dc_tagm_[(size_t)k].assign(m + (size_t)k * D, m + (size_t)(k + 1) * D);
    }
  }
  if (!dc_journal_ || K >= dc_jK0_) dc_slot_.resize((size_t)std::max(K, dc_journal_ ? dc_jK0_ : 0));
  dc_K_ = K;
  // constants + normalisation
  const bool have_old = qz_[cur_].K == K;  // the buffer holds K columns of q_old
  const bool hashed_old = have_old && qz_[cur_].hash_ok;
#ifdef LC_TEST_HOOKS
  // tests: before the sweep trusts the stored fingerprints, recompute every one of them from the buffer.  Any writer of
  // qZ that left hash_ok standing over rows it changed (without marking them QHASH_NONE) is caught here, on whatever
  // path a learner reached it.
  static const bool verify = std::getenv("LC_TEST_VERIFY_QHASH") != nullptr;
  if (verify && hashed_old && NP_ > 0) {
    DevBuf<int64_t> bad;
    bad.reserve(1);
    LC_HIP(hipMemsetAsync(bad.p, 0, sizeof(int64_t), stream_));
    LC_HIP(lck::launch_qhash_verify(qz_[cur_].buf.p, NP_, K, NP_, qz_[cur_].hash.p, reinterpret_cast<unsigned long long*>(bad.p), stream_));
    unsigned long long nbad = 0;
    LC_HIP(hipMemcpyAsync(&nbad, bad.p, sizeof(nbad), hipMemcpyDeviceToHost, stream_));
    LC_HIP(hipStreamSynchronize(stream_));
    if (nbad) throw std::runtime_error("LC_TEST_VERIFY_QHASH: " + std::to_string(nbad) + " rows carry a fingerprint that is not theirs (K = " + std::to_string(K) + ")");
    std::cerr << "[cache] fingerprints verified, K " << K << std::endl;
  }
#endif
  ensure_qz(qz_[cur_], K, false);
  qz_[cur_].K = K;
  const int64_t grid = lck::softmax_cached_grid(NP_);
  fzpart_.reserve((size_t)std::max<int64_t>(grid, 1));
  if (LLk) llpart_.reserve((size_t)std::max<int64_t>(grid, 1) * K);
  red_.reserve((size_t)1 + K);
  const bool delta = delta_tol >= 0.0 && have_old;
  const int nred = LLk ? 1 + K : 1;
  hred_.resize((size_t)nred);
  constexpr bool direct_env = true;  // (the copy-back command instead: measured slower)
  const bool direct = direct_env && !distributed() && NP_ > 0;
  if (NP_ > 0) {
    // [c_jk table | the K slots of the clusters' columns (ints)] in one upload
    hpack_.assign((size_t)J_ * K + (size_t)(K + 1) / 2, 0.0);
    std::memcpy(hpack_.data(), c, (size_t)J_ * K * sizeof(double));
    std::memcpy(hpack_.data() + (size_t)J_ * K, dc_slot_.data(), (size_t)K * sizeof(int));
    params_.reserve(hpack_.size());
    LC_HIP(hipMemcpyAsync(params_.p, hpack_.data(), hpack_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
    lck::CachedNormLaunch a;
    a.dcache = dc_slab_.p;
    a.ldc = NP_;
    a.fresh = nullptr;
    a.ldf = 0;
    a.colmap = reinterpret_cast<const int*>(params_.p + (size_t)J_ * K);
    a.ctab = params_.p;
    a.K = K;
    a.rginfo = J_ > 1 ? rginfo_.p : nullptr;
    a.nrows = Nj_[0];
    a.NP = NP_;
    a.qZ = qz_[cur_].buf.p;
    a.ldq = NP_;
    a.fz_part = fzpart_.p;
    a.ll_part = LLk ? llpart_.p : nullptr;
    bool rm_room = bound_static_ok() && K <= lck::BOUND_MAX_K;  // (what recompute_bounded needs next time -- where it can run)
    if (rm_room) {
      try {
        rm_max_.reserve((size_t)NP_);
        rm_arg_.reserve((size_t)NP_);
      } catch (const AllocFailure&) {
        (void)hipGetLastError();
        rm_room = false;
      }
    }
    if (rm_room) {
      a.rmax = rm_max_.p;
      a.ramax = rm_arg_.p;
      rm_valid_ = true;
      rm_K_ = K;
      rm_ver_.assign(dc_ver_.begin(), dc_ver_.begin() + std::min<size_t>(dc_ver_.size(), (size_t)K));
      rm_ver_.resize((size_t)K, 0);
      rm_c_.assign(c, c + K);
    } else {
      rm_valid_ = false;
    }
    if (delta) {
      dq_.reserve((size_t)NP_ * std::max(K, dc_cap_));  // (as wide as the slab: re-allocated only when that grows)
      amax_.reserve((size_t)NP_);
      a.dq = dq_.p;
      a.ldd = K;  // row-major [NP x K]
      dq_ld_ = K;
      a.amax = amax_.p;
      a.dq_tol = delta_tol;
      dq_tol_ = delta_tol;
      dq_maskd_.reserve(2);
      LC_HIP(hipMemsetAsync(dq_maskd_.p, 0, 2 * sizeof(int64_t), stream_));
      a.colmask = reinterpret_cast<unsigned long long*>(dq_maskd_.p);
      static const bool no_hash = lck::test_switch("LC_SPLIT_NO_QHASH") != nullptr;  // (tests: read every old value)
      if (!no_hash) {
        qz_[cur_].hash.reserve((size_t)NP_);
        a.qhash = qz_[cur_].hash.p;
        a.qhash_in = hashed_old ? 1 : 0;
        static const bool trace = std::getenv("LC_TRACE_PHASES") != nullptr;
        if (trace) std::cerr << "[cache] sweep K " << K << ", fingerprints " << (hashed_old ? "compared" : "written") << std::endl;
      }
    }
    EvPair ev{};
    if (timing_) {
      ev.a = timing_event();
      ev.b = timing_event();
      ev.kind = 0;
      LC_HIP(hipEventRecord(ev.a, stream_));
    }
    LC_HIP(lck::launch_softmax_cached(a, stream_));
    if (timing_) {
      LC_HIP(hipEventRecord(ev.b, stream_));
      pending_.push_back(ev);
    }
    redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
    double* dst = direct ? hred_.data() : red_.p;  // (single rank: the folds write into the page-locked host buffer)
    LC_HIP(lck::launch_reduce_partials(fzpart_.p, (int)grid, 1, dst, stream_, redtmp_.p));
    if (LLk) LC_HIP(lck::launch_reduce_partials(llpart_.p, (int)grid, K, dst + 1, stream_, redtmp_.p));
  } else {
    LC_HIP(hipMemsetAsync(red_.p, 0, (size_t)(1 + K) * sizeof(double), stream_));
  }
  if (!direct) {
    allreduce(red_.p, nred);
    LC_HIP(hipMemcpyAsync(hred_.data(), red_.p, (size_t)nred * sizeof(double), hipMemcpyDeviceToHost, stream_));
  }
  dq_mask_ok_ = false;
  if (delta && NP_ > 0) {
    hmask_.resize(2);
    LC_HIP(hipMemcpyAsync(hmask_.data(), dq_maskd_.p, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, stream_));
  }
  run_overlap();
  LC_HIP(hipStreamSynchronize(stream_));  // (also covers the host vectors the asynchronous copies read)
  if (delta && NP_ > 0) {
    std::memcpy(dq_mask_, hmask_.data(), sizeof(dq_mask_));
    dq_mask_ok_ = true;
  }
  if (Fz) *Fz = hred_[0];
  if (LLk) std::copy(hred_.begin() + 1, hred_.begin() + 1 + K, LLk);
  if (delta) dq_K_ = K;  // (a rank without rows keeps an empty delta and still joins delta_suffstat's sums)
  qz_[cur_].hash_ok = delta && NP_ > 0 && qz_[cur_].hash.p != nullptr && lck::test_switch("LC_SPLIT_NO_QHASH") == nullptr;
  return nch;
}

bool Context::delta_suffstat(int K1, double max_frac, double* dNk, double* dxs, double* dxxs, double* dNjk) {
  use_device();
  if (dq_K_ != K1 || K1 < 1) throw std::invalid_argument("no responsibilities delta of that width");
  dq_K_ = 0;  // single use: the next E-step overwrites the responsibilities it refers to
  RowSelection sel;
  select_rows_col(amax_.p, dq_tol_, sel);
  delta_rows_ = sel.M;
  double cnt[2] = {(double)sel.M, (double)Ntot_};
  allreduce_values(cnt, 2);  // every rank takes the same branch
  if (cnt[0] > max_frac * cnt[1]) return false;
  // the clusters any moved row moved in (the sweep's column mask): a split trial moves rows between the two halves of one
  // cluster and, here and there, a neighbour -- 2 to 4 columns of K + 1.  The other clusters' differences are all zero and
  // so are their statistics: they are not computed (single rank: the mask is this rank's own)
  std::vector<int> cols;
  if (dq_mask_ok_ && !distributed() && K1 <= 128) {
    for (int k = 0; k < K1; ++k)
      if ((dq_mask_[k >> 6] >> (k & 63)) & 1ull) cols.push_back(k);
  } else {
    for (int k = 0; k < K1; ++k) cols.push_back(k);
  }
  const int nc = (int)cols.size();
  const int D = D_;
  std::fill(dNk, dNk + K1, 0.0);
  std::fill(dxs, dxs + (size_t)K1 * D, 0.0);
  std::fill(dxxs, dxxs + (size_t)K1 * D * D, 0.0);
  std::fill(dNjk, dNjk + (size_t)J_ * K1, 0.0);
  if (nc == 0) return true;
  Context sub(device_, stream_);
  sub.inherit_comm(*this);
  sub.skip_zero_ = false;
  sub.set_data_gather(*this, sel);
  QZ& q = sub.qz_[sub.cur_];
  {
    RelaxedFit lend;
    sub.ensure_qz(q, nc, false);
  }
  q.K = nc;
  if (sub.NP_ > 0) {
    LC_HIP(hipMemsetAsync(q.buf.p, 0, (size_t)sub.NP_ * nc * sizeof(double), stream_));  // padding rows carry nothing
    if (nc == K1) {
      LC_HIP(lck::launch_gather_rowmajor(dq_.p, dq_ld_, K1, sel.idx.p, sel.M, sel.starts_d.p, sub.goff_d_.p, J_, q.buf.p,
                                         sub.NP_, stream_));
    } else {
      dq_colsd_.reserve((size_t)nc);
      LC_HIP(hipMemcpyAsync(dq_colsd_.p, cols.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, stream_));
      LC_HIP(lck::launch_gather_rowmajor_cols(dq_.p, dq_ld_, dq_colsd_.p, nc, sel.idx.p, sel.M, sel.starts_d.p, sub.goff_d_.p,
                                              J_, q.buf.p, sub.NP_, stream_));
    }
  }
  if (nc == K1) {
    sub.suffstat(nullptr, dNk, dxs, dxxs, dNjk);
    return true;
  }
  std::vector<double> n2((size_t)nc), x2((size_t)nc * D), xx2((size_t)nc * D * D), nj2((size_t)J_ * nc);
  sub.suffstat(nullptr, n2.data(), x2.data(), xx2.data(), nj2.data());  // (synchronises: `cols` has been read by then)
  for (int t = 0; t < nc; ++t) {
    const int k = cols[(size_t)t];
    dNk[k] = n2[(size_t)t];
    std::copy(x2.begin() + (size_t)t * D, x2.begin() + (size_t)(t + 1) * D, dxs + (size_t)k * D);
    std::copy(xx2.begin() + (size_t)t * D * D, xx2.begin() + (size_t)(t + 1) * D * D, dxxs + (size_t)k * D * D);
    for (int j = 0; j < J_; ++j) dNjk[(size_t)j * K1 + k] = nj2[(size_t)j * nc + t];
  }
  return true;
}

void Context::colsums(double* Njk) {
  use_device();
  const int K = qz_[cur_].K;
  if (K < 1) throw std::invalid_argument("qZ has not been set");
  red_.reserve((size_t)std::max(1 + K, J_ * K));
  if (NP_ > 0) {
    redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
    LC_HIP(lck::launch_group_colsum(qz_[cur_].buf.p, NP_, K, goff_d_.p, J_, red_.p, stream_, redtmp_.p, NP_));
  } else
    LC_HIP(hipMemsetAsync(red_.p, 0, (size_t)J_ * K * sizeof(double), stream_));
  if (!group_sharded_) allreduce(red_.p, (int64_t)J_ * K);
  LC_HIP(hipMemcpyAsync(Njk, red_.p, (size_t)J_ * K * sizeof(double), hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
}

void Context::estep_diag(int K, const double* av, const double* w2, const double* w1, const double* c, double* Fz,
                         double* LLk, bool raw) {
  if (K < 1) throw std::invalid_argument("K must be >= 1");
  if (NP_ == 0 && !distributed()) {
    if (Fz) *Fz = -0.0;
    if (LLk) std::fill(LLk, LLk + K, 0.0);
    qz_[cur_].K = K;
    return;
  }
  LC_HIP(hipSetDevice(device_));
  const int D = D_, DP = DP_;
  hpack_.assign((size_t)K * 3 * DP + (size_t)J_ * K, 0.0);
  bool no_w1 = true, only_w1 = true;
  for (int k = 0; k < K; ++k) {
    double* P = hpack_.data() + (size_t)k * DP;
    std::copy(av + (size_t)k * D, av + (size_t)(k + 1) * D, P);
    std::copy(w2 + (size_t)k * D, w2 + (size_t)(k + 1) * D, P + (size_t)K * DP);
    std::copy(w1 + (size_t)k * D, w1 + (size_t)(k + 1) * D, P + (size_t)2 * K * DP);
    for (int d = 0; d < D; ++d) {
      no_w1 = no_w1 && w1[(size_t)k * D + d] == 0.0;
      only_w1 = only_w1 && w2[(size_t)k * D + d] == 0.0;
    }
  }
  std::memcpy(hpack_.data() + (size_t)K * 3 * DP, c, (size_t)J_ * K * sizeof(double));
  const int mode = only_w1 ? 2 : no_w1 ? 1 : 0;  // same arithmetic with the identically-zero terms left out
  // Matrix-pipe path (estep_diag_mfma_kernel): the log-likelihood expanded around a centre mu is bilinear in
  // [x'^2, x'].  Exact for the exponential family (linear in x); for the quadratic term the expansion cancels, so it is
  // taken only when max_d max_k |w2_kd| * reach_d^2 is small, reach_d = the farthest 6-sigma edge of any cluster from
  // the centre (the mean of the cluster centres): the absolute error of log q~ is then about D * cond * 2^-53.
  size_t wt_off = 0, mu_off = 0, ck_off = 0;
  {
    static const int force = lck::test_switch("LC_ED_MFMA") ? std::atoi(lck::test_switch("LC_ED_MFMA")) : -1;  // (tests: 0 never, 1 always)
    const int64_t nw = lck::estep_diag_mfma_weights(DP, K, mode);
    bool use = nw > 0 && force != 0;
    std::vector<double> muv((size_t)DP, 0.0);
    if (use && mode != 2) {
      double cond = 0.0;
      for (int d = 0; d < D; ++d) {
        double m = 0.0;
        for (int k = 0; k < K; ++k) m += av[(size_t)k * D + d];
        m /= K;
        muv[(size_t)d] = m;
        double wmax = 0.0, reach = 0.0;
        for (int k = 0; k < K; ++k) {
          const double w = std::fabs(w2[(size_t)k * D + d]);
          wmax = std::max(wmax, w);
          const double sig = w > 0 ? std::sqrt(0.5 / w) : 0.0;
          reach = std::max(reach, std::fabs(av[(size_t)k * D + d] - m) + 6.0 * sig);
        }
        cond = std::max(cond, wmax * reach * reach);
      }
      if (!(cond <= 4096.0) && force != 1) use = false;  // (NaN-safe)
    }
    if (use) {
      if (timing_) times_.estep_diag_mfma_calls += 1;
      const int NT = DP / 4, NTF = mode == 2 ? NT : 2 * NT, KT = (K + 3) / 4;
      const size_t base = hpack_.size();
      mu_off = base;
      ck_off = mu_off + (size_t)DP;
      wt_off = (ck_off + (size_t)K + 1) & ~(size_t)1;  // 16-byte aligned (double2 loads)
      hpack_.resize(wt_off + (size_t)nw);
      std::memcpy(hpack_.data() + (size_t)K * 3 * DP, c, (size_t)J_ * K * sizeof(double));  // (resize may have moved the block)
      for (int k = 0; k < K; ++k) {  // re-pack after a possible reallocation
        double* P = hpack_.data() + (size_t)k * DP;
        std::fill(P, P + DP, 0.0);
        std::fill(P + (size_t)K * DP, P + (size_t)K * DP + DP, 0.0);
        std::fill(P + (size_t)2 * K * DP, P + (size_t)2 * K * DP + DP, 0.0);
        std::copy(av + (size_t)k * D, av + (size_t)(k + 1) * D, P);
        std::copy(w2 + (size_t)k * D, w2 + (size_t)(k + 1) * D, P + (size_t)K * DP);
        std::copy(w1 + (size_t)k * D, w1 + (size_t)(k + 1) * D, P + (size_t)2 * K * DP);
      }
      std::copy(muv.begin(), muv.end(), hpack_.begin() + mu_off);
      double* ck = hpack_.data() + ck_off;
      double* wt = hpack_.data() + wt_off;
      std::fill(ck, ck + K + 1, 0.0);
      std::fill(wt, wt + nw, 0.0);
      // W[k][f]: f < NT*4 quadratic weights (w2), then linear weights (w1 - 2 w2 a'); tile (it, jt) element
      // (lo, hh) = W[4 it + lo][column 16 q + jr + 4 hh of its half], jt = 4 q + jr: the kernel's column groups (a lane
      // holds four contiguous columns per sixteen)
      auto put = [&](int k, int f, double v) {
        const int it = k >> 2, lo = k & 3, half = f / DP, d = f % DP;
        const int jt = half * NT + 4 * (d / 16) + (d % 4), hh = (d % 16) / 4;
        wt[((size_t)it * NTF + jt) * 16 + lo + 4 * hh] = v;
      };
      (void)KT;
      for (int k = 0; k < K; ++k) {
        double cs = 0.0;
        for (int d = 0; d < D; ++d) {
          const double a2 = w2[(size_t)k * D + d], a1 = w1[(size_t)k * D + d];
          const double ac = av[(size_t)k * D + d] - muv[(size_t)d];
          if (mode == 2) {
            put(k, d, a1);
          } else {
            put(k, d, a2);
            put(k, DP + d, a1 - 2.0 * a2 * ac);
            cs += a2 * ac * ac + a1 * muv[(size_t)d];
          }
        }
        ck[k] = cs;
      }
    }
  }
  params_.reserve(hpack_.size());
  LC_HIP(hipMemcpyAsync(params_.p, hpack_.data(), hpack_.size() * sizeof(double), hipMemcpyHostToDevice, stream_));
  ensure_qz(qz_[cur_], K, false);
  qz_[cur_].K = K;
  const int64_t nrg = NP_ / lck::RG, grid = lck::estep_diag_grid(nrg);
  fzpart_.reserve((size_t)std::max<int64_t>(grid, 1));
  llpart_.reserve((size_t)std::max<int64_t>(grid, 1) * K);
  red_.reserve((size_t)1 + K);
  lck::DiagEstepLaunch a;
  a.DP = DP;
  a.D = D;
  a.X = X_.p;
  a.nrg = nrg;
  a.rginfo = J_ > 1 ? rginfo_.p : nullptr;
  a.nrows = Nj_[0];
  a.params = params_.p;
  a.ctab = params_.p + (size_t)K * 3 * DP;  // params: [3][K][DP]
  a.K = K;
  a.qZ = qz_[cur_].buf.p;
  a.ldq = NP_;
  a.fz_part = fzpart_.p;
  a.ll_part = LLk ? llpart_.p : nullptr;
  a.raw = raw ? 1 : 0;
  a.mode = mode;
  if (wt_off) {
    a.wt = params_.p + wt_off;
    a.mu = params_.p + mu_off;
    a.constk = params_.p + ck_off;
    sink_.reserve(256);
    a.sink = sink_.p;
    a.ngroups = J_;
  }
  EvPair ev{};
  if (timing_) {
    ev.a = timing_event();
    ev.b = timing_event();
    ev.kind = 0;
    LC_HIP(hipEventRecord(ev.a, stream_));
  }
  LC_HIP(lck::launch_estep_diag(a, stream_));
  if (timing_) {
    LC_HIP(hipEventRecord(ev.b, stream_));
    pending_.push_back(ev);
  }
  if (raw) {
    LC_HIP(hipStreamSynchronize(stream_));
    return;
  }
  if (grid > 0) {
    redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
    LC_HIP(lck::launch_reduce_partials(fzpart_.p, (int)grid, 1, red_.p, stream_, redtmp_.p));
    if (LLk) LC_HIP(lck::launch_reduce_partials(llpart_.p, (int)grid, K, red_.p + 1, stream_, redtmp_.p));
    else LC_HIP(hipMemsetAsync(red_.p + 1, 0, (size_t)K * sizeof(double), stream_));
  } else {
    LC_HIP(hipMemsetAsync(red_.p, 0, (size_t)(1 + K) * sizeof(double), stream_));
  }
  allreduce(red_.p, 1 + K);
  hred_.resize((size_t)1 + K);
  LC_HIP(hipMemcpyAsync(hred_.data(), red_.p, (size_t)(1 + K) * sizeof(double), hipMemcpyDeviceToHost, stream_));
  run_overlap();
  LC_HIP(hipStreamSynchronize(stream_));
  if (Fz) *Fz = hred_[0];
  if (LLk) std::copy(hred_.begin() + 1, hred_.end(), LLk);
}

void Context::suffstat_diag(const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk) {
  const int K = qz_[cur_].K, D = D_, DP = DP_;
  if (K < 1) throw std::invalid_argument("qZ has not been set");
  LC_HIP(hipSetDevice(device_));
  const int64_t SS = 1 + 2 * (int64_t)DP;
  const size_t nout = (size_t)K * SS + (size_t)J_ * K;
  ssout_.reserve(nout);
  double* njk_d = ssout_.p + (size_t)K * SS;
  const bool own_counts = J_ > 1 || group_sharded() || smask != nullptr;
  if (NP_ > 0) {
    // about four blocks per resident slot (2 blocks per CU), whole 32-row batches; the partial records
    // (chunks x row classes x K x (1 + 2 DP) doubles) stay small next to the data
    const int rs = lck::suffstat_diag_rsplit(K), nslice = (K + 63) / 64;
    int64_t want = std::min<int64_t>(std::max<int64_t>(2048 / nslice, 1), (NP_ + 255) / 256);
    if (want < 1) want = 1;
    int64_t rows = ((NP_ + want - 1) / want + 31) / 32 * 32;
    const int nchunks = (int)((NP_ + rows - 1) / rows);
    const int nparts = nchunks * rs;
    sspart_reserve((size_t)nparts * K * SS);
    sspart_clean_ = false;  // (another record layout)
    lck::DiagStatLaunch a;
    a.DP = DP;
    a.X = X_.p;
    a.NP = NP_;
    a.qZ = qz_[cur_].buf.p;
    a.ldq = NP_;
    a.K = K;
    a.rginfo = nullptr;
    a.smask = nullptr;
    if (smask && J_ > 1) {
      smask_.reserve((size_t)J_ * K);
      LC_HIP(hipMemcpyAsync(smask_.p, smask, (size_t)J_ * K, hipMemcpyHostToDevice, stream_));
      a.rginfo = rginfo_.p;
      a.smask = smask_.p;
    }
    a.partial = sspart_.p;
    a.nchunks = nchunks;
    a.chunk_rows = rows;
    a.second = xxs ? 1 : 0;
    EvPair ev{};
    if (timing_) {
      ev.a = timing_event();
      ev.b = timing_event();
      ev.kind = 1;
      LC_HIP(hipEventRecord(ev.a, stream_));
    }
    LC_HIP(lck::launch_suffstat_diag(a, stream_));
    if (timing_) {
      LC_HIP(hipEventRecord(ev.b, stream_));
      pending_.push_back(ev);
    }
    LC_HIP(lck::launch_reduce_partials(sspart_.p, nparts, (int64_t)K * SS, ssout_.p, stream_));
    if (own_counts) {
      redtmp_.reserve((size_t)lck::REDUCE_TMP_ELEMS * 64);
      LC_HIP(lck::launch_group_colsum(qz_[cur_].buf.p, NP_, K, goff_d_.p, J_, njk_d, stream_, redtmp_.p, NP_));
    }
    else LC_HIP(hipMemsetAsync(njk_d, 0, (size_t)K * sizeof(double), stream_));
    if (smask && J_ == 1)
      for (int k = 0; k < K; ++k)
        if (!smask[k]) LC_HIP(hipMemsetAsync(ssout_.p + (size_t)k * SS, 0, (size_t)SS * sizeof(double), stream_));
  } else {
    LC_HIP(hipMemsetAsync(ssout_.p, 0, nout * sizeof(double), stream_));
  }
  allreduce(ssout_.p, group_sharded_ ? (int64_t)K * SS : (int64_t)nout);
  hss_.resize(nout);
  LC_HIP(hipMemcpyAsync(hss_.data(), ssout_.p, nout * sizeof(double), hipMemcpyDeviceToHost, stream_));
  LC_HIP(hipStreamSynchronize(stream_));
  for (int k = 0; k < K; ++k) {
    const double* rec = hss_.data() + (size_t)k * SS;
    const bool off = false;  // (see suffstat)
    if (Nk) Nk[k] = off ? 0.0 : rec[0];
    for (int d = 0; d < D; ++d) {
      if (xs) xs[(size_t)k * D + d] = off ? 0.0 : rec[1 + d];
      if (xxs) xxs[(size_t)k * D + d] = off ? 0.0 : rec[1 + DP + d];
    }
  }
  if (Njk) {
    if (!own_counts)
      for (int k = 0; k < K; ++k) Njk[k] = hss_[(size_t)k * SS];
    else
      std::copy(hss_.begin() + (size_t)K * SS, hss_.end(), Njk);
  }
}

// ---------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------
KernelTimes Context::timing_get() {
  use_device();
  if (!pending_.empty()) LC_HIP(hipStreamSynchronize(stream_));
  for (auto& p : pending_) {
    float ms = 0.f;
    LC_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    if (p.kind == 3) {
      times_.allreduce_ms += ms;
      times_.allreduce_calls += 1;
    } else if (p.kind == 2) {
      times_.fused_ms += ms;
      times_.fused_calls += 1;
    } else if (p.kind == 0) {
      times_.estep_ms += ms;
      times_.estep_calls += 1;
    } else {
      times_.suffstat_ms += ms;
      times_.suffstat_calls += 1;
    }
    evpool_.push_back(p.a);  // (back to the pool: creating an event costs microseconds, and a timed iteration of the
    evpool_.push_back(p.b);  //  small configuration lasts 180 of them)
  }
  pending_.clear();
  return times_;
}

hipEvent_t Context::timing_event() {
  if (!evpool_.empty()) {
    hipEvent_t e = evpool_.back();
    evpool_.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  LC_HIP(hipEventCreate(&e));
  return e;
}

void Context::timing_prepare(int events) {
  use_device();
  while ((int)evpool_.size() < events) {
    hipEvent_t e = nullptr;
    LC_HIP(hipEventCreate(&e));
    evpool_.push_back(e);
  }
}

void Context::timing_reset() {
  (void)timing_get();
  times_ = KernelTimes();
}

}  // namespace lcc

namespace lck {
const char* test_switch(const char* name) {
#ifdef LC_TEST_HOOKS
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
}  // namespace lck

#ifdef LC_TEST_HOOKS
// libcluster_hip_testhooks.so only (tests/test_gpu_comm.py): the device-side rank-order sum of LIBCLUSTER_COMM=rccl-gather
// on host data -- `world` blocks of `count` doubles in, `count` out.  0 on success, the HIP error code otherwise.
extern "C" __attribute__((visibility("default"))) int lc_test_rank_order_sum(const double* in, int world, long long count, double* out) {
  if (!in || !out || world < 1 || count < 1) return -1;
  double *din = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&din), (size_t)world * (size_t)count * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dout), (size_t)count * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(din, in, (size_t)world * (size_t)count * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = lck::launch_rank_order_sum(din, world, count, dout, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)count * sizeof(double), hipMemcpyDeviceToHost);
  if (din) (void)hipFree(din);
  if (dout) (void)hipFree(dout);
  return (int)e;
}
#endif
