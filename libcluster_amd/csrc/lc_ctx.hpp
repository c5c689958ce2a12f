// Device context: owns the device-resident copy of one data set (all groups,
// padded), the responsibility matrix qZ and every workspace of the hot path,
// and drives the kernels of lc_kernels_*.hip.  One context per learn*() call
// (and one per split sub-problem); contexts are independent.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "lc_comm.hpp"
#include "lc_kernels.h"

namespace lcc {

struct HipFailure : std::runtime_error {
  explicit HipFailure(const std::string& s) : std::runtime_error(s) {}
};
// hipMalloc said no (after the block cache was trimmed): optional accelerations catch this one and take their ordinary path
struct AllocFailure : HipFailure {
  explicit AllocFailure(const std::string& s) : HipFailure(s) {}
};
// Context::estep_cache: the device has no room for the distance cache at this width (nothing was changed; the caller
// runs the ordinary E-step instead)
struct CacheNoRoom : std::runtime_error {
  explicit CacheNoRoom(const std::string& s) : std::runtime_error(s) {}
};

// all-reduce hook: sum `count` doubles in place across ranks (device buffer),
// enqueued on / ordered with `stream`.  Returns 0 on success.
typedef int (*allreduce_fn)(void* user, void* device_buf, int64_t count, void* stream);

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;  // elements
  int device = -1;  // the GPU the block lives on (tagged when it is taken, so it returns to the right cache bin)
  ~DevBuf() { release(); }
  void release();  // back to the block cache (lc_ctx.cpp), not to the driver
  // grow (contents NOT preserved)
  void reserve(size_t n);
};

// page-locked host staging buffer (grow-only): asynchronous copies to / from it do not bounce through the
// driver's own pinned pool, which is what a pageable std::vector costs on every small per-iteration transfer
struct PinnedBuf {
  double* p = nullptr;
  size_t cap = 0, n = 0;
  ~PinnedBuf();
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  void resize(size_t count);  // contents NOT preserved when it grows
  void assign(size_t count, double v) {
    resize(count);
    for (size_t i = 0; i < count; ++i) p[i] = v;
  }
  double* data() { return p; }
  const double* data() const { return p; }
  size_t size() const { return n; }
  double& operator[](size_t i) { return p[i]; }
  const double& operator[](size_t i) const { return p[i]; }
  double* begin() { return p; }
  double* end() { return p + n; }
};

// an ordered set of (padded, global) row indices of one context, kept on the device
struct RowSelection {
  DevBuf<int64_t> idx;          // [M] ascending
  DevBuf<int64_t> starts_d;     // [J+1] first position of every group inside idx
  std::vector<int64_t> starts;  // host copy
  int64_t M = 0;
};

// return every cached device / page-locked block to the driver
void trim_cache();
// while one of these lives on the calling thread, a device allocation may be served by a cached block up to 16 times
// the request (2 otherwise): for the short-lived buffers of sub-problems
struct RelaxedFit {
  RelaxedFit();
  ~RelaxedFit();
  RelaxedFit(const RelaxedFit&) = delete;
  RelaxedFit& operator=(const RelaxedFit&) = delete;
};
// Cached blocks are handed back only to the host thread that released them (one thread = one stream = ordered re-use).
// A thread that is about to end calls this AFTER synchronising its stream: its blocks become available to everybody.
void cache_release_thread();

struct KernelTimes {
  double estep_ms = 0, suffstat_ms = 0, fused_ms = 0;  // fused: E-step + statistics in one launch (small observations)
  int64_t estep_calls = 0, suffstat_calls = 0, fused_calls = 0;
  int64_t estep_diag_mfma_calls = 0;  // separable families: E-step launches that took estep_diag_mfma_kernel (of estep_calls)
  // the exchange step (events around the collective on the context's stream: the sum itself plus the wait for the
  // slowest rank to arrive) and the host's wall time per phase of a VBEM iteration (vbem): what a multi-GPU run
  // needs to tell a slow collective from a slow M-step from a straggling rank
  double allreduce_ms = 0;
  int64_t allreduce_calls = 0;
  double host_stats_ms = 0, host_mstep_ms = 0, host_estep_ms = 0, host_fenergy_ms = 0;
  int64_t host_iters = 0;
};

class Context {
 public:
  Context(int device, hipStream_t stream);
  ~Context();
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;

  // ---- data ---------------------------------------------------------------
  // Host groups -> device (row-major, padded).  Element (n,d) of group j is at
  // Xj[j][n*row_stride + d*col_stride].
  void set_data(int J, const double* const* Xj, const int64_t* Nj, int D, int64_t row_stride, int64_t col_stride);
  // Single-group synthetic mixture generated on the device (+ initial qZ, K columns)
  void synth(int64_t N, int D, int K, const double* mu, const double* L, uint64_t seed, int64_t row_offset,
             double hard);
  // J groups; cdf: J x K cumulative mixing proportions (host) or null; group_ids: J global ids (Philox
  // counters: a group looks the same on whichever rank generates it) or null = 0..J-1
  void synth_groups(int J, const int64_t* Nj, int D, int K, const double* mu, const double* L, const double* cdf,
                    uint64_t seed, const int64_t* group_ids, int64_t row_offset, double hard);
  // rows [row0, row0+n) of group j -> row-major n x D host buffer
  void get_rows(int j, int64_t row0, int64_t n, double* out) const;

  int J() const { return J_; }
  int D() const { return D_; }
  int DP() const { return DP_; }
  int K() const { return qz_[cur_].K; }
  int64_t N(int j) const { return Nj_[j]; }
  int64_t Ntotal() const { return Ntot_; }
  int64_t NP() const { return NP_; }
  int device() const { return device_; }
  hipStream_t stream() const { return stream_; }
  void set_stream(hipStream_t s) { stream_ = s; }
  void synchronize() const;
  void set_allreduce(allreduce_fn f, void* user) {
    ar_fn_ = f;
    ar_user_ = user;
  }
  // native collective (RCCL or host-staged, lc_comm.hpp); takes precedence over the hook
  void set_comm(std::shared_ptr<lcm::Comm> c) { comm_ = std::move(c); }
  const std::shared_ptr<lcm::Comm>& comm() const { return comm_; }
  bool distributed() const { return ar_fn_ != nullptr || comm_ != nullptr; }
  // sum of one host value over all ranks (identity without a hook); via the device hook
  double allreduce_value(double v);
  void allreduce_values(double* v, int n);
  // Sharding of a distributed run: false (default) = every rank holds rows of the SAME groups (BGMM/VDP
  // row blocks): counts are summed; true = every rank holds WHOLE, different groups (GMC): the per-group
  // counts stay local, only cluster statistics and scalars are summed.
  void set_group_sharded(bool on) { group_sharded_ = on; }
  // statistics pass: skip (4-row step, cluster) pairs whose responsibilities are all exactly 0.0 (bit-identical
  // results; pays off once qZ is mostly hard).  Always on in sparse mode.
  void set_skip_zero(bool on) { skip_zero_ = on; }
  bool group_sharded() const { return group_sharded_ && distributed(); }
  // a context for a sub-problem of this one: same device, stream and all-reduce hook
  void inherit_comm(const Context& parent) {
    ar_fn_ = parent.ar_fn_;
    ar_user_ = parent.ar_user_;
    comm_ = parent.comm_;
    group_sharded_ = parent.group_sharded_;
    skip_zero_ = parent.skip_zero_;
  }

  // ---- qZ -----------------------------------------------------------------
  void qz_fill(int K, double value);  // K columns = value on valid rows
  void qz_set(int j, const double* q, int K, int64_t row_stride, int64_t col_stride);
  void qz_get(int j, double* q, int64_t row_stride, int64_t col_stride) const;
  void qz_get_all(double* out) const;  // [Ntotal x K] row-major, groups concatenated
  // out[j] = [N_j x K] column-major (Eigen's default layout); ld[j] (optional) = column stride of out[j] when the
  // group's rows are a block of a taller matrix (row-sharded learners), default N_j
  void qz_get_all_colmajor(double* const* out, const int64_t* ld = nullptr) const;
  void qz_get_rows(int j, int64_t row0, int64_t n, double* q, int64_t row_stride, int64_t col_stride) const;
  void qz_get_column(int j, int k, double* out) const;  // N(j) doubles
  void qz_keep_columns(const std::vector<int>& keep);    // prune_clusters
  void qz_clone_to_alt();                                // alt <- copy of current (capacity K+1)
  void qz_swap_alt();                                    // current <-> alt
  // ---- split search on the device (partobs / splitobs / auglabels) ----------
  // rows with qZ[.,k] > thresh, in order (partobs' index part, comutils.cpp:56-72)
  void select_rows(int k, double thresh, RowSelection& sel);
  void select_rows_col(const double* col, double thresh, RowSelection& sel);  // ... of any device column of NP values
  // this context := the selected rows of src, group structure kept (partobs' copy part)
  void set_data_gather(const Context& src, const RowSelection& sel);
  // qZ := the selected rows of column `col` of src's current qZ (one column; this context holds src's selected rows)
  void qz_gather_column(const Context& src, const RowSelection& sel, int col);
  // qZ := [s, 1-s], s = ((x-m).v >= 0)   (splitobs + cluster.cpp:446-449); m, v: D host doubles
  void qz_init_split(const double* m, const double* v);
  // ExpGamma::splitobs (distributions.cpp:575-581): s = (x.v > mean over the group's rows of x.v)
  void qz_init_split_mean(const double* v);
  // auglabels (comutils.cpp:75-104) on the current buffer (K -> K+1 columns): selected rows whose
  // refined second responsibility in `sub` exceeds 0.5 move their column-k mass to the new column
  void qz_split_from(const Context& sub, const RowSelection& sel, int k);

  // ---- hot path -------------------------------------------------------------
  // A: K x D x D row-major lower-triangular whiteners, m: K x D, c: J x K.
  // Writes new responsibilities into qZ (K columns).  Fz = -sum logZ; LLk[k] =
  // sum_n q_nk (log q~_nk - c_jk)  (the data term of cluster.cpp:409-410).
  // raw = true stops after log q~ (c_jk - 0.5 d^2) has been written to qZ: GaussWish::Eloglike.
  // target (raw mode only): write the K columns of c_jk - 0.5 d^2 there (leading dimension NP) instead of into qZ
  void estep(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk, bool raw = false,
             double* target = nullptr);
  // Model selection (cluster.cpp:564-629): the distances -0.5 d^2_k(x_n) of the clusters, kept on the device
  // [K x NP], every column tagged on the host with the whitener and mean it was computed from.  An E-step through the
  // cache recomputes the columns whose tag differs IN ANY BIT from the cluster it is asked about (raw estep_kernel),
  // then adds the constants c_jk and normalises over all K columns (softmax_cached_kernel): the same outputs as
  // estep().  Between two iterations of a converging model, and between the candidates of a split round, most
  // clusters' posteriors are bit for bit the same, so most columns are re-used.  Returns the number of recomputed
  // columns; *stale = how many of them had a valid (but different) tag.
  // delta_tol >= 0: also leave every row's largest |q_new - q_old| (against the responsibilities being overwritten,
  // when the buffer held K columns) and, for the rows where it exceeds delta_tol, q_new - q_old, on the device for
  // delta_suffstat(); rows that come out bit for bit as they were are not even written.
  bool dcache_eligible(int K) const;  // a property of the shape (identical on every rank)
  int estep_cache(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk, double delta_tol,
                  int* stale = nullptr);
  // a split candidate works on the cache in place: from journal_begin() on, a column is copied aside before it is
  // first overwritten; rollback() puts columns, tags and width back, journal_end() keeps the new state
  void dcache_journal_begin();
  void dcache_rollback();
  void dcache_journal_end();
  void dcache_invalidate();
  void dcache_release();  // ... and give the memory back
  // The CHANGE of the statistics caused by the last estep_cache(delta_tol = tau): sum over the rows whose
  // responsibilities moved by more than tau in some column of (q_new - q_old) x the usual terms -- the statistics are linear in q, so
  // adding it to the statistics of q_old gives those of q_new up to tau * sum_n |x_n x_n^T| (rows that moved by <= tau
  // are left out).  The rows are compacted in order and gathered into a sub-context (as partobs does), so the cost is
  // proportional to the rows that moved.  Returns false, having computed nothing, when more than max_frac of all rows
  // (summed over ranks) moved -- the caller then runs the ordinary pass.
  bool delta_suffstat(int K1, double max_frac, double* dNk, double* dxs, double* dxxs, double* dNjk);
  int delta_pending() const { return dq_K_; }  // width of the move the last estep_cache(delta_tol >= 0) left (0: none)
  int64_t delta_rows() const { return delta_rows_; }  // rows the last delta_suffstat() found moved (this rank)
  // Small observations (D <= 16, K <= 16, Gauss-Wishart, dense): the E-step AND the statistics of the responsibilities it
  // produces, in one pass (lc_kernels_fused.hip).  Same outputs as estep() followed by suffstat(nullptr, ...).
  // Returns false, having done nothing, when the shape has no fused path.
  bool estep_suffstat_fused(int K, const double* A, const double* m, const double* c, double* Fz, double* LLk,
                            double* Nk, double* xs, double* xxs, double* Njk);
  // Sufficient statistics of the current qZ.  smask: J x K (1 = accumulate) or null.
  // Nk[K], xs[K*D], xxs[K*D*D] (row-major, symmetric), Njk[J*K].
  void suffstat(const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk);
  void colsums(double* Njk);  // J x K column sums of the current qZ
  // Diagonal / exponential families: log q~[n,k] = c_jk + sum_d (w2_kd (x_nd - a_kd)^2 + w1_kd x_nd);
  // a, w2, w1: K x D host arrays.  Same outputs as estep().
  void estep_diag(int K, const double* a, const double* w2, const double* w1, const double* c, double* Fz,
                  double* LLk, bool raw = false);
  // N_k, x_s = sum q x [K*D], xx_s = sum q x^2 [K*D] (elementwise), Njk[J*K]
  void suffstat_diag(const unsigned char* smask, double* Nk, double* xs, double* xxs, double* Njk);

  // Host work to run while the NEXT normalising E-step is on the device: the E-step calls it once, after its last
  // launch and before it waits for the stream (vbem hands over the free-energy terms that depend on the posteriors
  // only).  A hook that is still pending when the E-step returns was not called; the owner runs it itself.
  void set_overlap(std::function<void()> f) { overlap_ = std::move(f); }
  bool overlap_pending() const { return (bool)overlap_; }
  void run_overlap() {
    if (!overlap_) return;
    std::function<void()> f = std::move(overlap_);
    overlap_ = nullptr;
    f();
  }

  // ---- timing ---------------------------------------------------------------
  void timing_enable(bool on) {
    timing_ = on;
    if (on) timing_prepare(1024);  // (two events per timed launch: created here, outside anybody's timed region)
  }
  bool timing_enabled() const { return timing_; }
  void timing_host_phases(double stats_ms, double mstep_ms, double estep_ms, double fenergy_ms) {  // one VBEM iteration
    times_.host_stats_ms += stats_ms;
    times_.host_mstep_ms += mstep_ms;
    times_.host_estep_ms += estep_ms;
    times_.host_fenergy_ms += fenergy_ms;
    times_.host_iters += 1;
  }
  KernelTimes timing_get();  // resolves pending events (synchronises the stream)
  void timing_prepare(int events);  // events created ahead of a timed region (they are recycled afterwards)
  void timing_reset();

 private:
  struct QZ {
    DevBuf<double> buf;
    int cap = 0;  // columns allocated
    int K = 0;    // columns in use
    // cluster(): a 64-bit fingerprint per row of what the buffer holds (softmax_cached_kernel keeps it, and skips the
    // old values of the rows it leaves unchanged).  Every other writer of the buffer goes through ensure_qz(), which
    // clears hash_ok; qz_clone_to_alt copies it, qz_split_from marks the rows it rewrites.
    DevBuf<int64_t> hash;
    bool hash_ok = false;
  };
  void ensure_qz(QZ& q, int K, bool preserve);
  void build_layout(int J, const int64_t* Nj, int D);
  void allreduce(double* dbuf, int64_t count);
  // pack the E-step parameter stream (tiles of A_k in consumption order, -b_k; then the J x K table c) into hpack_
  void pack_estep_params(int K, const double* A, const double* m, const double* c);
  void use_device() const;         // hipSetDevice(device_): every method that allocates, launches or copies starts here
  void require_gw_width() const;  // throws for DP > 128 (full-covariance kernels)
  int build_sparse_worklist(const unsigned char* smask, int K, int64_t SS, lck::SuffstatLaunch& a);

  int device_;
  hipStream_t stream_;
  allreduce_fn ar_fn_ = nullptr;
  void* ar_user_ = nullptr;
  std::shared_ptr<lcm::Comm> comm_;
  bool group_sharded_ = false;
  bool skip_zero_ = false;

  int J_ = 0, D_ = 0, DP_ = 0;
  int DC_ = 0;  // active width of the Gauss-Wishart E-step / feature-GEMM statistics (lck::estep_active_width): DP_ - 8 or DP_
  std::vector<int64_t> Nj_, goff_;  // goff_: padded row offsets, size J+1
  int64_t Ntot_ = 0, NP_ = 0;
  DevBuf<double> X_;
  DevBuf<int> rginfo_;      // only when J > 1
  DevBuf<int64_t> goff_d_;  // J+1
  DevBuf<unsigned char> ssitems_;  // sparse statistics: work items, and [klist | kptr | krec]
  DevBuf<int> ssints_;
  const int* sskptr_ = nullptr;
  const int* sskrec_ = nullptr;
  QZ qz_[2];
  // distance cache: slab of dc_cap_ column SLOTS [dc_cap_ x NP]; cluster k's column of -0.5 d^2 lives in slot dc_slot_[k]
  // (k < dc_K_), host tags say for which posterior it was computed; scratch for recomputed columns that find no run of
  // free slots.  Round 5: a column that changes under the journal (a split candidate's trial) is written into a FREE
  // slot and the map is switched -- the old slot is what a rollback returns to, a commit frees it: no column is ever
  // copied (round 4: 2 456 device-to-device copies of 80 MB per model selection at N = 10M, 66 ms).
  DevBuf<double> dc_slab_, dfresh_;
  int dc_cap_ = 0, dc_K_ = 0;
  int dc_room_K_ = 0;  // widest K for which all ranks found room
  std::vector<int> dc_slot_;            // [dc_K_] slot of cluster k's column
  std::vector<unsigned char> dc_used_;  // [dc_cap_] slot in use (by the map or by the journal)
  std::vector<std::vector<double>> dc_tagA_, dc_tagm_;
  struct SavedColumn {
    int col = -1;
    int slot = -1;  // where the column's journaled content stays
    uint64_t ver = 0;
    std::vector<double> A, m;
  };
  // which rows a recomputed column matters for (lck::BoundSelectLaunch): every sweep leaves each row's largest log q~ and
  // its cluster; valid for cluster j while column j is the version it was then (dc_ver_)
  std::vector<uint64_t> dc_ver_;  // [dc_K_] version of cluster k's column (bumped on every recomputation)
  uint64_t dc_vernext_ = 0;
  DevBuf<double> rm_max_, bs_need_, bs_x_, bs_out_;
  DevBuf<int> rm_arg_;
  bool rm_valid_ = false;
  int rm_K_ = 0;
  std::vector<uint64_t> rm_ver_;
  std::vector<double> rm_c_;
  int64_t clone_resyncs_ = 0, clone_fulls_ = 0;  // split trials whose working copy was re-synchronised row by row / copied whole (trace)
  int64_t bound_rows_ = 0, bound_passes_ = 0;  // rows recomputed / passes taken by the bounded recomputation (trace)
  bool bound_static_ok() const;  // the bounded recomputation's shape preconditions (one group, one rank, enough rows)
  bool recompute_bounded(int K, const std::vector<int>& changed, const std::vector<int>& oldslot,
                         const std::vector<std::vector<double>>& oldA, const std::vector<std::vector<double>>& oldm,
                         const std::vector<int>& dest, const double* A, const double* m, const double* c, bool delta);
  int dc_find_run(int n) const;  // first run of n free slots (-1: none)
  std::vector<std::unique_ptr<SavedColumn>> dc_saved_;
  bool dc_journal_ = false;
  int dc_jK0_ = 0;
  DevBuf<double> sink_;  // estep_diag_mfma_kernel's store sink
  DevBuf<double> dq_, amax_;  // estep_cache(delta_tol): q_new - q_old [K x NP] (moved rows), per-row max |.|
  DevBuf<int64_t> dq_maskd_;  // [2] device: columns with any non-zero difference (CachedNormLaunch::colmask)
  DevBuf<int> dq_colsd_;
  uint64_t dq_mask_[2] = {0, 0};
  bool dq_mask_ok_ = false;
  double dq_tol_ = 0.0;
  int dq_K_ = 0;
  int64_t dq_ld_ = 0;  // row stride of dq_ (row-major: a moved row's differences are contiguous)
  int64_t delta_rows_ = 0;
  int cur_ = 0;

  DevBuf<double> params_, ctab_, fzpart_, llpart_, red_, redtmp_, sspart_, ssout_, ssext_;
  // sspart_ holds zeros in every Gauss-Wishart record entry of the columns past the active width (suffstat()): true once
  // it has been cleared and until it is re-allocated or lent to another record layout
  bool sspart_clean_ = false;
  void sspart_reserve(size_t n) {
    if (n > sspart_.cap) {
      sspart_.reserve(n);
      sspart_clean_ = false;
    }
  }
  DevBuf<unsigned char> smask_;
  DevBuf<int> selcnt_;
  DevBuf<int64_t> seloff_;
  DevBuf<double> mv_;
  PinnedBuf hpack_, hred_, hss_, hmask_;
  std::function<void()> overlap_;

  bool timing_ = false;
  struct EvPair {
    hipEvent_t a, b;
    int kind;
  };
  std::vector<EvPair> pending_;
  std::vector<hipEvent_t> evpool_;  // recycled timing events
  hipEvent_t timing_event();
  KernelTimes times_;
};

}  // namespace lcc
