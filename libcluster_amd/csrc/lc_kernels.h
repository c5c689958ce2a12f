// Internal launcher interface between the C-ABI layer and the gfx950 kernels.
// Everything here is plain pointers + sizes; all pointers are DEVICE pointers
// unless a name ends in _h.  Launchers enqueue on `stream` and do not sync.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lck {

// ---- data layout constants -------------------------------------------------
// X on device: row-major [NP x DP] doubles, DP = D rounded up to a multiple of 16 up to 128 (a multiple of 64 beyond that),
// pad columns zero.  Groups are padded to multiples of 16 rows (a "row-group"),
// pad rows zero.  qZ on device: column-major, qZ[k*ldq + row], ldq = NP.
constexpr int RG = 16;  // rows per row-group (one MFMA column block)

inline int padded_dim(int D) {
  return D <= 16 ? 16 : D <= 32 ? 32 : D <= 48 ? 48 : D <= 64 ? 64 : D <= 80 ? 80 : D <= 96 ? 96 : D <= 112 ? 112 : D <= 128 ? 128 : -1;
}
// LDS row stride (doubles) of a staged X batch in the statistics kernels: DP + 16 puts the two rows a ds_read_b64
// half-wave touches on disjoint banks when the stride is 16 mod 32; 48, 80 and 112 need 32 more for that (16 keeps
// the stride it was tuned with)
// (DP = 16: no padding at all -- 32 dwords per row put the two rows of a half-wave on the two halves of the 64 banks;
// the DP + 16 = 32 doubles of round 2 put them on the SAME banks: the bank conflicts of suffstat_kernel<16> in its profile)
__host__ __device__ constexpr int lds_row_stride(int DP) { return DP == 16 ? 16 : (DP + 16) % 32 != 16 ? DP + 32 : DP + 16; }
// wider observations are padded to a multiple of 64 columns: the separable (diagonal / exponential) families process
// them in 128-column blocks with a possible half block at the end (any D), the Gauss-Wishart kernels in 64-column
// panels / 64 x 64 whitener blocks
inline int padded_dim_wide(int D) { return D <= 128 ? padded_dim(D) : (D + 63) / 64 * 64; }
constexpr int GW_MAX_DP = 1024;  // Gauss-Wishart kernels: panel / chunk streaming beyond 128 (tested to 512)
inline int ntiles(int DP) { int nt = DP / 4; return nt * (nt + 1) / 2; }
// doubles per cluster in the packed E-step parameter stream
inline int pstride(int DP) { return ntiles(DP) * 16 + DP; }
// wide observations (DP > 128): the whitener streams as 64 x 64 blocks (I, J <= I), row-major; one chunk = 256 tiles
// of 16 doubles + the 64 entries of -b_I
constexpr int WIDE_CHUNK = 256 * 16 + 64;
inline int wide_chunks(int DP) { int np = DP / 64; return np * (np + 1) / 2; }
// doubles per cluster in the packed E-step parameter stream (the constant table c_jk follows K of these): the ONE
// place that knows both layouts -- every packer and every launch that places `ctab` asks here
// (narrow layouts: the stream holds the tiles of the ACTIVE width DC <= DP only -- estep_active_width)
inline int64_t estep_pstride(int DP, int DC) { return DP > 128 ? (int64_t)wide_chunks(DP) * WIDE_CHUNK : (int64_t)pstride(DC); }
inline int64_t estep_pstride(int DP) { return estep_pstride(DP, DP); }
// Active width of the Gauss-Wishart E-step and feature-GEMM statistics (round 6): the columns D ... DP - 1 of the padded
// layout are zero, and both kernels work in 4-column tiles / patches -- they walk the tiles of DC = D rounded up to a
// multiple of 8 (of 4 up to 48 columns) instead of those of DP (D = 23: 21 whitener tiles per cluster instead of 36, 24 feature tiles instead of
// 39).  Same X layout, same records (the entries of the idle columns are never written: zero), same results bit for bit
// (the skipped products are 0 * 0).  DP = 16 has the fused pass's own narrow instances (NTA).
// (granularity: 4 columns at the 32- and 48-column layouts, where the padding is the larger share -- D = 17 walks 20 columns,
//  not 24 --, 8 columns beyond: one more instance per layout there instead of three)
inline int estep_active_width(int D, int DP) {
  if (DP < 32 || DP > 128) return DP;
  const int g = DP <= 48 ? 4 : 8, lo = DP <= 48 ? DP - 12 : DP - 8;
  const int dc = (D + g - 1) / g * g;
  return dc < DP ? (dc < lo ? lo : dc) : DP;
}
// doubles per cluster in a stats record: [N_k, s_k[DP], S_k[DP*DP]]
inline int64_t stat_stride(int DP) { return 1 + (int64_t)DP + (int64_t)DP * DP; }

// Kernels that need more than 64 KB of dynamic LDS must be told so once per device (function attributes are per
// device; one process may hold contexts on several).  granted = a static per launcher instance.
struct LdsGrant {
  size_t granted[16] = {};
};
inline hipError_t grant_dynamic_lds(const void* fn, size_t bytes, LdsGrant& g) {
  if (bytes <= 64 * 1024) return hipSuccess;
  int dev = -1;
  const bool tracked = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16;
  if (tracked && bytes <= g.granted[dev]) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && tracked) g.granted[dev] = bytes;
  return e;
}

// rginfo word per row-group: (group << 5) | nvalid   (nvalid in 0..16)
inline int rginfo_pack(int group, int nvalid) { return (group << 5) | nvalid; }

struct EstepLaunch {
  int DP;
  int DC = 0;            // active width (estep_active_width; 0 = DP): tiles of the columns >= DC are neither packed nor walked
  const double* X;       // [NP x DP]
  int64_t nrg;           // number of row-groups (NP / 16)
  const int* rginfo;     // [nrg] or nullptr (single group; nvalid from nrows)
  int64_t nrows;         // valid rows when rginfo == nullptr
  const double* params;  // [K x estep_pstride(DP, DC)] packed tiles + b
  const double* ctab;    // [J x K] c_jk (may hold -inf for sparse-inactive)
  int K;
  double* qZ;            // [K x ldq]
  int64_t ldq;
  double* fz_part;       // [estep_grid(...)]
  double* ll_part;       // [estep_grid(...) x K], or nullptr: skip the split-ordering data term
  int raw = 0;           // 1: stop after writing log q~ (no log-sum-exp, fz/ll untouched)
  int sparse = 0;        // 1: ctab holds -inf entries; waves skip clusters inactive for all their rows
  int lq_lds = 0;        // filled in by launch_estep: log q~ waits in LDS (D <= 48, small K) instead of in qZ
};
int estep_rows_per_block(int DP);
// blocks of the launch (and partial sums the caller provides): a function of the WHOLE launch -- DP, DC, K, raw, sparse, nrg
// have to be set (D = 64 / 80 run four row groups per wave where the log q~ table fits in LDS: other rows per block)
int64_t estep_grid(const EstepLaunch& a);
hipError_t launch_estep(const EstepLaunch& a, hipStream_t stream);

// ---- small observations: E-step + the next iteration's statistics in one persistent pass (lc_kernels_fused.hip) ----
struct FusedLaunch {
  int DP;
  int D = 16;            // observation width before padding (D <= 8: the half-width instance)
  const double* X;       // [NP x DP]
  int64_t nrg;           // row groups (NP / 16)
  const int* rginfo;     // [nrg] or nullptr (single group)
  int64_t nrows;         // valid rows when rginfo == nullptr
  const double* params;  // [K x pstride(DP)] as for estep_kernel
  const double* ctab;    // [J x K]
  int K;
  double* qZ;            // [K x ldq]: written once (the new responsibilities)
  int64_t ldq;
  // ONE partial record per block, folded by a single reduce_partials launch:
  //   [K x stat_stride(DP) statistics of the NEW responsibilities | Fz | LL_k (K; zeros when !want_ll)]
  double* partial;       // [grid x fused_record(DP, K)]
  bool want_ll;          // also the split-ordering data term
  int grid;              // fused_plan(...)
  int ngroups = 1;       // J: rows of ctab (the table waits in LDS when J x K <= FUSED_CT_CAP)
  // share (per mille) of a CU's tiles that goes to the SECOND of its two blocks (blocks grid / 2 ... grid - 1); 0: every
  // block the same (launch_fused sets it; the kernel's tile deal is a function of (blockIdx, grid, nrg) alone)
  int yshare = 0;
};
constexpr int FUSED_CT_CAP = 1024;
inline int64_t fused_record(int DP, int K) { return (int64_t)K * (1 + DP + (int64_t)DP * DP) + 1 + K; }
bool fused_eligible(int DP, int K);          // a property of the shape alone (identical on every rank)
int fused_plan(int DP, int64_t nrg, int K);  // persistent blocks for nrg row groups (0 when there are none)
hipError_t launch_fused(const FusedLaunch& a, hipStream_t stream);

// sparse work item of suffstat_kernel: rows [r0, r1) of one group x clusters klist[kofs .. kofs + kcnt)
struct SSItem {
  int64_t r0, r1;
  int kofs, kcnt;
  int64_t rec0;  // partial record of the item's first cluster
};
struct SuffstatLaunch {
  int DP;
  int DC = 0;               // active width (estep_active_width; 0 = DP): the feature-GEMM kernel deals out its patches only
  const double* X;
  int64_t NP;               // padded rows (multiple of 16)
  const double* qZ;
  int64_t ldq;
  int K;
  const int* rginfo;        // needed only with smask
  const unsigned char* smask;  // [J x K] 1 = accumulate (sparse), or nullptr
  double* partial;          // [nchunks x K x stat_stride(DP)]
  int nchunks;
  int64_t chunk_rows;       // multiple of 4
  int nslice = 1;           // filled in by launch_suffstat
  int skip_zero = 0;        // 1: use the variant that skips (4-row step, cluster) pairs with all-zero q (exact)
  const SSItem* items = nullptr;  // sparse work list (device) or nullptr: dense (chunk, slice) grid
  const int* klist = nullptr;     // active cluster lists the items point into
  int nitems = 0;
  // ragged K (see suffstat_extra_records): records per chunk = K + extra; 0 = K
  int KR = 0;
  int slice0 = 0, rs = 1, klast0 = 0, nklast = 0;  // filled in by launch_suffstat for the row-split launch of the last slice
  // wide observations (DP > 128), filled in by launch_suffstat: the kernel works on 64-column panels
  int* occ_out = nullptr;         // suffstat_plan's question to the few-cluster kernel: resident blocks per CU of the instance this launch would take (nothing is launched)
  int64_t ldx = 0;                // row stride of X
  int DPW = 0, colA = 0, colB = 0;  // record width, first column of the A-side / B-side panel
};
// choose a chunking for (NP, K); returns nchunks and sets chunk_rows
int suffstat_plan(int DP, int64_t NP, int K, int64_t* chunk_rows, int DC = 0);
hipError_t launch_suffstat(const SuffstatLaunch& a, hipStream_t stream);
int suffstat_clusters_per_block(int DP, int K);  // 4 waves x clusters per wave
const char* suffstat_kernel_name(int DP, int K, int DC = 0);  // "suffstat_kernel" or "suffstat_feat_kernel" (dense pass of this shape)
// When the last cluster slice of the dense pass fills only one or two of its four waves, the idle waves take over part
// of the active waves' rows (2 or 4 row classes) and write partial records of their own: `extra` more records per
// chunk, laid out after the K regular ones ([row class - 1][cluster of the last slice]).  Returns extra (0: no split);
// klast0 = first cluster of the last slice.  launch_fold_extra adds them into their clusters after the reduction.
int suffstat_extra_records(int DP, int K, bool skip_or_items, int* klast0, int DC = 0);
hipError_t launch_fold_extra(double* rec, int64_t SS, int K, int klast0, int extra, hipStream_t stream);
hipError_t launch_reduce_records(const double* partial, int64_t n, int K, const int* kptr, const int* krec, double* out,
                                 hipStream_t stream);

// out[e] = sum_c partial[c*n + e]  (fixed order => deterministic)
// tmp (optional): REDUCE_TMP_ELEMS * 64 doubles of scratch, enables the two-stage path for very many records
constexpr int REDUCE_TMP_ELEMS = 512;
hipError_t launch_reduce_partials(const double* partial, int nparts, int64_t n, double* out, hipStream_t stream,
                                  double* tmp = nullptr);
// out[j*K+k] = sum over rows of group j of qZ[k*ldq + row]; goff = padded row offsets [J+1]
// tmp (optional, REDUCE_TMP_ELEMS * 64 doubles) and rows (total padded rows) enable the sliced path for few large groups
hipError_t launch_group_colsum(const double* qZ, int64_t ldq, int K, const int64_t* goff, int J, double* out,
                               hipStream_t stream, double* tmp = nullptr, int64_t rows = 0);
// qZ[:, 0..K) = value on valid rows, 0 on pad rows
hipError_t launch_fill_qz(double* qZ, int64_t ldq, int K, const int* rginfo, int64_t nrows, int64_t nrg, double value,
                          hipStream_t stream);
// ---- diagonal Gaussian / exponential families (NormGamma, ExpGamma) --------------------------
struct DiagEstepLaunch {
  int DP, D;
  const double* X;
  int64_t nrg;
  const int* rginfo;
  int64_t nrows;
  const double* params;  // [3][K][DP]: a, w2, w1 (padding columns zero)
  int mode = 0;          // 0: general, 1: w1 == 0 (NormGamma), 2: a == w2 == 0 (ExpGamma)
  const double* ctab;    // [J x K]
  int K;
  double* qZ;
  int64_t ldq;
  double* fz_part;       // [ceil(NP/64)]
  double* ll_part;       // [ceil(NP/64) x K] or nullptr
  int raw = 0;
  // matrix-pipe path (estep_diag_mfma_kernel), chosen by lc_ctx.cpp when the expansion around `mu` is well conditioned:
  const double* wt = nullptr;      // packed weight tiles [ceil(K/4)][NTF][16]; nullptr: the difference-form VALU kernel
  const double* mu = nullptr;      // [DP] centre (unused in mode 2)
  const double* constk = nullptr;  // [K] sum_d (w2 a'^2 + w1 mu)
  int ngroups = 1;                 // J: rows of ctab (with several groups the plain instance keeps the table in LDS when J x K <= EDM_CT_CAP)
  double* sink = nullptr;          // [256] scratch: where the lanes of padding clusters / missing row groups store (see the kernel)
};
constexpr int EDM_CT_CAP = 1024;
inline int64_t estep_diag_grid(int64_t nrg) { return (nrg * RG + 63) / 64; }
int64_t estep_diag_mfma_weights(int DP, int K, int mode);
hipError_t launch_estep_diag(const DiagEstepLaunch& a, hipStream_t stream);

struct DiagStatLaunch {
  int DP;
  const double* X;
  int64_t NP;
  const double* qZ;
  int64_t ldq;
  int K;
  const int* rginfo;
  const unsigned char* smask;
  double* partial;       // [nchunks * rsplit x K x (1 + 2 DP)], rsplit = suffstat_diag_rsplit(K)
  int nchunks;
  int64_t chunk_rows;    // multiple of 32
  int second = 1;        // 0: skip the second moments (ExpGamma)
  int nslice = 0, rsplit = 0;  // filled in by launch_suffstat_diag
  int ldx = 0, col0 = 0, DPT = 0;  // ditto: row stride of X, first column and total padded width (wide D)
};
int suffstat_diag_rsplit(int K);
hipError_t launch_suffstat_diag(const DiagStatLaunch& a, hipStream_t stream);

// ---- split-search data passes (partobs / splitobs / auglabels on the device) ----
int select_blocks(int64_t NP);  // number of per-block counts select_count produces
hipError_t launch_select_count(const double* qcol, int64_t NP, double thresh, int* counts, hipStream_t stream);
hipError_t launch_select_compact(const double* qcol, int64_t NP, double thresh, const int64_t* offsets, int64_t* idx,
                                 hipStream_t stream);
hipError_t launch_group_starts(const int64_t* idx, int64_t M, const int64_t* goff, int J, int64_t* starts,
                               hipStream_t stream);
// qT[row * K + k] = qZ[k * ldq + row]
hipError_t launch_transpose_qz(const double* qZ, int64_t ldq, int K, int64_t NP, double* qT, hipStream_t stream);
// dst[c, gathered row of p] = src[c, idx[p]] for K columns (column-major, leading dimensions lds / ldd)
hipError_t launch_gather_cols(const double* src, int64_t lds, int K, const int64_t* idx, int64_t M, const int64_t* starts,
                              const int64_t* goff_sub, int J, double* dst, int64_t ldd, hipStream_t stream);
hipError_t launch_gather_rowmajor(const double* src, int64_t lds, int K, const int64_t* idx, int64_t M,
                                  const int64_t* starts, const int64_t* goff_sub, int J, double* dst, int64_t ldd,
                                  hipStream_t stream);
// the same for the nc source columns cols[0 .. nc) (device array): dst[c, gathered row of p] = src[idx[p], cols[c]]
hipError_t launch_gather_rowmajor_cols(const double* src, int64_t lds, const int* cols, int nc, const int64_t* idx, int64_t M,
                                       const int64_t* starts, const int64_t* goff_sub, int J, double* dst, int64_t ldd,
                                       hipStream_t stream);
hipError_t launch_gather_rows(const double* X, int DP, const int64_t* idx, int64_t M, const int64_t* starts,
                              const int64_t* goff_sub, int J, double* Xdst, hipStream_t stream);
hipError_t launch_split_init(const double* X, int DP, int D, int64_t NP, const int* rginfo, int64_t nrows,
                             const double* mv, double* q, int64_t ldq, int mode, const double* thr,
                             hipStream_t stream);
// qhash (or nullptr): the rows it rewrites lose their fingerprint (CachedNormLaunch::qhash)
hipError_t launch_aug_from_sub(double* q, int64_t ldq, int k, int K, const int64_t* idx, int64_t M,
                               const int64_t* starts, const int64_t* goff_sub, int J, const double* qsub1,
                               hipStream_t stream, int64_t* qhash = nullptr);

// split search: normalise K columns of log q~ = c_jk + (cached | freshly computed) -0.5 d^2 (softmax_cached_kernel)
struct CachedNormLaunch {
  const double* dcache;  // [Kc x ldc] cached -0.5 d^2 of the round's clusters
  int64_t ldc;
  const double* fresh;   // [nfresh x ldf] the recomputed columns
  int64_t ldf;
  const int* colmap;     // [K] device: >= 0 cache column, < 0 fresh column -(v + 1); nullptr: cache column j for cluster j
  const double* ctab;    // [J x K]
  int K;
  const int* rginfo;     // or nullptr (single group)
  int64_t nrows, NP;
  double* qZ;
  int64_t ldq;
  double* fz_part;       // [softmax_cached_grid(NP)]
  // optional (both or neither): how far this E-step moved the responsibilities it overwrites
  double* ll_part = nullptr;  // [softmax_cached_grid(NP) x K] or nullptr: sum_n q_nk (log q~_nk - c_jk) partials
  double* dq = nullptr;    // [NP x ldd], ROW-major (ldd >= K): q_new - q_old, written for the rows with amax > dq_tol only
  double dq_tol = 0.0;
  int64_t ldd = 0;
  double* amax = nullptr;  // [NP] max_j |q_new - q_old| of the row
  // optional, with dq: a 64-bit fingerprint of every row of qZ (QHASH_NONE: unknown).  The sweep writes the fingerprint of
  // what it leaves in a row; with qhash_in it first compares: a row whose new values have the fingerprint of the old
  // ones is unchanged and its K old values are not even read (between the candidates of a split round that is almost
  // every row: half of the sweep's traffic)
  int64_t* qhash = nullptr;  // [NP]
  int qhash_in = 0;          // the fingerprints describe the responsibilities being overwritten
  // optional (both or neither): every row's largest log q~ and the cluster it belongs to -- what the NEXT recomputation of
  // a few columns needs to tell which rows those columns cannot reach (BoundSelectLaunch)
  double* rmax = nullptr;  // [NP]
  int* ramax = nullptr;    // [NP]
  // optional, with dq: bit j of colmask[j / 64] is set when any row's q_new - q_old is non-zero in column j (two words,
  // zeroed by the caller): delta_suffstat forms the moved rows' statistics for those clusters only
  unsigned long long* colmask = nullptr;
};
// Model selection: which rows does a recomputed column of -0.5 d^2 matter for?  (Context::estep_cache, round 5.)
// A split candidate changes two columns and the raw pass that recomputes them reads ALL of X for them (1.7 ms at N = 10M,
// D = 64: X-bound) -- although, for all but the candidate's own rows and their neighbours, the new responsibilities are
// exactly 0.0.  For a reference cluster `ref` (the column's own previous version, or the parent of a split) whose column
// is still in the slab,
//     d^2_new(x) = |B y + b|^2 >= (sigma_min(B) |y| - |b|)^2,   y = A_ref (x - m_ref), B = A_new A_ref^-1, b = A_new (m_ref - m_new)
// bounds the new column from ABOVE by a function of the old one:  log q~_new <= c_new - 0.5 (sigma sqrt(d^2_ref) - |b|)_+^2.
// A row whose largest log q~ over the UNCHANGED columns (rmax + the change of that cluster's constant) exceeds that bound
// by more than T gets -inf in the new column instead of its true value -- the same bits out of the normalisation sweep:
// with T > 745.2 exp(log q~ - max) underflows to 0.0 either way; in the moved-row sweeps, which store q < 2^-300 as zero,
// T > 208 suffices (an addend below 2^-54 of a sum that contains the row's exp(0) = 1 never changes it).  The other rows
// are flagged in `need` (1.0 / 0.0), gathered, and recomputed by the ordinary raw E-step.
constexpr int BOUND_MAX_COLS = 8, BOUND_MAX_K = 72;
struct BoundSelectLaunch {
  int ncol = 0, K = 0;
  int64_t NP = 0;
  const double* ref[BOUND_MAX_COLS];  // [NP] -0.5 d^2 of column t's reference cluster
  double* dest[BOUND_MAX_COLS];       // [NP] where column t goes: -inf is written for the rows that need no value
  double sigma[BOUND_MAX_COLS], bnorm[BOUND_MAX_COLS], cnew[BOUND_MAX_COLS];
  const double* rmax;
  const int* ramax;
  double T;
  unsigned char usable[BOUND_MAX_K];  // cluster j's column (and so rmax of its rows) is what it was when rmax was written
  double dcj[BOUND_MAX_K];            // change of c_j since then
  double* need;                       // [NP] out: 1.0 = recompute this row's columns
};
hipError_t launch_bound_select(const BoundSelectLaunch& a, hipStream_t stream);
// Xdst[p] = X[idx[p]] (rows of DP doubles), p < M
hipError_t launch_gather_rows_plain(const double* X, int DP, const int64_t* idx, int64_t M, double* Xdst, hipStream_t stream);
// dest[t][idx[p]] = src[t * lds + p], t < ncol, p < M
hipError_t launch_scatter_cols(const double* src, int64_t lds, int ncol, double* const* dest_host, const int64_t* idx, int64_t M,
                               hipStream_t stream);
// Switches of the test suite (the literal schedule of the split search, kernel instances forced on or off): read from
// the environment by libcluster_hip_testhooks.so ONLY -- in the shipped library this returns nullptr for every name, so
// no environment variable can change which kernel runs or what a learner returns (defined in lc_ctx.cpp).
const char* test_switch(const char* name);
constexpr int64_t QHASH_NONE = (int64_t)0x8000000000000000ull;
// out[i] = ((in[0][i] + in[1][i]) + in[2][i]) + ... over `world` blocks of `count` doubles, rank 0 first: the additions of
// the host transport (lc_comm.cpp, HostComm::allreduce_sum), on the device, behind ncclAllGather (LIBCLUSTER_COMM=rccl-gather)
hipError_t launch_rank_order_sum(const double* gathered, int world, int64_t count, double* out, hipStream_t stream);
// tests: rows whose stored fingerprint is neither QHASH_NONE nor that of the K values in the buffer are counted in *bad
hipError_t launch_qhash_verify(const double* qZ, int64_t ldq, int K, int64_t NP, const int64_t* qhash,
                               unsigned long long* bad, hipStream_t stream);
// dst row := src row (K values, zeros in dst's columns K .. Kdst-1, the fingerprint) wherever the rows' fingerprints differ
hipError_t launch_qz_resync(double* dst, const double* src, int64_t ldq, int K, int Kdst, int64_t NP, int64_t* dhash,
                            const int64_t* shash, hipStream_t stream);
int64_t softmax_cached_grid(int64_t NP);
int softmax_cached_max_k();  // widest K the sweep is built for
hipError_t launch_softmax_cached(const CachedNormLaunch& a, hipStream_t stream);

// synthetic mixture generator (bench): Philox4x32-10, counter = global row.
struct SynthLaunch {
  int DP, D, K;
  double* X;               // [NP x DP]
  double* qZ;              // [K x ldq] initial responsibilities, or nullptr
  int64_t ldq;
  int64_t nrows;           // valid rows
  int64_t NP;
  int64_t row_offset;      // global index of row 0 (rank sharding)
  uint64_t seed;
  const double* mu;        // [K x D]
  const double* L;         // [K x D x D] lower Cholesky factors, row-major
  double hard;             // q on the true label (0.9); rest (1-hard)/(K-1)
  // grouped data (all null / 0 for a single group): row-group info, padded group offsets, per-group
  // cumulative mixing proportions [J x K] (null = uniform labels), global id of local group 0
  const int* rginfo = nullptr;
  const int64_t* goff = nullptr;
  const double* cdf = nullptr;
  const int64_t* gids = nullptr;  // [J] global group ids (Philox counters), or null: group_base + local index
  int64_t group_base = 0;
};
hipError_t launch_synth(const SynthLaunch& a, hipStream_t stream);

}  // namespace lck
