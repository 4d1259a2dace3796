// smx_api.hip -- C ABI (include/smx.h): validation, plan selection, twiddle-table cache, dispatch.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>

#include "../../include/smx.h"
#include "smx_kernels.h"
#include "smx_tables.h"

using namespace smx;

namespace {

thread_local std::string t_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_err = buf;
  return code;
}

// (the sticky error of somebody else's earlier failure -- e.g. an invalidated stream capture -- is
// cleared first, so that it is not reported as the failure of this launch)
#define HIP_TRY(expr)                                                                    \
  do {                                                                                   \
    (void)hipGetLastError();                                                             \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(SMX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                  __FILE__, __LINE__);                                                   \
  } while (0)

// Plan knobs.  Process-wide defaults (smx_set_option) and, on top of them, a per-thread stack of overrides
// (smx_options_push / smx_options_pop): plan choice is an argument of the calling context, two users of one
// process can run different plans, and nothing a caller memoises per shape goes stale behind its back --
// g_opt_epoch changes with every change of a default.
std::atomic<int> o_nsplit{0}, o_placement{2}, o_force_direct{0}, o_round{512}, o_full8{1}, o_fourstep{1};
std::atomic<int> o_fs_bgroups{0}, o_fold_gradw{0}, o_decim16{1}, o_conv1{1}, o_st_plain{-1};
std::atomic<unsigned long long> g_opt_epoch{1}, g_tab_epoch{1};
constexpr int OPT_DEPTH = 8;
thread_local smx_options t_opt_stack[OPT_DEPTH];
thread_local int t_opt_depth = 0;

smx_options default_opts() {
  smx_options o;
  o.nsplit = o_nsplit.load(); o.placement = o_placement.load(); o.round = o_round.load();
  o.force_direct = o_force_direct.load(); o.full8 = o_full8.load(); o.fourstep = o_fourstep.load();
  o.fs_bgroups = o_fs_bgroups.load();
  o.fold_gradw = o_fold_gradw.load();
  o.decim16 = o_decim16.load();
  o.conv1 = o_conv1.load();
  o.st_plain = o_st_plain.load();
  return o;
}
smx_options cur_opts() { return t_opt_depth > 0 ? t_opt_stack[t_opt_depth - 1] : default_opts(); }

// ---- twiddle cache, keyed by (device, N) ---------------------------------------------------------
// Tables are uploaded with a blocking hipMemcpy the first time an N is seen on a device.  That must
// not happen while `stream` is being captured into a hipGraph (the copy would invalidate the capture):
// such a call fails cleanly and asks for smx_prepare(N) up front.  The cache is bounded: past
// `table_cache_entries` distinct (device, N) the least recently used tables are freed after a device
// synchronise (variable-length workloads); hipGraphs captured with an evicted N must be re-captured,
// so keep the bound above the number of sequence lengths a graph-replaying process uses.
struct Tables { cf* tw = nullptr; cf* bt = nullptr; cf* tq = nullptr; cf* v16 = nullptr; cf* b16 = nullptr; };
struct TableEntry { Tables t; std::map<int, cf*> group_bt; unsigned long long used = 0; int pins = 0; };
std::mutex g_mu;
std::map<std::pair<int, int>, TableEntry> g_tables;
unsigned long long g_tick = 0;
std::atomic<int> o_table_cap{256};

// What get_tables hands to a call: the entry stays PINNED until the call has enqueued its launches (the
// destructor runs when the entry point returns), so another thread's eviction cannot free tables between
// the lookup and the launch; once unpinned, the evictor's hipDeviceSynchronize covers the enqueued work.
struct TableRef : Tables {
  std::pair<int, int> key{-1, -1};
  TableRef() = default;
  TableRef(const TableRef&) = delete;
  TableRef& operator=(const TableRef&) = delete;
  ~TableRef() {
    if (key.first < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tables.find(key);
    if (it != g_tables.end() && it->second.pins > 0) --it->second.pins;
  }
};

bool capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t e = hipStreamIsCapturing(s, &st);
  if (e != hipSuccess) { (void)hipGetLastError(); return true; }   // e.g. legacy stream during a global capture
  return st != hipStreamCaptureStatusNone;
}

// Victims: least recently used, on the CURRENT device only (hipDeviceSynchronize below covers exactly that
// device's queues; entries of other devices wait for a call made there), never pinned, never the entry just made.
void evict_locked(int dev, int keepN) {
  while ((int)g_tables.size() > o_table_cap.load()) {
    auto victim = g_tables.end();
    for (auto it = g_tables.begin(); it != g_tables.end(); ++it)
      if (it->first.first == dev && it->first.second != keepN && it->second.pins == 0 &&
          (victim == g_tables.end() || it->second.used < victim->second.used))
        victim = it;
    if (victim == g_tables.end()) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(victim->second.t.tw);
    if (victim->second.t.bt) (void)hipFree(victim->second.t.bt);
    if (victim->second.t.tq) (void)hipFree(victim->second.t.tq);
    if (victim->second.t.v16) (void)hipFree(victim->second.t.v16);
    if (victim->second.t.b16) (void)hipFree(victim->second.t.b16);
    for (auto& kv : victim->second.group_bt) (void)hipFree(kv.second);
    g_tables.erase(victim);
    g_tab_epoch++;
  }
}

int get_tables(int N, TableRef* out, hipStream_t s) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find({dev, N});
  if (it == g_tables.end()) {
    if (capturing(s))
      return fail(SMX_ERR_UNSUPPORTED,
                  "twiddle tables for N=%d are not on device %d yet and the stream is being captured: "
                  "call smx_prepare(%d) (or run one eager call of this shape) before the capture", N, dev, N);
    TableEntry e;
    std::vector<cf> tw = make_tw(N);
    HIP_TRY(hipMalloc((void**)&e.t.tw, tw.size() * sizeof(cf)));
    HIP_TRY(hipMemcpy(e.t.tw, tw.data(), tw.size() * sizeof(cf), hipMemcpyHostToDevice));
    if (N % M == 0) {
      std::vector<cf> bt = make_bt(N, N / M);
      HIP_TRY(hipMalloc((void**)&e.t.bt, bt.size() * sizeof(cf)));
      HIP_TRY(hipMemcpy(e.t.bt, bt.data(), bt.size() * sizeof(cf), hipMemcpyHostToDevice));
      std::vector<cf> tq = make_tq(N);
      HIP_TRY(hipMalloc((void**)&e.t.tq, tq.size() * sizeof(cf)));
      HIP_TRY(hipMemcpy(e.t.tq, tq.data(), tq.size() * sizeof(cf), hipMemcpyHostToDevice));
    }
    if (N % 16 == 0 && N % M != 0) {           // sixteen-row decimation (make_plan)
      std::vector<cf> v = make_v16(N), bb = make_b16(N);
      HIP_TRY(hipMalloc((void**)&e.t.v16, v.size() * sizeof(cf)));
      HIP_TRY(hipMemcpy(e.t.v16, v.data(), v.size() * sizeof(cf), hipMemcpyHostToDevice));
      HIP_TRY(hipMalloc((void**)&e.t.b16, bb.size() * sizeof(cf)));
      HIP_TRY(hipMemcpy(e.t.b16, bb.data(), bb.size() * sizeof(cf), hipMemcpyHostToDevice));
    }
    g_tables[{dev, N}] = e;
    evict_locked(dev, N);
    it = g_tables.find({dev, N});
  }
  it->second.used = ++g_tick;
  ++it->second.pins;
  out->tw = it->second.t.tw; out->bt = it->second.t.bt; out->tq = it->second.t.tq;
  out->v16 = it->second.t.v16; out->b16 = it->second.t.b16;
  out->key = {dev, N};
  return SMX_OK;
}

// residue twiddles of band group `group` (>= 1; group 0 is Tables::bt); get_tables(N) came first
int get_group_bt(int N, int group, cf** out, hipStream_t s) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find({dev, N});
  if (it == g_tables.end()) return fail(SMX_ERR_INVALID, "internal: tables of N=%d missing", N);
  auto gt = it->second.group_bt.find(group);
  if (gt != it->second.group_bt.end()) { *out = gt->second; return SMX_OK; }
  if (capturing(s))
    return fail(SMX_ERR_UNSUPPORTED,
                "band-group tables for N=%d are not on the device yet and the stream is being captured: "
                "run one eager call of this shape before the capture", N);
  std::vector<cf> bt = make_bt(N, N / M, group);
  cf* p = nullptr;
  HIP_TRY(hipMalloc((void**)&p, bt.size() * sizeof(cf)));
  HIP_TRY(hipMemcpy(p, bt.data(), bt.size() * sizeof(cf), hipMemcpyHostToDevice));
  it->second.group_bt[group] = p;
  *out = p;
  return SMX_OK;
}

// ---- plan ----------------------------------------------------------------------------------------
struct Plan {
  int path, k, L, nb, nsplit, lc, nwg;
  int groups;     // band groups of 512 bins (1 unless k > 512)
  int nedge;      // bins 512, 1024, ... < k: left to the edge kernels when groups > 1
  bool fs;        // four-step path (k_fs_a / k_fs_f / k_fs_b): more than 512 bins at the tile counts of fs_tiles();
  int fs_nsplit, fs_lc;   // takes precedence over the band groups and over full8 (option "fourstep" = 0: off)
  bool conv1 = false;   // smx_conv_*: one launch per direction (k_conv1: n_fft <= 2048)
  int conv1_nj = 16;    // ... channel pairs per workgroup of that launch (16: 512 threads; 8: 256 threads)
  bool full8;     // N = 2048 with k > 512: the eight-band kernel (k_full8) takes every call that runs
                  // forward half and inverse half together; the band groups remain the plan of the
                  // phase-split backward (and define the workspace layout, which must not depend on
                  // which of the two a call uses)
};

int check_shape(int B, int N, int D, int F) {
  if (B <= 0 || N <= 0 || D <= 0 || F <= 0)
    return fail(SMX_ERR_INVALID, "shape must be positive: B=%d N=%d D=%d F=%d", B, N, D, F);
  if ((long long)N * D >= (1ll << 40)) return fail(SMX_ERR_INVALID, "N*D too large");
  return SMX_OK;
}

// General problem: x / y have R rows, the transform length is N >= R (rows >= R are zero padding /
// cropped), k bins are kept, k <= min(F, N/2 + 1) -- k = N/2 + 1 includes the Nyquist bin.
// The reference layer's own shapes are R = N, k = min(F, N/2) (layer_shape).
struct Shape { int B, R, D, F, N, k; };
Shape layer_shape(int B, int N, int D, int F) { return Shape{B, N, D, F, N, F < N / 2 ? F : N / 2}; }

int check_shape(const Shape& h) {
  if (int rc = check_shape(h.B, h.N, h.D, h.F)) return rc;
  if (h.R <= 0 || h.R > h.N)
    return fail(SMX_ERR_INVALID, "rows must be in [1, n_fft]: rows=%d n_fft=%d", h.R, h.N);
  if (h.k < 0 || h.k > h.F || h.k > h.N / 2 + 1)
    return fail(SMX_ERR_INVALID, "k must be in [0, min(F, n_fft/2 + 1)]: k=%d F=%d n_fft=%d", h.k, h.F, h.N);
  return SMX_OK;
}
int shape_from(const smx_shape* sh, Shape* out) {
  if (!sh) return fail(SMX_ERR_INVALID, "shape is NULL");
  *out = Shape{sh->B, sh->rows, sh->D, sh->F, sh->n_fft, sh->k};
  return check_shape(*out);
}

// tile counts the four-step path takes: the column transform in one thread's registers (5 ... 16; even 18 ... 32
// by one radix-2 step over two half-length transforms) or shared by L / 16 threads (64, 128, 256)
static bool fs_tiles(int L) {
  return (L >= 5 && L <= 32) || L == 64 || L == 128 || L == 256;      // (odd 17 ... 31: round 3)
}
// ... plus every L = L1 L2 the two-level
// columns take with a first-level length 9 ... 15: 36 ... 60 step 4, 72 ... 120 step 8, 144 ... 240 step 16
static bool fs_tiles_filter(int L) {
  int l1, l2;
  return fs_tiles(L) || (L >= 33 && L <= 256 && fs_two_level(L, &l1, &l2));
}

// residue (256-point plan) or tile (sixteen-row plan) chunks per (batch row, d-tile): L = items to cut
static int choose_nsplit(int forced, int nwg, int L, double bytes) {
  int ns = forced;
  if (ns <= 0) {
    // One fused launch per direction whenever the (b, d-tile) pairs alone fill the chip (2 WG/CU).
    // Otherwise the residues are cut into chunks (three more launches, ~30 us of fixed cost):
    //  * large tensors are bandwidth-bound: aim at ONE resident round of 512 workgroups (2 per CU) --
    //    with the XCD-aware placement C3 measured 59 % of the roofline at 8 chunks (512 workgroups)
    //    against 54 % at 16 and 48-51 % at 6, 10, 12 (partial second round);
    //  * small tensors are latency-bound (a workgroup walks 2L tiles at ~2.5 us each): split only if
    //    that walk is longer than the chunked walk plus the fixed cost, and keep one resident round.
    //    (measured: (32,2048,256) 46 us fused vs 65 us split; (2,4096,256) 66 vs 30.)
    ns = 1;
    if (nwg < 384) {
      if (bytes >= 128.0 * (1 << 20)) {
        ns = 512 / nwg;
        if (ns < 1) ns = 1;
      } else {
        int cand = 512 / nwg;
        if (cand > L) cand = L;
        if (cand > 1) {
          const int lc = (L + cand - 1) / cand;
          if (5 * L > 6 * lc + 30) ns = cand;
        }
      }
    }
  }
  return ns;
}

Plan make_plan(const Shape& h) {
  const int B = h.B, N = h.N, D = h.D;
  Plan p{};
  const smx_options opt = cur_opts();
  p.k = h.k;
  p.groups = 1;
  // (a batch row of 2 GiB or more does not fit the 32-bit offsets of the streaming kernels' buffer accesses, RowBuf)
  const bool rows32 = (unsigned long long)h.R * (unsigned long long)D * 4ull < (1ull << 31);
  const bool fast = !opt.force_direct && N % M == 0 && D % 2 == 0 && p.k >= 1 && rows32;
  if (!fast) {
    // N = 16 P, not a multiple of 256: sixteen-row decimation (k_fused16) for the layer-sized filters (k <= 256;
    // zero-padded rows included: functional.spectral_mix runs N = 8 (odd) as the even bins of 2 N); everything else -- and every call this plan's kernels do not serve (dropout,
    // phase-split backward, synthesis alone) -- runs the DFT products of the direct plan on the same workspace
    if (!opt.force_direct && opt.decim16 != 0 && N % 16 == 0 && N % M != 0 && D % 2 == 0 && p.k >= 1 && p.k <= 256 &&
        p.k <= N / 2) {        // (not the Nyquist bin: its two slots +-N/2 would not cancel to an exact +0 imaginary part)
      p.path = SMX_PATH_DECIM16;
      p.L = (N / 16 + 15) / 16;               // tiles of 16 residues
      p.nb = p.k > 128 ? 2 : 1;
      p.nwg = B * ((D + DT - 1) / DT);
      int ns = choose_nsplit(opt.nsplit, p.nwg, p.L, 4.0 * B * (double)h.R * D);
      if (ns > p.L) ns = p.L;
      p.lc = (p.L + ns - 1) / ns;
      p.nsplit = (p.L + p.lc - 1) / p.lc;
      return p;
    }
    p.path = SMX_PATH_DIRECT; p.nsplit = 1; return p;
  }
  p.path = SMX_PATH_DECIMATED;
  p.L = N / M;
  // bands by the bins below the Nyquist bin: with k = N/2 + 1 that bin (f = 128 L) is the self-paired
  // edge slot of the NB = L kernels (L in 1, 2, 4), an ordinary +-pair of a wider band set (L = 3), or
  // an edge / interior bin of the band groups
  const int kb = p.k > N / 2 ? N / 2 : p.k;
  p.nb = kb > 256 ? 4 : kb > 128 ? 2 : 1;
  p.nwg = B * ((D + DT - 1) / DT);
  if (kb > 512) {
    p.full8 = p.L == 8 && opt.full8 != 0;
    const int fsm = opt.fourstep;
    p.fs = fsm != 0 && fs_tiles_filter(p.L);
    if (p.fs) {
      p.full8 = false;
      int ns = 512 / p.nwg;                       // one resident round of tile workgroups, as on the split plan
      if (ns < 1) ns = 1;
      if (ns > p.L) ns = p.L;
      p.fs_lc = (p.L + ns - 1) / ns;
      p.fs_nsplit = (p.L + p.fs_lc - 1) / p.fs_lc;
    }
    // More than 512 bins: the four-band kernels run once per group of 512 bins (group g: |f| in
    // [512 g, 512 g + 512), its own residue-twiddle table, later groups add to y); the bins that are
    // multiples of 512 pair across groups and go through the literal-DFT kernels instead.  x is read
    // and y re-written once per group -- 12 B/sample more per extra group, against O(N k) per column
    // for the direct path.  One fused launch per group (no residue split).
    p.groups = (kb + 511) / 512;
    p.nedge = (p.k - 1) / 512;
    p.nsplit = 1; p.lc = p.L;
    return p;
  }
  int ns = choose_nsplit(opt.nsplit, p.nwg, p.L, 4.0 * B * (double)h.R * D);
  if (ns > p.L) ns = p.L;
  p.lc = (p.L + ns - 1) / ns;
  p.nsplit = (p.L + p.lc - 1) / p.lc;
  return p;
}

inline size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

struct Ws {
  size_t z = 0, zs = 0, s = 0, slab = 0, gbp = 0, spec0 = 0, spec1 = 0, spec2 = 0, lnp = 0, total = 0;
  size_t s_group = 0;            // bytes of one band group's parked spectrum
  size_t edge0 = 0, edge1 = 0;   // (B, nedge, D) complex: edge-bin spectrum / filtered edge bins
  size_t edgep = 0;              // per-chunk partial sums of launch_edge_spectrum
  size_t wt = 0;                 // (k, D) complex: the filter packed for the unpack phase
  size_t fs = 0;                 // four-step path: [B*ndt][L][16][256] complex tile spectra
  size_t gscp = 0;               // four-step path: partial sums of the row-scale gradient
  size_t sync = 0;               // SYNC_WORDS counters of the launches that fold the parameter gradients in
  size_t rows = 0;               // band-group plan: a (B, N, D) float copy of the input (masked g; LayerNorm(x) of the block)
};

// The first SYNC_BYTES of EVERY workspace layout are the sync area (flag words of the launches that fold the
// parameter-gradient reduction in): the same place whatever the shape, never used as scratch, so the "zero between
// calls" state survives a workspace that layers of different shapes share.
constexpr size_t SYNC_BYTES = 65536;

Ws ws_layout(const Plan& p, int B, int N, int D) {
  Ws w;
  size_t o = SYNC_BYTES;
  w.sync = 0;
  if (p.path == SMX_PATH_DECIM16) {
    w.slab = o; o += al((size_t)B * p.k * D * sizeof(cf));
    w.gbp = o; o += al((size_t)B * D * sizeof(float));
    w.wt = o; o += al((size_t)p.k * D * sizeof(cf));
    const size_t per = (size_t)16 * p.nb * TPB * sizeof(cf);
    w.s = o; o += al((size_t)p.nwg * per);           // filtered spectrum (split plan; parked by a phase-split backward)
    if (p.nsplit > 1) {
      w.z = o; o += al((size_t)p.nwg * p.nsplit * per);
      w.zs = o; o += al((size_t)p.nwg * per);
    }
  }
  if (p.path == SMX_PATH_DECIMATED) {
    const size_t per = (size_t)16 * p.nb * TPB * sizeof(cf);
    w.z = o; o += al((size_t)p.nwg * p.nsplit * per);
    w.zs = o; o += al((size_t)p.nwg * per);
    w.s_group = al((size_t)p.nwg * per);
    w.s = o; o += w.s_group * p.groups;
    if (p.groups > 1) {
      const size_t e = al((size_t)B * p.nedge * D * sizeof(cf));
      w.edge0 = o; o += e;
      w.edge1 = o; o += e;
      w.edgep = o;
      o += al((size_t)edge_chunks(B, N, D) * B * p.nedge * D * 2 * sizeof(double));
      if (!p.fs && !p.full8) {   // the group launches re-read their input while the output accumulates: no in-place form
        w.rows = o; o += al((size_t)B * N * D * sizeof(float));
      }
    }
    w.wt = o; o += al((size_t)p.k * D * sizeof(cf));
    if (p.fs) {
      w.fs = o; o += al((size_t)p.nwg * p.L * EX * sizeof(cf));
      w.gscp = o; o += al((size_t)p.nwg * fs_column_blocks(p.L) * 16 * sizeof(cf));
    }
    w.slab = o; o += al((size_t)B * p.k * D * sizeof(cf));
    w.gbp = o; o += al((size_t)B * D * sizeof(float));
  } else {
    const size_t spec = al((size_t)B * (p.k > 0 ? p.k : 1) * D * sizeof(cf));
    w.spec0 = o; o += spec;
    w.spec1 = o; o += spec;
    w.spec2 = o; o += spec;
  }
  if (ln_supported(D)) {       // grad_gamma / grad_beta partials of smx_block_backward
    w.lnp = o; o += al((size_t)ln_num_blocks((long long)B * N) * 2 * D * sizeof(float));
  }
  w.total = o;
  return w;
}

int need_ws(const Ws& w, void* ws, size_t bytes) {
  if (w.total == 0) return SMX_OK;
  if (!ws) return fail(SMX_ERR_WORKSPACE, "workspace is NULL, %zu bytes required", w.total);
  if (bytes < w.total)
    return fail(SMX_ERR_WORKSPACE, "workspace has %zu bytes, %zu required", bytes, w.total);
  if ((uintptr_t)ws & 255) return fail(SMX_ERR_INVALID, "workspace must be 256-byte aligned");
  return SMX_OK;
}

DecimArgs decim_args(const Plan& p, const Tables& t, const Shape& h, char* ws, const Ws& w) {
  const int B = h.B, N = h.N, D = h.D, F = h.F;
  DecimArgs a{};
  a.tw = t.tw; a.bt = t.bt; a.tq = t.tq;
  a.g.B = B; a.g.N = N; a.g.D = D; a.g.F = F; a.g.k = p.k; a.g.L = p.L; a.g.R = h.R;
  a.g.inv_n = (float)(1.0 / (double)N);
  if (p.path == SMX_PATH_DECIM16) { a.g.P = N / 16; a.v16 = t.v16; a.b16 = t.b16; }
  const smx_options opt = cur_opts();
  a.placement = opt.placement;
  a.round = opt.round;
  // Store policy of the streamed output (y / grad_x, B R D floats): rows written with the default write-back policy.
  // Measured at step level on five shapes (profiles/r04_store_policy.txt): L2 + Infinity Cache take about 64 MiB of
  // dirty lines per launch sequence at no cost -- those stores retire at cache speed and drain while the next
  // launch reads -- all-streaming stores leave that unused (C2 +9 % step time), all-cached ones overflow it (+13 %).
  {
    const double mib = 4.0 * B * (double)h.R * D / (1 << 20);
    // (the residue-split plan writes the whole tensor in ONE write-only launch: four rows up to 768 MiB -- C3 0.448 ms
    //  against 0.452 with two and 0.480 with none)
    const int aut = mib <= (p.nsplit > 1 ? 768 : 320) ? 4 : mib <= 768 ? 2 : mib <= 1536 ? 1 : 0;
    a.st_plain = opt.st_plain < 0 ? aut : (opt.st_plain >= 4 ? 4 : opt.st_plain == 3 ? 2 : opt.st_plain);
    a.g.st_plain = a.st_plain;            // (the pointer-addressed tile stores read it from the geometry)
  }
  a.nsplit = p.nsplit; a.lc = p.lc;
  a.sum_in_f = p.nb == 1 && p.nsplit <= 64;       // one band: k_split_f sums the chunk partials itself (no k_split_sum)
  a.ws_z = (cf*)(ws + w.z);
  a.ws_zs = (cf*)(ws + w.zs);
  a.ws_s = (cf*)(ws + w.s);
  return a;
}

// dropout parameters of one call: thr = round(p * 65536) (0 = off)
struct DropCfg { unsigned thr = 0; float scale = 1.f; const unsigned long long* rng = nullptr; };
int drop_cfg(float p, const void* rng_state, DropCfg* out) {
  if (!(p >= 0.f) || p >= 1.f) return fail(SMX_ERR_INVALID, "dropout p must be in [0, 1), got %g", (double)p);
  long thr = lroundf(p * 65536.f);
  if (thr > 65535) thr = 65535;
  if (thr <= 0) return SMX_OK;
  if (!rng_state) return fail(SMX_ERR_INVALID, "rng_state is NULL with dropout p > 0");
  if ((uintptr_t)rng_state & 7) return fail(SMX_ERR_INVALID, "rng_state must be 8-byte aligned");
  out->thr = (unsigned)thr;
  out->scale = 65536.f / (float)(65536 - thr);
  out->rng = (const unsigned long long*)rng_state;
  return SMX_OK;
}
void set_drop(DecimArgs& a, const DropCfg& dc) { a.drop_thr = dc.thr; a.drop_scale = dc.scale; a.rng = dc.rng; }

// The filter in the layout of the unpack phase (FilterArgs::wt).  `ready`: a copy packed by an earlier
// call with the same weights (backward reusing forward's) -- nothing is launched.  `keep`: caller buffer
// that receives the packed copy.  Otherwise it goes to the workspace, and without a workspace (allowed
// for the single-launch forward) the kernels gather from (D,F) directly.
int pack_filter(DecimArgs& a, const Plan& p, const Ws& w, void* workspace, size_t workspace_bytes,
                const float* w_re, const float* w_im, int D, int F, const float* ready, float* keep,
                hipStream_t s) {
  a.fa.wt = nullptr;
  if (((uintptr_t)ready | (uintptr_t)keep) & 15)
    return fail(SMX_ERR_INVALID, "filter_pack must be 16-byte aligned");
  // Small problems are launch-latency-bound: the extra 5 us launch costs more than the gathers it saves
  // ((8,512,256): 22 -> 17 us per forward).  The rule depends on the shape only, so a forward / backward
  // pair always agrees on whether filter_pack holds anything.
  if ((double)a.g.B * a.g.R * a.g.D < 8.0 * (1 << 20)) return SMX_OK;
  // One band: every workgroup (k_fused<1, .>, k_split_f<1, .>) stages its own slice of (D,F) through LDS
  // (prefetch_w / stage_w in smx_core.h); no packed copy is read or written on either plan.
  if (p.nb == 1 && p.groups == 1) return SMX_OK;
  if (ready) { a.fa.wt = ready; return SMX_OK; }
  cf* wt = (cf*)keep;
  if (!wt) {
    if (!workspace || workspace_bytes < w.total || ((uintptr_t)workspace & 255)) return SMX_OK;
    wt = (cf*)((char*)workspace + w.wt);
  }
  HIP_TRY(launch_pack_w(w_re, w_im, wt, D, F, p.k, s));
  a.fa.wt = (const float*)wt;
  return SMX_OK;
}

// ---- k > 512: band groups (see make_plan) --------------------------------------------------------
// Point `a` at band group g: its residue-twiddle table, bin offset, parked-spectrum slot; only the
// first group stores (the others add) and carries the bias.
int set_group(DecimArgs& a, const Plan& p, const Tables& t, const Ws& w, char* ws, int N, int g,
              const float* bias, hipStream_t s) {
  a.bt = t.bt;
  if (g > 0) if (int rc = get_group_bt(N, g, const_cast<cf**>(&a.bt), s)) return rc;
  a.fa.goff = 512 * g;
  a.fa.multi = p.groups > 1;
  a.fa.bias = g == 0 ? bias : nullptr;
  a.accumulate = g > 0;
  a.ws_s = (cf*)(ws + w.s + (size_t)g * w.s_group);
  return SMX_OK;
}

// the bins 512, 1024, ... < k of a multi-group plan, through the literal-DFT kernels
DirectArgs edge_args(const Plan& p, const Tables& t, const Shape& h) {
  DirectArgs e{h.B, h.N, h.D, h.F, p.nedge, t.tw};
  e.f0 = 512; e.fstep = 512;
  e.R = h.R;
  return e;
}
DirectArgs direct_args(const Plan& p, const Tables& t, const Shape& h) {
  DirectArgs d{h.B, h.N, h.D, h.F, p.k, t.tw};
  d.R = h.R;
  return d;
}

}  // namespace

extern "C" {

int smx_version(void) { return SMX_VERSION; }

__global__ void k_diag_clock(unsigned long long* out2, int spin) {
  const unsigned long long c0 = clock64(), r0 = wall_clock64();       // s_memtime / s_memrealtime
  float a = (float)threadIdx.x, b = 1.0000001f;
  for (int i = 0; i < spin; ++i) a = __builtin_fmaf(a, b, 1e-7f);
  const unsigned long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { out2[0] = c1 - c0; out2[1] = r1 - r0; }
  if (a == 12345.678f) out2[1] = 0;                                    // keeps the chain alive
}
int smx_diag_clock(unsigned long long* out2, int spin, void* stream) {
  if (!out2 || spin <= 0) return fail(SMX_ERR_INVALID, "out2 must be non-NULL and spin positive");
  hipLaunchKernelGGL(k_diag_clock, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, spin);
  HIP_TRY(hipGetLastError());
  return SMX_OK;
}
const char* smx_last_error(void) { return t_err.c_str(); }

int smx_set_option(const char* name, int value) {
  if (!name) return fail(SMX_ERR_INVALID, "option name is NULL");
  std::atomic<int>* o = nullptr;
  if (!strcmp(name, "nsplit")) o = &o_nsplit;
  else if (!strcmp(name, "placement") || !strcmp(name, "stagger")) o = &o_placement;
  else if (!strcmp(name, "round")) o = &o_round;
  else if (!strcmp(name, "force_direct")) o = &o_force_direct;
  else if (!strcmp(name, "full8")) o = &o_full8;
  else if (!strcmp(name, "fourstep")) o = &o_fourstep;
  else if (!strcmp(name, "fs_bgroups")) o = &o_fs_bgroups;
  else if (!strcmp(name, "fold_gradw")) o = &o_fold_gradw;
  else if (!strcmp(name, "decim16")) o = &o_decim16;
  else if (!strcmp(name, "conv1")) o = &o_conv1;
  else if (!strcmp(name, "st_plain")) o = &o_st_plain;
  else if (!strcmp(name, "tiled_dft")) { set_tiled_dft(value); g_opt_epoch++; return SMX_OK; }
  else if (!strcmp(name, "table_cache_entries")) { o_table_cap = value < 1 ? 1 : value; return SMX_OK; }
  else return fail(SMX_ERR_INVALID, "unknown option '%s'", name);
  *o = value;
  g_opt_epoch++;
  return SMX_OK;
}

int smx_options_default(smx_options* out) {
  if (!out) return fail(SMX_ERR_INVALID, "out is NULL");
  *out = default_opts();
  return SMX_OK;
}
int smx_options_push(const smx_options* opts) {
  if (!opts) return fail(SMX_ERR_INVALID, "opts is NULL");
  if (t_opt_depth >= OPT_DEPTH) return fail(SMX_ERR_INVALID, "smx_options_push nested deeper than %d", OPT_DEPTH);
  t_opt_stack[t_opt_depth++] = *opts;
  return SMX_OK;
}
int smx_options_pop(void) {
  if (t_opt_depth <= 0) return fail(SMX_ERR_INVALID, "smx_options_pop without a matching push");
  --t_opt_depth;
  return SMX_OK;
}
unsigned long long smx_options_epoch(void) { return g_opt_epoch.load(); }
unsigned long long smx_tables_epoch(void) { return g_tab_epoch.load(); }

// Compile-time switches this binary was built with that change what the kernels compute or how they are tuned.
// "SMX_AB_*" are timing ablations (tools/ab.sh) that return WRONG RESULTS by design: the Python loader refuses
// such a library (tensor_cuda_fft_amd._lib.load).
const char* smx_build_flags(void) {
  return ""
#ifdef SMX_AB_NO_FWD
         " SMX_AB_NO_FWD"
#endif
#ifdef SMX_AB_NO_UNPACK
         " SMX_AB_NO_UNPACK"
#endif
#ifdef SMX_AB_NO_INV
         " SMX_AB_NO_INV"
#endif
#if !SMX_NT_LOAD
         " SMX_NT_LOAD=0"
#endif
#if !SMX_NT_STORE
         " SMX_NT_STORE=0"
#endif
#ifdef SMX_BUILD_NOTE
         " " SMX_BUILD_NOTE
#endif
      ;
}

static int plan_query_impl(const Shape& h, smx_plan* out) {
  if (!out) return fail(SMX_ERR_INVALID, "out is NULL");
  Plan p = make_plan(h);
  out->path = p.path; out->k = p.k; out->L = p.L; out->bands = p.full8 ? 8 : p.fs ? 0 : p.nb;
  out->nsplit = p.fs ? p.fs_nsplit : p.nsplit;
  out->workgroups = p.path != SMX_PATH_DIRECT ? p.nwg * out->nsplit : 0;
  out->groups = (p.full8 || p.fs) ? 1 : p.groups;
  return SMX_OK;
}
int smx_plan_query(int B, int N, int D, int F, smx_plan* out) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  return plan_query_impl(layer_shape(B, N, D, F), out);
}
int smx_plan_query_ex(const smx_shape* shape, smx_plan* out) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  return plan_query_impl(h, out);
}

int smx_workspace_bytes(int B, int N, int D, int F, size_t* out) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  if (!out) return fail(SMX_ERR_INVALID, "out is NULL");
  const Shape h = layer_shape(B, N, D, F);
  *out = ws_layout(make_plan(h), B, N, D).total;
  return SMX_OK;
}
int smx_workspace_bytes_ex(const smx_shape* shape, size_t* out) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  if (!out) return fail(SMX_ERR_INVALID, "out is NULL");
  *out = ws_layout(make_plan(h), h.B, h.N, h.D).total;
  return SMX_OK;
}

int smx_prepare(int N) {
  if (N <= 0) return fail(SMX_ERR_INVALID, "N must be positive");
  TableRef t;
  return get_tables(N, &t, nullptr);
}

int smx_forward(const float* x, const float* w_re, const float* w_im, const float* bias, float* y,
                float* xk_save, void* workspace, size_t workspace_bytes, int B, int N, int D, int F,
                int conj_w, void* stream) {
  return smx_forward_dropout(x, w_re, w_im, bias, y, xk_save, workspace, workspace_bytes, B, N, D, F,
                             conj_w, 0.f, nullptr, nullptr, stream);
}

static int forward_impl(const Shape& h, const float* x, const float* w_re, const float* w_im,
                        const float* bias, float* y, float* xk_save, void* workspace,
                        size_t workspace_bytes, int conj_w, float dropout_p, const void* rng_state,
                        float* filter_pack, void* stream, const float* row_scale = nullptr);

int smx_forward_dropout(const float* x, const float* w_re, const float* w_im, const float* bias,
                        float* y, float* xk_save, void* workspace, size_t workspace_bytes, int B,
                        int N, int D, int F, int conj_w, float dropout_p, const void* rng_state,
                        float* filter_pack, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  return forward_impl(layer_shape(B, N, D, F), x, w_re, w_im, bias, y, xk_save, workspace, workspace_bytes,
                      conj_w, dropout_p, rng_state, filter_pack, stream);
}

int smx_forward_ex(const smx_shape* shape, const float* x, const float* w_re, const float* w_im,
                   const float* bias, float* y, float* xk_save, void* workspace, size_t workspace_bytes,
                   int conj_w, float* filter_pack, const float* row_scale, void* stream) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  return forward_impl(h, x, w_re, w_im, bias, y, xk_save, workspace, workspace_bytes, conj_w, 0.f, nullptr,
                      filter_pack, stream, row_scale);
}

int smx_row_scale_supported(const smx_shape* shape) {
  Shape h;
  if (shape_from(shape, &h)) return 0;
  const Plan p = make_plan(h);
  // (the eight-band kernel of option fourstep = 0 is left out: it only serves calls that run both halves of the
  // backward together, and a caller must be able to rely on this answer for every phase split)
  return p.path == SMX_PATH_DECIMATED && (p.groups == 1 || p.fs) ? 1 : 0;
}

static int forward_impl(const Shape& h, const float* x, const float* w_re, const float* w_im,
                        const float* bias, float* y, float* xk_save, void* workspace,
                        size_t workspace_bytes, int conj_w, float dropout_p, const void* rng_state,
                        float* filter_pack, void* stream, const float* row_scale) {
  const int B = h.B, N = h.N, D = h.D, F = h.F;
  DropCfg dc;
  if (int rc = drop_cfg(dropout_p, rng_state, &dc)) return rc;
  const bool pack_ready = (conj_w & SMX_FILTER_PACK_READY) != 0;      // filter_pack already holds W
  conj_w &= 1;
  if (pack_ready && !filter_pack) return fail(SMX_ERR_INVALID, "SMX_FILTER_PACK_READY without filter_pack");
  if (!x || !w_re || !w_im || !y) return fail(SMX_ERR_INVALID, "x, w_re, w_im, y must be non-NULL");
  if (((uintptr_t)x | (uintptr_t)y) & 7) return fail(SMX_ERR_INVALID, "x and y must be 8-byte aligned");
  if ((uintptr_t)xk_save & 15) return fail(SMX_ERR_INVALID, "xk_save must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const Plan p = make_plan(h);
  if (dc.thr && h.R < N) return fail(SMX_ERR_UNSUPPORTED, "fused dropout is not available with zero-padded rows");
  const Ws w = ws_layout(p, B, N, D);
  TableRef t;
  if (int rc = get_tables(N, &t, s)) return rc;
  char* ws = (char*)workspace;
  if (p.path == SMX_PATH_DECIM16 && !row_scale) {
    DecimArgs a = decim_args(p, t, h, ws, w);
    a.in = x; a.out = y;
    a.fa.w_re = w_re; a.fa.w_im = w_im; a.fa.bias = bias; a.fa.conj_w = conj_w;
    a.fa.xk_out = xk_save;
    set_drop(a, dc);
    if (int rc = pack_filter(a, p, w, workspace, workspace_bytes, w_re, w_im, D, F,
                             pack_ready ? filter_pack : nullptr, pack_ready ? nullptr : filter_pack, s))
      return rc;
    if (p.nsplit == 1) {
      HIP_TRY(launch_fused16(a, p.nb, 0, s));
    } else {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      HIP_TRY(launch_split16_a(a, p.nb, false, s));
      HIP_TRY(launch_split_f(a, p.nb, 0, s));
      HIP_TRY(launch_split16_b(a, p.nb, true, s));
    }
    return SMX_OK;
  }
  if (p.path == SMX_PATH_DECIMATED) {
    if (p.nsplit > 1) if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
    DecimArgs a = decim_args(p, t, h, ws, w);
    a.in = x; a.out = y;
    a.fa.w_re = w_re; a.fa.w_im = w_im; a.fa.bias = bias; a.fa.conj_w = conj_w;
    a.fa.xk_out = xk_save;
    a.fa.sc = row_scale;
    if (row_scale && !(p.groups == 1 || p.fs))
      return fail(SMX_ERR_UNSUPPORTED, "row_scale is not available on the band-group plan (smx_row_scale_supported)");
    // More than 512 bins (four-step, eight-band and band-group plans): their tile stores have no fused mask; the same mask -- a pure
    // function of (generator state, batch row, element) -- goes on y in one more pass (round 4; refused before).
    const bool post_drop = dc.thr && (p.fs || p.full8 || p.groups > 1);
    set_drop(a, post_drop ? DropCfg{} : dc);
    if (int rc = pack_filter(a, p, w, workspace, workspace_bytes, w_re, w_im, D, F,
                             pack_ready ? filter_pack : nullptr, pack_ready ? nullptr : filter_pack, s))
      return rc;
    if (p.fs) {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      a.ws_f = (cf*)(ws + w.fs); a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
      HIP_TRY(launch_fs_a(a, s));
      HIP_TRY(launch_fs_f(a, 0, s));
      HIP_TRY(launch_fs_b(a, s));
      if (post_drop) HIP_TRY(launch_dropout_rows(y, y, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
      return SMX_OK;
    }
    if (p.full8) {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      HIP_TRY(launch_full8(a, 0, s));
      if (post_drop) HIP_TRY(launch_dropout_rows(y, y, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
      return SMX_OK;
    }
    if (p.groups > 1) {
      if (x == y) return fail(SMX_ERR_INVALID, "the band-group plan cannot transform in place (y must not alias x)");
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      for (int g = 0; g < p.groups; ++g) {
        if (int rc = set_group(a, p, t, w, ws, N, g, bias, s)) return rc;
        HIP_TRY(launch_fused(a, 4, 0, s));
      }
      DirectArgs e = edge_args(p, t, h);
      cf* xe = xk_save ? (cf*)xk_save : (cf*)(ws + w.edge0);
      e.rows = xk_save ? p.k : 0;
      HIP_TRY(launch_edge_spectrum(x, xe, (double*)(ws + w.edgep), e, s));
      HIP_TRY(launch_direct_filter(xe, w_re, w_im, conj_w, (cf*)(ws + w.edge1), e, s));
      e.rows = 0;
      HIP_TRY(launch_edge_synth_acc((cf*)(ws + w.edge1), y, e, s));
      if (post_drop) HIP_TRY(launch_dropout_rows(y, y, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
      return SMX_OK;
    }
    if (p.nsplit == 1) {
      // the forward launch leaves the workspace's sync area zero: a backward call on the same workspace may then
      // skip its own clearing (SMX_PHASE_SYNC_CLEAN)
      if (workspace && workspace_bytes >= w.total && !((uintptr_t)workspace & 255) &&
          sync_words(B, D) * sizeof(unsigned) <= SYNC_BYTES)
        a.sync = (unsigned*)(ws + w.sync);
      HIP_TRY(launch_fused(a, p.nb, 0, s));
    } else {
      HIP_TRY(launch_split_a(a, p.nb, false, s));
      HIP_TRY(launch_split_f(a, p.nb, 0, s));
      HIP_TRY(launch_split_b(a, p.nb, true, s));
    }
    return SMX_OK;
  }
  if (row_scale) return fail(SMX_ERR_UNSUPPORTED, "row_scale is not available on the direct plan (smx_row_scale_supported)");
  if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
  DirectArgs d = direct_args(p, t, h);
  cf* xk = xk_save ? (cf*)xk_save : (cf*)(ws + w.spec0);
  cf* sk = (cf*)(ws + w.spec1);
  HIP_TRY(launch_direct_spectrum(x, xk, d, s));
  HIP_TRY(launch_direct_filter(xk, w_re, w_im, conj_w, sk, d, s));
  HIP_TRY(launch_direct_synth(sk, bias, y, d, s));
  if (dc.thr) HIP_TRY(launch_dropout_rows(y, y, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
  return SMX_OK;
}

int smx_backward(const float* g, const float* xk, const float* w_re, const float* w_im,
                 float* grad_x, float* gw_re, float* gw_im, float* gbias, void* workspace,
                 size_t workspace_bytes, int B, int N, int D, int F, int phases, void* stream) {
  return smx_backward_dropout(g, xk, w_re, w_im, grad_x, gw_re, gw_im, gbias, workspace,
                              workspace_bytes, B, N, D, F, phases, 0.f, nullptr, nullptr, stream);
}

static int backward_impl(const Shape& h, const float* g, const float* xk, const float* w_re,
                         const float* w_im, float* grad_x, float* gw_re, float* gw_im, float* gbias,
                         void* workspace, size_t workspace_bytes, int phases, float dropout_p,
                         const void* rng_state, const float* filter_pack, void* stream,
                         const float* row_scale = nullptr, float* grad_row_scale = nullptr);

int smx_backward_dropout(const float* g, const float* xk, const float* w_re, const float* w_im,
                         float* grad_x, float* gw_re, float* gw_im, float* gbias, void* workspace,
                         size_t workspace_bytes, int B, int N, int D, int F, int phases,
                         float dropout_p, const void* rng_state, const float* filter_pack,
                         void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  return backward_impl(layer_shape(B, N, D, F), g, xk, w_re, w_im, grad_x, gw_re, gw_im, gbias, workspace,
                       workspace_bytes, phases, dropout_p, rng_state, filter_pack, stream);
}

int smx_backward_ex(const smx_shape* shape, const float* g, const float* xk, const float* w_re,
                    const float* w_im, float* grad_x, float* gw_re, float* gw_im, float* gbias,
                    void* workspace, size_t workspace_bytes, int phases, const float* filter_pack,
                    const float* row_scale, float* grad_row_scale, void* stream) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  return backward_impl(h, g, xk, w_re, w_im, grad_x, gw_re, gw_im, gbias, workspace, workspace_bytes, phases,
                       0.f, nullptr, filter_pack, stream, row_scale, grad_row_scale);
}

static int backward_impl(const Shape& h, const float* g, const float* xk, const float* w_re,
                         const float* w_im, float* grad_x, float* gw_re, float* gw_im, float* gbias,
                         void* workspace, size_t workspace_bytes, int phases, float dropout_p,
                         const void* rng_state, const float* filter_pack, void* stream,
                         const float* row_scale, float* grad_row_scale) {
  const int B = h.B, N = h.N, D = h.D, F = h.F;
  DropCfg dc;
  if (int rc = drop_cfg(dropout_p, rng_state, &dc)) return rc;
  if (!g || !w_re || !w_im) return fail(SMX_ERR_INVALID, "g, w_re, w_im must be non-NULL");
  const bool sync_clean = (phases & SMX_PHASE_SYNC_CLEAN) != 0;
  phases &= ~SMX_PHASE_SYNC_CLEAN;
  if (phases < 1 || phases > 7) return fail(SMX_ERR_INVALID, "phases must be a combination of 1, 2, 4");
  if ((phases & SMX_PHASE_INVERSE) && !grad_x) return fail(SMX_ERR_INVALID, "grad_x is NULL");
  const bool want_w = gw_re || gw_im || gbias;
  if (want_w && !(gw_re && gw_im && gbias))
    return fail(SMX_ERR_INVALID, "gw_re, gw_im, gbias must be given together");
  if (((uintptr_t)g | (uintptr_t)grad_x) & 7)
    return fail(SMX_ERR_INVALID, "g and grad_x must be 8-byte aligned");
  if ((uintptr_t)xk & 15) return fail(SMX_ERR_INVALID, "xk must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const Plan p = make_plan(h);
  if (dc.thr && h.R < N) return fail(SMX_ERR_UNSUPPORTED, "fused dropout is not available with zero-padded rows");
  if ((want_w || dc.thr || grad_row_scale) && !xk && p.k > 0)
    return fail(SMX_ERR_INVALID, "xk (saved spectrum) is NULL");
  if ((row_scale || grad_row_scale) &&
      !(p.path == SMX_PATH_DECIMATED && (p.groups == 1 || p.fs)))
    return fail(SMX_ERR_UNSUPPORTED, "row_scale is not available on this plan (smx_row_scale_supported)");
  if (grad_row_scale && !row_scale) return fail(SMX_ERR_INVALID, "grad_row_scale without row_scale");
  const Ws w = ws_layout(p, B, N, D);
  if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
  TableRef t;
  if (int rc = get_tables(N, &t, s)) return rc;
  char* ws = (char*)workspace;
  const bool do_spec = phases & SMX_PHASE_SPECTRUM, do_inv = phases & SMX_PHASE_INVERSE;
  const bool do_par = (phases & SMX_PHASE_PARAMS) && want_w;

  if (p.path == SMX_PATH_DECIM16) {
    if (row_scale || grad_row_scale)
      return fail(SMX_ERR_UNSUPPORTED, "row_scale is not available on this plan (smx_row_scale_supported)");
    DecimArgs a = decim_args(p, t, h, ws, w);
    a.in = g; a.out = grad_x;
    a.fa.w_re = w_re; a.fa.w_im = w_im; a.fa.conj_w = 1;
    a.fa.xk_in = xk; a.fa.pslab = (float*)(ws + w.slab); a.fa.gb_part = (float*)(ws + w.gbp);
    set_drop(a, dc);
    const int mode = (want_w || dc.thr) ? 1 : 0;                  // (the mask is applied by the mode-1 load)
    if (do_spec) {
      if (int rc = pack_filter(a, p, w, workspace, workspace_bytes, w_re, w_im, D, F, filter_pack, nullptr, s)) return rc;
      if (p.nsplit > 1) {
        HIP_TRY(launch_split16_a(a, p.nb, true, s));
        HIP_TRY(launch_split_f(a, p.nb, mode, s));
        if (do_inv) HIP_TRY(launch_split16_b(a, p.nb, false, s));
      } else if (do_inv) {
        HIP_TRY(launch_fused16(a, p.nb, mode, s));
      } else {                                                     // SPECTRUM alone: products out, spectrum parked
        DecimArgs sp = a;
        sp.out = nullptr;
        HIP_TRY(launch_fused16(sp, p.nb, mode, s));
      }
    } else if (do_inv) {
      HIP_TRY(launch_split16_b(a, p.nb, false, s));               // (nsplit == 1: one chunk = every tile)
    }
    if (do_par)
      HIP_TRY(launch_gradw_slab((cf*)(ws + w.slab), (float*)(ws + w.gbp), gw_re, gw_im, gbias, B, D, F, p.k, s));
    return SMX_OK;
  }
  if (p.path == SMX_PATH_DECIMATED) {
    DecimArgs a = decim_args(p, t, h, ws, w);
    a.in = g; a.out = grad_x;
    a.fa.w_re = w_re; a.fa.w_im = w_im; a.fa.conj_w = 1;
    a.fa.xk_in = xk; a.fa.pslab = (float*)(ws + w.slab); a.fa.gb_part = (float*)(ws + w.gbp);
    // mode 0 with xk_out == NULL: input gradient only (with dropout the mask goes on the LOADED tile,
    // which only the mode-1 instantiation does: use it, the products land in the workspace unused)
    const bool pre_drop = dc.thr && (p.fs || p.full8 || p.groups > 1);      // (see below)
    const int mode = (want_w || (dc.thr && !pre_drop) || grad_row_scale) ? 1 : 0;
    a.fa.sc = row_scale; a.fa.gsc = grad_row_scale;
    if (p.fs && grad_row_scale) a.fa.gsc_part = (cf*)(ws + w.gscp);
    // four-step / eight-band plans: the masked upstream gradient is staged in grad_x first (their kernels read the whole
    // input column before the first store to it, so transforming grad_x in place is safe)
    set_drop(a, pre_drop ? DropCfg{} : dc);
    if (pre_drop && do_spec) {
      if (p.fs || p.full8) {
        if (!grad_x) return fail(SMX_ERR_INVALID, "dropout with more than 512 bins needs grad_x as scratch");
        if (p.full8 && !do_inv) return fail(SMX_ERR_UNSUPPORTED, "dropout on the eight-band plan needs the spectrum and inverse phases in one call");
        HIP_TRY(launch_dropout_rows(g, grad_x, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
        a.in = grad_x;
      } else {             // band groups: the launches re-read g while grad_x accumulates -> the workspace's row copy
        HIP_TRY(launch_dropout_rows(g, (float*)(ws + w.rows), B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
        a.in = (const float*)(ws + w.rows);
        g = a.in;          // (the edge-bin spectrum below reads g as well)
      }
    }
    if (do_spec)
      if (int rc = pack_filter(a, p, w, workspace, workspace_bytes, w_re, w_im, D, F, filter_pack,
                               nullptr, s))
        return rc;
    if (p.fs) {
      a.ws_f = (cf*)(ws + w.fs); a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
      // Option "fs_bgroups" = G > 0: slab rows summed over G batch groups inside k_fs_f (a rule of the shape
      // and the option only, so that a separate SMX_PHASE_PARAMS call reads the layout the SPECTRUM call
      // wrote).  Measured at (64,1024,512), G = 8: k_gradw 71 -> 16 us, but k_fs_f<8,1> 215 -> 286 us -- one
      // thread walking 8 batch rows exposes the load latency 9216 independent workgroups hide.  Off by default.
      const int bgo = cur_opts().fs_bgroups;
      const int bg = (bgo > 0 && p.L >= 5 && p.L <= 16 && B >= 2 * bgo) ? bgo : 0;
      a.fs_bgroups = bg;
      if (do_spec) {
        HIP_TRY(launch_fs_a(a, s));
        HIP_TRY(launch_fs_f(a, mode, s));
      }
      if (do_inv) HIP_TRY(launch_fs_b(a, s));
      if (do_par)
        HIP_TRY(launch_gradw_slab((cf*)(ws + w.slab), (float*)(ws + w.gbp), gw_re, gw_im, gbias,
                                  bg ? (B + (B + bg - 1) / bg - 1) / ((B + bg - 1) / bg) : B, D, F, p.k, s));
      return SMX_OK;
    }
    if (p.full8 && do_spec && do_inv) {
      HIP_TRY(launch_full8(a, mode, s));
      if (do_par)
        HIP_TRY(launch_gradw_slab((cf*)(ws + w.slab), (float*)(ws + w.gbp), gw_re, gw_im, gbias, B,
                                  D, F, p.k, s));
      return SMX_OK;
    }
    if (p.groups > 1) {
      DirectArgs e = edge_args(p, t, h);
      cf* ge = (cf*)(ws + w.edge0);
      cf* se = (cf*)(ws + w.edge1);
      if (do_spec) {
        HIP_TRY(launch_edge_spectrum(g, ge, (double*)(ws + w.edgep), e, s));
        HIP_TRY(launch_direct_filter(ge, w_re, w_im, 1, se, e, s));
        if (want_w) HIP_TRY(launch_edge_slab((const cf*)xk, ge, (cf*)(ws + w.slab), p.k, e, s));
      }
      for (int gi = 0; gi < p.groups; ++gi) {
        if (int rc = set_group(a, p, t, w, ws, N, gi, nullptr, s)) return rc;
        if (do_spec && do_inv) {
          HIP_TRY(launch_fused(a, 4, mode, s));
        } else if (do_spec) {
          DecimArgs h = a;
          h.out = nullptr;
          HIP_TRY(launch_fused(h, 4, mode, s));
        } else if (do_inv) {
          HIP_TRY(launch_split_b(a, 4, false, s));
        }
      }
      if (do_inv) HIP_TRY(launch_edge_synth_acc(se, grad_x, e, s));
      if (do_par)
        HIP_TRY(launch_gradw_slab((cf*)(ws + w.slab), (float*)(ws + w.gbp), gw_re, gw_im, gbias, B,
                                  D, F, p.k, s));
      return SMX_OK;
    }
    if (do_spec && do_inv && p.nsplit == 1) {
      // Option fold_gradw = 1: one launch for everything -- the parameter-gradient reduction rides behind the
      // transform workgroups (gradw_tail in smx_decim.hip) instead of a separate k_gradw launch + two kernel
      // boundaries.  Bit-identical, wait-free for the producers, and MEASURED SLOWER (profiles/r03_fold_gradw_ab.txt:
      // C2 backward 111.0 -> 115.8 us, C5 290 -> 304 us with the flag words already clean, 4 us more when the
      // library clears them): the reduction workgroups inherit the transform kernel's footprint (two per CU) and
      // can only start when transform workgroups retire, so the tail is longer than k_gradw's 7.5 us + boundaries.
      // Off by default.
      const bool fold = do_par && mode == 1 && p.nb <= 2 && cur_opts().fold_gradw != 0 &&
                        sync_words(B, D) * sizeof(unsigned) <= SYNC_BYTES;
      if (fold) {
        a.sync = (unsigned*)(ws + w.sync);
        a.n_cons = gradw_tail_blocks(D, F, true);
        a.gw_re = gw_re; a.gw_im = gw_im; a.gbias = gbias;
        a.fa.slab_agent = 1;
        if (!sync_clean) HIP_TRY(hipMemsetAsync(a.sync, 0, sync_words(B, D) * sizeof(unsigned), s));
      }
      HIP_TRY(launch_fused(a, p.nb, mode, s));
      if (fold) return SMX_OK;
    } else {
      if (do_spec) {
        if (p.nsplit == 1) {          // forward half + filter in one launch, S parked in the workspace
          DecimArgs h = a;
          h.out = nullptr;
          HIP_TRY(launch_fused(h, p.nb, mode, s));
        } else {
          HIP_TRY(launch_split_a(a, p.nb, true, s));
          HIP_TRY(launch_split_f(a, p.nb, mode, s));
        }
      }
      if (do_inv) HIP_TRY(launch_split_b(a, p.nb, false, s));
    }
    if (do_par)
      HIP_TRY(launch_gradw_slab((cf*)(ws + w.slab), (float*)(ws + w.gbp), gw_re, gw_im, gbias, B,
                                D, F, p.k, s));
    return SMX_OK;
  }

  // direct path: spec0 = G (k' = max(k,1) bins so grad_bias is available when k == 0), spec1 = S
  DirectArgs d = direct_args(p, t, h);
  cf* gk = (cf*)(ws + w.spec0);
  cf* sk = (cf*)(ws + w.spec1);
  if (do_spec) {
    DirectArgs dg = d;
    if (p.k == 0) dg.k = 1;
    if (dc.thr) {            // masked upstream gradient, staged in grad_x (overwritten by the synthesis)
      if (!grad_x) return fail(SMX_ERR_INVALID, "dropout on the direct plan needs grad_x as scratch");
      HIP_TRY(launch_dropout_rows(g, grad_x, B, (long long)N * D, dc.thr, dc.scale, dc.rng, s));
      g = grad_x;
    }
    HIP_TRY(launch_direct_spectrum(g, gk, dg, s));
    HIP_TRY(launch_direct_filter(gk, w_re, w_im, 1, sk, d, s));
  }
  if (do_par) {
    if (p.k == 0) {
      HIP_TRY(hipMemsetAsync(gw_re, 0, (size_t)D * F * sizeof(float), s));
      HIP_TRY(hipMemsetAsync(gw_im, 0, (size_t)D * F * sizeof(float), s));
      // grad_bias = sum_b G[b,0,d].re : reuse the reduction with one bin and a dummy X
      // (its weight-gradient outputs go to spec2, which nothing else uses)
      float* scratch = (float*)(ws + w.spec2);
      HIP_TRY(launch_gradw_spectra(gk, gk, scratch, scratch + D, gbias, B, N, D, 1, 1, s));
    } else {
      HIP_TRY(launch_gradw_spectra((const cf*)xk, gk, gw_re, gw_im, gbias, B, N, D, F, p.k, s));
    }
  }
  if (do_inv) HIP_TRY(launch_direct_synth(sk, nullptr, grad_x, d, s));
  return SMX_OK;
}

// complex sequence FFT: (A) tile spectra + (F) columns written straight out; its own plan (any L the column
// kernel is instantiated for, also below the 512-bin threshold of the filter plans)
static bool cfft_plan(const Shape& h, Plan* p) {
  if (h.N % M != 0 || h.D % 2 != 0 || h.R > h.N) return false;
  const int L = h.N / M;
  if (!(L == 2 || L == 4 || fs_tiles_filter(L))) return false;
  *p = Plan{};
  p->path = SMX_PATH_DECIMATED; p->L = L; p->k = h.N / 2 + 1; p->nb = 4; p->groups = 1;
  p->nwg = h.B * ((h.D + DT - 1) / DT);
  int ns = 512 / p->nwg;
  if (ns < 1) ns = 1;
  if (ns > L) ns = L;
  p->fs = true;
  p->fs_lc = (L + ns - 1) / ns;
  p->fs_nsplit = (L + p->fs_lc - 1) / p->fs_lc;
  p->nsplit = 1; p->lc = L;
  return true;
}
int smx_cfft_workspace_bytes(const smx_shape* shape, size_t* out) {
  if (!shape || !out) return fail(SMX_ERR_INVALID, "NULL argument");
  Shape h{shape->B, shape->rows, shape->D, shape->n_fft / 2 + 1, shape->n_fft, shape->n_fft / 2 + 1};
  if (int rc = check_shape(h)) return rc;
  Plan p;
  if (!cfft_plan(h, &p)) return fail(SMX_ERR_UNSUPPORTED, "smx_cfft_ex does not take this shape");
  *out = SYNC_BYTES + al((size_t)p.nwg * p.L * EX * sizeof(cf));      // the sync area stays untouched (ws_layout)
  return SMX_OK;
}
int smx_cfft_ex(const smx_shape* shape, const float* z, float* out, void* workspace, size_t workspace_bytes,
                void* stream) {
  if (!shape) return fail(SMX_ERR_INVALID, "shape is NULL");
  Shape h{shape->B, shape->rows, shape->D, shape->n_fft / 2 + 1, shape->n_fft, shape->n_fft / 2 + 1};
  if (int rc = check_shape(h)) return rc;
  if (!z || !out) return fail(SMX_ERR_INVALID, "z and out must be non-NULL");
  if (((uintptr_t)z | (uintptr_t)out) & 7) return fail(SMX_ERR_INVALID, "z and out must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  Plan p;
  if (!cfft_plan(h, &p))
    return fail(SMX_ERR_UNSUPPORTED, "smx_cfft_ex needs n_fft = 256 L with L in {2, 4, 5..32, 36..64 step 4, 72..128 step 8, 144..256 step 16} and an even D; "
                                     "compose it from smx_spectrum_ex otherwise");
  const size_t need = SYNC_BYTES + al((size_t)p.nwg * p.L * EX * sizeof(cf));
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))
    return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes (smx_cfft_workspace_bytes)", need);
  TableRef t;
  if (int rc = get_tables(h.N, &t, s)) return rc;
  Ws dummy;
  DecimArgs a = decim_args(p, t, h, (char*)workspace, dummy);
  a.ws_z = a.ws_zs = a.ws_s = nullptr;
  a.in = z; a.out = nullptr;
  a.fa.xk_out = out;
  a.ws_f = (cf*)((char*)workspace + SYNC_BYTES); a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
  HIP_TRY(launch_fs_a(a, s));
  HIP_TRY(launch_fs_f(a, 3, s));
  return SMX_OK;
}

// ---- rank-one filter: the causal FFT convolution of fft_lm on the four-step path ------------------------
namespace {
struct ConvWs { size_t fs = 0, pp = 0, rp = 0, total = 0, save = 0; };
static bool conv_plan(const Shape& h, Plan* p) {
  // its own plan: tile spectra / columns / inverse tiles for n_fft = 512 ... 65536 (four columns of L values each
  // fit one thread's registers in backward up to L = 16; above that L / 16 threads share a column pair)
  if (h.N % M != 0 || h.D % 2 != 0 || h.R > h.N) return false;
  const int L = h.N / M;
  if (!(L == 2 || L == 4 || L == 8 || L == 16 || L == 32 || L == 64 || L == 128 || L == 256)) return false;
  *p = Plan{};
  p->path = SMX_PATH_DECIMATED; p->L = L; p->k = h.N / 2 + 1; p->nb = 4; p->groups = 1;
  p->nwg = h.B * ((h.D + DT - 1) / DT);
  int ns = 512 / p->nwg;
  if (ns < 1) ns = 1;
  if (ns > L) ns = L;
  p->fs = true;
  p->fs_lc = (L + ns - 1) / ns;
  p->fs_nsplit = (L + p->fs_lc - 1) / p->fs_lc;
  p->nsplit = 1; p->lc = L;
  // One launch per direction where the zero padding allows it (k_conv1).  Its workgroup owns 32 channels (512 threads,
  // one per CU) or 16 (256 threads, two per CU, 64-byte row segments); measured, hipGraph fwd+bwd at n_fft 2048
  // (profiles/r03_f2_conv1_ab.txt), by the count w of (batch row, 32-channel tile) items:
  //   w = 16: 0.061 ms three launches / 0.078 / 0.069    w = 64: 0.089 / 0.093 / 0.084     w = 128: 0.116 / 0.098 / 0.090
  //   w = 256: 0.182 / 0.114 / 0.108                     w = 512:   -   / 0.186 / 0.181    w = 1024: 0.663 / 0.350 / 0.367
  // so: below 48 items the three launches (they cut the residues into chunks to fill the chip), up to 768 the
  // 16-channel workgroups, above that the 32-channel ones.
  // Option "conv1": 1 = that rule (default), 0 = never, 2 / 3 = wherever the shape allows it with 32- / 16-channel
  // workgroups.
  const int c1 = cur_opts().conv1;
  p->conv1 = c1 != 0 && conv1_supported(h.N, h.R) && (c1 >= 2 || p->nwg >= 48);
  p->conv1_nj = c1 == 3 ? 8 : c1 == 2 ? 16 : (p->nwg <= 768 ? 8 : 16);
  return true;
}
static ConvWs conv_ws(const Plan& p, const Shape& h) {
  ConvWs w;
  size_t o = SYNC_BYTES;     // as every layout: the first 64 KiB are the flag words of the folded reductions, never
                             // scratch -- a layer backward that trusts them (SMX_PHASE_SYNC_CLEAN) may share this buffer
  w.save = (size_t)p.nwg * p.L * EX * sizeof(cf);      // (k_conv1: [workgroup][16 L / 2][512], the same bytes;
  if (p.conv1 && p.conv1_nj == 8)                       //  on 16-channel workgroups a ragged last tile rounds up)
    w.save = (size_t)conv1_workgroups(h.B, h.D, 8) * (p.L / 2 * 16) * 256 * sizeof(cf);
  w.fs = o; o += p.conv1 ? 0 : al(w.save);              // tile spectra in flight: the three-launch form only
  const size_t nwg = p.conv1 ? (size_t)conv1_workgroups(h.B, h.D, p.conv1_nj) : (size_t)p.nwg;
  w.pp = o; o += al((nwg + 32) * h.N * sizeof(cf));                 // partials (+ 32 spare rows: the staging area of the former two-stage sum)
  w.rp = o; o += al((size_t)p.nwg * conv_column_blocks(p.L) * 16 * sizeof(cf));
  w.total = o;
  return w;
}
int conv_shape(const smx_shape* sh, Shape* h) {
  if (!sh) return fail(SMX_ERR_INVALID, "shape is NULL");
  *h = Shape{sh->B, sh->rows, sh->D, sh->n_fft / 2 + 1, sh->n_fft, sh->n_fft / 2 + 1};
  return check_shape(*h);
}
}  // namespace

int smx_conv_supported(const smx_shape* shape) {
  Shape h; Plan p;
  if (conv_shape(shape, &h)) return 0;
  return conv_plan(h, &p) ? 1 : 0;
}
int smx_conv_workspace_bytes(const smx_shape* shape, size_t* workspace_bytes, size_t* save_bytes) {
  Shape h; Plan p;
  if (int rc = conv_shape(shape, &h)) return rc;
  if (!conv_plan(h, &p)) return fail(SMX_ERR_UNSUPPORTED, "smx_conv_* needs n_fft = 512 ... 65536 (a power of two), rows <= n_fft and an even channel count");
  const ConvWs w = conv_ws(p, h);
  if (workspace_bytes) *workspace_bytes = w.total;
  if (save_bytes) *save_bytes = w.save;
  return SMX_OK;
}

// t: the caller's TableRef -- the tables stay pinned until the entry point has enqueued its launches
static int conv_args(const Shape& h, const Plan& p, const ConvWs& w, void* workspace, size_t workspace_bytes,
                     const float* h_re, const float* h_im, const float* row_scale, hipStream_t s, TableRef& t,
                     DecimArgs* out) {
  if (!workspace || workspace_bytes < w.total || ((uintptr_t)workspace & 255))
    return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes", w.total);
  if (!h_re || !h_im) return fail(SMX_ERR_INVALID, "h_re, h_im must be non-NULL");
  if (int rc = get_tables(h.N, &t, s)) return rc;
  Ws dummy;
  DecimArgs a = decim_args(p, t, h, (char*)workspace, dummy);
  a.ws_z = a.ws_zs = a.ws_s = nullptr;
  a.ws_f = (cf*)((char*)workspace + w.fs);
  a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
  a.ca.h_re = h_re; a.ca.h_im = h_im; a.ca.sc = row_scale;
  a.ca.p_part = (cf*)((char*)workspace + w.pp);
  a.ca.r_part = (cf*)((char*)workspace + w.rp);
  a.out_scale = row_scale;
  *out = a;
  return SMX_OK;
}

int smx_conv_forward(const smx_shape* shape, const float* x, const float* h_re, const float* h_im,
                     const float* row_scale, float* y, float* x_spectra, void* workspace,
                     size_t workspace_bytes, void* stream) {
  Shape h; Plan p;
  if (int rc = conv_shape(shape, &h)) return rc;
  if (!conv_plan(h, &p)) return fail(SMX_ERR_UNSUPPORTED, "smx_conv_* needs n_fft = 512 ... 65536 (a power of two), rows <= n_fft and an even channel count");
  if (!x || !y) return fail(SMX_ERR_INVALID, "x and y must be non-NULL");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)x_spectra) & 7) return fail(SMX_ERR_INVALID, "x, y, x_spectra must be 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const ConvWs w = conv_ws(p, h);
  DecimArgs a;
  TableRef t;
  if (int rc = conv_args(h, p, w, workspace, workspace_bytes, h_re, h_im, row_scale, s, t, &a)) return rc;
  a.in = x; a.out = y;
  if (p.conv1) {
    a.ws_f = (cf*)x_spectra;                        // packed spectrum of x for backward, or NULL (inference)
    HIP_TRY(launch_conv1(a, p.conv1_nj, 0, nullptr, nullptr, nullptr, s));
    return SMX_OK;
  }
  cf* filtered = a.ws_f;
  if (x_spectra) a.ws_f = (cf*)x_spectra;           // (A) writes the tile spectra of x where backward finds them
  HIP_TRY(launch_fs_a(a, s));
  a.conv_src = a.ws_f;
  a.ws_f = filtered;
  HIP_TRY(launch_fs_conv(a, 0, nullptr, nullptr, nullptr, s));
  HIP_TRY(launch_fs_b(a, s));
  return SMX_OK;
}

int smx_conv_backward(const smx_shape* shape, const float* g, const float* x_spectra, const float* h_re,
                      const float* h_im, const float* row_scale, float* grad_x, float* grad_h_re,
                      float* grad_h_im, float* grad_row_scale, void* workspace, size_t workspace_bytes,
                      void* stream) {
  Shape h; Plan p;
  if (int rc = conv_shape(shape, &h)) return rc;
  if (!conv_plan(h, &p)) return fail(SMX_ERR_UNSUPPORTED, "smx_conv_* needs n_fft = 512 ... 65536 (a power of two), rows <= n_fft and an even channel count");
  if (!g || !x_spectra || !grad_x) return fail(SMX_ERR_INVALID, "g, x_spectra, grad_x must be non-NULL");
  if (((uintptr_t)g | (uintptr_t)grad_x | (uintptr_t)x_spectra) & 7)
    return fail(SMX_ERR_INVALID, "g, grad_x, x_spectra must be 8-byte aligned");
  if ((grad_h_re == nullptr) != (grad_h_im == nullptr))
    return fail(SMX_ERR_INVALID, "grad_h_re and grad_h_im must be given together");
  hipStream_t s = (hipStream_t)stream;
  const ConvWs w = conv_ws(p, h);
  DecimArgs a;
  TableRef t;
  if (int rc = conv_args(h, p, w, workspace, workspace_bytes, h_re, h_im, row_scale, s, t, &a)) return rc;
  a.in = g; a.out = grad_x;
  a.ca.xs = (const cf*)x_spectra;
  if (p.conv1) {
    HIP_TRY(launch_conv1(a, p.conv1_nj, 1, grad_h_re, grad_h_im, grad_row_scale, s));
    return SMX_OK;
  }
  HIP_TRY(launch_fs_a(a, s));
  a.conv_src = a.ws_f;
  HIP_TRY(launch_fs_conv(a, 1, grad_h_re, grad_h_im, grad_row_scale, s));
  HIP_TRY(launch_fs_b(a, s));
  return SMX_OK;
}

int smx_phase_filter(const float* magnitude, const float* phase, int D, int k, int n_fft, float* w_re, float* w_im,
                     void* stream) {
  if (D <= 0 || k <= 0 || n_fft <= 0 || k > n_fft / 2 + 1) return fail(SMX_ERR_INVALID, "need D > 0, 0 < k <= n_fft / 2 + 1");
  if (!magnitude || !phase || !w_re || !w_im) return fail(SMX_ERR_INVALID, "magnitude, phase, w_re, w_im must be non-NULL");
  HIP_TRY(launch_phase_filter(magnitude, phase, D, k, n_fft, w_re, w_im, (hipStream_t)stream));
  return SMX_OK;
}
int smx_phase_filter_backward(const float* magnitude, const float* phase, const float* grad_w_re, const float* grad_w_im,
                              int D, int k, int n_fft, int row_pitch, float* grad_magnitude, float* grad_phase,
                              void* stream) {
  if (D <= 0 || k <= 0 || n_fft <= 0 || k > n_fft / 2 + 1 || row_pitch < k)
    return fail(SMX_ERR_INVALID, "need D > 0, 0 < k <= n_fft / 2 + 1, row_pitch >= k");
  if (!magnitude || !phase || !grad_w_re || !grad_w_im) return fail(SMX_ERR_INVALID, "magnitude, phase, grad_w_re, grad_w_im must be non-NULL");
  HIP_TRY(launch_phase_filter_bwd(magnitude, phase, grad_w_re, grad_w_im, D, k, n_fft, row_pitch, grad_magnitude,
                                  grad_phase, (hipStream_t)stream));
  return SMX_OK;
}

int smx_conv_response(int n_fft, int taps, const float* kernel, const float* gate_logits, const float* mask,
                      float* h_re, float* h_im, void* stream) {
  if (n_fft < 2 || taps < 1 || taps > n_fft) return fail(SMX_ERR_INVALID, "need 1 <= taps <= n_fft, n_fft >= 2");
  if (!kernel || !h_re || !h_im) return fail(SMX_ERR_INVALID, "kernel, h_re, h_im must be non-NULL");
  hipStream_t s = (hipStream_t)stream;
  TableRef t;
  if (int rc = get_tables(n_fft, &t, s)) return rc;
  HIP_TRY(launch_conv_response(kernel, gate_logits, mask, t.tw, n_fft, taps, h_re, h_im, s));
  return SMX_OK;
}
int smx_conv_response_backward(int n_fft, int taps, int n_logits, const float* kernel, const float* gate_logits,
                               const float* mask, const float* grad_h_re, const float* grad_h_im, float* grad_kernel,
                               float* grad_gate_logits, void* stream) {
  if (n_fft < 2 || taps < 1 || taps > n_fft) return fail(SMX_ERR_INVALID, "need 1 <= taps <= n_fft, n_fft >= 2");
  if (!kernel || !grad_h_re || !grad_h_im) return fail(SMX_ERR_INVALID, "kernel, grad_h_re, grad_h_im must be non-NULL");
  if (grad_gate_logits && (!gate_logits || n_logits < n_fft / 2 + 1))
    return fail(SMX_ERR_INVALID, "grad_gate_logits needs gate_logits with at least n_fft / 2 + 1 entries");
  hipStream_t s = (hipStream_t)stream;
  TableRef t;
  if (int rc = get_tables(n_fft, &t, s)) return rc;
  HIP_TRY(launch_conv_response_bwd(kernel, gate_logits, mask, t.tw, n_fft, taps, n_logits, grad_h_re, grad_h_im,
                                   grad_kernel, grad_gate_logits, s));
  return SMX_OK;
}

static int spectrum_impl(const Shape& h, const float* x, float* xk, void* workspace,
                         size_t workspace_bytes, void* stream);
int smx_spectrum(const float* x, float* xk, void* workspace, size_t workspace_bytes, int B, int N,
                 int D, int F, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  return spectrum_impl(layer_shape(B, N, D, F), x, xk, workspace, workspace_bytes, stream);
}
int smx_spectrum_ex(const smx_shape* shape, const float* x, float* xk, void* workspace,
                    size_t workspace_bytes, void* stream) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  return spectrum_impl(h, x, xk, workspace, workspace_bytes, stream);
}
static int spectrum_impl(const Shape& h, const float* x, float* xk, void* workspace,
                         size_t workspace_bytes, void* stream) {
  const int B = h.B, N = h.N, D = h.D;
  if (h.k == 0) return SMX_OK;
  if (!x || !xk) return fail(SMX_ERR_INVALID, "x and xk must be non-NULL");
  if ((uintptr_t)x & 7) return fail(SMX_ERR_INVALID, "x must be 8-byte aligned");
  if ((uintptr_t)xk & 15) return fail(SMX_ERR_INVALID, "xk must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const Plan p = make_plan(h);
  const Ws w = ws_layout(p, B, N, D);
  TableRef t;
  if (int rc = get_tables(N, &t, s)) return rc;
  if (p.path == SMX_PATH_DECIM16) {
    DecimArgs a = decim_args(p, t, h, (char*)workspace, w);
    a.in = x; a.out = nullptr;
    a.fa.xk_out = xk;
    a.ws_s = nullptr;                        // nothing is parked (the workspace may be absent altogether)
    if (p.nsplit == 1) {
      HIP_TRY(launch_fused16(a, p.nb, 2, s));
    } else {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      HIP_TRY(launch_split16_a(a, p.nb, false, s));
      HIP_TRY(launch_split_f(a, p.nb, 2, s));
    }
    return SMX_OK;
  }
  if (p.path == SMX_PATH_DECIMATED) {
    if (p.nsplit > 1) if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
    DecimArgs a = decim_args(p, t, h, (char*)workspace, w);
    a.in = x; a.out = nullptr;
    // mode 2: unpack only -- no weights are read, no S is produced
    a.fa.xk_out = xk;
    a.ws_s = nullptr;
    if (p.fs) {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      a.ws_f = (cf*)((char*)workspace + w.fs); a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
      HIP_TRY(launch_fs_a(a, s));
      HIP_TRY(launch_fs_f(a, 2, s));
      return SMX_OK;
    }
    if (p.full8) {
      HIP_TRY(launch_full8(a, 2, s));
      return SMX_OK;
    }
    if (p.groups > 1) {
      if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
      for (int g = 0; g < p.groups; ++g) {
        if (int rc = set_group(a, p, t, w, (char*)workspace, N, g, nullptr, s)) return rc;
        a.ws_s = nullptr;
        HIP_TRY(launch_fused(a, 4, 2, s));
      }
      DirectArgs e = edge_args(p, t, h);
      e.rows = p.k;
      HIP_TRY(launch_edge_spectrum(x, (cf*)xk, (double*)((char*)workspace + w.edgep), e, s));
      return SMX_OK;
    }
    if (p.nsplit == 1) HIP_TRY(launch_fused(a, p.nb, 2, s));
    else {
      HIP_TRY(launch_split_a(a, p.nb, false, s));
      HIP_TRY(launch_split_f(a, p.nb, 2, s));
    }
    return SMX_OK;
  }
  DirectArgs d = direct_args(p, t, h);
  HIP_TRY(launch_direct_spectrum(x, (cf*)xk, d, s));
  return SMX_OK;
}

// N = 2048 with more than 512 bins (the reference's default causal-convolution length): the two halves of the
// pair run as ONE launch each with the eight bands in registers -- the tensor and the spectrum cross HBM once
// (measured at (64,1024,512): rfft 153 us against 203 us on the four-step path, which round-trips a workspace).
// These two launches touch no workspace (none is required, whatever smx_workspace_bytes_ex says for the filter
// plans of the same shape); option full8 = 0 sends the pair down the four-step path like every other length.
static bool pair_full8(const Plan& p) {
  return p.path == SMX_PATH_DECIMATED && p.L == 8 && p.k > 512 && cur_opts().full8 != 0;
}

int smx_rfft_ex(const smx_shape* shape, const float* x, float* spec, float scale, int hermitian,
                void* workspace, size_t workspace_bytes, void* stream) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  const Plan p8 = make_plan(h);
  if (h.k > 0 && pair_full8(p8)) {
    if (!x || !spec) return fail(SMX_ERR_INVALID, "x and spec must be non-NULL");
    if (((uintptr_t)x & 7) || ((uintptr_t)spec & 15))
      return fail(SMX_ERR_INVALID, "x must be 8-byte and spec 16-byte aligned");
    TableRef t;
    if (int rc = get_tables(h.N, &t, (hipStream_t)stream)) return rc;
    DecimArgs a = decim_args(p8, t, h, (char*)workspace, ws_layout(p8, h.B, h.N, h.D));
    a.in = x; a.out = nullptr; a.fa.xk_out = spec; a.ws_s = nullptr;
    HIP_TRY(launch_full8(a, 2, (hipStream_t)stream));
  } else if (int rc = spectrum_impl(h, x, spec, workspace, workspace_bytes, stream)) return rc;
  if (h.k > 0 && (scale != 1.f || hermitian))
    HIP_TRY(launch_scale_bins((const cf*)spec, (cf*)spec, h.B, h.k, h.D, h.N, scale, hermitian,
                              (hipStream_t)stream));
  return SMX_OK;
}

int smx_irfft_ex(const smx_shape* shape, const float* spec, float* y, float scale, int hermitian,
                 void* workspace, size_t workspace_bytes, void* stream) {
  Shape h;
  if (int rc = shape_from(shape, &h)) return rc;
  const int B = h.B, N = h.N, D = h.D;
  if (!y) return fail(SMX_ERR_INVALID, "y must be non-NULL");
  hipStream_t s = (hipStream_t)stream;
  if (h.k == 0) {
    HIP_TRY(hipMemsetAsync(y, 0, (size_t)B * h.R * D * sizeof(float), s));
    return SMX_OK;
  }
  if (!spec) return fail(SMX_ERR_INVALID, "spec must be non-NULL");
  if ((uintptr_t)y & 7) return fail(SMX_ERR_INVALID, "y must be 8-byte aligned");
  if ((uintptr_t)spec & 15) return fail(SMX_ERR_INVALID, "spec must be 16-byte aligned");
  const Plan p = make_plan(h);
  const Ws w = ws_layout(p, B, N, D);
  TableRef t;
  if (int rc = get_tables(N, &t, s)) return rc;
  char* ws = (char*)workspace;
  if (p.path == SMX_PATH_DECIMATED && (p.fs || p.groups == 1 || pair_full8(p))) {
    const bool l8 = pair_full8(p);
    if (!l8 && (p.fs || p.nsplit > 1)) if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
    DecimArgs a = decim_args(p, t, h, ws, w);
    a.in = nullptr; a.out = y;
    a.fa.xk_in = spec; a.fa.sp_scale = scale; a.fa.sp_herm = hermitian;
    if (l8) {
      HIP_TRY(launch_synth8(a, s));
    } else if (p.fs) {
      a.ws_f = (cf*)(ws + w.fs); a.nsplit = p.fs_nsplit; a.lc = p.fs_lc;
      HIP_TRY(launch_fs_f(a, 4, s));
      HIP_TRY(launch_fs_b(a, s));
    } else if (p.nsplit == 1) {
      HIP_TRY(launch_synth(a, p.nb, s));
    } else {
      DecimArgs park = a;
      park.out = nullptr;
      HIP_TRY(launch_synth(park, p.nb, s));
      HIP_TRY(launch_split_b(a, p.nb, false, s));
    }
    return SMX_OK;
  }
  // DFT products: the direct plan, and the band-group plans (k > 512 at tile counts the four-step path does
  // not take) -- weighted copy of the spectrum in the workspace, then the synthesis kernels of the direct path
  if (int rc = need_ws(w, workspace, workspace_bytes)) return rc;
  cf* sk = (cf*)(ws + (p.path == SMX_PATH_DECIMATED ? w.slab : w.spec1));
  DirectArgs d{B, N, D, h.F, p.k, t.tw};
  d.R = h.R;
  HIP_TRY(launch_scale_bins((const cf*)spec, sk, B, p.k, D, N, scale, hermitian, s));
  HIP_TRY(launch_direct_synth(sk, nullptr, y, d, s));
  return SMX_OK;
}

int smx_grad_w(const float* xk, const float* gk, float* gw_re, float* gw_im, float* gbias, int B,
               int N, int D, int F, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  if (!gw_re || !gw_im) return fail(SMX_ERR_INVALID, "gw_re, gw_im must be non-NULL");
  const int k = F < N / 2 ? F : N / 2;
  if (k > 0 && (!xk || !gk)) return fail(SMX_ERR_INVALID, "xk, gk must be non-NULL");
  HIP_TRY(launch_gradw_spectra((const cf*)xk, (const cf*)gk, gw_re, gw_im, gbias, B, N, D, F, k,
                               (hipStream_t)stream));
  return SMX_OK;
}

int smx_wfilter_forward(const float* x_freq, const float* w_re, const float* w_im, float* out, int B,
                        int N, int D, int F, int conj_w, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  if (!x_freq || !w_re || !w_im || !out) return fail(SMX_ERR_INVALID, "NULL argument");
  const int k = F < N / 2 ? F : N / 2;
  HIP_TRY(launch_wfilter((const cf*)x_freq, w_re, w_im, conj_w, (cf*)out, B, N, D, F, k,
                         (hipStream_t)stream));
  return SMX_OK;
}

int smx_wfilter_grad_w(const float* x_freq, const float* g_freq, float* gw_re, float* gw_im, int B,
                       int N, int D, int F, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  if (!x_freq || !g_freq || !gw_re || !gw_im) return fail(SMX_ERR_INVALID, "NULL argument");
  const int k = F < N / 2 ? F : N / 2;
  HIP_TRY(launch_wfilter_gradw((const cf*)x_freq, (const cf*)g_freq, gw_re, gw_im, B, N, D, F, k,
                               (hipStream_t)stream));
  return SMX_OK;
}

int smx_cmul(const float* x, const float* w, float* out, long long batch, long long inner,
             int conj_w, void* stream) {
  if (batch < 0 || inner < 0) return fail(SMX_ERR_INVALID, "negative size");
  if (batch * inner == 0) return SMX_OK;
  if (!x || !w || !out) return fail(SMX_ERR_INVALID, "NULL argument");
  HIP_TRY(launch_cmul((const cf*)x, (const cf*)w, conj_w, (cf*)out, batch, inner,
                      (hipStream_t)stream));
  return SMX_OK;
}

int smx_cmul_grad_w(const float* x, const float* g, float* gw, long long batch, long long inner,
                    void* stream) {
  if (batch < 0 || inner < 0) return fail(SMX_ERR_INVALID, "negative size");
  if (inner == 0) return SMX_OK;
  if (!x || !g || !gw) return fail(SMX_ERR_INVALID, "NULL argument");
  HIP_TRY(launch_cmul_gradw((const cf*)x, (const cf*)g, (cf*)gw, batch, inner,
                            (hipStream_t)stream));
  return SMX_OK;
}

int smx_rng_next(void* state, void* saved, void* stream) {
  if (!state || !saved) return fail(SMX_ERR_INVALID, "state and saved must be non-NULL");
  if (((uintptr_t)state | (uintptr_t)saved) & 7) return fail(SMX_ERR_INVALID, "8-byte alignment required");
  HIP_TRY(launch_rng_next((unsigned long long*)state, (unsigned long long*)saved, (hipStream_t)stream));
  return SMX_OK;
}

int smx_block_supported(int D) { return ln_supported(D) ? 1 : 0; }

int smx_block_forward(const float* x, const float* ln_w, const float* ln_b, float eps,
                      const float* w_re, const float* w_im, const float* bias, float* y,
                      float* xk_save, float* ln_stats, void* workspace, size_t workspace_bytes,
                      int B, int N, int D, int F, void* stream) {
  return smx_block_forward_dropout(x, ln_w, ln_b, eps, w_re, w_im, bias, y, xk_save, ln_stats,
                                   workspace, workspace_bytes, B, N, D, F, 0.f, nullptr, nullptr,
                                   stream);
}

int smx_block_forward_dropout(const float* x, const float* ln_w, const float* ln_b, float eps,
                              const float* w_re, const float* w_im, const float* bias, float* y,
                              float* xk_save, float* ln_stats, void* workspace,
                              size_t workspace_bytes, int B, int N, int D, int F, float dropout_p,
                              const void* rng_state, float* filter_pack, void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  DropCfg dc;
  if (int rc = drop_cfg(dropout_p, rng_state, &dc)) return rc;
  if (!ln_supported(D)) return fail(SMX_ERR_UNSUPPORTED, "LayerNorm width D=%d is not supported", D);
  if (!x || !w_re || !w_im || !y || !ln_stats)
    return fail(SMX_ERR_INVALID, "x, w_re, w_im, y, ln_stats must be non-NULL");
  if (x == y) return fail(SMX_ERR_INVALID, "y must not alias x");
  if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ln_stats) & 7)
    return fail(SMX_ERR_INVALID, "x, y and ln_stats must be 8-byte aligned");
  if (D % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)ln_w | (uintptr_t)ln_b) & 15))
    return fail(SMX_ERR_INVALID, "x, y, ln_w, ln_b must be 16-byte aligned");
  if ((uintptr_t)xk_save & 15) return fail(SMX_ERR_INVALID, "xk_save must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long long rows = (long long)B * N;
  const Shape h = layer_shape(B, N, D, F);
  const Plan p = make_plan(h);
  const bool by_groups = p.groups > 1 && !p.fs && !p.full8;
  HIP_TRY(launch_ln_stats(x, (cf*)ln_stats, rows, D, eps, s));
  if (p.path == SMX_PATH_DECIMATED && p.nsplit == 1 && p.groups == 1 && !p.fs && !p.full8) {
    TableRef t;
    if (int rc = get_tables(N, &t, s)) return rc;
    const Ws w = ws_layout(p, B, N, D);
    DecimArgs a = decim_args(p, t, h, (char*)workspace, w);
    a.in = x; a.out = y;
    a.fa.w_re = w_re; a.fa.w_im = w_im; a.fa.bias = bias; a.fa.conj_w = 0;
    a.fa.xk_out = xk_save;
    a.ln_stats = (const cf*)ln_stats; a.ln_w = ln_w; a.ln_b = ln_b;
    set_drop(a, dc);
    if (int rc = pack_filter(a, p, w, workspace, workspace_bytes, w_re, w_im, D, F, nullptr, filter_pack,
                             s))
      return rc;
    if (workspace && workspace_bytes >= w.total && !((uintptr_t)workspace & 255) &&
        sync_words(B, D) * sizeof(unsigned) <= SYNC_BYTES)
      a.sync = (unsigned*)((char*)workspace + w.sync);        // left zero for smx_block_backward (SYNC_CLEAN)
    HIP_TRY(launch_fused_block(a, p.nb, s));
    return SMX_OK;
  }
  // other plans: normalise into y, transform y in place (every kernel reads its whole input column
  // before the first store to it), add x
  float* hbuf = y;
  if (by_groups) {          // no in-place form there: LayerNorm(x) goes to the workspace's row copy
    const Ws w = ws_layout(p, B, N, D);
    if (!workspace || workspace_bytes < w.total || ((uintptr_t)workspace & 255))
      return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes", w.total);
    hbuf = (float*)((char*)workspace + w.rows);
  }
  HIP_TRY(launch_ln_apply(x, (const cf*)ln_stats, ln_w, ln_b, hbuf, rows, D, s));
  if (int rc = smx_forward_dropout(hbuf, w_re, w_im, bias, y, xk_save, workspace, workspace_bytes, B, N, D,
                                   F, 0, dropout_p, rng_state, filter_pack, stream))
    return rc;
  HIP_TRY(launch_add_rows(y, x, (size_t)rows * D, s));
  return SMX_OK;
}

int smx_block_backward(const float* g, const float* x, const float* ln_stats, const float* ln_w,
                       const float* xk, const float* w_re, const float* w_im, float* grad_x,
                       float* g_ln_w, float* g_ln_b, float* gw_re, float* gw_im, float* gbias,
                       void* workspace, size_t workspace_bytes, int B, int N, int D, int F,
                       int phases, void* stream) {
  return smx_block_backward_dropout(g, x, ln_stats, ln_w, xk, w_re, w_im, grad_x, g_ln_w, g_ln_b, gw_re,
                                    gw_im, gbias, workspace, workspace_bytes, B, N, D, F, phases, 0.f,
                                    nullptr, nullptr, stream);
}

int smx_block_backward_dropout(const float* g, const float* x, const float* ln_stats,
                               const float* ln_w, const float* xk, const float* w_re,
                               const float* w_im, float* grad_x, float* g_ln_w, float* g_ln_b,
                               float* gw_re, float* gw_im, float* gbias, void* workspace,
                               size_t workspace_bytes, int B, int N, int D, int F, int phases,
                               float dropout_p, const void* rng_state, const float* filter_pack,
                               void* stream) {
  if (int rc = check_shape(B, N, D, F)) return rc;
  if (!ln_supported(D)) return fail(SMX_ERR_UNSUPPORTED, "LayerNorm width D=%d is not supported", D);
  if ((phases & 7) < 1 || phases > 15) return fail(SMX_ERR_INVALID, "phases must be a combination of 1, 2, 4");
  if ((phases & SMX_PHASE_INVERSE) && (!x || !ln_stats || !grad_x))
    return fail(SMX_ERR_INVALID, "x, ln_stats, grad_x must be non-NULL");
  if (g == grad_x || x == grad_x) return fail(SMX_ERR_INVALID, "grad_x must not alias g or x");
  if (D % 4 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)grad_x | (uintptr_t)ln_w) & 15))
    return fail(SMX_ERR_INVALID, "x, g, grad_x, ln_w must be 16-byte aligned");

  if (int rc = smx_backward_dropout(g, xk, w_re, w_im, grad_x, gw_re, gw_im, gbias, workspace,
                                    workspace_bytes, B, N, D, F, phases, dropout_p, rng_state,
                                    filter_pack, stream))
    return rc;
  if (phases & SMX_PHASE_INVERSE) {
    const Ws w = ws_layout(make_plan(layer_shape(B, N, D, F)), B, N, D);
    HIP_TRY(launch_ln_bwd(grad_x, x, g, (const cf*)ln_stats, ln_w, (float*)((char*)workspace + w.lnp),
                          g_ln_w, g_ln_b, (long long)B * N, D, (hipStream_t)stream));
  }
  return SMX_OK;
}

// ---- BicameralBlock's time path (smx_time.hip) ---------------------------------------------------------------------
static int dw_check(int B, int T, int C) {
  if (B <= 0 || T <= 0 || C <= 0) return fail(SMX_ERR_INVALID, "shape must be positive: B=%d T=%d C=%d", B, T, C);
  if (B > 65535) return fail(SMX_ERR_UNSUPPORTED, "smx_dwconv3_* takes at most 65535 batch rows");
  if ((unsigned long long)B * ((T + 31) / 32) * ((C + 255) / 256) >= (1ull << 31))
    return fail(SMX_ERR_UNSUPPORTED, "smx_dwconv3_*: tensor too large for one launch");
  return SMX_OK;
}
int smx_dwconv3_workspace_bytes(int B, int T, int C, size_t* out) {
  if (!out) return fail(SMX_ERR_INVALID, "out is NULL");
  if (int rc = dw_check(B, T, C)) return rc;
  *out = SYNC_BYTES + al(dwconv3_workspace_bytes(B, T, C));          // behind the sync area, as every layout (ws_layout)
  return SMX_OK;
}
int smx_dwconv3_forward(const float* x, const float* w, const float* bias, const float* scale, float* y, int B, int T,
                        int C, void* stream) {
  if (int rc = dw_check(B, T, C)) return rc;
  if (!x || !w || !y) return fail(SMX_ERR_INVALID, "x, w, y must be non-NULL");
  if (x == y) return fail(SMX_ERR_INVALID, "y must not alias x");
  HIP_TRY(launch_dwconv3_fwd(x, w, bias, scale, y, B, T, C, (hipStream_t)stream));
  return SMX_OK;
}
int smx_dwconv3_backward(const float* g, const float* x, const float* w, const float* bias, const float* scale,
                         float* grad_x, float* grad_w, float* grad_bias, float* grad_scale, void* workspace,
                         size_t workspace_bytes, int B, int T, int C, void* stream) {
  if (int rc = dw_check(B, T, C)) return rc;
  if (!g || !x || !w) return fail(SMX_ERR_INVALID, "g, x, w must be non-NULL");
  if (grad_x == g || grad_x == x) return fail(SMX_ERR_INVALID, "grad_x must not alias g or x");
  if (grad_scale && !scale) return fail(SMX_ERR_INVALID, "grad_scale without scale");
  const size_t need = SYNC_BYTES + al(dwconv3_workspace_bytes(B, T, C));
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))
    return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes (smx_dwconv3_workspace_bytes)", need);
  HIP_TRY(launch_dwconv3_bwd(g, x, w, bias, scale, grad_x, grad_w, grad_bias, grad_scale,
                             (float*)((char*)workspace + SYNC_BYTES), B, T, C, (hipStream_t)stream));
  return SMX_OK;
}

// ---- SpectralLayerNorm (smx_time.hip) ---------------------------------------------------------------------------------
int smx_spectral_ln_supported(int C) { return spectral_ln_supported(C) ? 1 : 0; }
static int sln_check(int B, int F, int C) {
  if (B <= 0 || F <= 0 || C <= 0) return fail(SMX_ERR_INVALID, "shape must be positive: B=%d F=%d C=%d", B, F, C);
  if (!spectral_ln_supported(C)) return fail(SMX_ERR_UNSUPPORTED, "smx_spectral_ln_* takes C <= 1024, got %d", C);
  if ((long long)B * F >= (1ll << 33)) return fail(SMX_ERR_UNSUPPORTED, "too many rows");
  return SMX_OK;
}
int smx_spectral_ln_forward(const float* z, const float* gamma, const float* beta, float eps, float* out, int planar,
                            int B, int F, int C, void* stream) {
  if (int rc = sln_check(B, F, C)) return rc;
  if (!z || !gamma || !beta || !out) return fail(SMX_ERR_INVALID, "z, gamma, beta, out must be non-NULL");
  if (((uintptr_t)z | (uintptr_t)out) & 7) return fail(SMX_ERR_INVALID, "z and out must be 8-byte aligned");
  HIP_TRY(launch_spectral_ln_fwd((const cf*)z, gamma, beta, eps, (cf*)out, planar, B, F, C, (hipStream_t)stream));
  return SMX_OK;
}
int smx_spectral_ln_backward(const float* g, const float* z, const float* gamma, const float* beta, float eps,
                             float* grad_z, float* grad_gamma, float* grad_beta, int planar, int B, int F, int C,
                             void* stream) {
  if (int rc = sln_check(B, F, C)) return rc;
  if (!g || !z || !gamma || !beta) return fail(SMX_ERR_INVALID, "g, z, gamma, beta must be non-NULL");
  if (((uintptr_t)g | (uintptr_t)z | (uintptr_t)grad_z) & 7) return fail(SMX_ERR_INVALID, "g, z, grad_z must be 8-byte aligned");
  HIP_TRY(launch_spectral_ln_bwd((const cf*)g, (const cf*)z, gamma, beta, eps, (cf*)grad_z, grad_gamma, grad_beta, planar,
                                 B, F, C, (hipStream_t)stream));
  return SMX_OK;
}
static int planar_check(int B, int F, int C) {
  if (B <= 0 || F <= 0 || C <= 0) return fail(SMX_ERR_INVALID, "shape must be positive: B=%d F=%d C=%d", B, F, C);
  return SMX_OK;
}
int smx_planar_cmul_forward(const float* h, const float* f_re, const float* f_im, float* out, int B, int F, int C,
                            void* stream) {
  if (int rc = planar_check(B, F, C)) return rc;
  if (!h || !f_re || !f_im || !out) return fail(SMX_ERR_INVALID, "h, f_re, f_im, out must be non-NULL");
  HIP_TRY(launch_pcmul_fwd(h, f_re, f_im, out, B, F, C, (hipStream_t)stream));
  return SMX_OK;
}
int smx_planar_cmul_backward(const float* g, const float* h, const float* f_re, const float* f_im, float* grad_h,
                             float* grad_f_re, float* grad_f_im, int B, int F, int C, void* stream) {
  if (int rc = planar_check(B, F, C)) return rc;
  if (!g || !h || !f_re || !f_im) return fail(SMX_ERR_INVALID, "g, h, f_re, f_im must be non-NULL");
  HIP_TRY(launch_pcmul_bwd(g, h, f_re, f_im, grad_h, grad_f_re, grad_f_im, B, F, C, (hipStream_t)stream));
  return SMX_OK;
}
int smx_planar_add(const float* a, const float* planar, float* y, long long n, void* stream) {
  if (n <= 0 || !planar || !y) return fail(SMX_ERR_INVALID, "n must be positive, planar and y non-NULL");
  if (((uintptr_t)a | (uintptr_t)y) & 7) return fail(SMX_ERR_INVALID, "a and y must be 8-byte aligned");
  HIP_TRY(launch_add_planar((const cf*)a, planar, (cf*)y, n, (hipStream_t)stream));
  return SMX_OK;
}
int smx_planar_split(const float* g, float* planar, long long n, void* stream) {
  if (n <= 0 || !g || !planar) return fail(SMX_ERR_INVALID, "n must be positive, g and planar non-NULL");
  if ((uintptr_t)g & 7) return fail(SMX_ERR_INVALID, "g must be 8-byte aligned");
  HIP_TRY(launch_to_planar((const cf*)g, planar, n, (hipStream_t)stream));
  return SMX_OK;
}

// ---- the gate chain of the twin blocks (smx_time.hip) -------------------------------------------------------------------
static int gate_check(int B, int F, int C) {
  if (B <= 0 || F <= 0 || C <= 0) return fail(SMX_ERR_INVALID, "shape must be positive: B=%d F=%d C=%d", B, F, C);
  if (C % 2) return fail(SMX_ERR_UNSUPPORTED, "smx_spectral_gate_* takes an even channel count, got %d", C);
  if (B > 65535) return fail(SMX_ERR_UNSUPPORTED, "smx_spectral_gate_* takes at most 65535 batch rows");
  return SMX_OK;
}
int smx_spectral_gate_workspace_bytes(int B, int F, int C, size_t* out) {
  if (int rc = gate_check(B, F, C)) return rc;
  if (!out) return fail(SMX_ERR_INVALID, "out must be non-NULL");
  *out = SYNC_BYTES + al(gate_workspace_bytes(B, F, C));            // behind the sync area, as every layout (ws_layout)
  return SMX_OK;
}
int smx_spectral_gate_forward(const float* x, const float* a, const float* u, const float* p, const float* q,
                              const float* m, float* y, int B, int F, int C, void* stream) {
  if (int rc = gate_check(B, F, C)) return rc;
  if (!x || !a || !y) return fail(SMX_ERR_INVALID, "x, a, y must be non-NULL");
  if (((uintptr_t)x | (uintptr_t)y) & 15) return fail(SMX_ERR_INVALID, "x and y must be 16-byte aligned");
  if ((uintptr_t)a & 7) return fail(SMX_ERR_INVALID, "a must be 8-byte aligned");
  HIP_TRY(launch_gate_fwd((const cf*)x, (const cf*)a, u, p, q, m, (cf*)y, B, F, C, (hipStream_t)stream));
  return SMX_OK;
}
int smx_spectral_gate_backward(const float* g, const float* x, const float* a, const float* u, const float* p,
                               const float* q, const float* m, float* grad_x, float* s1, float* rc, float* rp,
                               void* workspace, size_t workspace_bytes, int B, int F, int C, void* stream) {
  if (int r = gate_check(B, F, C)) return r;
  if (!g || !x || !a) return fail(SMX_ERR_INVALID, "g, x, a must be non-NULL");
  if (((uintptr_t)g | (uintptr_t)x | (uintptr_t)grad_x) & 15) return fail(SMX_ERR_INVALID, "g, x, grad_x must be 16-byte aligned");
  if (((uintptr_t)a | (uintptr_t)s1) & 7) return fail(SMX_ERR_INVALID, "a and s1 must be 8-byte aligned");
  if (grad_x == g || grad_x == x) return fail(SMX_ERR_INVALID, "grad_x must not alias g or x");
  const size_t need = SYNC_BYTES + al(gate_workspace_bytes(B, F, C));
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))
    return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes (smx_spectral_gate_workspace_bytes)", need);
  HIP_TRY(launch_gate_bwd((const cf*)g, (const cf*)x, (const cf*)a, u, p, q, m, (cf*)grad_x, (cf*)s1, rc, rp,
                          (cf*)((char*)workspace + SYNC_BYTES), B, F, C, (hipStream_t)stream));
  return SMX_OK;
}

// ---- BicameralBlock's fusion line (smx_time.hip) --------------------------------------------------------------------------
int smx_mix_workspace_bytes(size_t* out) {
  if (!out) return fail(SMX_ERR_INVALID, "out must be non-NULL");
  *out = SYNC_BYTES + al(mix_workspace_bytes());
  return SMX_OK;
}
static int mix_check(long long n, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4) {
  if (n <= 0 || n % 4) return fail(SMX_ERR_UNSUPPORTED, "smx_mix_* takes a positive element count that is a multiple of 4, got %lld", n);
  if (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3 | (uintptr_t)p4) & 15)
    return fail(SMX_ERR_INVALID, "tensors must be 16-byte aligned");
  return SMX_OK;
}
int smx_mix_forward(const float* r, const float* a, const float* b, const float* c, const float* w, float c3, float* out,
                    long long n, void* stream) {
  if (!r || !a || !b || !w || !out) return fail(SMX_ERR_INVALID, "r, a, b, w, out must be non-NULL");
  if (int rc = mix_check(n, r, a, b, c, out)) return rc;
  HIP_TRY(launch_mix_fwd(r, a, b, c, w, c3, out, n, (hipStream_t)stream));
  return SMX_OK;
}
int smx_mix_backward(const float* g, const float* a, const float* b, const float* w, float c3, float* grad_a, float* grad_b,
                     float* grad_c, float* grad_w, void* workspace, size_t workspace_bytes, long long n, void* stream) {
  if (!g || !a || !b || !w) return fail(SMX_ERR_INVALID, "g, a, b, w must be non-NULL");
  if (int rc = mix_check(n, g, a, b, grad_a, grad_b)) return rc;
  if ((uintptr_t)grad_c & 15) return fail(SMX_ERR_INVALID, "tensors must be 16-byte aligned");
  const size_t need = SYNC_BYTES + al(mix_workspace_bytes());
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))
    return fail(SMX_ERR_WORKSPACE, "workspace must be 256-byte aligned and hold %zu bytes (smx_mix_workspace_bytes)", need);
  HIP_TRY(launch_mix_bwd(g, a, b, w, c3, grad_a, grad_b, grad_c, grad_w, (float*)((char*)workspace + SYNC_BYTES), n,
                         (hipStream_t)stream));
  return SMX_OK;
}

}  // extern "C"
