// libdsgd_hip -- MI355X (gfx950 / CDNA4) engine behind include/dsgd.h.
//
// Hot path of zifeo/distributed-sgd re-designed for one MI355X per process:
//   * the CSR shard, w, g and dimSparsity stay resident in HBM (288 GB/GPU); w/g/ds are
//     3 x 189 KB and live in the XCD L2s, so the only HBM stream is the CSR itself
//     (8 B per non-zero + 12 B per row -- SURVEY.md 8(d));
//   * gradient kernels: whole row ranges stream the matrix split by column rank (hot tiles with weights and gradient
//     in LDS, a row-ordered cold stream); index-list batches run the mini-batch engine (dsgd_batch.hpp); every sum is
//     accumulated in fixed point (exact, order-independent) and rounded once;
//   * exact reduce + regularise + aggregate + update are one fused kernel over D+1 columns;
//   * the synchronous master's Vec.mean over workers is one ncclAllReduce on the same stream.
// Citations "ref:" are relative to /root/reference/src/main/scala/epfl/distributed/.
//
// This file is written for gfx950 only: wave64, no CUDA paths, no compatibility layers.

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/dsgd.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    if (e__ != hipSuccess) return fail(DSGD_EHIP, "%s: %s", #expr, hipGetErrorString(e__));   \
  } while (0)

#define DSGD_TRY(expr)        \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != DSGD_OK) return rc__; \
  } while (0)

// ------------------------------------------------------------------------------------------------
// RCCL, resolved lazily so that single-GPU users never map the (very large) library
// ------------------------------------------------------------------------------------------------
namespace rccl {
typedef struct ncclComm* comm_t;
typedef struct {
  char internal[DSGD_UNIQUE_ID_BYTES];
} unique_id_t;
enum { kFloat32 = 7, kInt64 = 4, kUint32 = 3, kSum = 0 };
static int (*GetUniqueId)(unique_id_t*) = nullptr;
static int (*CommInitRank)(comm_t*, int, unique_id_t, int) = nullptr;
static int (*CommDestroy)(comm_t) = nullptr;
static int (*CommAbort)(comm_t) = nullptr;   // optional: only used to give up a group that not every rank joined
static int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
static int (*GroupStart)() = nullptr;
static int (*GroupEnd)() = nullptr;
static const char* (*GetErrorString)(int) = nullptr;
static std::once_flag once;
static bool ok = false;

static void load() {
  void* h = nullptr;
#ifdef DSGD_TEST_COLLECTIVE_SEAM
  // TEST BUILDS ONLY (tests/rccl_stub/build_seam.py compiles this file a second time with -DDSGD_TEST_COLLECTIVE_SEAM into
  // tests/rccl_stub/libdsgd_hip_seam.so): DSGD_RCCL_LIB=<path> names the library that serves the collectives -- a
  // host-staged shim that lets several ranks share ONE device, which RCCL refuses (the world = 2 arithmetic on a one-GPU
  // box).  The product library (distributed-sgd_amd/lib/libdsgd_hip.so) is compiled without this block: no environment
  // variable can put anything in RCCL's place there.
  if (const char* forced = getenv("DSGD_RCCL_LIB")) {
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
  }
#endif
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);  // reuse a copy the process already has
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) return;
  GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
  CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
  CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
  CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
  AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
  GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
  GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
  GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
  ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && GroupStart && GroupEnd && GetErrorString;
}
static bool available() {
  std::call_once(once, load);
  return ok;
}
}  // namespace rccl

#define RCCL_TRY(expr)                                                                          \
  do {                                                                                          \
    int r__ = (expr);                                                                           \
    if (r__ != 0) return fail(DSGD_ERCCL, "%s: %s", #expr, rccl::GetErrorString(r__));          \
  } while (0)

#include "dsgd_kernels.hpp"
#include "dsgd_batch.hpp"
#include "dsgd_cs.hpp"
#include "dsgd_dense.hpp"
#include "dsgd_fstep.hpp"
#include "dsgd_tcol.hpp"
#include "dsgd_shuffle.hpp"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct dsgd_plan {
  int* d_idx = nullptr;
  std::vector<long long> offsets;  // n_steps * n_workers + 1
  WorkSeg* d_segs = nullptr;       // n_steps * n_workers
  long long n_steps = 0;
  int n_workers = 0;
  long long max_items = 0;  // largest single list
  long long max_step_rows = 0;  // largest step (all workers together)
  bool fits = false;            // every list fits the staged sub-batch of dsgd_plan_kernel (rows and work items)
  long long fits_rows = -1;     // ... of the data set with this many rows (the one loaded at plan creation)
  // virtual tiles (dsgd_vt_grad_kernel): the lists laid out over the split streams, built at the first run that needs them
  std::vector<int> h_idx;       // host copy of the lists (a plan drawn on the device fetches it only if a host builder asks: plan_host_idx)
  bool idx_trusted = false;     // the lists were drawn by the library inside the caller's row ranges: nothing to validate
  VtLane* d_vt_lanes = nullptr; // 64 descriptors per tile
  WorkSeg* d_vt_segs = nullptr; // tile range of every list, then (n_lists further entries) its range of d_vt_long
  MbRec* d_vt_long = nullptr;   // rows of the lists that sit in no tile (long-row list, more than 64 cold entries)
  uint4* d_vt_packed = nullptr; // plans up to vt_pack_mb: the tiles' own copy of the rows they touch (4 KiB per tile)
  std::vector<long long> vt_off;       // n_lists + 1 tile offsets
  std::vector<int> vt_grid, vt_gxt, vt_shift;  // per step: workgroups per worker, those of them that walk tiles, shift
  long long vt_layout = -1;     // the layout generation the tiles were built for (-1: not built)
  bool vt_ok = false;           // every row of every list sits in the tiled streams
  // column slices (dsgd_cs_step_kernel): the lists laid out per (slice, step), built at the first run that can use them
  CsHdr* d_cs_hdr = nullptr;
  unsigned int* d_cs_meta = nullptr;
  unsigned short* d_cs_rf = nullptr;
  uint4* d_cs_col = nullptr;
  float4* d_cs_val = nullptr;
  unsigned short* d_cs_cl = nullptr;
  int cs_G = 0, cs_spl = 0, cs_nt = 0, cs_slot_stride = 0, cs_row_stride = 0, cs_cl_stride = 0;
  std::vector<int> cs_shift;    // per step
  long long cs_layout = -1;     // the layout generation (column ranking) the slices were built for
  bool cs_ok = false;
  bool cs_device_built = false; // laid out by dsgd_cs_layout_kernel (false: by the host, DSGD_CS_HOST_LAYOUT=1)
  size_t cs_bytes[6] = {0, 0, 0, 0, 0, 0};   // sizes of the six layout arrays (hdr, meta, rf, col, val, cl) as taken from the cache
  size_t idx_bytes = 0, segs_bytes = 0;
  // what the plan's set-up enqueued on the context's BUILD stream (lists uploaded, cells laid out) ends with this event;
  // the first run makes the launch stream wait for it
  hipEvent_t built_ev = nullptr;
  bool built_pending = false;
  // dsgd_plan_record: gate decisions and regulariser scalars of the steps run (column-slice plans)
  unsigned int* d_gate_rec = nullptr;
  float* d_s_rec = nullptr;
  int gate_words = 0;
  size_t gate_bytes = 0, s_bytes = 0;
};

struct FusedArgs {
  int hg, n_wg, hc, nc, n_wgc;
  double inv_scale, inv_scale_cold;
};

struct dsgd_ctx {
  dsgd_config cfg{};
  int dp = 0;  // D + 1
  std::mutex mu;
  hipStream_t stream = nullptr;
  // data
  long long n_rows = 0, nnz = 0;
  long long* d_row_ptr = nullptr;
  int* d_col = nullptr;
  float* d_val = nullptr;
  signed char* d_label = nullptr;
  int group = 16;  // lanes per row of the row-wise prediction / evaluation kernels, from the mean row length
  // column layout: external keys <-> internal frequency ranks (identity until prepare_layout)
  int* d_perm = nullptr;   // key  -> rank
  bool layout_ready = false;
  int hw_eval = DSGD_LDS_FLOATS;
  long long stream_min = 131072;  // row ranges with at least this many rows use the streaming kernels (DSGD_STREAM_MIN).  Four
                                  //   launches of the split streams cost ~50 us whatever the range; the row-wise kernel + reduce
                                  //   take 37 us for 18,519 rows and 57 us for 80,000 (streaming: 49 / 63 us;
                                  //   profiles/r04_stream_min.txt)
  // nnz-streaming kernels (contiguous row ranges)
  StreamSeg* d_ssegs = nullptr;
  int ssegs_cap = 0;
  std::vector<StreamSeg> ssegs_last;
  // The matrix is SPLIT by column rank into a hot stream (rank < hsplit: wave tiles, weights and gradient in LDS, no
  // gathers) and a cold stream (col - hsplit, val, row) handled by two small kernels whose LDS holds the cold weights /
  // the cold gradient.
  std::vector<long long> h_row_ptr;     // host copy of the internal row_ptr (tile building at layout time)
  std::vector<long long> h_crow_ptr;    // cold ENTRIES before each row
  std::vector<long long> h_hrp, h_ctp;  // slot offsets of the hot / cold stream (virtual tiles are built from them)
  std::vector<unsigned short> h_ccol;   // host copy of the 16-bit cold ranks (a virtual tile's descriptor carries its cold rank)
  long long layout_gen = 0;             // bumped whenever the split streams are rebuilt
  bool layout_seen_by_build = false;    // the build stream is ordered behind the kernels that wrote the current split streams
  bool cs_enable = true;                // DSGD_CS=0: small steps of resident plans through the row-parallel kernels
  int cs_g = 0;                         // DSGD_CS_G: slices (8 or 16; 0 = 8 up to four hosted workers, 16 beyond)
  long long cs_max_mb = 8192;           // DSGD_CS_MAX_MB: largest column-slice layout of one plan (device-built; the host
                                        //   builder of DSGD_CS_HOST_LAYOUT=1 holds a copy of it in host memory as well)
  int cs_nt = 0;                        // DSGD_CS_NT=256: tuning runs with 256 lanes per slice where a plan's steps fit them
  unsigned int cs_tag0 = 0;             // column-slice steps launched so far (the exchange granules' tags run on)
  float* d_cs_w = nullptr;              // the weights slice-major while cs_w_G != 0: then d_w is STALE -- every entry point that
  float* d_cs_ds = nullptr;             //   is not a column-slice launch converts back first (bind); dimSparsity likewise (a copy)
  int cs_w_G = 0;                       // slices of the slice-major state (0: the weights are in d_w, rank order)
  unsigned long long* d_cs_x = nullptr; // exchange buffer of dsgd_cs_step_kernel: [2][CS_MAX_G][CS_XSTRIDE] granules
  unsigned int* d_cs_sync = nullptr;    // its arrival counter and abort word
  bool cs_host_layout = false;          // DSGD_CS_HOST_LAYOUT=1: a plan's slices laid out by the host (rounds 1-4; kept as the cross-check)
  bool cs_req = false;                  // DSGD_CS_REQ=1: per-request steps of the reference's sizes through dsgd_cs_request_kernel.  OFF by
                                        //   default: a slice's workgroup lays the step out before it runs it, one dependent trip to
                                        //   memory per row and wave -- measured 126 us per 3 x 100 request against 40 us through the
                                        //   row-parallel kernels (profiles/r05_probe_v1.json); the kernel is kept, tested, and is what
                                        //   the abort-path test drives
  int cs_test_skip = 0;                 // (test builds: DSGD_TEST_CS_SKIP_PUBLISH -- slice 1 goes silent from this step of a launch on)
  unsigned int* d_cs_max = nullptr;     // the layout kernels' maxima and flags (4 words) ...
  unsigned int* h_cs_max = nullptr;     // ... and where pass 1's come back to (pinned)
  hipStream_t build_stream = nullptr;   // a plan's set-up (uploads, layout kernels) runs here, beside the launch stream
  // device blocks that plans hand back (dsgd_plan_destroy) and take again (dsgd_plan_create): an epoch of the reference is
  // one plan (core/Master.scala:179-199), so plans come and go with every epoch -- hipMalloc / hipFree per plan (the
  // latter synchronises the device) would sit in the fit loop
  struct CacheBlock {
    void* p = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;            // recorded on the launch stream when the block came back: its last reader
    unsigned long long tick = 0;        // cache_tick when it came back: a block no plan took for CACHE_MAX_AGE hand-backs is freed
  };
  std::vector<CacheBlock> cache;
  size_t cache_bytes = 0;
  unsigned long long cache_tick = 0;
  size_t cache_cap = (size_t)8 << 30;   // DSGD_CACHE_MB (dsgd_cache_trim gives blocks back on request)
  std::vector<hipEvent_t> ev_pool;      // events of blocks in use, for the next ones
  // scratch of dsgd_plan_create_from_seed, kept across epochs (grow-only): a hipFree per epoch would wait for the whole
  // device -- i.e. for the epoch that is running on the launch stream while the next one's lists are drawn
  void* seed_scratch[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t seed_scratch_bytes[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  // the one-step layout of per-request steps (dsgd_cs_request_kernel), strides at their maxima, per slice count
  struct ReqLayout {
    CsHdr* hdr = nullptr;
    unsigned int* meta = nullptr;
    unsigned short* rf = nullptr;
    unsigned short* col = nullptr;
    float* val = nullptr;
    unsigned short* cl = nullptr;
    int G = 0;
  } req_layout;
  bool vt_enable = true;                // DSGD_VT=0: index-list steps of resident plans through dsgd_mb_grad_kernel
  long long vt_pack_mb = 2048;          // DSGD_VT_PACK_MB: plans whose packed copy fits get one (0: descriptors only)
  int vt_tpw = 1;                       // DSGD_VT_TPW: virtual tiles per wave the grid is sized for (measured: 1 beats 2-4 up to B = 65,536)
  std::vector<signed char> h_label;
  // wave tiles over d_hcol/d_hval
  WTile* d_wtiles = nullptr;
  unsigned short* d_wmeta = nullptr;
  long long n_wtiles = 0;
  std::vector<int> h_wtile_r0;          // first row of every wave tile (+ sentinel n_rows)
  std::vector<long long> wlong_rows;    // rows that fit no wave tile (sorted): one wave per row, from the whole CSR
  std::vector<long long> wlong_weight;  // prefix sums of their weights in the row chunks' balance (slots; fstep_layout)
  int* d_wlong_rows = nullptr;          // the same list on the device
  int* d_part = nullptr;                // per-workgroup partial sums of the wseg gradient kernel: part_wgs x part_stride
  long long part_wgs = 0;
  int part_stride = 0;
  int hsplit = (DSGD_LDS_FLOATS - 16 * WS_COEF_STRIDE - 4 - 64) / 2;         // hot ranks: 2 * hsplit words of LDS
  unsigned short* d_hcol = nullptr;     // hot stream (16-bit ranks < hsplit), WS_PAD elements of padding
  float* d_hval = nullptr;
  long long* d_hrow_ptr = nullptr;      // n_rows + 1 (rows of the long list: empty)
  long long hot_nnz = 0;
  void* d_ccol = nullptr;               // cold stream in row order: rank - hsplit (16-bit words, or 32-bit when there
  float* d_cval = nullptr;              // are more than 65536 cold columns), value; WS_PAD elements of padding
  bool cold_col16 = false;
  long long* d_ctp = nullptr;           // n_rows + 1: slot offsets of the cold stream (a tiled row owns >= 1 slot)
  long long coldm_nnz = 0;              // slots of the cold stream
  WTile* d_ctiles = nullptr;            // wave tiles over d_ccol/d_cval (same records and lane descriptors as the hot ones)
  unsigned short* d_cmeta = nullptr;
  long long n_ctiles = 0;
  std::vector<int> h_ctile_r0;          // first row of every cold tile (+ sentinel n_rows)
  float* d_dcold = nullptr;             // n_rows: cold part of x.w (rows without cold entries stay 0)
  int* d_partc = nullptr;               // per-workgroup cold gradient partials: partc_wgs x partc_stride
  long long partc_wgs = 0;
  int partc_stride = 0;
  signed char* d_coef8 = nullptr;  // n_rows: gate coefficient y * [y (x . w) >= 0] of the last whole-range step
  const char* last_grad_kernel = "";
  // vectors
  float* d_w = nullptr;
  float* d_ds = nullptr;
  float* d_g = nullptr;  // g_cap x dp
  long long* d_g64 = nullptr;  // g_cap x dp fixed-point accumulators of the streaming kernel (zero between steps)
  float fix_scale = 4194304.0f;  // 2^FIX_SHIFT / vmax2
  int vexp = 0;                  // vmax2 = 2^vexp >= max |value|
  int last_shift = FIX_SHIFT;    // shift of the last gradient launch (streaming kernels or index-list kernel)
  std::vector<StreamSeg> bound_segs;   // split layout: the (ranges, grid) configuration bound_shift was measured for
  unsigned bound_grid = 0;
  int bound_shift = 0;
  unsigned int* d_bound = nullptr;
  int max_shift = FIX_SHIFT;     // DSGD_FIX_SHIFT: cap of the per-launch fixed-point shift of the split layout
  // Row chunks (csrc/dsgd_fstep.hpp): the whole gradient of a row range in ONE launch.  Per (row ranges, workgroups per
  // worker) configuration the host cuts the ranges into chunks balanced by stream bytes and lays out wave tiles of both
  // streams that end at the chunk boundaries; a few configurations stay cached (a fit alternates train steps only; the
  // tests and the bench's legs bring their own).
  struct FstepLayout {
    std::vector<long long> ranges;   // row_begin, row_end per worker
    int n_wg = 0;                    // workgroups (chunks) per worker
    WTile* d_tiles = nullptr;
    unsigned short* d_meta = nullptr;
    WTile* d_ctiles = nullptr;
    unsigned short* d_cmeta = nullptr;
    FChunk* d_chunks = nullptr;
    long long worst_rows = 1;        // rows of the largest chunk
    int shift = -1;                  // the measured fixed-point shift of the hot accumulators (-1: not measured yet)
    unsigned long long used = 0;
    // measured balance (round 6): the workgroups' own durations, summed by the kernel over `launches` launches, re-cut the
    // chunks once or twice per configuration (a chunk's share of the weight follows its workgroup's measured rate)
    unsigned long long* d_times = nullptr;
    unsigned long long* h_times = nullptr;   // pinned
    hipEvent_t times_ev = nullptr;
    std::vector<double> share;       // per chunk: its share of its worker's weight (empty: equal shares)
    int launches = 0, rebalances = 0;
    bool times_pending = false;
  };
  std::vector<FstepLayout> fstep_cache;
  unsigned long long fstep_clock = 0;
  bool fstep_enable = true;          // DSGD_FSTEP=0: row ranges through the three streaming launches / the row-wise kernel
  bool fstep_rebalance = false;      // DSGD_FSTEP_REBALANCE=1: the chunks re-cut by their workgroups' measured durations (measured: 98.4 ->
                                     //   97.5 us at N = 804,414, nothing at 2 M / 6.7 M rows, and two layout rebuilds per configuration --
                                     //   which workgroup is slow is not a stable property of its chunk: OFF, profiles/r06_fstep_wg_times.txt)
  int last_fstep_rebalances = 0;     // ... how often the configuration of the last chunked launch has been re-cut (dsgd_tuning_info)
  long long fstep_min = 65536;       // DSGD_FSTEP_MIN / DSGD_FSTEP_MAX: row ranges of this many rows in total take the chunked launch
  long long fstep_max = 1LL << 31;   //   (measured, whole-split steps, us: 80 K rows 56 -> 50-54, 643 K 118 -> 105, 1.6 M 213 -> 199,
                                     //    3.2 M 374 -> 343, 6.7 M 699 -> 663; at 18.5 K rows the row-wise kernel stays ahead: 37 vs 35-41)
  long long fstep_rows = 512;        // DSGD_FSTEP_ROWS: a chunk holds at least this many rows (fewer workgroups for small ranges:
                                     //   every workgroup moves 378 KB of LDS tiles in and out whatever its chunk)
  // Column lists (csrc/dsgd_tcol.hpp): whole-split steps of 10^3 .. 10^5 rows as dot + column-wise gradient + reduce, no
  // partials.  Per (row ranges) configuration the device sorts the ranges' entries by (worker, column) once; a few
  // configurations stay cached.
  struct TcolLayout {
    std::vector<long long> ranges;   // row_begin, row_end per worker
    long long gen = -1;              // layout_gen the ranked columns belong to
    long long n_ent = 0;             // entries of the ranges' rows
    int share = 0, n_wg = 0;         // entries per workgroup of the gradient kernel, its workgroups
    int bm_words = 0;                // words of the step's bitmap (every worker's range padded to 64 rows; a multiple of 4)
    unsigned int* d_ent_pk = nullptr;
    float* d_ent_val = nullptr;
    TcShare* d_shares = nullptr;
    int* d_key_of_cid = nullptr;
    int* d_bit_base = nullptr;       // [workers]
    unsigned int* d_bitmap = nullptr;   // the gate's decisions of the last step over these ranges
    unsigned long long used = 0;
  };
  std::vector<TcolLayout> tcol_cache;
  unsigned long long tcol_clock = 0;
  int tcol_miss_streak = 0;          // layouts built in a row without one being used again
  int tcol_cooldown = 0;             // steps the column lists still sit out after such a streak (then they try again)
  bool tcol_enable = true;           // DSGD_TCOL=0: such ranges through the row-wise kernel
  long long tcol_min = 512;          // DSGD_TCOL_MIN / DSGD_TCOL_MAX: row ranges of this many rows in total take the column lists
  long long tcol_max = 98303;        //   (above: row chunks, dsgd_fstep.hpp.  Measured, whole-split steps, us, row-wise / chunks / columns:
                                     //    4,800 rows 22.7 / 27.4 / 18.1; 18,519: 30.9 / 34.8 / 20.6; 40,000: 36.6 / 42.7 / 31.7; 80,441: 49 / 48 / 42.8;
                                     //    160,000: 73 / 57.5 / 73; 320,000: 117 / 65 / 162 -- profiles/r05_tcol_probe_*.json)
  long long tcol_max_nnz = 4500000;  // DSGD_TCOL_MAX_NNZ: ... and of at most this many non-zeros where row chunks can take over (round 6: the
                                     //   crossover is a matter of ENTRIES -- 80 K rows of 150 non-zeros: columns 67.8 us, chunks 51.7; of 40: 33.3 / 37.8;
                                     //   RCV1-like rows of 75: columns win up to ~60 K rows = 4.5 M entries, profiles/r06_dispatch_table.txt)
  int tcol_share = 0;                // DSGD_TCOL_SHARE: entries per workgroup of the gradient kernel (0: entries / CUs, within [1024, 8192])
  bool fused_apply_pending = false;
  FusedArgs fused_args{};
  float* d_redpart = nullptr;    // per-block partial sums of w.ds and |w|^2 of the fused reduce + apply kernel
  int redpart_cap = 0;
  // Pinned staging of the per-request entry points (Slave.gradient / Slave.forward hand over w and the sample indices
  // and get a dense vector back): a copy between pageable memory and the device is staged by the runtime, blocks the
  // calling thread and cannot overlap the kernels around it (151 us for a batch-size-100 dsgd_gradient, of which the
  // kernels are ~35: profiles/r02_boundary_latency.json).
  struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;   // the last copy FROM this buffer to the device
    bool armed = false;
  };
  Pinned pin_w, pin_idx, pin_segs, pin_out, pin_upd;
  // dsgd_update_grad: persistent device staging (keys, values) -- no allocation per call, nothing that synchronises the
  // device while the persistent engine is resident
  int* d_upd_key = nullptr;
  float* d_upd_dv = nullptr;
  long long upd_cap = 0;
  hipStream_t upd_stream = nullptr;
  float* d_pred = nullptr;     // dsgd_forward's predictions (grown on demand)
  long long pred_cap = 0;
  bool fix_bound = true;         // DSGD_FIX_BOUND=0: keep the data-independent bound (rows per workgroup x largest value)
  int g_cap = 0;
  float* d_gsum = nullptr;  // dp (all-reduce buffer / sum over hosted workers)
  float* d_tmp = nullptr;   // dp scratch (ranked order)
  float* d_io = nullptr;    // dp staging for vectors crossing the API in key order
  DevScalars* d_sc = nullptr;
  DevScalars* h_sc = nullptr;  // pinned
  // per-request steps: {n_active, err} written by the request's last kernel into host-mapped memory (no copy back);
  // n_active is read as a DIFFERENCE against the value the host last saw, so the request needs no memset either
  int* h_req = nullptr;                   // host-mapped index lists of a per-request step (REQ_MAPPED_ITEMS entries) ...
  int* d_req = nullptr;                   // ... their device address
  const int* cur_idx = nullptr;           // where stage_lists put the lists of the request in flight
  bool req_mapped = true;                 // DSGD_REQ_MAPPED=0: always the copy on the stream
  bool req_spin = true;                   // DSGD_REQ_SPIN=0: wait for the stream instead of polling the mailbox
  bool req_plan = false;                  // DSGD_REQ_PLAN=1: one-worker requests of <= 192 rows through the one-workgroup kernel (its
                                          //   launch re-derives s in fp64 and sets up 147 KB of LDS for ONE step: 37 us per request at the
                                          //   C ABI against 32 us through the row-parallel kernels -- profiles/r04_boundary_latency.json)
  unsigned long long* h_mail = nullptr;   // host-mapped: {n_active, err, sequence number of the request that wrote them, -}
  unsigned long long mail_seq = 0;        // requests answered through the mailbox so far
  unsigned long long* d_mail = nullptr;   // ... its device address
  bool ctr_known = false;                 // the host knows the device's n_active (ctr_last) and that err is clear
  unsigned long long ctr_last = 0;
  bool s_dirty = true;
  bool s_lazy = false;      // s and |w|^2 of the resident w are still per-block pairs in d_redpart[red_par] (fra_scalars)
  int red_par = 0;          // the half of d_redpart the last scalar-writing kernel wrote
  bool nsq_dirty = false;   // |w|^2 stale although s is current (after dsgd_plan_kernel)
  bool have_ds = false;
  // staging for host-provided index lists
  int* d_idx = nullptr;
  long long idx_cap = 0;
  WorkSeg* d_segs = nullptr;
  int segs_cap = 0;
  std::vector<WorkSeg> segs_last;  // what d_segs currently holds
  long long pending_samples = 0;   // rows enqueued by *_async calls since the last dsgd_synchronize
  // persistent Hogwild engine
  hipStream_t async_stream = nullptr;
  hipStream_t query_stream = nullptr;
  HogState* d_hog = nullptr;
  HogState* h_hog = nullptr;   // pinned
  float* d_gcold = nullptr;
  long long* d_asg = nullptr;  // begin[n], end[n]
  unsigned long long* d_hog_it = nullptr;   // per worker: iterations done (continues across exchange rounds)
  unsigned int* d_trace = nullptr;          // dsgd_async_set_trace: one record per update of the next engine runs
  long long trace_cap = 0;                  // records asked for
  long long trace_words = 0;                // words allocated at d_trace
  int trace_mw = 0;                         // mask words per record of the last traced run ((batch + 31) / 32)
  int trace_batch = 0;                      // ... and its batch size (a record carries one x . w per sampled row)
  float* d_tdot = nullptr;                  // traced runs: n_workers x batch, the x . w of every worker's mini-batch in flight
  long long tdot_words = 0;
  int* h_one = nullptr;        // pinned constant 1: source of the stop-flag copy
  // small-batch plan kernel (one persistent workgroup): cold strip and the multi-worker sum buffer
  unsigned long long* d_tprof = nullptr;   // DSGD_PLAN_PROF=1: phase cycle counters of dsgd_plan_kernel (tuning runs)
  float* d_plan_gcold = nullptr;
  bool plan_kernel = true;     // DSGD_PLAN_KERNEL=0: the multi-launch small-batch path
  int hog_hl = HOG_HL, hog_wl = HOG_WL;   // LDS-resident ranks of the Hogwild engine (accumulators / weight copy)
  long long plan_max_rows = 2048;   // steps with more rows in total use the multi-workgroup kernels
  int hog_workers = 0;      // capacity of the per-worker buffers
  int hog_n = 0, hog_batch = 0, hog_bug = 0;   // the running configuration
  float hog_lr = 0.0f;
  unsigned long long hog_seed = 0;
  long long exchange_every = 0;   // dsgd_async_set_exchange: cross-GPU exchange period in local updates (0 = none)
  float* d_wprev = nullptr;       // weights at the last exchange
  float* d_wdelta = nullptr;      // 2 x dp: all-reduced updates, this replica's own part
  bool async_running = false;
  bool join_in_progress = false;            // one thread blocks on the engine (without the mutex); the others wait for it
  std::condition_variable join_cv;
  int join_rc = DSGD_OK;                    // what that thread's join returned: every waiter reports it
  std::string join_err;
  // exchange mode: the rounds (engine launch, delta, all-reduce, apply) are enqueued by a helper thread so that
  // dsgd_async_start returns at once and dsgd_async_updates / dsgd_async_stop stay responsive
  std::thread exch_thread;
  std::atomic<bool> exch_done{true};
  int exch_rc = DSGD_OK;
  std::string exch_err;
  // comm
  rccl::comm_t comm = nullptr;
  bool comm_broken = false;   // a grouped collective failed half way: the communicator was aborted
  bool comm_broken_ok = false;   // (set only inside dsgd_comm_destroy: the one call a broken context accepts)
  int world = 1, rank = 0;
  // profiling of the gradient kernel
  bool prof = false;
  bool prof_main_only = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  size_t prof_used = 0;
  std::vector<int> prof_kind;
  double prof_kms[3] = {0.0, 0.0, 0.0};
  long long prof_kn[3] = {0, 0, 0};
  int n_cu = 256;
};

static int check_ctx(dsgd_ctx* c) {
  if (!c) return fail(DSGD_EINVAL, "null context");
  return DSGD_OK;
}
static int cs_sp(int dp, int G) { return (((dp + G - 1) / G) + 4) & ~3; }   // padded columns per slice (as the kernel computes it)
// Consecutive column-slice launches keep the weights slice-major (a launch then loads and stores ONE contiguous piece per
// workgroup); whatever else touches w first gets them back in rank order.  Every entry point binds, under the context
// mutex: the one place.
static int cs_unslice(dsgd_ctx* c) {
  if (!c->cs_w_G) return DSGD_OK;
  hipLaunchKernelGGL(dsgd_cs_unslice_kernel, dim3((c->dp + 255) / 256), dim3(256), 0, c->stream, c->d_cs_w, c->d_w, c->dp, c->cs_w_G,
                     cs_sp(c->dp, c->cs_w_G));
  HIP_TRY(hipGetLastError());
  c->cs_w_G = 0;
  return DSGD_OK;
}
static int bind(dsgd_ctx* c, bool keep_sliced = false) {  // host threads migrate (JVM pool): bind the device on every call
  HIP_TRY(hipSetDevice(c->cfg.device));
  if (c->comm_broken && !c->comm_broken_ok)
    return fail(DSGD_ERCCL, "the context's communicator was aborted after a collective that not every rank joined: "
                            "dsgd_comm_destroy, then attach a new one (without it the replicas would silently diverge)");
  if (c->cs_w_G && !keep_sliced) return cs_unslice(c);
  return DSGD_OK;
}
static CsrView view(dsgd_ctx* c) {
  CsrView v;
  v.n_rows = c->n_rows;
  v.row_ptr = c->d_row_ptr;
  v.col = c->d_col;
  v.val = c->d_val;
  v.label = c->d_label;
  return v;
}

// host -> device through a pinned buffer of the context: wait until the previous copy out of it has executed, then the
// caller fills it and enqueues the copy (pin_sent); device -> host: enqueue into pin_out, synchronise, copy out.
static int pin_acquire(dsgd_ctx::Pinned& b, size_t bytes) {
  if (b.armed) {
    HIP_TRY(hipEventSynchronize(b.ev));
    b.armed = false;
  }
  if (bytes > b.cap) {
    if (b.p) HIP_TRY(hipHostFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    const size_t cap = std::max<size_t>(bytes, 4096);
    HIP_TRY(hipHostMalloc(&b.p, cap, hipHostMallocDefault));
    b.cap = cap;
  }
  if (!b.ev) HIP_TRY(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  return DSGD_OK;
}
static int pin_sent(dsgd_ctx* c, dsgd_ctx::Pinned& b, hipStream_t on = nullptr) {
  HIP_TRY(hipEventRecord(b.ev, on ? on : c->stream));
  b.armed = true;
  return DSGD_OK;
}
static void pin_free(dsgd_ctx::Pinned& b) {
  if (b.ev) (void)hipEventDestroy(b.ev);
  if (b.p) (void)hipHostFree(b.p);
  b = dsgd_ctx::Pinned();
}

static int ensure_g(dsgd_ctx* c, int n_workers) {
  if (n_workers <= c->g_cap) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_g) HIP_TRY(hipFree(c->d_g));
  if (c->d_g64) HIP_TRY(hipFree(c->d_g64));
  c->d_g = nullptr;
  c->d_g64 = nullptr;
  HIP_TRY(hipMalloc(&c->d_g, sizeof(float) * (size_t)n_workers * c->dp));
  HIP_TRY(hipMalloc(&c->d_g64, sizeof(long long) * (size_t)n_workers * c->dp));
  HIP_TRY(hipMemsetAsync(c->d_g, 0, sizeof(float) * (size_t)n_workers * c->dp, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_g64, 0, sizeof(long long) * (size_t)n_workers * c->dp, c->stream));
  c->g_cap = n_workers;
  return DSGD_OK;
}
static int ensure_idx(dsgd_ctx* c, long long n) {
  if (n <= c->idx_cap) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_idx) HIP_TRY(hipFree(c->d_idx));
  c->d_idx = nullptr;
  long long cap = std::max<long long>(n, 2 * c->idx_cap);
  HIP_TRY(hipMalloc(&c->d_idx, sizeof(int) * (size_t)cap));
  c->idx_cap = cap;
  return DSGD_OK;
}
static int ensure_segs(dsgd_ctx* c, int n) {
  if (n <= c->segs_cap) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_segs) HIP_TRY(hipFree(c->d_segs));
  c->d_segs = nullptr;
  HIP_TRY(hipMalloc(&c->d_segs, sizeof(WorkSeg) * (size_t)n));
  c->segs_cap = n;
  c->segs_last.clear();
  return DSGD_OK;
}

// upload work segments; identical consecutive uploads (timed loops over the same ranges) are skipped
static int upload_segs(dsgd_ctx* c, const std::vector<WorkSeg>& segs) {
  const int n = (int)segs.size();
  DSGD_TRY(ensure_segs(c, n));
  if ((int)c->segs_last.size() == n && memcmp(c->segs_last.data(), segs.data(), sizeof(WorkSeg) * n) == 0) return DSGD_OK;
  // (ordered on the stream behind the launches that still read d_segs)
  DSGD_TRY(pin_acquire(c->pin_segs, sizeof(WorkSeg) * (size_t)n));
  memcpy(c->pin_segs.p, segs.data(), sizeof(WorkSeg) * (size_t)n);
  HIP_TRY(hipMemcpyAsync(c->d_segs, c->pin_segs.p, sizeof(WorkSeg) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  DSGD_TRY(pin_sent(c, c->pin_segs));
  c->segs_last = segs;
  return DSGD_OK;
}

// per-block partial sums of w.ds and |w|^2 (fra_scalars): one pair per FRA_COLS columns
static int ensure_redpart(dsgd_ctx* c) {
  const int blocks = (c->dp + FRA_COLS - 1) / FRA_COLS;
  if (blocks <= c->redpart_cap) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->d_redpart) HIP_TRY(hipFree(c->d_redpart));
  c->d_redpart = nullptr;
  HIP_TRY(hipMalloc(&c->d_redpart, sizeof(float) * 4 * (size_t)blocks));   // two halves: see red_out
  c->redpart_cap = blocks;
  return DSGD_OK;
}

// The per-block pairs (w . ds, |w|^2) live in one of two halves of d_redpart: a kernel that writes new pairs takes
// the half the previous one did not (a lazy consumer may still be reading the old pairs in the same launch).
static float* red_cur(dsgd_ctx* c) { return c->d_redpart + (size_t)c->red_par * 2 * (size_t)c->redpart_cap; }
static float* red_out(dsgd_ctx* c) { return c->d_redpart + (size_t)(1 - c->red_par) * 2 * (size_t)c->redpart_cap; }

// s = 2*lambda*(w.ds) must match the resident w.  allow_lazy: the caller goes on to the fused reduce + update, which
// adds the pairs itself; everybody else gets the scalars finalised in DevScalars.
static int ensure_s(dsgd_ctx* c, bool allow_lazy = false) {
  const int blocks = (c->dp + FRA_COLS - 1) / FRA_COLS;
  if (c->s_dirty) {
    c->nsq_dirty = false;
    DSGD_TRY(ensure_redpart(c));
    // (the summation order of the fused step: equal weights give bit-equal s whichever kernel left it)
    hipLaunchKernelGGL(dsgd_wstats_cols_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_w, c->d_ds, c->dp,
                       (float)c->cfg.lambda, c->d_sc, red_out(c));
    HIP_TRY(hipGetLastError());
    c->red_par ^= 1;
    c->s_dirty = false;
    c->s_lazy = false;
    return DSGD_OK;
  }
  if (c->s_lazy && !allow_lazy) {
    hipLaunchKernelGGL(dsgd_scalars_finalize_kernel, dim3(1), dim3(64), 0, c->stream, red_cur(c), (unsigned int)blocks,
                       (float)c->cfg.lambda, c->d_sc);
    HIP_TRY(hipGetLastError());
    c->s_lazy = false;
  }
  return DSGD_OK;
}

static int read_scalars(dsgd_ctx* c) {
  HIP_TRY(hipMemcpyAsync(c->h_sc, c->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DSGD_OK;
}
static int reset_counters(dsgd_ctx* c) {
  // err .. counts: everything after s_reg / wnorm2
  HIP_TRY(hipMemsetAsync((char*)c->d_sc + offsetof(DevScalars, err), 0, sizeof(DevScalars) - offsetof(DevScalars, err),
                         c->stream));
  c->ctr_known = true;   // (until the next gradient launch)
  c->ctr_last = 0;
  return DSGD_OK;
}
static int check_err_flag(dsgd_ctx* c) {
  const int err = c->h_sc->err;
  if (err) {
    HIP_TRY(hipMemsetAsync(&c->d_sc->err, 0, sizeof(int), c->stream));
    // the abort word FIRST, whatever else is set: left raised it would end every later column-slice launch at its first poll
    if ((err & (8 | 16)) && c->d_cs_sync) HIP_TRY(hipMemsetAsync(c->d_cs_sync, 0, sizeof(unsigned int) * 2, c->stream));
    if (err & 2)
      return fail(DSGD_ESTATE, "fixed-point gradient accumulator left its safe band; the step is invalid");
    if (err & 16) return 1;   // (a per-request step beyond the one-step layout: nothing was applied -- its peers' 8 comes with it)
    if (err & 8)
      return fail(DSGD_ESTATE, "the column-slice kernel's exchange between its workgroups timed out; the run is invalid "
                               "(DSGD_CS=0 selects the row-parallel kernels)");
    if (err & 4)
      return fail(DSGD_ESTATE, "a small-batch plan was created for other data than is loaded now (its lists no longer fit "
                               "the staged sub-batch): create the plan again");
    return fail(DSGD_ERANGE, "sample index / key outside the loaded data");
  }
  return DSGD_OK;
}

static int grid_for(dsgd_ctx* c, long long items, int group) {
  const long long groups_per_block = 256 / group;
  long long blocks = (items + groups_per_block - 1) / groups_per_block;
  const long long cap = (long long)c->n_cu * 8;  // memory-bound: ~8 blocks of 256 per CU, grid-stride the rest
  return (int)std::max<long long>(1, std::min(blocks, cap));
}

// kind 0: the main gradient kernel, 1 / 2: the cold stream's dot / gradient pass (dsgd_cold_kernel)
static int prof_begin(dsgd_ctx* c, size_t* slot, int kind = 0) {
  if (!c->prof || (kind != 0 && c->prof_main_only)) {   // (a skipped bracket: prof_end ignores the slot)
    *slot = (size_t)-1;
    return DSGD_OK;
  }
  if (c->prof_used == c->prof_ev.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    c->prof_ev.emplace_back(a, b);
    c->prof_kind.push_back(0);
  }
  *slot = c->prof_used++;
  c->prof_kind[*slot] = kind;
  HIP_TRY(hipEventRecord(c->prof_ev[*slot].first, c->stream));
  return DSGD_OK;
}
static int prof_end(dsgd_ctx* c, size_t slot) {
  if (!c->prof || slot == (size_t)-1) return DSGD_OK;
  HIP_TRY(hipEventRecord(c->prof_ev[slot].second, c->stream));
  return DSGD_OK;
}
static int prof_collect(dsgd_ctx* c) {  // stream must be idle
  for (size_t i = 0; i < c->prof_used; ++i) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, c->prof_ev[i].first, c->prof_ev[i].second));
    const int k = c->prof_kind[i];
    c->prof_kms[k] += ms;
    c->prof_kn[k]++;
  }
  c->prof_used = 0;
  return DSGD_OK;
}

static int ensure_part(dsgd_ctx* c, int** buf, long long* wgs, int* stride, long long need_wgs, int need_cols);

// Index-list batches spread over workgroups (dsgd_mb_grad_kernel): per-workgroup fixed-point partials, then the exact
// finish of the streaming path.  `allow_fused`: the caller goes on to launch_finish_sync (one hosted worker without
// peers then gets regulariser + update fused into the reduction); otherwise the sums land in d_g.
static int launch_grad_mb(dsgd_ctx* c, const int* d_idx, const WorkSeg* d_segs, int n_workers, long long max_items,
                          bool allow_fused) {
  const int hl = std::min(c->dp, MB_HL);
  const long long per_worker = std::max<long long>(1, c->n_cu / n_workers);
  long long rows_per_wg = std::max<long long>(64, (max_items + per_worker - 1) / per_worker);
  const long long wgs = std::max<long long>(1, (max_items + rows_per_wg - 1) / rows_per_wg);
  DSGD_TRY(ensure_part(c, &c->d_part, &c->part_wgs, &c->part_stride, wgs * n_workers, hl));
  int bits = 0;
  while ((1LL << bits) < rows_per_wg) ++bits;
  const int shift = 30 - bits;   // at most one contribution per row and column: a workgroup's sums stay below 2^30
  c->last_shift = shift;
  MbArgs a;
  a.m = view(c);
  a.w = c->d_w;
  a.idx = d_idx;
  a.segs = d_segs;
  a.part = c->d_part;
  a.g64_base = c->d_g64;
  a.g_stride = c->dp;
  a.sc = c->d_sc;
  a.qscale = std::ldexp(1.0f, shift - c->vexp);
  a.part_stride = c->part_stride;
  a.rows_per_wg = (int)rows_per_wg;
  a.hl = hl;
  a.wl = std::min(MB_WL, c->dp) & ~255;   // whole 1 KiB pieces
  a.dp = c->dp;
  a.tprof = c->d_tprof;
  const size_t lds = sizeof(float) * (size_t)mb_lds_words(hl, a.wl);
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  hipLaunchKernelGGL(dsgd_mb_grad_kernel, dim3((unsigned)wgs, n_workers), dim3(MB_THREADS), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  c->last_grad_kernel = "dsgd_mb_grad_kernel";
  const double inv = 1.0 / (double)a.qscale;
  c->fused_apply_pending = false;
  if (allow_fused) {
    DSGD_TRY(ensure_redpart(c));
    // (no cold partials here: nc = 0, and hc only has to lie at or beyond D + 1 -- a multiple of 4 keeps the reduce on
    //  its 16-byte loads; with hc = D + 1 = 47,237 it fell back to 4-byte loads: 21.5 us for 25 MB of partials)
    c->fused_args = {hl, (int)wgs, (c->dp + 3) & ~3, 0, 0, inv, inv};
    c->fused_apply_pending = true;   // launched by launch_finish_sync, which knows lr
    return DSGD_OK;
  }
  hipLaunchKernelGGL(dsgd_fix_reduce_kernel, dim3((c->dp + 63) / 64, n_workers), dim3(1024), 0, c->stream, c->d_g64, c->d_g,
                     (long long)c->dp, c->dp, hl, c->d_part, c->part_stride, (int)wgs, c->dp, 0, (const int*)nullptr, 0, 0, inv, inv);
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}

// ---- virtual tiles (dsgd_vt_grad_kernel) ---------------------------------------------------------------------
// Lay the lists of a plan out over the split streams: a row gets max(ceil(hot slots / 8), cold entries, 1) consecutive
// lanes of a 64-lane tile (whole rows per tile, every list starts a tile).  Not possible (vt_ok = false, the plan keeps
// using dsgd_mb_grad_kernel) when a row sits on the long-row list, holds more than 64 cold entries, or the streams are
// not in their 16-bit / 32-bit-addressable form.
// (returns 1 for a SOFT failure -- host or device memory for the layout could not be had: vt_build frees what was
//  allocated and the plan runs from dsgd_mb_grad_kernel, exactly as for lists the layout cannot hold)
#define VT_SOFT(expr)                 \
  do {                                \
    if ((expr) != hipSuccess) {       \
      (void)hipGetLastError();        \
      return 1;                       \
    }                                 \
  } while (0)
static int plan_host_idx(dsgd_ctx* c, dsgd_plan* p);
static int vt_build_impl(dsgd_ctx* c, dsgd_plan* p) {
  p->vt_layout = c->layout_gen;
  p->vt_ok = false;
  (void)hipFree(p->d_vt_lanes);
  (void)hipFree(p->d_vt_segs);
  (void)hipFree(p->d_vt_long);
  (void)hipFree(p->d_vt_packed);
  p->d_vt_lanes = nullptr;
  p->d_vt_segs = nullptr;
  p->d_vt_long = nullptr;
  p->d_vt_packed = nullptr;
  const int H = std::min(c->hsplit, c->dp);
  const long long n_lists = (long long)p->n_steps * p->n_workers;
  if (!c->cold_col16 || c->dp <= H || c->hot_nnz + WS_PAD >= (1LL << 32) || c->coldm_nnz + WS_PAD >= (1LL << 32)) return DSGD_OK;
  DSGD_TRY(plan_host_idx(c, p));
  if (c->h_hrp.size() != (size_t)c->n_rows + 1 || (long long)p->h_idx.size() != p->offsets[n_lists]) return DSGD_OK;
  if ((long long)c->h_ccol.size() != c->coldm_nnz) return DSGD_OK;
  const std::vector<long long>&hrp = c->h_hrp, &ctp = c->h_ctp, &crp = c->h_crow_ptr;
  std::vector<VtLane> lanes;
  lanes.reserve((size_t)p->offsets[n_lists] * 10);
  std::vector<int> tile_rows;
  std::vector<MbRec> long_rows;
  std::vector<long long> long_off((size_t)n_lists + 1, 0);
  p->vt_off.assign((size_t)n_lists + 1, 0);
  const VtLane empty{0u, 0u, 0u, 0u};
  for (long long li = 0; li < n_lists; ++li) {
    int used = 0, rows = 0;   // lanes and rows of the open tile
    for (long long t = p->offsets[li]; t < p->offsets[li + 1]; ++t) {
      const long long r = p->h_idx[(size_t)t];
      if (r < 0 || r >= c->n_rows) return DSGD_OK;       // (the mb kernel reports the bad index)
      const long long hot = hrp[r + 1] - hrp[r], cold = crp[r + 1] - crp[r];
      if (hot <= 0 || cold > 64) {   // long-row list / too many cold entries for a tile: one wave per row, whole CSR
        MbRec lr;
        lr.st = c->h_row_ptr[(size_t)r];
        lr.len = (int)(c->h_row_ptr[(size_t)r + 1] - c->h_row_ptr[(size_t)r]);
        lr.y = (float)c->h_label[(size_t)r];
        long_rows.push_back(lr);
        continue;
      }
      const int nl = (int)std::max<long long>(std::max<long long>((hot + 7) / 8, cold), 1);
      if (used + nl > 64) {                               // close the tile
        lanes.resize(lanes.size() + (size_t)(64 - used), empty);
        tile_rows.push_back(rows);
        used = 0;
        rows = 0;
      }
      const unsigned int ypos = c->h_label[(size_t)r] > 0 ? VT_YPOS : 0u;
      for (int j = 0; j < nl; ++j) {
        VtLane L;
        const long long left = hot - 8LL * j;
        const unsigned int cnt = (unsigned int)std::max<long long>(0, std::min<long long>(8, left));
        L.hp = cnt ? (unsigned int)(hrp[r] + 8LL * j) : 0u;
        L.info = cnt | (j == 0 ? VT_START : 0u) | (j == nl - 1 ? VT_LAST : 0u) | ypos | (j < cold ? VT_COLD : 0u) |
                 ((unsigned int)rows << 16);
        L.cp = j < cold ? (unsigned int)(ctp[r] + j) : 0u;
        L.crank = j < cold ? (unsigned int)c->h_ccol[(size_t)(ctp[r] + j)] : 0u;
        lanes.push_back(L);
      }
      used += nl;
      ++rows;
    }
    lanes.resize(lanes.size() + (size_t)(64 - used), empty);   // (a list is never empty: its last tile is open)
    tile_rows.push_back(rows);
    p->vt_off[(size_t)li + 1] = (long long)tile_rows.size();
    long_off[(size_t)li + 1] = (long long)long_rows.size();
  }
  // per step: the grid (workgroups per worker) and the fixed-point shift from the rows ONE workgroup can meet
  p->vt_grid.assign((size_t)p->n_steps, 1);
  p->vt_gxt.assign((size_t)p->n_steps, 1);
  p->vt_shift.assign((size_t)p->n_steps, 21);
  const long long per_worker = std::max<long long>(1, c->n_cu / p->n_workers);
  std::vector<WorkSeg> segs((size_t)n_lists * 2);
  for (long long li = 0; li < n_lists; ++li) {
    segs[(size_t)(n_lists + li)].begin = long_off[(size_t)li];
    segs[(size_t)(n_lists + li)].end = long_off[(size_t)li + 1];
  }
  for (long long s = 0; s < p->n_steps; ++s) {
    long long max_t = 1;
    for (int k = 0; k < p->n_workers; ++k) {
      const long long li = s * p->n_workers + k;
      segs[(size_t)li].begin = p->vt_off[(size_t)li];
      segs[(size_t)li].end = p->vt_off[(size_t)li + 1];
      max_t = std::max(max_t, segs[(size_t)li].end - segs[(size_t)li].begin);
    }
    long long max_long = 0;
    for (int k = 0; k < p->n_workers; ++k)
      max_long = std::max(max_long, long_off[(size_t)(s * p->n_workers + k) + 1] - long_off[(size_t)(s * p->n_workers + k)]);
    // workgroups of their own for the rows outside the tiled streams (one wave per row), beside the tile workgroups
    const long long gl = max_long ? std::max<long long>(1, std::min<long long>((max_long + 15) / 16, std::max<long long>(1, per_worker / 8))) : 0;
    const long long gx = std::max<long long>(1, std::min(per_worker - gl, (max_t + 16LL * c->vt_tpw - 1) / (16LL * c->vt_tpw)));
    long long worst = 1;
    std::vector<long long> rows_of((size_t)gx);
    for (int k = 0; k < p->n_workers; ++k) {
      const WorkSeg& sg = segs[(size_t)(s * p->n_workers + k)];
      std::fill(rows_of.begin(), rows_of.end(), 0);
      for (long long t = sg.begin; t < sg.end; ++t) rows_of[(size_t)(((t - sg.begin) / 16) % gx)] += tile_rows[(size_t)t];
      for (long long v : rows_of) worst = std::max(worst, v);
    }
    if (gl) worst = std::max(worst, (max_long + 16 * gl - 1) / (16 * gl) * 16);   // rows one long-row workgroup can meet
    int bits = 0;
    while ((1LL << bits) < worst) ++bits;
    p->vt_gxt[(size_t)s] = (int)gx;
    p->vt_grid[(size_t)s] = (int)(gx + gl);
    p->vt_shift[(size_t)s] = std::min(21, 30 - bits);   // (<= 21: the fixed-point conversion is one fma against 1.5 * 2^23)
  }
  VT_SOFT(hipMalloc(&p->d_vt_lanes, sizeof(VtLane) * std::max<size_t>(lanes.size(), 64)));
  VT_SOFT(hipMalloc(&p->d_vt_segs, sizeof(WorkSeg) * segs.size()));
  VT_SOFT(hipMalloc(&p->d_vt_long, sizeof(MbRec) * std::max<size_t>(long_rows.size(), 1)));
  HIP_TRY(hipMemcpy(p->d_vt_lanes, lanes.data(), sizeof(VtLane) * lanes.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_vt_segs, segs.data(), sizeof(WorkSeg) * segs.size(), hipMemcpyHostToDevice));
  if (!long_rows.empty())
    HIP_TRY(hipMemcpy(p->d_vt_long, long_rows.data(), sizeof(MbRec) * long_rows.size(), hipMemcpyHostToDevice));
  // small and mid-size plans get their own copy of the rows they touch, in tile order (dsgd_vt_pack_kernel): one
  // coalesced round trip per step instead of descriptors -> scattered rows
  const long long n_tiles = (long long)tile_rows.size();
  // (bounded by DSGD_VT_PACK_MB and by a quarter of the device memory that is free right now: a plan's copy must not be
  //  what makes the next allocation of the data path fail)
  size_t mem_free = 0, mem_total = 0;
  if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) {
    (void)hipGetLastError();
    mem_free = 0;
  }
  if (n_tiles > 0 && n_tiles * 4096 <= c->vt_pack_mb * (1LL << 20) && (size_t)n_tiles * 4096 <= mem_free / 4 &&
      hipMalloc(&p->d_vt_packed, (size_t)n_tiles * 4096) != hipSuccess) {
    (void)hipGetLastError();      // (no room for the copy: the plan runs from its descriptors)
    p->d_vt_packed = nullptr;
  }
  if (p->d_vt_packed) {
    VtArgs a{};
    a.hcol = c->d_hcol;
    a.hval = c->d_hval;
    a.ccol = reinterpret_cast<const unsigned short*>(c->d_ccol);
    a.cval = c->d_cval;
    a.w = c->d_w;
    a.lanes = p->d_vt_lanes;
    a.hsplit = H;
    const long long blocks = (n_tiles + 3) / 4;   // four tiles (waves) per block
    hipLaunchKernelGGL(dsgd_vt_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, c->stream, a, n_tiles, p->d_vt_packed);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  p->vt_ok = true;
  return DSGD_OK;
}
#undef VT_SOFT
static int vt_build(dsgd_ctx* c, dsgd_plan* p) {
  int rc;
  try {
    rc = vt_build_impl(c, p);
  } catch (const std::bad_alloc&) {   // (160 bytes of host memory per listed row; nothing may unwind across the C ABI)
    rc = 1;
  }
  if (rc == 1) {
    (void)hipFree(p->d_vt_lanes);
    (void)hipFree(p->d_vt_segs);
    (void)hipFree(p->d_vt_long);
    (void)hipFree(p->d_vt_packed);
    p->d_vt_lanes = nullptr;
    p->d_vt_segs = nullptr;
    p->d_vt_long = nullptr;
    p->d_vt_packed = nullptr;
    p->vt_ok = false;   // (vt_layout is stamped: the plan keeps the row-wise kernel until the layout changes)
    return DSGD_OK;
  }
  return rc;
}

static int launch_grad_vt(dsgd_ctx* c, dsgd_plan* p, long long step) {
  const int H = std::min(c->hsplit, c->dp);
  const int gx = p->vt_grid[(size_t)step], shift = p->vt_shift[(size_t)step];
  DSGD_TRY(ensure_part(c, &c->d_part, &c->part_wgs, &c->part_stride, (long long)gx * p->n_workers, H));
  c->last_shift = shift;
  VtArgs a;
  a.hcol = c->d_hcol;
  a.hval = c->d_hval;
  a.ccol = reinterpret_cast<const unsigned short*>(c->d_ccol);
  a.cval = c->d_cval;
  a.w = c->d_w;
  a.lanes = p->d_vt_lanes;
  a.packed = p->d_vt_packed;
  a.tsegs = p->d_vt_segs + step * p->n_workers;
  a.lsegs = p->d_vt_segs + ((long long)p->n_steps + step) * p->n_workers;
  a.long_recs = p->d_vt_long;
  a.mfull = view(c);
  a.part = c->d_part;
  a.g64_base = c->d_g64;
  a.g_stride = c->dp;
  a.sc = c->d_sc;
  a.qscale = std::ldexp(1.0f, shift - c->vexp);
  a.cold_scale = c->fix_scale;
  a.part_stride = c->part_stride;
  a.hsplit = H;
  a.gx_tiles = p->vt_gxt[(size_t)step];
  const size_t lds = sizeof(float) * (size_t)(((H + 4) & ~3) + 16 * 64 + H + 64);
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  if (a.packed) hipLaunchKernelGGL(dsgd_vt_grad_kernel<true>, dim3((unsigned)gx, p->n_workers), dim3(1024), lds, c->stream, a);
  else hipLaunchKernelGGL(dsgd_vt_grad_kernel<false>, dim3((unsigned)gx, p->n_workers), dim3(1024), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  c->last_grad_kernel = "dsgd_vt_grad_kernel";
  DSGD_TRY(ensure_redpart(c));
  // hot ranks: the workgroups' partials at this launch's scale; cold ranks: the 64-bit accumulators at the cold scale
  c->fused_args = {H, gx, H, 0, 0, 1.0 / (double)a.qscale, 1.0 / (double)c->fix_scale};
  c->fused_apply_pending = true;
  return DSGD_OK;
}

// ---- column slices (dsgd_cs_step_kernel) ---------------------------------------------------------------------
// Lay the lists of a plan out per (slice, step): slice b holds the entries whose rank is b (mod G), a row's entries
// inside a slice in slots of <= CS_L.  Not possible (cs_ok = false, the plan keeps the row-parallel kernels) beyond
// CS_MAX_K hosted workers, for steps of more than CS_MAX_SLOTS rows or slots, when the slices do not fit LDS, or when the
// layout would exceed DSGD_CS_MAX_MB.  Returns 1 for a soft failure (memory): the caller frees and falls back.
#define CS_SOFT(expr)                 \
  do {                                \
    if ((expr) != hipSuccess) {       \
      (void)hipGetLastError();        \
      return 1;                       \
    }                                 \
  } while (0)
// ---- device blocks of plans, cached across plans (see dsgd_ctx::CacheBlock) -------------------------------------
// take: the smallest cached block of at least `bytes` and at most twice that; the BUILD stream waits for the block's last
// reader on the launch stream (an event recorded when it came back).  Otherwise hipMalloc.
static int cache_take(dsgd_ctx* c, void** out, size_t bytes, size_t* got) {
  bytes = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
  int best = -1;
  for (int i = 0; i < (int)c->cache.size(); ++i)
    if (c->cache[i].bytes >= bytes && c->cache[i].bytes <= 2 * bytes + (1u << 16) && (best < 0 || c->cache[i].bytes < c->cache[best].bytes)) best = i;
  if (best >= 0) {
    dsgd_ctx::CacheBlock blk = c->cache[(size_t)best];
    c->cache.erase(c->cache.begin() + best);
    c->cache_bytes -= blk.bytes;
    if (blk.ev) {
      if (c->build_stream) HIP_TRY(hipStreamWaitEvent(c->build_stream, blk.ev, 0));
      c->ev_pool.push_back(blk.ev);
    }
    *out = blk.p;
    *got = blk.bytes;
    return DSGD_OK;
  }
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, bytes);
  if (e != hipSuccess && !c->cache.empty()) {   // (memory is tight: give the cached blocks back and try once more)
    (void)hipGetLastError();
    for (auto& b : c->cache) {
      (void)hipFree(b.p);
      if (b.ev) c->ev_pool.push_back(b.ev);
    }
    c->cache.clear();
    c->cache_bytes = 0;
    e = hipMalloc(&q, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 1;   // soft: the caller falls back
  }
  *out = q;
  *got = bytes;
  return DSGD_OK;
}
// blocks beyond `keep_bytes` (oldest first) or older than CACHE_MAX_AGE hand-backs go back to the device: a block whose
// last reader has finished is freed at once, one still being read stays (hipFree would wait for the device)
constexpr unsigned long long CACHE_MAX_AGE = 48;   // (an epoch's plan hands back ~10 blocks: blocks unused for ~5 plans)
static void cache_trim(dsgd_ctx* c, size_t keep_bytes, bool aged_only) {
  for (size_t i = 0; i < c->cache.size();) {
    dsgd_ctx::CacheBlock& b = c->cache[i];
    const bool aged = c->cache_tick - b.tick > CACHE_MAX_AGE;
    const bool over = !aged_only && c->cache_bytes > keep_bytes;
    if ((aged || over) && (!b.ev || hipEventQuery(b.ev) == hipSuccess)) {
      (void)hipFree(b.p);
      if (b.ev) c->ev_pool.push_back(b.ev);
      c->cache_bytes -= b.bytes;
      c->cache.erase(c->cache.begin() + (long)i);
    } else {
      (void)hipGetLastError();   // (hipErrorNotReady of the query)
      ++i;
    }
  }
}
// give back: the block's last reader is whatever the launch stream holds now
static void cache_give(dsgd_ctx* c, void* q, size_t bytes) {
  if (!q) return;
  ++c->cache_tick;
  if ((c->cache_tick & 15) == 0) cache_trim(c, c->cache_cap, true);
  if (bytes == 0 || c->cache_bytes + bytes > c->cache_cap || c->cache.size() >= 256) {
    (void)hipFree(q);   // (not ours to keep: allocated outside the cache, or the cache is full)
    return;
  }
  dsgd_ctx::CacheBlock blk;
  blk.p = q;
  blk.bytes = bytes;
  if (!c->ev_pool.empty()) {
    blk.ev = c->ev_pool.back();
    c->ev_pool.pop_back();
  } else if (hipEventCreateWithFlags(&blk.ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(q);
    return;
  }
  if (hipEventRecord(blk.ev, c->stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c->stream);
  }
  blk.tick = c->cache_tick;
  c->cache.push_back(blk);
  c->cache_bytes += bytes;
}
static void cache_drop_all(dsgd_ctx* c) {   // (dsgd_destroy: the streams are idle)
  for (auto& b : c->cache) {
    (void)hipFree(b.p);
    if (b.ev) (void)hipEventDestroy(b.ev);
  }
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  c->cache.clear();
  c->ev_pool.clear();
  c->cache_bytes = 0;
}

static void cs_free(dsgd_ctx* c, dsgd_plan* p) {
  void* q[6] = {p->d_cs_hdr, p->d_cs_meta, p->d_cs_rf, p->d_cs_col, p->d_cs_val, p->d_cs_cl};
  for (int i = 0; i < 6; ++i) {
    cache_give(c, q[i], p->cs_bytes[i]);   // (size 0: a block the host builder allocated itself -- freed)
    p->cs_bytes[i] = 0;
  }
  p->d_cs_cl = nullptr;
  p->d_cs_hdr = nullptr;
  p->d_cs_meta = nullptr;
  p->d_cs_rf = nullptr;
  p->d_cs_col = nullptr;
  p->d_cs_val = nullptr;
  p->cs_ok = false;
}
// the host copy of a plan's lists: plans drawn on the device (dsgd_plan_create_from_seed) have none until a HOST builder
// (virtual tiles, DSGD_CS_HOST_LAYOUT=1) needs it
static int plan_host_idx(dsgd_ctx* c, dsgd_plan* p) {
  const long long n_lists = (long long)p->n_steps * p->n_workers;
  const long long n = p->offsets.empty() ? 0 : p->offsets[(size_t)n_lists];
  if ((long long)p->h_idx.size() == n || !p->d_idx) return DSGD_OK;
  if (p->built_pending && p->built_ev) HIP_TRY(hipEventSynchronize(p->built_ev));
  p->h_idx.resize((size_t)n);
  HIP_TRY(hipMemcpy(p->h_idx.data(), p->d_idx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
  return DSGD_OK;
}
static int ensure_build_stream(dsgd_ctx* c) {
  if (!c->build_stream) HIP_TRY(hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking));
  return DSGD_OK;
}
// the arrays every column-slice launch shares: the exchange buffer and its abort word (each for itself: a failure between
// the two must not leave the first behind alone)
static int ensure_cs_exchange(dsgd_ctx* c) {
  if (!c->d_cs_x) {
    unsigned long long* x = nullptr;
    if (hipMalloc(&x, sizeof(unsigned long long) * 2 * CS_MAX_G * CS_XSTRIDE) != hipSuccess) {
      (void)hipGetLastError();
      return 1;
    }
    if (hipMemset(x, 0, sizeof(unsigned long long) * 2 * CS_MAX_G * CS_XSTRIDE) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(x);
      return 1;
    }
    c->d_cs_x = x;
  }
  if (!c->d_cs_sync) {
    unsigned int* y = nullptr;
    if (hipMalloc(&y, sizeof(unsigned int) * 2) != hipSuccess) {
      (void)hipGetLastError();
      return 1;
    }
    if (hipMemset(y, 0, sizeof(unsigned int) * 2) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(y);
      return 1;
    }
    c->d_cs_sync = y;
  }
  return DSGD_OK;
}
static int ensure_cs_max(dsgd_ctx* c) {
  if (!c->d_cs_max) HIP_TRY(hipMalloc(&c->d_cs_max, sizeof(unsigned int) * 4));
  if (!c->h_cs_max) HIP_TRY(hipHostMalloc(&c->h_cs_max, sizeof(unsigned int) * 4, hipHostMallocDefault));
  return DSGD_OK;
}
// slices for K hosted workers: 8 up to four (wider slices, fewer peers in the exchange), 16 beyond or for a wide model
static int cs_pick_G(const dsgd_ctx* c, int K, bool request) {
  auto fits = [&](int G) {
    const int words = request ? cs_req_lds_words(c->dp, G, K) : cs_lds_words(c->dp, G, K);
    return c->dp >= 4 * G && words <= DSGD_LDS_FLOATS && (c->dp + G - 1) / G <= 65536;
  };
  int G = c->cs_g ? c->cs_g : (K <= 4 ? 8 : 16);
  if (!c->cs_g && G == 8 && !fits(8)) G = 16;
  return fits(G) ? G : 0;
}

// A plan's column slices laid out by the device (csrc/dsgd_cs.hpp: dsgd_cs_layout_kernel), on the build stream: pass 1
// counts (one read-back of four words: the strides), pass 2 fills.  Returns 1 for a soft failure (memory).
static int cs_build_device_impl(dsgd_ctx* c, dsgd_plan* p) {
  if (!c->layout_seen_by_build) {   // (once per column layout: the layout kernels read the ranked CSR the launch stream wrote)
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->layout_seen_by_build = true;
  }
  cs_free(c, p);
  p->cs_layout = c->layout_gen;
  p->cs_device_built = true;
  const int K = p->n_workers;
  const long long n_steps = p->n_steps, n_lists = n_steps * K;
  if (K > CS_MAX_K || p->max_step_rows > CS_MAX_SLOTS) return DSGD_OK;
  const int G = cs_pick_G(c, K, false);
  if (!G) return DSGD_OK;
  if (n_steps * (long long)G > 0x7fffffffLL) return DSGD_OK;
  if (!p->idx_trusted) {
    if ((long long)p->h_idx.size() != p->offsets[(size_t)n_lists]) return DSGD_OK;
    for (long long r : p->h_idx)
      if (r < 0 || r >= c->n_rows) return DSGD_OK;   // (the row-wise kernel reports the bad index)
  }
  p->cs_shift.assign((size_t)n_steps, 21);
  for (long long st = 0; st < n_steps; ++st) {
    long long worst_list = 1;
    for (int k = 0; k < K; ++k) worst_list = std::max(worst_list, p->offsets[(size_t)(st * K + k) + 1] - p->offsets[(size_t)(st * K + k)]);
    int bits = 0;
    while ((1LL << bits) < worst_list) ++bits;
    p->cs_shift[(size_t)st] = 30 - bits;   // at most one contribution per row and column: a worker's sums stay below 2^30
  }
  DSGD_TRY(ensure_build_stream(c));
  DSGD_TRY(ensure_cs_max(c));
  if (int rc = ensure_cs_exchange(c)) return rc;
  hipStream_t bs = c->build_stream;
  CsBuildArgs ba{};
  ba.m = view(c);
  ba.idx = p->d_idx;
  ba.segs = p->d_segs;
  ba.maxima = c->d_cs_max;
  ba.n_steps_plan = n_steps;
  ba.dp = c->dp;
  ba.G = G;
  ba.K = K;
  HIP_TRY(hipMemsetAsync(c->d_cs_max, 0, sizeof(unsigned int) * 4, bs));
  hipLaunchKernelGGL(dsgd_cs_layout_kernel<false>, dim3((unsigned)(n_steps * G)), dim3(CS_THREADS), 0, bs, ba);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->h_cs_max, c->d_cs_max, sizeof(unsigned int) * 4, hipMemcpyDeviceToHost, bs));
  HIP_TRY(hipStreamSynchronize(bs));
  const long long max_slots = std::max<long long>(1, c->h_cs_max[0]), max_cols = std::max<long long>(1, c->h_cs_max[1]);
  const long long max_rows = std::max<long long>(1, p->max_step_rows);
  if (c->h_cs_max[2] & 1u) return DSGD_OK;
  if (max_slots > CS_MAX_SLOTS || max_rows > CS_MAX_SLOTS || max_cols > (long long)CS_MAX_CLT * CS_THREADS) return DSGD_OK;
  const long long big = std::max(max_slots, max_rows);
  const bool one_fits = big <= CS_THREADS && max_cols <= 4 * CS_THREADS;                  // one slot per lane
  const bool narrow_fits = big <= 2 * CS_THREADS_NARROW && max_cols <= 8 * CS_THREADS_NARROW;
  const int slot_stride = (int)((max_slots + 63) / 64 * 64), row_stride = (int)((max_rows + 1 + 63) / 64 * 64);
  const int cl_stride = (int)((max_cols + CS_THREADS - 1) / CS_THREADS * CS_THREADS);
  const long long cells = (long long)G * n_steps;
  const long long bytes = cells * ((long long)slot_stride * (4 + CS_L * 2 + CS_L * 4) + (long long)row_stride * 2 + (long long)cl_stride * 2 + 8);
  if (bytes > c->cs_max_mb * (1LL << 20)) return DSGD_OK;
  const size_t want[6] = {sizeof(CsHdr) * (size_t)cells,
                          sizeof(unsigned int) * (size_t)(cells * slot_stride),
                          sizeof(unsigned short) * (size_t)(cells * row_stride),
                          sizeof(unsigned short) * (size_t)(cells * slot_stride * CS_L),
                          sizeof(float) * (size_t)(cells * slot_stride * CS_L),
                          sizeof(unsigned short) * (size_t)(cells * cl_stride)};
  void* got[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < 6; ++i) {
    if (int rc = cache_take(c, &got[i], want[i], &p->cs_bytes[i])) {
      for (int j = 0; j < i; ++j) {
        cache_give(c, got[j], p->cs_bytes[j]);
        p->cs_bytes[j] = 0;
      }
      return rc;
    }
  }
  p->d_cs_hdr = static_cast<CsHdr*>(got[0]);
  p->d_cs_meta = static_cast<unsigned int*>(got[1]);
  p->d_cs_rf = static_cast<unsigned short*>(got[2]);
  p->d_cs_col = static_cast<uint4*>(got[3]);
  p->d_cs_val = static_cast<float4*>(got[4]);
  p->d_cs_cl = static_cast<unsigned short*>(got[5]);
  ba.hdr = p->d_cs_hdr;
  ba.slot_meta = p->d_cs_meta;
  ba.row_first = p->d_cs_rf;
  ba.col = reinterpret_cast<unsigned short*>(p->d_cs_col);
  ba.val = reinterpret_cast<float*>(p->d_cs_val);
  ba.clist = p->d_cs_cl;
  ba.slot_stride = slot_stride;
  ba.row_stride = row_stride;
  ba.cl_stride = cl_stride;
  hipLaunchKernelGGL(dsgd_cs_layout_kernel<true>, dim3((unsigned)(n_steps * G)), dim3(CS_THREADS), 0, bs, ba);
  HIP_TRY(hipGetLastError());
  p->cs_G = G;
  p->cs_nt = (c->cs_nt == CS_THREADS_NARROW && narrow_fits) ? CS_THREADS_NARROW : CS_THREADS;
  p->cs_spl = (p->cs_nt == CS_THREADS && one_fits) ? 1 : 2;
  p->cs_slot_stride = slot_stride;
  p->cs_row_stride = row_stride;
  p->cs_cl_stride = cl_stride;
  p->cs_ok = true;
  return DSGD_OK;
}
static int cs_build_impl(dsgd_ctx* c, dsgd_plan* p) {
  cs_free(c, p);
  p->cs_layout = c->layout_gen;
  p->cs_device_built = false;
  // (the lists were uploaded on the build stream: the gather below runs on the launch stream)
  if (p->built_pending) {
    HIP_TRY(hipStreamWaitEvent(c->stream, p->built_ev, 0));
    p->built_pending = false;
  }
  const int K = p->n_workers;
  const long long n_steps = p->n_steps, n_lists = n_steps * K;
  if (K > CS_MAX_K || p->max_step_rows > CS_MAX_SLOTS) return DSGD_OK;
  int G = c->cs_g ? c->cs_g : (K <= 4 ? 8 : 16);
  if (!c->cs_g && G == 8 && cs_lds_words(c->dp, 8, K) > DSGD_LDS_FLOATS) G = 16;   // (a wide model: narrower slices)
  if (c->dp < 4 * G || cs_lds_words(c->dp, G, K) > DSGD_LDS_FLOATS || (c->dp + G - 1) / G > 65536) return DSGD_OK;
  DSGD_TRY(plan_host_idx(c, p));
  if (c->h_row_ptr.size() != (size_t)c->n_rows + 1 || (long long)p->h_idx.size() != p->offsets[n_lists]) return DSGD_OK;
  const long long N = p->offsets[n_lists];
  std::vector<long long> pre((size_t)N + 1);
  pre[0] = 0;
  for (long long t = 0; t < N; ++t) {
    const long long r = p->h_idx[(size_t)t];
    if (r < 0 || r >= c->n_rows) return DSGD_OK;   // (the row-wise kernel reports the bad index)
    pre[(size_t)t + 1] = pre[(size_t)t] + (c->h_row_ptr[(size_t)r + 1] - c->h_row_ptr[(size_t)r]);
  }
  const long long E = pre[(size_t)N];
  if (E > (64LL << 20)) return DSGD_OK;   // (the rows' entries pass through host memory once: 8 bytes each)
  // the listed rows' (rank, value) pairs: the ranked CSR lives on the device only
  std::vector<int> ecol((size_t)std::max<long long>(E, 1));
  std::vector<float> eval((size_t)std::max<long long>(E, 1));
  {
    long long* d_pre = nullptr;
    int* d_ecol = nullptr;
    float* d_eval = nullptr;
    auto drop = [&]() {
      (void)hipFree(d_pre);
      (void)hipFree(d_ecol);
      (void)hipFree(d_eval);
    };
    hipError_t e = hipMalloc(&d_pre, sizeof(long long) * pre.size());
    if (e == hipSuccess) e = hipMalloc(&d_ecol, sizeof(int) * ecol.size());
    if (e == hipSuccess) e = hipMalloc(&d_eval, sizeof(float) * eval.size());
    if (e == hipSuccess) e = hipMemcpyAsync(d_pre, pre.data(), sizeof(long long) * pre.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      const int blocks = (int)std::max<long long>(1, std::min<long long>((N + 3) / 4, (long long)c->n_cu * 8));
      hipLaunchKernelGGL(dsgd_rows_gather_kernel, dim3(blocks), dim3(256), 0, c->stream, view(c), p->d_idx, N, d_pre, d_ecol, d_eval);
      e = hipGetLastError();
    }
    if (e == hipSuccess && E > 0) e = hipMemcpyAsync(ecol.data(), d_ecol, sizeof(int) * (size_t)E, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && E > 0) e = hipMemcpyAsync(eval.data(), d_eval, sizeof(float) * (size_t)E, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    drop();
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return 1;
    }
  }
  // pass 1: slots and distinct columns per (slice, step) -> the strides of the layout
  long long max_slots = 1, max_rows = 1, max_cols = 1;
  p->cs_shift.assign((size_t)n_steps, 21);
  {
    std::vector<int> cnt((size_t)G);
    std::vector<long long> seen((size_t)c->dp, -1);   // the last step that touched the rank
    for (long long s = 0; s < n_steps; ++s) {
      std::vector<long long> slots((size_t)G, 0), cols((size_t)G, 0);
      long long worst_list = 1;
      for (int k = 0; k < K; ++k) worst_list = std::max(worst_list, p->offsets[(size_t)(s * K + k) + 1] - p->offsets[(size_t)(s * K + k)]);
      for (long long t = p->offsets[(size_t)(s * K)]; t < p->offsets[(size_t)((s + 1) * K)]; ++t) {
        std::fill(cnt.begin(), cnt.end(), 0);
        for (long long e = pre[(size_t)t]; e < pre[(size_t)t + 1]; ++e) {
          const int rank = ecol[(size_t)e];
          ++cnt[(size_t)(rank % G)];
          if (seen[(size_t)rank] != s) {
            seen[(size_t)rank] = s;
            ++cols[(size_t)(rank % G)];
          }
        }
        for (int b = 0; b < G; ++b) slots[(size_t)b] += (cnt[(size_t)b] + CS_L - 1) / CS_L;
      }
      for (int b = 0; b < G; ++b) {
        max_slots = std::max(max_slots, slots[(size_t)b]);
        max_cols = std::max(max_cols, cols[(size_t)b]);
      }
      max_rows = std::max(max_rows, p->offsets[(size_t)((s + 1) * K)] - p->offsets[(size_t)(s * K)]);
      int bits = 0;
      while ((1LL << bits) < worst_list) ++bits;
      p->cs_shift[(size_t)s] = 30 - bits;   // at most one contribution per row and column: a worker's sums stay below 2^30
    }
  }
  if (max_slots > CS_MAX_SLOTS || max_rows > CS_MAX_SLOTS || max_cols > (long long)CS_MAX_CLT * CS_THREADS) return DSGD_OK;
  const long long big = std::max(max_slots, max_rows);
  const bool one_fits = big <= CS_THREADS && max_cols <= 4 * CS_THREADS;                  // one slot per lane
  const bool narrow_fits = big <= 2 * CS_THREADS_NARROW && max_cols <= 8 * CS_THREADS_NARROW;
  const int slot_stride = (int)((max_slots + 63) / 64 * 64), row_stride = (int)((max_rows + 1 + 63) / 64 * 64);
  const int cl_stride = (int)((max_cols + CS_THREADS - 1) / CS_THREADS * CS_THREADS);
  const long long cells = (long long)G * n_steps;
  const long long bytes = cells * ((long long)slot_stride * (4 + CS_L * 2 + CS_L * 4) + (long long)row_stride * 2 + (long long)cl_stride * 2 + 8);
  if (bytes > c->cs_max_mb * (1LL << 20)) return DSGD_OK;
  // pass 2: the layout
  std::vector<CsHdr> hdr((size_t)cells);
  std::vector<unsigned int> meta((size_t)(cells * slot_stride), 0u);
  std::vector<unsigned short> rf((size_t)(cells * row_stride), (unsigned short)0);
  std::vector<unsigned short> col((size_t)(cells * slot_stride * CS_L), (unsigned short)0);   // [cell][2 pieces][slot][8]
  std::vector<float> val((size_t)(cells * slot_stride * CS_L), 0.0f);                         // [cell][4 pieces][slot][4]
  std::vector<unsigned short> clist((size_t)(cells * cl_stride), (unsigned short)0xffffu);
  {
    std::vector<std::vector<std::pair<unsigned short, float>>> bucket((size_t)G);
    std::vector<std::vector<unsigned short>> touched((size_t)G);
    std::vector<long long> seen((size_t)c->dp, -1);
    for (long long s = 0; s < n_steps; ++s) {
      std::vector<int> cur((size_t)G, 0);
      for (auto& v : touched) v.clear();
      int r = 0;
      for (int k = 0; k < K; ++k) {
        for (long long t = p->offsets[(size_t)(s * K + k)]; t < p->offsets[(size_t)(s * K + k) + 1]; ++t, ++r) {
          for (auto& v : bucket) v.clear();
          for (long long e = pre[(size_t)t]; e < pre[(size_t)t + 1]; ++e) {
            const int rank = ecol[(size_t)e];
            bucket[(size_t)(rank % G)].emplace_back((unsigned short)(rank / G), eval[(size_t)e]);
            if (seen[(size_t)rank] != s) {
              seen[(size_t)rank] = s;
              touched[(size_t)(rank % G)].push_back((unsigned short)(rank / G));
            }
          }
          const unsigned short ypos = c->h_label[(size_t)p->h_idx[(size_t)t]] > 0 ? (unsigned short)0x8000u : (unsigned short)0;
          for (int b = 0; b < G; ++b) {
            const long long cell = (long long)b * n_steps + s;
            rf[(size_t)(cell * row_stride + r)] = (unsigned short)(cur[(size_t)b] | ypos);
            const auto& bk = bucket[(size_t)b];
            for (size_t j0 = 0; j0 < bk.size(); j0 += CS_L) {
              const long long slot = cell * slot_stride + cur[(size_t)b];
              meta[(size_t)slot] = (unsigned int)r | ((unsigned int)k << 16);
              const long long sl = cur[(size_t)b];
              for (size_t j = j0; j < std::min(bk.size(), j0 + CS_L); ++j) {
                const long long q = (long long)(j - j0);   // entry q of the slot: piece q / 8 of its columns, q / 4 of its values
                col[(size_t)((((cell * 2 + q / 8) * slot_stride) + sl) * 8 + q % 8)] = bk[j].first;
                val[(size_t)((((cell * 4 + q / 4) * slot_stride) + sl) * 4 + q % 4)] = bk[j].second;
              }
              ++cur[(size_t)b];
            }
          }
        }
      }
      for (int b = 0; b < G; ++b) {
        const long long cell = (long long)b * n_steps + s;
        rf[(size_t)(cell * row_stride + r)] = (unsigned short)cur[(size_t)b];   // the sentinel: one past the last row's slots
        hdr[(size_t)cell].counts = (unsigned int)cur[(size_t)b] | ((unsigned int)r << 16);
        hdr[(size_t)cell].shift = p->cs_shift[(size_t)s] | (int)((unsigned int)touched[(size_t)b].size() << 16);
        std::sort(touched[(size_t)b].begin(), touched[(size_t)b].end());
        std::copy(touched[(size_t)b].begin(), touched[(size_t)b].end(), clist.begin() + (size_t)(cell * cl_stride));
      }
    }
  }
  CS_SOFT(hipMalloc(&p->d_cs_hdr, sizeof(CsHdr) * hdr.size()));
  CS_SOFT(hipMalloc(&p->d_cs_meta, sizeof(unsigned int) * meta.size()));
  CS_SOFT(hipMalloc(&p->d_cs_rf, sizeof(unsigned short) * rf.size()));
  CS_SOFT(hipMalloc(&p->d_cs_col, sizeof(unsigned short) * col.size()));
  CS_SOFT(hipMalloc(&p->d_cs_val, sizeof(float) * val.size()));
  CS_SOFT(hipMalloc(&p->d_cs_cl, sizeof(unsigned short) * clist.size()));
  CS_SOFT(hipMemcpy(p->d_cs_cl, clist.data(), sizeof(unsigned short) * clist.size(), hipMemcpyHostToDevice));
  CS_SOFT(hipMemcpy(p->d_cs_hdr, hdr.data(), sizeof(CsHdr) * hdr.size(), hipMemcpyHostToDevice));
  CS_SOFT(hipMemcpy(p->d_cs_meta, meta.data(), sizeof(unsigned int) * meta.size(), hipMemcpyHostToDevice));
  CS_SOFT(hipMemcpy(p->d_cs_rf, rf.data(), sizeof(unsigned short) * rf.size(), hipMemcpyHostToDevice));
  CS_SOFT(hipMemcpy(p->d_cs_col, col.data(), sizeof(unsigned short) * col.size(), hipMemcpyHostToDevice));
  CS_SOFT(hipMemcpy(p->d_cs_val, val.data(), sizeof(float) * val.size(), hipMemcpyHostToDevice));
  if (ensure_cs_exchange(c)) return 1;
  p->cs_G = G;
  p->cs_nt = (c->cs_nt == CS_THREADS_NARROW && narrow_fits) ? CS_THREADS_NARROW : CS_THREADS;
  p->cs_spl = (p->cs_nt == CS_THREADS && one_fits) ? 1 : 2;
  p->cs_slot_stride = slot_stride;
  p->cs_row_stride = row_stride;
  p->cs_cl_stride = cl_stride;
  p->cs_ok = true;
  return DSGD_OK;
}
#undef CS_SOFT
static int cs_build(dsgd_ctx* c, dsgd_plan* p) {
  int rc;
  try {
    rc = c->cs_host_layout ? cs_build_impl(c, p) : cs_build_device_impl(c, p);
  } catch (const std::bad_alloc&) {   // (nothing may unwind across the C ABI)
    rc = 1;
  }
  if (rc == 1) {
    cs_free(c, p);   // (cs_layout is stamped: the plan keeps the row-parallel kernels until the layout changes)
    return DSGD_OK;
  }
  return rc;
}

// the weights slice-major for G slices (they stay so until something else touches w: bind)
static int cs_ensure_sliced(dsgd_ctx* c, int G) {
  if (c->cs_w_G == G) return DSGD_OK;
  DSGD_TRY(cs_unslice(c));
  const int Sp = cs_sp(c->dp, G), n = G * Sp;
  const size_t cap = sizeof(float) * (size_t)(c->dp + (CS_MAX_G + 1) * 8);
  if (!c->d_cs_w) HIP_TRY(hipMalloc(&c->d_cs_w, cap));
  if (!c->d_cs_ds) HIP_TRY(hipMalloc(&c->d_cs_ds, cap));
  hipLaunchKernelGGL(dsgd_cs_slice_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_w, c->d_cs_w, c->dp, G, Sp);
  hipLaunchKernelGGL(dsgd_cs_slice_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->d_ds, c->d_cs_ds, c->dp, G, Sp);
  HIP_TRY(hipGetLastError());
  c->cs_w_G = G;   // (dimSparsity can only change through an entry point that converts back first)
  return DSGD_OK;
}
// a step's granules carry the count of column-slice steps the context has launched: nothing to clear between launches
// (until the 32-bit count would wrap)
static int cs_take_tags(dsgd_ctx* c, unsigned long long n_launch, unsigned int* tag0) {
  if ((unsigned long long)c->cs_tag0 + n_launch + 1ull >= (1ull << 32)) {
    HIP_TRY(hipMemsetAsync(c->d_cs_x, 0, sizeof(unsigned long long) * 2 * CS_MAX_G * CS_XSTRIDE, c->stream));
    c->cs_tag0 = 0;
  }
  *tag0 = c->cs_tag0;
  c->cs_tag0 += (unsigned int)n_launch;
  return DSGD_OK;
}
static void cs_common_args(dsgd_ctx* c, CsArgs& a, int G, int K, float lr) {
  a.w = c->d_cs_w;
  a.ds = c->d_cs_ds;
  a.xbuf = c->d_cs_x;
  a.sync = c->d_cs_sync;
  a.sc = c->d_sc;
  a.tprof = c->d_tprof;
  a.lr = lr;
  a.lambda = (float)c->cfg.lambda;
  a.vexp = c->vexp;
  a.dp = c->dp;
  a.G = G;
  a.K = K;
  a.mail = nullptr;
  a.mail_seq = 0ull;
  a.gate_rec = nullptr;
  a.s_rec = nullptr;
  a.gate_words = 0;
  a.test_skip_publish = c->cs_test_skip;
}
static void cs_after_launch(dsgd_ctx* c, int shift) {
  c->last_grad_kernel = "dsgd_cs_step_kernel";
  c->last_shift = shift;
  c->fused_apply_pending = false;
  c->s_dirty = true;    // the kernel carries s = 2 lambda (w . ds) itself; whoever needs it next recomputes it from the weights
  c->s_lazy = false;
  c->nsq_dirty = false;
}

// the steps [step_begin, step_end) of a plan with column slices: ONE launch
static int launch_cs(dsgd_ctx* c, dsgd_plan* p, long long step_begin, long long step_end, float lr) {
  CsArgs a;
  a.hdr = p->d_cs_hdr;
  a.slot_meta = p->d_cs_meta;
  a.row_first = p->d_cs_rf;
  a.col = p->d_cs_col;
  a.val = p->d_cs_val;
  a.clist = p->d_cs_cl;
  a.cl_stride = p->cs_cl_stride;
  DSGD_TRY(cs_ensure_sliced(c, p->cs_G));
  cs_common_args(c, a, p->cs_G, p->n_workers, lr);
  a.n_steps_plan = p->n_steps;
  a.step_begin = step_begin;
  a.step_end = step_end;
  a.slot_stride = p->cs_slot_stride;
  a.row_stride = p->cs_row_stride;
  a.gate_rec = p->d_gate_rec;
  a.s_rec = p->d_s_rec;
  a.gate_words = p->gate_words;
  const size_t lds = sizeof(float) * (size_t)cs_lds_words(c->dp, a.G, a.K);
  DSGD_TRY(cs_take_tags(c, (unsigned long long)(step_end - step_begin), &a.tag0));
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  if (p->cs_nt == CS_THREADS_NARROW)
    hipLaunchKernelGGL((dsgd_cs_step_kernel<CS_THREADS_NARROW, 2, 8>), dim3((unsigned)a.G), dim3(CS_THREADS_NARROW), lds, c->stream, a);
  else if (p->cs_spl == 1)
    hipLaunchKernelGGL((dsgd_cs_step_kernel<CS_THREADS, 1, 4>), dim3((unsigned)a.G), dim3(CS_THREADS), lds, c->stream, a);
  else
    hipLaunchKernelGGL((dsgd_cs_step_kernel<CS_THREADS, 2, 8>), dim3((unsigned)a.G), dim3(CS_THREADS), lds, c->stream, a);
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  cs_after_launch(c, p->cs_shift[(size_t)(step_end - 1)]);
  return DSGD_OK;
}

// A per-request step of the reference's sizes (<= CS_MAX_K hosted workers, <= CS_MAX_SLOTS rows): the lists staged by
// stage_lists (c->cur_idx, c->d_segs) go through dsgd_cs_request_kernel -- every slice lays its cell out and runs the step,
// ONE launch.  Returns 1 when the path is not available (the caller takes the row-parallel kernels).
static int cs_request_ok(dsgd_ctx* c, const int32_t* const* idx_per_worker, const int64_t* n_per_worker, int K, int* G_out) {
  if (!c->cs_enable || !c->cs_req || c->comm || c->prof || K > CS_MAX_K) return 1;
  long long tot = 0;
  for (int k = 0; k < K; ++k) {
    if (n_per_worker[k] <= 0 || !idx_per_worker[k]) return 1;   // (stage_lists reports it)
    tot += n_per_worker[k];
  }
  if (tot > CS_MAX_SLOTS) return 1;
  for (int k = 0; k < K; ++k)
    for (int64_t t = 0; t < n_per_worker[k]; ++t)
      if (idx_per_worker[k][t] < 0 || idx_per_worker[k][t] >= c->n_rows) return 1;   // (the row-wise kernel reports the bad index)
  const int G = cs_pick_G(c, K, true);
  if (!G) return 1;
  *G_out = G;
  return DSGD_OK;
}
static int launch_cs_request(dsgd_ctx* c, int G, int K, long long worst_list, float lr) {
  dsgd_ctx::ReqLayout& L = c->req_layout;
  constexpr int SLOT_STRIDE = CS_MAX_SLOTS, ROW_STRIDE = CS_MAX_SLOTS + 64, CL_STRIDE = CS_MAX_CLT * CS_THREADS;
  if (L.G < G) {   // (grown once: 8 -> 16 slices)
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(L.hdr);
    (void)hipFree(L.meta);
    (void)hipFree(L.rf);
    (void)hipFree(L.col);
    (void)hipFree(L.val);
    (void)hipFree(L.cl);
    L = dsgd_ctx::ReqLayout();
    HIP_TRY(hipMalloc(&L.hdr, sizeof(CsHdr) * (size_t)G));
    HIP_TRY(hipMalloc(&L.meta, sizeof(unsigned int) * (size_t)G * SLOT_STRIDE));
    HIP_TRY(hipMalloc(&L.rf, sizeof(unsigned short) * (size_t)G * ROW_STRIDE));
    HIP_TRY(hipMalloc(&L.col, sizeof(unsigned short) * (size_t)G * SLOT_STRIDE * CS_L));
    HIP_TRY(hipMalloc(&L.val, sizeof(float) * (size_t)G * SLOT_STRIDE * CS_L));
    HIP_TRY(hipMalloc(&L.cl, sizeof(unsigned short) * (size_t)G * CL_STRIDE));
    L.G = G;
  }
  if (ensure_cs_exchange(c)) return fail(DSGD_ENOMEM, "out of device memory (column-slice exchange buffer)");
  DSGD_TRY(ensure_cs_max(c));
  DSGD_TRY(cs_ensure_sliced(c, G));
  CsArgs a;
  a.hdr = L.hdr;
  a.slot_meta = L.meta;
  a.row_first = L.rf;
  a.col = reinterpret_cast<const uint4*>(L.col);
  a.val = reinterpret_cast<const float4*>(L.val);
  a.clist = L.cl;
  a.cl_stride = CL_STRIDE;
  cs_common_args(c, a, G, K, lr);
  a.n_steps_plan = 1;
  a.step_begin = 0;
  a.step_end = 1;
  a.slot_stride = SLOT_STRIDE;
  a.row_stride = ROW_STRIDE;
  a.mail = c->d_mail;
  a.mail_seq = ++c->mail_seq;
  CsBuildArgs ba{};
  ba.m = view(c);
  ba.idx = c->cur_idx;
  ba.segs = c->d_segs;
  ba.hdr = L.hdr;
  ba.slot_meta = L.meta;
  ba.row_first = L.rf;
  ba.col = L.col;
  ba.val = L.val;
  ba.clist = L.cl;
  ba.maxima = c->d_cs_max;   // (a request only ever raises flags here; nobody reads the maxima)
  ba.n_steps_plan = 1;
  ba.slot_stride = SLOT_STRIDE;
  ba.row_stride = ROW_STRIDE;
  ba.cl_stride = CL_STRIDE;
  ba.dp = c->dp;
  ba.G = G;
  ba.K = K;
  const size_t lds = sizeof(float) * (size_t)cs_req_lds_words(c->dp, G, K);
  DSGD_TRY(cs_take_tags(c, 1ull, &a.tag0));
  c->ctr_known = false;
  hipLaunchKernelGGL(dsgd_cs_request_kernel, dim3((unsigned)G), dim3(CS_THREADS), lds, c->stream, a, ba);
  HIP_TRY(hipGetLastError());
  int bits = 0;
  while ((1LL << bits) < worst_list) ++bits;
  cs_after_launch(c, 30 - bits);
  c->last_grad_kernel = "dsgd_cs_request_kernel";
  return DSGD_OK;
}

// gradient of n_workers index lists (or row ranges too small for the streaming kernels) living at d_segs (device);
// max_items = largest list
static int launch_grad(dsgd_ctx* c, const int* d_idx, const WorkSeg* d_segs, int n_workers, long long max_items,
                       bool allow_fused = false) {
  return launch_grad_mb(c, d_idx, d_segs, n_workers, max_items, allow_fused);
}

// regularise each hosted worker's sum, aggregate (locally and across ranks), update w -- in three parts, so that ONE host
// thread can drive several contexts (dsgd_sync_step_devices): every context's kernels in front of the collective are
// enqueued first, then all the all-reduces inside one ncclGroup, then the updates behind them.
//   finish_pre         without peers: the whole fused reduce + regularise + mean + update (nothing else follows);
//                      with peers: exact column sums + regulariser + sum over the hosted workers -> d_gsum
//   finish_collective  the synchronous master's Future.sequence + Vec.mean (ref: core/Master.scala:190-194) as ONE all-reduce
//                      of D+1 floats over xGMI, ordered on the same stream as the kernels around it
//   finish_post        w <- w - lr * sum / (workers x world), the next regulariser scalar
static int finish_pre(dsgd_ctx* c, int n_workers, float lr, bool mail) {
  const int dp = c->dp;
  const int cblocks = (dp + FRA_COLS - 1) / FRA_COLS;
  if (!c->fused_apply_pending) return fail(DSGD_ESTATE, "internal: no gradient partials pending");
  c->fused_apply_pending = false;
  const FusedArgs& f = c->fused_args;
  if (c->s_dirty) return fail(DSGD_ESTATE, "internal: regulariser scalar not prepared");
  if (!c->comm) {
    hipLaunchKernelGGL(dsgd_fix_reduce_apply_kernel<true>, dim3(cblocks), dim3(1024), 0, c->stream, c->d_g64, (long long)dp,
                       n_workers, c->d_w, c->d_ds, dp, f.hg, c->d_part, c->part_stride, f.n_wg, f.hc, f.nc, c->d_partc, c->partc_stride, f.n_wgc,
                       f.inv_scale, f.inv_scale_cold, lr, (float)c->cfg.lambda, c->d_sc, red_out(c), (float*)nullptr,
                       (const float*)red_cur(c), c->s_lazy ? 1 : 0, mail ? c->d_mail : (unsigned long long*)nullptr);
    HIP_TRY(hipGetLastError());
    c->red_par ^= 1;
    c->s_lazy = true;   // the new pairs stay in d_redpart: the next step adds them itself, anyone else asks ensure_s
    c->s_dirty = false;
    return DSGD_OK;
  }
  hipLaunchKernelGGL(dsgd_fix_reduce_apply_kernel<false>, dim3(cblocks), dim3(1024), 0, c->stream, c->d_g64, (long long)dp,
                     n_workers, c->d_w, c->d_ds, dp, f.hg, c->d_part, c->part_stride, f.n_wg, f.hc, f.nc, c->d_partc, c->partc_stride, f.n_wgc,
                     f.inv_scale, f.inv_scale_cold, lr, (float)c->cfg.lambda, c->d_sc, red_out(c), c->d_gsum,
                     (const float*)red_cur(c), c->s_lazy ? 1 : 0, (unsigned long long*)nullptr);
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}
static int finish_collective(dsgd_ctx* c) {
  if (!c->comm) return DSGD_OK;
  RCCL_TRY(rccl::AllReduce(c->d_gsum, c->d_gsum, (size_t)c->dp, rccl::kFloat32, rccl::kSum, c->comm, c->stream));
  return DSGD_OK;
}
static int finish_post(dsgd_ctx* c, int n_workers, float lr) {
  if (!c->comm) return DSGD_OK;
  const int cblocks = (c->dp + FRA_COLS - 1) / FRA_COLS;
  const float k_total = (float)n_workers * (float)c->world;
  hipLaunchKernelGGL(dsgd_apply_cols_kernel, dim3(cblocks), dim3(256), 0, c->stream, c->d_w, c->d_gsum, c->d_ds, c->dp, k_total, lr,
                     (float)c->cfg.lambda, c->d_sc, red_out(c), 0);
  HIP_TRY(hipGetLastError());
  c->red_par ^= 1;
  c->s_lazy = true;
  c->s_dirty = false;
  return DSGD_OK;
}
static int launch_finish_sync(dsgd_ctx* c, int n_workers, float lr, bool mail = false) {
  DSGD_TRY(finish_pre(c, n_workers, lr, mail));
  DSGD_TRY(finish_collective(c));
  return finish_post(c, n_workers, lr);
}

// ---- column layout -------------------------------------------------------------------------------------
static int launch_permute_in(dsgd_ctx* c, const float* d_in, float* d_out) {
  hipLaunchKernelGGL(dsgd_permute_in_kernel, dim3((c->dp + 255) / 256), dim3(256), 0, c->stream, d_in, d_out, c->d_perm,
                     c->dp);
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}
static int launch_permute_out(dsgd_ctx* c, const float* d_in, float* d_out) {
  hipLaunchKernelGGL(dsgd_permute_out_kernel, dim3((c->dp + 255) / 256), dim3(256), 0, c->stream, d_in, d_out, c->d_perm,
                     c->dp);
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}
static int set_identity_perm(dsgd_ctx* c) {
  std::vector<int> id(c->dp);
  for (int j = 0; j < c->dp; ++j) id[j] = j;
  HIP_TRY(hipMemcpy(c->d_perm, id.data(), sizeof(int) * c->dp, hipMemcpyHostToDevice));
  return DSGD_OK;
}
static int count_columns(dsgd_ctx* c, long long nnz, unsigned int* d_cnt) {
  HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * c->dp, c->stream));
  if (nnz > 0) {
    const int hcnt = std::min(c->dp, DSGD_LDS_FLOATS);
    const int blocks = (int)std::max<long long>(1, std::min<long long>((nnz + 4095) / 4096, c->n_cu));
    hipLaunchKernelGGL(dsgd_colcount_kernel, dim3(blocks), dim3(1024), sizeof(unsigned int) * hcnt, c->stream, c->d_col,
                       c->d_val, nnz, d_cnt, c->dp, hcnt, c->d_sc);
    HIP_TRY(hipGetLastError());
  }
  return DSGD_OK;
}

// ---- wave tiles ------------------------------------------------------------------------------------------
// Tiles of WHOLE rows of the hot stream with <= WS_MAXNNZ non-zeros and <= WS_MAXROWS rows.  Lane l owns the 8
// contiguous slots [8l, 8l+8) of the 512-slot window at pos0 (a multiple of 8 slots: one 16-byte load of eight 16-bit
// column ranks per lane).  Lane descriptor (16 bits): row-start bits | label sign of the row ENDING at each start << 8
// (the kernel derives each lane's first row with a wave prefix sum over the start bits).  Rows of length 0 in
// `row_ptr` are skipped: the caller emptied them and put them on the long-row list.
struct HostTiles {
  std::vector<WTile> wt;
  std::vector<int> r0;              // first row of every tile + sentinel n_rows
  std::vector<unsigned short> meta16;   // n_tiles x 64
};
// wave tiles of WHOLE rows over the rows [rb, re) of a stream given by its slot offsets (a window of <= WS_SLOTS slots that
// starts at a multiple of 8; rows without a slot or longer than a tile close the open tile and stay out), appended to wt
static void append_wave_tiles(const long long* row_ptr, long long rb, long long re, std::vector<WTile>& wt, int max_rows) {
  const long long amask = ~7LL;
  long long start = -1;  // first row of the open tile
  auto close = [&](long long end_row) {
    if (start < 0) return;
    WTile t;
    t.pos0 = row_ptr[start] & amask;
    t.r0 = (int)start;
    t.info = (int)(((end_row - start) & 0xffff) | ((row_ptr[end_row] - t.pos0) << 16));
    wt.push_back(t);
    start = -1;
  };
  for (long long i = rb; i < re; ++i) {
    const long long len = row_ptr[i + 1] - row_ptr[i];
    if (len > WS_MAXNNZ || len == 0) {
      close(i);
      continue;
    }
    if (start >= 0 && (row_ptr[i + 1] - (row_ptr[start] & amask) > WS_SLOTS - 1 || i - start >= max_rows)) close(i);
    if (start < 0) start = i;
  }
  close(re);
}
// the 64 lane descriptors of every tile: row-start bits of the lane's eight slots | label signs of the rows ending there << 8
static void wave_tile_meta(const long long* row_ptr, const signed char* label, const std::vector<WTile>& wt,
                           std::vector<unsigned short>& meta16) {
  const long long n_tiles = (long long)wt.size();
  meta16.assign((size_t)std::max<long long>(n_tiles, 1) * 64, (unsigned short)0);
  for (long long t = 0; t < n_tiles; ++t) {
    const WTile& T = wt[(size_t)t];
    const int t_nrows = (int)(short)(T.info & 0xffff);
    int cur_row = 0, next = 0;
    for (int l = 0; l < 64; ++l) {  // lane l owns slots [8l, 8l+8)
      unsigned int bits = 0, ys = 0;
      for (int k = 0; k < 8; ++k) {
        const long long slot = 8LL * l + k;
        bool st = false;
        if (next < t_nrows && row_ptr[T.r0 + next] - T.pos0 == slot) st = true;
        else if (next == t_nrows && row_ptr[T.r0 + t_nrows] - T.pos0 == slot) st = true;  // end mark
        if (st) {
          bits |= 1u << k;
          // the row that ends here is local row cur_row (1-based); row 0 = leading padding
          if (cur_row >= 1 && label[T.r0 + cur_row - 1] > 0) ys |= 1u << k;
          ++cur_row;
          ++next;
        }
      }
      meta16[(size_t)t * 64 + l] = (unsigned short)(bits | (ys << 8));
    }
  }
}
static void build_wave_tiles(const long long* row_ptr, long long n_rows, const signed char* label, HostTiles& out,
                             int max_rows = WS_MAXROWS) {
  std::vector<WTile>& wt = out.wt;
  std::vector<int>& wr0 = out.r0;
  wt.clear();
  wr0.clear();
  append_wave_tiles(row_ptr, 0, n_rows, wt, max_rows);
  for (const WTile& t : wt) wr0.push_back(t.r0);
  wr0.push_back((int)n_rows);
  wave_tile_meta(row_ptr, label, wt, out.meta16);
  if (wt.empty()) {
    WTile t;
    t.pos0 = 0;
    t.r0 = 0;
    t.info = 0xffff;   // -1 rows, 0 slots
    wt.push_back(t);
  }
}
static int upload_wave_tiles(dsgd_ctx* c, HostTiles& ht) {
  (void)hipFree(c->d_wtiles);
  (void)hipFree(c->d_wmeta);
  (void)hipFree(c->d_wlong_rows);
  c->d_wtiles = nullptr;
  c->d_wmeta = nullptr;
  c->d_wlong_rows = nullptr;
  c->n_wtiles = (long long)ht.r0.size() - 1;
  c->h_wtile_r0.swap(ht.r0);
  HIP_TRY(hipMalloc(&c->d_wtiles, sizeof(WTile) * ht.wt.size()));
  const size_t meta_bytes = sizeof(unsigned short) * ht.meta16.size();
  HIP_TRY(hipMalloc(&c->d_wmeta, meta_bytes));
  HIP_TRY(hipMemcpy(c->d_wtiles, ht.wt.data(), sizeof(WTile) * ht.wt.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_wmeta, ht.meta16.data(), meta_bytes, hipMemcpyHostToDevice));
  c->bound_segs.clear();
  std::vector<int> lr(c->wlong_rows.begin(), c->wlong_rows.end());
  lr.push_back(0);
  HIP_TRY(hipMalloc(&c->d_wlong_rows, sizeof(int) * lr.size()));
  HIP_TRY(hipMemcpy(c->d_wlong_rows, lr.data(), sizeof(int) * lr.size(), hipMemcpyHostToDevice));
  c->ssegs_last.clear();
  return DSGD_OK;
}

// split the ranked CSR into the hot stream (rank < hsplit) and the cold stream (rank - hsplit), both in row order with
// wave tiles of whole rows; rows whose hot or cold part exceeds a wave tile stay on the long-row list and in neither.
static void fstep_drop_all(dsgd_ctx* c);
static void tcol_drop_all(dsgd_ctx* c);
static int build_split(dsgd_ctx* c) {
  fstep_drop_all(c);
  tcol_drop_all(c);   // (the chunked tile tables index the streams built here; the column lists are sorted by the ranks about to change)
  (void)hipFree(c->d_hcol); (void)hipFree(c->d_hval); (void)hipFree(c->d_hrow_ptr);
  (void)hipFree(c->d_ccol); (void)hipFree(c->d_cval); (void)hipFree(c->d_ctp); (void)hipFree(c->d_ctiles); (void)hipFree(c->d_cmeta);
  (void)hipFree(c->d_dcold); (void)hipFree(c->d_coef8);
  c->d_hcol = nullptr; c->d_hval = nullptr; c->d_hrow_ptr = nullptr;
  c->d_ccol = nullptr; c->d_cval = nullptr; c->d_ctp = nullptr; c->d_ctiles = nullptr; c->d_cmeta = nullptr;
  c->d_dcold = nullptr; c->d_coef8 = nullptr;
  const long long n_rows = c->n_rows;
  const int H = std::min(c->hsplit, c->dp);
  HIP_TRY(hipMalloc(&c->d_coef8, (size_t)std::max<long long>(n_rows, 1)));
  HIP_TRY(hipMemset(c->d_coef8, 0, (size_t)std::max<long long>(n_rows, 1)));
  HIP_TRY(hipMalloc(&c->d_dcold, sizeof(float) * (size_t)std::max<long long>(n_rows, 1)));
  HIP_TRY(hipMemset(c->d_dcold, 0, sizeof(float) * (size_t)std::max<long long>(n_rows, 1)));
  // cold entries per row
  int* d_cnt = nullptr;
  HIP_TRY(hipMalloc(&d_cnt, sizeof(int) * (size_t)std::max<long long>(n_rows, 1)));
  CsrView m = view(c);
  {
    const int blocks = (int)std::max<long long>(1, std::min<long long>((n_rows + 15) / 16, (long long)c->n_cu * 16));
    hipLaunchKernelGGL(dsgd_split_count_kernel<16>, dim3(blocks), dim3(256), 0, c->stream, m, H, d_cnt);
  }
  std::vector<int> cnt((size_t)n_rows);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t)n_rows, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_cnt);
  if (e != hipSuccess) return fail(DSGD_EHIP, "split counts: %s", hipGetErrorString(e));
  // hrp / ctp: slot offsets of the two streams (a tiled row owns at least one slot in each: an explicit zero when it
  // has no entry there); crp: the cold ENTRIES before each row (what dsgd_range_nnz reports)
  std::vector<long long>&hrp = c->h_hrp, &ctp = c->h_ctp, &crp = c->h_crow_ptr;
  hrp.assign((size_t)n_rows + 1, 0);
  ctp.assign((size_t)n_rows + 1, 0);
  ++c->layout_gen;
  c->layout_seen_by_build = false;
  crp.assign((size_t)n_rows + 1, 0);
  c->wlong_rows.clear();
  hrp[0] = 0;
  ctp[0] = 0;
  const std::vector<long long>& rp = c->h_row_ptr;
  const bool has_cold = c->dp > H;
  for (long long i = 0; i < n_rows; ++i) {
    const long long len = rp[i + 1] - rp[i], cold = cnt[(size_t)i], hot = len - cold;
    if (hot > WS_MAXNNZ || cold > WS_MAXNNZ) {   // served whole from the ranked CSR, one wave per row
      c->wlong_rows.push_back(i);
      hrp[i + 1] = hrp[i];
      ctp[i + 1] = ctp[i];
      crp[i + 1] = crp[i];
    } else {
      hrp[i + 1] = hrp[i] + std::max<long long>(hot, 1);
      ctp[i + 1] = ctp[i] + (has_cold ? std::max<long long>(cold, 1) : 0);
      crp[i + 1] = crp[i] + cold;
    }
  }
  // (what a long row weighs when row chunks are cut: its own non-zeros plus ~1,300 slots' worth of its wave's waiting --
  //  0.7 us per long row measured against 75 us for a chunk of ~200 K slots, profiles/r06_fstep_wg_times.txt)
  c->wlong_weight.assign(c->wlong_rows.size() + 1, 0);
  for (size_t k = 0; k < c->wlong_rows.size(); ++k) {
    const long long i = c->wlong_rows[k];
    c->wlong_weight[k + 1] = c->wlong_weight[k] + (rp[(size_t)i + 1] - rp[(size_t)i]) + 1300;
  }
  c->hot_nnz = hrp[n_rows];
  c->coldm_nnz = ctp[n_rows];
  HIP_TRY(hipMalloc(&c->d_hcol, sizeof(unsigned short) * (size_t)(c->hot_nnz + WS_PAD)));
  HIP_TRY(hipMalloc(&c->d_hval, sizeof(float) * (size_t)(c->hot_nnz + WS_PAD)));
  HIP_TRY(hipMemset(c->d_hcol + c->hot_nnz, 0, sizeof(unsigned short) * WS_PAD));
  HIP_TRY(hipMemset(c->d_hval + c->hot_nnz, 0, sizeof(float) * WS_PAD));
  HIP_TRY(hipMalloc(&c->d_hrow_ptr, sizeof(long long) * hrp.size()));
  HIP_TRY(hipMemcpy(c->d_hrow_ptr, hrp.data(), sizeof(long long) * hrp.size(), hipMemcpyHostToDevice));
  // 16-bit cold ids whenever there are at most 65536 cold columns (DSGD_COLD_UNPACKED=1 forces the 32-bit form: tests)
  c->cold_col16 = c->dp - H <= 65536 && !getenv("DSGD_COLD_UNPACKED");
  const size_t csz = c->cold_col16 ? sizeof(unsigned short) : sizeof(unsigned int);
  HIP_TRY(hipMalloc(&c->d_ccol, csz * (size_t)(c->coldm_nnz + WS_PAD)));
  HIP_TRY(hipMalloc(&c->d_cval, sizeof(float) * (size_t)(c->coldm_nnz + WS_PAD)));
  HIP_TRY(hipMemset((char*)c->d_ccol + csz * (size_t)c->coldm_nnz, 0, csz * WS_PAD));
  HIP_TRY(hipMemset(c->d_cval + c->coldm_nnz, 0, sizeof(float) * WS_PAD));
  HIP_TRY(hipMalloc(&c->d_ctp, sizeof(long long) * ctp.size()));
  HIP_TRY(hipMemcpy(c->d_ctp, ctp.data(), sizeof(long long) * ctp.size(), hipMemcpyHostToDevice));
  {
    const int blocks = (int)std::max<long long>(1, std::min<long long>((n_rows + 3) / 4, (long long)c->n_cu * 16));
    if (c->cold_col16)
      hipLaunchKernelGGL(dsgd_split_fill_kernel<true>, dim3(blocks), dim3(256), 0, c->stream, m, H, c->d_hrow_ptr, c->d_ctp,
                         c->d_hcol, c->d_hval, c->d_ccol, c->d_cval);
    else
      hipLaunchKernelGGL(dsgd_split_fill_kernel<false>, dim3(blocks), dim3(256), 0, c->stream, m, H, c->d_hrow_ptr, c->d_ctp,
                         c->d_hcol, c->d_hval, c->d_ccol, c->d_cval);
    HIP_TRY(hipGetLastError());
  }
  c->h_ccol.clear();
  if (c->cold_col16 && c->vt_enable && c->coldm_nnz > 0) {   // (75 MB for the 8.4 M-row shard)
    c->h_ccol.resize((size_t)c->coldm_nnz);
    HIP_TRY(hipMemcpyAsync(c->h_ccol.data(), c->d_ccol, sizeof(unsigned short) * (size_t)c->coldm_nnz, hipMemcpyDeviceToHost, c->stream));
  }
  HostTiles ht;
  build_wave_tiles(hrp.data(), n_rows, c->h_label.data(), ht);
  DSGD_TRY(upload_wave_tiles(c, ht));
  // the cold stream's own tiles (label signs unused there)
  HostTiles hc;
  build_wave_tiles(ctp.data(), n_rows, c->h_label.data(), hc, CT_MAXROWS);
  c->n_ctiles = (long long)hc.r0.size() - 1;
  c->h_ctile_r0.swap(hc.r0);
  HIP_TRY(hipMalloc(&c->d_ctiles, sizeof(WTile) * hc.wt.size()));
  HIP_TRY(hipMalloc(&c->d_cmeta, sizeof(unsigned short) * hc.meta16.size()));
  HIP_TRY(hipMemcpy(c->d_ctiles, hc.wt.data(), sizeof(WTile) * hc.wt.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_cmeta, hc.meta16.data(), sizeof(unsigned short) * hc.meta16.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return DSGD_OK;
}

// Rank the columns by how often they occur in the loaded rows (summed over ranks when a
// communicator is attached, so every replica uses the same order) and relabel the CSR columns.
// Runs once, lazily, at the first compute call after dsgd_load_csr.
// (in three parts -- the counts, their all-reduce, the ranking and the split -- so that one host thread can prepare several
//  contexts: dsgd_build_dim_sparsity_devices)
static int layout_begin(dsgd_ctx* c, unsigned int** d_cnt_out) {
  *d_cnt_out = nullptr;
  if (c->layout_ready) return DSGD_OK;
  unsigned int* d_cnt = nullptr;
  HIP_TRY(hipMalloc(&d_cnt, sizeof(unsigned int) * c->dp));
  const int rc = count_columns(c, c->nnz, d_cnt);
  if (rc) {
    (void)hipFree(d_cnt);
    return rc;
  }
  *d_cnt_out = d_cnt;
  return DSGD_OK;
}
static int layout_collective(dsgd_ctx* c, unsigned int* d_cnt) {
  if (!d_cnt || !c->comm) return DSGD_OK;
  int r = rccl::AllReduce(d_cnt, d_cnt, (size_t)c->dp, rccl::kUint32, rccl::kSum, c->comm, c->stream);
  if (r) return fail(DSGD_ERCCL, "ncclAllReduce(column counts): %s", rccl::GetErrorString(r));
  return DSGD_OK;
}
static int layout_finish(dsgd_ctx* c, unsigned int* d_cnt) {   // (takes ownership of d_cnt)
  if (!d_cnt) return DSGD_OK;
  std::vector<unsigned int> cnt(c->dp);
  int rc = DSGD_OK;
  {
    hipError_t e = hipMemcpyAsync(cnt.data(), d_cnt, sizeof(unsigned int) * c->dp, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(DSGD_EHIP, "column counts: %s", hipGetErrorString(e));
  }
  (void)hipFree(d_cnt);
  DSGD_TRY(rc);
  std::vector<int> order(c->dp);
  for (int j = 0; j < c->dp; ++j) order[j] = j;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cnt[a] > cnt[b]; });  // ties: ascending key
  std::vector<int> perm(c->dp);
  for (int r = 0; r < c->dp; ++r) perm[order[r]] = r;
  // bring the resident vectors (identity layout so far) into ranked order
  HIP_TRY(hipMemcpy(c->d_perm, perm.data(), sizeof(int) * c->dp, hipMemcpyHostToDevice));
  DSGD_TRY(launch_permute_in(c, c->d_w, c->d_tmp));
  HIP_TRY(hipMemcpyAsync(c->d_w, c->d_tmp, sizeof(float) * c->dp, hipMemcpyDeviceToDevice, c->stream));
  DSGD_TRY(launch_permute_in(c, c->d_ds, c->d_tmp));
  HIP_TRY(hipMemcpyAsync(c->d_ds, c->d_tmp, sizeof(float) * c->dp, hipMemcpyDeviceToDevice, c->stream));
  if (c->nnz > 0) {
    const int blocks = (int)std::min<long long>((c->nnz + 255) / 256, (long long)c->n_cu * 8);
    hipLaunchKernelGGL(dsgd_remap_cols_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_col, c->nnz, c->d_perm);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  DSGD_TRY(build_split(c));
  c->layout_ready = true;
  c->s_dirty = true;
  return DSGD_OK;
}
static int prepare_layout(dsgd_ctx* c) {
  if (c->layout_ready) return DSGD_OK;
  unsigned int* d_cnt = nullptr;
  DSGD_TRY(layout_begin(c, &d_cnt));
  const int rc = layout_collective(c, d_cnt);
  if (rc) {
    (void)hipFree(d_cnt);
    return rc;
  }
  return layout_finish(c, d_cnt);
}
// back to the identity layout (before new data is loaded): resident vectors return to key order
static int reset_layout(dsgd_ctx* c) {
  if (c->layout_ready) {
    DSGD_TRY(launch_permute_out(c, c->d_w, c->d_tmp));
    HIP_TRY(hipMemcpyAsync(c->d_w, c->d_tmp, sizeof(float) * c->dp, hipMemcpyDeviceToDevice, c->stream));
    DSGD_TRY(launch_permute_out(c, c->d_ds, c->d_tmp));
    HIP_TRY(hipMemcpyAsync(c->d_ds, c->d_tmp, sizeof(float) * c->dp, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  DSGD_TRY(set_identity_perm(c));
  c->layout_ready = false;
  c->s_dirty = true;
  return DSGD_OK;
}

// ---- nnz-streaming launches (contiguous row ranges) ----------------------------------------------------
static int upload_ssegs(dsgd_ctx* c, const std::vector<StreamSeg>& segs) {
  const int n = (int)segs.size();
  if (n > c->ssegs_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_ssegs);
    c->d_ssegs = nullptr;
    HIP_TRY(hipMalloc(&c->d_ssegs, sizeof(StreamSeg) * (size_t)n));
    c->ssegs_cap = n;
    c->ssegs_last.clear();
  }
  if ((int)c->ssegs_last.size() == n && memcmp(c->ssegs_last.data(), segs.data(), sizeof(StreamSeg) * n) == 0) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(c->d_ssegs, segs.data(), sizeof(StreamSeg) * n, hipMemcpyHostToDevice));
  c->ssegs_last = segs;
  return DSGD_OK;
}
static StreamSeg make_sseg(long long rb, long long re) {
  StreamSeg s{};
  s.row_begin = rb;
  s.row_end = re;
  return s;
}
// tile and long-row ranges of every worker's row range (wave tiles of whatever stream the mode uses)
static void locate_segs(dsgd_ctx* c, std::vector<StreamSeg>& segs, long long* max_tiles, long long* max_long,
                        long long* max_ctiles = nullptr) {
  const std::vector<int>& wr0 = c->h_wtile_r0;  // n_wtiles + 1 entries (sentinel n_rows)
  *max_tiles = 1;
  *max_long = 0;
  if (max_ctiles) *max_ctiles = 0;
  for (StreamSeg& s : segs) {
    // first tile whose rows reach row_begin: the last tile with r0 <= row_begin (it may end before row_begin when a
    // long row sits in between -- then all its rows are masked) ... one past the last tile with r0 < row_end
    long long tb = (std::upper_bound(wr0.begin(), wr0.end() - 1, (int)s.row_begin) - wr0.begin()) - 1;
    if (tb < 0) tb = 0;
    long long te = std::lower_bound(wr0.begin(), wr0.end() - 1, (int)s.row_end) - wr0.begin();
    if (te < tb) te = tb;
    s.tile_begin = tb;
    s.tile_end = te;
    *max_tiles = std::max(*max_tiles, te - tb);
    // the rows that fit no tile: a range of the sorted long-row list, handled by the same kernel (one wave per row)
    s.long_begin = std::lower_bound(c->wlong_rows.begin(), c->wlong_rows.end(), s.row_begin) - c->wlong_rows.begin();
    s.long_end = std::lower_bound(c->wlong_rows.begin(), c->wlong_rows.end(), s.row_end) - c->wlong_rows.begin();
    *max_long = std::max(*max_long, s.long_end - s.long_begin);
    // the cold stream's tiles of the range, located the same way
    const std::vector<int>& cr0 = c->h_ctile_r0;   // n_ctiles + 1 entries (sentinel n_rows)
    s.ctile_begin = s.ctile_end = 0;
    if (cr0.size() > 1) {
      long long cb = (std::upper_bound(cr0.begin(), cr0.end() - 1, (int)s.row_begin) - cr0.begin()) - 1;
      if (cb < 0) cb = 0;
      long long ce = std::lower_bound(cr0.begin(), cr0.end() - 1, (int)s.row_end) - cr0.begin();
      if (ce < cb) ce = cb;
      s.ctile_begin = cb;
      s.ctile_end = ce;
    }
    if (max_ctiles) *max_ctiles = std::max(*max_ctiles, s.ctile_end - s.ctile_begin);
  }
}
static int ensure_part(dsgd_ctx* c, int** buf, long long* wgs, int* stride, long long need_wgs, int need_cols) {
  if (need_wgs <= *wgs && need_cols <= *stride) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  (void)hipFree(*buf);
  *buf = nullptr;
  *wgs = std::max<long long>(need_wgs, c->n_cu);
  *stride = std::max(*stride, (need_cols + 63) / 64 * 64);
  HIP_TRY(hipMalloc(buf, sizeof(int) * (size_t)*wgs * (size_t)*stride));
  return DSGD_OK;
}

// whole row ranges: hot stream in wave tiles, cold stream before (x.w) and after (gradient) it
template <bool SCATTER>
static int launch_stream(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs) {
  const int n_workers = (int)row_segs.size();
  std::vector<StreamSeg> segs(row_segs);
  long long max_tiles, max_long, max_ctiles;
  locate_segs(c, segs, &max_tiles, &max_long, &max_ctiles);
  DSGD_TRY(upload_ssegs(c, segs));
  const int H = std::min(c->hsplit, c->dp);
  const int nc = c->dp - H;                                     // cold columns
  // ... of which in the LDS tile of the cold kernels (next to 16 strips and the gradient kernel's 64 dummy words)
  const int nc_lds = std::min(nc, DSGD_LDS_FLOATS - 16 * CT_STRIP - 64 - 4);
  const long long per_worker = std::max<long long>(1, c->n_cu / n_workers);
  dim3 gridc((unsigned)std::max<long long>(1, std::min(per_worker, (max_ctiles + 15) / 16)), n_workers);
  const bool cold = nc > 0 && max_ctiles > 0;
  const bool wide = nc > nc_lds;
  const size_t lds_cdot = sizeof(float) * (size_t)(((nc_lds + 3) & ~3) + 16 * CT_STRIP);
  const size_t lds_cgrad = sizeof(float) * (size_t)(((nc_lds + 64 + 3) & ~3) + 16 * CT_STRIP);
  if (cold) {
    size_t slot_c = 0;
    DSGD_TRY(prof_begin(c, &slot_c, 1));
#define DSGD_COLD_LAUNCH(C16, GR, WD, LDS)                                                                               \
  hipLaunchKernelGGL((dsgd_cold_kernel<C16, GR, WD>), gridc, dim3(1024), LDS, c->stream, c->d_ctiles, c->d_cmeta,      \
                     c->d_ccol, c->d_cval, c->d_ssegs, c->d_w, c->d_dcold, c->d_coef8, c->d_g64, (long long)c->dp, c->d_sc, \
                     H, nc_lds, c->fix_scale, c->d_partc, c->partc_stride)
#define DSGD_COLD_DISPATCH(GR, LDS)                                       \
  do {                                                                    \
    if (c->cold_col16 && !wide) DSGD_COLD_LAUNCH(true, GR, false, LDS);   \
    else if (c->cold_col16) DSGD_COLD_LAUNCH(true, GR, true, LDS);        \
    else if (!wide) DSGD_COLD_LAUNCH(false, GR, false, LDS);              \
    else DSGD_COLD_LAUNCH(false, GR, true, LDS);                          \
  } while (0)
    DSGD_COLD_DISPATCH(false, lds_cdot);
    HIP_TRY(hipGetLastError());
    DSGD_TRY(prof_end(c, slot_c));
  }
  long long bx = std::min(per_worker, std::max((max_tiles + 15) / 16, (max_long + 15) / 16));
  dim3 grid((unsigned)std::max<long long>(1, bx), n_workers);
  const int hw = H, hg = SCATTER ? H : 0;
  const size_t lds = sizeof(float) * (size_t)(16 * WS_COEF_STRIDE + hw + hg + (SCATTER ? 64 : 0) + 4);
  if (SCATTER) {
    DSGD_TRY(ensure_part(c, &c->d_part, &c->part_wgs, &c->part_stride, (long long)grid.x * grid.y, hg));
    if (cold) DSGD_TRY(ensure_part(c, &c->d_partc, &c->partc_wgs, &c->partc_stride, (long long)gridc.x * gridc.y, nc_lds));
  }
  // Fixed-point scale of THIS launch: a column receives at most one contribution per row, |contribution| <= 2^shift,
  // and workgroup b of a worker owns the 16-tile groups b, b + grid.x, ... of its tile range (plus its share of the
  // long rows): with rows(b) <= 2^bits, shift = 30 - bits keeps every 32-bit LDS sum below 2^30.  (Rows between the
  // first rows of two tiles bound the rows of a tile from above.)
  float main_scale = c->fix_scale;
  if (SCATTER) {
    const std::vector<int>& wr0 = c->h_wtile_r0;
    long long worst = 1;
    std::vector<long long> rows_of((size_t)grid.x);
    for (const StreamSeg& sg : segs) {
      std::fill(rows_of.begin(), rows_of.end(), 0);
      long long g = 0;
      for (long long t = sg.tile_begin; t < sg.tile_end; t += 16, ++g) {
        const long long t1 = std::min<long long>(t + 16, sg.tile_end);
        rows_of[(size_t)(g % grid.x)] += (long long)wr0[(size_t)t1] - (long long)wr0[(size_t)t];
      }
      const long long n_long = sg.long_end - sg.long_begin;
      const long long long_share = (n_long + 16LL * grid.x - 1) / (16LL * grid.x) * 16;
      for (long long v : rows_of) worst = std::max(worst, v + long_share);
    }
    int bits = 0;
    while ((1LL << bits) < worst) ++bits;
    const int shift0 = std::max(1, std::min(c->max_shift, 30 - bits));   // safe for ANY data: rows x largest value
    int shift = shift0;
    if (c->fix_bound && shift0 < c->max_shift) {
      // data-dependent refinement (dsgd_wseg_bound_kernel): measured once per (ranges, grid) configuration
      const bool hit = c->bound_grid == grid.x && c->bound_segs.size() == segs.size() &&
                       memcmp(c->bound_segs.data(), segs.data(), sizeof(StreamSeg) * segs.size()) == 0;
      if (!hit) {
        if (!c->d_bound) HIP_TRY(hipMalloc(&c->d_bound, sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->d_bound, 0, sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(dsgd_wseg_bound_kernel, grid, dim3(1024), sizeof(unsigned int) * (size_t)(hg + 16), c->stream,
                           c->d_hrow_ptr, c->d_hcol, c->d_hval, c->d_wtiles, c->n_wtiles, c->n_rows, view(c),
                           c->d_wlong_rows, c->d_ssegs, hg, std::ldexp(1.0f, shift0 - c->vexp), c->d_bound);
        HIP_TRY(hipGetLastError());
        unsigned int amax = 0;
        HIP_TRY(hipMemcpyAsync(&amax, c->d_bound, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        int s2 = shift0;
        const double room = (double)(1LL << 30) - (double)worst;
        while (s2 < c->max_shift && std::ldexp((double)amax, s2 + 1 - shift0) <= room) ++s2;
        c->bound_segs = segs;
        c->bound_grid = grid.x;
        c->bound_shift = s2;
      }
      shift = c->bound_shift;
    }
    main_scale = std::ldexp(1.0f, shift - c->vexp);
    c->last_shift = shift;
  }
  CsrView mh = view(c);
  mh.row_ptr = c->d_hrow_ptr;
  mh.col = reinterpret_cast<const int*>(c->d_hcol);   // 16-bit ranks; the kernel reinterprets the pointer
  mh.val = c->d_hval;
  CsrView mf = view(c);
  size_t slot = 0;
  if (SCATTER) DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  hipLaunchKernelGGL(dsgd_wseg_kernel<SCATTER>, grid, dim3(1024), lds, c->stream, mh, mf, c->d_wtiles, c->d_wmeta, c->d_w,
                     c->d_g64, (long long)c->dp, c->d_ssegs, c->d_sc, hw, hg, main_scale, c->d_coef8, c->d_wlong_rows,
                     SCATTER ? c->d_part : nullptr, c->part_stride, c->d_dcold, c->fix_scale);
  HIP_TRY(hipGetLastError());
  if (SCATTER) DSGD_TRY(prof_end(c, slot));
  if (!SCATTER) return DSGD_OK;
  c->last_grad_kernel = "dsgd_wseg_kernel<true>";
  if (cold) {
    size_t slot_g = 0;
    DSGD_TRY(prof_begin(c, &slot_g, 2));
    DSGD_COLD_DISPATCH(true, lds_cgrad);
    HIP_TRY(hipGetLastError());
    DSGD_TRY(prof_end(c, slot_g));
  }
  // the exact column sums of every hosted worker go straight into regularise + sum (+ mean + update + next s when
  // there are no peers): g itself is never materialised
  DSGD_TRY(ensure_redpart(c));
  c->fused_args = {hg, (int)grid.x, H, cold ? nc_lds : 0, (int)gridc.x, 1.0 / (double)main_scale, 1.0 / (double)c->fix_scale};
  c->fused_apply_pending = true;   // launched by launch_finish_sync, which knows lr
  return DSGD_OK;
}

// ---- row chunks: the whole gradient of a row range in one launch (csrc/dsgd_fstep.hpp) -------------------------------
static void fstep_free(dsgd_ctx::FstepLayout& L) {
  (void)hipFree(L.d_times);
  if (L.h_times) (void)hipHostFree(L.h_times);
  if (L.times_ev) (void)hipEventDestroy(L.times_ev);
  (void)hipFree(L.d_tiles);
  (void)hipFree(L.d_meta);
  (void)hipFree(L.d_ctiles);
  (void)hipFree(L.d_cmeta);
  (void)hipFree(L.d_chunks);
  L = dsgd_ctx::FstepLayout();
}
static void fstep_drop_all(dsgd_ctx* c) {   // (the split streams are about to change, or the context goes away)
  if (c->fstep_cache.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (dsgd_ctx::FstepLayout& L : c->fstep_cache) fstep_free(L);
  c->fstep_cache.clear();
}
// can the chunked launch serve this context's layout at all?  (16-bit cold ids, every cold column inside the LDS tile)
static bool fstep_possible(const dsgd_ctx* c) {
  const int H = std::min(c->hsplit, c->dp);
  const int nc = c->dp - H;
  const int nc_lds = std::min(nc, DSGD_LDS_FLOATS - 16 * CT_STRIP - 64 - 4);
  return c->fstep_enable && nc > 0 && nc <= nc_lds && c->cold_col16 && c->coldm_nnz > 0 && c->n_rows < (1LL << 31);
}
static int fstep_build(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, int n_wg, dsgd_ctx::FstepLayout& L);
static int fstep_layout(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, int n_wg, dsgd_ctx::FstepLayout** out) {
  std::vector<long long> key;
  for (const StreamSeg& sg : row_segs) {
    key.push_back(sg.row_begin);
    key.push_back(sg.row_end);
  }
  for (dsgd_ctx::FstepLayout& L : c->fstep_cache)
    if (L.n_wg == n_wg && L.ranges == key) {
      L.used = ++c->fstep_clock;
      *out = &L;
      return DSGD_OK;
    }
  if (c->fstep_cache.size() >= 8) {   // the least recently used configuration makes room
    size_t v = 0;
    for (size_t i = 1; i < c->fstep_cache.size(); ++i)
      if (c->fstep_cache[i].used < c->fstep_cache[v].used) v = i;
    HIP_TRY(hipStreamSynchronize(c->stream));
    fstep_free(c->fstep_cache[v]);
    c->fstep_cache.erase(c->fstep_cache.begin() + (long)v);
  }
  dsgd_ctx::FstepLayout L;
  L.ranges = key;
  L.n_wg = n_wg;
  DSGD_TRY(fstep_build(c, row_segs, n_wg, L));
  L.used = ++c->fstep_clock;
  c->fstep_cache.push_back(L);
  *out = &c->fstep_cache.back();
  return DSGD_OK;
}
// the tile tables and chunk records of a configuration (L.share: the chunks' shares of their worker's weight, or empty)
static int fstep_build(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, int n_wg, dsgd_ctx::FstepLayout& L) {
  const int n_workers = (int)row_segs.size();
  const std::vector<long long>&hrp = c->h_hrp, &ctp = c->h_ctp;
  std::vector<WTile> wt, ct;
  std::vector<FChunk> chunks((size_t)n_workers * (size_t)n_wg);
  long long worst = 1;
  for (int k = 0; k < n_workers; ++k) {
    const long long rb = row_segs[(size_t)k].row_begin, re = row_segs[(size_t)k].row_end;
    // a row's weight: its slots in the two streams (6 bytes each) plus what every row costs (descriptor share, dcold, coef8);
    // a LONG row (in neither stream) its own non-zeros plus what its wave waits for it: ~10 us, five tiles' worth (round 6:
    // the chunks with the most long rows were the launch's last, 0.7-0.9 us per long row)
    const std::vector<long long>& lrw = c->wlong_weight;   // prefix sums over the long-row list
    auto W = [&](long long i) {
      const size_t nl = (size_t)(std::lower_bound(c->wlong_rows.begin(), c->wlong_rows.end(), i) - c->wlong_rows.begin());
      return hrp[(size_t)i] + ctp[(size_t)i] + 2 * i + lrw[nl];
    };
    const long long w0 = W(rb), wtot = W(re) - w0;
    long long cut = rb;
    double share_total = 0.0, share_run = 0.0;
    const bool shares = L.share.size() == (size_t)n_workers * (size_t)n_wg;
    for (int b = 0; b < n_wg; ++b) share_total += shares ? L.share[(size_t)k * (size_t)n_wg + (size_t)b] : 1.0;
    for (int b = 0; b < n_wg; ++b) {
      long long nxt = re;
      share_run += shares ? L.share[(size_t)k * (size_t)n_wg + (size_t)b] : 1.0;
      if (b + 1 < n_wg) {
        const long long target = w0 + (long long)((double)wtot * share_run / share_total);
        long long lo = cut, hi = re;   // first row i in [cut, re] with W(i) >= target
        while (lo < hi) {
          const long long mid = (lo + hi) >> 1;
          if (W(mid) >= target) hi = mid;
          else lo = mid + 1;
        }
        nxt = lo;
      }
      FChunk& ch = chunks[(size_t)k * (size_t)n_wg + (size_t)b];
      ch.row_begin = (int)cut;
      ch.row_end = (int)nxt;
      ch.tile_begin = (int)wt.size();
      append_wave_tiles(hrp.data(), cut, nxt, wt, WS_MAXROWS);
      ch.tile_end = (int)wt.size();
      ch.ctile_begin = (int)ct.size();
      append_wave_tiles(ctp.data(), cut, nxt, ct, CT_MAXROWS);
      {
        // The chunk's cold tiles are walked twice by 16 waves with a stride of 16: 29-35 tiles of ~90 rows left some waves
        // three tiles and most two (phases A and C: 11 of a workgroup's 70 us).  Smaller tiles in a number just below a
        // multiple of 16 give every wave the same count (round 6).
        const size_t c0 = (size_t)ch.ctile_begin;
        auto waste = [](size_t n) { return n == 0 ? 1.0 : (double)((n + 15) / 16 * 16) / (double)n; };
        size_t n_best = ct.size() - c0;
        if (n_best % 16 != 0 && nxt > cut) {
          int mr_best = CT_MAXROWS;
          const long long target = (long long)((n_best + 15) / 16 * 16);
          int mr = (int)std::min<long long>(CT_MAXROWS, std::max<long long>(8, (nxt - cut + target - 1) / target));
          std::vector<WTile> tmp;
          for (int tries = 0; tries < 6 && mr >= 8 && waste(n_best) > 1.04; ++tries, --mr) {
            tmp.clear();
            append_wave_tiles(ctp.data(), cut, nxt, tmp, mr);
            if (waste(tmp.size()) < waste(n_best) && tmp.size() <= 3 * (size_t)target) {
              n_best = tmp.size();
              mr_best = mr;
            }
          }
          if (mr_best != CT_MAXROWS) {
            ct.resize(c0);
            append_wave_tiles(ctp.data(), cut, nxt, ct, mr_best);
          }
        }
      }
      ch.ctile_end = (int)ct.size();
      ch.long_begin = (int)(std::lower_bound(c->wlong_rows.begin(), c->wlong_rows.end(), cut) - c->wlong_rows.begin());
      ch.long_end = (int)(std::lower_bound(c->wlong_rows.begin(), c->wlong_rows.end(), nxt) - c->wlong_rows.begin());
      worst = std::max(worst, nxt - cut);
      cut = nxt;
    }
  }
  std::vector<unsigned short> meta, cmeta;
  wave_tile_meta(hrp.data(), c->h_label.data(), wt, meta);
  wave_tile_meta(ctp.data(), c->h_label.data(), ct, cmeta);
  WTile pad;   // (the tables are never empty: a clamped record read must stay inside them)
  pad.pos0 = 0;
  pad.r0 = 0;
  pad.info = 0xffff;
  if (wt.empty()) wt.push_back(pad);
  if (ct.empty()) ct.push_back(pad);
  L.worst_rows = worst;
  L.shift = -1;
  (void)hipFree(L.d_tiles);
  (void)hipFree(L.d_meta);
  (void)hipFree(L.d_ctiles);
  (void)hipFree(L.d_cmeta);
  (void)hipFree(L.d_chunks);
  L.d_tiles = nullptr;
  L.d_meta = nullptr;
  L.d_ctiles = nullptr;
  L.d_cmeta = nullptr;
  L.d_chunks = nullptr;
  hipError_t e = hipMalloc(&L.d_tiles, sizeof(WTile) * wt.size());
  if (e == hipSuccess) e = hipMalloc(&L.d_meta, sizeof(unsigned short) * meta.size());
  if (e == hipSuccess) e = hipMalloc(&L.d_ctiles, sizeof(WTile) * ct.size());
  if (e == hipSuccess) e = hipMalloc(&L.d_cmeta, sizeof(unsigned short) * cmeta.size());
  if (e == hipSuccess) e = hipMalloc(&L.d_chunks, sizeof(FChunk) * chunks.size());
  if (e == hipSuccess) e = hipMemcpy(L.d_tiles, wt.data(), sizeof(WTile) * wt.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.d_meta, meta.data(), sizeof(unsigned short) * meta.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.d_ctiles, ct.data(), sizeof(WTile) * ct.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.d_cmeta, cmeta.data(), sizeof(unsigned short) * cmeta.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(L.d_chunks, chunks.data(), sizeof(FChunk) * chunks.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !L.d_times) {
    e = hipMalloc(&L.d_times, sizeof(unsigned long long) * chunks.size());
    if (e == hipSuccess) e = hipHostMalloc(&L.h_times, sizeof(unsigned long long) * chunks.size(), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L.times_ev, hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMemset(L.d_times, 0, sizeof(unsigned long long) * chunks.size());
  L.launches = 0;
  L.times_pending = false;
  if (e != hipSuccess) {
    fstep_free(L);
    return fail(DSGD_EHIP, "row-chunk layout: %s", hipGetErrorString(e));
  }
  return DSGD_OK;
}
// Measured balance (opt-in, DSGD_FSTEP_REBALANCE=1): after FSTEP_CAL launches of a configuration the workgroups' summed durations come back (asynchronously);
// a later launch that finds them cuts the chunks again -- chunk b's share of the weight times (mean time / its time),
// within 15 % -- twice at most.  The slowest workgroup ends the launch: at N = 804,414 it ran 8 us behind the average of 75,
// the same workgroups every time (profiles/r06_fstep_wg_times.txt).  Integer sums: the re-cut changes no bit.
constexpr int FSTEP_CAL = 24;
static int fstep_maybe_rebalance(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, dsgd_ctx::FstepLayout* L) {
  if (!c->fstep_rebalance || L->rebalances >= 2 || !L->d_times || c->d_tprof) return DSGD_OK;
  const size_t n = (size_t)L->n_wg * row_segs.size();
  if (!L->times_pending) {
    if (L->launches >= FSTEP_CAL) {
      HIP_TRY(hipMemcpyAsync(L->h_times, L->d_times, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipEventRecord(L->times_ev, c->stream));
      L->times_pending = true;
    }
    return DSGD_OK;
  }
  if (hipEventQuery(L->times_ev) != hipSuccess) {
    (void)hipGetLastError();
    return DSGD_OK;
  }
  // (the copy covered exactly the launches enqueued in front of it; launches since then run on the old cut and are dropped)
  std::vector<double> share(n, 1.0);
  const bool had = L->share.size() == n;
  bool usable = true;
  for (size_t k = 0; k < row_segs.size() && usable; ++k) {
    double mean = 0.0;
    for (int b = 0; b < L->n_wg; ++b) mean += (double)L->h_times[k * (size_t)L->n_wg + (size_t)b];
    mean /= (double)L->n_wg;
    if (!(mean > 0.0)) usable = false;
    for (int b = 0; b < L->n_wg && usable; ++b) {
      const size_t i = k * (size_t)L->n_wg + (size_t)b;
      const double t = (double)L->h_times[i];
      double f = t > 0.0 ? mean / t : 1.0;
      f = std::min(1.15, std::max(0.85, f));
      share[i] = (had ? L->share[i] : 1.0) * f;
    }
  }
  L->times_pending = false;
  ++L->rebalances;
  if (!usable) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));   // (the tables in use go away)
  L->share = share;
  return fstep_build(c, row_segs, L->n_wg, *L);
}

// workgroups per worker of the chunked launch for these ranges (0: not this path)
static int fstep_grid(const dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, long long tot, long long ent) {
  // (from DSGD_FSTEP_MIN rows on -- or from 8,192 rows on when the range holds more entries than the column lists take)
  if (!fstep_possible(c) || tot > c->fstep_max || (tot < c->fstep_min && !(ent > c->tcol_max_nnz && tot >= 8192))) return 0;
  const int n_workers = (int)row_segs.size();
  if (n_workers > c->n_cu) return 0;
  long long smallest = tot;
  for (const StreamSeg& sg : row_segs) smallest = std::min(smallest, sg.row_end - sg.row_begin);
  const long long per_worker = std::max<long long>(1, c->n_cu / n_workers);
  return (int)std::max<long long>(1, std::min(per_worker, smallest / std::max<long long>(1, c->fstep_rows)));
}

static int launch_fstep(dsgd_ctx* c, const std::vector<StreamSeg>& row_segs, int n_wg) {
  const int n_workers = (int)row_segs.size();
  dsgd_ctx::FstepLayout* L = nullptr;
  DSGD_TRY(fstep_layout(c, row_segs, n_wg, &L));
  DSGD_TRY(fstep_maybe_rebalance(c, row_segs, L));
  ++L->launches;
  c->last_fstep_rebalances = L->rebalances;
  const int H = std::min(c->hsplit, c->dp);
  const int nc = c->dp - H;   // (fstep_possible: all of them inside the LDS tile)
  dim3 grid((unsigned)n_wg, (unsigned)n_workers);
  DSGD_TRY(ensure_part(c, &c->d_part, &c->part_wgs, &c->part_stride, (long long)n_wg * n_workers, H));
  DSGD_TRY(ensure_part(c, &c->d_partc, &c->partc_wgs, &c->partc_stride, (long long)n_wg * n_workers, nc));
  // LDS: the largest of the three phases' tiles
  const size_t lds_a = sizeof(float) * (size_t)(((nc + 3) & ~3) + 16 * CT_STRIP);
  const size_t lds_b = sizeof(float) * (size_t)(16 * WS_COEF_STRIDE + 2 * H + 64 + 4 + 4);   // (+ 4: the hot tiles' counter)
  const size_t lds_c = sizeof(float) * (size_t)(((nc + 64 + 3) & ~3) + 16 * CT_STRIP);
  const size_t lds = std::max(lds_a, std::max(lds_b, lds_c));
  // the fixed-point scale of the hot accumulators: one contribution per row and column, |contribution| <= 2^shift, a
  // chunk's rows known -- refined once per configuration by the data (dsgd_fstep_bound_kernel)
  int bits = 0;
  while ((1LL << bits) < L->worst_rows) ++bits;
  const int shift0 = std::max(1, std::min(c->max_shift, 30 - bits));
  int shift = shift0;
  if (c->fix_bound && shift0 < c->max_shift) {
    if (L->shift < 0) {
      if (!c->d_bound) HIP_TRY(hipMalloc(&c->d_bound, sizeof(unsigned int)));
      HIP_TRY(hipMemsetAsync(c->d_bound, 0, sizeof(unsigned int), c->stream));
      hipLaunchKernelGGL(dsgd_fstep_bound_kernel, grid, dim3(1024), sizeof(unsigned int) * (size_t)(H + 16), c->stream, c->d_hrow_ptr,
                         c->d_hcol, c->d_hval, L->d_chunks, view(c), c->d_wlong_rows, H, std::ldexp(1.0f, shift0 - c->vexp), c->d_bound);
      HIP_TRY(hipGetLastError());
      unsigned int amax = 0;
      HIP_TRY(hipMemcpyAsync(&amax, c->d_bound, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      int s2 = shift0;
      const double room = (double)(1LL << 30) - (double)L->worst_rows;
      while (s2 < c->max_shift && std::ldexp((double)amax, s2 + 1 - shift0) <= room) ++s2;
      L->shift = s2;
    }
    shift = L->shift;
  }
  const float main_scale = std::ldexp(1.0f, shift - c->vexp);
  c->last_shift = shift;
  CsrView mh = view(c);
  mh.row_ptr = c->d_hrow_ptr;
  mh.col = reinterpret_cast<const int*>(c->d_hcol);   // 16-bit ranks; the kernel reinterprets the pointer
  mh.val = c->d_hval;
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  hipLaunchKernelGGL(dsgd_fstep_kernel, grid, dim3(1024), lds, c->stream, mh, view(c), L->d_tiles, L->d_meta, L->d_ctiles, L->d_cmeta,
                     (const void*)c->d_ccol, c->d_cval, L->d_chunks, c->d_w, c->d_g64, (long long)c->dp, c->d_sc, H, nc, main_scale,
                     c->fix_scale, c->d_coef8, c->d_dcold, c->d_wlong_rows, c->d_part, c->part_stride, c->d_partc, c->partc_stride,
                     c->d_tprof, (c->fstep_rebalance && L->rebalances < 2) ? L->d_times : nullptr);
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  c->last_grad_kernel = "dsgd_fstep_kernel";
  DSGD_TRY(ensure_redpart(c));
  c->fused_args = {H, n_wg, H, nc, n_wg, 1.0 / (double)main_scale, 1.0 / (double)c->fix_scale};
  c->fused_apply_pending = true;   // launched by launch_finish_sync, which knows lr
  return DSGD_OK;
}

// ---- column lists: whole-split steps of 10^3 .. 10^5 rows (csrc/dsgd_tcol.hpp) ----------------------------------------
static void tcol_free(dsgd_ctx::TcolLayout& L) {
  (void)hipFree(L.d_ent_pk);
  (void)hipFree(L.d_ent_val);
  (void)hipFree(L.d_shares);
  (void)hipFree(L.d_key_of_cid);
  (void)hipFree(L.d_bit_base);
  (void)hipFree(L.d_bitmap);
  L = dsgd_ctx::TcolLayout();
}
static void tcol_drop_all(dsgd_ctx* c) {   // (the ranked CSR is about to change, or the context goes away)
  if (c->tcol_cache.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (dsgd_ctx::TcolLayout& L : c->tcol_cache) tcol_free(L);
  c->tcol_cache.clear();
}
static bool tcol_wanted(const dsgd_ctx* c, long long tot, int n_workers) {
  return c->tcol_enable && tot >= c->tcol_min && tot <= c->tcol_max && n_workers <= 64 && c->n_rows < (1LL << 31) &&
         (long long)n_workers * c->dp < (1LL << 24) && tot + 64LL * n_workers <= TC_MAX_BITS;
}
static dim3 tcol_row_grid(long long mx, int n_workers) {
  return dim3((unsigned)std::max<long long>(1, std::min<long long>((mx + TC_ROWS_PER_WG - 1) / TC_ROWS_PER_WG, 8192)), (unsigned)n_workers);
}
// the layout of these ranges (c->d_segs holds them): 1 = not possible (too many entries, no memory: the caller's other path)
static int tcol_layout(dsgd_ctx* c, const std::vector<WorkSeg>& segs, long long mx, dsgd_ctx::TcolLayout** out) {
  const int n_workers = (int)segs.size();
  std::vector<long long> key;
  for (const WorkSeg& sg : segs) {
    key.push_back(sg.begin);
    key.push_back(sg.end);
  }
  for (dsgd_ctx::TcolLayout& L : c->tcol_cache)
    if (L.gen == c->layout_gen && L.ranges == key) {
      L.used = ++c->tcol_clock;
      if (L.n_wg == 0) return 1;   // (declined before -- no memory, too many entries: not tried again step after step)
      c->tcol_miss_streak = 0;
      *out = &L;
      return DSGD_OK;
    }
  // A caller that never repeats a configuration (mini-batches over ever new row ranges) would pay a layout -- milliseconds --
  // for every 20 us step: more misses in a row than the cache holds, and the column lists leave such a context alone.
  // Not for good: after TCOL_COOLDOWN declined steps they try again -- a fit that settles on fixed splits after
  // a phase of ever new ranges gets its column lists back; a context that keeps changing pays 4 layouts per 256 steps.
  if (c->tcol_cooldown > 0) {
    if (--c->tcol_cooldown == 0) c->tcol_miss_streak = 8;
    return 1;
  }
  if (++c->tcol_miss_streak > 12) {
    c->tcol_cooldown = 256;
    return 1;
  }
  for (size_t i = 0; i < c->tcol_cache.size();)   // layouts of an earlier ranking
    if (c->tcol_cache[i].gen != c->layout_gen) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      tcol_free(c->tcol_cache[i]);
      c->tcol_cache.erase(c->tcol_cache.begin() + (long)i);
    } else {
      ++i;
    }
  if (c->tcol_cache.size() >= 8) {   // the least recently used configuration makes room
    size_t v = 0;
    for (size_t i = 1; i < c->tcol_cache.size(); ++i)
      if (c->tcol_cache[i].used < c->tcol_cache[v].used) v = i;
    HIP_TRY(hipStreamSynchronize(c->stream));
    tcol_free(c->tcol_cache[v]);
    c->tcol_cache.erase(c->tcol_cache.begin() + (long)v);
  }
  const int n_keys = n_workers * c->dp;
  unsigned int *d_cnt = nullptr, *d_ptr = nullptr, *d_cursor = nullptr;
  int* d_cid = nullptr;
  unsigned long long* d_tot = nullptr;
  dsgd_ctx::TcolLayout L;
  auto give_up = [&](int rc) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_cnt);
    (void)hipFree(d_ptr);
    (void)hipFree(d_cursor);
    (void)hipFree(d_cid);
    (void)hipFree(d_tot);
    tcol_free(L);
    if (rc == 2) return 1;   // out of memory / a failed call: declined for THIS step only (memory may be there the next time)
    if (rc == 1) {   // remembered as declined (n_wg = 0): the caller's other path takes this configuration from now on
      L.ranges = key;
      L.gen = c->layout_gen;
      L.used = ++c->tcol_clock;
      c->tcol_cache.push_back(L);
    }
    return rc;
  };
#define TC_SOFT(expr)                        \
  do {                                       \
    if ((expr) != hipSuccess) {              \
      (void)hipGetLastError();               \
      return give_up(2);                     \
    }                                        \
  } while (0)
  // every worker's rows padded to whole 64-row blocks of the bitmap (a workgroup of the dot kernel writes two whole words)
  std::vector<int> bit_base((size_t)n_workers);
  long long bits = 0;
  for (int k = 0; k < n_workers; ++k) {
    bit_base[(size_t)k] = (int)bits;
    bits += ((segs[(size_t)k].end - segs[(size_t)k].begin + 63) / 64) * 64;
  }
  if (bits > TC_MAX_BITS) return give_up(1);
  L.bm_words = (int)(((bits >> 5) + 3) & ~3LL);
  TC_SOFT(hipMalloc(&d_cnt, sizeof(unsigned int) * (size_t)n_keys));
  TC_SOFT(hipMalloc(&d_ptr, sizeof(unsigned int) * ((size_t)n_keys + 1)));
  TC_SOFT(hipMalloc(&d_cursor, sizeof(unsigned int) * (size_t)n_keys));
  TC_SOFT(hipMalloc(&d_cid, sizeof(int) * (size_t)n_keys));
  TC_SOFT(hipMalloc(&d_tot, sizeof(unsigned long long) * 2));
  TC_SOFT(hipMalloc(&L.d_bit_base, sizeof(int) * (size_t)n_workers));
  TC_SOFT(hipMalloc(&L.d_bitmap, sizeof(unsigned int) * (size_t)L.bm_words));
  TC_SOFT(hipMemcpyAsync(L.d_bit_base, bit_base.data(), sizeof(int) * (size_t)n_workers, hipMemcpyHostToDevice, c->stream));
  TC_SOFT(hipMemsetAsync(L.d_bitmap, 0, sizeof(unsigned int) * (size_t)L.bm_words, c->stream));
  TC_SOFT(hipMemsetAsync(d_cnt, 0, sizeof(unsigned int) * (size_t)n_keys, c->stream));
  const dim3 grid = tcol_row_grid(mx, n_workers);
  hipLaunchKernelGGL(dsgd_tc_count_kernel, grid, dim3(TC_THREADS), 0, c->stream, view(c), c->d_segs, c->dp, d_cnt);
  hipLaunchKernelGGL(dsgd_tc_scan_kernel, dim3(1), dim3(TC_THREADS), 0, c->stream, d_cnt, n_keys, d_ptr, d_cursor, d_cid, d_tot);
  TC_SOFT(hipGetLastError());
  unsigned long long tot[2] = {0, 0};
  TC_SOFT(hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, c->stream));
  TC_SOFT(hipStreamSynchronize(c->stream));   // (also: bit_base is a local)
  if (tot[0] == 0 || tot[0] >= (1ULL << 31)) return give_up(1);
  L.ranges = key;
  L.gen = c->layout_gen;
  L.n_ent = (long long)tot[0];
  long long share = (L.n_ent + c->n_cu - 1) / std::max(1, c->n_cu);
  share = std::max<long long>(1024, std::min<long long>(TC_MAX_SHARE, (share + 63) & ~63LL));
  if (c->tcol_share > 0) share = std::max<long long>(64, std::min<long long>(TC_MAX_SHARE, (c->tcol_share + 3) & ~3));   // (whole 16-byte pieces)
  L.share = (int)share;
  L.n_wg = (int)((L.n_ent + share - 1) / share);
  {   // whole 16-byte pieces; the last one's padding is zero
    const size_t n_pad = ((size_t)L.n_ent + 3) & ~(size_t)3;
    TC_SOFT(hipMalloc(&L.d_ent_pk, sizeof(unsigned int) * n_pad));
    TC_SOFT(hipMalloc(&L.d_ent_val, sizeof(float) * n_pad));
    TC_SOFT(hipMemsetAsync(L.d_ent_pk + (n_pad - 4), 0, sizeof(unsigned int) * 4, c->stream));
    TC_SOFT(hipMemsetAsync(L.d_ent_val + (n_pad - 4), 0, sizeof(float) * 4, c->stream));
  }
  TC_SOFT(hipMalloc(&L.d_shares, sizeof(TcShare) * (size_t)L.n_wg));
  TC_SOFT(hipMalloc(&L.d_key_of_cid, sizeof(int) * (size_t)std::max<unsigned long long>(1, tot[1])));
  hipLaunchKernelGGL(dsgd_tc_shares_kernel, dim3((unsigned)((L.n_wg + 255) / 256)), dim3(256), 0, c->stream, d_ptr, d_cid, n_keys, L.n_ent,
                     L.share, L.n_wg, L.d_shares);
  hipLaunchKernelGGL(dsgd_tc_fill_kernel, grid, dim3(TC_THREADS), 0, c->stream, view(c), c->d_segs, c->dp, d_cursor, d_cid, L.d_shares, L.share,
                     L.d_bit_base, L.d_ent_pk, L.d_ent_val, L.d_key_of_cid);
  TC_SOFT(hipGetLastError());
  TC_SOFT(hipStreamSynchronize(c->stream));
#undef TC_SOFT
  (void)hipFree(d_cnt);
  (void)hipFree(d_ptr);
  (void)hipFree(d_cursor);
  (void)hipFree(d_cid);
  (void)hipFree(d_tot);
  L.used = ++c->tcol_clock;
  c->tcol_cache.push_back(L);
  *out = &c->tcol_cache.back();
  return DSGD_OK;
}
// the gradient of the ranges in c->d_segs by column lists; 1 = declined (nothing launched: the caller's other path)
static int launch_tcol(dsgd_ctx* c, const std::vector<WorkSeg>& segs, long long mx) {
  const int n_workers = (int)segs.size();
  dsgd_ctx::TcolLayout* L = nullptr;
  const int lrc = tcol_layout(c, segs, mx, &L);
  if (lrc != DSGD_OK) return lrc;
  const int shift = std::max(1, std::min(c->max_shift, 30));   // 64-bit sums: nothing to bound (|contribution| <= 2^shift)
  c->last_shift = shift;
  const float scale = std::ldexp(1.0f, shift - c->vexp);
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  c->ctr_known = false;
  // 32 rows (512 lanes) per workgroup while that leaves the CUs a few workgroups each: finer grains fill them evenly
  // (18,519 rows: 290 workgroups of 64 rows leave 34 CUs with two and the rest with one)
  dim3 dgrid = tcol_row_grid(mx, n_workers);
  hipLaunchKernelGGL(dsgd_tc_dot_kernel<TC_THREADS>, dgrid, dim3(TC_THREADS), 0, c->stream, view(c), c->d_w, c->d_segs, L->d_bit_base,
                     L->d_bitmap, std::min(TC_WL, c->dp & ~3));
  HIP_TRY(hipGetLastError());
  TcGradArgs a;
  a.ent_pk = L->d_ent_pk;
  a.ent_val = L->d_ent_val;
  a.shares = L->d_shares;
  a.key_of_cid = L->d_key_of_cid;
  a.bitmap = L->d_bitmap;
  a.g64 = c->d_g64;
  a.sc = c->d_sc;
  a.n_ent = L->n_ent;
  a.share = L->share;
  a.bm_words = L->bm_words;
  a.scale = scale;
  {
    const dim3 g2((unsigned)L->n_wg);
    const size_t lds = sizeof(long long) * (size_t)L->share + sizeof(unsigned int) * (size_t)L->bm_words + 64;
    if (L->share <= 4 * TC_THREADS) hipLaunchKernelGGL(dsgd_tc_grad_kernel<1>, g2, dim3(TC_THREADS), lds, c->stream, a);
    else hipLaunchKernelGGL(dsgd_tc_grad_kernel<2>, g2, dim3(TC_THREADS), lds, c->stream, a);
  }
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  c->last_grad_kernel = "dsgd_tc_grad_kernel";
  DSGD_TRY(ensure_redpart(c));
  const double inv = 1.0 / (double)scale;
  c->fused_args = {0, 0, (c->dp + 3) & ~3, 0, 0, inv, inv};   // no partials: the exact sums are in the 64-bit accumulators
  c->fused_apply_pending = true;   // launched by launch_finish_sync, which knows lr
  return DSGD_OK;
}

static int hog_raise_stop(dsgd_ctx* c);

// small-batch steps as ONE persistent workgroup (dsgd_plan_kernel): eligible when no collective sits between the
// gradient and the update and a step is small enough for one CU
// does the list fit the staged sub-batch of dsgd_plan_kernel (at most PLAN_CAP rows and PLAN_CAP work items of 128
// non-zeros)?  Row lengths from the host copy of the internal row pointers.
static bool list_fits_staged(const dsgd_ctx* c, const int32_t* idx, long long n) {
  if (n > PLAN_CAP || c->h_row_ptr.size() != (size_t)c->n_rows + 1) return false;
  long long items = 0;
  for (long long t = 0; t < n; ++t) {
    const long long r = idx[t];
    if (r < 0 || r >= c->n_rows) return false;
    items += (c->h_row_ptr[(size_t)r + 1] - c->h_row_ptr[(size_t)r] + BT_CH - 1) / BT_CH;
  }
  return items <= PLAN_CAP;
}
static bool plan_kernel_ok(const dsgd_ctx* c, long long step_rows, int n_workers) {
  // (several hosted workers: their batches would run one after the other in the one workgroup -- 44 us for 3 x 100 --
  //  while dsgd_mb_grad_kernel gives every worker its own workgroups: 32 us with the four-launch finish, less fused)
  return c->plan_kernel && !c->comm && n_workers == 1 && step_rows <= c->plan_max_rows;
}
static int launch_plan_kernel(dsgd_ctx* c, const int* d_idx, const WorkSeg* d_segs, long long step_begin,
                              long long step_end, float lr, bool mail = false) {
  if (!c->d_plan_gcold) {
    const size_t strip = (size_t)std::max(1, c->dp - plan_hl(c->dp));
    HIP_TRY(hipMalloc(&c->d_plan_gcold, sizeof(float) * strip));
    HIP_TRY(hipMemsetAsync(c->d_plan_gcold, 0, sizeof(float) * strip, c->stream));
  }
  PlanArgs a;
  a.m = view(c);
  a.w = c->d_w;
  a.ds = c->d_ds;
  a.gcold = c->d_plan_gcold;
  a.idx = d_idx;
  a.segs = d_segs;
  a.sc = c->d_sc;
  a.step_begin = step_begin;
  a.step_end = step_end;
  a.lr = lr;
  a.lambda = (float)c->cfg.lambda;
  a.tprof = c->d_tprof;
  a.mail = mail ? c->d_mail : nullptr;
  a.mail_seq = mail ? ++c->mail_seq : 0ull;
  a.vexp = c->vexp;
  a.dp = c->dp;
  const size_t lds = sizeof(float) * (size_t)plan_lds_words(c->dp);
  size_t slot = 0;
  DSGD_TRY(prof_begin(c, &slot));
  // (every list fits the staged sub-batch: the callers checked the row lengths)
  c->ctr_known = false;
  hipLaunchKernelGGL(dsgd_plan_kernel, dim3(1), dim3(PLAN_THREADS), lds, c->stream, a);
  c->last_grad_kernel = "dsgd_plan_kernel";
  HIP_TRY(hipGetLastError());
  DSGD_TRY(prof_end(c, slot));
  c->s_lazy = false;
  c->s_dirty = false;     // the kernel leaves s = 2*lambda*(w . ds) of the weights it ends with ...
  c->nsq_dirty = true;    // ... but not |w|^2 (needed only by dsgd_loss_acc)
  return DSGD_OK;
}

static int require_data(dsgd_ctx* c) {
  if (!c->d_row_ptr) return fail(DSGD_ESTATE, "no data loaded (dsgd_load_csr)");
  return DSGD_OK;
}
// the synchronous entry points own w; they are refused while the lock-free engine is updating it
static int require_sync_mode(dsgd_ctx* c) {
  if (c->async_running) return fail(DSGD_ESTATE, "async computation running (dsgd_async_stop / dsgd_async_wait first)");
  return DSGD_OK;
}
static int require_ds(dsgd_ctx* c) {
  if (!c->have_ds) return fail(DSGD_ESTATE, "dimSparsity not set (dsgd_set_dim_sparsity / dsgd_build_dim_sparsity)");
  return DSGD_OK;
}

extern "C" {

int dsgd_abi_version(void) { return DSGD_ABI_VERSION; }
const char* dsgd_last_error(void) { return g_err; }

int dsgd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  int ok = 0;
  for (int i = 0; i < n; ++i) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  return ok;
}

int dsgd_create(const dsgd_config* cfg, dsgd_ctx** out) {
  if (!cfg || !out) return fail(DSGD_EINVAL, "null argument");
  if (cfg->n_features < 1) return fail(DSGD_EINVAL, "n_features must be >= 1");
  if (!(cfg->lambda == cfg->lambda)) return fail(DSGD_EINVAL, "lambda is NaN");
  if (cfg->flags != DSGD_F_DEFAULT) return fail(DSGD_EINVAL, "unknown flags 0x%x (none are defined)", cfg->flags);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(DSGD_EUNSUPPORTED, "no HIP device visible: libdsgd_hip has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= n) return fail(DSGD_EINVAL, "device %d out of range (%d devices)", cfg->device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DSGD_EUNSUPPORTED, "device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);
  dsgd_ctx* c = new (std::nothrow) dsgd_ctx();
  if (!c) return fail(DSGD_ENOMEM, "out of host memory");
  c->cfg = *cfg;
  c->dp = cfg->n_features + 1;
  c->n_cu = prop.multiProcessorCount;
  auto bail = [&](int rc) {
    dsgd_destroy(c);
    return rc;
  };
#define HIP_TRY_B(expr)                                                                               \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess) return bail(fail(DSGD_EHIP, "%s: %s", #expr, hipGetErrorString(e__)));     \
  } while (0)
  HIP_TRY_B(hipSetDevice(cfg->device));
  HIP_TRY_B(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY_B(hipMalloc(&c->d_w, sizeof(float) * (c->dp + 1)));  // + the zero slot w[dp] of the wseg kernels
  HIP_TRY_B(hipMemsetAsync(c->d_w + c->dp, 0, sizeof(float), c->stream));
  HIP_TRY_B(hipMalloc(&c->d_ds, sizeof(float) * c->dp));
  HIP_TRY_B(hipMalloc(&c->d_gsum, sizeof(float) * c->dp));
  HIP_TRY_B(hipMalloc(&c->d_tmp, sizeof(float) * c->dp));
  HIP_TRY_B(hipMalloc(&c->d_io, sizeof(float) * c->dp));
  HIP_TRY_B(hipMalloc(&c->d_perm, sizeof(int) * c->dp));
  HIP_TRY_B(hipMalloc(&c->d_sc, sizeof(DevScalars)));
  HIP_TRY_B(hipHostMalloc(&c->h_sc, sizeof(DevScalars), hipHostMallocDefault));
  HIP_TRY_B(hipHostMalloc(&c->h_mail, 4 * sizeof(unsigned long long), hipHostMallocMapped));
  c->h_mail[0] = c->h_mail[1] = c->h_mail[2] = c->h_mail[3] = 0;
  HIP_TRY_B(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_mail), c->h_mail, 0));
  HIP_TRY_B(hipMemsetAsync(c->d_w, 0, sizeof(float) * c->dp, c->stream));
  HIP_TRY_B(hipMemsetAsync(c->d_ds, 0, sizeof(float) * c->dp, c->stream));
  HIP_TRY_B(hipMemsetAsync(c->d_gsum, 0, sizeof(float) * c->dp, c->stream));
  HIP_TRY_B(hipMemsetAsync(c->d_sc, 0, sizeof(DevScalars), c->stream));
  int rc = ensure_g(c, 1);
  if (rc) return bail(rc);
  rc = set_identity_perm(c);
  if (rc) return bail(rc);
  c->hw_eval = std::min(c->dp, DSGD_LDS_FLOATS);
  // Switches that stay: one per live decision (A/B measurements, tests that force a path), all read once here.
  if (const char* e = getenv("DSGD_FIX_SHIFT")) c->max_shift = std::max(8, std::min(FIX_SHIFT, atoi(e)));   // cap of the fixed-point shift
  if (const char* e = getenv("DSGD_FIX_BOUND")) c->fix_bound = atoi(e) != 0;      // 0: data-independent bound only
  if (const char* e = getenv("DSGD_PLAN_KERNEL")) c->plan_kernel = atoi(e) != 0;  // 0: small batches through the multi-workgroup kernel
  if (const char* e = getenv("DSGD_VT")) c->vt_enable = atoi(e) != 0;             // 0: plans' index lists through dsgd_mb_grad_kernel
  if (const char* e = getenv("DSGD_CS")) c->cs_enable = atoi(e) != 0;             // 0: small plan steps through the row-parallel kernels
  if (const char* e = getenv("DSGD_CS_G")) c->cs_g = atoi(e) == 16 ? 16 : (atoi(e) == 8 ? 8 : 0);
  if (const char* e = getenv("DSGD_CS_MAX_MB")) c->cs_max_mb = std::max(0, atoi(e));
  if (const char* e = getenv("DSGD_CS_HOST_LAYOUT")) c->cs_host_layout = atoi(e) != 0;
  if (const char* e = getenv("DSGD_CS_REQ")) c->cs_req = atoi(e) != 0;
  if (const char* e = getenv("DSGD_CACHE_MB")) c->cache_cap = (size_t)std::max(0, atoi(e)) << 20;
#ifdef DSGD_TEST_COLLECTIVE_SEAM
  if (const char* e = getenv("DSGD_TEST_CS_SKIP_PUBLISH")) c->cs_test_skip = std::max(0, atoi(e));
#endif
  if (const char* e = getenv("DSGD_REQ_MAPPED")) c->req_mapped = atoi(e) != 0;
  if (const char* e = getenv("DSGD_REQ_PLAN")) c->req_plan = atoi(e) != 0;
  if (const char* e = getenv("DSGD_STREAM_MIN")) c->stream_min = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_FSTEP")) c->fstep_enable = atoi(e) != 0;
  if (const char* e = getenv("DSGD_FSTEP_MIN")) c->fstep_min = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_FSTEP_MAX")) c->fstep_max = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_FSTEP_ROWS")) c->fstep_rows = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_FSTEP_REBALANCE")) c->fstep_rebalance = atoi(e) != 0;
  if (const char* e = getenv("DSGD_TCOL")) c->tcol_enable = atoi(e) != 0;
  if (const char* e = getenv("DSGD_TCOL_MIN")) c->tcol_min = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_TCOL_MAX")) c->tcol_max = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_TCOL_MAX_NNZ")) c->tcol_max_nnz = std::max(1LL, atoll(e));
  if (const char* e = getenv("DSGD_TCOL_SHARE")) c->tcol_share = std::max(0, atoi(e));
  if (const char* e = getenv("DSGD_REQ_SPIN")) c->req_spin = atoi(e) != 0;
  if (const char* e = getenv("DSGD_CS_NT")) c->cs_nt = atoi(e) == CS_THREADS_NARROW ? CS_THREADS_NARROW : 0;
  if (const char* e = getenv("DSGD_VT_TPW")) c->vt_tpw = std::max(1, atoi(e));
  if (const char* e = getenv("DSGD_VT_PACK_MB")) c->vt_pack_mb = std::max(0, atoi(e));
  if (const char* e = getenv("DSGD_HSPLIT")) c->hsplit = atoi(e);                 // hot/cold split rank (tests: wide models)
  if (getenv("DSGD_PLAN_PROF") && atoi(getenv("DSGD_PLAN_PROF"))) {
    // (16 counters + four words per workgroup of the chunked launch's LAST run: start, end, cycles in the hot tiles, XCC id)
    HIP_TRY_B(hipMalloc(&c->d_tprof, sizeof(unsigned long long) * (16 + 4 * 1024)));
    HIP_TRY_B(hipMemsetAsync(c->d_tprof, 0, sizeof(unsigned long long) * (16 + 4 * 1024), c->stream));
  }
  c->hsplit = std::max(1, std::min(c->hsplit, (DSGD_LDS_FLOATS - 16 * WS_COEF_STRIDE - 4 - 64) / 2));   // (< 65536: 16-bit ranks)
  if (c->hsplit >= 8) c->hsplit &= ~3;   // 16-byte aligned tile boundaries: the LDS tiles are staged / written back in 16-byte pieces
  const int lds_max = DSGD_LDS_FLOATS * (int)sizeof(float);
#define DSGD_ATTR(fn) HIP_TRY_B(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max))
  DSGD_ATTR(dsgd_eval_kernel<64>);
  DSGD_ATTR(dsgd_eval_kernel<32>);
  DSGD_ATTR(dsgd_eval_kernel<16>);
  DSGD_ATTR(dsgd_eval_kernel<8>);
  DSGD_ATTR(dsgd_colcount_kernel);
  DSGD_ATTR((dsgd_hogwild_kernel<false, false>));
  DSGD_ATTR((dsgd_hogwild_kernel<true, false>));
  DSGD_ATTR((dsgd_hogwild_kernel<false, true>));
  DSGD_ATTR((dsgd_hogwild_kernel<true, true>));
  DSGD_ATTR(dsgd_mb_grad_kernel);
  DSGD_ATTR(dsgd_vt_grad_kernel<true>);
  DSGD_ATTR(dsgd_vt_grad_kernel<false>);
  DSGD_ATTR(dsgd_plan_kernel);
  DSGD_ATTR((dsgd_cs_step_kernel<CS_THREADS, 1, 4>));
  DSGD_ATTR((dsgd_cs_step_kernel<CS_THREADS, 2, 8>));
  DSGD_ATTR((dsgd_cs_step_kernel<CS_THREADS_NARROW, 2, 8>));
  DSGD_ATTR(dsgd_wseg_kernel<true>);
  DSGD_ATTR(dsgd_wseg_kernel<false>);
  DSGD_ATTR(dsgd_wseg_bound_kernel);
  DSGD_ATTR(dsgd_fstep_kernel);
  DSGD_ATTR(dsgd_fstep_bound_kernel);
  {   // the column lists' gradient kernel: its table, the bitmap, 16 words
    const int tc_lds = (int)(sizeof(long long) * TC_MAX_SHARE + TC_MAX_BITS / 8 + 64);
#define DSGD_ATTR_TC(fn) HIP_TRY_B(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, tc_lds))
    DSGD_ATTR_TC(dsgd_tc_grad_kernel<1>);
    DSGD_ATTR_TC(dsgd_tc_grad_kernel<2>);
#undef DSGD_ATTR_TC
  }
  DSGD_ATTR((dsgd_cold_kernel<true, false, false>));
  DSGD_ATTR((dsgd_cold_kernel<true, false, true>));
  DSGD_ATTR((dsgd_cold_kernel<false, false, false>));
  DSGD_ATTR((dsgd_cold_kernel<false, false, true>));
  DSGD_ATTR((dsgd_cold_kernel<true, true, false>));
  DSGD_ATTR((dsgd_cold_kernel<true, true, true>));
  DSGD_ATTR((dsgd_cold_kernel<false, true, false>));
  DSGD_ATTR((dsgd_cold_kernel<false, true, true>));
#undef DSGD_ATTR
  HIP_TRY_B(hipStreamSynchronize(c->stream));
#undef HIP_TRY_B
  *out = c;
  return DSGD_OK;
}

int dsgd_destroy(dsgd_ctx* c) {
  if (!c) return DSGD_OK;
  std::unique_lock<std::mutex> lk(c->mu);   // a call still running on another thread finishes first
  (void)hipSetDevice(c->cfg.device);
  // The persistent Hogwild kernel only exits on its stop flag: raise it and wait BEFORE any hipFree (hipFree
  // synchronises the device -- it would wait forever on that kernel, or free memory the kernel still reads).
  if (c->async_stream) {
    (void)hog_raise_stop(c);
    // a thread inside dsgd_async_wait / dsgd_async_stop blocks on the engine WITHOUT the mutex (async_wait_released) and
    // comes back for it: it is the one thread that joins exch_thread, and it still uses the context afterwards -- wait
    // until it has left (the stop flag above lets it finish)
    c->join_cv.wait(lk, [c] { return !c->join_in_progress; });
    if (c->exch_thread.joinable()) c->exch_thread.join();
    (void)hipStreamSynchronize(c->async_stream);
    c->async_running = false;
  }
  if (c->upd_stream) (void)hipStreamSynchronize(c->upd_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm && rccl::available()) rccl::CommDestroy(c->comm);
  for (auto& e : c->prof_ev) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  (void)hipFree(c->d_row_ptr);
  (void)hipFree(c->d_col);
  (void)hipFree(c->d_val);
  (void)hipFree(c->d_label);
  (void)hipFree(c->d_w);
  (void)hipFree(c->d_ds);
  (void)hipFree(c->d_g);
  (void)hipFree(c->d_g64);
  (void)hipFree(c->d_gsum);
  (void)hipFree(c->d_tmp);
  (void)hipFree(c->d_io);
  (void)hipFree(c->d_perm);
  (void)hipFree(c->d_sc);
  (void)hipFree(c->d_idx);
  (void)hipFree(c->d_segs);
  (void)hipFree(c->d_ssegs);
  (void)hipFree(c->d_coef8);
  (void)hipFree(c->d_wtiles);
  (void)hipFree(c->d_wmeta);
  fstep_drop_all(c);
  tcol_drop_all(c);
  (void)hipFree(c->d_wlong_rows);
  (void)hipFree(c->d_part);
  (void)hipFree(c->d_partc);
  (void)hipFree(c->d_hcol);
  (void)hipFree(c->d_hval);
  (void)hipFree(c->d_hrow_ptr);
  (void)hipFree(c->d_ccol);
  (void)hipFree(c->d_cval);
  (void)hipFree(c->d_ctp);
  (void)hipFree(c->d_ctiles);
  (void)hipFree(c->d_cmeta);
  (void)hipFree(c->d_dcold);
  (void)hipFree(c->d_bound);
  (void)hipFree(c->d_redpart);
  (void)hipFree(c->d_pred);
  pin_free(c->pin_w);
  pin_free(c->pin_idx);
  pin_free(c->pin_segs);
  pin_free(c->pin_out);
  pin_free(c->pin_upd);
  (void)hipFree(c->d_upd_key);
  (void)hipFree(c->d_upd_dv);
  if (c->upd_stream) (void)hipStreamDestroy(c->upd_stream);
  if (c->async_stream) (void)hipStreamDestroy(c->async_stream);
  if (c->query_stream) (void)hipStreamDestroy(c->query_stream);
  (void)hipFree(c->d_hog);
  (void)hipFree(c->d_gcold);
  (void)hipFree(c->d_asg);
  if (c->h_hog) (void)hipHostFree(c->h_hog);
  if (c->h_one) (void)hipHostFree(c->h_one);
  (void)hipFree(c->d_hog_it);
  (void)hipFree(c->d_trace);
  (void)hipFree(c->d_tdot);
  (void)hipFree(c->d_wprev);
  (void)hipFree(c->d_wdelta);
  (void)hipFree(c->d_tprof);
  (void)hipFree(c->d_plan_gcold);
  if (c->build_stream) {
    (void)hipStreamSynchronize(c->build_stream);
    (void)hipStreamDestroy(c->build_stream);
  }
  cache_drop_all(c);
  for (int i = 0; i < 9; ++i) (void)hipFree(c->seed_scratch[i]);
  (void)hipFree(c->req_layout.hdr);
  (void)hipFree(c->req_layout.meta);
  (void)hipFree(c->req_layout.rf);
  (void)hipFree(c->req_layout.col);
  (void)hipFree(c->req_layout.val);
  (void)hipFree(c->req_layout.cl);
  (void)hipFree(c->d_cs_max);
  if (c->h_cs_max) (void)hipHostFree(c->h_cs_max);
  (void)hipFree(c->d_cs_x);
  (void)hipFree(c->d_cs_sync);
  (void)hipFree(c->d_cs_w);
  (void)hipFree(c->d_cs_ds);
  if (c->h_sc) (void)hipHostFree(c->h_sc);
  if (c->h_mail) (void)hipHostFree(c->h_mail);
  if (c->h_req) (void)hipHostFree(c->h_req);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  lk.unlock();
  delete c;
  return DSGD_OK;
}

int dsgd_load_csr(dsgd_ctx* c, int64_t n_rows, const int64_t* row_ptr_in, const int32_t* col_in, const float* val_in,
                  const int8_t* label) {
  DSGD_TRY(check_ctx(c));
  if (n_rows < 1 || !row_ptr_in || !label) return fail(DSGD_EINVAL, "n_rows must be >= 1 and arrays non-null");
  if (row_ptr_in[0] != 0) return fail(DSGD_EINVAL, "row_ptr[0] must be 0");
  const int64_t* row_ptr = row_ptr_in;
  const int32_t* col = col_in;
  const float* val = val_in;
  int64_t nnz = row_ptr[n_rows];
  if (nnz < 0 || (nnz > 0 && (!col || !val))) return fail(DSGD_EINVAL, "bad nnz / null col,val");
  for (int64_t i = 0; i < n_rows; ++i)
    if (row_ptr[i + 1] < row_ptr[i]) return fail(DSGD_EINVAL, "row_ptr not monotone at row %lld", (long long)i);
  // keys must be valid Sparse keys for a vector of size D (ref: math/Sparse.scala:61-68 accepts 0..size)
  float vmax = 0.0f;
  for (int64_t p = 0; p < nnz; ++p) {
    if (col[p] < 0 || col[p] > c->cfg.n_features)
      return fail(DSGD_ERANGE, "column id %d at nnz %lld outside [0, %d]", col[p], (long long)p, c->cfg.n_features);
    const float a = std::fabs(val[p]);
    if (!(a <= 3.0e38f)) return fail(DSGD_EINVAL, "non-finite value at nnz %lld", (long long)p);  // Vec.scala:14 NaN guard
    vmax = std::max(vmax, a);
  }
  for (int64_t i = 0; i < n_rows; ++i)
    if (label[i] != 1 && label[i] != -1) return fail(DSGD_EINVAL, "label[%lld] = %d, expected +1/-1", (long long)i, label[i]);
  // A row is a Map[Int, Number] in the reference (math/Sparse.scala:11): a key occurs once.  The fixed-point bound of
  // the streaming kernels ("one contribution per row and column") relies on it, so duplicates are rejected here
  // (ascending rows -- the RCV1 files -- take the cheap path).
  {
    std::vector<int32_t> keys;
    for (int64_t i = 0; i < n_rows; ++i) {
      const int64_t b = row_ptr[i], e = row_ptr[i + 1];
      bool ascending = true;
      for (int64_t p = b + 1; p < e && ascending; ++p) ascending = col[p] > col[p - 1];
      if (ascending) continue;
      keys.assign(col + b, col + e);
      std::sort(keys.begin(), keys.end());
      if (std::adjacent_find(keys.begin(), keys.end()) != keys.end())
        return fail(DSGD_EINVAL, "row %lld holds a key twice (a Sparse vector is a map: one value per key)", (long long)i);
    }
  }
  // Internal CSR: an empty row (Sparse.zeros) gets ONE explicit zero on key 0.  x.w, the gate and the
  // gradient are unchanged (the product and y*x are 0 and every counting kernel skips abs(v) <= 1e-20),
  // and the streaming kernels can rely on "every row owns at least one slot".
  std::vector<int64_t> prow;
  std::vector<int32_t> pcol;
  std::vector<float> pval;
  {
    int64_t n_empty = 0;
    for (int64_t i = 0; i < n_rows; ++i) n_empty += row_ptr[i + 1] == row_ptr[i];
    if (n_empty > 0) {
      prow.resize((size_t)n_rows + 1);
      pcol.reserve((size_t)(nnz + n_empty));
      pval.reserve((size_t)(nnz + n_empty));
      prow[0] = 0;
      for (int64_t i = 0; i < n_rows; ++i) {
        if (row_ptr[i + 1] == row_ptr[i]) {
          pcol.push_back(0);
          pval.push_back(0.0f);
        } else {
          pcol.insert(pcol.end(), col + row_ptr[i], col + row_ptr[i + 1]);
          pval.insert(pval.end(), val + row_ptr[i], val + row_ptr[i + 1]);
        }
        prow[i + 1] = (int64_t)pcol.size();
      }
      row_ptr = prow.data();
      col = pcol.data();
      val = pval.data();
      nnz = row_ptr[n_rows];
    }
  }
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_sync_mode(c));   // the persistent engine reads the matrix that would be freed here
  HIP_TRY(hipStreamSynchronize(c->stream));
  DSGD_TRY(reset_layout(c));
  (void)hipFree(c->d_row_ptr);
  (void)hipFree(c->d_col);
  (void)hipFree(c->d_val);
  (void)hipFree(c->d_label);
  c->d_row_ptr = nullptr;
  c->d_col = nullptr;
  c->d_val = nullptr;
  c->d_label = nullptr;
  HIP_TRY(hipMalloc(&c->d_row_ptr, sizeof(long long) * (size_t)(n_rows + 1)));
  if (n_rows >= (int64_t)1 << 31) return fail(DSGD_EUNSUPPORTED, "more than 2^31-1 rows per context");
  // padding: the streaming kernels read whole windows (up to 512 slots) without clamping
  HIP_TRY(hipMalloc(&c->d_col, sizeof(int) * (size_t)(nnz + WS_PAD)));
  HIP_TRY(hipMalloc(&c->d_val, sizeof(float) * (size_t)(nnz + WS_PAD)));
  HIP_TRY(hipMemset(c->d_col + nnz, 0, sizeof(int) * WS_PAD));
  HIP_TRY(hipMemset(c->d_val + nnz, 0, sizeof(float) * WS_PAD));
  HIP_TRY(hipMalloc(&c->d_label, (size_t)n_rows));
  HIP_TRY(hipMemcpy(c->d_row_ptr, row_ptr, sizeof(long long) * (size_t)(n_rows + 1), hipMemcpyHostToDevice));
  if (nnz) {
    HIP_TRY(hipMemcpy(c->d_col, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_val, val, sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemcpy(c->d_label, label, (size_t)n_rows, hipMemcpyHostToDevice));
  c->n_rows = n_rows;
  c->nnz = nnz;
  {
    int e = 0;
    std::frexp(vmax > 0.0f ? vmax : 1.0f, &e);  // vmax = f * 2^e, f in [0.5, 1)  ->  vmax2 = 2^e >= vmax
    if (vmax > 0.0f && std::ldexp(1.0f, e - 1) == vmax) e -= 1;  // vmax itself a power of two
    c->vexp = e;   // vmax2 = 2^e
    c->fix_scale = std::ldexp(1.0f, FIX_SHIFT - e);
  }
  c->h_row_ptr.assign(row_ptr, row_ptr + n_rows + 1);
  c->h_label.assign(label, label + n_rows);
  c->ssegs_last.clear();
  // (the wave tiles cover the hot stream and are built with the layout, at the first compute call)
  const double mean = (double)nnz / (double)n_rows;
  c->group = mean > 192.0 ? 64 : (mean > 96.0 ? 32 : (mean > 12.0 ? 16 : 8));
  return DSGD_OK;
}

int dsgd_n_rows(dsgd_ctx* c, int64_t* n_rows, int64_t* nnz) {
  DSGD_TRY(check_ctx(c));
  if (n_rows) *n_rows = c->n_rows;
  if (nnz) *nnz = c->nnz;
  return DSGD_OK;
}

int dsgd_set_dim_sparsity(dsgd_ctx* c, const float* ds) {
  DSGD_TRY(check_ctx(c));
  if (!ds) return fail(DSGD_EINVAL, "null ds");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_sync_mode(c));
  HIP_TRY(hipMemcpyAsync(c->d_io, ds, sizeof(float) * c->dp, hipMemcpyHostToDevice, c->stream));
  DSGD_TRY(launch_permute_in(c, c->d_io, c->d_ds));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->have_ds = true;
  c->s_dirty = true;
  return DSGD_OK;
}

// dimSparsity in three parts around its all-reduce (one host thread, several contexts: dsgd_build_dim_sparsity_devices)
static int ds_begin(dsgd_ctx* c, long long n_train, unsigned int** d_cnt_out) {   // (layout ready)
  *d_cnt_out = nullptr;
  long long nnz_train = 0;
  HIP_TRY(hipMemcpy(&nnz_train, c->d_row_ptr + n_train, sizeof(long long), hipMemcpyDeviceToHost));
  unsigned int* d_cnt = nullptr;
  HIP_TRY(hipMalloc(&d_cnt, sizeof(unsigned int) * c->dp));
  int rc = reset_counters(c);
  if (!rc) rc = count_columns(c, nnz_train, d_cnt);
  if (rc) {
    (void)hipFree(d_cnt);
    return rc;
  }
  *d_cnt_out = d_cnt;
  return DSGD_OK;
}
static int ds_collective(dsgd_ctx* c, unsigned int* d_cnt) {
  if (!c->comm) return DSGD_OK;
  // the reference counts over the WHOLE train set (Main.scala:57-60); shards sum their counts
  int r = rccl::AllReduce(d_cnt, d_cnt, (size_t)c->dp, rccl::kUint32, rccl::kSum, c->comm, c->stream);
  if (r) return fail(DSGD_ERCCL, "ncclAllReduce(feature counts): %s", rccl::GetErrorString(r));
  return DSGD_OK;
}
static int ds_finish(dsgd_ctx* c, unsigned int* d_cnt, float* ds_out) {   // (takes ownership of d_cnt)
  int rc = DSGD_OK;
  unsigned int cnt_key0 = 0;
  {
    // Main.scala:60 does buff(idx - 1): a feature id 0 in a train row would index buff(-1)
    int rank0 = 0;
    hipError_t e = hipMemcpyAsync(&rank0, c->d_perm, sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(&cnt_key0, d_cnt + rank0, sizeof(unsigned int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(DSGD_EHIP, "dimSparsity: %s", hipGetErrorString(e));
  }
  if (!rc) {
    hipLaunchKernelGGL(dsgd_ds_kernel, dim3((c->dp + 255) / 256), dim3(256), 0, c->stream, d_cnt, c->d_perm, c->d_ds, c->dp);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) rc = fail(DSGD_EHIP, "dimSparsity kernels: %s", hipGetErrorString(le));
  }
  if (!rc) rc = read_scalars(c);
  (void)hipFree(d_cnt);
  DSGD_TRY(rc);
  DSGD_TRY(check_err_flag(c));
  if (cnt_key0) return fail(DSGD_ERANGE, "feature id 0 cannot be counted by Main.scala:60 (buff(idx - 1))");
  c->have_ds = true;
  c->s_dirty = true;
  if (ds_out) {
    DSGD_TRY(launch_permute_out(c, c->d_ds, c->d_io));
    HIP_TRY(hipMemcpyAsync(ds_out, c->d_io, sizeof(float) * c->dp, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return DSGD_OK;
}
static int ds_checks(dsgd_ctx* c, long long n_train) {
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_sync_mode(c));
  if (n_train < 1 || n_train > c->n_rows) return fail(DSGD_EINVAL, "n_train %lld outside [1, %lld]", n_train, c->n_rows);
  return DSGD_OK;
}

int dsgd_build_dim_sparsity(dsgd_ctx* c, int64_t n_train, float* ds_out) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(ds_checks(c, n_train));
  DSGD_TRY(prepare_layout(c));
  unsigned int* d_cnt = nullptr;
  DSGD_TRY(ds_begin(c, n_train, &d_cnt));
  const int rc = ds_collective(c, d_cnt);
  if (rc) {
    (void)hipFree(d_cnt);
    return rc;
  }
  return ds_finish(c, d_cnt, ds_out);
}

static int set_weights_locked(dsgd_ctx* c, const float* w) {
  const size_t bytes = sizeof(float) * (size_t)c->dp;
  DSGD_TRY(pin_acquire(c->pin_w, bytes));
  memcpy(c->pin_w.p, w, bytes);   // the caller may reuse w as soon as this returns
  HIP_TRY(hipMemcpyAsync(c->d_io, c->pin_w.p, bytes, hipMemcpyHostToDevice, c->stream));
  DSGD_TRY(pin_sent(c, c->pin_w));
  DSGD_TRY(launch_permute_in(c, c->d_io, c->d_w));
  c->s_dirty = true;
  return DSGD_OK;
}

int dsgd_set_weights(dsgd_ctx* c, const float* w) {
  DSGD_TRY(check_ctx(c));
  if (!w) return fail(DSGD_EINVAL, "null w");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_sync_mode(c));
  return set_weights_locked(c, w);
}

int dsgd_get_weights(dsgd_ctx* c, float* w_out) {
  DSGD_TRY(check_ctx(c));
  if (!w_out) return fail(DSGD_EINVAL, "null w_out");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  const size_t bytes = sizeof(float) * (size_t)c->dp;
  DSGD_TRY(pin_acquire(c->pin_out, bytes));
  DSGD_TRY(launch_permute_out(c, c->d_w, c->d_io));
  HIP_TRY(hipMemcpyAsync(c->pin_out.p, c->d_io, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(w_out, c->pin_out.p, bytes);
  return DSGD_OK;
}

// stage host index lists for n_workers workers; returns the largest list length
constexpr long long REQ_MAPPED_ITEMS = 16384;   // index entries of one request read in place from host memory (64 KiB)
static int stage_lists(dsgd_ctx* c, const int32_t* const* idx_per_worker, const int64_t* n_per_worker, int n_workers,
                       long long* max_items, long long* total) {
  long long tot = 0, mx = 0;
  for (int k = 0; k < n_workers; ++k) {
    if (n_per_worker[k] <= 0)
      return fail(DSGD_EINVAL, "worker %d has an empty sample list: Vec.sum requires a non-empty list", k);
    if (!idx_per_worker[k]) return fail(DSGD_EINVAL, "null index list for worker %d", k);
    tot += n_per_worker[k];
    mx = std::max<long long>(mx, n_per_worker[k]);
  }
  std::vector<WorkSeg> segs(n_workers);
  long long off = 0;
  if (c->req_mapped && tot <= REQ_MAPPED_ITEMS) {
    // the reference's batch sizes: the lists go into a host-mapped buffer the kernels read in place -- no copy on the
    // stream in front of the launch (every per-request entry point returns behind its kernels: the buffer is free again)
    if (!c->h_req) {
      HIP_TRY(hipHostMalloc(&c->h_req, sizeof(int) * (size_t)REQ_MAPPED_ITEMS, hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_req), c->h_req, 0));
    }
    for (int k = 0; k < n_workers; ++k) {
      memcpy(c->h_req + off, idx_per_worker[k], sizeof(int) * (size_t)n_per_worker[k]);
      segs[k].begin = off;
      segs[k].end = off + n_per_worker[k];
      off += n_per_worker[k];
    }
    c->cur_idx = c->d_req;
  } else {
    DSGD_TRY(ensure_idx(c, tot));
    // one copy for all lists, on the stream (behind the previous step's kernels, which may still be reading d_idx)
    DSGD_TRY(pin_acquire(c->pin_idx, sizeof(int) * (size_t)tot));
    for (int k = 0; k < n_workers; ++k) {
      memcpy(static_cast<int*>(c->pin_idx.p) + off, idx_per_worker[k], sizeof(int) * (size_t)n_per_worker[k]);
      segs[k].begin = off;
      segs[k].end = off + n_per_worker[k];
      off += n_per_worker[k];
    }
    HIP_TRY(hipMemcpyAsync(c->d_idx, c->pin_idx.p, sizeof(int) * (size_t)tot, hipMemcpyHostToDevice, c->stream));
    DSGD_TRY(pin_sent(c, c->pin_idx));
    c->cur_idx = c->d_idx;
  }
  DSGD_TRY(upload_segs(c, segs));
  *max_items = mx;
  *total = tot;
  return DSGD_OK;
}

int dsgd_gradient(dsgd_ctx* c, const float* w, const int32_t* idx, int64_t n, float* g_out, dsgd_batch_stats* stats) {
  DSGD_TRY(check_ctx(c));
  if (!g_out) return fail(DSGD_EINVAL, "null g_out");
  if (n <= 0 || !idx) return fail(DSGD_EINVAL, "Cannot sum an empty list of vectors");  // ref: math/Vec.scala:129
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));   // g accumulators, s and (with w != NULL) the weights belong to the running engine
  DSGD_TRY(prepare_layout(c));
  if (w) DSGD_TRY(set_weights_locked(c, w));
  DSGD_TRY(ensure_s(c));
  DSGD_TRY(reset_counters(c));
  long long mx = 0, tot = 0;
  const int64_t nn = n;
  DSGD_TRY(stage_lists(c, &idx, &nn, 1, &mx, &tot));
  DSGD_TRY(launch_grad(c, c->cur_idx, c->d_segs, 1, mx));
  hipLaunchKernelGGL(dsgd_regularize_kernel, dim3((c->dp + 1023) / 1024, 1), dim3(1024), 0, c->stream, c->d_g,
                     (long long)c->dp, c->dp, c->d_sc);
  HIP_TRY(hipGetLastError());
  const size_t bytes = sizeof(float) * (size_t)c->dp;
  DSGD_TRY(pin_acquire(c->pin_out, bytes));
  DSGD_TRY(launch_permute_out(c, c->d_g, c->d_io));
  HIP_TRY(hipMemcpyAsync(c->pin_out.p, c->d_io, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_g, 0, bytes, c->stream));
  DSGD_TRY(read_scalars(c));   // the one synchronisation of the call
  memcpy(g_out, c->pin_out.p, bytes);
  DSGD_TRY(prof_collect(c));
  DSGD_TRY(check_err_flag(c));
  if (stats) {
    stats->n_samples = n;
    stats->n_active = (int64_t)c->h_sc->n_active;
  }
  return DSGD_OK;
}

int dsgd_apply(dsgd_ctx* c, const float* g_mean, float lr) {
  DSGD_TRY(check_ctx(c));
  if (!g_mean) return fail(DSGD_EINVAL, "null g_mean");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));
  HIP_TRY(hipMemcpyAsync(c->d_io, g_mean, sizeof(float) * c->dp, hipMemcpyHostToDevice, c->stream));
  DSGD_TRY(launch_permute_in(c, c->d_io, c->d_gsum));
  DSGD_TRY(ensure_redpart(c));
  hipLaunchKernelGGL(dsgd_apply_cols_kernel, dim3((c->dp + FRA_COLS - 1) / FRA_COLS), dim3(256), 0, c->stream, c->d_w,
                     c->d_gsum, c->d_ds, c->dp, 1.0f, lr, (float)c->cfg.lambda, c->d_sc, red_out(c), 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->red_par ^= 1;
  c->s_lazy = false;
  c->s_dirty = false;
  return DSGD_OK;
}

static int finish_stats(dsgd_ctx* c, dsgd_batch_stats* stats, long long total) {
  DSGD_TRY(read_scalars(c));
  DSGD_TRY(prof_collect(c));
  DSGD_TRY(check_err_flag(c));
  if (stats) {
    stats->n_samples = total;
    stats->n_active = (int64_t)c->h_sc->n_active;
  }
  return DSGD_OK;
}

// the statistics of a per-request step from the host-mapped mailbox its last kernel wrote: n_active as the difference
// against the value the host last saw (no memset in front of the request), the error flags as they are
static int finish_mail(dsgd_ctx* c, dsgd_batch_stats* stats, long long total, unsigned long long before, bool seq = false) {
  // (seq: the request's ONE workgroup wrote its sequence number behind the two words, with release order, as its last
  //  act -- the host polls the mapped word instead of waiting for the stream's completion signal; what the request
  //  enqueues next is ordered behind the kernel by the stream as always)
  bool seen = false;
  if (seq) {
    const volatile unsigned long long* m = c->h_mail;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int spin = 0; !seen; ++spin) {
      seen = __atomic_load_n(&m[2], __ATOMIC_ACQUIRE) == c->mail_seq;
      if (!seen && (spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
    }
  }
  if (!seen) HIP_TRY(hipStreamSynchronize(c->stream));
  const unsigned long long now = c->h_mail[0];
  const int err = (int)c->h_mail[1];
  c->ctr_last = now;
  c->ctr_known = err == 0;
  if (err) {   // (rare: the flags are cleared for the next request, as check_err_flag does)
    c->h_sc->err = err;
    return check_err_flag(c);
  }
  if (stats) {
    stats->n_samples = total;
    stats->n_active = (int64_t)(now - before);
  }
  return DSGD_OK;
}

int dsgd_sync_step(dsgd_ctx* c, const int32_t* const* idx_per_worker, const int64_t* n_per_worker, int32_t n_workers,
                   float lr, dsgd_batch_stats* stats) {
  DSGD_TRY(check_ctx(c));
  if (n_workers < 1 || !idx_per_worker || !n_per_worker) return fail(DSGD_EINVAL, "need at least one worker");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));
  DSGD_TRY(prepare_layout(c));
  long long mx = 0, tot = 0;
  {
    long long t = 0;
    for (int k = 0; k < n_workers; ++k) t += std::max<long long>(0, n_per_worker[k]);
    bool fits = c->req_plan && plan_kernel_ok(c, t, n_workers);
    for (int k = 0; k < n_workers && fits; ++k)
      fits = idx_per_worker[k] && list_fits_staged(c, idx_per_worker[k], n_per_worker[k]);
    if (fits && c->prof) {   // the reference's batch sizes: one persistent workgroup does the whole closure
      DSGD_TRY(reset_counters(c));
      DSGD_TRY(stage_lists(c, idx_per_worker, n_per_worker, n_workers, &mx, &tot));
      DSGD_TRY(launch_plan_kernel(c, c->cur_idx, c->d_segs, 0, 1, lr));
      return finish_stats(c, stats, tot);
    }
    if (fits) {   // ... its statistics through the host-mapped mailbox (see below)
      if (!c->ctr_known) DSGD_TRY(reset_counters(c));
      const unsigned long long before = c->ctr_last;
      DSGD_TRY(stage_lists(c, idx_per_worker, n_per_worker, n_workers, &mx, &tot));
      DSGD_TRY(launch_plan_kernel(c, c->cur_idx, c->d_segs, 0, 1, lr, true));
      return finish_mail(c, stats, tot, before, c->req_spin);
    }
  }
  {
    // the reference's own sizes (application.conf:15,27): the step laid out and run by the column-slice kernel, one launch
    int G = 0;
    if (cs_request_ok(c, idx_per_worker, n_per_worker, n_workers, &G) == DSGD_OK) {
      if (!c->ctr_known) DSGD_TRY(reset_counters(c));
      const unsigned long long before = c->ctr_last;
      DSGD_TRY(stage_lists(c, idx_per_worker, n_per_worker, n_workers, &mx, &tot));
      DSGD_TRY(launch_cs_request(c, G, n_workers, mx, lr));
      const int rc = finish_mail(c, stats, tot, before, c->req_spin);
      if (rc != 1) return rc;
      // (the step does not fit the one-step layout -- more than CS_MAX_SLOTS slots or 4,096 columns in one slice: nothing
      //  was applied; the row-parallel kernels below take it.  They work on the rank-ordered vector: the slice-major
      //  copy the request made is dropped -- left "live" it would overwrite their update at the next bind)
      DSGD_TRY(cs_unslice(c));
    }
  }
  DSGD_TRY(ensure_g(c, n_workers));
  DSGD_TRY(ensure_s(c, true));
  if (c->comm || c->prof) {   // (the collective path / profiling brackets: the plain read-back)
    DSGD_TRY(reset_counters(c));
    DSGD_TRY(stage_lists(c, idx_per_worker, n_per_worker, n_workers, &mx, &tot));
    DSGD_TRY(launch_grad(c, c->cur_idx, c->d_segs, n_workers, mx, true));
    DSGD_TRY(launch_finish_sync(c, n_workers, lr));
    return finish_stats(c, stats, tot);
  }
  // the request's statistics come back through the host-mapped mailbox the fused reduce writes (no memset in front, no
  // copy behind: two calls and ~7 us of GPU time per request)
  if (!c->ctr_known) DSGD_TRY(reset_counters(c));
  const unsigned long long before = c->ctr_last;
  DSGD_TRY(stage_lists(c, idx_per_worker, n_per_worker, n_workers, &mx, &tot));
  DSGD_TRY(launch_grad(c, c->cur_idx, c->d_segs, n_workers, mx, true));
  DSGD_TRY(launch_finish_sync(c, n_workers, lr, true));
  return finish_mail(c, stats, tot, before);
}

// (finish = false: stop behind the gradient kernels -- the caller splits the finish around a grouped collective)
static int ranges_enqueue(dsgd_ctx* c, const int64_t* row_begin, const int64_t* row_end, int n_workers, float lr,
                          long long* total, bool finish = true) {
  if (n_workers < 1 || !row_begin || !row_end) return fail(DSGD_EINVAL, "need at least one worker");
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));
  std::vector<WorkSeg> segs(n_workers);
  long long mx = 0, tot = 0;
  for (int k = 0; k < n_workers; ++k) {
    if (row_end[k] <= row_begin[k])
      return fail(DSGD_EINVAL, "worker %d has an empty row range: Vec.sum requires a non-empty list", k);
    if (row_begin[k] < 0 || row_end[k] > c->n_rows)
      return fail(DSGD_ERANGE, "worker %d range [%lld, %lld) outside the %lld loaded rows", k, (long long)row_begin[k],
                  (long long)row_end[k], c->n_rows);
    segs[k].begin = row_begin[k];
    segs[k].end = row_end[k];
    mx = std::max<long long>(mx, row_end[k] - row_begin[k]);
    tot += row_end[k] - row_begin[k];
  }
  DSGD_TRY(prepare_layout(c));
  DSGD_TRY(ensure_g(c, n_workers));
  DSGD_TRY(ensure_s(c, true));
  std::vector<StreamSeg> ssegs(n_workers);
  for (int k = 0; k < n_workers; ++k) ssegs[k] = make_sseg(row_begin[k], row_end[k]);
  // 10^3 .. 10^5 rows: column lists (csrc/dsgd_tcol.hpp) -- dot, column-wise gradient, reduce: no partials
  int trc = 1;
  long long ent = tot * 75;   // non-zeros of the ranges (the host's copy of the row offsets; else RCV1's mean row)
  if (c->h_row_ptr.size() == (size_t)c->n_rows + 1) {
    ent = 0;
    for (int k = 0; k < n_workers; ++k) ent += c->h_row_ptr[(size_t)row_end[k]] - c->h_row_ptr[(size_t)row_begin[k]];
  }
  const bool chunks_could = fstep_grid(c, ssegs, tot, ent) > 0;
  if (tcol_wanted(c, tot, n_workers) && !(chunks_could && ent > c->tcol_max_nnz)) {
    DSGD_TRY(upload_segs(c, segs));
    trc = launch_tcol(c, segs, mx);
    if (trc != DSGD_OK && trc != 1) return trc;
  }
  const int fwg = trc == 1 ? fstep_grid(c, ssegs, tot, ent) : 0;
  if (trc == DSGD_OK) {
  } else if (fwg > 0) {
    // shards of 10^4 .. 2 * 10^6 rows: row chunks, the three passes of the split streams in ONE launch (csrc/dsgd_fstep.hpp)
    DSGD_TRY(launch_fstep(c, ssegs, fwg));
  } else if (tot >= c->stream_min) {
    // whole contiguous ranges: the nnz-streaming kernel (coalesced 16-byte loads, no per-row latency chain)
    DSGD_TRY(launch_stream<true>(c, ssegs));  // (the profiling events bracket the main kernel only)
  } else {
    DSGD_TRY(upload_segs(c, segs));
    DSGD_TRY(launch_grad(c, nullptr, c->d_segs, n_workers, mx, true));
  }
  if (finish) DSGD_TRY(launch_finish_sync(c, n_workers, lr));
  *total = tot;
  return DSGD_OK;
}

int dsgd_sync_step_ranges(dsgd_ctx* c, const int64_t* row_begin, const int64_t* row_end, int32_t n_workers, float lr,
                          dsgd_batch_stats* stats) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(reset_counters(c));
  long long tot = 0;
  DSGD_TRY(ranges_enqueue(c, row_begin, row_end, n_workers, lr, &tot));
  return finish_stats(c, stats, tot);
}

int dsgd_sync_step_ranges_async(dsgd_ctx* c, const int64_t* row_begin, const int64_t* row_end, int32_t n_workers,
                                float lr) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  long long tot = 0;
  DSGD_TRY(ranges_enqueue(c, row_begin, row_end, n_workers, lr, &tot));
  c->pending_samples += tot;
  return DSGD_OK;
}

int dsgd_synchronize(dsgd_ctx* c, dsgd_batch_stats* stats) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  DSGD_TRY(read_scalars(c));
  DSGD_TRY(prof_collect(c));
  const long long act = (long long)c->h_sc->n_active;
  const int err = c->h_sc->err;
  const long long pending = c->pending_samples;
  c->pending_samples = 0;   // (whatever the outcome: the rows of a rejected run are not carried into the next report)
  DSGD_TRY(reset_counters(c));
  if ((err & (8 | 16)) && c->d_cs_sync) HIP_TRY(hipMemsetAsync(c->d_cs_sync, 0, sizeof(unsigned int) * 2, c->stream));   // the abort word, first
  if (err & 2)
    return fail(DSGD_ESTATE, "fixed-point gradient accumulator left its safe band; the steps since the last "
                             "synchronize are invalid");
  if (err & 8)
    return fail(DSGD_ESTATE, "the column-slice kernel's exchange between its workgroups timed out; the steps since the last "
                             "synchronize are invalid (DSGD_CS=0 selects the row-parallel kernels)");
  if (err) return fail(DSGD_ERANGE, "sample index / key outside the loaded data");
  if (stats) {
    stats->n_active = act;
    stats->n_samples = pending;
  }
  return DSGD_OK;
}

int dsgd_cache_trim(dsgd_ctx* c, int64_t keep_bytes, int64_t* held_out) {
  DSGD_TRY(check_ctx(c));
  if (keep_bytes < 0) return fail(DSGD_EINVAL, "negative keep_bytes");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  cache_trim(c, (size_t)keep_bytes, false);
  if (held_out) *held_out = (int64_t)c->cache_bytes;
  return DSGD_OK;
}

int dsgd_plan_create_n(dsgd_ctx* c, const int32_t* idx, int64_t n_idx, const int64_t* offsets, int64_t n_steps,
                       int32_t n_workers, dsgd_plan** out) {
  DSGD_TRY(check_ctx(c));
  if (!idx || !offsets || !out || n_steps < 1 || n_workers < 1 || n_idx < 0) return fail(DSGD_EINVAL, "bad plan arguments");
  if (n_steps > INT64_MAX / n_workers) return fail(DSGD_EINVAL, "n_steps * n_workers overflows");
  if (offsets[n_steps * n_workers] != n_idx)
    return fail(DSGD_EINVAL, "offsets end at %lld but idx holds %lld entries", (long long)offsets[n_steps * n_workers], (long long)n_idx);
  return dsgd_plan_create(c, idx, offsets, n_steps, n_workers, out);
}

// A plan's frame: offsets checked, the two device blocks taken from the cache, the list ranges uploaded (build stream).
// The caller fills p->d_idx on the build stream and calls plan_finish; on failure everything is given back.
static void plan_abandon(dsgd_ctx* c, dsgd_plan* p) {
  (void)hipStreamSynchronize(c->build_stream);
  cs_free(c, p);
  cache_give(c, p->d_idx, p->idx_bytes);
  cache_give(c, p->d_segs, p->segs_bytes);
  if (p->built_ev) (void)hipEventDestroy(p->built_ev);
  delete p;
}
static int plan_frame(dsgd_ctx* c, const int64_t* offsets, int64_t n_steps, int32_t n_workers, dsgd_plan** out) {
  const int64_t n_lists = n_steps * n_workers;
  if (offsets[0] != 0) return fail(DSGD_EINVAL, "offsets[0] must be 0");
  long long mx = 0;
  for (int64_t i = 0; i < n_lists; ++i) {
    if (offsets[i + 1] <= offsets[i])
      return fail(DSGD_EINVAL, "list %lld is empty: Vec.sum requires a non-empty list", (long long)i);
    mx = std::max<long long>(mx, offsets[i + 1] - offsets[i]);
  }
  dsgd_plan* p = new (std::nothrow) dsgd_plan();
  if (!p) return fail(DSGD_ENOMEM, "out of host memory");
  p->n_steps = n_steps;
  p->n_workers = n_workers;
  p->max_items = mx;
  p->offsets.assign(offsets, offsets + n_lists + 1);
  for (int64_t st = 0; st < n_steps; ++st)
    p->max_step_rows = std::max<long long>(p->max_step_rows, offsets[(st + 1) * n_workers] - offsets[st * n_workers]);
  p->fits_rows = c->n_rows;
  std::vector<WorkSeg> segs((size_t)n_lists);
  for (int64_t i = 0; i < n_lists; ++i) {
    segs[i].begin = offsets[i];
    segs[i].end = offsets[i + 1];
  }
  // the lists go up on the context's BUILD stream (beside whatever the launch stream is running: the next epoch's plan
  // can be set up while this epoch's steps run), into blocks earlier plans handed back
  int rc = ensure_build_stream(c);
  void *qi = nullptr, *qs = nullptr;
  const size_t ib = sizeof(int) * (size_t)offsets[n_lists], sb = sizeof(WorkSeg) * (size_t)n_lists;
  if (rc == DSGD_OK && (cache_take(c, &qi, ib, &p->idx_bytes) || cache_take(c, &qs, sb, &p->segs_bytes)))
    rc = fail(DSGD_ENOMEM, "out of device memory (plan of %lld index entries)", (long long)offsets[n_lists]);
  p->d_idx = static_cast<int*>(qi);
  p->d_segs = static_cast<WorkSeg*>(qs);
  if (rc == DSGD_OK && hipEventCreateWithFlags(&p->built_ev, hipEventDisableTiming) != hipSuccess) rc = fail(DSGD_EHIP, "hipEventCreate");
  hipError_t e = hipSuccess;
  if (rc == DSGD_OK) e = hipMemcpyAsync(p->d_segs, segs.data(), sb, hipMemcpyHostToDevice, c->build_stream);
  if (rc == DSGD_OK && e == hipSuccess) e = hipStreamSynchronize(c->build_stream);   // (segs is a local: staged before it goes)
  if (rc == DSGD_OK && e != hipSuccess) rc = fail(DSGD_EHIP, "plan upload: %s", hipGetErrorString(e));
  if (rc != DSGD_OK) {
    char msg[sizeof(g_err)];
    snprintf(msg, sizeof(msg), "%s", g_err);
    plan_abandon(c, p);
    return fail(rc, "%s", msg);
  }
  *out = p;
  return DSGD_OK;
}
// the lists are in p->d_idx (enqueued on the build stream): lay the plan out for the device, close its set-up
static int plan_finish(dsgd_ctx* c, dsgd_plan* p, dsgd_plan** out) {
  int rc = DSGD_OK;
  // the column slices of the reference's own step sizes are laid out NOW (by the device, on the build stream), not inside
  // the first dsgd_plan_run -- a call its callers time; without a column layout yet (no data / no dimSparsity) at the first run
  if (c->cs_enable && !c->comm && c->d_row_ptr && c->have_ds && !c->async_running) {
    rc = prepare_layout(c);
    if (rc == DSGD_OK) rc = cs_build(c, p);
  }
  if (rc == DSGD_OK && hipEventRecord(p->built_ev, c->build_stream) != hipSuccess) rc = fail(DSGD_EHIP, "hipEventRecord");
  if (rc != DSGD_OK) {
    char msg[sizeof(g_err)];
    snprintf(msg, sizeof(msg), "%s", g_err);
    plan_abandon(c, p);
    return fail(rc, "%s", msg);
  }
  p->built_pending = true;
  *out = p;
  return DSGD_OK;
}

int dsgd_plan_create(dsgd_ctx* c, const int32_t* idx, const int64_t* offsets, int64_t n_steps, int32_t n_workers,
                     dsgd_plan** out) {
  DSGD_TRY(check_ctx(c));
  if (!idx || !offsets || !out || n_steps < 1 || n_workers < 1) return fail(DSGD_EINVAL, "bad plan arguments");
  const int64_t n_lists = n_steps * n_workers;
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));   // (nothing here touches w: slice-major weights stay as they are)
  dsgd_plan* p = nullptr;
  DSGD_TRY(plan_frame(c, offsets, n_steps, n_workers, &p));
  p->h_idx.assign(idx, idx + offsets[n_lists]);
  p->fits = true;
  for (int64_t i = 0; i < n_lists && p->fits; ++i) p->fits = list_fits_staged(c, idx + offsets[i], offsets[i + 1] - offsets[i]);
  // (pageable sources: the runtime stages them and returns when they are staged; ordered on the build stream)
  hipError_t e = hipMemcpyAsync(p->d_idx, idx, sizeof(int) * (size_t)offsets[n_lists], hipMemcpyHostToDevice, c->build_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->build_stream);
  if (e != hipSuccess) {
    plan_abandon(c, p);
    return fail(DSGD_EHIP, "plan upload: %s", hipGetErrorString(e));
  }
  return plan_finish(c, p, out);
}

// slot `i` of the from-seed scratch with room for `bytes`
static hipError_t seed_scratch(dsgd_ctx* c, int i, size_t bytes, void** out) {
  if (c->seed_scratch_bytes[i] < bytes) {
    (void)hipFree(c->seed_scratch[i]);
    c->seed_scratch[i] = nullptr;
    c->seed_scratch_bytes[i] = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    const hipError_t e = hipMalloc(&c->seed_scratch[i], want);
    if (e != hipSuccess) return e;
    c->seed_scratch_bytes[i] = want;
  }
  *out = c->seed_scratch[i];
  return hipSuccess;
}
// ---- an epoch's lists drawn on the device, draw for draw the reference's stream (csrc/dsgd_shuffle.hpp) ---------------------
int dsgd_plan_create_from_seed(dsgd_ctx* c, uint64_t* jstate, const int64_t* split_begin, const int64_t* split_end, int32_t n_splits,
                               int64_t max_samples, int32_t batch_size, dsgd_plan** out, int64_t* n_steps_out, int64_t* draws_out) {
  DSGD_TRY(check_ctx(c));
  if (!jstate || !split_begin || !split_end || !out || !n_steps_out || n_splits < 1 || batch_size < 1)
    return fail(DSGD_EINVAL, "bad arguments");
  *out = nullptr;
  *n_steps_out = 0;
  if (draws_out) *draws_out = 0;
  long long max_len = 0;
  for (int k = 0; k < n_splits; ++k) {
    const long long len = split_end[k] - split_begin[k];
    if (len < 1 || split_begin[k] < 0) return fail(DSGD_EINVAL, "worker %d has no rows", k);
    max_len = std::max(max_len, len);
  }
  if (batch_size > JR_MAX_TAKE || max_len > JR_MAX_LEN)
    return fail(DSGD_EUNSUPPORTED, "device-drawn lists serve batches up to %d rows of splits up to %d rows (draw them on the host)", JR_MAX_TAKE, JR_MAX_LEN);
  // steps the reference runs before an empty slice (core/Master.scala:184-188; the slave's Vec.sum throws on it)
  long long n_steps = 0;
  for (long long b = 0; b < max_samples; b += batch_size) {
    bool ok = true;
    for (int k = 0; k < n_splits; ++k) ok = ok && b < split_end[k] - split_begin[k];
    if (!ok) break;
    ++n_steps;
  }
  if (n_steps == 0) return DSGD_OK;
  const long long n_shuf = n_steps * n_splits;
  std::vector<int64_t> offsets((size_t)n_shuf + 1, 0);
  std::vector<long long> start((size_t)n_shuf + 1, 0);   // nominal raw index of every shuffle's start (no rejections)
  for (long long q = 0; q < n_shuf; ++q) {
    const long long len = split_end[q % n_splits] - split_begin[q % n_splits], b = (q / n_splits) * (long long)batch_size;
    offsets[(size_t)q + 1] = offsets[(size_t)q] + std::min<long long>(batch_size, len - b);
    start[(size_t)q + 1] = start[(size_t)q] + (len >= 2 ? len - 1 : 0);
  }
  const long long nominal = start[(size_t)n_shuf];
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  for (int k = 0; k < n_splits; ++k)
    if (split_end[k] > c->n_rows) return fail(DSGD_ERANGE, "worker %d's rows [%lld, %lld) outside the %lld loaded", k, (long long)split_begin[k], (long long)split_end[k], c->n_rows);
  DSGD_TRY(ensure_build_stream(c));
  hipStream_t bs = c->build_stream;
  const unsigned long long s0 = *jstate & JR_MASK;
  const bool seed_prof = getenv("DSGD_SEED_PROF") != nullptr;   // (tuning: where an epoch's 14 ms go)
  const auto tp0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (seed_prof) fprintf(stderr, "[dsgd_plan_create_from_seed] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
  };
  // ---- pass A on the device: the candidates for a rejection; the walk over them here ----
  std::vector<JrShuf> shuf((size_t)n_shuf);
  std::vector<int> rej_list;
  long long rej_total = 0;
  {
    const unsigned int cand_min = (unsigned int)(0x80000000ULL - (unsigned long long)max_len);
    const int per_lane = (int)std::max<long long>(1, std::min<long long>(1024, (1LL << 30) / max_len));
    long long scan = nominal + 64 + nominal / 4096;
    for (int attempt = 0;; ++attempt) {
      const long long span = (long long)JR_SCAN_THREADS * per_lane;
      const long long n_wg = (scan + span - 1) / span;
      const long long cap = (long long)((double)scan * (double)max_len / 2147483648.0 * 2.0) + 65536;
      long long* d_ci = nullptr;
      unsigned int* d_cu = nullptr;
      unsigned long long *d_base = nullptr, *d_tot = nullptr;
      auto drop = [&]() {};   // (the buffers are the context's scratch: kept)
      hipError_t e = n_wg > 0x7fffffffLL ? hipErrorInvalidValue : seed_scratch(c, 0, sizeof(long long) * (size_t)cap, (void**)&d_ci);
      if (e == hipSuccess) e = seed_scratch(c, 1, sizeof(unsigned int) * (size_t)cap, (void**)&d_cu);
      if (e == hipSuccess) e = seed_scratch(c, 2, sizeof(unsigned long long) * (size_t)n_wg, (void**)&d_base);
      if (e == hipSuccess) e = seed_scratch(c, 3, sizeof(unsigned long long) * 2, (void**)&d_tot);
      if (e == hipSuccess) e = hipMemsetAsync(d_tot, 0, sizeof(unsigned long long) * 2, bs);
      std::vector<unsigned long long> base((size_t)n_wg);
      unsigned long long tot[2] = {0, 0};
      if (e == hipSuccess) {
        JrScanArgs sa;
        sa.s0 = s0;
        sa.scan = scan;
        sa.per_lane = per_lane;
        sa.cand_min = cand_min;
        sa.cap = cap;
        sa.cand_i = d_ci;
        sa.cand_u = d_cu;
        sa.wg_base = d_base;
        sa.total = d_tot;
        hipLaunchKernelGGL(dsgd_jr_scan_kernel, dim3((unsigned)n_wg), dim3(JR_SCAN_THREADS), 0, bs, sa);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, bs);
      if (e == hipSuccess) e = hipMemcpyAsync(base.data(), d_base, sizeof(unsigned long long) * (size_t)n_wg, hipMemcpyDeviceToHost, bs);
      if (e == hipSuccess) e = hipStreamSynchronize(bs);
      std::vector<long long> ci;
      std::vector<unsigned int> cu;
      if (e == hipSuccess && tot[1] == 0 && tot[0] > 0) {
        ci.resize((size_t)tot[0]);
        cu.resize((size_t)tot[0]);
        e = hipMemcpyAsync(ci.data(), d_ci, sizeof(long long) * ci.size(), hipMemcpyDeviceToHost, bs);
        if (e == hipSuccess) e = hipMemcpyAsync(cu.data(), d_cu, sizeof(unsigned int) * cu.size(), hipMemcpyDeviceToHost, bs);
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
      }
      lap("scan + candidates on the host");
      drop();
      lap("... scan buffers freed");
      if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DSGD_EHIP, "scanning the random stream: %s", hipGetErrorString(e));
      }
      if (tot[1] != 0) return fail(DSGD_EUNSUPPORTED, "more rejection candidates than the device scan holds (draw the lists on the host)");
      // walk the candidates in raw order (workgroup by workgroup, each block in order): `rej` = rejections so far.  Raw index
      // i serves draw d = i - rej of the epoch; draw d belongs to shuffle j (start[j] <= d < start[j + 1]), bound len_j - (d - start[j])
      long long rej = 0, j = 0;
      bool beyond = false;
      rej_list.clear();
      for (long long q = 0; q < n_shuf; ++q) shuf[(size_t)q] = JrShuf{start[(size_t)q], 0, 0};
      long long next_begin = 0;   // shuffles < next_begin have their record closed
      auto close_upto = [&](long long jj) {   // shuffles before jj start with the rejections counted so far ... hmm: see below
        for (; next_begin < jj; ++next_begin) {
          shuf[(size_t)next_begin].rej_end = (int)rej_list.size();
          if (next_begin + 1 < n_shuf) {
            shuf[(size_t)next_begin + 1].raw0 = start[(size_t)next_begin + 1] + rej;
            shuf[(size_t)next_begin + 1].rej_begin = (int)rej_list.size();
          }
        }
      };
      for (long long w = 0; w < n_wg && !beyond; ++w) {
        const unsigned long long bw = base[(size_t)w];
        const unsigned long long at = bw >> 16, cnt = bw & 0xffffULL;
        for (unsigned long long t = 0; t < cnt; ++t) {
          const long long d = ci[(size_t)(at + t)] - rej;
          if (d >= nominal) {
            beyond = true;
            break;
          }
          while (start[(size_t)j + 1] <= d) ++j;
          close_upto(j);
          const long long len = split_end[j % n_splits] - split_begin[j % n_splits];
          const unsigned int n = (unsigned int)(len - (d - start[(size_t)j]));
          if ((n & (n - 1)) != 0 && cu[(size_t)(at + t)] >= (0x80000000u / n) * n) {
            rej_list.push_back((int)(ci[(size_t)(at + t)] - shuf[(size_t)j].raw0));
            ++rej;
          }
        }
      }
      close_upto(n_shuf);
      if (nominal + rej > scan) {   // more rejections than the margin scanned: scan further and walk again
        if (attempt >= 3) return fail(DSGD_EUNSUPPORTED, "the random stream keeps outrunning its scan (draw the lists on the host)");
        scan = nominal + rej + 64 + rej / 8;
        continue;
      }
      rej_total = rej;
      lap("candidates walked");
      break;
    }
  }
  for (long long q = 0; q < n_shuf; ++q)
    if (shuf[(size_t)q].rej_end - shuf[(size_t)q].rej_begin > JR_MAX_REJ)
      return fail(DSGD_EUNSUPPORTED, "more than %d rejections inside one shuffle (draw the lists on the host)", JR_MAX_REJ);
  // ---- the plan's frame, then pass B straight into its index buffer ----
  dsgd_plan* p = nullptr;
  DSGD_TRY(plan_frame(c, offsets.data(), n_steps, n_splits, &p));
  lap("plan frame");
  p->idx_trusted = true;
  p->fits = false;   // (the one-workgroup kernel's test reads the lists on the host; these plans run on column slices)
  JrShuf* d_shuf = nullptr;
  int *d_rej = nullptr, *d_err = nullptr;
  long long *d_sb = nullptr, *d_off = nullptr;
  auto drop2 = [&]() {};   // (scratch of the context)
  std::vector<long long> sb2(2 * (size_t)n_splits);
  for (int k = 0; k < n_splits; ++k) {
    sb2[(size_t)k] = split_begin[k];
    sb2[(size_t)n_splits + (size_t)k] = split_end[k];
  }
  int h_err = 0;
  hipError_t e = seed_scratch(c, 4, sizeof(JrShuf) * (size_t)n_shuf, (void**)&d_shuf);
  if (e == hipSuccess) e = seed_scratch(c, 5, sizeof(int) * std::max<size_t>(1, rej_list.size()), (void**)&d_rej);
  if (e == hipSuccess) e = seed_scratch(c, 6, sizeof(int), (void**)&d_err);
  if (e == hipSuccess) e = seed_scratch(c, 7, sizeof(long long) * sb2.size(), (void**)&d_sb);
  if (e == hipSuccess) e = seed_scratch(c, 8, sizeof(long long) * offsets.size(), (void**)&d_off);
  if (e == hipSuccess) e = hipMemcpyAsync(d_shuf, shuf.data(), sizeof(JrShuf) * (size_t)n_shuf, hipMemcpyHostToDevice, bs);
  if (e == hipSuccess && !rej_list.empty()) e = hipMemcpyAsync(d_rej, rej_list.data(), sizeof(int) * rej_list.size(), hipMemcpyHostToDevice, bs);
  if (e == hipSuccess) e = hipMemsetAsync(d_err, 0, sizeof(int), bs);
  if (e == hipSuccess) e = hipMemcpyAsync(d_sb, sb2.data(), sizeof(long long) * sb2.size(), hipMemcpyHostToDevice, bs);
  if (e == hipSuccess) e = hipMemcpyAsync(d_off, offsets.data(), sizeof(long long) * offsets.size(), hipMemcpyHostToDevice, bs);
  if (e == hipSuccess) {
    static const JrAffine back_one = jr_inverse_step();
    static const JrAffine back_block = jr_power(back_one, JR_SLICE_THREADS);
    JrSliceArgs a;
    a.s0 = s0;
    a.back_block = back_block;
    a.back_one = back_one;
    a.shuf = d_shuf;
    a.rej = d_rej;
    a.split_begin = d_sb;
    a.split_end = d_sb + n_splits;
    a.offsets = d_off;
    a.idx_out = p->d_idx;
    a.n_splits = n_splits;
    a.batch_size = batch_size;
    a.err = d_err;
    hipLaunchKernelGGL(dsgd_jr_slice_kernel, dim3((unsigned)n_shuf), dim3(JR_SLICE_THREADS), jr_slice_lds_bytes(max_len), bs, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, bs);
  if (e == hipSuccess) e = hipStreamSynchronize(bs);
  lap("lists drawn (slice kernel)");
  drop2();
  lap("... slice buffers freed");
  if (e != hipSuccess || h_err) {
    (void)hipGetLastError();
    plan_abandon(c, p);
    if (h_err) return fail(DSGD_EUNSUPPORTED, "a shuffle left the limits of the device form (draw the lists on the host)");
    return fail(DSGD_EHIP, "drawing the lists: %s", hipGetErrorString(e));
  }
  DSGD_TRY(plan_finish(c, p, out));
  lap("plan laid out");
  *n_steps_out = n_steps;
  *jstate = jr_jump_dev(s0, (unsigned long long)(nominal + rej_total));
  if (draws_out) *draws_out = nominal + rej_total;
  return DSGD_OK;
}

// the lists of a plan as the device holds them (tests: the device-drawn lists against csrc/jrand.c's)
int dsgd_plan_read_lists(dsgd_ctx* c, dsgd_plan* p, int32_t* idx_out, int64_t n, int64_t* offsets_out, int64_t n_offsets) {
  DSGD_TRY(check_ctx(c));
  if (!p || n < 0 || n_offsets < 0 || (n > 0 && !idx_out) || (n_offsets > 0 && !offsets_out)) return fail(DSGD_EINVAL, "bad arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  const long long n_lists = (long long)p->n_steps * p->n_workers;
  if (n > p->offsets[(size_t)n_lists] || n_offsets > n_lists + 1) return fail(DSGD_EINVAL, "more entries asked for than the plan holds");
  if (p->built_pending && p->built_ev) HIP_TRY(hipEventSynchronize(p->built_ev));
  if (n) HIP_TRY(hipMemcpy(idx_out, p->d_idx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n_offsets; ++i) offsets_out[i] = p->offsets[(size_t)i];
  return DSGD_OK;
}

// gate decisions and regulariser scalars of a column-slice plan's steps on record (include/dsgd.h)
int dsgd_plan_record(dsgd_ctx* c, dsgd_plan* p, int32_t on) {
  DSGD_TRY(check_ctx(c));
  if (!p) return fail(DSGD_EINVAL, "null plan");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  HIP_TRY(hipStreamSynchronize(c->stream));
  cache_give(c, p->d_gate_rec, p->gate_bytes);
  cache_give(c, p->d_s_rec, p->s_bytes);
  p->d_gate_rec = nullptr;
  p->d_s_rec = nullptr;
  p->gate_words = 0;
  if (!on) return DSGD_OK;
  const int words = (int)((std::max<long long>(p->max_step_rows, 1) + 31) / 32);
  void *g = nullptr, *sv = nullptr;
  if (cache_take(c, &g, sizeof(unsigned int) * (size_t)p->n_steps * words, &p->gate_bytes) ||
      cache_take(c, &sv, sizeof(float) * (size_t)p->n_steps, &p->s_bytes)) {
    cache_give(c, g, p->gate_bytes);
    return fail(DSGD_ENOMEM, "out of device memory (plan record)");
  }
  p->d_gate_rec = static_cast<unsigned int*>(g);
  p->d_s_rec = static_cast<float*>(sv);
  p->gate_words = words;
  // (blocks from the cache were ordered against the BUILD stream: order them against the launch stream, which clears them)
  HIP_TRY(hipStreamSynchronize(c->build_stream ? c->build_stream : c->stream));
  HIP_TRY(hipMemsetAsync(p->d_gate_rec, 0, sizeof(unsigned int) * (size_t)p->n_steps * words, c->stream));
  HIP_TRY(hipMemsetAsync(p->d_s_rec, 0, sizeof(float) * (size_t)p->n_steps, c->stream));
  return DSGD_OK;
}

int dsgd_plan_read_record(dsgd_ctx* c, dsgd_plan* p, int64_t step_begin, int64_t step_end, uint32_t* gate_mask, float* s_used,
                          int32_t* mask_words_out) {
  DSGD_TRY(check_ctx(c));
  if (!p) return fail(DSGD_EINVAL, "null plan");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  if (mask_words_out) *mask_words_out = p->gate_words;
  if (!gate_mask && !s_used) return DSGD_OK;
  if (!p->d_gate_rec) return fail(DSGD_ESTATE, "the plan keeps no record (dsgd_plan_record)");
  if (!(p->cs_ok && p->cs_layout == c->layout_gen))
    return fail(DSGD_EUNSUPPORTED, "only plans that run on column slices keep a record (see dsgd_plan_info)");
  if (step_begin < 0 || step_end > p->n_steps || step_end < step_begin) return fail(DSGD_EINVAL, "steps outside the plan");
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t n = (size_t)(step_end - step_begin);
  if (gate_mask && n)
    HIP_TRY(hipMemcpy(gate_mask, p->d_gate_rec + (size_t)step_begin * p->gate_words, sizeof(unsigned int) * n * (size_t)p->gate_words,
                      hipMemcpyDeviceToHost));
  if (s_used && n) HIP_TRY(hipMemcpy(s_used, p->d_s_rec + step_begin, sizeof(float) * n, hipMemcpyDeviceToHost));
  return DSGD_OK;
}

#ifdef DSGD_TEST_COLLECTIVE_SEAM
// TEST BUILDS ONLY (tests/rccl_stub/libdsgd_hip_seam.so; the product library has no such symbol): from step `from_step`
// (1-based, counted inside each launch) on, slice 1 of every column-slice launch withholds its granules; 0 = off.
extern "C" int dsgd_test_cs_skip_publish(dsgd_ctx* c, int32_t from_step) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  c->cs_test_skip = from_step < 0 ? 0 : from_step;
  return DSGD_OK;
}
#endif

int dsgd_plan_info(dsgd_ctx* c, dsgd_plan* p, int32_t* vals, int32_t n) {
  DSGD_TRY(check_ctx(c));
  if (!p || !vals || n < 0 || n > 8) return fail(DSGD_EINVAL, "bad plan_info arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  const bool cs = c->cs_enable && !c->comm && p->cs_ok && p->cs_layout == c->layout_gen;
  const bool one_wg = !cs && plan_kernel_ok(c, p->max_step_rows, p->n_workers) && p->fits && p->fits_rows == c->n_rows;
  const bool vt = !cs && !one_wg && c->vt_enable && p->vt_ok && p->vt_layout == c->layout_gen;
  const int32_t all[8] = {cs ? 1 : (one_wg ? 2 : (vt ? 3 : (p->cs_layout == c->layout_gen || p->vt_layout == c->layout_gen ? 4 : 0))),
                          cs ? p->cs_G : 0, cs ? p->cs_slot_stride : 0, cs ? p->cs_row_stride : 0, cs ? p->cs_cl_stride : 0,
                          cs ? p->cs_spl : 0, (cs && p->cs_device_built) ? 1 : 0, p->gate_words};
  for (int i = 0; i < n; ++i) vals[i] = all[i];
  return DSGD_OK;
}

int dsgd_plan_destroy(dsgd_ctx* c, dsgd_plan* p) {
  DSGD_TRY(check_ctx(c));
  if (!p) return DSGD_OK;
  std::lock_guard<std::mutex> lk(c->mu);
  c->comm_broken_ok = true;   // (a context whose communicator was given up can still let go of its plans)
  const int brc = bind(c, true);
  c->comm_broken_ok = false;
  DSGD_TRY(brc);
  // the plan's blocks go back to the context's cache behind everything the launch stream still holds (an event per block:
  // no hipFree, no device synchronisation -- an epoch of the reference is one plan); a set-up that never ran is ordered
  // in front of that first
  if (p->built_pending) HIP_TRY(hipStreamWaitEvent(c->stream, p->built_ev, 0));
  cache_give(c, p->d_idx, p->idx_bytes);
  cache_give(c, p->d_segs, p->segs_bytes);
  cs_free(c, p);
  cache_give(c, p->d_gate_rec, p->gate_bytes);
  cache_give(c, p->d_s_rec, p->s_bytes);
  if (p->d_vt_lanes || p->d_vt_segs || p->d_vt_long || p->d_vt_packed) {   // (the larger steps' tiles: allocated per plan)
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(p->d_vt_lanes);
    (void)hipFree(p->d_vt_segs);
    (void)hipFree(p->d_vt_long);
    (void)hipFree(p->d_vt_packed);
  }
  if (p->built_ev) {
    // (the event may still be pending on the build stream: destroying a recorded event is allowed, its resources go when it completes)
    (void)hipEventDestroy(p->built_ev);
  }
  delete p;
  return DSGD_OK;
}

int dsgd_plan_run(dsgd_ctx* c, dsgd_plan* p, int64_t step_begin, int64_t step_end, float lr) {
  DSGD_TRY(check_ctx(c));
  if (!p) return fail(DSGD_EINVAL, "null plan");
  if (step_begin < 0 || step_end > p->n_steps || step_end < step_begin)
    return fail(DSGD_EINVAL, "steps [%lld, %lld) outside the plan's %lld steps", (long long)step_begin, (long long)step_end,
                p->n_steps);
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));
  DSGD_TRY(prepare_layout(c));
  // the reference's own batch sizes: column slices, the whole range of steps in ONE launch (csrc/dsgd_cs.hpp)
  if (c->cs_enable && !c->comm) {
    if (p->cs_layout != c->layout_gen) {   // (created before the column layout existed, or the layout changed since)
      DSGD_TRY(cs_build(c, p));
      if (p->cs_device_built && c->build_stream) {
        HIP_TRY(hipEventRecord(p->built_ev, c->build_stream));
        p->built_pending = true;
      }
    }
  }
  if (p->built_pending) {   // the plan's set-up ran beside the launch stream: wait for it, once
    HIP_TRY(hipStreamWaitEvent(c->stream, p->built_ev, 0));
    p->built_pending = false;
  }
  if (c->cs_enable && !c->comm) {
    if (p->cs_ok && p->cs_layout == c->layout_gen) {
      if (step_end > step_begin) DSGD_TRY(launch_cs(c, p, step_begin, step_end, lr));
      c->pending_samples += p->offsets[step_end * p->n_workers] - p->offsets[step_begin * p->n_workers];
      return DSGD_OK;
    }
  }
  DSGD_TRY(cs_unslice(c));   // (bound with the slice-major weights kept: the row-parallel kernels below read d_w)
  if (plan_kernel_ok(c, p->max_step_rows, p->n_workers) && p->fits && p->fits_rows == c->n_rows) {
    if (step_end > step_begin) DSGD_TRY(launch_plan_kernel(c, p->d_idx, p->d_segs, step_begin, step_end, lr));
    c->pending_samples += p->offsets[step_end * p->n_workers] - p->offsets[step_begin * p->n_workers];
    return DSGD_OK;
  }
  DSGD_TRY(ensure_g(c, p->n_workers));
  DSGD_TRY(ensure_s(c, true));
  if (c->vt_enable && p->vt_layout != c->layout_gen) DSGD_TRY(vt_build(c, p));
  const bool vt = c->vt_enable && p->vt_ok && p->vt_layout == c->layout_gen;
  for (int64_t s = step_begin; s < step_end; ++s) {
    if (vt) {   // the lists as virtual tiles over the split streams
      DSGD_TRY(launch_grad_vt(c, p, s));
      DSGD_TRY(launch_finish_sync(c, p->n_workers, lr));
      c->pending_samples += p->offsets[(s + 1) * p->n_workers] - p->offsets[s * p->n_workers];
      continue;
    }
    const WorkSeg* segs = p->d_segs + s * p->n_workers;
    long long mx = 0;
    for (int k = 0; k < p->n_workers; ++k)
      mx = std::max<long long>(mx, p->offsets[s * p->n_workers + k + 1] - p->offsets[s * p->n_workers + k]);
    DSGD_TRY(launch_grad(c, p->d_idx, segs, p->n_workers, mx, true));
    DSGD_TRY(launch_finish_sync(c, p->n_workers, lr));
    c->pending_samples += p->offsets[(s + 1) * p->n_workers] - p->offsets[s * p->n_workers];
  }
  return DSGD_OK;
}

int dsgd_forward(dsgd_ctx* c, const float* w, const int32_t* idx, int64_t n, float* pred_out) {
  DSGD_TRY(check_ctx(c));
  if (n < 0 || (n > 0 && (!idx || !pred_out))) return fail(DSGD_EINVAL, "bad forward arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  if (w) DSGD_TRY(require_sync_mode(c));   // replacing the weights under the lock-free engine is refused
  DSGD_TRY(prepare_layout(c));
  if (w) DSGD_TRY(set_weights_locked(c, w));
  if (n == 0) return DSGD_OK;  // samplesIdx.map over an empty Seq is an empty reply (ref: core/Slave.scala:133)
  DSGD_TRY(ensure_idx(c, n));
  DSGD_TRY(reset_counters(c));
  if (n > c->pred_cap) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->d_pred) HIP_TRY(hipFree(c->d_pred));
    c->d_pred = nullptr;
    c->pred_cap = 0;
    const long long cap = std::max<long long>(n, 4096);
    HIP_TRY(hipMalloc(&c->d_pred, sizeof(float) * (size_t)cap));
    c->pred_cap = cap;
  }
  float* d_pred = c->d_pred;
  DSGD_TRY(pin_acquire(c->pin_idx, sizeof(int) * (size_t)n));
  memcpy(c->pin_idx.p, idx, sizeof(int) * (size_t)n);
  HIP_TRY(hipMemcpyAsync(c->d_idx, c->pin_idx.p, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
  DSGD_TRY(pin_sent(c, c->pin_idx));
  DSGD_TRY(pin_acquire(c->pin_out, sizeof(float) * (size_t)n));
  const int G = c->group;
  dim3 grid(grid_for(c, n, G));
  CsrView m = view(c);
  switch (G) {
    case 64: hipLaunchKernelGGL(dsgd_forward_kernel<64>, grid, dim3(256), 0, c->stream, m, c->d_w, c->d_idx, (long long)n, d_pred, c->d_sc); break;
    case 32: hipLaunchKernelGGL(dsgd_forward_kernel<32>, grid, dim3(256), 0, c->stream, m, c->d_w, c->d_idx, (long long)n, d_pred, c->d_sc); break;
    case 16: hipLaunchKernelGGL(dsgd_forward_kernel<16>, grid, dim3(256), 0, c->stream, m, c->d_w, c->d_idx, (long long)n, d_pred, c->d_sc); break;
    default: hipLaunchKernelGGL(dsgd_forward_kernel<8>, grid, dim3(256), 0, c->stream, m, c->d_w, c->d_idx, (long long)n, d_pred, c->d_sc); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(c->pin_out.p, d_pred, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  DSGD_TRY(read_scalars(c));
  memcpy(pred_out, c->pin_out.p, sizeof(float) * (size_t)n);
  return check_err_flag(c);
}

// evaluation in three parts around the all-reduce of its tallies (one host thread, several contexts: dsgd_loss_acc_devices)
static int eval_enqueue(dsgd_ctx* c, const float* w, int64_t row_begin, int64_t row_end) {
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  if (row_end <= row_begin)  // samples.map(...).reduce on an empty collection throws (ref: SparseSVM.scala:21-23)
    return fail(DSGD_EINVAL, "empty row range: reduce on an empty sample list");
  if (row_begin < 0 || row_end > c->n_rows)
    return fail(DSGD_ERANGE, "rows [%lld, %lld) outside the %lld loaded rows", (long long)row_begin, (long long)row_end, c->n_rows);
  if (w) DSGD_TRY(require_sync_mode(c));
  DSGD_TRY(prepare_layout(c));
  if (w) DSGD_TRY(set_weights_locked(c, w));
  // |w|^2 of the loss: cached with s while the synchronous kernels own w; while the lock-free engine runs w moves
  // under the cache, so every check recomputes it (MasterAsync's leaky loss check, core/MasterAsync.scala:96-162,
  // compares successive losses).  The engine keeps its own s (HogState), so refreshing d_sc here disturbs nothing.
  if (c->async_running || c->nsq_dirty) c->s_dirty = true;
  DSGD_TRY(ensure_s(c));  // also refreshes |w|^2
  if (c->async_running) c->s_dirty = true;
  DSGD_TRY(reset_counters(c));
  if (!c->async_running && row_end - row_begin >= 4096) {
    std::vector<StreamSeg> ssegs(1, make_sseg(row_begin, row_end));
    DSGD_TRY(launch_stream<false>(c, ssegs));
  } else {
    const int G = c->group;
    // beside the persistent Hogwild engine (2 waves x 232 VGPRs per SIMD, 136 KB of LDS) only ONE more wave per SIMD
    // and 16 KB of LDS fit a CU: 256-lane blocks with a small weight tile, one per CU
    const int bs = c->async_running ? 256 : 1024;
    const long long groups_per_block = bs / G;
    const long long rows = row_end - row_begin;
    dim3 grid((unsigned)std::max<long long>(1, std::min<long long>(c->n_cu, (rows + groups_per_block - 1) / groups_per_block)));
    // small ranges do not amortise staging 160 KiB of weights per workgroup: shrink the LDS tile
    const int hw = std::min(c->async_running ? std::min(c->hw_eval, 4096) : (rows >= 4096 ? c->hw_eval : std::min(c->hw_eval, 1024)),
                            DSGD_LDS_FLOATS - 4);
    const size_t lds = sizeof(float) * (size_t)(hw + 4);   // (+ the workgroup's three tally words)
    CsrView m = view(c);
    switch (G) {
      case 64: hipLaunchKernelGGL(dsgd_eval_kernel<64>, grid, dim3(bs), lds, c->stream, m, c->d_w, (long long)row_begin, (long long)row_end, c->d_sc, hw); break;
      case 32: hipLaunchKernelGGL(dsgd_eval_kernel<32>, grid, dim3(bs), lds, c->stream, m, c->d_w, (long long)row_begin, (long long)row_end, c->d_sc, hw); break;
      case 16: hipLaunchKernelGGL(dsgd_eval_kernel<16>, grid, dim3(bs), lds, c->stream, m, c->d_w, (long long)row_begin, (long long)row_end, c->d_sc, hw); break;
      default: hipLaunchKernelGGL(dsgd_eval_kernel<8>, grid, dim3(bs), lds, c->stream, m, c->d_w, (long long)row_begin, (long long)row_end, c->d_sc, hw); break;
    }
    HIP_TRY(hipGetLastError());
  }
  return DSGD_OK;
}
static int eval_collective(dsgd_ctx* c) {
  if (!c->comm) return DSGD_OK;
  // shard-wise evaluation: three tallies + row count summed over ranks
  RCCL_TRY(rccl::AllReduce(c->d_sc->counts, c->d_sc->counts, 4, rccl::kInt64, rccl::kSum, c->comm, c->stream));
  return DSGD_OK;
}
static int eval_read(dsgd_ctx* c, double* loss, double* acc, int64_t* counts) {
  long long tallies[4] = {0, 0, 0, 0};
  DSGD_TRY(read_scalars(c));
  for (int i = 0; i < 4; ++i) tallies[i] = (long long)c->h_sc->counts[i];
  const double n = (double)tallies[3];
  if (loss) *loss = c->cfg.lambda * (double)c->h_sc->wnorm2 + ((double)tallies[1] + 2.0 * (double)tallies[2]) / n;
  if (acc) *acc = (double)tallies[0] / n;
  if (counts) {
    counts[0] = tallies[0];
    counts[1] = tallies[1];
    counts[2] = tallies[2];
  }
  return DSGD_OK;
}

int dsgd_loss_acc(dsgd_ctx* c, const float* w, int64_t row_begin, int64_t row_end, double* loss, double* acc,
                  int64_t* counts) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(eval_enqueue(c, w, row_begin, row_end));
  DSGD_TRY(eval_collective(c));
  return eval_read(c, loss, acc, counts);
}

int dsgd_async_step(dsgd_ctx* c, const int32_t* idx, int64_t n, float lr, float* delta_out, dsgd_batch_stats* stats) {
  DSGD_TRY(check_ctx(c));
  if (n <= 0 || !idx) return fail(DSGD_EINVAL, "Cannot sum an empty list of vectors");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  DSGD_TRY(require_sync_mode(c));
  DSGD_TRY(prepare_layout(c));
  DSGD_TRY(ensure_s(c));
  DSGD_TRY(reset_counters(c));
  long long mx = 0, tot = 0;
  const int64_t nn = n;
  DSGD_TRY(stage_lists(c, &idx, &nn, 1, &mx, &tot));
  DSGD_TRY(launch_grad(c, c->cur_idx, c->d_segs, 1, mx));
  hipLaunchKernelGGL(dsgd_async_finish_kernel, dim3(1), dim3(1024), 0, c->stream, c->d_w, c->d_g, c->dp, c->d_ds, (float)n,
                     lr, (float)c->cfg.lambda, delta_out ? c->d_tmp : (float*)nullptr, c->d_sc);
  HIP_TRY(hipGetLastError());
  c->s_dirty = false;
  if (delta_out) {
    DSGD_TRY(launch_permute_out(c, c->d_tmp, c->d_io));
    HIP_TRY(hipMemcpyAsync(delta_out, c->d_io, sizeof(float) * c->dp, hipMemcpyDeviceToHost, c->stream));
  }
  return finish_stats(c, stats, tot);
}

int dsgd_update_grad(dsgd_ctx* c, const int32_t* key, const float* dv, int64_t nnz) {
  DSGD_TRY(check_ctx(c));
  if (nnz < 0 || (nnz > 0 && (!key || !dv))) return fail(DSGD_EINVAL, "bad update arguments");
  if (nnz == 0) return DSGD_OK;
  for (int64_t i = 0; i < nnz; ++i)   // the keys are host data: validated here, no device-side error flag to wait for
    if (key[i] < 0 || key[i] >= c->dp) return fail(DSGD_ERANGE, "key %d at position %lld outside [0, %d]", key[i], (long long)i, c->dp - 1);
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  // The reference applies a peer's update while its own asyncTask runs (core/Slave.scala:177-185 is an RPC handler on
  // another pool thread).  Here: pinned staging + a persistent device buffer (hipMalloc / hipFree per call -- hipFree
  // synchronises the DEVICE, i.e. waits for the persistent engine to reach its update budget), the copy and the kernel
  // on a side stream while the engine is resident, atomic adds to w, and the engine's incremental regulariser scalar
  // told about the foreign update.
  if (!c->upd_stream) HIP_TRY(hipStreamCreateWithFlags(&c->upd_stream, hipStreamNonBlocking));
  if (!c->d_upd_key) {
    const long long cap = std::max<long long>(c->dp, 4096);   // a Sparse delta holds at most D + 1 entries; longer inputs go in pieces
    HIP_TRY(hipMalloc(&c->d_upd_key, sizeof(int) * (size_t)cap));
    HIP_TRY(hipMalloc(&c->d_upd_dv, sizeof(float) * (size_t)cap));
    c->upd_cap = cap;
  }
  // (an engine that has reached its budget but has not been joined yet is not live: its kernel has exited, nothing adds
  //  to w any more and nobody reads HogState::s_reg -- the update takes the plain path with the Sparse filter pass)
  const bool live = c->async_running && !(c->exch_done.load() && hipStreamQuery(c->async_stream) == hipSuccess);
  hipStream_t st = live ? c->upd_stream : c->stream;
  for (int64_t o = 0; o < nnz; o += c->upd_cap) {
    const long long n = std::min<long long>(c->upd_cap, nnz - o);
    DSGD_TRY(pin_acquire(c->pin_upd, (sizeof(int) + sizeof(float)) * (size_t)n));
    int* pk = static_cast<int*>(c->pin_upd.p);
    float* pv = reinterpret_cast<float*>(pk + n);
    memcpy(pk, key + o, sizeof(int) * (size_t)n);
    memcpy(pv, dv + o, sizeof(float) * (size_t)n);
    HIP_TRY(hipMemcpyAsync(c->d_upd_key, pk, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c->d_upd_dv, pv, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, st));
    DSGD_TRY(pin_sent(c, c->pin_upd, st));
    const int blocks = (int)std::min<long long>((n + 255) / 256, 64);
    hipLaunchKernelGGL(dsgd_update_grad_kernel, dim3(blocks), dim3(256), 0, st, c->d_w, c->d_perm, c->d_ds, c->d_upd_key,
                       c->d_upd_dv, n, (float)c->cfg.lambda, live ? &c->d_hog->s_reg : (float*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  if (!live) {
    // the Sparse filter pass rewrites every w[j] non-atomically: only while no engine is adding to w
    hipLaunchKernelGGL(dsgd_filter_kernel, dim3((c->dp + 255) / 256), dim3(256), 0, st, c->d_w, c->dp);
    HIP_TRY(hipGetLastError());
  }
  c->s_dirty = true;
  HIP_TRY(hipStreamSynchronize(st));   // the update is applied when the call returns (the RPC's Ack, proto.proto:40)
  return DSGD_OK;
}

// ---- Hogwild persistent engine -------------------------------------------------------------------------
static int async_refresh(dsgd_ctx* c) {  // copy the engine's counters to the host without touching its stream
  HIP_TRY(hipMemcpyAsync(c->h_hog, c->d_hog, sizeof(HogState), hipMemcpyDeviceToHost, c->query_stream));
  HIP_TRY(hipStreamSynchronize(c->query_stream));
  return DSGD_OK;
}

// the master's loss check must become resident beside the engine: its 16 KiB weight tile and the engine's LDS share a CU
// (RCV1: D + 1 = 47,237)
static_assert(sizeof(float) * ((size_t)hog_lds_words(HOG_HL, HOG_WL, 47237) + 4096 + 64) <= 160 * 1024, "Hogwild + eval LDS");
// one launch of the persistent kernel on async_stream: workers run until the TOTAL update count reaches max_updates
static int hog_launch(dsgd_ctx* c, long long max_updates) {
  HogArgs a;
  a.m = view(c);
  a.w = c->d_w;
  a.ds = c->d_ds;
  a.gcold = c->d_gcold;
  a.asg_begin = c->d_asg;
  a.asg_end = c->d_asg + c->hog_n;
  a.it = c->d_hog_it;
  a.st = c->d_hog;
  a.max_updates = max_updates;
  a.seed = c->hog_seed;
  a.lr = c->hog_lr;
  a.lambda = (float)c->cfg.lambda;
  int bits = 0;
  while ((1 << bits) < c->hog_batch) ++bits;
  const int shift = 30 - bits;   // at most one contribution per row and column: sums stay below 2^30
  a.qscale = std::ldexp(1.0f, shift - c->vexp);
  a.inv_qscale = std::ldexp(1.0f, c->vexp - shift);
  a.batch = c->hog_batch;
  a.positional_bug = c->hog_bug;
  a.hl = std::min(c->dp, c->hog_hl) & ~3;   // (whole quads of ranks; the rest is cold strip)
  a.dp = c->dp;
  a.wl = std::min(c->hog_wl, c->dp) & ~255;
  a.hh = hog_hh(a.hl);
  a.direct = c->hog_n <= HOG_DIRECT_MAX ? 1 : 0;
  a.trace = c->trace_cap > 0 ? c->d_trace : nullptr;
  a.trace_cap = c->trace_cap;
  a.tdot = c->d_tdot;
  const size_t lds = sizeof(float) * (size_t)hog_lds_words(a.hl, a.wl, c->dp);
  a.tprof = c->d_tprof;
  const dim3 grid(c->hog_n), block(HOG_THREADS);
  if (a.trace) {
    if (c->d_tprof) hipLaunchKernelGGL((dsgd_hogwild_kernel<true, true>), grid, block, lds, c->async_stream, a);
    else hipLaunchKernelGGL((dsgd_hogwild_kernel<false, true>), grid, block, lds, c->async_stream, a);
  } else {
    if (c->d_tprof) hipLaunchKernelGGL((dsgd_hogwild_kernel<true, false>), grid, block, lds, c->async_stream, a);
    else hipLaunchKernelGGL((dsgd_hogwild_kernel<false, false>), grid, block, lds, c->async_stream, a);
  }
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}

// one round of the cross-GPU asynchronous mode on async_stream: local updates up to `upto`, then
// d_local = w_prev - w (what this replica subtracted since the last exchange); all-reduce; the peers' part
// d_sum - d_local is subtracted on top (a replica applies its own updates as it goes and its peers' updates when they
// arrive: core/Slave.scala:99-105,177-185).  With one rank the peers' part is exactly zero.
static int exchange_round(dsgd_ctx* c, long long upto) {
  const int blocks = (c->dp + 1023) / 1024;
  DSGD_TRY(hog_launch(c, upto));
  hipLaunchKernelGGL(dsgd_exchange_delta_kernel, dim3(blocks), dim3(1024), 0, c->async_stream, c->d_w, c->d_wprev,
                     c->d_wdelta, c->d_wdelta + c->dp, c->dp);
  HIP_TRY(hipGetLastError());
  RCCL_TRY(rccl::AllReduce(c->d_wdelta, c->d_wdelta, (size_t)c->dp, rccl::kFloat32, rccl::kSum, c->comm, c->async_stream));
  hipLaunchKernelGGL(dsgd_exchange_apply_kernel, dim3(1), dim3(1024), 0, c->async_stream, c->d_w, c->d_wprev,
                     c->d_wdelta, c->d_wdelta + c->dp, c->d_ds, c->dp, (float)c->cfg.lambda, c->d_hog, c->world);
  HIP_TRY(hipGetLastError());
  return DSGD_OK;
}

int dsgd_async_start(dsgd_ctx* c, const int64_t* assigned_begin, const int64_t* assigned_end, int32_t n_workers, int32_t batch,
                     float lr, int64_t max_updates, uint64_t seed, int32_t positional_bug) {
  DSGD_TRY(check_ctx(c));
  if (!assigned_begin || !assigned_end || n_workers < 1) return fail(DSGD_EINVAL, "need at least one worker");
  if (batch < 1 || batch > HOG_MAX_BATCH) return fail(DSGD_EINVAL, "batch %d outside [1, %d]", batch, HOG_MAX_BATCH);
  if (max_updates < 0) return fail(DSGD_EINVAL, "negative max_updates");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_ds(c));
  // ref: core/Slave.scala:161 "Async computation already running, can't be initialized unless stopped first"
  if (c->async_running) return fail(DSGD_ESTATE, "async computation already running: stop it first");
  if (n_workers > c->n_cu)   // every worker is a resident workgroup (they never yield): one per CU at most
    return fail(DSGD_EINVAL, "%d workers on a device with %d compute units", n_workers, c->n_cu);
  std::vector<long long> asg(2 * (size_t)n_workers);
  for (int k = 0; k < n_workers; ++k) {
    const long long b = assigned_begin[k], e = assigned_end[k];
    if (e <= b) return fail(DSGD_EINVAL, "worker %d has no assigned samples", k);
    if (b < 0 || e > c->n_rows) return fail(DSGD_ERANGE, "worker %d range [%lld, %lld) outside the %lld loaded rows", k, b, e, c->n_rows);
    if (batch > e - b) return fail(DSGD_EINVAL, "batch %d larger than worker %d's %lld assigned samples", batch, k, e - b);
    if (positional_bug && e - b > c->n_rows) return fail(DSGD_ERANGE, "positional sampling outside the data");
    asg[k] = b;
    asg[n_workers + k] = e;
  }
  const bool exchange = c->comm && c->exchange_every > 0;
  long long n_rounds = 1;
  if (exchange) {
    // ref: core/Slave.scala:103-105 gossips every update to every peer; across GPUs the replicas exchange the SUM of
    // their updates every `exchange_every` local updates (dsgd_async_set_exchange) -- all ranks must enqueue the same
    // number of collectives, so the update budget has to be finite and identical on every rank.
    n_rounds = (max_updates + c->exchange_every - 1) / c->exchange_every;
    if (n_rounds < 1) n_rounds = 1;
    if (n_rounds > 65536)
      return fail(DSGD_EINVAL, "max_updates / exchange_every = %lld rounds: more than 65536 (give a finite update budget)", n_rounds);
  }
  DSGD_TRY(prepare_layout(c));
  DSGD_TRY(ensure_s(c));
  if (!c->async_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->async_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->query_stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->d_hog, sizeof(HogState)));
    HIP_TRY(hipHostMalloc(&c->h_hog, sizeof(HogState), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&c->h_one, sizeof(int), hipHostMallocDefault));
    *c->h_one = 1;
  }
  const int hl = std::min(c->dp, c->hog_hl) & ~3;
  const size_t strip = (size_t)std::max(1, c->dp - hl);
  if (n_workers > c->hog_workers) {
    (void)hipFree(c->d_gcold);
    (void)hipFree(c->d_asg);
    (void)hipFree(c->d_hog_it);
    c->d_gcold = nullptr;
    c->d_asg = nullptr;
    c->d_hog_it = nullptr;
    HIP_TRY(hipMalloc(&c->d_gcold, sizeof(float) * (size_t)n_workers * strip));
    HIP_TRY(hipMalloc(&c->d_asg, sizeof(long long) * 2 * (size_t)n_workers));
    HIP_TRY(hipMalloc(&c->d_hog_it, sizeof(unsigned long long) * (size_t)n_workers));
    c->hog_workers = n_workers;
  }
  if (c->trace_cap > 0) {   // traced run: records of hog_trace_words(batch) words (header, gate masks, one x . w per row)
    c->trace_mw = (batch + 31) / 32;
    c->trace_batch = batch;
    const long long need = c->trace_cap * hog_trace_words(batch);
    if (need > c->trace_words) {
      (void)hipFree(c->d_trace);
      c->d_trace = nullptr;
      c->trace_words = 0;
      HIP_TRY(hipMalloc(&c->d_trace, sizeof(unsigned int) * (size_t)need));
      c->trace_words = need;
    }
    const long long need_dot = (long long)n_workers * batch;
    if (need_dot > c->tdot_words) {
      (void)hipFree(c->d_tdot);
      c->d_tdot = nullptr;
      c->tdot_words = 0;
      HIP_TRY(hipMalloc(&c->d_tdot, sizeof(float) * (size_t)need_dot));
      c->tdot_words = need_dot;
    }
  }
  if (exchange && !c->d_wprev) {
    HIP_TRY(hipMalloc(&c->d_wprev, sizeof(float) * c->dp));
    HIP_TRY(hipMalloc(&c->d_wdelta, sizeof(float) * 2 * c->dp));
  }
  HIP_TRY(hipMemsetAsync(c->d_gcold, 0, sizeof(float) * (size_t)n_workers * strip, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_hog_it, 0, sizeof(unsigned long long) * (size_t)n_workers, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_asg, asg.data(), sizeof(long long) * asg.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->d_hog, 0, sizeof(HogState), c->stream));
  HIP_TRY(hipMemcpyAsync(&c->d_hog->s_reg, &c->d_sc->s_reg, sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  if (exchange) HIP_TRY(hipMemcpyAsync(c->d_wprev, c->d_w, sizeof(float) * c->dp, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // asg is a stack-lifetime host buffer; everything is in place before launch
  c->hog_n = n_workers;
  c->hog_batch = batch;
  c->hog_lr = lr;
  c->hog_seed = seed;
  c->hog_bug = positional_bug;
  c->exch_rc = DSGD_OK;
  c->exch_err.clear();
  if (!exchange) {
    DSGD_TRY(hog_launch(c, max_updates));
  } else {
    // The rounds are enqueued by a helper thread: a HIP queue holds a bounded number of launches (and a collective
    // may block its caller), so enqueuing thousands of rounds here would hold the context's mutex for most of the run
    // and make dsgd_async_updates / dsgd_async_stop wait behind it.
    if (c->exch_thread.joinable()) c->exch_thread.join();
    c->exch_done.store(false);
    const long long every = c->exchange_every;
    c->exch_thread = std::thread([c, n_rounds, max_updates, every]() {
      int rc = hipSetDevice(c->cfg.device) == hipSuccess ? DSGD_OK : fail(DSGD_EHIP, "hipSetDevice in the exchange thread");
      for (long long r = 0; r < n_rounds && rc == DSGD_OK; ++r) rc = exchange_round(c, std::min<long long>(max_updates, (r + 1) * every));
      if (rc != DSGD_OK) {
        c->exch_rc = rc;
        c->exch_err = g_err;   // (thread-local message of THIS thread: carried to the joining caller)
      }
      c->exch_done.store(true);
    });
  }
  c->async_running = true;
  c->s_dirty = true;
  return DSGD_OK;
}

int dsgd_async_set_exchange(dsgd_ctx* c, int64_t every_updates) {
  DSGD_TRY(check_ctx(c));
  if (every_updates < 0) return fail(DSGD_EINVAL, "negative exchange period");
  std::lock_guard<std::mutex> lk(c->mu);
  if (c->async_running) return fail(DSGD_ESTATE, "async computation running");
  c->exchange_every = every_updates;
  return DSGD_OK;
}

int dsgd_async_updates(dsgd_ctx* c, int64_t* updates, int32_t* running) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (!c->d_hog) {
    if (updates) *updates = 0;
    if (running) *running = 0;
    return DSGD_OK;
  }
  DSGD_TRY(async_refresh(c));
  if (updates) *updates = (int64_t)c->h_hog->updates;
  const bool busy = c->async_running && (!c->exch_done.load() || hipStreamQuery(c->async_stream) == hipErrorNotReady);
  if (running) *running = busy ? 1 : 0;
  return DSGD_OK;
}

// raise the engine's stop flag: a 4-byte copy on the side stream (the kernel polls device memory, not the PCIe bus)
static int hog_raise_stop(dsgd_ctx* c) {
  if (!c->d_hog || !c->h_one) return DSGD_OK;
  HIP_TRY(hipMemcpyAsync(&c->d_hog->stop, c->h_one, sizeof(int), hipMemcpyHostToDevice, c->query_stream));
  HIP_TRY(hipStreamSynchronize(c->query_stream));
  return DSGD_OK;
}

static int async_join(dsgd_ctx* c) {
  if (c->exch_thread.joinable()) c->exch_thread.join();   // (it never takes the context's mutex)
  HIP_TRY(hipStreamSynchronize(c->async_stream));
  if (c->upd_stream) HIP_TRY(hipStreamSynchronize(c->upd_stream));
  c->async_running = false;
  c->s_dirty = true;  // w moved under the scalar the synchronous kernels cache
  if (c->exch_rc != DSGD_OK) return fail(c->exch_rc, "exchange round: %s", c->exch_err.c_str());
  DSGD_TRY(async_refresh(c));
  if (c->h_hog->err) return fail(DSGD_ERANGE, "the lock-free engine sampled a row outside the loaded data");
  return DSGD_OK;
}

// Block until the engine has left the device -- WITHOUT the context's mutex: dsgd_update_grad, dsgd_loss_acc and
// dsgd_async_updates are RPC handlers of other pool threads in the reference (core/Slave.scala:177-185 runs while
// asyncTask does) and must not queue up behind a waiter for as long as the update budget lasts.  One thread does the
// blocking; a second waiter sleeps on the condition variable.
static int async_wait_released(dsgd_ctx* c, std::unique_lock<std::mutex>& lk) {
  if (c->join_in_progress) {
    c->join_cv.wait(lk, [c] { return !c->join_in_progress; });
    // the joining thread's result is every waiter's result (a failed run must not read as a success to the thread that
    // happened to come second)
    if (c->join_rc != DSGD_OK) return fail(c->join_rc, "%s", c->join_err.c_str());
    return DSGD_OK;
  }
  c->join_in_progress = true;
  hipStream_t st = c->async_stream;
  lk.unlock();
  if (c->exch_thread.joinable()) c->exch_thread.join();   // (it never takes the context's mutex)
  const hipError_t e = hipStreamSynchronize(st);
  lk.lock();
  int rc = DSGD_OK;
  if (e != hipSuccess) rc = fail(DSGD_EHIP, "hipStreamSynchronize(async stream): %s", hipGetErrorString(e));
  else if (c->async_running) rc = async_join(c);   // (its own synchronisations return at once now)
  c->join_rc = rc;
  c->join_err = rc == DSGD_OK ? "" : g_err;
  c->join_in_progress = false;   // (cleared LAST: dsgd_destroy and the other waiters go on only when the result is stored)
  c->join_cv.notify_all();
  return rc;
}

int dsgd_async_stop(dsgd_ctx* c) {  // ref: SlaveImpl.stopAsync, core/Slave.scala:187-195
  DSGD_TRY(check_ctx(c));
  std::unique_lock<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (!c->async_running) return DSGD_OK;
  DSGD_TRY(hog_raise_stop(c));
  return async_wait_released(c, lk);
}

int dsgd_async_stats(dsgd_ctx* c, int64_t* counters, double* s_engine, double* s_exact) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (!c->d_hog) return fail(DSGD_ESTATE, "the lock-free engine has never been started on this context");
  DSGD_TRY(async_refresh(c));
  if (counters) {
    counters[0] = (int64_t)c->h_hog->updates;
    counters[1] = (int64_t)c->h_hog->samples;
    counters[2] = (int64_t)c->h_hog->active;
    counters[3] = (int64_t)c->h_hog->atomics;
  }
  if (s_engine) *s_engine = (double)c->h_hog->s_reg;
  if (s_exact) {
    // the same summation the synchronous kernels use (fra_scalars), from the weights as they are now; the engine keeps
    // its own scalar (HogState), so refreshing the synchronous one disturbs nothing
    DSGD_TRY(require_ds(c));
    c->s_dirty = true;
    DSGD_TRY(ensure_s(c));
    DSGD_TRY(read_scalars(c));
    if (c->async_running) c->s_dirty = true;
    *s_exact = (double)c->h_sc->s_reg;
  }
  return DSGD_OK;
}

int dsgd_async_set_trace(dsgd_ctx* c, int64_t capacity) {
  DSGD_TRY(check_ctx(c));
  if (capacity < 0) return fail(DSGD_EINVAL, "negative trace capacity");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (c->async_running) return fail(DSGD_ESTATE, "async computation running");
  c->trace_cap = capacity;   // (the buffer is sized at dsgd_async_start: a record's length depends on the batch size)
  if (capacity == 0) {
    (void)hipFree(c->d_trace);   // (no engine is resident: hipFree's device synchronisation returns)
    (void)hipFree(c->d_tdot);
    c->d_trace = nullptr;
    c->d_tdot = nullptr;
    c->trace_words = 0;
    c->tdot_words = 0;
    c->trace_mw = 0;
    c->trace_batch = 0;
  }
  return DSGD_OK;
}

int dsgd_async_read_trace(dsgd_ctx* c, int32_t* worker, uint32_t* iteration, int64_t* read_at, float* s_used,
                          int32_t* n_active, uint32_t* gate_mask, int64_t n, int64_t* n_out, int32_t* mask_words_out) {
  DSGD_TRY(check_ctx(c));
  if (n < 0 || (n > 0 && (!worker || !iteration || !read_at || !s_used || !n_active || !gate_mask)))
    return fail(DSGD_EINVAL, "bad trace arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (c->async_running) return fail(DSGD_ESTATE, "async computation running (dsgd_async_wait / dsgd_async_stop first)");
  if (!c->d_trace || !c->d_hog || c->trace_mw == 0)
    return fail(DSGD_ESTATE, "no traced run on this context (dsgd_async_set_trace, then dsgd_async_start)");
  DSGD_TRY(async_refresh(c));
  const long long have = std::min<long long>((long long)c->h_hog->updates, c->trace_cap);
  const long long m = std::min<long long>(have, n);
  const int mw = c->trace_mw, rw = (int)hog_trace_words(c->trace_batch);
  if (n_out) *n_out = have;
  if (mask_words_out) *mask_words_out = mw;
  if (m == 0) return DSGD_OK;
  std::vector<unsigned int> recs;
  try {
    recs.resize((size_t)m * (size_t)rw);
  } catch (const std::bad_alloc&) {
    return fail(DSGD_ENOMEM, "out of host memory");
  }
  HIP_TRY(hipMemcpy(recs.data(), c->d_trace, sizeof(unsigned int) * recs.size(), hipMemcpyDeviceToHost));
  for (long long i = 0; i < m; ++i) {
    const unsigned int* r = recs.data() + (size_t)i * (size_t)rw;
    worker[i] = (int32_t)r[0];
    iteration[i] = r[1];
    read_at[i] = (int64_t)(((unsigned long long)r[3] << 32) | r[2]);
    memcpy(&s_used[i], &r[4], sizeof(float));
    n_active[i] = (int32_t)r[5];
    memcpy(gate_mask + (size_t)i * (size_t)mw, r + HOG_TRACE_HDR, sizeof(unsigned int) * (size_t)mw);
  }
  return DSGD_OK;
}

int dsgd_async_read_trace_dots(dsgd_ctx* c, int64_t* seen_from, float* dots, int64_t n, int32_t* batch_out) {
  DSGD_TRY(check_ctx(c));
  if (n < 0 || (n > 0 && (!seen_from || !dots))) return fail(DSGD_EINVAL, "bad trace arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (c->async_running) return fail(DSGD_ESTATE, "async computation running (dsgd_async_wait / dsgd_async_stop first)");
  if (!c->d_trace || !c->d_hog || c->trace_mw == 0)
    return fail(DSGD_ESTATE, "no traced run on this context (dsgd_async_set_trace, then dsgd_async_start)");
  DSGD_TRY(async_refresh(c));
  const long long have = std::min<long long>((long long)c->h_hog->updates, c->trace_cap);
  const long long m = std::min<long long>(have, n);
  const int B = c->trace_batch, mw = c->trace_mw, rw = (int)hog_trace_words(B);
  if (batch_out) *batch_out = B;
  if (m == 0) return DSGD_OK;
  std::vector<unsigned int> recs;
  try {
    recs.resize((size_t)m * (size_t)rw);
  } catch (const std::bad_alloc&) {
    return fail(DSGD_ENOMEM, "out of host memory");
  }
  HIP_TRY(hipMemcpy(recs.data(), c->d_trace, sizeof(unsigned int) * recs.size(), hipMemcpyDeviceToHost));
  for (long long i = 0; i < m; ++i) {
    const unsigned int* r = recs.data() + (size_t)i * (size_t)rw;
    const unsigned long long read_at = ((unsigned long long)r[3] << 32) | r[2];
    // the low word of seen_from, at most 2^32 - 1 updates below read_at; a launch's first iteration knows only the run's start
    seen_from[i] = (r[7] & 1u) ? 0 : (int64_t)(read_at - (unsigned long long)((unsigned int)read_at - r[6]));
    memcpy(dots + (size_t)i * (size_t)B, r + HOG_TRACE_HDR + mw, sizeof(float) * (size_t)B);
  }
  return DSGD_OK;
}

int dsgd_async_wait(dsgd_ctx* c) {
  DSGD_TRY(check_ctx(c));
  std::unique_lock<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (!c->async_running) return DSGD_OK;
  return async_wait_released(c, lk);
}

int dsgd_comm_unique_id(char* id_out) {
  if (!id_out) return fail(DSGD_EINVAL, "null id_out");
  if (!rccl::available()) return fail(DSGD_ERCCL, "librccl could not be loaded: %s", dlerror());
  rccl::unique_id_t id;
  RCCL_TRY(rccl::GetUniqueId(&id));
  memcpy(id_out, id.internal, DSGD_UNIQUE_ID_BYTES);
  return DSGD_OK;
}

int dsgd_comm_init(dsgd_ctx* c, const char* unique_id, int32_t world_size, int32_t rank) {
  DSGD_TRY(check_ctx(c));
  if (!unique_id || world_size < 1 || rank < 0 || rank >= world_size) return fail(DSGD_EINVAL, "bad communicator arguments");
  if (!rccl::available()) return fail(DSGD_ERCCL, "librccl could not be loaded");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  if (c->comm) return fail(DSGD_ESTATE, "communicator already attached");
  rccl::unique_id_t id;
  memcpy(id.internal, unique_id, DSGD_UNIQUE_ID_BYTES);
  RCCL_TRY(rccl::CommInitRank(&c->comm, world_size, id, rank));
  c->comm_broken = false;
  c->world = world_size;
  c->rank = rank;
  return DSGD_OK;
}

int dsgd_comm_destroy(dsgd_ctx* c) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  c->comm_broken_ok = true;
  const int brc = bind(c);
  c->comm_broken_ok = false;
  DSGD_TRY(brc);
  c->comm_broken = false;   // (an aborted communicator is gone already: the context may attach a new one)
  if (!c->comm) {
    c->world = 1;
    c->rank = 0;
    return DSGD_OK;
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  RCCL_TRY(rccl::CommDestroy(c->comm));
  c->comm = nullptr;
  c->world = 1;
  c->rank = 0;
  return DSGD_OK;
}

// ---- ONE host thread driving several contexts -------------------------------------------------------------------------
// The reference's dev role runs the master and every slave in ONE JVM (Main.scala:144-158): one thread of that JVM must be
// able to step all the GPUs of a node.  A collective blocks its caller until every rank has joined, so the per-context
// entry points above cannot be called one after the other from one thread; these take all the contexts at once: every
// context's kernels in front of a collective are enqueued first, then all the collectives inside one
// ncclGroupStart / ncclGroupEnd, then what follows them -- the same kernels, sums and order as N processes, one per GPU.
}  // extern "C" (the helpers below are C++)
namespace {
struct MultiLock {
  std::vector<dsgd_ctx*> held;
  int lock(dsgd_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) return fail(DSGD_EINVAL, "need at least one context");
    std::vector<dsgd_ctx*> order(ctxs, ctxs + n);
    for (dsgd_ctx* c : order)
      if (!c) return fail(DSGD_EINVAL, "null context");
    std::sort(order.begin(), order.end());   // one locking order for every caller
    if (std::adjacent_find(order.begin(), order.end()) != order.end()) return fail(DSGD_EINVAL, "a context appears twice");
    for (dsgd_ctx* c : order) {
      c->mu.lock();
      held.push_back(c);
    }
    return DSGD_OK;
  }
  ~MultiLock() {
    for (auto it = held.rbegin(); it != held.rend(); ++it) (*it)->mu.unlock();
  }
};
// every context of a grouped call joins the SAME collectives: communicators of one size, attached to all or to none
int group_shape(dsgd_ctx* const* ctxs, int n) {
  for (int i = 0; i < n; ++i) {
    if ((ctxs[i]->comm != nullptr) != (ctxs[0]->comm != nullptr) || ctxs[i]->world != ctxs[0]->world)
      return fail(DSGD_ESTATE, "the contexts of a grouped call must share one communicator shape (dsgd_comm_init_all)");
    if (ctxs[i]->layout_ready != ctxs[0]->layout_ready)
      return fail(DSGD_ESTATE, "some contexts have their column layout and some do not: load the data of all of them first");
  }
  if (n > 1 && !ctxs[0]->comm) return fail(DSGD_ESTATE, "several contexts without a communicator (dsgd_comm_init_all)");
  return DSGD_OK;
}
// phase 2 of every grouped call: fn(context) enqueues that context's collective
template <class Fn>
int grouped(dsgd_ctx* const* ctxs, int n, Fn fn) {
  if (!ctxs[0]->comm) return DSGD_OK;
  // all or nothing: whatever can be refused is refused BEFORE the group opens (a context that cannot be bound, a
  // communicator a failed group left behind) -- a collective that only some ranks joined would hang every stream
  for (int i = 0; i < n; ++i) {
    if (ctxs[i]->comm_broken)
      return fail(DSGD_ERCCL, "context %d: its communicator was abandoned after a collective that not every rank joined "
                              "(dsgd_comm_destroy + dsgd_comm_init_all)", i);
    DSGD_TRY(bind(ctxs[i]));
  }
  RCCL_TRY(rccl::GroupStart());
  int rc = DSGD_OK, joined = 0;
  for (int i = 0; i < n && rc == DSGD_OK; ++i) {
    rc = bind(ctxs[i]);
    if (rc == DSGD_OK) rc = fn(ctxs[i], i);
    if (rc == DSGD_OK) ++joined;
  }
  char first_err[sizeof(g_err)];
  snprintf(first_err, sizeof(first_err), "%s", g_err);
  const int re = rccl::GroupEnd();   // (always: a group that was started is ended)
  if (rc != DSGD_OK && joined > 0) {
    // some ranks' collectives are enqueued and the others' never will be: abort the communicators (their streams come
    // back with an error instead of hanging) and refuse further collectives on them
    for (int i = 0; i < n; ++i) {
      ctxs[i]->comm_broken = true;
      if (rccl::CommAbort && ctxs[i]->comm) {
        (void)rccl::CommAbort(ctxs[i]->comm);
        ctxs[i]->comm = nullptr;
      }
    }
    return fail(rc, "%s (the group's communicators were aborted: only %d of %d ranks had joined)", first_err, joined, n);
  }
  if (rc == DSGD_OK && re != 0) rc = fail(DSGD_ERCCL, "ncclGroupEnd: %s", rccl::GetErrorString(re));
  return rc;
}
}  // namespace
extern "C" {

int dsgd_comm_init_all(dsgd_ctx* const* ctxs, int32_t n_ctx) {
  MultiLock ml;
  DSGD_TRY(ml.lock(ctxs, n_ctx));
  if (!rccl::available()) return fail(DSGD_ERCCL, "librccl could not be loaded");
  for (int i = 0; i < n_ctx; ++i)
    if (ctxs[i]->comm) return fail(DSGD_ESTATE, "communicator already attached to context %d", i);
  rccl::unique_id_t id;
  RCCL_TRY(rccl::GetUniqueId(&id));
  RCCL_TRY(rccl::GroupStart());
  int rc = DSGD_OK;
  for (int i = 0; i < n_ctx && rc == DSGD_OK; ++i) {
    rc = bind(ctxs[i]);
    if (rc == DSGD_OK) {
      const int r = rccl::CommInitRank(&ctxs[i]->comm, n_ctx, id, i);
      if (r) rc = fail(DSGD_ERCCL, "ncclCommInitRank(rank %d of %d): %s", i, n_ctx, rccl::GetErrorString(r));
    }
  }
  const int re = rccl::GroupEnd();
  if (rc == DSGD_OK && re != 0) rc = fail(DSGD_ERCCL, "ncclGroupEnd: %s", rccl::GetErrorString(re));
  if (rc != DSGD_OK) {
    for (int i = 0; i < n_ctx; ++i) ctxs[i]->comm = nullptr;   // (whatever was created is abandoned: the call failed as a whole)
    return rc;
  }
  for (int i = 0; i < n_ctx; ++i) {
    ctxs[i]->world = n_ctx;
    ctxs[i]->rank = i;
  }
  return DSGD_OK;
}

int dsgd_build_dim_sparsity_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* n_train_per_ctx) {
  MultiLock ml;
  DSGD_TRY(ml.lock(ctxs, n_ctx));
  if (!n_train_per_ctx) return fail(DSGD_EINVAL, "null n_train_per_ctx");
  DSGD_TRY(group_shape(ctxs, n_ctx));
  std::vector<unsigned int*> cnt((size_t)n_ctx, nullptr);
  auto drop = [&]() {
    for (auto& p : cnt) {
      (void)hipFree(p);
      p = nullptr;
    }
  };
  int rc = DSGD_OK;
  for (int i = 0; i < n_ctx && !rc; ++i) {
    rc = bind(ctxs[i]);
    if (!rc) rc = ds_checks(ctxs[i], n_train_per_ctx[i]);
  }
  // the column ranking: counts, their sum over the contexts, one order for all
  for (int i = 0; i < n_ctx && !rc; ++i) {
    rc = bind(ctxs[i]);
    if (!rc) rc = layout_begin(ctxs[i], &cnt[(size_t)i]);
  }
  if (!rc) rc = grouped(ctxs, n_ctx, [&](dsgd_ctx* c, int i) { return layout_collective(c, cnt[(size_t)i]); });
  for (int i = 0; i < n_ctx && !rc; ++i) {
    rc = bind(ctxs[i]);
    if (!rc) {
      rc = layout_finish(ctxs[i], cnt[(size_t)i]);   // (takes the buffer)
      cnt[(size_t)i] = nullptr;
    }
  }
  // dimSparsity: feature counts of every context's train rows, summed (Main.scala:57-60 counts the whole train set)
  for (int i = 0; i < n_ctx && !rc; ++i) {
    rc = bind(ctxs[i]);
    if (!rc) rc = ds_begin(ctxs[i], n_train_per_ctx[i], &cnt[(size_t)i]);
  }
  if (!rc) rc = grouped(ctxs, n_ctx, [&](dsgd_ctx* c, int i) { return ds_collective(c, cnt[(size_t)i]); });
  for (int i = 0; i < n_ctx && !rc; ++i) {
    rc = bind(ctxs[i]);
    if (!rc) {
      rc = ds_finish(ctxs[i], cnt[(size_t)i], nullptr);
      cnt[(size_t)i] = nullptr;
    }
  }
  drop();
  return rc;
}

}  // extern "C"
static int devices_step_checks(dsgd_ctx* const* ctxs, int n_ctx) {
  DSGD_TRY(group_shape(ctxs, n_ctx));
  for (int i = 0; i < n_ctx; ++i) {
    dsgd_ctx* c = ctxs[i];
    DSGD_TRY(bind(c));
    DSGD_TRY(require_data(c));
    DSGD_TRY(require_ds(c));
    DSGD_TRY(require_sync_mode(c));
    if (!c->layout_ready) return fail(DSGD_ESTATE, "context %d has no column layout yet (dsgd_build_dim_sparsity_devices)", i);
  }
  return DSGD_OK;
}
static int devices_step_finish(dsgd_ctx* const* ctxs, int n_ctx, int n_workers, float lr, const std::vector<long long>& totals,
                               dsgd_batch_stats* stats) {
  DSGD_TRY(grouped(ctxs, n_ctx, [](dsgd_ctx* c, int) { return finish_collective(c); }));
  for (int i = 0; i < n_ctx; ++i) {
    DSGD_TRY(bind(ctxs[i]));
    DSGD_TRY(finish_post(ctxs[i], n_workers, lr));
  }
  dsgd_batch_stats sum{0, 0};
  for (int i = 0; i < n_ctx; ++i) {
    DSGD_TRY(bind(ctxs[i]));
    dsgd_batch_stats one{0, 0};
    DSGD_TRY(finish_stats(ctxs[i], &one, totals[(size_t)i]));
    sum.n_samples += one.n_samples;
    sum.n_active += one.n_active;
  }
  if (stats) *stats = sum;
  return DSGD_OK;
}
extern "C" {

int dsgd_sync_step_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int32_t* const* idx_per_worker,
                           const int64_t* n_per_worker, int32_t workers_per_ctx, float lr, dsgd_batch_stats* stats) {
  MultiLock ml;
  DSGD_TRY(ml.lock(ctxs, n_ctx));
  if (workers_per_ctx < 1 || !idx_per_worker || !n_per_worker) return fail(DSGD_EINVAL, "need at least one worker per context");
  DSGD_TRY(devices_step_checks(ctxs, n_ctx));
  std::vector<long long> totals((size_t)n_ctx, 0);
  for (int i = 0; i < n_ctx; ++i) {   // every context's gradient kernels and its part of the finish in front of the collective
    dsgd_ctx* c = ctxs[i];
    DSGD_TRY(bind(c));
    DSGD_TRY(ensure_g(c, workers_per_ctx));
    DSGD_TRY(ensure_s(c, true));
    DSGD_TRY(reset_counters(c));
    long long mx = 0;
    DSGD_TRY(stage_lists(c, idx_per_worker + (size_t)i * workers_per_ctx, n_per_worker + (size_t)i * workers_per_ctx, workers_per_ctx,
                         &mx, &totals[(size_t)i]));
    DSGD_TRY(launch_grad(c, c->cur_idx, c->d_segs, workers_per_ctx, mx, true));
    DSGD_TRY(finish_pre(c, workers_per_ctx, lr, false));
  }
  return devices_step_finish(ctxs, n_ctx, workers_per_ctx, lr, totals, stats);
}

int dsgd_sync_step_ranges_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* row_begin, const int64_t* row_end,
                                  int32_t workers_per_ctx, float lr, dsgd_batch_stats* stats) {
  MultiLock ml;
  DSGD_TRY(ml.lock(ctxs, n_ctx));
  if (workers_per_ctx < 1 || !row_begin || !row_end) return fail(DSGD_EINVAL, "need at least one worker per context");
  DSGD_TRY(devices_step_checks(ctxs, n_ctx));
  std::vector<long long> totals((size_t)n_ctx, 0);
  for (int i = 0; i < n_ctx; ++i) {
    dsgd_ctx* c = ctxs[i];
    DSGD_TRY(bind(c));
    DSGD_TRY(reset_counters(c));
    DSGD_TRY(ranges_enqueue(c, row_begin + (size_t)i * workers_per_ctx, row_end + (size_t)i * workers_per_ctx, workers_per_ctx, lr,
                            &totals[(size_t)i], false));
    DSGD_TRY(finish_pre(c, workers_per_ctx, lr, false));
  }
  return devices_step_finish(ctxs, n_ctx, workers_per_ctx, lr, totals, stats);
}

int dsgd_loss_acc_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* row_begin, const int64_t* row_end, double* loss,
                          double* acc, int64_t* counts) {
  MultiLock ml;
  DSGD_TRY(ml.lock(ctxs, n_ctx));
  if (!row_begin || !row_end) return fail(DSGD_EINVAL, "null row ranges");
  DSGD_TRY(group_shape(ctxs, n_ctx));
  for (int i = 0; i < n_ctx; ++i) {
    DSGD_TRY(bind(ctxs[i]));
    if (!ctxs[i]->layout_ready) return fail(DSGD_ESTATE, "context %d has no column layout yet (dsgd_build_dim_sparsity_devices)", i);
    DSGD_TRY(eval_enqueue(ctxs[i], nullptr, row_begin[i], row_end[i]));
  }
  DSGD_TRY(grouped(ctxs, n_ctx, [](dsgd_ctx* c, int) { return eval_collective(c); }));
  for (int i = n_ctx - 1; i >= 0; --i) {   // (every replica holds the summed tallies and the same weights: context 0 answers)
    DSGD_TRY(bind(ctxs[i]));
    DSGD_TRY(eval_read(ctxs[i], i == 0 ? loss : nullptr, i == 0 ? acc : nullptr, i == 0 ? counts : nullptr));
  }
  return DSGD_OK;
}

int dsgd_prof_enable(dsgd_ctx* c, int32_t on) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  HIP_TRY(hipStreamSynchronize(c->stream));
  DSGD_TRY(prof_collect(c));
  c->prof = on != 0;
  c->prof_main_only = on == 2;   // 2: bracket only the dominant gradient kernel (two event records per step, not six)
  return DSGD_OK;
}

int dsgd_prof_read(dsgd_ctx* c, double* ms_avg, int64_t* n_launches, int32_t reset) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  HIP_TRY(hipStreamSynchronize(c->stream));
  DSGD_TRY(prof_collect(c));
  if (ms_avg) *ms_avg = c->prof_kn[0] ? c->prof_kms[0] / (double)c->prof_kn[0] : 0.0;
  if (n_launches) *n_launches = c->prof_kn[0];
  if (reset) {
    for (int k = 0; k < 3; ++k) {
      c->prof_kms[k] = 0.0;
      c->prof_kn[k] = 0;
    }
  }
  return DSGD_OK;
}

int dsgd_prof_read_kinds(dsgd_ctx* c, double* ms_avg3, int64_t* n_launches3) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  HIP_TRY(hipStreamSynchronize(c->stream));
  DSGD_TRY(prof_collect(c));
  for (int k = 0; k < 3; ++k) {
    if (ms_avg3) ms_avg3[k] = c->prof_kn[k] ? c->prof_kms[k] / (double)c->prof_kn[k] : 0.0;
    if (n_launches3) n_launches3[k] = c->prof_kn[k];
  }
  return DSGD_OK;
}

int dsgd_range_nnz(dsgd_ctx* c, int64_t row_begin, int64_t row_end, int64_t* nnz, int64_t* cold_nnz) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  DSGD_TRY(require_data(c));
  if (row_begin < 0 || row_end < row_begin || row_end > c->n_rows) return fail(DSGD_ERANGE, "row range [%lld, %lld) outside [0, %lld)", (long long)row_begin, (long long)row_end, c->n_rows);
  DSGD_TRY(prepare_layout(c));
  if (nnz) *nnz = c->h_row_ptr[(size_t)row_end] - c->h_row_ptr[(size_t)row_begin];
  if (cold_nnz) {
    const bool split = c->h_crow_ptr.size() == (size_t)c->n_rows + 1;
    *cold_nnz = split ? c->h_crow_ptr[(size_t)row_end] - c->h_crow_ptr[(size_t)row_begin] : 0;
  }
  return DSGD_OK;
}

int dsgd_debug_cycles(dsgd_ctx* c, uint64_t* out8, int32_t reset) {
  DSGD_TRY(check_ctx(c));
  if (!out8) return fail(DSGD_EINVAL, "null out8");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c, true));
  for (int i = 0; i < 16; ++i) out8[i] = 0;
  if (!c->d_tprof) return DSGD_OK;
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(out8, c->d_tprof, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost));
  if (const char* path = getenv("DSGD_FSTEP_DUMP")) {   // tuning runs: the per-workgroup words of the chunked launch's last run
    std::vector<unsigned long long> v(4 * 1024);
    HIP_TRY(hipMemcpy(v.data(), c->d_tprof + 16, sizeof(unsigned long long) * v.size(), hipMemcpyDeviceToHost));
    if (FILE* f = fopen(path, "a")) {
      fprintf(f, "# wg start end hot_tile_cycles xcc\n");
      for (int i = 0; i < 1024; ++i)
        if (v[4 * i + 1]) fprintf(f, "%d %llu %llu %llu %llu\n", i, v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
      fclose(f);
    }
  }
  if (reset) HIP_TRY(hipMemset(c->d_tprof, 0, sizeof(unsigned long long) * 16));
  return DSGD_OK;
}

int dsgd_tuning_info(dsgd_ctx* c, int32_t* vals, int32_t n) {
  DSGD_TRY(check_ctx(c));
  if (!vals || n < 0 || n > 7) return fail(DSGD_EINVAL, "bad tuning_info arguments");
  std::lock_guard<std::mutex> lk(c->mu);
  const int32_t all[7] = {4 /* split layout: the only one */, std::min(c->hsplit, c->dp), c->last_shift,
                          c->cold_col16 ? 1 : 0, c->plan_kernel ? 1 : 0, c->fix_bound ? 1 : 0, c->last_fstep_rebalances};
  for (int i = 0; i < n; ++i) vals[i] = all[i];
  return DSGD_OK;
}

int dsgd_column_ranks(dsgd_ctx* c, int32_t* rank_of_key) {
  DSGD_TRY(check_ctx(c));
  if (!rank_of_key) return fail(DSGD_EINVAL, "null rank_of_key");
  std::lock_guard<std::mutex> lk(c->mu);
  DSGD_TRY(bind(c));
  DSGD_TRY(require_data(c));
  DSGD_TRY(require_sync_mode(c));
  DSGD_TRY(prepare_layout(c));
  HIP_TRY(hipMemcpy(rank_of_key, c->d_perm, sizeof(int) * (size_t)c->dp, hipMemcpyDeviceToHost));
  return DSGD_OK;
}

const char* dsgd_grad_kernel_name(dsgd_ctx* c) {
  if (!c) return "";
  return c->last_grad_kernel;
}

int dsgd_device_ptrs(dsgd_ctx* c, void** w_dev, void** g_dev, void** stream) {
  DSGD_TRY(check_ctx(c));
  std::lock_guard<std::mutex> lk(c->mu);
  // (binding converts slice-major weights back: after a column-slice dsgd_plan_run the live weights are NOT in d_w until
  //  some entry point binds -- this one does, on the stream it hands out)
  DSGD_TRY(bind(c));
  if (w_dev) *w_dev = c->d_w;
  if (g_dev) *g_dev = c->d_gsum;
  if (stream) *stream = c->stream;
  return DSGD_OK;
}

// ---- K8: dense logistic mini-batch step (no reference counterpart; include/dsgd.h) ------------------------------
struct dsgd_dense {
  int D = 0, device = 0, n_cu = 256;
  std::mutex mu;
  hipStream_t stream = nullptr;
  long long n_rows = 0;
  float* d_X = nullptr;
  float* d_y = nullptr;
  float* d_w = nullptr;
  float* d_g = nullptr;
  float* d_gpart = nullptr;   // n_wg x D
  double* d_lpart = nullptr;  // n_wg x 2
  double* h_lpart = nullptr;  // pinned
  int n_wg = 0;
  bool mfma = false;          // DSGD_DENSE_MFMA=1: forward product with v_mfma_f32_16x16x4_f32 (D a multiple of 256, <= 4096)
  rccl::comm_t comm = nullptr;
  int world = 1;
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
  double ms_sum = 0.0;
  long long ms_n = 0;
};
static int dn_bind(dsgd_dense* d) {
  if (!d) return fail(DSGD_EINVAL, "null dense object");
  HIP_TRY(hipSetDevice(d->device));
  return DSGD_OK;
}
static int dn_collect(dsgd_dense* d) {
  for (size_t i = 0; i < d->ev_used; ++i) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev[i].first, d->ev[i].second));
    d->ms_sum += ms;
    d->ms_n++;
  }
  d->ev_used = 0;
  return DSGD_OK;
}
static int dn_launch(dsgd_dense* d, long long rb, long long re, bool grad) {
  DenseArgs a;
  a.X = d->d_X;
  a.y = d->d_y;
  a.w = d->d_w;
  a.gpart = grad ? d->d_gpart : nullptr;
  a.lpart = d->d_lpart;
  a.row_begin = rb;
  a.row_end = re;
  a.D = d->D;
  const int rows_per_block = d->mfma ? 16 : DN_ROWS;
  const long long n_blocks = (re - rb + rows_per_block - 1) / rows_per_block;
  const int grid = (int)std::max<long long>(1, std::min<long long>(d->mfma ? d->n_cu : d->n_wg, n_blocks));
  size_t slot = (size_t)-1;
  if (d->prof && grad) {
    if (d->ev_used == d->ev.size() && d->ev.size() >= 4096) {   // bounded pool: collect what has completed so far
      HIP_TRY(hipStreamSynchronize(d->stream));
      DSGD_TRY(dn_collect(d));
    }
    if (d->ev_used == d->ev.size()) {
      hipEvent_t x, y;
      HIP_TRY(hipEventCreate(&x));
      HIP_TRY(hipEventCreate(&y));
      d->ev.emplace_back(x, y);
    }
    slot = d->ev_used++;
    HIP_TRY(hipEventRecord(d->ev[slot].first, d->stream));
  }
  if (d->mfma)   // the variant on the matrix cores (forward product only; csrc/dsgd_dense.hpp)
    hipLaunchKernelGGL(dsgd_dense_step_mfma_kernel, dim3(grid), dim3(d->D / 4), sizeof(float) * (size_t)d->D, d->stream, a);
  else
    hipLaunchKernelGGL(dsgd_dense_step_kernel, dim3(grid), dim3(d->D / DN_COLS), 0, d->stream, a);
  HIP_TRY(hipGetLastError());
  if (slot != (size_t)-1) HIP_TRY(hipEventRecord(d->ev[slot].second, d->stream));
  return grid;
}

int dsgd_dense_create(int32_t n_features, int32_t device, dsgd_dense** out) {
  if (!out) return fail(DSGD_EINVAL, "null argument");
  if (n_features < 512 || n_features > 8192 || n_features % 512)
    return fail(DSGD_EINVAL, "n_features must be a multiple of 512 in [512, 8192] (one lane owns 8 columns)");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(DSGD_EUNSUPPORTED, "no HIP device visible: libdsgd_hip has no CPU fallback");
  if (device < 0 || device >= n) return fail(DSGD_EINVAL, "device %d out of range (%d devices)", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(DSGD_EUNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  dsgd_dense* d = new (std::nothrow) dsgd_dense();
  if (!d) return fail(DSGD_ENOMEM, "out of host memory");
  d->D = n_features;
  d->device = device;
  d->n_cu = prop.multiProcessorCount;
  d->n_wg = 2 * d->n_cu;   // two workgroups per CU: one computes while the other's loads are in flight
  if (const char* e = getenv("DSGD_DENSE_MFMA")) d->mfma = atoi(e) != 0;
  if (d->mfma && n_features > 4096) {
    delete d;
    return fail(DSGD_EUNSUPPORTED, "the MFMA variant handles at most 4096 features (one wave per 256 columns)");
  }
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&d->d_w, sizeof(float) * d->D);
  if (e == hipSuccess) e = hipMalloc(&d->d_g, sizeof(float) * d->D);
  if (e == hipSuccess) e = hipMalloc(&d->d_gpart, sizeof(float) * (size_t)d->n_wg * d->D);
  if (e == hipSuccess) e = hipMalloc(&d->d_lpart, sizeof(double) * 2 * d->n_wg);
  if (e == hipSuccess) e = hipHostMalloc(&d->h_lpart, sizeof(double) * 2 * d->n_wg, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMemset(d->d_w, 0, sizeof(float) * d->D);
  if (e != hipSuccess) {
    dsgd_dense_destroy(d);
    return fail(DSGD_EHIP, "dense create: %s", hipGetErrorString(e));
  }
  *out = d;
  return DSGD_OK;
}

int dsgd_dense_destroy(dsgd_dense* d) {
  if (!d) return DSGD_OK;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->comm && rccl::available()) rccl::CommDestroy(d->comm);
  for (auto& e : d->ev) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  (void)hipFree(d->d_X);
  (void)hipFree(d->d_y);
  (void)hipFree(d->d_w);
  (void)hipFree(d->d_g);
  (void)hipFree(d->d_gpart);
  (void)hipFree(d->d_lpart);
  if (d->h_lpart) (void)hipHostFree(d->h_lpart);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  return DSGD_OK;
}

static int dn_alloc_rows(dsgd_dense* d, long long n_rows) {
  HIP_TRY(hipStreamSynchronize(d->stream));
  (void)hipFree(d->d_X);
  (void)hipFree(d->d_y);
  d->d_X = nullptr;
  d->d_y = nullptr;
  d->n_rows = 0;
  HIP_TRY(hipMalloc(&d->d_X, sizeof(float) * (size_t)n_rows * (size_t)d->D));
  HIP_TRY(hipMalloc(&d->d_y, sizeof(float) * (size_t)n_rows));
  d->n_rows = n_rows;
  return DSGD_OK;
}

int dsgd_dense_generate(dsgd_dense* d, int64_t n_rows, uint64_t seed) {
  DSGD_TRY(dn_bind(d));
  if (n_rows < 1) return fail(DSGD_EINVAL, "n_rows must be >= 1");
  std::lock_guard<std::mutex> lk(d->mu);
  DSGD_TRY(dn_alloc_rows(d, n_rows));
  hipLaunchKernelGGL(dsgd_dense_generate_kernel, dim3((unsigned)std::min<long long>(n_rows, (long long)d->n_cu * 8)), dim3(256), 0,
                     d->stream, d->d_X, d->d_y, (long long)n_rows, d->D, (unsigned long long)seed);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(d->stream));
  return DSGD_OK;
}

int dsgd_dense_load(dsgd_dense* d, int64_t n_rows, const float* X, const float* y) {
  DSGD_TRY(dn_bind(d));
  if (n_rows < 1 || !X || !y) return fail(DSGD_EINVAL, "n_rows must be >= 1 and arrays non-null");
  for (int64_t i = 0; i < n_rows; ++i)
    if (y[i] != 0.0f && y[i] != 1.0f) return fail(DSGD_EINVAL, "label[%lld] = %g, expected 0 or 1", (long long)i, (double)y[i]);
  std::lock_guard<std::mutex> lk(d->mu);
  DSGD_TRY(dn_alloc_rows(d, n_rows));
  HIP_TRY(hipMemcpy(d->d_X, X, sizeof(float) * (size_t)n_rows * (size_t)d->D, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d->d_y, y, sizeof(float) * (size_t)n_rows, hipMemcpyHostToDevice));
  return DSGD_OK;
}

int dsgd_dense_set_weights(dsgd_dense* d, const float* w) {
  DSGD_TRY(dn_bind(d));
  if (!w) return fail(DSGD_EINVAL, "null w");
  std::lock_guard<std::mutex> lk(d->mu);
  HIP_TRY(hipStreamSynchronize(d->stream));
  HIP_TRY(hipMemcpy(d->d_w, w, sizeof(float) * d->D, hipMemcpyHostToDevice));
  return DSGD_OK;
}
int dsgd_dense_get_weights(dsgd_dense* d, float* w_out) {
  DSGD_TRY(dn_bind(d));
  if (!w_out) return fail(DSGD_EINVAL, "null w_out");
  std::lock_guard<std::mutex> lk(d->mu);
  HIP_TRY(hipStreamSynchronize(d->stream));
  HIP_TRY(hipMemcpy(w_out, d->d_w, sizeof(float) * d->D, hipMemcpyDeviceToHost));
  return DSGD_OK;
}

static int dn_check_range(dsgd_dense* d, long long rb, long long re) {
  if (!d->d_X) return fail(DSGD_ESTATE, "no data (dsgd_dense_generate / dsgd_dense_load)");
  if (re <= rb) return fail(DSGD_EINVAL, "empty row range");
  if (rb < 0 || re > d->n_rows) return fail(DSGD_ERANGE, "rows [%lld, %lld) outside the %lld resident rows", rb, re, d->n_rows);
  return DSGD_OK;
}

int dsgd_dense_step(dsgd_dense* d, int64_t row_begin, int64_t row_end, float lr) {
  DSGD_TRY(dn_bind(d));
  std::lock_guard<std::mutex> lk(d->mu);
  DSGD_TRY(dn_check_range(d, row_begin, row_end));
  const int grid = dn_launch(d, row_begin, row_end, true);
  if (grid < 0) return grid;
  const float scale = lr / ((float)(row_end - row_begin) * (float)d->world);   // every rank contributes an equal batch
  hipLaunchKernelGGL(dsgd_dense_reduce_kernel, dim3((d->D + 63) / 64), dim3(1024), 0, d->stream, d->d_gpart, grid, d->D, d->d_g,
                     d->comm ? (float*)nullptr : d->d_w, scale);
  HIP_TRY(hipGetLastError());
  if (d->comm) {
    RCCL_TRY(rccl::AllReduce(d->d_g, d->d_g, (size_t)d->D, rccl::kFloat32, rccl::kSum, d->comm, d->stream));
    hipLaunchKernelGGL(dsgd_dense_apply_kernel, dim3((d->D + 255) / 256), dim3(256), 0, d->stream, d->d_w, d->d_g, d->D, scale);
    HIP_TRY(hipGetLastError());
  }
  return DSGD_OK;
}

int dsgd_dense_synchronize(dsgd_dense* d) {
  DSGD_TRY(dn_bind(d));
  std::lock_guard<std::mutex> lk(d->mu);
  HIP_TRY(hipStreamSynchronize(d->stream));
  return dn_collect(d);
}

int dsgd_dense_loss(dsgd_dense* d, int64_t row_begin, int64_t row_end, double* loss, double* acc) {
  DSGD_TRY(dn_bind(d));
  std::lock_guard<std::mutex> lk(d->mu);
  DSGD_TRY(dn_check_range(d, row_begin, row_end));
  const int grid = dn_launch(d, row_begin, row_end, false);
  if (grid < 0) return grid;
  HIP_TRY(hipMemcpyAsync(d->h_lpart, d->d_lpart, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  double l = 0.0, c = 0.0;
  for (int b = 0; b < grid; ++b) {
    l += d->h_lpart[2 * b];
    c += d->h_lpart[2 * b + 1];
  }
  const double n = (double)(row_end - row_begin);
  if (loss) *loss = l / n;
  if (acc) *acc = c / n;
  return DSGD_OK;
}

int dsgd_dense_comm_init(dsgd_dense* d, const char* unique_id, int32_t world_size, int32_t rank) {
  DSGD_TRY(dn_bind(d));
  if (!unique_id || world_size < 1 || rank < 0 || rank >= world_size) return fail(DSGD_EINVAL, "bad communicator arguments");
  if (!rccl::available()) return fail(DSGD_ERCCL, "librccl could not be loaded");
  std::lock_guard<std::mutex> lk(d->mu);
  if (d->comm) return fail(DSGD_ESTATE, "communicator already attached");
  rccl::unique_id_t id;
  memcpy(id.internal, unique_id, DSGD_UNIQUE_ID_BYTES);
  RCCL_TRY(rccl::CommInitRank(&d->comm, world_size, id, rank));
  d->world = world_size;
  return DSGD_OK;
}

int dsgd_dense_prof(dsgd_dense* d, int32_t enable, double* kernel_ms_avg, int64_t* n_launches) {
  DSGD_TRY(dn_bind(d));
  std::lock_guard<std::mutex> lk(d->mu);
  HIP_TRY(hipStreamSynchronize(d->stream));
  DSGD_TRY(dn_collect(d));
  if (kernel_ms_avg) *kernel_ms_avg = d->ms_n ? d->ms_sum / (double)d->ms_n : 0.0;
  if (n_launches) *n_launches = d->ms_n;
  d->ms_sum = 0.0;
  d->ms_n = 0;
  d->prof = enable != 0;
  return DSGD_OK;
}

}  // extern "C"
