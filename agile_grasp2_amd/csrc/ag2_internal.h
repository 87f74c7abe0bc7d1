// ag2_internal.h -- host-side context of libag2hip.so and the launchers of its kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "ag2_c.h"
#include "ag2_device.h"

namespace ag2 {

// Grow-only device buffer.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t reserve(size_t need) {
    if (need <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    size_t want = need + need / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) bytes = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return (T*)p; }
};

// Device-side counters / flags block (one per context, zeroed per call).
struct DevStats {
  // --- zeroed before every hypothesis run (everything before `bounds`) ---
  // The counters that thousands of workgroups hit with atomics during one kernel -- and whose return
  // values some of them wait for -- sit on cache lines of their own: atomics on ONE line are served
  // one after the other by the L2 channel that owns it, whatever their addresses within it.
  unsigned long long sum_k2, sum_kcrop, sum_p;
  unsigned int n_frames, n_hyp, n_pruned_keep;
  unsigned int err_flags;            // bit0 arena overflow, bit1 list arena overflow (split sweep), bit3 global sweep scratch overflow
  unsigned int n_list;               // hypotheses that go on to be scored (after the prune)
  unsigned int n_sel;                // scored hypotheses with score >= min_score_diff
  unsigned int n_clu;                // selected hypotheses that survive the clustering
  unsigned int max_p;                // largest in-box list of the run (picks the image renderers)
  unsigned int max_k_over;           // longest cropped list that did not fit the sweep's global scratch
  alignas(128) unsigned long long arena_top;  // in-box points reserved in the arena (returning atomic per hypothesis)
  alignas(128) unsigned long long list_top;   // split sweep: points reserved in the list arena (returning atomic per sample)
  alignas(128) unsigned int n_pairs;          // split sweep: (sample, orientation) pairs queued for k_sweep_orient
  alignas(128) unsigned int n_overflow;       // samples handed on to the long-list stage
  alignas(128) unsigned int work_next[3];     // k_sweep work queues, one per stage (items beyond the first grid)
  unsigned int topk_blocks;                   // k_topk: workgroups that have written their part (the last one writes the done flag)
  // --- per cloud (zeroed by k_init_stats when the grid is rebuilt) ---
  alignas(128) unsigned int bounds[7];        // ordered-int min xyz, max xyz, n_valid
  unsigned int pad1;
  unsigned long long sum_k1;         // neighbours visited by k_normals
};

// float <-> int whose signed order matches the float order (atomicMin/atomicMax on bounds)
__host__ __device__ __forceinline__ int f2ord(float f) {
  int i;
  __builtin_memcpy(&i, &f, 4);
  return i ^ ((i >> 31) & 0x7fffffff);
}
__host__ __device__ __forceinline__ float ord2f(int i) {
  const int b = i ^ ((i >> 31) & 0x7fffffff);
  float f;
  __builtin_memcpy(&f, &b, 4);
  return f;
}

// Device-side state of the preprocessing kernels (k_preprocess.hip).
struct PreStats {
  int mn[3], mx[3];            // ordered-int bounds of the points that pass the workspace filter
  unsigned int n_keep;         // their count
  unsigned int sel_prefix[3];  // radix select of the sub-sampling threshold: (hash hi, hash lo, index)
  unsigned int sel_remaining;
  unsigned int pad;
};

// Device-side record of one pass of the fixed-shape GPU front end (k_preprocess.hip: workspace filter,
// voxel grid and uniform sub-sampling without a host round trip).  Written by its kernels, read by the
// host together with the results.
struct PreFrame {
  float mn[3];                 // minimum of the points that pass the filter = origin of the voxel lattice
  float cell;
  int dims[3];
  int words;                   // bitmap words the lattice needs (0: nothing to mark)
  unsigned n_keep, n_vox, n_cand;
  unsigned flags;              // kPre* below; non-zero: the caller repeats the step on the general path
  unsigned long long thr;      // sub-sampling: hashes up to thr are candidates
  unsigned long long kth_h;    // the num_samples-th smallest key (hash, index)
  unsigned kth_i, pad;
};
constexpr unsigned kPreGridTooLarge = 1u;   // the voxel lattice needs more bitmap words than reserved (or is too fine)
constexpr unsigned kPreTooManyVoxels = 2u;  // more voxels than the cloud buffer holds
constexpr unsigned kPreCandOverflow = 4u;   // more sub-sampling candidates than the candidate list holds
constexpr unsigned kPreCandShort = 8u;      // fewer candidates than samples (the hash filter missed: p < 1e-12)
constexpr unsigned kPreAllPoints = 16u;     // no more voxels than num_samples: every point is a sample (general path)

// Uniform sub-sampling = the num_samples smallest of the keys (draw_u64(seed, stream, i), i).  The keys are
// uniform 64-bit hashes, so the num_samples-th smallest is close to num_samples / n x 2^64: hashes up to a
// threshold eight standard deviations above that are collected as candidates (a few per cent more than
// num_samples) and the exact selection runs over the candidates only.
__host__ __device__ inline double cand_mean(unsigned k) { return (double)k + 8.0 * __builtin_sqrt((double)k) + 16.0; }
__host__ __device__ inline unsigned long long cand_threshold(unsigned k, unsigned n) {
  if (n == 0u || k == 0u) return 0ull;
  const double p = cand_mean(k) / (double)n;
  if (p >= 1.0) return ~0ull;
  return (unsigned long long)(p * 18446744073709551616.0);
}
inline size_t cand_capacity(size_t k) {
  const double mu = cand_mean((unsigned)k);
  return (size_t)(mu + 8.0 * __builtin_sqrt(mu) + 64.0);
}

struct LeNetDev {
  bool loaded = false;
  DevBuf w1p, b1, w2p, b2, w3p, b3, w4, b4;  // packed for the MFMA lane layout (k_lenet.hip)
  DevBuf w1x, w2x, w3x;                      // conv / ip1 weights split into 3 bf16 terms (k_lenet_x3.hip)
  bool use_x3 = true;                        // false: the f32-input MFMA convolutions (AG2_LENET_F32=1)
  bool use_bands = true;                     // false: k_lenet_conv_x3, one workgroup per image (AG2_LENET_WHOLE=1)
};

}  // namespace ag2

struct ag2_ctx {
  ag2_params p;
  int device = 0;
  std::string err;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev[16] = {};

  // cloud
  size_t n = 0;           // points given
  size_t n_valid = 0;     // finite points (grid-resident)
  ag2::GridDesc grid{};
  bool has_cloud = false, has_normals = false;
  bool normals_pending = false;  // k_normals launched, stats not collected yet
  bool grid_pending = false;     // grid kernels launched, duration not collected yet
  float min_z = 0.f;
  ag2::DevBuf d_xyz_in;    // packed float4 (x,y,z, cam mask bits) in ORIGINAL order
  ag2::DevBuf d_key;       // int2 per original point: cell key (-1 invalid), arrival rank in the cell
  ag2::DevBuf d_bounds;    // per-workgroup extent partials of k_bounds (8 ints each)
  ag2::DevBuf d_gpos;      // sweep stage 0: per-workgroup position lists that exceed its LDS
  ag2::DevBuf d_export_list;  // occupied slots, for ag2_export_candidates_compact_device
  int bounds_blocks = 0;   // > 0: the fused pack left that many partials for build_grid
  ag2::DevBuf d_cell;      // uint32 cell_start[ncells+1]
  ag2::DevBuf d_perm;      // int32 sorted position -> original index
  ag2::DevBuf d_sorted;    // float4 (x,y,z, cam mask bits) in sorted order
  ag2::DevBuf d_nrm;       // float4 (nx,ny,nz,0) in sorted order
  ag2::DevBuf d_scan;      // control words of the chained scan
  ag2::DevBuf d_stats;     // DevStats
  ag2::DevBuf d_hc;        // HandConst
  ag2::HandConst hc{};

  // preprocessing (k_preprocess.hip)
  ag2::DevBuf d_raw;       // float4 (x,y,z, cam mask bits) as handed over, before filter / voxel grid
  ag2::DevBuf d_raw_nrm;   // float4 normals that came with the raw cloud
  ag2::DevBuf d_pre;       // float4 points that passed the workspace filter
  ag2::DevBuf d_pflags;    // uint32 keep flags -> exclusive prefix
  ag2::DevBuf d_bitmap;    // occupancy bit per voxel, ascending (ix, iy, iz) key
  ag2::DevBuf d_wrank;     // uint32 per bitmap word: rank of its first voxel
  ag2::DevBuf d_first;     // int32 per voxel: smallest point index inside
  ag2::DevBuf d_prestats;  // PreStats
  ag2::DevBuf d_hist;      // 65536-bin digit histogram of the sub-sampling radix select
  ag2::DevBuf d_samples;   // int32 sample indices left resident by ag2_subsample_uniformly
  ag2::DevBuf d_preframe;  // PreFrame
  ag2::DevBuf d_cand;      // sub-sampling candidates: uint64 hash[cap] then uint32 index[cap]
  size_t n_resident_samples = 0;
  int last_vox_dims[3] = {0, 0, 0};  // voxel lattice of the last ag2_preprocess_cloud* call (shapes of a raw stream)
  bool origin_set = false;    // ag2_set_grid_origin: grid origin (and cloud minimum) of the whole cloud
  float origin[3] = {0, 0, 0};
  bool bounds_known = false;  // set by the front end for the next grid build: extent of d_xyz_in
  float known_min[3] = {0, 0, 0}, known_max[3] = {0, 0, 0};

  // samples / hypotheses of the last generate call
  size_t s = 0;
  uint64_t slot_base = 0;
  ag2::DevBuf d_sample_q;  // float4 per sample (query xyz, valid flag)
  ag2::DevBuf d_frames;    // 12 doubles per sample + valid
  ag2::DevBuf d_frame_ok;  // int per sample
  ag2::DevBuf d_table;     // ag2_hypothesis[s*R] fixed-slot table
  ag2::DevBuf d_tab_off;   // int64 arena offset per slot
  ag2::DevBuf d_tab_keep;  // uint8 prune flag per slot
  ag2::DevBuf d_arena;     // 6 doubles per in-box point (U.xyz, Y.xyz)
  size_t arena_points = 0;
  ag2::DevBuf d_overflow;  // int sample ids that need the global-memory sweep
  ag2::DevBuf d_gscratch;  // global cropped-list scratch for the overflow path
  ag2::DevBuf d_lists;     // split sweep: cropped lists (float4: centred xyz, sorted position) of the samples with a passing orientation
  ag2::DevBuf d_pairs;     // split sweep: SweepPair queue
  ag2::DevBuf d_obox;      // split sweep: per-workgroup closing-region index lists beyond the LDS part
  // ag2_detect without the host round trip in its middle: shapes learned from the previous call
  size_t spec_cap_img = 0, spec_s = 0;
  int spec_max_p = 0, spec_prune = -1;
  long long spec_runs = 0, spec_fallbacks = 0;
  bool defer_hyp_stats = false;    // set around launch_sweep by a caller that compacts the slot table next
  bool hyp_stats_pending = false;  // the split sweep has left its statistics to compact_slots_async
  size_t list_ints = 0;    // capacity of d_lists in points; grown on demand like the arena
  int sweep_gcap = 1 << 16;  // points per workgroup of that scratch; grows to the longest list met
  int sweep_g2 = 1024;       // workgroups of the stage that uses it
  size_t cell_bytes_last = 0;     // bytes of d_cell the last grid build cleared (counters + scan control words)
  size_t cell_bytes_prezeroed = 0;  // ... and how many the pack of the CURRENT cloud has cleared ahead of its grid build
  const void* cell_prezeroed_at = nullptr;
  int cell_count_ahead_cap = 0;   // > 0: pack_device_xyz also queued k_cell_count (grid derived on the device from the
                                  // extent partials) for grids of at most this many cells; build_grid skips its own
  unsigned bounds_seq = 0;   // ... of the last extent pass whose follower raises the flag in the small area
  bool bounds_flag_armed = false;
  unsigned topk_seq = 0;     // sequence number of the last k_topk launch (the done flag in the page-locked small area)
  // A rank of a multi-GPU job (ag2_detect without a result buffer: the merge selects) in ONE host round trip --
  // none at all inside the detect: the tail is launched at the shapes the previous call left, and whether they
  // held travels in the header of the exported list (k_export_selected), where EVERY rank sees it after the
  // exchange and takes the same decision (ag2_merge_*: AG2_ERR_RETRY).  The statistics come back in page-locked
  // memory with the export and are taken up at the next point that synchronises anyway (rank_spec_collect).
  struct RankSpec {
    bool pending = false;      // a one-trip rank detect whose statistics have not been taken up yet
    unsigned cap_img = 0;      // images its tail was launched for
    int render_cap = 0;        // in-box points its renderers take
    int stage1_skipped = 0;    // its sweep left the long-list stage out
    size_t s = 0;
  } rank_spec;
  int sweep_no_overflow_runs = 0;   // consecutive runs that handed no sample to the long-list stage
  bool sweep_may_skip_stage1 = false;  // set by the one-round-trip detect around its sweep: it checks and repeats
  bool sweep_stage1_skipped = false;   // the last sweep did not launch the long-list stage
  int sweep_gpos_cap = 0;    // longest list the sweep's first stage keeps (k_sweep.hip: kGposCap / kGposCapBig, adaptive)
  ag2::DevBuf d_list;      // int compacted slot ids (hypotheses in order)
  ag2::DevBuf d_list2;     // int compacted slot ids after prune / for scoring
  ag2::DevBuf d_images;    // uint8 n_img x 10800 (HWC)
  ag2::DevBuf d_logits;    // float n_img x 2
  ag2::DevBuf d_act1;      // LeNet intermediates (pooled2: n x 7200 float)
  ag2::DevBuf d_fcpart;    // ip1 split-K partial sums [ksplit][n_pad][512]
  ag2::DevBuf d_tmp;       // misc staging
  ag2::DevBuf d_sel;       // ag2_hypothesis: scored records with score >= min_score_diff, list order (+ count trailer)
  ag2::DevBuf d_merge;     // ag2_hypothesis: the gathered selected lists of all ranks, flattened (ag2_merge_selected_device)
  ag2::DevBuf d_gather;    // one process, several GPUs: the ranks' compact lists on the root's GPU (ag2_gather_begin)
  ag2::DevBuf d_xchg;      // ... and a rank's own list before the peer copy
  size_t gather_world = 0, gather_cap = 0;
  std::vector<uint8_t> gather_delivered;  // per rank: ag2_gather_selected has completed since ag2_gather_begin
  const void* d_last_sel = nullptr;    // what the last ag2_detect selected from (d_sel or d_cluster) ...
  const unsigned* d_last_nsel = nullptr;  // ... and its count on the device
  ag2::DevBuf d_flags;     // uint32 flags / prefix for slot compaction
  ag2::DevBuf d_desc;      // image descriptors: int64 arena offset[n] then int32 count[n]
  ag2::DevBuf d_rlist;     // renderers of the larger images: 8 counters, then one image list per renderer (k_image.hip)
  ag2::DevBuf d_cluster;   // ag2_hypothesis: clustered hands, compacted (k_cluster.hip)
  ag2::DevBuf d_cluster_tmp;  // ag2_hypothesis: moved hands before the compaction
  int min_inliers = 0;     // HandleSearch::setMinInliers; 0 = no clustering inside ag2_detect
  // page-locked host staging (small read-backs, sample indices, result records): copies from / to it
  // are true asynchronous DMA, so the host keeps enqueueing while the GPU works
  void* h_pin = nullptr;
  void* h_pin_dev = nullptr;      // the same block as the device sees it (kernels that write results where the host reads them)
  bool bounds_in_pin = false;     // the extent partials of the current cloud were written straight into pin_small
  size_t h_pin_bytes = 0;
  std::vector<ag2_hypothesis> h_hyps;   // compacted hypotheses of the last generate call
  std::vector<int32_t> h_slots;         // their slot ids
  std::vector<int64_t> h_offsets;       // their arena offsets
  std::vector<uint8_t> h_keep;          // their prune flags
  int stage_timing = 2;    // which per-stage HIP events are recorded (ag2_set_stage_timing)
  size_t n_img = 0;
  int max_p = 0;           // largest in-box point list of the last hypothesis run
  size_t desc_stride = 0;  // > 0: the compaction also left the image descriptors in d_desc (offsets,
                           // then counts at + desc_stride)

  // frame mode (ag2_frame.hip): the launchers use device-resident shapes (grid description, list
  // lengths) and launch sizes fixed at the maxima below, so that a frame can be captured in a hipGraph
  bool fm_on = false;
  size_t fm_n_max = 0;       // points per frame (the cloud is padded with non-finite points up to it)
  size_t fm_s_max = 0;       // samples per frame (the index list is padded with -1 up to it)
  size_t fm_cap_cells = 0;   // grid cells
  ag2::DevBuf d_griddesc;    // GridDesc written by k_cell_count (from the extent partials) or by the front end
  bool fm_skip_stage1 = false; // frame mode: the sequence leaves the sweep's long-list stage out (learned: no frame so far needed it)
  bool fm_grid_ready = false;  // frame mode behind the GPU front end: d_griddesc is written by k_vox_emit_frame
  const ag2::FrameArgs* fm_args_dev = nullptr;  // device view of the page-locked per-frame scalars
  struct ag2_frame_state* fm = nullptr;         // owned by ag2_frame.hip

  // hipFuncSetAttribute is per device: which of the large-dynamic-LDS kernels this context has prepared on ITS
  // device (bits kAttr*); a process-wide flag left devices >= 1 without the attribute
  unsigned func_attr_done = 0;
  ag2::DevBuf d_sweep_prof;  // diagnostic only (AG2_SWEEP_PROF): per-phase cycle sums of the sweep kernels
  // how the host waits for a kernel's flag in page-locked memory (ag2_set_wait_mode): 0 the stream
  // (hipStreamSynchronize), 1 poll; spin_us: busy polling before the poller starts yielding its core
  int wait_poll = 1;
  int wait_spin_us = 50;
  int64_t poll_fallbacks = 0;   // waits that polled 5 ms without the flag and went on to wait for the stream
  int64_t poll_yields = 0;      // sched_yield calls of those waits
  int64_t last_wait_us = 0, last_submit_us = 0;  // host time inside the last frame_wait / frame_submit

  ag2::LeNetDev net;
  ag2_counters cnt{};
  ag2_times times{};
};

namespace ag2 {

int set_err(ag2_ctx* c, int code, const std::string& msg);
enum { kAttrRender = 1u, kAttrLenetConv = 2u, kAttrLenetX3 = 4u, kAttrLenetX3b = 8u };
#define AG2_HIP(c, expr)                                                                  \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return ag2::set_err(c, AG2_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// ag2_context.hip
int collect_normals_stats(ag2_ctx* c);
int after_cloud(ag2_ctx* c);
// page-locked staging of at least `bytes` (+ kPinSmall bytes in front for small read-backs); growing
// synchronises the stream first.  Layout: [0, kPinSmall) small area, [kPinSmall, ...) bulk area.
constexpr size_t kPinSmall = 20480;
// offsets in the small area of the sequence numbers a kernel's last workgroup writes behind its results for the
// host to poll (wait_flag): k_topk's, k_bounds'
constexpr size_t kPinDoneFlag = 16384, kPinBoundsFlag = 16384 + 64;  // (the first 16 KB: the extent partials)
// ... and where k_export_selected leaves the statistics of a rank's one-trip detect (RankSpec below)
constexpr size_t kPinRankStats = 16384 + 128;
static_assert(kPinRankStats + sizeof(ag2::DevStats) <= kPinSmall, "pin_small: room for the rank statistics");
constexpr unsigned kPinFlags = hipHostMallocCoherent | hipHostMallocMapped;  // every block a kernel writes for the host to poll
int pin_reserve(ag2_ctx* c, size_t bulk_bytes);
inline char* pin_small(ag2_ctx* c) { return (char*)c->h_pin; }
inline char* pin_bulk(ag2_ctx* c) { return (char*)c->h_pin + kPinSmall; }
inline char* pin_small_dev(ag2_ctx* c) { return (char*)c->h_pin_dev; }
inline char* pin_bulk_dev(ag2_ctx* c) { return (char*)c->h_pin_dev + kPinSmall; }
// k_grid.hip
int build_grid(ag2_ctx* c);
// Per-stage timing events cost a few microseconds of stream serialisation each (about 3 % of a cfg2
// step for all of them), so how many are recorded is a setting (ag2_set_stage_timing): 0 none,
// 1 only the two around the whole sweep (the dominant kernel; sweep_ms is then its total and
// sweep_overflow_ms 0), 2 all.  Stages whose events
// are not recorded report 0 ms.
inline bool stage_event_on(const ag2_ctx* c, int i) {
  return c->stage_timing >= 2 || (c->stage_timing == 1 && (i == 1 || i == 11));
}
inline hipError_t stage_event(ag2_ctx* c, int i) {
  return stage_event_on(c, i) ? hipEventRecord(c->ev[i], c->stream) : hipSuccess;
}
inline void stage_elapsed(ag2_ctx* c, float* ms, int a, int b) {
  *ms = 0.f;
  if (stage_event_on(c, a) && stage_event_on(c, b) &&
      hipEventElapsedTime(ms, c->ev[a], c->ev[b]) != hipSuccess) {
    *ms = 0.f;                  // (an event of the pair was never recorded at this level)
    (void)hipGetLastError();    // not an error of the path: do not leave it for the next check
  }
}
inline hipError_t stage_sync(ag2_ctx* c, int i) {
  return stage_event_on(c, i) ? hipEventSynchronize(c->ev[i]) : hipSuccess;
}
int gather_normals(ag2_ctx* c);  // d_tmp (float4, original order) -> d_nrm (sorted order)
int pack_device_xyz(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes, float4* dst,
                    bool with_bounds = false, size_t n_pad = 0);
// frame mode: grid description derived on the device + cell count + scan + scatter + cell sort at the
// fixed maxima
int launch_grid_frame(ag2_ctx* c, unsigned* cell, unsigned* zeroed_ctl);
// ag2_frame.hip
void frame_release(ag2_ctx* c);
size_t scan_ctl_words(int n);
int scan_exclusive_u32(ag2_ctx* c, unsigned* d, int n, unsigned* zeroed_ctl = nullptr);
int scan_popc_u32(ag2_ctx* c, const unsigned* bitmap, int words, unsigned* rank, unsigned* zeroed_ctl = nullptr);
// k_preprocess.hip: the fixed-shape front end of a frame (ag2_frame.hip)
struct FrontShapes {
  size_t raw_max;     // raw points per frame (the raw cloud is padded with non-finite points up to it)
  size_t cap_words;   // bitmap words of the voxel lattice
  size_t n_max;       // voxels (= points of the processed cloud; padded with non-finite points up to it)
  size_t num_samples;
  size_t cand_cap;
  float cell;
  int filter_workspace;
};
int front_pack_raw(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes, const FrontShapes& fs);
int enqueue_front_frame(ag2_ctx* c, const FrontShapes& fs);
// k_normals.hip
int launch_normals(ag2_ctx* c);
// k_sweep.hip
int upload_samples(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                   bool clear_run = false);
int launch_sample_queries(ag2_ctx* c, const int* d_idx, size_t s, bool clear_run);
int launch_frames(ag2_ctx* c, size_t s, uint64_t slot_base, uint64_t seed);
// what the device top-k leaves for the host beside the records: statistics, grid, counts
struct FrameOut {
  DevStats st;
  GridDesc g;
  unsigned n_out;  // records that follow (top-k)
  unsigned topk_overflow;
  PreFrame pre;    // ag2_detect_frame_raw: what the front end of the frame found (voxels, flags)
  unsigned done_seq;  // frame mode: FrameArgs::seq, written last by the last workgroup of k_topk (frame_wait polls it)
};
// ag2_frame.hip: top num_selected of d_recs[0 .. min(*d_n, cap)) by (score desc, position asc) -> d_out, *d_fo
int launch_topk(ag2_ctx* c, const ag2_hypothesis* d_recs, const unsigned* d_n, size_t cap, size_t k_cap,
                ag2_hypothesis* d_out, FrameOut* d_fo, const GridDesc* gp, const PreFrame* pfp = nullptr);
// waits until the word at flag_off of the page-locked small area holds `want` (a kernel's last workgroup writes
// it behind the kernel's results), polling for up to ~2 ms, then for the stream the ordinary way
constexpr int kBoundsBlocks = 128;  // workgroups of the extent pass = 32-byte records of partials (4 KB)
int wait_flag(ag2_ctx* c, size_t flag_off, unsigned want);
int wait_flag_at(ag2_ctx* c, const volatile unsigned* flag, unsigned want);  // the same for a word anywhere in page-locked memory
inline int wait_topk(ag2_ctx* c) { return wait_flag(c, kPinDoneFlag, c->topk_seq); }
int launch_sweep(ag2_ctx* c, size_t s, uint64_t slot_base, bool emit_lists, bool run_cleared = false);
void sweep_adapt_gpos(ag2_ctx* c, size_t n_samples, size_t n_overflow);
int launch_hyp_stats(ag2_ctx* c, size_t n_slots);  // k_sweep_orient.hip: n_hyp, sum_p, max_p from the slot table
// k_select.hip
int compact_slots(ag2_ctx* c, size_t n_slots, int mode, DevBuf& out_list, size_t* n_out);
int compact_slots_async(ag2_ctx* c, size_t n_slots, int mode, DevBuf& out_list, unsigned* d_count,
                        bool with_descs = false);
int score_and_select_async(ag2_ctx* c, const int* d_list, size_t n_img, unsigned* d_count,
                           const unsigned* d_n = nullptr);
int gather_records(ag2_ctx* c, const int* d_list, size_t n, std::vector<ag2_hypothesis>& recs,
                   std::vector<int64_t>* offs, std::vector<uint8_t>* keep);
int export_candidates_compact(ag2_ctx* c, void* d_dst, size_t cap_records);
int export_selected_compact(ag2_ctx* c, void* d_dst, size_t cap_records);
int rank_spec_collect(ag2_ctx* c, bool stream_is_idle);  // ag2_pipeline.hip
int merge_selected(ag2_ctx* c, const void* d_gathered, size_t world, size_t cap_records, ag2_hypothesis* selected,
                   size_t cap, size_t* n_selected, size_t* n_total);
int make_image_descs(ag2_ctx* c, const int* d_list, size_t n);
// k_lenet_x3.hip
int lenet_pack_weights_x3(ag2_ctx* c, const float* conv1_w, const float* conv2_w);
int launch_lenet_conv_x3(ag2_ctx* c, const uint8_t* d_images, size_t n, float* d_pooled2,
                         const unsigned* d_n = nullptr);
int lenet_pack_fc_x3(ag2_ctx* c, const float* w3p_7200x512);
int launch_lenet_fc1_x3(ag2_ctx* c, size_t n, int* n_pad_out, int* ksplit_out,
                        const unsigned* d_n = nullptr);
// k_cluster.hip
int cluster_async(ag2_ctx* c, const ag2_hypothesis* d_in, size_t n_max, const unsigned* d_n,
                  int min_inliers, unsigned* d_count);
// k_image.hip
int launch_render(ag2_ctx* c, const double* d_arena, const long long* d_off, const int* d_cnt,
                  size_t n_img, uint8_t* d_out, int max_p, const unsigned* d_n = nullptr);
int render_capacity_for(int max_p);
// k_lenet.hip
int lenet_pack_weights(ag2_ctx* c, const float* c1w, const float* c1b, const float* c2w,
                       const float* c2b, const float* f1w, const float* f1b, const float* f2w,
                       const float* f2b);
int launch_lenet(ag2_ctx* c, const uint8_t* d_images, size_t n, float* d_logits, int ev_mid,
                 const unsigned* d_n = nullptr);

}  // namespace ag2
