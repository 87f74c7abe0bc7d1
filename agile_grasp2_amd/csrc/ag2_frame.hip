// ag2_frame.hip -- BASELINE.json configuration 5: a stream of clouds, one `ag2_detect_frame` per
// frame, the per-frame pipeline captured in a hipGraph.
//
// The caller this serves is the live-topic loop of the reference's node
// (src/nodes/grasp_detection_node.cpp:69-95 run(), :123-143 detectGraspPosesInTopic): one cloud in,
// one grasp list out, again and again with clouds of nearly the same size.  A frame is
//   search grid -> PCA normals -> local frames -> hand sweep -> prune -> grasp images -> LeNet ->
//   score threshold -> top-k
// i.e. ag2_set_cloud_device + ag2_compute_normals + ag2_detect, and returns the same bytes.
//
// What makes the sequence capturable: nothing in it waits for the host.  The three read-backs of the
// step-by-step path (grid extent, list length, results + host top-k) are gone --
//   * the extent partials are reduced on the device into a GridDesc that the kernels read from
//     memory (k_grid_desc),
//   * every list length (hypotheses to score, selected hypotheses, sweep overflow queue) is consumed
//     where it was produced; launches cover fixed maximum shapes and the surplus workgroups leave,
//   * the top-k by score runs on the device (k_topk), and ONE copy at the end brings statistics,
//     grid description and the selected records into page-locked memory.
// Fixed maximum shapes: the cloud is padded with non-finite points to fm_n_max (no stage sees a
// non-finite point), the sample list with -1 to fm_s_max, the cell table covers fm_cap_cells, the
// image list S*R.  What changes per frame and would be a by-value kernel argument (seed, slot base)
// sits in page-locked memory (FrameArgs).  The pack + extent kernel that reads the caller's buffer
// runs in front of the graph, outside it (its pointer and count change with every frame).
//
// A frame that does not fit the captured shapes (more points, samples or cells; arena or sweep
// scratch too small) is detected from the flags that come back with the results and is repeated on
// the step-by-step path, which sizes the buffers; the graph is captured again afterwards.
//
// ag2_detect_frame_raw starts one step earlier, at the RAW cloud of the sensor: the workspace filter,
// the 3 mm voxel grid and the uniform sub-sampling (GraspDetector::preprocessPointCloud,
// grasp_detector.cpp:285-335; CloudCamera::filterWorkspace / voxelizeCloud / subsampleUniformly,
// cloud_camera.cpp:89-178) run on the device inside the same captured sequence (k_preprocess.hip:
// enqueue_front_frame) -- the processed cloud, the search grid's description and the sample indices never
// leave HBM, and there is still exactly one host synchronisation per frame.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <sched.h>
#include <vector>

#include "ag2_internal.h"

using namespace ag2;

namespace ag2 {


}  // namespace ag2

struct ag2_frame_state {
  bool use_graph = true;
  bool shapes_known = false;
  size_t cap_img = 0;        // images rendered / scored per frame at most (learned: a multiple of what the
                             // frames so far needed, in whole batches of 256; at most fm_s_max * R)
  size_t k_cap = 0;          // records copied back per frame
  int do_prune = -1;         // of the captured sequence
  // page-locked block: FrameArgs | idx[s_max] | FrameOut | records[k_cap]
  char* h_pin = nullptr;
  size_t h_pin_bytes = 0;
  size_t off_idx = 0, off_out = 0, off_rec = 0;
  ag2::DevBuf d_raw;         // staging of a cloud handed over in host memory
  int cap_p = 0;             // in-box points per image the captured renderers can take
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool graph_valid = false;
  unsigned long long sig_at_capture = 0;  // addresses / sizes the captured graph holds
  ag2_frame_info info{};
  // ag2_detect_frame_raw: the shapes of the front end (raw == false: frames arrive preprocessed)
  bool raw = false;
  ag2::FrontShapes fs{};
  // a frame that has been submitted and not yet waited for (ag2_submit_frame* / ag2_wait_frame)
  struct Pending {
    bool active = false;
    bool finished = false;       // ran step by step inside the submit: the results below are final
    bool replay = false;         // the fixed-shape sequence went out as a graph launch
    bool unsupported = false;
    int rc = 0;
    const void* xyz = nullptr;   // the caller's buffer (device clouds) or the device staging copy
    int on_device = 0;
    size_t n = 0, stride = 12;
    bool raw = false;
    int filter_ws = 1;
    double voxel_size = 0.003;
    size_t num_samples = 0;
    uint64_t sample_seed = 0, seed = 0;
    int do_prune = 1;
    std::vector<int32_t> idx;    // copy of the caller's sample indices
    std::vector<ag2_hypothesis> recs;
    size_t n_selected = 0, n_scored = 0, n_voxels = 0;
  } pend;
  unsigned seq = 0;              // sequence number of the last frame enqueued at fixed shapes (FrameArgs::seq)
  char* h_stage = nullptr;       // page-locked staging of a cloud handed over in host memory (true async H2D)
  size_t h_stage_bytes = 0;
};

namespace ag2 {

// ---- top num_selected by score on the device ------------------------------------------------------
// grasp_detector.cpp:239-252: partial_sort by score, descending; ties by position in the list (the
// order the step-by-step path's host sort uses).  Every record finds its rank by counting the
// records that come before it in that order -- n^2 compares over a few hundred to a few thousand
// selected records, the scores staged through LDS -- and the first k are written at their rank.
constexpr int kTopkThreads = 256;
__global__ void __launch_bounds__(kTopkThreads) k_topk(const ag2_hypothesis* __restrict__ recs,
                                                       const unsigned* __restrict__ d_n, int cap,
                                                       int k_want, int k_cap,
                                                       ag2_hypothesis* __restrict__ out, FrameOut* fo,
                                                       DevStats* __restrict__ st,
                                                       const GridDesc* __restrict__ gp,
                                                       const PreFrame* __restrict__ pfp, unsigned seq,
                                                       unsigned* done_flag, const FrameArgs* __restrict__ fa) {
  if (fa) seq = (unsigned)fa->seq;  // frame mode: the sequence number of THIS replay (the argument is frozen in the graph)
  // out / fo: page-locked host memory seen through its device view -- the last kernel of a frame
  // writes the results where the host reads them, no copy operation follows
  __shared__ double sc[kTopkThreads];
  const int n = min((int)*d_n, cap);
  int k = (k_want >= 0 && k_want < n) ? k_want : n;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    fo->topk_overflow = (k > k_cap) ? (unsigned)k : 0u;
    fo->n_out = (unsigned)min(k, k_cap);
    fo->st = *st;  // (every kernel that updates the statistics has finished)
    if (gp) fo->g = *gp;
    if (pfp) fo->pre = *pfp;
  }
  k = min(k, k_cap);
  // Every workgroup reports when its writes into host memory are out; the last one writes the call's sequence
  // number behind them (page-locked memory), which the host may poll instead of waiting for the stream
  // (wait_topk).
  // (a system-scope fence is a write-back of the L2 on this GPU: only waves that wrote something pay for one.)
  // EVERY wave that stored into host memory fences its own stores before the barrier: a workgroup barrier does
  // not wait for another wave's outstanding stores (no vmcnt(0) at workgroup scope outside tgsplit mode), so a
  // fence by thread 0 alone would cover wave 0's stores only.
  auto report = [&](bool wrote) {
    if (!done_flag) return;  // uniform
    if (__any(wrote)) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(&st->topk_blocks, 1u) == gridDim.x - 1u) {
        __threadfence_system();
        *reinterpret_cast<volatile unsigned*>(done_flag) = seq;
      }
    }
  };
  const int base = blockIdx.x * kTopkThreads;
  if (base >= n) {  // uniform
    report(blockIdx.x == 0 && threadIdx.x == 0);  // (thread 0 of workgroup 0 wrote the statistics)
    return;
  }
  const int i = base + threadIdx.x;
  __shared__ unsigned long long keys[kRankKeys];
  __shared__ int inexact;
  int rank = rank_by_keys(recs, n, i, keys, &inexact);
  const bool general = rank < 0;  // (uniform) longer than the key stage, or a score that is no float
  const double si = (general && i < n) ? recs[i].score : 0.0;
  if (general) rank = 0;
  for (int j0 = 0; general && j0 < n; j0 += kTopkThreads) {
    __syncthreads();
    sc[threadIdx.x] = (j0 + (int)threadIdx.x < n) ? recs[j0 + threadIdx.x].score : -__builtin_inf();
    __syncthreads();
    // (scores beyond n are padded so that they never count.)  Eight LDS reads in flight and no branch:
    // one read per iteration, each waited for, behind a short-circuit compare made this loop the
    // whole kernel.
    for (int t = 0; t < kTopkThreads; t += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = sc[t + u];
#pragma unroll
      for (int u = 0; u < 8; u++)
        rank += (int)(v[u] > si) | ((int)(v[u] == si) & (int)(j0 + t + u < i));
    }
  }
  const bool wrote = i < n && rank < k;
  if (wrote) out[rank] = recs[i];
  report(wrote || (blockIdx.x == 0 && threadIdx.x == 0));
}

// Polling returns ~10 us earlier than the runtime's wait for the kernel's completion signal (the results
// behind which the flag sits are in host memory by then; two such waits per detect step: 0.713 -> 0.692 ms).
// Cost and bounds: the waiting thread spins on its core for `wait_spin_us` (50 us by default: a detect step's two
// waits are normally shorter than that by the time the host reaches them), then goes on polling but yields the
// core between looks (sched_yield: a caller with more contexts than spare cores -- ag2_pipe, one thread per
// device -- does not hold cores for whole steps), and after 5 ms waits for the stream the ordinary way.
// ag2_set_wait_mode(c, 0, 0) / AG2_POLL=0 turn the polling off.  After the stream wait the flag MUST hold the
// sequence number (the kernel that writes it has finished): anything else is reported, not ignored.
int wait_flag(ag2_ctx* c, size_t flag_off, unsigned want) {
  if (!c->h_pin_dev) {
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  return wait_flag_at(c, reinterpret_cast<const volatile unsigned*>(pin_small(c) + flag_off), want);
}
int wait_flag_at(ag2_ctx* c, const volatile unsigned* flag, unsigned want) {
  if (c->wait_poll && flag) {
    const auto t0 = std::chrono::steady_clock::now();
    const auto spin_for = std::chrono::microseconds(c->wait_spin_us);
    bool yielding = false;
    for (;;) {
      for (int spin = 0; spin < 256; spin++) {
        if (*flag == want) {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          return 0;
        }
        __builtin_ia32_pause();
      }
      const auto dt = std::chrono::steady_clock::now() - t0;
      if (dt > std::chrono::milliseconds(5)) break;
      if (yielding || dt > spin_for) {
        yielding = true;
        c->poll_yields++;
        sched_yield();
      }
    }
    c->poll_fallbacks++;
  }
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  if (c->wait_poll && flag && *flag != want)
    return set_err(c, AG2_ERR_STATE, "the stream is idle but the kernel's flag does not hold the call's sequence number (have " +
                                         std::to_string(*flag) + ", want " + std::to_string(want) + ")");
  return 0;
}

int launch_topk(ag2_ctx* c, const ag2_hypothesis* d_recs, const unsigned* d_n, size_t cap, size_t k_cap,
                ag2_hypothesis* d_out, FrameOut* d_fo, const GridDesc* gp, const PreFrame* pfp) {
  hipLaunchKernelGGL(k_topk, dim3((unsigned)((std::max<size_t>(cap, 1) + kTopkThreads - 1) / kTopkThreads)),
                     dim3(kTopkThreads), 0, c->stream, d_recs, d_n, (int)cap, c->p.num_selected, (int)k_cap, d_out,
                     d_fo, c->d_stats.as<DevStats>(), gp, pfp, (++c->topk_seq == 0u ? ++c->topk_seq : c->topk_seq),
                     c->h_pin_dev ? reinterpret_cast<unsigned*>(pin_small_dev(c) + kPinDoneFlag) : (unsigned*)nullptr,
                     (const FrameArgs*)nullptr);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

namespace {

int frame_pin_reserve(ag2_ctx* c, ag2_frame_state* f) {
  const size_t off_idx = 64;
  const size_t off_out = (off_idx + c->fm_s_max * 4 + 63) & ~size_t(63);
  const size_t off_rec = (off_out + sizeof(FrameOut) + 63) & ~size_t(63);
  const size_t need = off_rec + f->k_cap * sizeof(ag2_hypothesis) + 64;
  const bool moved = off_out != f->off_out;
  f->off_idx = off_idx;
  f->off_out = off_out;
  f->off_rec = off_rec;
  if (need <= f->h_pin_bytes) {
    if (moved) {  // the results' place holds something else: below every sequence number before it is polled
      AG2_HIP(c, hipStreamSynchronize(c->stream));
      reinterpret_cast<FrameOut*>(f->h_pin + off_out)->done_seq = 0u;
    }
    return 0;
  }
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  f->h_pin = nullptr;
  f->h_pin_bytes = 0;
  AG2_HIP(c, hipHostMalloc((void**)&f->h_pin, need, kPinFlags));
  f->h_pin_bytes = need;
  reinterpret_cast<FrameOut*>(f->h_pin + off_out)->done_seq = 0u;
  void* dv = nullptr;
  AG2_HIP(c, hipHostGetDevicePointer(&dv, f->h_pin, 0));
  c->fm_args_dev = (const FrameArgs*)dv;
  return 0;
}

// A captured graph holds the addresses (and the few sizes) it was captured with: every buffer a frame
// touches goes into this signature, and a graph is replayed only while it still matches.
unsigned long long frame_signature(const ag2_ctx* c, const ag2_frame_state* f) {
  const void* ptrs[] = {c->d_xyz_in.p, c->d_bounds.p, c->d_griddesc.p, c->d_key.p, c->d_cell.p, c->d_perm.p,
                        c->d_sorted.p, c->d_nrm.p, c->d_stats.p, c->d_hc.p, c->d_sample_q.p, c->d_frames.p,
                        c->d_frame_ok.p, c->d_table.p, c->d_tab_off.p, c->d_tab_keep.p, c->d_arena.p,
                        c->d_overflow.p, c->d_gscratch.p, c->d_gpos.p, c->d_lists.p, c->d_pairs.p, c->d_list2.p, c->d_images.p,
                        c->d_logits.p, c->d_act1.p, c->d_fcpart.p, c->d_sel.p, c->d_flags.p, c->d_desc.p,
                        c->d_scan.p, f->h_pin, c->net.w1x.p, c->net.w2x.p, c->net.w3x.p,
                        c->net.b1.p, c->net.b2.p, c->net.b3.p, c->net.w4.p, c->net.b4.p, (const void*)c->stream,
                        c->d_raw.p, c->d_bitmap.p, c->d_wrank.p, c->d_preframe.p, c->d_cand.p, c->d_samples.p, c->d_rlist.p,
                        c->d_cluster.p, c->d_cluster_tmp.p};
  unsigned cell_bits = 0;
  memcpy(&cell_bits, &f->fs.cell, 4);
  const unsigned long long vals[] = {c->fm_n_max, c->fm_s_max, c->fm_cap_cells, c->arena_points, c->list_ints, f->cap_img,
                                     (unsigned long long)c->sweep_gcap, (unsigned long long)c->sweep_g2,
                                     (unsigned long long)c->sweep_gpos_cap * 2ull + (c->fm_skip_stage1 ? 1ull : 0ull),
                                     (unsigned long long)c->p.num_selected, f->k_cap, (unsigned long long)f->cap_p,
                                     (unsigned long long)c->min_inliers,
                                     (unsigned long long)c->origin_set, (unsigned long long)f->raw, f->fs.raw_max,
                                     f->fs.cap_words, f->fs.cand_cap, (unsigned long long)cell_bits,
                                     (unsigned long long)f->fs.filter_workspace};
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](unsigned long long v) {
    for (int b = 0; b < 8; b++) {
      h ^= (v >> (8 * b)) & 255ull;
      h *= 1099511628211ull;
    }
  };
  for (const void* p : ptrs) mix((unsigned long long)(uintptr_t)p);
  for (unsigned long long v : vals) mix(v);
  for (int a = 0; a < 3; a++) {
    unsigned u;
    memcpy(&u, &c->origin[a], 4);
    mix(u);
  }
  for (int k = 0; k < 6; k++) {  // (the workspace is a by-value argument of the front end's kernels)
    unsigned long long u;
    memcpy(&u, &c->p.workspace[k], 8);
    mix(u);
  }
  return h;
}

void drop_graph(ag2_frame_state* f) {
  if (f->exec) (void)hipGraphExecDestroy(f->exec);
  if (f->graph) (void)hipGraphDestroy(f->graph);
  f->exec = nullptr;
  f->graph = nullptr;
  f->graph_valid = false;
}

// Everything of a frame behind the pack + extent kernel, on c->stream, at the fixed maximum shapes;
// nothing in here waits for the host or allocates once the buffers have their size (the first,
// uncaptured run gives them that).
int enqueue_frame(ag2_ctx* c, ag2_frame_state* f, int do_prune) {
  const size_t n_max = c->fm_n_max, s_max = c->fm_s_max, cap_cells = c->fm_cap_cells;
  const int R = c->p.num_orientations;
  const size_t n_slots = s_max * (size_t)R;
  DevStats* st = c->d_stats.as<DevStats>();
  // -- the front end of a raw frame: workspace filter, voxel grid, uniform sub-sampling ---------------
  if (f->raw) {
    const int rc0 = enqueue_front_frame(c, f->fs);
    if (rc0) return rc0;
  }
  GridDesc* gp = c->d_griddesc.as<GridDesc>();
  // -- K0 search grid (its description derived from the extent partials inside k_cell_count, or left
  // in memory by the front end) ---------------------------------------------------------------------
  AG2_HIP(c, c->d_key.reserve(n_max * 8));
  const size_t cell_words = ((cap_cells + 1) + 3) & ~size_t(3);
  const size_t ctl_words = (scan_ctl_words((int)cap_cells + 1) + 3) & ~size_t(3);
  AG2_HIP(c, c->d_cell.reserve((cell_words + ctl_words) * 4));
  AG2_HIP(c, c->d_perm.reserve(n_max * 4));
  AG2_HIP(c, c->d_sorted.reserve((n_max + 1) * 16));  // + the NaN point behind the cloud
  AG2_HIP(c, c->d_nrm.reserve(n_max * 16));
  unsigned* cell = c->d_cell.as<unsigned>();
  AG2_HIP(c, hipMemsetAsync(cell, 0, (cell_words + ctl_words) * 4, c->stream));
  int rc = launch_grid_frame(c, cell, cell + cell_words);
  if (rc) return rc;
  // -- K1 normals ----------------------------------------------------------------------------------
  rc = launch_normals(c);
  if (rc) return rc;
  // -- samples (indices in the page-locked block, padded with -1), K2 frames, K3 sweep -------------
  AG2_HIP(c, c->d_sample_q.reserve(s_max * 16));
  AG2_HIP(c, c->d_frames.reserve(s_max * 12 * 8));
  AG2_HIP(c, c->d_frame_ok.reserve(s_max * 4));
  AG2_HIP(c, c->d_tab_keep.reserve(std::max<size_t>(n_slots, 1)));
  // (a raw frame's sample indices were left in d_samples by the front end)
  rc = launch_sample_queries(c, f->raw ? c->d_samples.as<int>() : (const int*)((const char*)c->fm_args_dev + f->off_idx),
                             s_max, true);
  if (rc) return rc;
  rc = launch_frames(c, s_max, 0, 0);
  if (rc) return rc;
  c->defer_hyp_stats = true;  // (the compaction below takes the statistics along)
  c->hyp_stats_pending = false;
  rc = launch_sweep(c, s_max, 0, /*emit_lists=*/true, /*run_cleared=*/true);
  c->defer_hyp_stats = false;
  if (rc) return rc;
  // -- prune (predicate evaluated in the sweep) + image descriptors ------------------------------
  rc = compact_slots_async(c, n_slots, do_prune ? 1 : 0, c->d_list2, &st->n_list, /*with_descs=*/true);
  if (rc) return rc;
  if (c->desc_stride == 0) return set_err(c, AG2_ERR_CAPACITY, "frame mode: more than 65536 table slots");
  // -- K4 images, K5 LeNet, K6 score / threshold (at most cap_img images: the kernels clamp the list
  // length they read, the host sees a longer list in the statistics and repeats the frame) -----------
  const size_t cap_img = std::min(f->cap_img, n_slots);
  AG2_HIP(c, c->d_images.reserve(cap_img * 10800));
  AG2_HIP(c, c->d_logits.reserve(cap_img * 8));
  const unsigned* d_n = &st->n_list;
  rc = launch_render(c, c->d_arena.as<double>(), c->d_desc.as<long long>(),
                     (const int*)(c->d_desc.as<long long>() + c->desc_stride), cap_img,
                     c->d_images.as<uint8_t>(), f->cap_p, d_n);
  if (rc) return rc;
  rc = launch_lenet(c, c->d_images.as<uint8_t>(), cap_img, c->d_logits.as<float>(), -1, d_n);
  if (rc) return rc;
  rc = score_and_select_async(c, c->d_list2.as<int>(), cap_img, &st->n_sel, d_n);
  if (rc) return rc;
  // -- grasp clusters between the threshold and the top-k (grasp_detector.cpp:228-236), when asked for ----
  const ag2_hypothesis* d_res = c->d_sel.as<ag2_hypothesis>();
  const unsigned* d_nres = &st->n_sel;
  if (c->min_inliers > 0) {
    rc = cluster_async(c, d_res, cap_img, d_nres, c->min_inliers, &st->n_clu);
    if (rc) return rc;
    d_res = c->d_cluster.as<ag2_hypothesis>();
    d_nres = &st->n_clu;
  }
  // -- top-k on the device, written straight into the page-locked block ------------------------------
  ag2_hypothesis* d_rec = (ag2_hypothesis*)((char*)c->fm_args_dev + f->off_rec);
  FrameOut* d_fo = (FrameOut*)((char*)c->fm_args_dev + f->off_out);
  hipLaunchKernelGGL(k_topk, dim3((unsigned)((cap_img + kTopkThreads - 1) / kTopkThreads)), dim3(kTopkThreads), 0,
                     c->stream, d_res, d_nres, (int)cap_img, c->p.num_selected,
                     (int)f->k_cap, d_rec, d_fo, st, gp,
                     f->raw ? c->d_preframe.as<PreFrame>() : (const PreFrame*)nullptr, 0u, &d_fo->done_seq,
                     c->fm_args_dev);
  AG2_HIP(c, hipGetLastError());
  return 0;
}

// what one frame call asks for (either entry point)
struct FrameIn {
  const void* xyz = nullptr;
  int on_device = 0;
  size_t n = 0, stride = 12;
  bool raw = false;  // ag2_detect_frame_raw
  int filter_ws = 1;
  double voxel_size = 0.003;
  size_t num_samples = 0;
  uint64_t sample_seed = 0;
  const int32_t* sample_idx = nullptr;  // ag2_detect_frame
  size_t s = 0;
  uint64_t seed = 0;
  int do_prune = 1;
};

// the step-by-step path for one frame (also what sizes the buffers and teaches the shapes)
int frame_stepwise(ag2_ctx* c, const void* d_xyz, const FrameIn& in, ag2_hypothesis* selected, size_t cap,
                   size_t* n_selected, size_t* n_scored, size_t* n_voxels, size_t* s_used) {
  c->fm_on = false;
  c->fm_grid_ready = false;
  int rc = 0;
  size_t s = in.s;
  if (in.raw) {
    size_t m = 0, k = 0;
    rc = ag2_preprocess_cloud_device(c, d_xyz, in.n, in.stride, in.filter_ws, 1, in.voxel_size, &m);
    if (!rc) rc = ag2_subsample_uniformly(c, in.num_samples, in.sample_seed, nullptr, 0, &k);  // indices stay on the device
    if (n_voxels) *n_voxels = m;
    s = k;
  } else {
    rc = ag2_set_cloud_device(c, d_xyz, in.n, in.stride);
  }
  if (s_used) *s_used = s;
  if (!rc) rc = ag2_compute_normals(c);
  if (!rc)
    rc = ag2_detect(c, in.raw ? nullptr : in.sample_idx, nullptr, s, 0, in.seed, in.do_prune, selected, cap, n_selected,
                    nullptr, 0, n_scored);
  return rc;
}

// clears the frame-mode switches of the context on every way out of the fixed-shape block
struct FrameModeGuard {
  ag2_ctx* c;
  explicit FrameModeGuard(ag2_ctx* ctx) : c(ctx) {}
  ~FrameModeGuard() {
    c->fm_on = false;
    c->fm_grid_ready = false;
  }
};

// shapes for the frames to come, from a frame that just ran step by step (the context holds its results)
void learn_shapes(ag2_ctx* c, ag2_frame_state* f, const FrameIn& in, size_t n_vox, size_t s_used) {
  const int R = c->p.num_orientations;
  const size_t n = in.n;
  if (f->raw != in.raw) {
    c->fm_n_max = c->fm_s_max = 0;
    f->fs = FrontShapes{};
  }
  f->raw = in.raw;
  const size_t n_cloud = in.raw ? n_vox : n;
  c->fm_n_max = std::max(c->fm_n_max, n_cloud + n_cloud / 8 + 1024);
  c->fm_s_max = in.raw ? s_used : std::max(c->fm_s_max, s_used);
  // cell table: this frame's grid with 4 cells (4 cm) of room per axis and a fifth on top -- the scan, the cell
  // sort and the fill run over the whole table, so generous room costs every frame (2.3 x the cells cost the
  // headline cloud 40 us per frame); a frame that outgrows it is repeated step by step and the table grows
  long long cells = 1;
  for (int a = 0; a < 3; a++) cells *= (long long)(c->grid.dims[a] + 4);
  cells = std::min<long long>(cells + cells / 5, 1ll << 30);
  c->fm_cap_cells = std::max<size_t>(c->fm_cap_cells, (size_t)cells);
  if (in.raw) {
    f->fs.raw_max = std::max(f->fs.raw_max, n + n / 8 + 1024);
    // bitmap of the voxel lattice: this frame's with half as much again (at least 16 voxels) of room per
    // axis -- an object taller than anything seen so far must not cost a frame --, but never more than the
    // workspace can hold when it filters
    double words = 1.0;
    for (int a = 0; a < 3; a++) words *= (double)(c->last_vox_dims[a] + std::max(16, c->last_vox_dims[a] / 2));
    words = words / 32.0 + 2.0;
    if (in.filter_ws) {
      double wsw = 1.0;
      for (int a = 0; a < 3; a++)
        wsw *= floor((c->p.workspace[2 * a + 1] - c->p.workspace[2 * a]) / (double)(float)in.voxel_size) + 2.0;
      if (wsw >= 1.0) words = std::min(words, wsw / 32.0 + 2.0);
    }
    f->fs.cap_words = std::max(f->fs.cap_words, (size_t)std::min(words, 1.0e9));
    f->fs.n_max = c->fm_n_max;
    f->fs.num_samples = s_used;
    f->fs.cand_cap = cand_capacity(s_used);
    f->fs.cell = (float)in.voxel_size;
    f->fs.filter_workspace = in.filter_ws ? 1 : 0;
  }
  // images: twice what this frame scored, in whole batches of 256 (BASELINE.json configs[4]:
  // batch_size 256), never below the shapes already in force
  const size_t want_img = (((size_t)c->n_img * 2 + 512) + 255) / 256 * 256;
  f->cap_p = std::max(f->cap_p, 2 * c->max_p);  // which renderers the sequence launches
  f->cap_img = std::min(std::max(f->cap_img, want_img), c->fm_s_max * (size_t)R);
  f->k_cap = (c->p.num_selected >= 0) ? std::min<size_t>((size_t)c->p.num_selected, f->cap_img) : f->cap_img;
  f->shapes_known = c->fm_s_max * (size_t)R <= 65536;
  // the sweep's long-list stage: left out of the sequence while no frame has queued a sample for it (a frame that
  // does is repeated step by step and the shapes are learned again, with the stage)
  c->fm_skip_stage1 = c->cnt.n_overflow_samples == 0;
}

FrameIn pending_in(const ag2_frame_state::Pending& p) {
  FrameIn in;
  in.xyz = p.xyz;
  in.on_device = p.on_device;
  in.n = p.n;
  in.stride = p.stride;
  in.raw = p.raw;
  in.filter_ws = p.filter_ws;
  in.voxel_size = p.voxel_size;
  in.num_samples = p.num_samples;
  in.sample_seed = p.sample_seed;
  in.sample_idx = p.idx.empty() ? nullptr : p.idx.data();
  in.s = p.idx.size();
  in.seed = p.seed;
  in.do_prune = p.do_prune;
  return in;
}

// the frame step by step, its results kept in the pending record, the shapes learned
int run_pending_stepwise(ag2_ctx* c, ag2_frame_state* f) {
  ag2_frame_state::Pending& p = f->pend;
  const FrameIn in = pending_in(p);
  const size_t s_req = in.raw ? in.num_samples : in.s;
  const size_t cap = std::max<size_t>(1, s_req * (size_t)c->p.num_orientations);
  p.recs.resize((c->p.num_selected >= 0) ? std::max<size_t>(1, std::min<size_t>(cap, (size_t)c->p.num_selected)) : cap);
  f->info.stepwise_runs++;
  size_t n_vox = in.n, s_used = in.s;
  p.n_selected = p.n_scored = 0;
  const int rc = frame_stepwise(c, p.xyz, in, p.recs.data(), p.recs.size(), &p.n_selected, &p.n_scored, &n_vox, &s_used);
  p.n_voxels = n_vox;
  p.finished = true;
  p.rc = rc;
  if (rc) return rc;
  // a raw frame with no more voxels than num_samples takes every point as a sample (grasp_detector.cpp:322-330):
  // that is not what the fixed-shape sub-sampling does, such frames stay on this path
  if (!p.unsupported && !(in.raw && s_used != in.num_samples)) learn_shapes(c, f, in, n_vox, s_used);
  return 0;
}

// ---- submit: everything of a frame up to (not including) the wait for its results -------------------
int frame_submit_impl(ag2_ctx* c, const FrameIn& in) {
  if (c->p.n_cams != 1) return set_err(c, AG2_ERR_ARG, "frames are single-camera clouds");
  if (in.stride < 12 || in.stride % 4 != 0) return set_err(c, AG2_ERR_ARG, "bad stride");
  if (in.n > (size_t)1 << 30) return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 points");
  if (in.raw && !((float)in.voxel_size > 0.f)) return set_err(c, AG2_ERR_ARG, "voxel_size must be positive");
  if (!c->net.loaded) return set_err(c, AG2_ERR_STATE, "lenet weights not loaded");
  if (!c->fm) c->fm = new ag2_frame_state();
  ag2_frame_state* f = c->fm;
  if (f->pend.active) return set_err(c, AG2_ERR_STATE, "a frame is in flight on this context: ag2_wait_frame first");
  const int R = c->p.num_orientations;
  const size_t n = in.n;
  ag2_frame_state::Pending& p = f->pend;
  p = ag2_frame_state::Pending{};
  // A cloud in host memory goes through page-locked staging (the host copies, then a true asynchronous
  // DMA: the call does not wait for the transfer) into a device staging buffer.
  const void* d_xyz = in.xyz;
  if (!in.on_device && n) {
    const size_t bytes = n * in.stride;
    if (bytes > f->h_stage_bytes) {
      AG2_HIP(c, hipStreamSynchronize(c->stream));
      if (f->h_stage) (void)hipHostFree(f->h_stage);
      f->h_stage = nullptr;
      f->h_stage_bytes = 0;
      const size_t want = bytes + bytes / 8 + 4096;
      AG2_HIP(c, hipHostMalloc((void**)&f->h_stage, want, hipHostMallocDefault));
      f->h_stage_bytes = want;
    }
    AG2_HIP(c, f->d_raw.reserve(bytes));
    memcpy(f->h_stage, in.xyz, bytes);
    AG2_HIP(c, hipMemcpyAsync(f->d_raw.p, f->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
    d_xyz = f->d_raw.p;
  }
  p.active = true;
  p.xyz = d_xyz;
  p.on_device = 1;
  p.n = n;
  p.stride = in.stride;
  p.raw = in.raw;
  p.filter_ws = in.filter_ws;
  p.voxel_size = in.voxel_size;
  p.num_samples = in.num_samples;
  p.sample_seed = in.sample_seed;
  p.seed = in.seed;
  p.do_prune = in.do_prune;
  if (!in.raw && in.s) p.idx.assign(in.sample_idx, in.sample_idx + in.s);
  f->info.frames++;
  const size_t s_req = in.raw ? in.num_samples : in.s;
  // Frames the captured sequence cannot take: more than 65536 table slots, an empty frame, the f32-input LeNet
  // kernels.
  p.unsupported = s_req * (size_t)R > 65536 || n == 0 || s_req == 0 || !c->net.use_x3;
  if (f->shapes_known && f->raw != in.raw) {  // the stream changed its entry point: learn the shapes again
    f->shapes_known = false;
    drop_graph(f);
  }
  bool stepwise = p.unsupported || !f->shapes_known;
  if (!stepwise) {
    if (in.raw)
      stepwise = n > f->fs.raw_max || in.num_samples != c->fm_s_max || (float)in.voxel_size != f->fs.cell ||
                 (in.filter_ws ? 1 : 0) != f->fs.filter_workspace;
    else
      stepwise = n > c->fm_n_max || in.s > c->fm_s_max;
  }
  if (stepwise) {
    // the first frame, frames that outgrow the shapes, unsupported settings: step by step, here and now
    const int rc = run_pending_stepwise(c, f);
    if (rc) p.active = false;
    return rc;
  }
  // ---- the frame at fixed shapes: pack + extent in front, then the sequence (graph or plain) ----
  FrameModeGuard guard(c);
  c->fm_on = true;
  c->fm_grid_ready = in.raw;
  auto fail = [&](int rc) {
    p.active = false;
    return rc;
  };
  int rc = frame_pin_reserve(c, f);
  if (rc) return fail(rc);
  FrameArgs* fa = (FrameArgs*)f->h_pin;
  fa->seed = in.seed;
  fa->slot_base = 0;
  fa->sample_seed = in.sample_seed;
  if (++f->seq == 0u) f->seq = 1u;  // (zero is what the flag starts from)
  fa->seq = f->seq;
  if (!in.raw) {
    int32_t* hidx = (int32_t*)(f->h_pin + f->off_idx);
    memcpy(hidx, in.sample_idx, in.s * 4);
    for (size_t i = in.s; i < c->fm_s_max; i++) hidx[i] = -1;
  }
  if (c->d_xyz_in.reserve(c->fm_n_max * 16) != hipSuccess || c->d_bounds.reserve((size_t)kBoundsBlocks * 8 * 4) != hipSuccess ||
      c->d_griddesc.reserve(sizeof(GridDesc)) != hipSuccess)
    return fail(set_err(c, AG2_ERR_HIP, "frame: device allocation failed"));
  if (in.raw) rc = front_pack_raw(c, d_xyz, n, in.stride, f->fs);
  else rc = pack_device_xyz(c, d_xyz, n, in.stride, c->d_xyz_in.as<float4>(), /*with_bounds=*/true, c->fm_n_max);
  if (rc) return fail(rc);
  p.replay = f->use_graph && f->graph_valid && f->do_prune == in.do_prune && f->sig_at_capture == frame_signature(c, f);
  if (p.replay) {
    if (hipGraphLaunch(f->exec, c->stream) != hipSuccess) return fail(set_err(c, AG2_ERR_HIP, "hipGraphLaunch failed"));
    f->info.graph_replays++;
  } else {
    drop_graph(f);
    const int lvl = c->stage_timing;
    c->stage_timing = 0;  // (events are not part of the sequence)
    rc = enqueue_frame(c, f, in.do_prune);
    c->stage_timing = lvl;
    if (rc) return fail(rc);
    f->info.plain_runs++;
  }
  return 0;
}

// ---- wait: the results of the submitted frame -------------------------------------------------------
int frame_wait_impl(ag2_ctx* c, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
                    size_t* n_voxels) {
  ag2_frame_state* f = c->fm;
  if (!f || !f->pend.active) return set_err(c, AG2_ERR_STATE, "no frame in flight on this context");
  ag2_frame_state::Pending& p = f->pend;
  const int R = c->p.num_orientations;
  if (!p.finished) {
    const FrameIn in = pending_in(p);
    const FrameOut* fo = (const FrameOut*)(f->h_pin + f->off_out);
    if (wait_flag_at(c, &fo->done_seq, f->seq) != 0) {  // (k_topk's flag behind the results, then the stream)
      p.active = false;
      return AG2_ERR_HIP;
    }
    const unsigned flags = fo->st.err_flags;
    const bool bad = (flags & (1u | 2u | 8u)) != 0 || fo->g.ncells < 0 || fo->topk_overflow != 0 ||
                     (size_t)fo->st.n_list > std::min(f->cap_img, c->fm_s_max * (size_t)R) ||
                     (int)fo->st.max_p > render_capacity_for(f->cap_p) || (in.raw && fo->pre.flags != 0u) ||
                     (c->fm_skip_stage1 && fo->st.n_overflow > 0);
    if (bad) {
      if (fo->g.ncells == -2) {
        p.active = false;
        return set_err(c, AG2_ERR_ARG, "a point lies below the grid origin given to ag2_set_grid_origin");
      }
      f->info.fallbacks++;
      f->info.last_fallback = (int64_t)(flags & (1u | 2u | 8u)) | (fo->g.ncells < 0 ? 16 : 0) |
                              (fo->topk_overflow ? 32 : 0) |
                              ((size_t)fo->st.n_list > std::min(f->cap_img, c->fm_s_max * (size_t)R) ? 64 : 0) |
                              ((int)fo->st.max_p > render_capacity_for(f->cap_p) ? 128 : 0) |
                              ((c->fm_skip_stage1 && fo->st.n_overflow > 0) ? (1ll << 40) : 0) |
                              (in.raw ? ((int64_t)fo->pre.flags << 8) : 0);
      f->shapes_known = false;  // learn the shapes again from the step-by-step run
      const int rc = run_pending_stepwise(c, f);
      if (rc) {
        p.active = false;
        return rc;
      }
    } else {
      // capture for the frames to come (right after the run that gave every buffer its size)
      if (!p.replay && f->use_graph) {
        if (c->stream == nullptr) {
          f->info.capture_refused++;  // the legacy default stream cannot be captured
        } else {
          FrameModeGuard guard(c);
          c->fm_on = true;
          c->fm_grid_ready = in.raw;
          const unsigned long long sig0 = frame_signature(c, f);
          const int lvl = c->stage_timing;
          c->stage_timing = 0;
          (void)hipStreamSynchronize(c->stream);  // (the results were polled for: let the runtime see the stream idle)
          hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
          int rc2 = (e == hipSuccess) ? enqueue_frame(c, f, in.do_prune) : AG2_ERR_HIP;
          hipGraph_t g = nullptr;
          if (e == hipSuccess) e = hipStreamEndCapture(c->stream, &g);
          c->stage_timing = lvl;
          if (e == hipSuccess && rc2 == 0 && g && sig0 == frame_signature(c, f) &&
              hipGraphInstantiate(&f->exec, g, nullptr, nullptr, 0) == hipSuccess) {
            // (the executable graph's first launch would otherwise carry its upload to the device)
            (void)hipGraphUpload(f->exec, c->stream);
            (void)hipStreamSynchronize(c->stream);
            f->graph = g;
            f->graph_valid = true;
            f->sig_at_capture = sig0;
            f->do_prune = in.do_prune;
            f->info.captures++;
          } else {
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
            f->info.capture_failed++;
          }
        }
      }
      // results
      const size_t n_cloud = in.raw ? (size_t)fo->pre.n_vox : in.n;
      const size_t s = in.raw ? in.num_samples : in.s;
      c->n = n_cloud;
      c->n_valid = (size_t)fo->g.n_valid;
      c->grid = fo->g;
      c->min_z = fo->g.min_z;
      c->has_cloud = c->has_normals = true;
      c->normals_pending = c->grid_pending = false;
      c->bounds_known = false;
      c->s = s;
      c->slot_base = 0;
      c->n_resident_samples = in.raw ? s : 0;
      c->n_img = fo->st.n_list;
      c->max_p = (int)fo->st.max_p;
      memset(&c->cnt, 0, sizeof(c->cnt));
      c->cnt.n_points = (int64_t)n_cloud;
      c->cnt.n_valid_points = fo->g.n_valid;
      c->cnt.n_samples = (int64_t)s;
      c->cnt.n_frames = fo->st.n_frames;
      c->cnt.n_hypotheses = fo->st.n_hyp;
      c->cnt.n_pruned = fo->st.n_list;
      c->cnt.n_scored = fo->st.n_list;
      c->cnt.sum_k1 = (int64_t)fo->st.sum_k1;
      c->cnt.sum_k2 = (int64_t)fo->st.sum_k2;
      c->cnt.sum_kcrop = (int64_t)fo->st.sum_kcrop;
      c->cnt.sum_p = (int64_t)fo->st.sum_p;
      c->cnt.n_overflow_samples = fo->st.n_overflow;
      c->cnt.list_points = (int64_t)fo->st.list_top;
      const size_t k = fo->n_out;
      c->cnt.n_selected = (int64_t)k;
      *n_selected = k;
      if (n_scored) *n_scored = fo->st.n_list;
      if (n_voxels) *n_voxels = n_cloud;
      if (k > cap) {
        // the caller's buffer is too small: the frame stays retrievable (a second wait with more room brings it;
        // *n_selected says how much) instead of being lost
        const ag2_hypothesis* r = (const ag2_hypothesis*)(f->h_pin + f->off_rec);
        p.recs.assign(r, r + k);
        p.n_selected = k;
        p.n_scored = fo->st.n_list;
        p.n_voxels = n_cloud;
        p.finished = true;
        return set_err(c, AG2_ERR_CAPACITY, "detect_frame: output capacity too small (the frame is kept: wait again with room for n_selected records)");
      }
      p.active = false;
      if (k) memcpy(selected, f->h_pin + f->off_rec, k * sizeof(ag2_hypothesis));
      return 0;
    }
  }
  // the frame ran step by step (inside the submit, or just now as a fallback), or an earlier wait found the
  // caller's buffer too small: its results were kept
  *n_selected = p.n_selected;
  if (n_scored) *n_scored = p.n_scored;
  if (n_voxels) *n_voxels = p.n_voxels;
  if (p.n_selected > cap)
    return set_err(c, AG2_ERR_CAPACITY, "detect_frame: output capacity too small (the frame is kept: wait again with room for n_selected records)");
  p.active = false;
  if (p.n_selected) memcpy(selected, p.recs.data(), p.n_selected * sizeof(ag2_hypothesis));
  return 0;
}

// (host time inside each half, for callers that have to attribute a slow frame: ag2_get_wait_info)
inline int64_t us_since(std::chrono::steady_clock::time_point t0) {
  return (int64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}
int frame_submit(ag2_ctx* c, const FrameIn& in) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = frame_submit_impl(c, in);
  c->last_submit_us = us_since(t0);
  return rc;
}
int frame_wait(ag2_ctx* c, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
               size_t* n_voxels) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = frame_wait_impl(c, selected, cap, n_selected, n_scored, n_voxels);
  c->last_wait_us = us_since(t0);
  return rc;
}

int detect_frame_impl(ag2_ctx* c, const FrameIn& in, ag2_hypothesis* selected, size_t cap, size_t* n_selected,
                      size_t* n_scored, size_t* n_voxels) {
  const int rc = frame_submit(c, in);
  if (rc) return rc;
  const int rw = frame_wait(c, selected, cap, n_selected, n_scored, n_voxels);
  // (the synchronous call has no second half to come back to: a frame the wait kept for a caller with a larger
  //  buffer -- AG2_ERR_CAPACITY, *n_selected says how large -- is dropped, the next call starts a new one)
  if (rw && c->fm) c->fm->pend.active = false;
  return rw;
}

}  // namespace

void frame_release(ag2_ctx* c) {
  ag2_frame_state* f = c->fm;
  if (!f) return;
  drop_graph(f);
  if (f->h_pin) (void)hipHostFree(f->h_pin);
  if (f->h_stage) (void)hipHostFree(f->h_stage);
  f->d_raw.release();
  delete f;
  c->fm = nullptr;
}

}  // namespace ag2

extern "C" {

int ag2_stream_configure(ag2_ctx* c, size_t max_points, size_t max_samples, int use_graph) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->fm) c->fm = new ag2_frame_state();
  ag2_frame_state* f = c->fm;
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  drop_graph(f);
  f->use_graph = use_graph != 0;
  f->shapes_known = false;
  c->fm_n_max = max_points;
  c->fm_s_max = max_samples;
  c->fm_cap_cells = 0;
  f->cap_img = f->k_cap = 0;
  f->cap_p = 0;
  f->raw = false;
  f->fs = FrontShapes{};
  memset(&f->info, 0, sizeof(f->info));
  return 0;
}

int ag2_detect_frame(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                     const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune,
                     ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored) {
  if (!c || !n_selected || (s && !sample_idx) || (n && !xyz)) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  FrameIn in;
  in.xyz = xyz;
  in.on_device = xyz_on_device;
  in.n = n;
  in.stride = stride_bytes;
  in.sample_idx = sample_idx;
  in.s = s;
  in.seed = seed;
  in.do_prune = do_prune;
  return detect_frame_impl(c, in, selected, cap, n_selected, n_scored, nullptr);
}

int ag2_detect_frame_raw(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                         int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                         uint64_t seed, int do_prune, ag2_hypothesis* selected, size_t cap, size_t* n_selected,
                         size_t* n_scored, size_t* n_voxels) {
  if (!c || !n_selected || (n && !xyz)) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  FrameIn in;
  in.xyz = xyz;
  in.on_device = xyz_on_device;
  in.n = n;
  in.stride = stride_bytes;
  in.raw = true;
  in.filter_ws = filter_workspace ? 1 : 0;
  in.voxel_size = voxel_size;
  in.num_samples = num_samples;
  in.sample_seed = sample_seed;
  in.seed = seed;
  in.do_prune = do_prune;
  return detect_frame_impl(c, in, selected, cap, n_selected, n_scored, n_voxels);
}

// ---- the asynchronous form: submit a frame, do something else (submit the next one to another context),
// wait for its results ----
int ag2_submit_frame(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                     const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune) {
  if (!c || (s && !sample_idx) || (n && !xyz)) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  FrameIn in;
  in.xyz = xyz;
  in.on_device = xyz_on_device;
  in.n = n;
  in.stride = stride_bytes;
  in.sample_idx = sample_idx;
  in.s = s;
  in.seed = seed;
  in.do_prune = do_prune;
  return frame_submit(c, in);
}

int ag2_submit_frame_raw(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                         int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                         uint64_t seed, int do_prune) {
  if (!c || (n && !xyz)) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  FrameIn in;
  in.xyz = xyz;
  in.on_device = xyz_on_device;
  in.n = n;
  in.stride = stride_bytes;
  in.raw = true;
  in.filter_ws = filter_workspace ? 1 : 0;
  in.voxel_size = voxel_size;
  in.num_samples = num_samples;
  in.sample_seed = sample_seed;
  in.seed = seed;
  in.do_prune = do_prune;
  return frame_submit(c, in);
}

int ag2_wait_frame(ag2_ctx* c, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
                   size_t* n_voxels) {
  if (!c || !n_selected) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  return frame_wait(c, selected, cap, n_selected, n_scored, n_voxels);
}

// ---- ag2_pipe: one caller thread, several frames in flight ------------------------------------------
// `depth` contexts on one GPU, each with its own stream, taken in turn: while the tail of frame k (LeNet,
// selection) runs on one stream, the front of frame k + 1 (transfer, front end, grid, normals, sweep) runs on
// the next -- one cloud alone does not fill 256 CUs.  Results come back in submission order.
struct ag2_pipe {
  std::vector<ag2_ctx*> ctx;
  size_t next_submit = 0, next_wait = 0, in_flight = 0;
  std::string err;
};

ag2_pipe* ag2_pipe_create(const ag2_params* p, int device_id, int depth) {
  if (!p || depth < 1 || depth > 8) return nullptr;
  ag2_pipe* q = new ag2_pipe();
  for (int k = 0; k < depth; k++) {
    ag2_ctx* c = ag2_create(p, device_id);
    if (!c) {
      ag2_pipe_destroy(q);
      return nullptr;
    }
    (void)ag2_set_stage_timing(c, 0);  // (events serialise the streams)
    q->ctx.push_back(c);
  }
  return q;
}

void ag2_pipe_destroy(ag2_pipe* q) {
  if (!q) return;
  for (ag2_ctx* c : q->ctx) ag2_destroy(c);
  delete q;
}

const char* ag2_pipe_last_error(const ag2_pipe* q) { return q ? q->err.c_str() : "null pipe"; }
ag2_ctx* ag2_pipe_context(ag2_pipe* q, int k) { return (q && k >= 0 && (size_t)k < q->ctx.size()) ? q->ctx[(size_t)k] : nullptr; }

int ag2_pipe_lenet_load(ag2_pipe* q, const float* c1w, const float* c1b, const float* c2w, const float* c2b,
                        const float* f1w, const float* f1b, const float* f2w, const float* f2b) {
  if (!q) return AG2_ERR_ARG;
  for (ag2_ctx* c : q->ctx) {
    const int rc = ag2_lenet_load(c, c1w, c1b, c2w, c2b, f1w, f1b, f2w, f2b);
    if (rc) {
      q->err = ag2_last_error(c);
      return rc;
    }
  }
  return 0;
}

static int pipe_submitted(ag2_pipe* q, ag2_ctx* c, int rc) {
  if (rc) {
    q->err = ag2_last_error(c);
    return rc;
  }
  q->next_submit = (q->next_submit + 1) % q->ctx.size();
  q->in_flight++;
  return 0;
}

int ag2_pipe_submit_raw(ag2_pipe* q, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                        int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                        uint64_t seed, int do_prune) {
  if (!q) return AG2_ERR_ARG;
  if (q->in_flight == q->ctx.size()) {
    q->err = "pipe full: ag2_pipe_wait first";
    return AG2_ERR_STATE;
  }
  ag2_ctx* c = q->ctx[q->next_submit];
  return pipe_submitted(q, c, ag2_submit_frame_raw(c, xyz, xyz_on_device, n, stride_bytes, filter_workspace, voxel_size,
                                                   num_samples, sample_seed, seed, do_prune));
}

int ag2_pipe_submit(ag2_pipe* q, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                    const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune) {
  if (!q) return AG2_ERR_ARG;
  if (q->in_flight == q->ctx.size()) {
    q->err = "pipe full: ag2_pipe_wait first";
    return AG2_ERR_STATE;
  }
  ag2_ctx* c = q->ctx[q->next_submit];
  return pipe_submitted(q, c, ag2_submit_frame(c, xyz, xyz_on_device, n, stride_bytes, sample_idx, s, seed, do_prune));
}

int ag2_pipe_wait(ag2_pipe* q, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
                  size_t* n_voxels) {
  if (!q || !n_selected) return AG2_ERR_ARG;
  if (q->in_flight == 0) {
    q->err = "pipe empty: nothing was submitted";
    return AG2_ERR_STATE;
  }
  ag2_ctx* c = q->ctx[q->next_wait];
  const int rc = ag2_wait_frame(c, selected, cap, n_selected, n_scored, n_voxels);
  if (rc) q->err = ag2_last_error(c);
  if (rc == AG2_ERR_CAPACITY) return rc;  // (the frame is kept: the caller waits again with a larger buffer)
  q->next_wait = (q->next_wait + 1) % q->ctx.size();
  q->in_flight--;
  return rc;
}

int ag2_get_frame_info(ag2_ctx* c, ag2_frame_info* out) {
  if (!c || !out) return AG2_ERR_ARG;
  memset(out, 0, sizeof(*out));
  if (c->fm) {
    *out = c->fm->info;
    out->max_points = (int64_t)c->fm_n_max;
    out->max_samples = (int64_t)c->fm_s_max;
    out->max_cells = (int64_t)c->fm_cap_cells;
    out->max_images = (int64_t)c->fm->cap_img;
    out->graph_ready = c->fm->graph_valid ? 1 : 0;
  }
  return 0;
}

}  // extern "C"
