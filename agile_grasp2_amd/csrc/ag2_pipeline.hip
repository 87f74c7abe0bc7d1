// ag2_pipeline.hip -- C-ABI entry points of the hypothesis path: frames, hand sweep, prune,
// images, LeNet scoring, selection.  Orchestration only; the kernels live in k_*.hip.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ag2_internal.h"

using namespace ag2;

namespace {

int check_samples(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s) {
  if (!c->has_cloud) return set_err(c, AG2_ERR_STATE, "no cloud set");
  if (!c->has_normals)
    return set_err(c, AG2_ERR_STATE, "normals missing: call ag2_compute_normals or pass normals");
  if (sample_idx && sample_xyz)
    return set_err(c, AG2_ERR_ARG, "at most one of sample_idx / sample_xyz may be given");
  if (!sample_idx && !sample_xyz && s > c->n_resident_samples)
    return set_err(c, AG2_ERR_ARG,
                   "no sample_idx / sample_xyz given and fewer than s indices left by "
                   "ag2_subsample_uniformly");
  return 0;
}

int reset_stats(ag2_ctx* c) {
  // keep bounds (grid) intact: zero everything before them
  AG2_HIP(c, hipMemsetAsync(c->d_stats.p, 0, offsetof(DevStats, bounds), c->stream));
  return 0;
}

// fixed-slot table for the exchange step: the record of every occupied slot, zeros for the empty
// ones (their records in d_table are stale: the table is not cleared between runs)
__global__ void k_export_table(const uint4* __restrict__ table, const unsigned char* __restrict__ keep,
                               int n16, uint4* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  constexpr int kPer = (int)(sizeof(ag2_hypothesis) / 16);
  dst[i] = keep[i / kPer] ? table[i] : make_uint4(0u, 0u, 0u, 0u);
}

int read_stats(ag2_ctx* c, DevStats* hs) {
  AG2_HIP(c, hipMemcpyAsync(pin_small(c), c->d_stats.p, sizeof(DevStats), hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(hs, pin_small(c), sizeof(DevStats));
  return 0;
}

// counters and stage times a finished sweep leaves (hs: its statistics record)
void note_sweep(ag2_ctx* c, size_t s, const DevStats& hs, int compact_mode) {
  c->cnt.n_samples = (int64_t)s;
  c->cnt.n_frames = hs.n_frames;
  c->cnt.n_hypotheses = hs.n_hyp;
  c->cnt.sum_k2 = (int64_t)hs.sum_k2;
  c->cnt.sum_kcrop = (int64_t)hs.sum_kcrop;
  c->cnt.sum_p = (int64_t)hs.sum_p;
  c->cnt.n_overflow_samples = hs.n_overflow;  // handed to the sweep's long-list stage
  sweep_adapt_gpos(c, s, hs.n_overflow);
  c->cnt.list_points = (int64_t)hs.list_top;
  if (compact_mode >= 0) c->n_img = hs.n_list;
  c->max_p = (int)hs.max_p;
  stage_elapsed(c, &c->times.frames_ms, 0, 1);
  if (stage_event_on(c, 2)) {
    stage_elapsed(c, &c->times.sweep_ms, 1, 2);
    stage_elapsed(c, &c->times.sweep_overflow_ms, 2, 11);
  } else {  // timing level 1: the sweep as a whole
    stage_elapsed(c, &c->times.sweep_ms, 1, 11);
    c->times.sweep_overflow_ms = 0.f;
  }
  if (c->grid_pending) {
    stage_elapsed(c, &c->times.grid_ms, 12, 13);
    c->grid_pending = false;
  }
  if (c->normals_pending) {  // the stream has been synchronised: k_normals is long done
    stage_elapsed(c, &c->times.normals_ms, 9, 10);
    c->cnt.sum_k1 = (int64_t)hs.sum_k1;
    c->normals_pending = false;
  }
}

// frames + sweep for s samples; fills the slot table (and the arena when emit_lists).
// compact_mode >= 0 additionally queues the order-preserving compaction of the slot table into
// d_list2 (0: every hypothesis, 1: those surviving the prune) BEFORE the one host read-back, so the
// caller learns the list length (c->n_img) without a second round trip.
int run_hypotheses(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                   uint64_t slot_base, uint64_t seed, bool emit_lists, int compact_mode = -1,
                   bool defer_read = false) {
  if (s * (size_t)c->p.num_orientations > ((size_t)1 << 30))
    return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 table slots");
  c->s = s;
  c->slot_base = slot_base;
  for (int attempt = 0; attempt < 6; attempt++) {
    // with index samples the kernel that fetches them also clears the slot states and the per-run
    // statistics (two fills less)
    const bool fused_clear = s > 0 && (sample_idx || !sample_xyz);
    int rc = 0;
    if (fused_clear)
      AG2_HIP(c, c->d_tab_keep.reserve(std::max<size_t>(s * (size_t)c->p.num_orientations, 1)));
    else
      rc = reset_stats(c);
    if (rc) return rc;
    rc = upload_samples(c, sample_idx, sample_xyz, s, fused_clear);
    if (rc) return rc;
    AG2_HIP(c, stage_event(c, 0));
    rc = launch_frames(c, s, slot_base, seed);
    if (rc) return rc;
    AG2_HIP(c, stage_event(c, 1));
    c->defer_hyp_stats = compact_mode >= 0;
    c->hyp_stats_pending = false;
    rc = launch_sweep(c, s, slot_base, emit_lists, fused_clear);  // records ev[2] and ev[11]
    c->defer_hyp_stats = false;
    if (rc) return rc;
    if (compact_mode >= 0) {
      rc = compact_slots_async(c, s * (size_t)c->p.num_orientations, compact_mode, c->d_list2,
                               &c->d_stats.as<DevStats>()->n_list, /*with_descs=*/true);
      if (rc) return rc;
    }
    if (defer_read) return 0;  // (ag2_detect's one-round-trip form reads the statistics at its end)
    DevStats hs;
    rc = read_stats(c, &hs);
    if (rc) return rc;
    if (hs.err_flags & 8u) {
      // A cropped neighbourhood is longer than the sweep's global scratch: size the scratch to the
      // longest list of this run and repeat the run.  The reference crops into a list of any
      // length (hand_search.cpp:329-349); the only limit here is device memory.
      const size_t need = (((size_t)hs.max_k_over + 1023) / 1024) * 1024;
      if (need > (size_t)0x7fffffff - 1024)
        return set_err(c, AG2_ERR_CAPACITY, "a cropped neighbourhood exceeds 2^31 points");
      size_t free_b = 0, total_b = 0;
      AG2_HIP(c, hipMemGetInfo(&free_b, &total_b));
      // memory the resized scratch may take: what is free now plus what the old scratch returns
      const size_t budget = std::min<size_t>((free_b + c->d_gscratch.bytes) / 2, (size_t)64 << 30);
      const size_t per_wg = need * 5 * 4;
      const size_t g2 = std::min<size_t>(1024, budget / (per_wg + per_wg / 4 + 1));
      if (g2 < 1)
        return set_err(c, AG2_ERR_CAPACITY,
                       "a cropped neighbourhood of " + std::to_string(hs.max_k_over) +
                           " points does not fit the device memory left for the sweep scratch");
      c->sweep_gcap = (int)need;
      c->sweep_g2 = (int)g2;
      continue;
    }
    if (hs.err_flags & 2u) {  // list arena of the split sweep too small: grow to what this run asked for, retry
      c->list_ints = std::max<size_t>((size_t)hs.list_top + ((size_t)hs.list_top >> 3), c->list_ints * 2);
      continue;
    }
    if (hs.err_flags & 1u) {  // arena too small: grow to what this run asked for, retry
      c->arena_points = std::max<size_t>((size_t)hs.arena_top + ((size_t)hs.arena_top >> 3), c->arena_points * 2);
      continue;
    }
    note_sweep(c, s, hs, compact_mode);
    return 0;
  }
  return set_err(c, AG2_ERR_CAPACITY, "point-list arena could not be sized");
}

// ag2_detect with ONE host round trip.  The step-by-step form reads the sweep's statistics in the
// middle (it needs the number of images to size the renderer's and LeNet's launches) and sorts the
// selected records on the host.  From the second call of a context on, the tail is launched at shapes
// learned from the previous call -- capacity for a quarter more images, the renderers the largest
// in-box list so far needs -- with the list length read on the device (the kernels of frame mode),
// the top num_selected are picked on the device (k_topk), and statistics + records come back in one
// copy.  The statistics then say whether the shapes held (and whether a buffer was too small):
// if not, kSpecRedo sends the caller through the step-by-step form, which also learns the new shapes.
// Results are byte for byte those of the step-by-step form (tests/test_gpu_lenet_detect.py).
constexpr int kSpecRedo = 1;
int detect_speculative(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                       uint64_t slot_base, uint64_t seed, int do_prune, ag2_hypothesis* selected, size_t cap,
                       size_t* n_selected, size_t* n_scored, bool rank_mode) {
  const size_t n_slots = s * (size_t)c->p.num_orientations;
  const size_t cap_img = std::min(c->spec_cap_img, n_slots);
  const size_t k_cap = (c->p.num_selected >= 0) ? std::min<size_t>((size_t)c->p.num_selected, cap_img) : cap_img;
  if (!rank_mode && k_cap > cap) return kSpecRedo;  // (the step-by-step form reports the caller's short buffer)
  if (rank_mode && !c->h_pin_dev) return kSpecRedo;
  c->sweep_may_skip_stage1 = true;
  int rc = run_hypotheses(c, sample_idx, sample_xyz, s, slot_base, seed, true, do_prune ? 1 : 0, /*defer_read=*/true);
  c->sweep_may_skip_stage1 = false;
  if (rc) return rc;
  if (c->desc_stride == 0) return kSpecRedo;  // (no descriptors from the compaction: more than 64 Ki slots)
  DevStats* st = c->d_stats.as<DevStats>();
  const unsigned* d_n = &st->n_list;
  AG2_HIP(c, c->d_images.reserve(cap_img * 10800));
  AG2_HIP(c, c->d_logits.reserve(cap_img * 8));
  AG2_HIP(c, stage_event(c, 3));
  rc = launch_render(c, c->d_arena.as<double>(), c->d_desc.as<long long>(),
                     (const int*)(c->d_desc.as<long long>() + c->desc_stride), cap_img,
                     c->d_images.as<uint8_t>(), c->spec_max_p, d_n);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 4));
  rc = launch_lenet(c, c->d_images.as<uint8_t>(), cap_img, c->d_logits.as<float>(), 5, d_n);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 6));
  // the top-k kernel writes statistics + records where the host reads them (page-locked memory through
  // its device view): no copy operation behind it
  const size_t out_bytes = sizeof(FrameOut) + k_cap * sizeof(ag2_hypothesis);
  rc = pin_reserve(c, out_bytes);
  if (rc) return rc;
  if (!c->h_pin_dev) return kSpecRedo;
  FrameOut* d_fo = reinterpret_cast<FrameOut*>(pin_bulk_dev(c));
  ag2_hypothesis* d_rec = reinterpret_cast<ag2_hypothesis*>(d_fo + 1);
  rc = score_and_select_async(c, c->d_list2.as<int>(), cap_img, &st->n_sel, d_n);
  if (rc) return rc;
  c->d_last_sel = c->d_sel.p;  // for ag2_export_selected_compact_device
  c->d_last_nsel = &st->n_sel;
  if (rank_mode) {
    // A rank that only feeds the merge: no clustering, no top-k, no read-back -- and NO wait.  Whether the shapes
    // held is decided on the device when the list is exported (k_export_selected), read by every rank after the
    // exchange (ag2_merge_*: AG2_ERR_RETRY); the statistics are taken up there as well (rank_spec_collect).
    AG2_HIP(c, stage_event(c, 7));
    c->rank_spec.pending = true;
    c->rank_spec.cap_img = (unsigned)cap_img;
    c->rank_spec.render_cap = render_capacity_for(c->spec_max_p);
    c->rank_spec.stage1_skipped = c->sweep_stage1_skipped ? 1 : 0;
    c->rank_spec.s = s;
    *n_selected = 0;
    if (n_scored) *n_scored = 0;  // (not known yet: ag2_get_counters after the merge / gather)
    return 0;
  }
  const ag2_hypothesis* d_res = c->d_sel.as<ag2_hypothesis>();
  const unsigned* d_nres = &st->n_sel;
  if (c->min_inliers > 0) {  // grasp clusters between the threshold and the top-k (grasp_detector.cpp:228-236)
    rc = cluster_async(c, d_res, cap_img, d_nres, c->min_inliers, &st->n_clu);
    if (rc) return rc;
    d_res = c->d_cluster.as<ag2_hypothesis>();
    d_nres = &st->n_clu;
  }
  rc = launch_topk(c, d_res, d_nres, cap_img, k_cap, d_rec, d_fo, nullptr);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 7));
  if (c->stage_timing >= 2) {  // (event 7 is behind k_topk: wait for the stream)
    AG2_HIP(c, hipStreamSynchronize(c->stream));
  } else {
    rc = wait_topk(c);
    if (rc) return rc;
    if (c->stage_timing == 1) AG2_HIP(c, stage_sync(c, 11));  // (long complete; the runtime takes note of it)
  }
  FrameOut fo;
  memcpy(&fo, pin_bulk(c), sizeof(FrameOut));
  const DevStats& hs = fo.st;
  // did the shapes hold?  (err_flags: a buffer of the sweep was too small; the step-by-step form grows it)
  if ((hs.err_flags & (1u | 2u | 8u)) || (size_t)hs.n_list > cap_img ||
      (int)hs.max_p > render_capacity_for(c->spec_max_p) || fo.topk_overflow ||
      (c->sweep_stage1_skipped && hs.n_overflow > 0)) {  // (samples were queued for the stage that was left out)
    c->sweep_no_overflow_runs = 0;
    return kSpecRedo;
  }
  note_sweep(c, s, hs, 1);
  const size_t n_img = hs.n_list;
  c->cnt.n_pruned = (int64_t)n_img;
  c->cnt.n_scored = (int64_t)n_img;
  c->cnt.n_selected = (int64_t)fo.n_out;
  *n_selected = fo.n_out;
  if (n_scored) *n_scored = n_img;
  stage_elapsed(c, &c->times.compact_ms, 11, 3);
  stage_elapsed(c, &c->times.render_ms, 3, 4);
  stage_elapsed(c, &c->times.lenet_conv_ms, 4, 5);
  stage_elapsed(c, &c->times.lenet_fc_ms, 5, 6);
  stage_elapsed(c, &c->times.select_ms, 6, 7);
  stage_elapsed(c, &c->times.total_ms, 8, 7);
  if (fo.n_out) memcpy(selected, pin_bulk(c) + sizeof(FrameOut), (size_t)fo.n_out * sizeof(ag2_hypothesis));
  // the shapes follow the workload (never below what just ran)
  c->spec_cap_img = std::max(c->spec_cap_img, ((n_img + n_img / 4 + 256 + 255) / 256) * 256);
  c->spec_max_p = std::max(c->spec_max_p, (int)hs.max_p);
  c->spec_runs++;
  return 0;
}

}  // namespace

namespace ag2 {
// Takes up what a rank's one-trip detect left: the statistics k_export_selected copied into page-locked memory
// (counters, stage times, the shapes the next call is launched at).  Called where the stream has been waited
// for anyway (merge, gather); elsewhere (stream_is_idle == false) it waits itself.
int rank_spec_collect(ag2_ctx* c, bool stream_is_idle) {
  if (!c->rank_spec.pending) return 0;
  if (!stream_is_idle) AG2_HIP(c, hipStreamSynchronize(c->stream));
  c->rank_spec.pending = false;
  DevStats hs;
  memcpy(&hs, pin_small(c) + kPinRankStats, sizeof(hs));
  const bool bad = (hs.err_flags & (1u | 2u | 8u)) != 0u || hs.n_list > c->rank_spec.cap_img ||
                   (int)hs.max_p > c->rank_spec.render_cap || (c->rank_spec.stage1_skipped && hs.n_overflow > 0u);
  if (bad) {
    c->spec_fallbacks++;
    c->sweep_no_overflow_runs = 0;
    c->spec_cap_img = 0;  // the next call runs step by step and learns the shapes again
    return 0;
  }
  note_sweep(c, c->rank_spec.s, hs, 1);
  const size_t n_img = hs.n_list;
  c->cnt.n_pruned = (int64_t)n_img;
  c->cnt.n_scored = (int64_t)n_img;
  c->cnt.n_selected = 0;
  stage_elapsed(c, &c->times.compact_ms, 11, 3);
  stage_elapsed(c, &c->times.render_ms, 3, 4);
  stage_elapsed(c, &c->times.lenet_conv_ms, 4, 5);
  stage_elapsed(c, &c->times.lenet_fc_ms, 5, 6);
  stage_elapsed(c, &c->times.select_ms, 6, 7);
  stage_elapsed(c, &c->times.total_ms, 8, 7);
  c->spec_cap_img = std::max(c->spec_cap_img, ((n_img + n_img / 4 + 256 + 255) / 256) * 256);
  c->spec_max_p = std::max(c->spec_max_p, (int)hs.max_p);
  c->spec_runs++;
  return 0;
}
}  // namespace ag2

extern "C" {

int ag2_local_frames(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                     uint64_t slot_base, uint64_t seed, double* frames, int32_t* valid) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = check_samples(c, sample_idx, sample_xyz, s);
  if (rc) return rc;
  rc = reset_stats(c);
  if (rc) return rc;
  rc = upload_samples(c, sample_idx, sample_xyz, s);
  if (rc) return rc;
  rc = launch_frames(c, s, slot_base, seed);
  if (rc) return rc;
  if (s) {
    AG2_HIP(c, hipMemcpyAsync(frames, c->d_frames.p, s * 96, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipMemcpyAsync(valid, c->d_frame_ok.p, s * 4, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < s; i++)
      if (!valid[i])
        for (int k = 0; k < 12; k++) frames[12 * i + k] = 0.0;
  }
  return 0;
}

int ag2_generate_hypotheses(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz,
                            size_t s, uint64_t slot_base, uint64_t seed, ag2_hypothesis* out,
                            size_t cap, size_t* n_out) {
  if (!c || !n_out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = check_samples(c, sample_idx, sample_xyz, s);
  if (rc) return rc;
  rc = run_hypotheses(c, sample_idx, sample_xyz, s, slot_base, seed, true);
  if (rc) return rc;
  size_t nh = 0;
  rc = compact_slots(c, s * (size_t)c->p.num_orientations, 0, c->d_list, &nh);
  if (rc) return rc;
  std::vector<uint8_t> keep;
  rc = gather_records(c, c->d_list.as<int>(), nh, c->h_hyps, &c->h_offsets, &keep);
  if (rc) return rc;
  c->h_keep = keep;
  c->h_slots.resize(nh);
  if (nh) {
    AG2_HIP(c, hipMemcpy(c->h_slots.data(), c->d_list.p, nh * 4, hipMemcpyDeviceToHost));
  }
  *n_out = nh;
  if (nh > cap) return set_err(c, AG2_ERR_CAPACITY, "generate_hypotheses: output capacity too small");
  if (nh) memcpy(out, c->h_hyps.data(), nh * sizeof(ag2_hypothesis));
  return 0;
}

int ag2_hyp_points(ag2_ctx* c, size_t h, double* pts, double* nrm) {
  if (!c || !pts || !nrm) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (h >= c->h_hyps.size()) return set_err(c, AG2_ERR_ARG, "hypothesis index out of range");
  const int P = c->h_hyps[h].n_points;
  const int64_t off = c->h_offsets[h];
  if (off < 0) return set_err(c, AG2_ERR_STATE, "no point list stored for this hypothesis");
  std::vector<double> buf((size_t)P * 6);
  AG2_HIP(c, hipMemcpy(buf.data(), c->d_arena.as<double>() + (size_t)off * 6, (size_t)P * 48,
                       hipMemcpyDeviceToHost));
  for (int b = 0; b < P; b++)
    for (int k = 0; k < 3; k++) {
      pts[3 * b + k] = buf[6 * (size_t)b + k];
      nrm[3 * b + k] = buf[6 * (size_t)b + 3 + k];
    }
  return 0;
}

int ag2_prune(ag2_ctx* c, uint8_t* keep, size_t n) {
  if (!c || !keep) return AG2_ERR_ARG;
  if (n != c->h_hyps.size()) return set_err(c, AG2_ERR_ARG, "prune: size mismatch");
  for (size_t i = 0; i < n; i++) keep[i] = c->h_keep[i] ? 1 : 0;
  return 0;
}

int ag2_render_images(ag2_ctx* c, size_t first, size_t count, uint8_t* out) {
  if (!c || !out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (first + count > c->h_hyps.size()) return set_err(c, AG2_ERR_ARG, "render: range");
  if (count == 0) return 0;
  AG2_HIP(c, stage_event(c, 3));
  int rc = make_image_descs(c, c->d_list.as<int>() + first, count);
  if (rc) return rc;
  AG2_HIP(c, c->d_images.reserve(count * 10800));
  rc = launch_render(c, c->d_arena.as<double>(), c->d_desc.as<long long>(),
                     (const int*)(c->d_desc.as<long long>() + count), count,
                     c->d_images.as<uint8_t>(), c->max_p);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 4));
  AG2_HIP(c, hipMemcpyAsync(out, c->d_images.p, count * 10800, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  stage_elapsed(c, &c->times.render_ms, 3, 4);
  return 0;
}

int ag2_render_images_from_points(ag2_ctx* c, size_t n, const int64_t* offsets, const double* pts,
                                  const double* nrm, uint8_t* out) {
  if (!c || !offsets || !out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (n == 0) return 0;
  const size_t tot = (size_t)offsets[n];
  std::vector<double> inter(std::max<size_t>(tot, 1) * 6);
  for (size_t b = 0; b < tot; b++)
    for (int k = 0; k < 3; k++) {
      inter[6 * b + k] = pts[3 * b + k];
      inter[6 * b + 3 + k] = nrm[3 * b + k];
    }
  std::vector<long long> off(n);
  std::vector<int> cnt(n);
  for (size_t i = 0; i < n; i++) {
    off[i] = offsets[i];
    cnt[i] = (int)(offsets[i + 1] - offsets[i]);
  }
  AG2_HIP(c, c->d_tmp.reserve(inter.size() * 8));
  AG2_HIP(c, c->d_desc.reserve(n * 12));
  AG2_HIP(c, c->d_images.reserve(n * 10800));
  long long* d_off = c->d_desc.as<long long>();
  int* d_cnt = (int*)(d_off + n);
  AG2_HIP(c, hipMemcpyAsync(c->d_tmp.p, inter.data(), inter.size() * 8, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipMemcpyAsync(d_off, off.data(), n * 8, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipMemcpyAsync(d_cnt, cnt.data(), n * 4, hipMemcpyHostToDevice, c->stream));
  int max_p = 0;
  for (size_t i = 0; i < n; i++) max_p = std::max(max_p, cnt[i]);
  const int rc = launch_render(c, c->d_tmp.as<double>(), d_off, d_cnt, n, c->d_images.as<uint8_t>(), max_p);
  if (rc) return rc;
  AG2_HIP(c, hipMemcpyAsync(out, c->d_images.p, n * 10800, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"

extern "C" {

int ag2_lenet_load(ag2_ctx* c, const float* c1w, const float* c1b, const float* c2w,
                   const float* c2b, const float* f1w, const float* f1b, const float* f2w,
                   const float* f2b) {
  if (!c || !c1w || !c1b || !c2w || !c2b || !f1w || !f1b || !f2w || !f2b) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  return lenet_pack_weights(c, c1w, c1b, c2w, c2b, f1w, f1b, f2w, f2b);
}

int ag2_lenet_forward(ag2_ctx* c, const uint8_t* images, size_t n, float* out) {
  if (!c || (n && (!images || !out))) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->net.loaded) return set_err(c, AG2_ERR_STATE, "lenet weights not loaded");
  if (n == 0) return 0;
  AG2_HIP(c, c->d_images.reserve(n * 10800));
  AG2_HIP(c, c->d_logits.reserve(n * 8));
  AG2_HIP(c, hipMemcpyAsync(c->d_images.p, images, n * 10800, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, stage_event(c, 4));
  const int rc = launch_lenet(c, c->d_images.as<uint8_t>(), n, c->d_logits.as<float>(), 5);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 6));
  AG2_HIP(c, hipMemcpyAsync(out, c->d_logits.p, n * 8, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  stage_elapsed(c, &c->times.lenet_conv_ms, 4, 5);
  stage_elapsed(c, &c->times.lenet_fc_ms, 5, 6);
  return 0;
}

int ag2_detect(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
               uint64_t slot_base, uint64_t seed, int do_prune, ag2_hypothesis* selected,
               size_t cap, size_t* n_selected, ag2_hypothesis* scored_all, size_t cap_all,
               size_t* n_scored) {
  if (!c || !n_selected) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = check_samples(c, sample_idx, sample_xyz, s);
  if (rc) return rc;
  if (!c->net.loaded) return set_err(c, AG2_ERR_STATE, "lenet weights not loaded");
  const size_t n_slots = s * (size_t)c->p.num_orientations;
  (void)n_slots;
  AG2_HIP(c, stage_event(c, 8));
  // The one-round-trip form, when the previous call on this context left its shapes and nothing asks
  // for the step-by-step one (all scored records, the multi-GPU export without read-back,
  // the f32-input LeNet kernels, AG2_DETECT_STEPWISE=1 for A/B).
  static const bool spec_off = getenv("AG2_DETECT_STEPWISE") != nullptr;
  rc = rank_spec_collect(c, /*stream_is_idle=*/false);  // (a rank's previous call nobody merged or gathered)
  if (rc) return rc;
  // (selected == NULL and cap == 0: a rank of a multi-GPU job -- the caller exports the list and the merge selects)
  const bool rank_mode = !selected && cap == 0 && !(scored_all && cap_all);
  const bool spec = !spec_off && (selected || rank_mode) && !(scored_all && cap_all) && c->net.use_x3 &&
                    !c->fm_on && c->spec_cap_img > 0 && c->spec_s == s && c->spec_prune == (do_prune ? 1 : 0) &&
                    n_slots > 0 && n_slots <= 65536;
  if (spec) {
    rc = detect_speculative(c, sample_idx, sample_xyz, s, slot_base, seed, do_prune, selected, cap, n_selected,
                            n_scored, rank_mode);
    if (rc != kSpecRedo) return rc;
    c->spec_fallbacks++;
    AG2_HIP(c, stage_event(c, 8));
  }
  // 1. hypotheses + 2. prune (predicate evaluated in the sweep; the survivor list is compacted on
  // the device and its length comes back with the sweep's statistics: one host round trip)
  rc = run_hypotheses(c, sample_idx, sample_xyz, s, slot_base, seed, true, do_prune ? 1 : 0);
  if (rc) return rc;
  const size_t n_img = c->n_img;
  // shapes for the next call's one-round-trip form
  c->spec_cap_img = ((n_img + n_img / 4 + 256 + 255) / 256) * 256;
  c->spec_max_p = c->max_p;
  c->spec_s = s;
  c->spec_prune = do_prune ? 1 : 0;
  c->cnt.n_pruned = (int64_t)n_img;
  AG2_HIP(c, c->d_images.reserve(std::max<size_t>(n_img, 1) * 10800));
  AG2_HIP(c, c->d_logits.reserve(std::max<size_t>(n_img, 1) * 8));
  size_t desc_stride = c->desc_stride;  // the compaction usually wrote the descriptors already
  if (desc_stride == 0) {
    rc = make_image_descs(c, c->d_list2.as<int>(), n_img);                     // 3a. images
    if (rc) return rc;
    desc_stride = n_img;
  }
  AG2_HIP(c, stage_event(c, 3));
  rc = launch_render(c, c->d_arena.as<double>(), c->d_desc.as<long long>(),
                     (const int*)(c->d_desc.as<long long>() + desc_stride), n_img,
                     c->d_images.as<uint8_t>(), c->max_p);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 4));
  rc = launch_lenet(c, c->d_images.as<uint8_t>(), n_img, c->d_logits.as<float>(), 5);  // 3b.
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 6));
  // 4. score = ip2[1] - ip2[0], keep score >= min_score_diff, gather in order -- all on the device;
  // one read-back brings the count and (for the usual small lists) the records themselves
  unsigned* d_nsel = &c->d_stats.as<DevStats>()->n_sel;
  rc = score_and_select_async(c, c->d_list2.as<int>(), n_img, d_nsel);
  if (rc) return rc;
  // 4b. grasp clusters (grasp_detector.cpp:228-236), only when HandleSearch::setMinInliers > 0
  const void* d_res = c->d_sel.p;
  const unsigned* d_nres = d_nsel;
  // What ag2_export_selected_compact_device puts on the wire is the list BEFORE the clustering: clusters
  // count inliers over the hands of all ranks, so a multi-GPU job clusters in the merge
  // (ag2_merge_selected_device), on the gathered list -- and a rank that only feeds the merge skips its own.
  c->d_last_sel = c->d_sel.p;
  c->d_last_nsel = d_nsel;
  const bool merge_follows = !selected && cap == 0 && !(scored_all && cap_all);
  if (c->min_inliers > 0 && !merge_follows) {
    unsigned* d_nclu = &c->d_stats.as<DevStats>()->n_clu;
    rc = cluster_async(c, c->d_sel.as<ag2_hypothesis>(), n_img, d_nsel, c->min_inliers, d_nclu);
    if (rc) return rc;
    d_res = c->d_cluster.p;
    d_nres = d_nclu;
  }
  AG2_HIP(c, stage_event(c, 7));
  if (merge_follows) {
    // Multi-GPU use: the caller exports the selected list (ag2_export_selected_compact_device) and the
    // top-k happens in ag2_merge_selected_device over all ranks' lists -- no local read-back, no local
    // top-k, and this call returns with the tail of the pipeline still queued on the stream.
    *n_selected = 0;
    c->cnt.n_scored = (int64_t)n_img;
    c->cnt.n_selected = 0;
    if (n_scored) *n_scored = n_img;
    return 0;
  }
  std::vector<ag2_hypothesis> anti;
  const ag2_hypothesis* recs = nullptr;  // the selected records, in list order
  unsigned n_anti = 0;
  {
    // The records are followed by their count (trailer written by the gather / cluster kernel), so
    // for the usual small lists ONE copy into page-locked memory brings both.
    const size_t rec_bytes = n_img * sizeof(ag2_hypothesis);
    const bool small = rec_bytes <= ((size_t)2 << 20);
    if (small && n_img) {
      rc = pin_reserve(c, rec_bytes + 16);
      if (rc) return rc;
      AG2_HIP(c, hipMemcpyAsync(pin_bulk(c), d_res, rec_bytes + 4, hipMemcpyDeviceToHost, c->stream));
      AG2_HIP(c, hipStreamSynchronize(c->stream));
      memcpy(&n_anti, pin_bulk(c) + rec_bytes, 4);
      recs = (const ag2_hypothesis*)pin_bulk(c);  // (nothing below writes the staging area)
    } else {
      AG2_HIP(c, hipMemcpyAsync(pin_small(c), d_nres, 4, hipMemcpyDeviceToHost, c->stream));
      AG2_HIP(c, hipStreamSynchronize(c->stream));
      memcpy(&n_anti, pin_small(c), 4);
      anti.resize(n_anti);
      if (n_anti)
        AG2_HIP(c, hipMemcpy(anti.data(), d_res, (size_t)n_anti * sizeof(ag2_hypothesis),
                             hipMemcpyDeviceToHost));
      recs = anti.data();
    }
  }
  // 5. top num_selected by score, descending (grasp_detector.cpp:239-252); ties by position.  Only
  // the order of the first k is needed: indices are (partially) sorted, the 176-byte records are
  // copied once.
  size_t k = n_anti;
  if (c->p.num_selected >= 0 && k > (size_t)c->p.num_selected) k = (size_t)c->p.num_selected;
  std::vector<uint32_t> order(n_anti);
  for (uint32_t i = 0; i < n_anti; i++) order[i] = i;
  const auto better = [recs](uint32_t a, uint32_t b) {
    return recs[a].score > recs[b].score || (recs[a].score == recs[b].score && a < b);
  };
  if (k < n_anti) std::partial_sort(order.begin(), order.begin() + k, order.end(), better);
  else std::sort(order.begin(), order.end(), better);
  *n_selected = k;
  c->cnt.n_scored = (int64_t)n_img;
  c->cnt.n_selected = (int64_t)k;
  if (n_scored) *n_scored = n_img;
  if (scored_all && cap_all) {
    if (n_img > cap_all) return set_err(c, AG2_ERR_CAPACITY, "detect: scored_all capacity too small");
    std::vector<ag2_hypothesis> all;
    rc = gather_records(c, c->d_list2.as<int>(), n_img, all, nullptr, nullptr);
    if (rc) return rc;
    if (n_img) memcpy(scored_all, all.data(), n_img * sizeof(ag2_hypothesis));
  }
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  stage_elapsed(c, &c->times.compact_ms, 11, 3);
  stage_elapsed(c, &c->times.render_ms, 3, 4);
  stage_elapsed(c, &c->times.lenet_conv_ms, 4, 5);
  stage_elapsed(c, &c->times.lenet_fc_ms, 5, 6);
  stage_elapsed(c, &c->times.select_ms, 6, 7);
  stage_elapsed(c, &c->times.total_ms, 8, 7);
  if (k > cap) return set_err(c, AG2_ERR_CAPACITY, "detect: output capacity too small");
  for (size_t i = 0; i < k; i++) selected[i] = recs[order[i]];
  return 0;
}

int ag2_export_candidates_device(ag2_ctx* c, void* d_dst, size_t bytes) {
  if (!c || !d_dst) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  const size_t need = c->s * (size_t)c->p.num_orientations * sizeof(ag2_hypothesis);
  if (bytes < need) return set_err(c, AG2_ERR_CAPACITY, "export: destination too small");
  if (need) {
    const int n16 = (int)(need / 16);
    hipLaunchKernelGGL(k_export_table, dim3((n16 + 255) / 256), dim3(256), 0, c->stream,
                       c->d_table.as<uint4>(), c->d_tab_keep.as<unsigned char>(), n16, (uint4*)d_dst);
    AG2_HIP(c, hipGetLastError());
  }
  return 0;
}

int ag2_export_selected_compact_device(ag2_ctx* c, void* d_dst, size_t bytes, size_t cap_records) {
  if (!c || !d_dst) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (cap_records > ((size_t)1 << 30)) return set_err(c, AG2_ERR_ARG, "export: cap too large");
  if (bytes < 16 + cap_records * sizeof(ag2_hypothesis))
    return set_err(c, AG2_ERR_CAPACITY, "export: destination too small");
  if (!c->d_last_nsel) return set_err(c, AG2_ERR_STATE, "export: no ag2_detect result to export");
  if (!c->d_last_sel || cap_records == 0) {  // a detect over no images: an empty list
    const unsigned hdr[4] = {0u, (unsigned)cap_records, 0u, 0u};
    AG2_HIP(c, hipMemcpyAsync(d_dst, hdr, 16, hipMemcpyHostToDevice, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
  }
  return export_selected_compact(c, d_dst, cap_records);
}

int ag2_merge_selected_device(ag2_ctx* c, const void* d_gathered, size_t world, size_t cap_records,
                              ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_total) {
  if (!c || !d_gathered || !n_selected || world == 0) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (world > 4096 || cap_records > ((size_t)1 << 24)) return set_err(c, AG2_ERR_ARG, "merge: sizes");
  return merge_selected(c, d_gathered, world, cap_records, selected, cap, n_selected, n_total);
}

int ag2_gather_begin(ag2_ctx* root, size_t world, size_t cap_records) {
  if (!root || world == 0) return AG2_ERR_ARG;
  (void)hipSetDevice(root->device);
  if (world > 4096 || cap_records > ((size_t)1 << 24)) return set_err(root, AG2_ERR_ARG, "gather: sizes");
  const size_t per = 16 + cap_records * sizeof(ag2_hypothesis);
  AG2_HIP(root, hipStreamSynchronize(root->stream));  // (a merge of the previous exchange may still read the buffer)
  AG2_HIP(root, root->d_gather.reserve(world * per));
  // No rank has delivered yet, and the merge refuses to run while one is missing: a rank that failed can no longer
  // leave the previous exchange's records in its place.  (Clearing the headers as well would cost a
  // fill and a wait per step; the delivery marks make stale headers unreachable.)
  root->gather_delivered.assign(world, 0);
  root->gather_world = world;
  root->gather_cap = cap_records;
  return 0;
}

int ag2_gather_selected(ag2_ctx* root, ag2_ctx* src, size_t rank) {
  if (!root || !src) return AG2_ERR_ARG;
  if (rank >= root->gather_world) return set_err(src, AG2_ERR_ARG, "gather: rank outside the world given to ag2_gather_begin");
  const size_t cap = root->gather_cap, per = 16 + cap * sizeof(ag2_hypothesis);
  (void)hipSetDevice(src->device);
  AG2_HIP(src, src->d_xchg.reserve(per));
  const int rc = ag2_export_selected_compact_device(src, src->d_xchg.p, per, cap);
  if (rc) return rc;
  char* dst = (char*)root->d_gather.p + rank * per;
  if (src->device == root->device)
    AG2_HIP(src, hipMemcpyAsync(dst, src->d_xchg.p, per, hipMemcpyDeviceToDevice, src->stream));
  else  // over xGMI; the runtime stages through the host when the devices have no peer access
    AG2_HIP(src, hipMemcpyPeerAsync(dst, root->device, src->d_xchg.p, src->device, per, src->stream));
  AG2_HIP(src, hipStreamSynchronize(src->stream));
  {  // (the stream is idle: what a one-trip rank detect left is taken up here)
    const int rcc = rank_spec_collect(src, /*stream_is_idle=*/true);
    if (rcc) return rcc;
  }
  root->gather_delivered[rank] = 1;  // (one byte per rank: the ranks' threads write distinct elements)
  return 0;
}

int ag2_merge_gathered(ag2_ctx* root, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_total) {
  if (!root || !n_selected) return AG2_ERR_ARG;
  (void)hipSetDevice(root->device);
  if (root->gather_world == 0) return set_err(root, AG2_ERR_STATE, "merge: no ag2_gather_begin");
  for (size_t r = 0; r < root->gather_world; r++)
    if (r >= root->gather_delivered.size() || !root->gather_delivered[r])
      return set_err(root, AG2_ERR_STATE, "merge: rank " + std::to_string(r) + " has not delivered its list (ag2_gather_selected)");
  return merge_selected(root, root->d_gather.p, root->gather_world, root->gather_cap, selected, cap, n_selected, n_total);
}

int ag2_export_candidates_compact_device(ag2_ctx* c, void* d_dst, size_t bytes, size_t cap_records) {
  if (!c || !d_dst) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (cap_records > ((size_t)1 << 30)) return set_err(c, AG2_ERR_ARG, "export: cap too large");
  if (bytes < 16 + cap_records * sizeof(ag2_hypothesis))
    return set_err(c, AG2_ERR_CAPACITY, "export: destination too small");
  return export_candidates_compact(c, d_dst, cap_records);
}

}  // extern "C"
