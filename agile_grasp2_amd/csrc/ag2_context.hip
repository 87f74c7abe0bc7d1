// ag2_context.hip -- context, memory management and the C-ABI entry points of libag2hip.so.
// See include/ag2_c.h for the contract and the reference code each entry point replaces.
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ag2_internal.h"

namespace ag2 {

int set_err(ag2_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

// Host-side derivation of the constants the kernels read (computed once, in double, with libm).
static void derive_constants(const ag2_params& p, HandConst& h, double* angles = nullptr) {
  memset(&h, 0, sizeof(h));
  // FingerHand ctor, finger_hand.cpp:7-12 (Eigen LinSpaced(i) = low + i * ((high-low)/(n-1)))
  const int n = 10;
  const double high = p.hand_outer_diameter - p.finger_width;
  const double step = (high - 0.0) / (double)(n - 1);
  for (int i = 0; i < n; i++) {
    const double fh = 0.0 + (double)i * step;
    h.fs[i] = (fh - p.hand_outer_diameter) + p.finger_width;
    h.fs[n + i] = fh;
  }
  for (int i = 0; i < 2 * n; i++) h.fsr[i] = h.fs[i] + p.finger_width;
  h.slot_step = step;
  h.slot_inv_step = (step > 0.0) ? 1.0 / step : 0.0;
  // 2 <=> finger narrower than two spacings less a hair: three exact candidates suffice
  h.slot_span = (step > 0.0 && p.finger_width / step < 2.0 - 1e-6) ? 2 : 20;
  // hand_search.cpp:179-180
  const int R = p.num_orientations;
  const double low = -1.0 * M_PI / 2.0, hi = M_PI / 2.0;
  const double astep = (hi - low) / (double)R;
  for (int i = 0; i < R; i++) {
    const double a = low + (double)i * astep;
    if (angles) angles[i] = a;
    h.cos_t[i] = cos(a);
    h.sin_t[i] = sin(a);
  }
  // finger_hand.cpp:118-122
  int nd = 0;
  for (double d = p.init_bite + 0.005; d <= p.hand_depth && nd < kMaxDepths; d += 0.005)
    h.depths[nd++] = d;
  h.n_depths = nd;
  for (int i = 0; i < 2; i++)
    for (int k = 0; k < 3; k++) h.cam_origin[i][k] = p.cam_origin[i][k];
  h.finger_width = p.finger_width;
  h.hand_outer_diameter = p.hand_outer_diameter;
  h.hand_depth = p.hand_depth;
  h.hand_height = p.hand_height;
  h.init_bite = p.init_bite;
  h.cos_fc = cos(30.0 * M_PI / 180.0);
  h.min_aperture = p.min_aperture;
  h.max_aperture = p.max_aperture;
  h.ws_min_x = (float)p.workspace[0];
  h.ws_max_x = (float)p.workspace[1];
  h.ws_min_y = (float)p.workspace[2];
  h.ws_max_y = (float)p.workspace[3];
  h.r2_taubin = (float)(p.nn_radius_taubin * p.nn_radius_taubin);
  h.r2_hands = (float)(p.nn_radius_hands * p.nn_radius_hands);
  h.r2_normals = (float)(p.normals_radius * p.normals_radius);
  h.rq_taubin = (float)p.nn_radius_taubin * 1.001f;
  h.rq_hands = (float)p.nn_radius_hands * 1.001f;
  h.rq_normals = (float)p.normals_radius * 1.001f;
  h.R = R;
  h.n_cams = p.n_cams;
  h.filter_half = p.filter_half_grasps;
}

static int upload_constants(ag2_ctx* c) {
  AG2_HIP(c, c->d_hc.reserve(sizeof(HandConst)));
  AG2_HIP(c, hipMemcpyAsync(c->d_hc.p, &c->hc, sizeof(HandConst), hipMemcpyHostToDevice, c->stream));
  return 0;
}

static int check_params(ag2_ctx* c) {
  const ag2_params& p = c->p;
  if (p.num_orientations < 1 || p.num_orientations > kMaxOrient)
    return set_err(c, AG2_ERR_ARG, "num_orientations must be in [1, 32]");
  if (p.n_cams < 1 || p.n_cams > 2) return set_err(c, AG2_ERR_ARG, "n_cams must be 1 or 2");
  if (!(p.grid_cell > 0) || !(p.nn_radius_hands > 0) || !(p.nn_radius_taubin > 0) ||
      !(p.normals_radius > 0))
    return set_err(c, AG2_ERR_ARG, "radii and grid_cell must be positive");
  // cells spanned per axis by [q - r', q + r'] with r' = 1.001 r: at most floor(2 r' / cell) + 2
  const double span = floor(2.0 * p.nn_radius_hands * 1.002 / p.grid_cell) + 2.0;
  if (span * span > (double)kMaxRows)
    return set_err(c, AG2_ERR_ARG,
                   "nn_radius_hands / grid_cell too large for the 512-row stencil table "
                   "(need (floor(2r/cell)+2)^2 <= 512)");
  if (!(p.hand_depth > 0) || !(p.hand_height > 0))
    return set_err(c, AG2_ERR_ARG, "hand_depth and hand_height must be positive");
  return 0;
}

// k_normals is launched without a host sync; its duration and neighbour counter are read here.
int collect_normals_stats(ag2_ctx* c) {
  if (c->grid_pending) {
    AG2_HIP(c, stage_sync(c, 13));
    stage_elapsed(c, &c->times.grid_ms, 12, 13);
    c->grid_pending = false;
  }
  if (!c->normals_pending) return 0;
  if (stage_event_on(c, 10)) AG2_HIP(c, hipEventSynchronize(c->ev[10]));
  else AG2_HIP(c, hipStreamSynchronize(c->stream));  // k_normals must be done before its counter is read
  stage_elapsed(c, &c->times.normals_ms, 9, 10);
  unsigned long long k1 = 0;
  AG2_HIP(c, hipMemcpy(&k1, (const char*)c->d_stats.p + offsetof(DevStats, sum_k1), 8, hipMemcpyDeviceToHost));
  c->cnt.sum_k1 = (int64_t)k1;
  c->normals_pending = false;
  return 0;
}

int pin_reserve(ag2_ctx* c, size_t bulk_bytes) {
  const size_t need = kPinSmall + bulk_bytes;
  if (need <= c->h_pin_bytes) return 0;
  // (unconditional: a NULL handle is the HIP default stream, a supported setting -- ag2_set_stream)
  AG2_HIP(c, hipStreamSynchronize(c->stream));  // nothing may still be copying
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  c->h_pin = nullptr;
  c->h_pin_dev = nullptr;
  c->h_pin_bytes = 0;
  const size_t want = need + need / 2 + 65536;
  // explicitly coherent (fine-grained, uncached on the device side) and mapped: kernels write results and the
  // flags the host polls into it, which is only sound while no device cache holds those lines back --
  // hipHostMallocDefault leaves that to HIP_HOST_COHERENT
  AG2_HIP(c, hipHostMalloc(&c->h_pin, want, kPinFlags));
  memset(c->h_pin, 0, kPinSmall);  // (the done flag of k_topk starts below every sequence number)
  c->h_pin_bytes = want;
  c->h_pin_dev = nullptr;
  AG2_HIP(c, hipHostGetDevicePointer(&c->h_pin_dev, c->h_pin, 0));
  return 0;
}

// d_xyz_in / n hold a new cloud: drop everything derived from the previous one, build the grid.
int after_cloud(ag2_ctx* c) {
  memset(&c->cnt, 0, sizeof(c->cnt));
  c->normals_pending = false;
  c->grid_pending = false;
  c->h_hyps.clear();
  c->h_slots.clear();
  c->h_offsets.clear();
  c->s = 0;
  c->n_img = 0;
  c->n_resident_samples = 0;
  AG2_HIP(c, stage_event(c, 12));
  const int rc = build_grid(c);  // one host round trip inside (cloud bounds -> grid dimensions)
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 13));
  c->grid_pending = true;        // duration collected at the next synchronisation point
  c->cnt.n_points = (int64_t)c->n;
  c->cnt.n_valid_points = (int64_t)c->n_valid;
  c->has_cloud = true;
  return 0;
}

}  // namespace ag2

using namespace ag2;

extern "C" {

int ag2_abi_version(void) { return AG2_ABI_VERSION; }

void ag2_default_params(ag2_params* p) {
  memset(p, 0, sizeof(*p));
  p->finger_width = 0.01;
  p->hand_outer_diameter = 0.09;
  p->hand_depth = 0.06;
  p->hand_height = 0.02;
  p->init_bite = 0.015;
  p->nn_radius_taubin = 0.01;
  p->nn_radius_hands = 0.1;
  p->normals_radius = 0.01;
  p->grid_cell = 0.01;
  p->num_orientations = 8;
  p->num_threads = 1;
  p->n_cams = 1;
  p->filter_half_grasps = 1;
  const double ws[6] = {-1e30, 1e30, -1e30, 1e30, -1e30, 1e30};
  for (int i = 0; i < 6; i++) p->workspace[i] = ws[i];
  p->min_aperture = 0.03;
  p->max_aperture = 0.07;
  p->min_score_diff = 500.0;
  p->num_selected = 50;
}

int ag2_hand_constants(const ag2_params* p, double* finger_spacing20, double* angles,
                       double* depths32, int32_t* n_depths) {
  if (!p || p->num_orientations < 1 || p->num_orientations > kMaxOrient) return AG2_ERR_ARG;
  HandConst h;
  derive_constants(*p, h, angles);
  if (finger_spacing20) memcpy(finger_spacing20, h.fs, sizeof(h.fs));
  if (depths32) memcpy(depths32, h.depths, sizeof(double) * (size_t)h.n_depths);
  if (n_depths) *n_depths = h.n_depths;
  return 0;
}

ag2_ctx* ag2_create(const ag2_params* p, int device_id) {
  if (!p) return nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "ag2_create: no HIP device available (libag2hip.so has no CPU fallback)\n");
    return nullptr;
  }
  if (device_id < 0 || device_id >= ndev) {
    fprintf(stderr, "ag2_create: device %d out of range (have %d)\n", device_id, ndev);
    return nullptr;
  }
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  ag2_ctx* c = new ag2_ctx();
  c->p = *p;
  c->device = device_id;
  if (p->debug_flags & 2) c->sweep_gcap = 1024;  // test knob: forces the scratch-resize path on small clouds
  if (check_params(c) != 0) {
    fprintf(stderr, "ag2_create: %s\n", c->err.c_str());
    delete c;
    return nullptr;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  c->own_stream = true;
  for (auto& e : c->ev) (void)hipEventCreate(&e);
  // how the host waits (ag2_set_wait_mode changes it per context): AG2_POLL=0 -> the stream; AG2_POLL_SPIN_US
  if (const char* e = getenv("AG2_POLL")) c->wait_poll = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("AG2_POLL_SPIN_US")) c->wait_spin_us = std::max(0, atoi(e));

  derive_constants(c->p, c->hc);
  if (c->d_stats.reserve(sizeof(DevStats)) != hipSuccess || upload_constants(c) != 0 ||
      pin_reserve(c, (size_t)1 << 20) != 0) {
    fprintf(stderr, "ag2_create: device allocation failed\n");
    ag2_destroy(c);
    return nullptr;
  }
  return c;
}

void ag2_destroy(ag2_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);  // NULL = the default stream: still to be waited for
  frame_release(c);
  DevBuf* bufs[] = {&c->d_griddesc, &c->d_lists, &c->d_pairs, &c->d_obox, &c->d_xyz_in, &c->d_key, &c->d_bounds, &c->d_gpos, &c->d_export_list, &c->d_cell, &c->d_perm, &c->d_sorted,
                    &c->d_nrm, &c->d_scan, &c->d_stats, &c->d_hc, &c->d_sample_q, &c->d_frames,
                    &c->d_frame_ok, &c->d_table, &c->d_tab_off, &c->d_tab_keep, &c->d_arena,
                    &c->d_overflow, &c->d_gscratch, &c->d_list, &c->d_list2, &c->d_images,
                    &c->d_logits, &c->d_act1, &c->d_fcpart, &c->d_tmp, &c->d_sel, &c->d_merge, &c->d_gather, &c->d_xchg, &c->d_flags, &c->d_desc,
                    &c->d_raw, &c->d_raw_nrm, &c->d_pre, &c->d_pflags, &c->d_bitmap, &c->d_wrank,
                    &c->d_first, &c->d_prestats, &c->d_hist, &c->d_samples, &c->d_preframe, &c->d_cand, &c->d_cluster, &c->d_cluster_tmp, &c->d_rlist, &c->net.w1p, &c->net.b1,
                    &c->net.w2p, &c->net.b2, &c->net.w3p, &c->net.b3, &c->net.w4, &c->net.b4,
                    &c->net.w1x, &c->net.w2x, &c->net.w3x, &c->d_sweep_prof};
  for (DevBuf* b : bufs) b->release();
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  for (auto& e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* ag2_last_error(const ag2_ctx* c) { return c ? c->err.c_str() : "null context"; }

int ag2_set_stream(ag2_ctx* c, void* hip_stream) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  c->own_stream = false;
  c->stream = (hipStream_t)hip_stream;  // NULL = the HIP default (null) stream, torch's default
  return 0;
}

int ag2_set_cloud(ag2_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                  const int32_t* cam_source, int n_cams, const double* normals) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (n_cams != c->p.n_cams) return set_err(c, AG2_ERR_ARG, "n_cams differs from ag2_params.n_cams");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return set_err(c, AG2_ERR_ARG, "bad stride");
  if (n > 0 && !xyz) return set_err(c, AG2_ERR_ARG, "xyz is NULL");
  if (n > (size_t)1 << 30) return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 points");
  c->n = n;
  c->has_cloud = c->has_normals = false;
  c->bounds_known = false;
  std::vector<float> pack(n * 4);
  const char* base = (const char*)xyz;
  for (size_t i = 0; i < n; i++) {
    const float* pt = (const float*)(base + i * stride_bytes);
    int mask = 0;
    for (int cam = 0; cam < n_cams; cam++) {
      const int v = cam_source ? cam_source[i * (size_t)n_cams + cam] : 1;  // cloud_camera.cpp:59
      if (v == 1) mask |= (1 << cam);
    }
    pack[4 * i] = pt[0];
    pack[4 * i + 1] = pt[1];
    pack[4 * i + 2] = pt[2];
    memcpy(&pack[4 * i + 3], &mask, 4);
  }
  AG2_HIP(c, c->d_xyz_in.reserve(std::max<size_t>(n, 1) * 16));
  if (n) AG2_HIP(c, hipMemcpyAsync(c->d_xyz_in.p, pack.data(), n * 16, hipMemcpyHostToDevice, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  int rc = after_cloud(c);
  if (rc) return rc;
  if (normals && c->n_valid) {
    // cloud_camera.cpp:27-31: normals originate from float PointNormal fields; stored as float.
    std::vector<float> nf(n * 4, 0.f);
    for (size_t i = 0; i < n; i++) {
      nf[4 * i] = (float)normals[3 * i];
      nf[4 * i + 1] = (float)normals[3 * i + 1];
      nf[4 * i + 2] = (float)normals[3 * i + 2];
    }
    AG2_HIP(c, c->d_tmp.reserve(n * 16));
    AG2_HIP(c, hipMemcpyAsync(c->d_tmp.p, nf.data(), n * 16, hipMemcpyHostToDevice, c->stream));
    rc = gather_normals(c);
    if (rc) return rc;
    AG2_HIP(c, hipStreamSynchronize(c->stream));
    c->has_normals = true;
  }
  return 0;
}

int ag2_set_cloud_device(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (c->p.n_cams != 1) return set_err(c, AG2_ERR_ARG, "device clouds are single-camera");
  if (stride_bytes < 12 || stride_bytes % 4 != 0) return set_err(c, AG2_ERR_ARG, "bad stride");
  if (n > 0 && !d_xyz) return set_err(c, AG2_ERR_ARG, "d_xyz is NULL");
  if (n > (size_t)1 << 30) return set_err(c, AG2_ERR_CAPACITY, "more than 2^30 points");
  c->n = n;
  c->has_cloud = c->has_normals = false;
  c->bounds_known = false;
  AG2_HIP(c, c->d_xyz_in.reserve(std::max<size_t>(n, 1) * 16));
  const int rc = pack_device_xyz(c, d_xyz, n, stride_bytes, c->d_xyz_in.as<float4>(), /*with_bounds=*/true);
  if (rc) return rc;
  return after_cloud(c);
}

int ag2_compute_normals(ag2_ctx* c) {
  if (!c) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->has_cloud) return set_err(c, AG2_ERR_STATE, "no cloud set");
  // asynchronous: the kernel time and the K1 counter are collected at the next point that
  // synchronises anyway (ag2::collect_normals_stats)
  // sum_k1 was zeroed with the rest of the per-cloud block when the grid was built (k_init_stats);
  // only a repeated call on the same cloud has to clear it again
  if (c->has_normals)
    AG2_HIP(c, hipMemsetAsync((char*)c->d_stats.p + offsetof(DevStats, sum_k1), 0, 8, c->stream));
  AG2_HIP(c, stage_event(c, 9));
  const int rc = launch_normals(c);
  if (rc) return rc;
  AG2_HIP(c, stage_event(c, 10));
  c->normals_pending = true;
  c->has_normals = true;
  return 0;
}

int ag2_get_normals(ag2_ctx* c, double* out) {
  if (!c || !out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->has_normals) return set_err(c, AG2_ERR_STATE, "no normals");
  const double nanv = nan("");
  for (size_t i = 0; i < 3 * c->n; i++) out[i] = nanv;
  if (c->n_valid == 0) return 0;
  std::vector<float> nf(c->n_valid * 4);
  std::vector<int32_t> perm(c->n_valid);
  AG2_HIP(c, hipMemcpyAsync(nf.data(), c->d_nrm.p, c->n_valid * 16, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipMemcpyAsync(perm.data(), c->d_perm.p, c->n_valid * 4, hipMemcpyDeviceToHost, c->stream));
  AG2_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t pos = 0; pos < c->n_valid; pos++) {
    const size_t i = (size_t)perm[pos];
    out[3 * i] = (double)nf[4 * pos];
    out[3 * i + 1] = (double)nf[4 * pos + 1];
    out[3 * i + 2] = (double)nf[4 * pos + 2];
  }
  return 0;
}

int ag2_get_grid_perm(ag2_ctx* c, int32_t* perm, size_t cap, size_t* n_valid) {
  if (!c || !n_valid) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->has_cloud) return set_err(c, AG2_ERR_STATE, "no cloud set");
  *n_valid = c->n_valid;
  if (cap < c->n_valid) return set_err(c, AG2_ERR_CAPACITY, "perm buffer too small");
  if (c->n_valid) {  // ordered behind the grid kernels on the context's stream
    AG2_HIP(c, hipMemcpyAsync(perm, c->d_perm.p, c->n_valid * 4, hipMemcpyDeviceToHost, c->stream));
    AG2_HIP(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

int ag2_set_stage_timing(ag2_ctx* c, int level) {
  if (!c) return AG2_ERR_ARG;
  if (level < 0 || level > 2) return set_err(c, AG2_ERR_ARG, "stage timing level must be 0, 1 or 2");
  (void)hipSetDevice(c->device);
  AG2_HIP(c, hipStreamSynchronize(c->stream));  // no half-recorded pairs
  c->stage_timing = level;
  memset(&c->times, 0, sizeof(c->times));
  c->grid_pending = false;  // (its events may not exist at the new level)
  return 0;
}

int ag2_set_wait_mode(ag2_ctx* c, int poll, int spin_us) {
  if (!c) return AG2_ERR_ARG;
  if (poll < 0 || poll > 1 || spin_us < 0) return set_err(c, AG2_ERR_ARG, "wait mode: poll 0 / 1, spin_us >= 0");
  c->wait_poll = poll;
  c->wait_spin_us = spin_us;
  return 0;
}

int ag2_get_wait_info(ag2_ctx* c, ag2_wait_info* out) {
  if (!c || !out) return AG2_ERR_ARG;
  memset(out, 0, sizeof(*out));
  out->poll = c->wait_poll;
  out->spin_us = c->wait_spin_us;
  out->poll_fallbacks = c->poll_fallbacks;
  out->poll_yields = c->poll_yields;
  out->last_submit_us = c->last_submit_us;
  out->last_wait_us = c->last_wait_us;
  return 0;
}

int ag2_set_grid_origin(ag2_ctx* c, const float* origin3) {
  if (!c) return AG2_ERR_ARG;
  c->origin_set = origin3 != nullptr;
  if (origin3) {
    for (int a = 0; a < 3; a++) {
      if (!std::isfinite(origin3[a])) {
        c->origin_set = false;
        return set_err(c, AG2_ERR_ARG, "grid origin must be finite");
      }
      c->origin[a] = origin3[a];
    }
  }
  return 0;
}

int ag2_get_counters(ag2_ctx* c, ag2_counters* out) {
  if (!c || !out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  (void)rank_spec_collect(c, /*stream_is_idle=*/false);
  (void)collect_normals_stats(c);
  *out = c->cnt;
  out->detect_one_trip = c->spec_runs;
  out->detect_redone = c->spec_fallbacks;
  return 0;
}

int ag2_get_stage_times(ag2_ctx* c, ag2_times* out) {
  if (!c || !out) return AG2_ERR_ARG;
  (void)hipSetDevice(c->device);
  (void)rank_spec_collect(c, /*stream_is_idle=*/false);
  (void)collect_normals_stats(c);
  *out = c->times;
  return 0;
}

}  // extern "C"
