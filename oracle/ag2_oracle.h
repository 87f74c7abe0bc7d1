/*
 * ag2_oracle.h -- C API of the CPU oracle for the agile_grasp2 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  The product (agile_grasp2_amd/csrc, libag2hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference (gwding/agile_grasp2) ships no tests, fixtures, golden vectors,
 * point clouds or trained weights, and cannot be compiled in this image (Eigen, PCL/FLANN, OpenCV,
 * Caffe, Boost and ROS are absent).  This oracle is a dependency-free restatement of the
 * reference's algorithm, every function citing the reference file:line it follows; its third-party
 * pieces (PCL radius search + normal estimation, Eigen eigen-solver, OpenCV dilate/convertTo,
 * Caffe LeNet layers) are restated from their documented behaviour and pinned against independent
 * numpy / scipy / torch-CPU re-derivations (tests/golden/make_golden.py), not against the
 * libraries themselves.
 */
#ifndef AG2_ORACLE_H
#define AG2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ag2o_ctx ag2o_ctx;

/* Same field order as ag2_params in include/ag2_c.h so a test can fill both from one dict. */
typedef struct ag2o_params {
  /* hand geometry -- HandSearch::Parameters, include/agile_grasp2/hand_search.h:72-91 */
  double finger_width;        /* 0.01  grasp_detector.cpp:37 */
  double hand_outer_diameter; /* 0.09  grasp_detector.cpp:38 */
  double hand_depth;          /* 0.06  grasp_detector.cpp:40 */
  double hand_height;         /* 0.02  grasp_detector.cpp:41 */
  double init_bite;           /* 0.015 grasp_detector.cpp:42 (launch files: 0.01) */
  /* neighbourhood radii */
  double nn_radius_taubin;    /* 0.01  grasp_detector.cpp:30 */
  double nn_radius_hands;     /* 0.1   grasp_detector.cpp:31 */
  double normals_radius;      /* 0.01  hard-coded at hand_search.cpp:91 */
  double grid_cell;           /* uniform-grid cell edge (new; no reference counterpart) */
  int32_t num_orientations;   /* 8     grasp_detector.cpp:32 */
  int32_t num_threads;        /* 1     grasp_detector.cpp:29 */
  int32_t n_cams;             /* rows of camera_source_, 1 or 2 */
  int32_t filter_half_grasps; /* true  grasp_detector.cpp:34 */
  double cam_origin[2][3];    /* translation column of cam_tf_left/right, local_frame.cpp:8-12 */
  /* prune / select -- grasp_detector.cpp:363-395, :198-207, :239-252 */
  double workspace[6];
  double min_aperture;        /* 0.03  grasp_detector.cpp:70 */
  double max_aperture;        /* 0.07  grasp_detector.cpp:71 */
  double min_score_diff;      /* 500   grasp_detector.cpp:51 */
  int32_t num_selected;       /* 50    grasp_detector.cpp:72 */
  int32_t reserved;
} ag2o_params;

/* 176-byte hypothesis record, identical layout to ag2_hypothesis in include/ag2_c.h. */
typedef struct ag2o_hypothesis {
  double axis[3], approach[3], binormal[3];
  double surface[3], bottom[3], top[3];
  double width;
  double score;
  int32_t sample_slot;   /* global sample slot (slot_base + i) */
  int32_t orientation;   /* 0..R-1 */
  uint8_t half_antipodal, full_antipodal;
  uint16_t reserved;
  int32_t n_points;      /* P: points in the closing region */
} ag2o_hypothesis;

typedef struct ag2o_counters {
  int64_t n_points, n_valid_points, n_samples, n_frames, n_hypotheses, n_pruned, n_scored, n_selected;
  int64_t sum_k1, sum_k2, sum_kcrop, sum_p; /* neighbour / crop / in-box counts, for roofline bytes */
  double t_normals, t_frames, t_hands, t_images, t_lenet, t_total; /* seconds */
} ag2o_counters;

void ag2o_default_params(ag2o_params* p);
/* finger_spacing_ (finger_hand.cpp:7-12), hand angles (hand_search.cpp:179-180), deepenHand depths
 * (finger_hand.cpp:118-122) as this restatement derives them; any output may be NULL. */
int ag2o_hand_constants(const ag2o_params* p, double* finger_spacing20, double* angles,
                        double* depths32, int32_t* n_depths);
ag2o_ctx* ag2o_create(const ag2o_params* p);
void ag2o_destroy(ag2o_ctx* c);
const char* ag2o_last_error(const ag2o_ctx* c);

/* xyz: n points, stride_bytes between points (12 for packed xyz, 32 for pcl::PointXYZRGBA).
 * cam_source: n_cams x n column-major int32 or NULL (= all ones, cloud_camera.cpp:59).
 * normals: 3 x n column-major double or NULL (then ag2o_compute_normals must run). */
int ag2o_set_cloud(ag2o_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                   const int32_t* cam_source, int n_cams, const double* normals);
int ag2o_compute_normals(ag2o_ctx* c);
int ag2o_get_normals(ag2o_ctx* c, double* out3xn);
/* sorted position -> original index (ascending (cell key, index)); invalid points at the end */
int ag2o_get_grid_perm(ag2o_ctx* c, int32_t* perm_n);
/* neighbours of q within r in canonical order, as ORIGINAL indices */
int ag2o_radius_search(ag2o_ctx* c, const float* q3, double r, int32_t* out_idx, size_t cap,
                       size_t* n_out);
/* frames: s x 12 doubles (sample, normal, binormal, curvature_axis); valid[s] */
int ag2o_local_frames(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                      uint64_t slot_base, uint64_t seed, double* frames_sx12, int32_t* valid);
/* exactly one of sample_idx / sample_xyz (3 x s column-major) is non-NULL */
int ag2o_generate_hypotheses(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz,
                             size_t s, uint64_t slot_base, uint64_t seed, ag2o_hypothesis* out,
                             size_t cap, size_t* n_out);
/* in-box (unit-box scaled) points and rotated normals of hypothesis h of the last generate call,
 * 3 x P column-major each */
int ag2o_hyp_points(ag2o_ctx* c, size_t h, double* pts3xp, double* nrm3xp);
/* keep[h] = 1 if hypothesis h of the last generate call survives pruneGraspsOnHandParameters */
int ag2o_prune(ag2o_ctx* c, uint8_t* keep, size_t n);
/* 60x60x3 uint8 HWC images of hypotheses [first, first+count) of the last generate call */
int ag2o_render_images(ag2o_ctx* c, size_t first, size_t count, uint8_t* out_hwc);
int ag2o_render_image_from_points(const double* pts3xp, const double* nrm3xp, size_t p,
                                  uint8_t* out_hwc);
/* Caffe blob order: conv OIHW, inner product out x in */
int ag2o_lenet_load(ag2o_ctx* c, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                    const float* conv2_b, const float* ip1_w, const float* ip1_b,
                    const float* ip2_w, const float* ip2_b);
int ag2o_lenet_forward(ag2o_ctx* c, const uint8_t* images_hwc, size_t n, float* ip2_out);
/* whole path: hypotheses -> [prune] -> images -> LeNet -> threshold -> top-k (score desc).
 * scores_all (optional, cap_all >= #hypotheses after prune) receives every scored hypothesis. */
int ag2o_detect(ag2o_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                uint64_t slot_base, uint64_t seed, int do_prune, ag2o_hypothesis* selected,
                size_t cap, size_t* n_selected, ag2o_hypothesis* scored_all, size_t cap_all,
                size_t* n_scored);
int ag2o_get_counters(ag2o_ctx* c, ag2o_counters* out);

/* HandleSearch::findClusters(hand_list, remove_inliers), handle_search.cpp:4-80.  Output in input
 * order; returns -2 (with *n_out set) when cap is too small.  min_inliers >= 1. */
int ag2o_find_clusters(const ag2o_hypothesis* hands, size_t n, int min_inliers, int remove_inliers,
                       ag2o_hypothesis* out, size_t cap, size_t* n_out);
/* HandleSearch::setMinInliers: > 0 makes ag2o_detect cluster before the top-k (grasp_detector.cpp:228-236) */
int ag2o_set_min_inliers(ag2o_ctx* c, int min_inliers);

/* Spatial tiles (multi-GPU): origin of the search grid for the clouds set afterwards
 * (NULL = automatic: per-axis minimum).  A tile passes the minimum of the WHOLE cloud, which makes
 * its binning, its canonical neighbour order and the prune's cloud minimum those of the unsplit run. */
int ag2o_set_grid_origin(ag2o_ctx* c, const float* origin3);

/* Preprocessing in front of the path (GraspDetector::preprocessPointCloud, grasp_detector.cpp:285-335):
 * CloudCamera::filterWorkspace (cloud_camera.cpp:89-121, bounds = params.workspace) and
 * CloudCamera::voxelizeCloud (:124-168); the result becomes the context's cloud.  flags bit 0:
 * a voxel's camera source comes from the first point that hit it (default: the literal
 * reference indexing, :137-152). */
int ag2o_preprocess_cloud(ag2o_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                          const int32_t* cam_source, int n_cams, const double* normals,
                          int filter_workspace, int voxelize, double voxel_size, int flags,
                          size_t* n_out);
/* current cloud: xyz n x 3 packed, cam_source n_cams x n column-major (either may be NULL) */
int ag2o_get_cloud(ag2o_ctx* c, float* xyz_nx3, int32_t* cam_source, size_t cap, size_t* n);
/* CloudCamera::subsampleUniformly (cloud_camera.cpp:171-178): min(num_samples, n) indices, ascending */
int ag2o_subsample_uniformly(ag2o_ctx* c, size_t num_samples, uint64_t seed, int32_t* idx_out,
                             size_t cap, size_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* AG2_ORACLE_H */
