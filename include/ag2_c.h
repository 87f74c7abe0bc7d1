/*
 * ag2_c.h -- C-ABI of libag2hip.so: the MI355X (gfx950) implementation of the agile_grasp2 hot
 * path  sample -> normals -> local frames -> hand search -> grasp image -> LeNet score -> select.
 *
 * The reference (gwding/agile_grasp2) has NO plugin/FFI layer: its ROS node links the C++ classes
 * directly (CMakeLists.txt:154-157).  This ABI is therefore the seam the C++ host mirror
 * (the headers under include/agile_grasp2/) sits on; each entry point names the reference code it replaces
 * (paths relative to the reference root).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - return 0 on success, a negative AG2_ERR_* otherwise; ag2_last_error() gives the message.
 *     Nothing aborts (the reference aborts through glog CHECK, caffe_classifier.cpp:16-34).
 *   - the caller owns every host buffer; the context owns all device memory.
 *   - a context is single-threaded (the reference's HandSearch is not re-entrant either,
 *     hand_search.cpp:14-15); distinct contexts, one per GPU, may run concurrently.
 *   - matrices are column-major like Eigen's: "3 x n" means element (r, c) at [c*3 + r].
 *   - there is no CPU fallback: without a usable HIP device ag2_create returns NULL.
 */
#ifndef AG2_C_H
#define AG2_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG2_ABI_VERSION 1

enum {
  AG2_OK = 0,
  AG2_ERR_ARG = -1,       /* bad argument / call order */
  AG2_ERR_HIP = -2,       /* HIP runtime error */
  AG2_ERR_CAPACITY = -3,  /* an output or internal buffer is too small (message says which) */
  AG2_ERR_STATE = -4,     /* missing cloud / normals / weights */
  AG2_ERR_RETRY = -5      /* ag2_merge_*: a rank ran its detect at shapes learned from its previous call and they
                             did not hold (its exported header says so, every rank reads it): nothing was merged;
                             EVERY rank repeats the step -- its next ag2_detect runs step by step and learns the
                             new shapes */
};

typedef struct ag2_ctx ag2_ctx;

/* Parameters: names and defaults of GraspDetector's ROS params (grasp_detector.cpp:19-80) and
 * HandSearch::Parameters (include/agile_grasp2/hand_search.h:72-91). */
typedef struct ag2_params {
  double finger_width;        /* 0.01  */
  double hand_outer_diameter; /* 0.09  */
  double hand_depth;          /* 0.06  */
  double hand_height;         /* 0.02  */
  double init_bite;           /* 0.015 (launch files use 0.01) */
  double nn_radius_taubin;    /* 0.01  */
  double nn_radius_hands;     /* 0.1   */
  double normals_radius;      /* 0.01, hard-coded at hand_search.cpp:91 */
  double grid_cell;           /* 0.01, edge of the uniform search grid (replaces the FLANN kd-tree) */
  int32_t num_orientations;   /* 8     */
  int32_t num_threads;        /* kept for API parity; ignored on the GPU */
  int32_t n_cams;             /* 1 or 2 */
  int32_t filter_half_grasps; /* 1     */
  double cam_origin[2][3];    /* translation column of cam_tf_left / cam_tf_right */
  double workspace[6];        /* [minX maxX minY maxY minZ maxZ] */
  double min_aperture;        /* 0.03 */
  double max_aperture;        /* 0.07 */
  double min_score_diff;      /* 500  */
  int32_t num_selected;       /* 50   */
  int32_t debug_flags;        /* 0. bit0: visit every radius neighbour in the hand sweep (no
                                 sphere/slab row culling) so counters.sum_k2 is exact; results
                                 are identical either way.  bit1: start the sweep's list arena at
                                 4 096 points instead of 16 Mi: exercises the grow-and-repeat path on
                                 small clouds; results identical */
} ag2_params;

/* One grasp hypothesis = the fixed part of GraspHypothesis
 * (include/agile_grasp2/grasp_hypothesis.h:297-312).  176 bytes; this is also the slot of the
 * fixed-slot candidate table exchanged between GPUs. */
typedef struct ag2_hypothesis {
  double axis[3], approach[3], binormal[3];
  double surface[3], bottom[3], top[3];
  double width;
  double score;
  int32_t sample_slot;  /* global sample slot = slot_base + position in the sample list */
  int32_t orientation;  /* 0 .. num_orientations-1 */
  uint8_t half_antipodal, full_antipodal;
  uint16_t reserved;
  int32_t n_points;     /* points in the closing region; 0 marks an empty table slot */
} ag2_hypothesis;

typedef struct ag2_counters {
  int64_t n_points, n_valid_points, n_samples, n_frames, n_hypotheses, n_pruned, n_scored, n_selected;
  int64_t sum_k1, sum_k2, sum_kcrop, sum_p;  /* measured neighbourhood sizes (roofline bytes) */
  int64_t n_overflow_samples;                /* samples whose cropped list did not fit the first sweep stage */
  int64_t list_points;                       /* split sweep: room taken in the list arena, points of 16 B (long lists
                                                reserve their candidate count, about 1.7 x their length) */
  int64_t detect_one_trip;                   /* ag2_detect calls of this context served in ONE host round trip (tail
                                                launched at the shapes the previous call left) ... */
  int64_t detect_redone;                     /* ... and those whose shapes did not hold (more images, a larger
                                                in-box list, a buffer too small) and that ran again step by step.
                                                Both are per context, not per cloud. */
} ag2_counters;

/* Device time of the last call per stage, milliseconds (HIP events on the context's stream).
 * Stage names follow the reference's own probes (hand_search.cpp:30,58,167,232;
 * grasp_detector.cpp:210,254). */
typedef struct ag2_times {
  float grid_ms;        /* K0 search grid (ag2_set_cloud*) */
  float normals_ms;     /* K1 k_normals */
  float frames_ms;      /* K2 k_frames */
  float sweep_ms;       /* K3 k_sweep, first stage (cropped list in LDS), up to the finger-placement gates */
  float compact_ms;     /* prune-flag compaction + image descriptors */
  float render_ms;      /* K4 k_render */
  float lenet_conv_ms;  /* K5 k_lenet_conv */
  float lenet_fc_ms;    /* K5 k_lenet_fc */
  float select_ms;      /* K6 score scatter, threshold compaction, record gather */
  float total_ms;       /* first to last event of the call */
  float sweep_overflow_ms; /* K3: k_sweep's second stage (long lists, written to the list arena) + k_sweep_orient (one
                              workgroup per surviving (sample, orientation) pair) + k_hyp_stats */
  float preprocess_ms;  /* workspace filter + voxel grid (ag2_preprocess_cloud*), without the grid build */
} ag2_times;

/* Frame mode (BASELINE.json configuration 5), see ag2_detect_frame. */
typedef struct ag2_frame_info {
  int64_t frames;           /* ag2_detect_frame calls */
  int64_t graph_replays;    /* ... served by launching the captured hipGraph */
  int64_t plain_runs;       /* ... served by the same fixed-shape sequence, launched kernel by kernel */
  int64_t stepwise_runs;    /* ... served by set_cloud_device + compute_normals + detect */
  int64_t captures;         /* graphs captured and instantiated */
  int64_t capture_failed, capture_refused;  /* refused: the legacy default stream cannot be captured */
  int64_t fallbacks;        /* fixed-shape frames repeated step by step (a buffer or table was too small) */
  int64_t max_points, max_samples, max_cells, max_images;  /* the fixed shapes in force */
  int64_t graph_ready;
  int64_t last_fallback;    /* why the last fallback happened: 1 point-list arena, 2 cropped-list arena, 8 sweep scratch too
                               small; 16 more grid cells than the table holds; 32 more selected records than the result
                               block; 64 more images than max_images; 128 an in-box list longer than the captured
                               renderers take; << 8: the front end's flags (ag2_detect_frame_raw): 1 voxel lattice larger
                               than the bitmap, 2 more voxels than max_points, 4 / 8 sub-sampling candidate list, 16 no
                               more voxels than num_samples; << 40: a sample needed the sweep's long-list stage, which the
                               sequence leaves out while no frame has needed it */
} ag2_frame_info;

int ag2_abi_version(void);
void ag2_default_params(ag2_params* p);
/* The constants the hand sweep derives from the parameters, computed on the host (no device is
 * touched): finger_spacing[20] = FingerHand's finger_spacing_ (finger_hand.cpp:7-12), angles[R] =
 * the hand orientations (hand_search.cpp:179-180), depths[<= 32] = deepenHand's depth sequence
 * (finger_hand.cpp:118-122).  Any output pointer may be NULL. */
int ag2_hand_constants(const ag2_params* p, double* finger_spacing20, double* angles,
                       double* depths32, int32_t* n_depths);

/* Replaces the constructors of GraspDetector (grasp_detector.cpp:15-81) / HandSearch::setParameters
 * (hand_search.cpp:64-80).  NULL when no HIP device is usable. */
ag2_ctx* ag2_create(const ag2_params* p, int device_id);
void ag2_destroy(ag2_ctx* c);
const char* ag2_last_error(const ag2_ctx* c);
/* Run on the caller's HIP stream (hipStream_t), e.g. torch's current stream; NULL = the HIP
 * default (null) stream.  Without this call the context uses a private non-blocking stream. */
int ag2_set_stream(ag2_ctx* c, void* hip_stream);

/* Replaces CloudCamera's data members (include/agile_grasp2/cloud_camera.h:178-183) and the kd-tree
 * build (hand_search.cpp:11-12).  xyz: n points, stride_bytes apart (12 packed, 32 for
 * pcl::PointXYZRGBA).  cam_source: n_cams x n int32 or NULL (= ones, cloud_camera.cpp:59).
 * normals: 3 x n double or NULL. */
int ag2_set_cloud(ag2_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                  const int32_t* cam_source, int n_cams, const double* normals);
/* Same, but xyz already lives in device memory (HBM-resident input for benchmarks/streams). */
int ag2_set_cloud_device(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes);
/* HandSearch::calculateNormalsOMP, hand_search.cpp:83-94. */
int ag2_compute_normals(ag2_ctx* c);
int ag2_get_normals(ag2_ctx* c, double* out3xn);
/* sorted position -> original index for the n_valid finite points (tests / debugging). */
int ag2_get_grid_perm(ag2_ctx* c, int32_t* perm, size_t cap, size_t* n_valid);
/* HandSearch::calculateLocalFrames, hand_search.cpp:97-170 / :238-317 (tests / debugging):
 * frames s x 12 doubles (sample, normal, binormal, curvature axis), valid[s]. */
int ag2_local_frames(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
                     uint64_t slot_base, uint64_t seed, double* frames_sx12, int32_t* valid);
/* HandSearch::generateHypotheses, hand_search.cpp:4-61 (frames + evaluateHands + calculateHand).
 * Exactly one of sample_idx (indices into the cloud, CloudCamera::getSampleIndices) and
 * sample_xyz (3 x s, CloudCamera::getSamples) is non-NULL.  Output in sample order. */
int ag2_generate_hypotheses(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz,
                            size_t s, uint64_t slot_base, uint64_t seed, ag2_hypothesis* out,
                            size_t cap, size_t* n_out);
/* GraspHypothesis::getPointsForLearning / getNormalsForLearning of hypothesis h of the last
 * generate call: 3 x n_points each. */
int ag2_hyp_points(ag2_ctx* c, size_t h, double* pts3xp, double* nrm3xp);
/* GraspDetector::pruneGraspsOnHandParameters, grasp_detector.cpp:363-395, on the last generate
 * call's hypotheses. */
int ag2_prune(ag2_ctx* c, uint8_t* keep, size_t n);
/* Learning::createGraspImages, learning.cpp:4-33, for hypotheses [first, first+count) of the last
 * generate call: count x 60 x 60 x 3 uint8 (HWC, channel order after the BGR2RGB swap). */
int ag2_render_images(ag2_ctx* c, size_t first, size_t count, uint8_t* out_hwc);
/* Same for caller-supplied point lists: hypothesis i owns columns [offsets[i], offsets[i+1]). */
int ag2_render_images_from_points(ag2_ctx* c, size_t n, const int64_t* offsets_np1,
                                  const double* pts3xp, const double* nrm3xp, uint8_t* out_hwc);
/* Classifier ctor's CopyTrainedLayersFrom, caffe_classifier.cpp:13-14.  Caffe blob order. */
int ag2_lenet_load(ag2_ctx* c, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                   const float* conv2_b, const float* ip1_w, const float* ip1_b,
                   const float* ip2_w, const float* ip2_b);
/* Classifier::ClassifyBatch / PredictBatch, caffe_classifier.cpp:70-127: n x 2 raw ip2 logits.
 * fp32 like Caffe.  By default conv1, conv2 and ip1 run on the bf16 matrix cores with every fp32
 * operand written as the exact sum of three bf16 terms and fp32 accumulation (deviation from a
 * sequential fp32 evaluation: that of a re-ordered fp32 sum); with AG2_LENET_F32=1 in the
 * environment when ag2_lenet_load is called they run on the f32-input matrix instructions. */
int ag2_lenet_forward(ag2_ctx* c, const uint8_t* images_hwc, size_t n, float* ip2_out);
/* GraspDetector::detectGraspPoses, grasp_detector.cpp:84-282 (antipodal_mode PREDICTION, no
 * clustering): hypotheses -> [prune] -> images -> LeNet -> score >= min_score_diff -> top
 * num_selected by score.  scored_all (optional) receives every scored hypothesis in order.
 * Host round trips: two on a context's first call (the sweep's statistics size the launches of the
 * renderer and of LeNet; the selected records come back and are sorted on the host), ONE from the
 * second call on when only the selection is asked for (scored_all == NULL): the tail is
 * launched at the shapes the previous call left with the list length read on the device, the top
 * num_selected are picked on the device, and if the statistics that come back with them say the
 * shapes did not hold the call runs again in the two-trip form.  Same bytes either way
 * (AG2_DETECT_STEPWISE=1 in the environment keeps the two-trip form, for A/B). */
int ag2_detect(ag2_ctx* c, const int32_t* sample_idx, const double* sample_xyz, size_t s,
               uint64_t slot_base, uint64_t seed, int do_prune, ag2_hypothesis* selected,
               size_t cap, size_t* n_selected, ag2_hypothesis* scored_all, size_t cap_all,
               size_t* n_scored);
/* Fixed-slot candidate table of the last detect call: s * num_orientations records, slot
 * (i * R + orientation), n_points == 0 where empty, score filled where scored.  Copied
 * device-to-device into d_dst (e.g. a torch tensor) for the RCCL all-gather. */
/* One frame of a cloud stream = ag2_set_cloud_device + ag2_compute_normals + ag2_detect (with index
 * samples, slot_base 0, without the scored-records output), same results byte for byte.  Replaces
 * the body of the node's live-topic loop, src/nodes/grasp_detection_node.cpp:69-95 (run) / :123-143
 * (detectGraspPosesInTopic -> GraspDetector::detectGraspPoses).  The per-frame pipeline runs at fixed
 * maximum shapes with no host round trip before the results and is captured in a hipGraph that later
 * frames replay; the first frame (and any frame that outgrows the shapes: more points, samples or
 * grid cells, a longer point-list arena) runs step by step and sets them.  xyz: n points stride_bytes
 * apart in device (xyz_on_device != 0) or host memory; single-camera clouds.
 * ag2_stream_configure is optional: it presets the maxima (0 = learn from the first frame) and can
 * turn the graph off (the fixed-shape sequence is then launched kernel by kernel). */
int ag2_stream_configure(ag2_ctx* c, size_t max_points, size_t max_samples, int use_graph);
int ag2_detect_frame(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                     const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune,
                     ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored);
/* The same, one step earlier: a frame of the RAW sensor cloud.  Replaces, per frame,
 * GraspDetector::preprocessPointCloud (grasp_detector.cpp:285-335: CloudCamera::filterWorkspace,
 * cloud_camera.cpp:89-121 -- bounds = ag2_params.workspace, when filter_workspace != 0 --, voxelizeCloud
 * :124-168 with voxel_size, subsampleUniformly :171-178 with num_samples and sample_seed) followed by
 * GraspDetector::detectGraspPoses (:84-282), i.e. what the node does for a cloud it loads
 * (grasp_detection_node.cpp:97-121) and what a live topic has to do for every frame (:123-143).
 * Same results byte for byte as ag2_preprocess_cloud_device + ag2_subsample_uniformly (indices left on the
 * device) + ag2_compute_normals + ag2_detect.  Filter, voxel grid and sub-sampling run on the device inside
 * the captured sequence: the processed cloud, the search grid and the sample indices never leave HBM and
 * the host synchronises once per frame.  A frame with no more voxels than num_samples (every point becomes
 * a sample, grasp_detector.cpp:322-330) and any frame that outgrows the shapes run step by step.
 * n_voxels (may be NULL): size of the processed cloud, which is the context's cloud afterwards
 * (ag2_get_cloud, ag2_get_normals; ag2_get_samples for the indices drawn). */
int ag2_detect_frame_raw(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                         int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                         uint64_t seed, int do_prune, ag2_hypothesis* selected, size_t cap,
                         size_t* n_selected, size_t* n_scored, size_t* n_voxels);
int ag2_get_frame_info(ag2_ctx* c, ag2_frame_info* out);

/* How the host waits for results (no counterpart in the reference, which computes on the calling thread).
 * The last kernel of a step writes its results and then a sequence number into coherent page-locked memory;
 * the host can POLL that word instead of waiting for the stream, which returns ~10 us earlier per wait (two
 * waits per detect step, one per frame).  Cost: the waiting thread occupies its core -- it spins for `spin_us`
 * microseconds (default 50), then keeps polling but yields the core between looks (sched_yield), and after
 * 5 ms falls back to hipStreamSynchronize.  A caller with more contexts than spare cores (ag2_pipe, one thread
 * per device) should lower spin_us or turn polling off.
 *   poll = 1 (default; AG2_POLL=0 in the environment changes the default), spin_us >= 0 (AG2_POLL_SPIN_US).
 * Results are byte-identical in every mode. */
int ag2_set_wait_mode(ag2_ctx* c, int poll, int spin_us);
typedef struct ag2_wait_info {
  int64_t poll, spin_us;
  int64_t poll_fallbacks;   /* waits that polled 5 ms without seeing the flag and went on to wait for the stream */
  int64_t poll_yields;      /* sched_yield calls made while polling */
  int64_t last_submit_us;   /* host time inside the last ag2_submit_frame* (or the submit half of ag2_detect_frame*) */
  int64_t last_wait_us;     /* host time inside the last ag2_wait_frame (or the wait half of ag2_detect_frame*) */
} ag2_wait_info;
int ag2_get_wait_info(ag2_ctx* c, ag2_wait_info* out);

/* ---- the asynchronous form of the two frame entries (no counterpart in the reference, whose node handles one
 * cloud at a time, grasp_detection_node.cpp:69-95) ----
 * ag2_submit_frame[_raw] = ag2_detect_frame[_raw] up to, and not including, the wait for the results: a cloud
 * in host memory is copied into page-locked staging and transferred by an asynchronous DMA, the per-frame
 * sequence is queued on the context's stream, and the call returns.  ag2_wait_frame brings the results of the
 * frame submitted last: same bytes as the synchronous call.  One frame per context may be in flight; no other
 * call on the context in between; a device-resident cloud must stay valid until the wait.  (A frame that has to
 * run step by step -- the first of a stream, one that outgrows the shapes -- does so inside the submit, or inside
 * the wait when the flags that come back say so.) */
int ag2_submit_frame(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                     const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune);
int ag2_submit_frame_raw(ag2_ctx* c, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                         int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                         uint64_t seed, int do_prune);
int ag2_wait_frame(ag2_ctx* c, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
                   size_t* n_voxels);
/* ag2_pipe: `depth` contexts on one GPU (own streams), taken in turn by ONE caller thread -- while LeNet and the
 * selection of frame k run on one stream, transfer, front end, grid, normals and sweep of frame k + 1 run on
 * the next; a single cloud does not fill the GPU.  submit returns AG2_ERR_STATE when `depth` frames are in
 * flight; wait returns the results of the oldest frame (submission order). */
typedef struct ag2_pipe ag2_pipe;
ag2_pipe* ag2_pipe_create(const ag2_params* p, int device_id, int depth);
void ag2_pipe_destroy(ag2_pipe* q);
const char* ag2_pipe_last_error(const ag2_pipe* q);
ag2_ctx* ag2_pipe_context(ag2_pipe* q, int k);  /* context k of the pipe (settings such as ag2_stream_configure) */
int ag2_pipe_lenet_load(ag2_pipe* q, const float* conv1_w, const float* conv1_b, const float* conv2_w,
                        const float* conv2_b, const float* ip1_w, const float* ip1_b, const float* ip2_w,
                        const float* ip2_b);
int ag2_pipe_submit(ag2_pipe* q, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                    const int32_t* sample_idx, size_t s, uint64_t seed, int do_prune);
int ag2_pipe_submit_raw(ag2_pipe* q, const void* xyz, int xyz_on_device, size_t n, size_t stride_bytes,
                        int filter_workspace, double voxel_size, size_t num_samples, uint64_t sample_seed,
                        uint64_t seed, int do_prune);
int ag2_pipe_wait(ag2_pipe* q, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_scored,
                  size_t* n_voxels);

int ag2_export_candidates_device(ag2_ctx* c, void* d_dst, size_t bytes);
/* Multi-GPU merge of the detect results (no counterpart in the single-process reference; this is the
 * step grasp_detector.cpp:239-252 -- top num_selected by score -- becomes when the samples are sharded):
 * ag2_export_selected_compact_device leaves the scored records of the last ag2_detect with score >=
 * min_score_diff (list order; BEFORE the clustering, also when ag2_set_min_inliers > 0) in d_dst as a
 * 16-byte header {count, cap, 0, 0} + min(count, cap) records; the ranks all-gather these buffers
 * (RCCL) and every rank calls ag2_merge_selected_device on the gathered world x (16 + cap x 176)
 * bytes: the lists concatenated in rank order (= sample order), then -- when ag2_set_min_inliers > 0 --
 * HandleSearch::findClusters over the WHOLE gathered list (handle_search.cpp:4-80 counts inliers over
 * all hands, whichever rank found them; grasp_detector.cpp:228-236), then the top num_selected by
 * score, ties by position.  n_total (may be NULL): records that took part (before the clustering).
 * A header whose count exceeds cap (a rank's list was cut) makes the merge return AG2_ERR_CAPACITY. */
int ag2_export_selected_compact_device(ag2_ctx* c, void* d_dst, size_t bytes, size_t cap_records);
/* (ag2_detect with selected == NULL and cap == 0 skips the local read-back and top-k and returns with the
 * tail of the pipeline still queued: what a rank calls when the merge follows.) */
int ag2_merge_selected_device(ag2_ctx* c, const void* d_gathered, size_t world, size_t cap_records,
                              ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_total);
/* The same exchange from ONE process that owns several GPUs (what the C++ host mirror uses:
 * GraspDetector::Params::devices -- one context per device, one host thread each; a process per GPU with an
 * RCCL all-gather is the other way, see bench.py).  ag2_gather_begin sizes a gather buffer on root's
 * GPU for `world` lists of cap_records records; ag2_gather_selected exports what src's last ag2_detect
 * selected from (as ag2_export_selected_compact_device) and copies it into position `rank` of that buffer --
 * a peer copy over xGMI when the devices differ -- and returns when it has landed; it touches only src and
 * its own slot of the buffer, so the ranks may call it from their threads concurrently.
 * ag2_merge_gathered is ag2_merge_selected_device on the buffer. */
int ag2_gather_begin(ag2_ctx* root, size_t world, size_t cap_records);
int ag2_gather_selected(ag2_ctx* root, ag2_ctx* src, size_t rank);
int ag2_merge_gathered(ag2_ctx* root, ag2_hypothesis* selected, size_t cap, size_t* n_selected, size_t* n_total);
/* The same candidates in compact form (what a multi-GPU job should put on the wire: the table is
 * mostly empty): a 16-byte header {uint32 count, uint32 cap_records, 0, 0} followed by
 * min(count, cap_records) records in slot order -- the occupied slots of the table above, so the
 * rank-order concatenation of all ranks' records is the reference's output order
 * (hand_search.cpp:223-228).  count > cap_records tells the receiver the list was cut: exchange the
 * full table instead.  bytes >= 16 + cap_records * sizeof(ag2_hypothesis); asynchronous on the
 * context's stream. */
int ag2_export_candidates_compact_device(ag2_ctx* c, void* d_dst, size_t bytes, size_t cap_records);
/* Spatial tiles (multi-GPU, for clouds too large to replicate): origin of the search grid -- and the
 * cloud minimum the prune test uses (pcl::getMinMax3D, grasp_detector.cpp:152-153) -- for the clouds
 * set afterwards; NULL = automatic (per-axis minimum of the cloud).  A rank that holds only a tile
 * (its samples' x-range plus a halo of nn_radius_hands + normals_radius, points in their original
 * relative order) passes the minimum of the WHOLE cloud: it then bins its points, orders every
 * neighbourhood and prunes exactly as the unsplit run, so its hypotheses are identical.  Every point
 * of a cloud set afterwards must be >= the origin on every axis. */
int ag2_set_grid_origin(ag2_ctx* c, const float* origin3);
/* Which of the HIP events behind ag2_get_stage_times are recorded.  Each costs a few microseconds of
 * stream serialisation (all of them about 3 % of a 1 ms detect step): 2 = every stage (default),
 * 1 = only the sweep as a whole (sweep_ms = both stages and the orientation kernel, sweep_overflow_ms = 0),
 * 0 = none.  Stages without events report 0 ms.
 * The reference has no counterpart (it times with std::clock() around whole calls,
 * grasp_detector.cpp:86-88, :262-266). */
int ag2_set_stage_timing(ag2_ctx* c, int level);

/* ---- the step in front of the path: GraspDetector::preprocessPointCloud, grasp_detector.cpp:285-335 ----
 * Steps 1-2 on the GPU: CloudCamera::filterWorkspace (cloud_camera.cpp:89-121; bounds =
 * ag2_params.workspace, strict, order kept) and CloudCamera::voxelizeCloud (:124-168; voxel value =
 * floor((p - min) / cell) * cell + min in float, output ascending in (ix, iy, iz)).  The result
 * becomes the context's cloud exactly as if handed to ag2_set_cloud; *n_out = its size.  Arguments as
 * ag2_set_cloud.  normals are carried through the filter; they cannot be combined with voxelize
 * (the reference leaves normals_ untouched there).  Non-finite points are always dropped.
 * flags bit 0: a voxel's camera source is that of the first point that hit it; default is the
 * reference's literal indexing (:137-152: k-th voxel in set order <- k-th first-hit point in scan
 * order), identical whenever all points carry the same mask.  The reference's linear indexing
 * of the 2-camera matrix in filterWorkspace (:107) is a bug and is not reproduced. */
#define AG2_PRE_OWNER_CAMERA_SOURCE 1
int ag2_preprocess_cloud(ag2_ctx* c, const float* xyz, size_t n, size_t stride_bytes,
                         const int32_t* cam_source, int n_cams, const double* normals,
                         int filter_workspace, int voxelize, double voxel_size, int flags,
                         size_t* n_out);
/* Same with the raw cloud already in device memory (single camera, no normals). */
int ag2_preprocess_cloud_device(ag2_ctx* c, const void* d_xyz, size_t n, size_t stride_bytes,
                                int filter_workspace, int voxelize, double voxel_size,
                                size_t* n_out);
/* CloudCamera::getCloudProcessed / getCameraSource: the context's current cloud, xyz n x 3 packed,
 * cam_source n_cams x n column-major (either may be NULL). */
int ag2_get_cloud(ag2_ctx* c, float* xyz_nx3, int32_t* cam_source, size_t cap, size_t* n);
/* Step 3, CloudCamera::subsampleUniformly (cloud_camera.cpp:171-178, grasp_detector.cpp:321-335):
 * min(num_samples, n) distinct indices into the current cloud, ascending.  The reference's
 * pcl::RandomSample is seeded from wall time; here point i gets the key
 * (draw(seed, i), i) and the num_samples smallest keys are taken.  The indices stay on the device:
 * a following ag2_generate_hypotheses / ag2_detect with sample_idx == sample_xyz == NULL and
 * s <= *n_out uses the first s of them.  idx_out may be NULL. */
int ag2_subsample_uniformly(ag2_ctx* c, size_t num_samples, uint64_t seed, int32_t* idx_out,
                            size_t cap, size_t* n_out);

/* CloudCamera::getSampleIndices of the indices the device drew last (ag2_subsample_uniformly,
 * ag2_detect_frame_raw) and still holds: *n of them, ascending. */
int ag2_get_samples(ag2_ctx* c, int32_t* idx, size_t cap, size_t* n);

/* ---- the step behind the scoring: HandleSearch::findClusters(hand_list, remove_inliers = false),
 * handle_search.cpp:4-80 ----
 * A hand is kept when at least min_inliers other hands have an axis within 15 degrees, a bottom
 * within 5 cm and within 5 mm in the plane orthogonal to its axis; it is moved by (mean inlier
 * bottom - own bottom) and takes the mean inlier score.  Output in input order.  min_inliers >= 1.
 * (remove_inliers = true is order-dependent and has no caller in the reference; the host mirror
 * keeps it as plain C++.) */
int ag2_find_clusters(ag2_ctx* c, const ag2_hypothesis* hands, size_t n, int min_inliers,
                      ag2_hypothesis* out, size_t cap, size_t* n_out);
/* HandleSearch::setMinInliers (handle_search.h:83-86; GraspDetector reads the ROS parameter
 * min_inliers, default 0, grasp_detector.cpp:59-65).  > 0 makes ag2_detect cluster the hands that
 * passed min_score_diff before the top num_selected are taken (grasp_detector.cpp:228-236). */
int ag2_set_min_inliers(ag2_ctx* c, int min_inliers);

int ag2_get_counters(ag2_ctx* c, ag2_counters* out);
int ag2_get_stage_times(ag2_ctx* c, ag2_times* out);

#ifdef __cplusplus
}
#endif
#endif /* AG2_C_H */
