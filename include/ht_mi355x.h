/*
 * ht_mi355x.h -- C-ABI of the MI355X-native hand-tracking hot path (libht_mi355x.so).
 *
 * This is the drop-in boundary for the per-frame path of IntelRealSense/hand_tracking_samples:
 *   depth tile -> CNN forward (third_party/cnn.h) -> heat-map decode -> dynamics-based pose solver
 *   (include/physmodel.h, third_party/physics.h, third_party/gjk.h) as driven by include/handtrack.h.
 * The reference has no FFI (it is header-only C++), so each entry point cites the reference interface it replaces.
 * C++ wrappers that keep the reference's own names (HandTracker::update, CNN::Eval ...) live in include/ht_handtrack.hpp;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes, no exceptions across the ABI, every function returns an int status
 * (HT_OK == 0); the caller owns all buffers it passes, the library owns its device memory; one context per
 * (GPU, stream); a context is not thread safe, but contexts on different GPUs may be used from one process or
 * from several threads: every entry point makes its context's device current for the call and restores the
 * caller's afterwards.  *_dev entry points take DEVICE pointers and a hipStream_t (passed as void*) and are
 * asynchronous; a NULL stream means the context's own stream (not the legacy default stream); the host-read helpers
 * (ht_capacity_events, ht_frames_overflow, ht_get_tracker_flags, ht_get_cnn_results) wait for the stream of the latest
 * *_dev call before they read.  The other entry points take HOST pointers and are synchronous.
 *
 * Layouts: a pose is 7 floats (position xyz, orientation quaternion xyzw) like Pose (third_party/geometric.h:111-125);
 * a body state is 13 floats (pose, linear momentum, angular momentum, physics.h:103-116);
 * a camera is 12 floats (focal xy, principal xy, depth_scale, pose) like DCamera (include/misc_image.h:30-55).
 */
#ifndef HT_MI355X_H
#define HT_MI355X_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HT_OK 0
#define HT_ERR_ARG 1          /* bad argument (NULL, size, batch > capacity) */
#define HT_ERR_IO 2           /* file missing / malformed */
#define HT_ERR_HIP 3          /* a HIP call failed (no device, out of memory, launch failure) */
#define HT_ERR_STATE 4        /* call sequence error (e.g. weights not loaded) */

#define HT_CNN_IN 4096        /* 64x64x1 input, handtrack.h:108 */
#define HT_CNN_OUT 2304       /* 8 heat-maps 16x16 + 16 rows x 16 bins, handtrack.h:117-118 */
#define HT_CNNB_COUNT 9458400 /* fp32 values in a .cnnb file, cnn.h:288,454,590 */
#define HT_CNN128_IN 16384    /* 128x128x1 input of the larger variant (BASELINE configs[4]) */
#define HT_CNNB128_COUNT 30429920 /* its weights: conv1 416, conv2 16448, fc1 12544*2048 + 2048, fc2 2048*2304 + 2304 */
#define HT_POSE 7
#define HT_STATE 13
#define HT_CAM 12
#define HT_MAX_POINTS 4096   /* stride of the point / cloud-row arrays of the stage calls, and the point capacity a context starts with (ht_reserve_points) */
#define HT_POINTS_LIMIT 76800 /* 320x240: the largest cloud a frame can carry (every pixel in range, subsample_fraction 1) */
#define HT_ANALYSIS 84        /* floats per frame, see ht_stage_decode */

typedef struct ht_ctx ht_ctx;

/* Tunables of HandTracker (handtrack.h:523-547) and the physics globals it sets (physics.h:34-47, handtrack.h:837-838). */
typedef struct ht_params
{
	float full_reset_on_error;      /* 0.6   */
	int   angles_only;              /* 0     */
	int   always_take_cnn;          /* 0     */
	float drangey;                  /* 0.7   */
	int   boundary_planes;          /* 1     */
	float microforce;               /* 1 (synthetic-tracker.cpp:92 sets 3) */
	float cloudforce_max_point;     /* 15    */
	float cloudforce_max_sum;       /* 3000  */
	int   mainthreadpasses;         /* 1 (synthetic-tracker.cpp:93 sets 3) */
	int   subsample_fraction;       /* 4     */
	int   min_point_num;            /* 400   */
	float accum_error_threshold;    /* 0     */
	float min_cray_prob;            /* 0     */
	int   steps, steps_keypoints, steps_keyangles, steps_palmangle, steps_cloudstart, steps_unibody;   /* 5 3 2 2 1 3 */
	int   physics_iterations, physics_iterations_post, physics_use_collision;                          /* 16 4 1 */
	float physics_weak_force;       /* 0.4 (physmodel.h:234) */
	float bone_sum_error_scale;     /* 4   (handtrack.h:369) */
	float unibody_force;            /* 0.1 (handtrack.h:450) */
	int   subsample_voxel;          /* 0: every subsample_fraction-th point; != 0: voxelsubsample (physmodel.h:66-118) for the main-thread cloud (handtrack.h:751) */
	float subsample_size;           /* 0   voxel edge in metres; subsample_fraction is then the least number of points a voxel must hold */
} ht_params;

/* ---- lifecycle --------------------------------------------------------------------------------------------------
 * ht_create      replaces  HandTracker::HandTracker() (handtrack.h:830-839): builds both hand models and the CNN topology.
 *                model_path: the reference's own model file (assets/model_hand.json: "controlcages" + "joints"), which is
 *                built on the host exactly as PhysModel::PhysModel + LoadHandModel build it (physmodel.h:444-475,
 *                handtrack.h:347-366: 2x subdivision, 48-vertex hull, mass properties, planes, ignore lists), or a model
 *                baked earlier by ht_model_bake (recognised by its "HTFX0001" magic).
 *                max_batch: number of independent tracker slots (frames processed per call).
 *                model_path == NULL gives a context without a hand model that only serves the CNN entry points (ht_cnn_*): what a stand-alone
 *                CNN object is in the reference (CNN PoseInitializerCNN(std::string), handtrack.h:103-130); tracker calls on it return HT_ERR_STATE.
 * ht_destroy     replaces  HandTracker::~HandTracker() (handtrack.h:841-844). */
int ht_create(const char *model_path, int max_batch, int device, ht_ctx **out);
int ht_destroy(ht_ctx *ctx);
/* ht_model_bake  replaces  PhysModel::PhysModel(const char *jsonfile) (physmodel.h:444-475) and, with flags & 1,
 *                LoadHandModel()'s post-processing (handtrack.h:350-358).  Host only (no device needed): writes the built
 *                model as a named-array container that ht_create also accepts. */
int ht_model_bake(const char *json_path, const char *out_path, int flags);
const char *ht_last_error(const ht_ctx *ctx);
int ht_get_params(const ht_ctx *ctx, ht_params *p);
int ht_set_params(ht_ctx *ctx, const ht_params *p);                    /* replaces HandTracker::load_config / visit_fields (handtrack.h:549-581, 822-828) */
int ht_model_info(const ht_ctx *ctx, int *n_bodies, int *n_joints, int *max_batch);
/* ht_scale        replaces  float HandTracker::scale(float s) (handtrack.h:591): PhysModel::scale (physmodel.h:196-219,304-319) on both models of
 *                every tracker slot; the caller multiplies its segment_scale (the compatibility header does). */
int ht_scale(ht_ctx *ctx, float s);
/* ht_config_read  replaces  HandTracker::load_config(const std::string &jsonfile) (handtrack.h:822-828): host only.  Applies the file to
 *                *params the way the reference's field decoder does: every field of visit_fields (handtrack.h:549-581) is assigned, one that
 *                the file does not give as a number becomes 0; a missing file leaves everything untouched.  segment_scale and
 *                prev_frame_error (optional) are the two listed fields that live outside ht_params. */
int ht_config_read(const char *jsonfile, ht_params *params, float *segment_scale, float *prev_frame_error);

/* ---- CNN ---------------------------------------------------------------------------------------------------------
 * ht_cnn_load_weights  replaces  CNN::loadb(std::istream&) (cnn.h:590) / PoseInitializerCNN (handtrack.h:103-130): n must be HT_CNNB_COUNT.
 * ht_cnn_eval          replaces  std::vector<float> CNN::Eval(const std::vector<float>&) (cnn.h:550-556) for B inputs at once:
 *                      in [B][4096] -> out [B][2304]. */
int ht_cnn_load_weights(ht_ctx *ctx, const float *weights, size_t n);
int ht_cnn_eval(ht_ctx *ctx, const float *in, float *out, int B);
int ht_cnn_eval_dev(ht_ctx *ctx, const float *d_in, float *d_out, int B, void *stream);
/* ht_cnn_*_sized       the same three calls for a CNN object built from the reference's layer classes (cnn.h:136-511: LConv, LActivation<TanH>, LMaxPool,
 *                      LFull, LSoftMaxChunked) in PoseInitializerCNN's order but on a `side` x `side` input: side = 64 is the net above; side = 128 is
 *                      BASELINE configs[4]'s input size (conv5 -> 124, pool -> 62 -> 31, conv4 -> 28, pool -> 14, FC 12544 -> 2048 -> 2304; weights
 *                      HT_CNNB128_COUNT values in the same .cnnb order, in [B][16384] -> out [B][2304]).  The two nets of a context are independent. */
int ht_cnn_load_weights_sized(ht_ctx *ctx, int side, const float *weights, size_t n);
int ht_cnn_eval_sized(ht_ctx *ctx, int side, const float *in, float *out, int B);
int ht_cnn_eval_sized_dev(ht_ctx *ctx, int side, const float *d_in, float *d_out, int B, void *stream);

/* ---- tracker -----------------------------------------------------------------------------------------------------
 * ht_tracker_reset    replaces  handmodel.SetPose(p) + othermodel.SetPose(p) with momenta, prev_frame_error and initializing
 *                     cleared (physmodel.h:435; what synthetic-tracker does implicitly at start).  poses [n][nb][7].
 * ht_get_state / ht_set_state   read / write PhysModel rigid-body state (which: 0 handmodel, 1 othermodel), [n][nb][13].
 * ht_update_sync      replaces  std::vector<Pose> HandTracker::update(Image<unsigned short>) (handtrack.h:748-785) for B independent
 *                     trackers, with the background CNN job of :755-768 run synchronously every frame (= update_cnn_model, :734-741,
 *                     then mainthreadpasses passes of :769-780).  depth [B][64*64] (64x64 tiles: HandSegmentVR is the identity,
 *                     :283-284), cams [B][12], poses_out [B][nb][7] = GetPoseUser() (physmodel.h:434).
 *                     cnn_out (optional, [B][2304]) receives HandTracker::cnn_output.
 * ht_update_dev       the same on device buffers, asynchronous on `stream`; d_start_poses (optional, [B][nb][7]) re-seeds every
 *                     tracker slot before the update (independent-frame batches, BASELINE config 3).  The call only enqueues (kernels, two
 *                     side streams forked off `stream` and joined back into it, one 8-byte copy to pinned host memory): it can be captured
 *                     into a HIP graph and replayed (tests/test_gpu_graph.py).  d_depth and d_poses_out may be PINNED HOST memory (hipHostMalloc: one
 *                     address for host and device): the input transform then reads the frames over the host link and the last solve writes the poses there --
 *                     for frames that arrive in host memory this beats uploads on a copy stream beside the step (bench.py host_io: 0.94 against 0.80 of the
 *                     device-resident rate at 1024 frames; tests/test_gpu_zero_copy.py).  d_cams is read by many kernels: keep it in device memory. */
int ht_tracker_reset(ht_ctx *ctx, int first, int n, const float *poses);
int ht_get_state(ht_ctx *ctx, int which, int first, int n, float *state);
int ht_set_state(ht_ctx *ctx, int which, int first, int n, const float *state);
int ht_get_tracker_flags(ht_ctx *ctx, int first, int n, float *prev_frame_error, int *initializing);     /* HandTracker::prev_frame_error / initializing (handtrack.h:546-547) */
int ht_set_tracker_flags(ht_ctx *ctx, int first, int n, const float *prev_frame_error, const int *initializing);
int ht_update_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int B, float *poses_out, float *cnn_out);
int ht_update_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, const float *d_start_poses, int B, float *d_poses_out, void *stream);
/* ht_update_frames_*  the same call on frames that are not 64x64 (w, h multiples of 4, at most 320x240), as HandTracker::update treats them
 *                     (handtrack.h:693-785): HandSegmentVR(dimage, 0xF, {0.1, drangey}, segment_scale) feeds the CNN (:697-701), the point cloud and
 *                     FitError come from the full frame with its own camera (:703-704, :751), and segment.cam.pose goes to PoseFromScratch,
 *                     UnibodyFit and MultiStepSim (:708-713).  depth [B][h*w], cams [B][12] = the frames' cameras.  Any number of in-range points is
 *                     taken, as handtrack.h:751 does: a call whose frames can carry more points than the context holds (w*h / subsample_fraction)
 *                     first grows the per-point arrays (ht_reserve_points; one synchronising re-allocation, 164 bytes per point and frame), so
 *                     ht_frames_overflow (frames of the last call whose cloud was cut) stays 0.
 * ht_reserve_points   grows the per-point arrays ahead of time to `points` per frame (<= HT_POINTS_LIMIT; never shrinks); ht_point_capacity reads it. */
int ht_update_frames_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B, float *poses_out, float *cnn_out);
int ht_update_frames_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int w, int h, float segment_scale, const float *d_start_poses, int B, float *d_poses_out, void *stream);
/* ht_update_direct_*  BASELINE configs[4] end to end (SURVEY 8d "config 5 (i)-(iii)"): the same call on side x side frames (side = 128) that are their own segment
 *                     -- what HandSegmentVR returns for a frame of the net's own size (handtrack.h:283-284) -- evaluated by the net of that input size
 *                     (ht_cnn_load_weights_sized; the reference's layer classes cnn.h:136-511 in the order of handtrack.h:108-118), decoded with
 *                     CNNOutputAnalysis(out, camsub(cam, side / 16)) (handtrack.h:218-241), then FitError, the reset branch, MultiStepSim, the accept step and
 *                     the main-thread passes exactly as handtrack.h:703-785.  The reference has no call of its own for this (PoseInitializerCNN is fixed at
 *                     64x64); oracle/ref_harness.cpp `e2e128` drives the reference's stage functions in that order.  side = 64 is ht_update_*.  d_depth of ht_update_direct_dev must be 16-byte aligned (HT_ERR_ARG otherwise).
 */
int ht_update_direct_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int side, int B, float *poses_out, float *cnn_out);
int ht_update_direct_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int side, const float *d_start_poses, int B, float *d_poses_out, void *stream);
int ht_frames_overflow(ht_ctx *ctx, int *frames_over);
int ht_reserve_points(ht_ctx *ctx, int points);
int ht_point_capacity(ht_ctx *ctx, int *points);
/* ht_update_cnn_model_sync  replaces  std::vector<Pose> HandTracker::update_cnn_model(Image<unsigned short>) (handtrack.h:734-741) and, with
 *                     apply_to_handmodel != 0, void HandTracker::kickstart(Image<unsigned short>) (:743-746) for B trackers: the CNN job alone --
 *                     othermodel is NOT re-seeded from handmodel first, no main-thread passes, no "initializing = 50" rule.  poses_out [B][nb][7] =
 *                     othermodel.GetPose() (physmodel.h:433, body poses, not rig space), accepted_out [B] = 1 where the reference returns that pose and
 *                     0 where it returns an empty vector (:720-722); kickstart copies the accepted poses into handmodel.  Frames w x h as
 *                     ht_update_frames_sync (64x64 tiles: segment_scale unused).  cnn_out optional.
 * ht_get_cnn_results  replaces  the members HandTracker::cnn_input, cnn_output and cnn_output_analysis (handtrack.h:583-585) after an update call:
 *                     cnn_input [n][4096], cnn_output [n][2304], analysis [n][HT_ANALYSIS] (layout: ht_stage_decode); any may be NULL. */
int ht_update_cnn_model_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B, int apply_to_handmodel,
                             float *poses_out, int *accepted_out, float *cnn_out);
int ht_get_cnn_results(ht_ctx *ctx, int first, int n, float *cnn_input, float *cnn_output, float *analysis);
/* The reference's OVERLAPPED update (handtrack.h:748-785: the CNN job on a background thread, collected by a later call) on two contexts of one device, `job` for the CNN job
 * and `main` for the caller's part.  The synchronous ht_update_* calls stay the definition parity is stated on (the reference's update() is timing dependent).
 * ht_job_start          replaces :757-758: othermodel.SetPose(handmodel.GetPose()); std::async(update_cnn_model_threadsafe(dimage)).  main's handmodel, prev_frame_error and
 *                       initializing are copied to `job`, the frame to pinned staging, the job is enqueued on job's stream; does not wait for it.
 * ht_job_poll           replaces pose_estimator.wait_for(1 ms) == ready (:760): *ready = 1 when the job has finished (hipEventQuery).  ht_job_wait blocks until it has.
 * ht_job_collect        replaces :761-768: handmodel.SetPose(results.pose) where the job accepted its pose (accepted_out [B], may be NULL), prev_frame_error as the job left it,
 *                       the job's initializing = max(initializing - 1, 0) applied to main's current value.  The job's cnn_input / cnn_output / analysis: ht_get_cnn_results(job).
 * ht_update_passes_sync replaces the caller's part of update() (:751-753, 769-785): the frame's cloud, mainthreadpasses x (HandModelEnhancements, cloud_chamber, FitPointCloud)
 *                       on handmodel, the "initializing = 50" rule, handmodel.GetPoseUser() into poses_out [B][nb][7].  Frames w x h as ht_update_frames_sync. */
int ht_update_passes_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, int B, float *poses_out);
int ht_job_start(ht_ctx *job, ht_ctx *main, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B);
int ht_job_poll(ht_ctx *job, int *ready);
int ht_job_wait(ht_ctx *job);
int ht_job_collect(ht_ctx *job, ht_ctx *main, int B, int *accepted_out);
/* ht_get_cnn_layers   the intermediate layers of the latest CNN evaluation of slots [first, first + n), for per-layer parity tests (the reference returns every
 *                     layer's output from its forward() calls, cnn.h:550-556): act1 [n][3600] after conv 5x5 + tanh + two max-pools (layer 3 of the list,
 *                     handtrack.h:108-111), act2 [n][2304] after conv 4x4 + tanh + pool (layer 6), act3 [n][2048] after the first fully connected layer + tanh
 *                     (layer 8), logits [n][2304] after the second one (layer 9, before the chunked soft-max).  Any pointer may be NULL. */
int ht_get_cnn_layers(ht_ctx *ctx, int first, int n, float *act1, float *act2, float *act3, float *logits);
/* ht_capacity_events  how often, since ht_create, a kernel hit a capacity the reference does not have: expanding-polytope runs cut short (gjk.h:417 /
 *                     hull.h:233-310 loop without bound; here at most 128 iterations, 96 vertices, 192 triangles); touching samples beyond the 192 the contact
 *                     kernel's per-frame pool holds, or beyond 40 five-sample patches (physics.h:451-462 keeps them all; every sample the pool holds IS kept and
 *                     solved -- 192 contacts per frame and launch since round 5, it was 96); solves in which a model's angular rows exceeded what the solver keeps.
 *                     All are 0 unless a scene or a model is out of the ordinary. */
int ht_capacity_events(ht_ctx *ctx, int *epa_cut_short, int *contacts_dropped, int *angular_rows_over);
/* The most touching samples / five-sample patches a frame of THIS model can produce (every colliding, non-ignoring pair once; five samples where neither body is smaller than
 * ContactPatch's 0.05 m proximity test, gjk.h:637) beside what the contact kernel's per-frame pool holds: samples_bound <= pool and patches_bound <= patch_slots proves that the
 * kernel keeps every contact the reference's unbounded list (physics.h:451-462) would hold -- true for the stock 17-bone hand (91 <= 192, 0 <= 40). */
int ht_contact_capacity(ht_ctx *ctx, int *samples_bound, int *patches_bound, int *pool, int *patch_slots);

/* ---- training ----------------------------------------------------------------------------------------------------------
 * ht_cnn_train        replaces  float CNN::Train(const std::vector<float> &x, const std::vector<float> &t, float alpha) (cnn.h:558-580) called for
 *                     n samples in sequence (batch-1 SGD as train-cnn.cpp:156-162 does with alpha = 0.001): inputs [n][4096], targets
 *                     [n][2304], mse_out [n] (optional) = the value Train returns for each sample.  The context's weights are updated in place.
 * ht_cnn_get_weights  replaces  CNN::saveb (cnn.h:591-593): the weights in .cnnb order.
 * ht_expected_cnn     replaces  GatherHandExpectedCNN(pose, hcam).cnn_expected (handtrack.h:160-173), host only: pose [17][7] and the TILE
 *                     camera [12] (the heat-map camera camsub(cam, 4) is formed inside) -> expected [2304]. */
int ht_cnn_train(ht_ctx *ctx, const float *inputs, const float *targets, int n, float alpha, float *mse_out);
int ht_cnn_get_weights(ht_ctx *ctx, float *w, size_t n);
int ht_expected_cnn(const float *pose, const float *cam, float *expected);
int ht_expected_cnn_full(const float *pose, const float *cam, float *expected, float *image_points /* [8][2], may be NULL */, float *vals /* [16], may be NULL */);

/* ---- a hand model that is not being tracked (host only, no device needed) ---------------------------------------------------
 * The reference's applications keep a second PhysModel to pose, draw and ray-cast their synthetic input
 * (`PhysModel fakehand = LoadHandModel();` synthetic-tracker.cpp:94, FakeDepth :69-76).
 * ht_model_open      replaces  PhysModel::PhysModel(const char *jsonfile) (physmodel.h:444-475) and, with hand_tweaks != 0, LoadHandModel()
 *                    (handtrack.h:347-366); a model baked by ht_model_bake is accepted too.  *out is set even on failure (ht_model_error), close it.
 * ht_model_body      per body: vertex / hull-triangle / plane counts, centre of mass in rig space (RigidBody::com) and the rest pose.
 * ht_model_body_mesh the body's collision vertices [nverts][3] (centre-of-mass frame) and hull triangles [ntris][3]: what PhysModel::GetMeshes()
 *                    hands to a renderer (physmodel.h:295-303), posed by the caller.
 * ht_model_hitcheck  replaces  PhysModel::HitCheck(v0, v1) (physmodel.h:287-294) for bodies at poses [nb][7]: nearest hit of the segment with the
 *                    bodies' hulls; *body = -1 and impact = v1 when nothing is hit. */
typedef struct ht_model ht_model;
int ht_model_open(const char *path, int hand_tweaks, ht_model **out);
int ht_model_close(ht_model *m);
const char *ht_model_error(const ht_model *m);
int ht_model_counts(const ht_model *m, int *n_bodies, int *n_joints);
int ht_model_body(const ht_model *m, int body, int *nverts, int *ntris, int *nplanes, float *com3, float *rest_pose7);
int ht_model_body_mesh(const ht_model *m, int body, float *verts, int *tris);
int ht_model_body_sdmesh(const ht_model *m, int body, int *nverts, float *verts);      /* PhysModel::sdmeshes (physmodel.h:258): the twice-subdivided control cage flat-shaded, three corner positions per triangle in the bone's rig frame; verts NULL: the count only */
int ht_model_hitcheck(const ht_model *m, const float *poses, const float *v0, const float *v1, float *impact3, float *normal3, int *body);

/* ---- segmentation: the step before the tracker for full-size frames --------------------------------------------------
 * ht_segment_vr       replaces  Image<unsigned short> HandSegmentVR(const Image<unsigned short> &depth, int entry_options = 0xF,
 *                     float2 wrange = {0.1f, 0.65f}, float diam = 0.17f) (handtrack.h:280-344) for B frames of w x h pixels:
 *                     depth [B][h*w], cams [B][12] -> tiles [B][64*64] and the segment cameras cams_out [B][12] (focal, principal
 *                     (32,32), depth_scale, pose = (0, rotation)), ready for ht_update_sync.  A 64x64 frame is passed through.
 *                     Frames up to 320x240 (the reference's camera), dimensions multiples of 4.
 * ht_segment_vr_dev   the same on device buffers, asynchronous on `stream`. */
int ht_segment_vr(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, int B, int entry_options, float wrange_lo, float wrange_hi, float diam,
                  uint16_t *tiles, float *cams_out);
int ht_segment_vr_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int w, int h, int B, int entry_options, float wrange_lo, float wrange_hi, float diam,
                      uint16_t *d_tiles, float *d_cams_out, void *stream);

/* ---- annotation fit loop ---------------------------------------------------------------------------------------------
 * ht_slowfit          replaces  void HandTracker::slowfit(const std::vector<float3> &points, int hold, const std::vector<Pose> &refpose,
 *                     int steps_ = 6, RigidBody *selectrb = NULL, const float3 &spoint, const float3 &rbpoint, const std::vector<float4> &crays)
 *                     (handtrack.h:786-821) on the handmodel of slots [0,B), with the points ht_stage_prepare left on the device.
 *                     refpose [B][nb][7] (NULL or hold = 0: no RelativeAngularConstraints), select_rb < 0: no nailed bone,
 *                     crays [B][8][4] (first ncray used for the reference's 8 model feature points). */
int ht_set_points(ht_ctx *ctx, int B, const float *points, int cap, const int *npoints);      /* caller-supplied clouds [B][cap][3] instead of ht_stage_prepare's */
int ht_slowfit(ht_ctx *ctx, int B, int hold, const float *refpose, int steps, int select_rb, const float *spoint, const float *rbpoint, const float *crays, int ncray);

/* ---- stage entry points (same kernels, exposed one reference function at a time so parity tests can pin each) -----
 * All take HOST buffers and are synchronous.  `which` selects the model (0 handmodel, 1 othermodel) of slots [0,B).
 * ht_stage_prepare    depth -> CNN input (handtrack.h:700) and sub-sampled point cloud (misc_image.h:409-417, physmodel.h:58-64):
 *                     cnn_in [B][4096] (optional), points [B][HT_MAX_POINTS][4] (optional), npoints [B] (optional).
 * ht_stage_decode     CNNOutputAnalysis (handtrack.h:218-241): cnn_out [B][2304] -> analysis [B][HT_ANALYSIS]:
 *                     crays 8x4 | image_points 8x2 | confidence 8 | vals 16 | wristroll pitch tilt | palmq 4 | finger_clenched 5.
 * ht_stage_fit_error  FitError (handtrack.h:371-399) of model `which` against the prepared points / depth: err [B].
 * ht_stage_cloud_rows CloudConstraints (physmodel.h:164-181) of model `which`, every `stride`-th prepared point, ray origin =
 *                     camera position if use_cam_origin else 0: rows [B][HT_MAX_POINTS][16] (layout: rb0 rb1 position0 position1 normal
 *                     targetdist targetspeednobias forcelimit.xy friction_master), nrows [B].
 * ht_stage_chamber    cloud_chamber (physmodel.h:486-496) with the five directions and the force limit of handtrack.h:774-778, of model `which` against the
 *                     prepared points: rows [B][5*nb][16], nrows [B] (0 unless boundary_planes is set and the frame has more than min_point_num points).
 * ht_stage_contacts   FindShapeShapeContacts (physics.h:451-462): contacts [B][cap][12] (rb0 rb1 normal p0w p1w separation), n [B].
 * ht_stage_fit        one PhysModel::FitPointCloud(points, chamber-rows-if-enabled, HandModelEnhancements) pass on handmodel
 *                     exactly as HandTracker::update runs it (handtrack.h:769-780).
 * ht_stage_multistep  HandTracker::MultiStepSim on othermodel with the given analysis (handtrack.h:642-690). */
int ht_stage_prepare(ht_ctx *ctx, const uint16_t *depth, const float *cams, int B, float *cnn_in, float *points, int *npoints);
int ht_stage_decode(ht_ctx *ctx, const float *cnn_out, const float *cams, int B, float *analysis);
int ht_stage_fit_error(ht_ctx *ctx, int which, int B, float *err);
int ht_stage_cloud_rows(ht_ctx *ctx, int which, int stride, int use_cam_origin, int B, float *rows, int *nrows);
int ht_stage_chamber(ht_ctx *ctx, int which, int B, float *rows, int *nrows);
int ht_stage_contacts(ht_ctx *ctx, int which, int B, int cap, float *contacts, int *ncontacts);
int ht_stage_fit(ht_ctx *ctx, int B);
int ht_stage_multistep(ht_ctx *ctx, const float *analysis, int B);
int ht_stage_multistep_range(ht_ctx *ctx, const float *analysis, int B, int from_step, int to_step);      /* steps [from_step, to_step) of MultiStepSim alone (handtrack.h:660-688) from othermodel's current state: single-step, teacher-forced comparisons */
int ht_stage_scratch_unibody(ht_ctx *ctx, const float *analysis, int B, int n_unibody);

/* ---- caller-built constraint rows ----------------------------------------------------------------------------------------
 * Row layouts: a linear row (LimitLinear, physics.h:267-308) is 16 floats: rb0 rb1 position0[3] position1[3] normal[3] targetdist targetspeednobias
 * forcelimit.x forcelimit.y friction_master (the layout ht_stage_cloud_rows returns); an angular row (LimitAngular, physics.h:239-265) is 8 floats:
 * rb0 rb1 axis[3] targetspin mintorque maxtorque.  A body is its index in PhysModel::rigidbodies, -1 = NULL.  Arrays are [B][cap][16] / [B][cap][8]
 * with per-frame counts; `which` selects the model (0 handmodel, 1 othermodel) of slots [0,B).  Host buffers, synchronous.
 * ht_fit_rows        replaces  void PhysModel::FitPointCloud(const std::vector<float3> &points, std::vector<LimitLinear> linears = {},
 *                    std::vector<LimitAngular> angulars = {}, float microforce = 1.0f) (physmodel.h:345-356): the caller's linear rows, then
 *                    CloudConstraints(points) limited to +-microforce (x physics_weak_force on bodies 0-2), then the joints' nailed rows; the caller's
 *                    angular rows, then the joints' range rows; PhysicsUpdate with collision rows; SanityCheck.  points [B][pcap][3].  As at every call
 *                    site of the reference the joint ranges are first brought up to date from the pose (HandModelEnhancements, handtrack.h:417-420,
 *                    434-440).  The caller's linear rows must act on one body from the world (rb0 == -1), as those of every such call site do
 *                    (landmark rays handtrack.h:672-673, boundary planes :774-778, the annotator's nail :803-810); HT_ERR_ARG otherwise.
 * ht_physics_update  replaces  void PhysicsUpdate(const std::vector<RigidBody*> &rigidbodies, std::vector<LimitLinear> &Linears,
 *                    std::vector<LimitAngular> &Angulars, const std::vector<std::vector<float3>*> &wgeom) (physics.h:543-587) with an empty wgeom: the
 *                    caller's rows are all there is (plus the collision rows when physics_use_collision is set), any mix of one- and two-body rows in
 *                    the caller's order; a friction row (friction_master -1 / -2) must directly follow its master row, as ConstrainContacts emits
 *                    them (physics.h:463-489).  Capacity: 32 groups of up to three consecutive two-body rows on one body pair, 126 angular rows. */
int ht_fit_rows(ht_ctx *ctx, int which, int B, const float *points, int pcap, const int *npoints, const float *linears, int lcap, const int *nlinears,
                const float *angulars, int acap, const int *nangulars, float microforce);
int ht_physics_update(ht_ctx *ctx, int which, int B, const float *linears, int lcap, const int *nlinears, const float *angulars, int acap, const int *nangulars);

/* ---- multi-GPU: the result gather -------------------------------------------------------------------------------------------
 * Frames are independent, so a host shards a batch one GPU (one process, one context) per contiguous block of frames and nothing crosses GPUs but the
 * results (BASELINE north star: "RCCL over xGMI for the result gather only"; the reference is a single-process CPU program and has no counterpart).
 * ht_comm_unique_id   rank 0 makes the 128-byte RCCL id (ncclGetUniqueId); the host program hands it to the other ranks (MPI, a socket, a file ...).
 * ht_comm_init        every rank joins with it (ncclCommInitRank on the context's device).  RCCL is loaded (dlopen) by these two calls only.
 * ht_comm_info        what the communicator itself reports: ranks and this rank's number (ncclCommCount / ncclCommUserRank).
 * ht_gather_poses_dev one ncclAllGather of d_local [frames][nb][7] into d_all [world][frames][nb][7] (device pointers), on the context's communication
 *                     stream behind everything enqueued on `stream` so far: the next step's kernels do not wait for it.  slot (0 / 1) names which of the
 *                     caller's two buffer pairs the call uses; ht_gather_wait(ctx, slot, stream) makes `stream` (NULL: the context's own stream, as everywhere)
 *                     wait for that slot's gather before the pair is reused or read, ht_gather_wait_host(ctx, slot) the calling thread.  A gather issued
 *                     into a slot whose previous gather was never waited for is refused (HT_ERR_STATE): the wait has to stand in front of the work that refills
 *                     the slot's buffers, nothing issued afterwards can order that.  ht_gather_wait_host synchronises the thread also after a stream-side wait.
 * ht_comm_available   1 when RCCL can be loaded here.  ncclCommInitRank blocks until every rank has arrived, so the ranks of a job agree on this (a MIN
 *                     over the host program's own channel) BEFORE any of them calls ht_comm_init; a rank without the library then cannot strand the others.
 * ht_comm_destroy     leaves the communicator (ht_destroy does it too). */
int ht_comm_available(void);
int ht_gather_wait_host(ht_ctx *ctx, int slot);
int ht_comm_unique_id(void *id128);
int ht_comm_init(ht_ctx *ctx, int world, int rank, const void *id128);
int ht_comm_info(ht_ctx *ctx, int *world, int *rank);
int ht_gather_poses_dev(ht_ctx *ctx, const float *d_local, float *d_all, int frames, int slot, void *stream);
int ht_gather_wait(ht_ctx *ctx, int slot, void *stream);
int ht_comm_destroy(ht_ctx *ctx);

/* ---- timing hooks for bench.py: HIP-event time (ms) per named phase accumulated since the last reset.
 * on = 1: only the dominant kernel ("solve") is bracketed (negligible perturbation, used inside the timed region);
 * on = 2: every phase is bracketed and the side streams are serialised (phase table). ----- */
int ht_profile_enable(ht_ctx *ctx, int on);
int ht_profile_read(ht_ctx *ctx, int reset, int max_entries, char *names, int name_stride, float *total_ms, int *launches, int *n_entries);
/* Tuning aid, no reference counterpart: with the environment variable HT_DEBUG_SKIP=2048 every k_solve launch accumulates per-frame
 * statistics (launches, cycles in chains / two-body linear / angular rows / all sweeps, steps, longest chain, row counts, prologue parts), 16 floats per frame. */
int ht_debug_solve_stats(ht_ctx *ctx, int B, float *out, int reset);
int ht_debug_contact_stats(ht_ctx *ctx, int B, float *out, int reset);      /* same for k_contacts: 24 floats per frame (5 about its wave's polytope runs, 7 unused, 12 about the frame) */
/* Test aid: pins which build of the solver kernel runs (0 = chosen per launch; 1-3 the LDS sizes for tile batches / small batches / large frames and models;
 * 4 = a build whose LDS arrays hold nothing, so every frame places its row records in HBM; 6 = the build with four angular-row slots per lane that models with more
 * than 18 joints take).  Placement only: results are identical bit for bit.  7 = the launcher's choice, with the two-body rows of every frame applied through the level
 * schedule (the sweeps of round 4) instead of block by block (csrc/ht_block.hpp): the same rows in the same order, another rounding.  8 = the exact-order sweeps of 5 under
 * the product's forked launch sequence (5 runs an update's kernels in order on one stream). */
int ht_debug_solver_build(ht_ctx *ctx, int which);
int ht_debug_reset_flags(ht_ctx *ctx, int *flags, int n);      /* test aid: flags[i] = 1 if tracker slot i took the full-reset branch (handtrack.h:706-711) in the latest update */
/* Test aid: 5 additionally selects the EXACT-ORDER instantiation of the solver -- the same rows swept by the reference's own LimitLinear::Iter / LimitAngular::Iter
 * (physics.h:251-265, 289-307) in the reference's row order, no fused multiply-adds (update entry points only; tests/test_gpu_exact_solver.py).
 * ht_debug_reset_organisation: how the latest update launched the full-reset branch (0 few frames, 1 many frames, -1 none yet). */
int ht_debug_reset_organisation(ht_ctx *ctx, int *many);
/* Test aid: pins the organisation of the contact kernel (0 = chosen per launch: the cooperative kernel whenever the model fits its LDS; 1 cooperative,
 * 2 one lane group per body pair).  Same contacts in the same order either way. */
int ht_debug_contact_kernel(ht_ctx *ctx, int which);
/* Test / measurement aid: 0 (default) = every solve makes its tables (joint groups, angular records, block couplings, chain lists) in k_solve's own one-wave prologue;
 * 1 = k_solve_prep makes them on a side stream beside the contact kernel (csrc/ht_prep.hip; round 6's experiment: k_solve 8 % shorter, the step longer -- profiles/r06_notes.md);
 * 2 = only the pose-only tables (joint groups, angular records, block couplings), on the side stream the cloud rows do not use: also measured slower.
 * Where the tables are made, never what they hold: results are identical bit for bit (tests/test_gpu_solve_tables.py). */
int ht_debug_solve_tables(ht_ctx *ctx, int mode);
int ht_debug_solve_tables_header(ht_ctx *ctx, int B, int *hdr);      /* tuning aid: the 32 header words of every frame's tables as the latest k_solve_prep left them ([B][32]; words 20-27: cycle stamps of a -DHT_TUNING build) */

#ifdef __cplusplus
}
#endif
#endif
