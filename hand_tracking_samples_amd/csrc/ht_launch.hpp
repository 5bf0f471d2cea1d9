// ht_launch.hpp -- host-callable launchers of the solver-side kernels (product code).
#pragma once
#include "ht_device.hpp"

struct solve_args
{
	const float *rows_pre; const int *n_pre; int pre_stride;      // chamber rows [B][pre_stride][HT_ROW] (may be null)
	const float *rows_cloud; const int *n_cloud;                    // cloud rows [B][pts_cap][HT_ROW] (may be null)
	const unsigned char *cloud_body;                                // [B][pts_cap]: k_cloud_rows has written the cloud rows' RECORDS into the frames' scratch slots (rows_cloud unused) and their bodies here
	const float *contacts; const int *ncontacts;                    // [B][HT_MAXCONTACT][HT_CONTACT] (may be null)
	// caller-built rows (ht_fit_rows / ht_physics_update = PhysModel::FitPointCloud's and PhysicsUpdate's row arguments; all may be null):
	const float *ang_user; const int *n_ang_user; int ang_user_stride;        // angular rows [B][stride][HT_AROW]: first in the angular list
	const float *lin_tail; const int *n_lin_tail; int lin_tail_stride;        // linear rows from the caller's first two-body row on [B][stride][HT_ROW]: ahead of the joint rows
	const unsigned short *lin_tail_pos; const int *n_tail_groups;             // their placement, made by the host: group << 2 | slot, bit 15 = the row's group is a contact triple
	int no_model_rows;                                                        // PhysicsUpdate: the caller's rows are all there is (no joint rows, no HandModelEnhancements)
	const float *analysis; const float *cams;                       // for ApplyAngles / landmark-ray rows / arm cone
	const int *active_flag;                                         // optional per-frame enable
	float *state;                                                   // [B][nb][HT_STATE_STRIDE] of the model being solved
	float *scratch; int scratch_stride;                             // [B][scratch_stride][HT_CREC] pre-computed single-body row records (ht_quad.hpp)
	int batch;                                                      // frames of the launch (the sums of over-size frames live behind all frames' records)
	int apply_angles; float drive_force; int ray_rows; int arm_cone; int zero_momenta; int steps_keyangles; float min_cray_prob;
	// slowfit (handtrack.h:786-821): landmark rays from the origin (sf_crays [B][8][4], first sf_ncray used), a bone nailed to a point, and
	// RelativeAngularConstraints against a reference pose (sf_refpose [B][nb][7], sf_hold = 1 or 2); all off when zero / null
	const float *sf_crays; int sf_ncray; int sf_select; float sf_spoint[3], sf_rbpoint[3]; const float *sf_refpose; int sf_hold;
	int *caps;                                                      // capacity counter: frames x launches whose angular rows exceeded the LDS records (may be null)
	int shared_gpu;                                                 // other kernels run beside this launch (the reset path): keep the small LDS footprint
	const int *frame_order; int *cost_out;                          // batches of several rounds per CU: block i takes frame frame_order[i] (the frames by what they took in the same launch of the previous update, longest first, so that the launch ends on short ones); cost_out [B]: what each took now.  Both may be null
	int two_body_levels;                                            // tests only (ht_debug_solver_build 7): the two-body rows by the level schedule for every frame, as until round 4 (otherwise only frames the blocked form does not hold)
	int force_build;                                                // 0: the launcher chooses k_solve's build; 1 small, 2 only, 3 mid, 4 tiny (every array in HBM): ht_debug_solver_build
	// the last solve of an update also delivers the poses (GetPoseUser physmodel.h:434 + the "initializing = 50" rule of handtrack.h:781-782), instead of a launch of its own
	float *out_poses; const int *out_npts; int *out_initializing; int out_min_point_num;
	int ang_extra_bound;                                            // angular rows a frame can have beyond the 13 CNN-driven ones and 6 per joint (slowfit: 3 per joint; caller-built rows: their largest count): picks the build
	float *exact_lin, *exact_ang;                                   // force_build 5 (tests only, ht_debug_exact_solver): the two-body linear rows [B][512][HT_ROW] and the angular rows [B][256][8] in the reference's layout, for the reference's own sweeps
	const float *tables;                                            // [B][TB_WORDS] (ht_solve_shared.hpp): k_solve_prep has made this solve's tables; null: k_solve's own prologue
	int dbg;                                                        // timing experiments only (HT_DEBUG_SKIP): 1 skip chains, 2 skip two-body linear, 4 skip angular
};

// k_solve_prep (csrc/ht_prep.hip): the tables of one solve, four waves per frame, on a side stream beside the contact kernel (layout: ht_solve_shared.hpp)
#define TB_WORDS 8928      // words of a frame's tables
struct prep_args
{
	const float *state; const float *analysis; const float *cams; const int *active_flag;
	float *scratch; int scratch_stride; int batch;
	const unsigned char *cloud_body; const int *n_cloud;      // the cloud rows' bodies (k_cloud_rows wrote their records into the scratch slots); null: a solve without cloud rows
	const float *ch_planes; const int *ch_on;                   // boundary planes [B][5][4] of the frames whose ch_on is set (k_chamber_planes); null: a solve without them
	float *rows_pre; int *n_pre; float ch_maxforce;             // their rows in the reference's layout [B][5 * nb][HT_ROW] and count, for a frame that falls back to k_solve's own prologue
	int apply_angles; float drive_force; int ray_rows; int arm_cone; int steps_keyangles; float min_cray_prob;
	int parts;                                                  // which tables this launch makes: 1 the pose-only ones (joints' groups, angular records, the blocks' couplings and edges: waves 0 and 1), 2 the chain tables (waves 2 and 3: lists, dealing, four-row couplings, landmark rays, boundary-plane rows)
	float *tables;                                              // [B][TB_WORDS]
	int dbg;
};
void ht_launch_solve_prep(const ht_model_dev &M, const ht_physics_dev &ph, const prep_args &a, int B, hipStream_t s);

// where k_cloud_rows puts the solver's records of its rows (ht_quad.hpp): the frames' scratch slots [B][stride][HT_CREC], the rows' bodies [B][pts_cap], the time step
struct cloud_records { float *scratch; int stride; unsigned char *body; float dt; };
void ht_launch_cloud_rows(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, const float *cams, const int *active_flag, int stride, int use_cam_origin, int mode,
                          const ht_params &par, float *rows, int *nrows, int B, hipStream_t s, float sf_ratio = 0.0f, float sf_wrist = 0.0f, const cloud_records *rec = nullptr);
// What follows a FitError in HandTracker::update and needs nothing but that frame's error rides on the kernel's last thread instead of a launch of its own:
// mode 1 = the full-reset decision (handtrack.h:706: flags[b] = angles_only || error > threshold), mode 2 = the accept step (handtrack.h:713-731).
struct ht_fit_after
{
	int mode;
	float reset_thr; int angles_only; int *flags, *nflags; int *list, *nlist; unsigned *nreset;      // list / nlist (may be null): the flagged frames, appended in no particular order; nreset (may be null): running counts of the frames flagged and of the decisions taken (one per update)
	float *hand; const float *other; const float *err_old; float *prev_err; int *initializing, *accepted; int nb, min_point_num, always_take_cnn; float accum_thr;
};
void ht_launch_fit_error(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, const uint16_t *depth, const float *cams, int w, int h, float scale, float *err, int B, hipStream_t s, const ht_fit_after *after = nullptr);
void ht_launch_chamber(const ht_model_dev &M, const float *state, const float4 *pts, const int *npts, int min_point_num, int enabled, float maxforce, float *rows, int *nch, int B, hipStream_t s,
                       const float *planes = nullptr, const int *planes_on = nullptr);      // planes [B][5][4] / planes_on [B] of ht_launch_chamber_planes: the scan is not made again
void ht_launch_chamber_planes(const ht_model_dev &M, const float4 *pts, const int *npts, int min_point_num, int enabled, float *planes, int *on, int B, hipStream_t s);      // the five planes alone [B][5][4], on [B]: once per update (k_solve_prep makes their rows pass by pass)
void ht_launch_contacts(const ht_model_dev &M, const float *state, float driftmax, float jiggle_sin, const int *active_flag, void *epa_ws, float *contacts, int *ncontacts, int B, hipStream_t s, bool beside_cloud_rows = false, int force_kernel = 0, int few_frames = 0,
                        const int *order = nullptr, int *work_out = nullptr);      // order / work_out (cooperative kernel only, both may be null): the frame of every (slot, block) as k_contact_order dealt them; where every live frame leaves what it cost
int ht_contacts_frames_per_block(const ht_model_dev &M, int B);
#define HT_CONTACT_SLOTS 16      // unmasked contact launches of an update that keep a work history: MultiStepSim step st -> slot st (< 8), main-thread pass i -> slot 8 + i
void ht_launch_order_by_points(const int *npts, int *order, int B, hipStream_t s);      // order = the frames by their point counts, most first (ht_model_dev::frame_order)
void ht_launch_rank_desc(const int *work, int *order, int B, int stride, unsigned slots, int nslots, hipStream_t s);      // order[slot][.] = the frames of every 4096-frame segment by work[slot][.], largest first
void ht_launch_contact_order(const int *work, int *order, int B, int nfr, int stride, unsigned slots, int nslots, int epb, hipStream_t s);
size_t ht_contacts_workspace_bytes(int B);
void ht_launch_solve(const ht_model_dev &M, const ht_physics_dev &ph, const solve_args &a, int B, hipStream_t s);
void ht_launch_set_pose(float *state, const float *src, int nb, int n, int mode, hipStream_t s);
void ht_launch_get_state(const float *state, float *dst, int nb, int n, hipStream_t s);
void ht_launch_clear_flags(float *prev_err, int *initializing, int n, hipStream_t s);
// the full-reset branch (PoseFromScratch, then n_unibody x UnibodyFit) of the listed frames, one block per frame at a time (k_reset, csrc/ht_cloud.hip)
void ht_launch_reset(const ht_model_dev &M, const ht_physics_dev &ph, float *state, const float4 *pts, const int *npts, const float *analysis, const float *cams, const int *list, const int *nlist,
                     int n_unibody, const ht_params &par, float *rows, int *nrows, float *scratch, int scratch_stride, int batch, int B, hipStream_t s, bool many_frames, int n_cu, bool exact = false);
void ht_launch_output(const ht_model_dev &M, const float *hand, const int *npts, int *initializing, int min_point_num, float *poses, int n, hipStream_t s, int raw = 0);
// ht_segment.hip
bool ht_segment_supported(int w, int h);
void ht_launch_segment(const uint16_t *depth, const float *cams, int w, int h, int entry_options, float wrange_hi, float diam, uint16_t *tiles, float *cams_out, int B, hipStream_t s);
void ht_launch_scale_state(float *state, int nb, int n, float s, hipStream_t st);      // ht_track.hip
// ht_train.hip
void ht_launch_train_step(float *w, float *W2p, const float *x, const float *target, float alpha, float *act, float *err, float *part, float *mse_out, hipStream_t s);
size_t ht_train_act_floats();
size_t ht_train_err_floats();
size_t ht_train_part_floats();
