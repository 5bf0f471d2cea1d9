// ht_host.hpp -- host-side context behind the C-ABI (product code).
#pragma once
#include <map>
#include <string>
#include <vector>
#include "ht_device.hpp"

struct ht_comm_state;      // RCCL communicator + communication stream (ht_comm.hip)
struct ht_prof_entry { std::vector<hipEvent_t> ev; size_t used; float total_ms; int launches; };

struct ht_ctx
{
	bool ready = false, have_weights = false, profile = false;
	bool cnn_only = false;          // created without a hand model: only the CNN entry points work
	bool profile_phases = false;     // also time the minor phases (serialises the side streams; used for the phase table, not for the timed region)
	int B = 0, device = 0;
	int solver_build = 0;           // ht_debug_solver_build: 0 = the launcher's choice; 5 = the exact-order instantiation (tests only): the reference's own sweeps, cloud rows in the reference's layout
	float *d_exact_lin = nullptr, *d_exact_ang = nullptr;      // its two-body linear rows [B][512][HT_ROW] and angular rows [B][256][8] (allocated when first asked for)
	int contact_kernel = 0;         // ht_debug_contact_kernel: 0 = the launcher's choice, 1 cooperative, 2 lane-per-pair
	std::string err;
	hipStream_t stream = nullptr;
	hipStream_t last_user_stream = nullptr;         // stream of the latest *_dev call (host-read helpers wait for it too)
	hipStream_t side[2] = { nullptr, nullptr };     // independent kernels of one fit step (cloud rows, contacts, chamber) run side by side
	hipEvent_t ev_fork = nullptr, ev_join[2] = { nullptr, nullptr }, ev_lap = nullptr;
	hipEvent_t ev_job = nullptr, ev_seed = nullptr; bool job_pending = false; void *h_job_in = nullptr; size_t h_job_cap = 0;      // ht_job_start: the CNN job of the overlapped update() in flight on this context, its pinned input staging
	ht_params par;
	ht_physics_dev phys;
	ht_model_dev model;
	ht_cnn_weights cnnw;
	ht_cnn_weights cnnw128; bool have_weights128 = false;         // the 128x128-input variant of the net (BASELINE configs[4]); buffers allocated on first use
	float *d_weights128 = nullptr, *d_in128 = nullptr, *d_act1_128 = nullptr, *d_act2_128 = nullptr;
	std::vector<float> h_bodyc, h_jointc;
	std::vector<float4> h_verts, h_planes;                        // host copies of the model geometry (ht_scale rewrites them)
	float *d_train = nullptr;                                    // training arena: layer outputs, errors, split-K partial sums (allocated on first use)
	float *d_sf_ref = nullptr, *d_sf_crays = nullptr;             // slowfit inputs [B][nb][7], [B][8][4] (allocated on first use)
	float4 *d_cverts_rw = nullptr;                                // padded copy of the collision vertices (ht_model_dev::cverts)
	float4 *d_verts_rw = nullptr, *d_planes_rw = nullptr; float *d_bodyc_rw = nullptr, *d_jointc_rw = nullptr;
	std::vector<void *> allocs;
	std::map<std::string, ht_prof_entry> prof;

	// device buffers (capacity B frames / tracker slots)
	float *d_weights = nullptr;
	uint16_t *d_depth = nullptr;
	uint16_t *d_seg_tiles = nullptr, *d_frames = nullptr; float *d_frame_cams = nullptr, *d_frame_cams_in = nullptr; int *d_overflow = nullptr; size_t frames_cap = 0;      // full-size frame path (ht_update_frames)
	float *d_cams = nullptr, *d_cnn_in = nullptr, *d_act1 = nullptr, *d_act2 = nullptr, *d_act3 = nullptr, *d_logits = nullptr, *d_cnn_out = nullptr, *d_analysis = nullptr;
	float4 *d_pts = nullptr; int *d_npts = nullptr;
	float4 *d_ptsv = nullptr; int *d_nptsv = nullptr;          // the main-thread cloud when subsample_voxel is set (allocated on first use; d_pts stays the CNN job's cloud)
	float *d_state[2] = { nullptr, nullptr };      // [B][nb][HT_STATE_STRIDE]: 0 handmodel, 1 othermodel
	float *d_prev_err = nullptr; int *d_initializing = nullptr;
	float *d_err_old = nullptr, *d_err_new = nullptr; int *d_flags = nullptr, *d_nflags = nullptr;
	int *d_flist = nullptr, *d_nflist = nullptr;      // the flagged frames of an update as a list [B] and its length (k_prepare clears it)
	// How many frames took the full-reset branch so far (device counter, copied to pinned host memory behind every update's reset kernel, never waited for):
	// an update that follows one with more reset frames than the device has CUs launches the reset branch and those frames' first contact step in their
	// many-frames organisation (two reset blocks per CU, four frames per contact block) instead of the few-frames one.  A hint only: both are always correct.
	unsigned *d_nreset = nullptr; volatile unsigned *h_nreset = nullptr;      // [2]: frames that reset, updates that counted them
	unsigned nreset_seen[2] = { 0, 0 }; bool many_reset = false, tail_pending = false; int n_cu = 256;
	int last_reset_many = -1;       // which organisation the latest update launched the reset branch in (ht_debug_reset_organisation): 0 few frames, 1 many
	float *d_rows = nullptr; int *d_nrows = nullptr;            // cloud rows [B][pts_cap][HT_ROW] in the reference's layout (stage calls, UnibodyFit, caller-built rows)
	unsigned char *d_rowbody = nullptr;                          // [B][pts_cap] body of every cloud row whose solver record k_cloud_rows wrote into d_scratch
	float *d_chamber = nullptr; int *d_nchamber = nullptr;      // [B][5*nb][HT_ROW]
	float *d_tables = nullptr;                                   // [B][TB_WORDS] the solve tables k_solve_prep makes and k_solve reads (ht_solve_shared.hpp)
	float *d_chplanes = nullptr; int *d_chon = nullptr; bool planes_valid = false;      // [B][5][4], [B]: the boundary planes of the update's main-thread cloud (k_chamber_planes: once per update); valid inside the update that made them
	int solve_tables = 0;                                       // 1 (ht_debug_solve_tables; tools/exp_tables.sh): k_solve_prep makes every solve's tables beside the contact kernel.  Measured slower in round 6 (profiles/r06_notes.md section 1): off
	int *d_accepted = nullptr;
	float *d_contacts = nullptr; int *d_ncontacts = nullptr;    // [B][HT_MAXCONTACT][HT_CONTACT]
	int *d_cwork = nullptr, *d_corder = nullptr;                  // [HT_CONTACT_SLOTS][cstride]: what every frame cost in every contact launch of the latest update; the assignment of frames to blocks made from it for the current one (ht_gjk.hip: k_contact_order)
	int *d_porder = nullptr;      // [B]: the frames by their point counts (ht_model_dev::frame_order inside an update of a batch of several rounds per CU)
	int *d_swork = nullptr, *d_sorder = nullptr; unsigned swork_mask = 0, sorder_mask = 0; int swork_B = 0;      // the same for the solves of a batch that takes several rounds per CU: what every frame's solve took, the launch order (longest first)
	int cstride = 0, cwork_B = 0; unsigned cwork_mask = 0, corder_mask = 0;      // slots whose work the latest update wrote (for cwork_B frames) / whose order the current update may use
	unsigned char *d_epa_ws = nullptr;                           // expanding-polytope workspace, one per (frame, wave)
	float *d_scratch = nullptr;                                  // solver row records [B][pts_cap + 5*nb + 32][20] (ht_quad.hpp)
	float *d_poses_out = nullptr, *d_start = nullptr;
	float *d_stage = nullptr;                                    // staging for host<->device state copies
	// caller-built constraint rows (ht_fit_rows / ht_physics_update), allocated on first use and grown to the largest call
	float *d_user_lin = nullptr; unsigned short *d_user_pos = nullptr; float *d_user_ang = nullptr; int *d_user_n = nullptr;      // [B][lin_cap][HT_ROW], [B][lin_cap], [B][ang_cap][HT_AROW], [4][B]
	int user_lin_cap = 0, user_ang_cap = 0;
	ht_comm_state *comm = nullptr;                               // multi-GPU pose gather (ht_comm_init), null on a single-GPU host
};

struct ht_prof_scope
{
	ht_ctx *ctx; ht_prof_entry *ent; hipStream_t stream; size_t slot;
	ht_prof_scope(ht_ctx *c, const char *name, hipStream_t s, bool minor_phase = false);
	~ht_prof_scope();
};

// rows of a frame's slot of the solver scratch: every point and chamber row, k_solve's read-ahead slack, the rows that pad a host body's chain
// to a multiple of 8 when a body beyond the 16th rides on its quad (7 per such body at most), and the tail that takes what does not fit k_solve's LDS
static inline size_t ht_scratch_rows(size_t pts_cap, size_t nb) { return pts_cap + 5 * nb + 32 + 7 * 16 + HT_SCRATCH_TAIL; }
int ht_alloc_buffers(ht_ctx *ctx);
int ht_alloc_solve_tables(ht_ctx *ctx);      // d_tables, d_chplanes, d_chon (first use of the solve-tables path)
int ht_reserve_points_locked(ht_ctx *ctx, int points);      // grows the per-point arrays (ht_api.hip); waits for the context's streams
// *_dev entry points: a NULL stream means the context's own stream (never the legacy default stream); the choice is remembered so that the
// host-read helpers (ht_capacity_events, ht_frames_overflow, ht_get_tracker_flags, ...) can wait for work enqueued on a caller's stream
static inline hipStream_t ht_user_stream(ht_ctx *ctx, void *stream) { hipStream_t s = stream ? (hipStream_t)stream : ctx->stream; ctx->last_user_stream = s; return s; }
static inline hipError_t ht_sync_all(ht_ctx *ctx)
{
	hipError_t e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess && ctx->last_user_stream && ctx->last_user_stream != ctx->stream) e = hipStreamSynchronize(ctx->last_user_stream);
	return e;
}
