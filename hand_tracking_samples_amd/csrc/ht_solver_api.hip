// ht_solver_api.hip -- tracker / solver entry points of the C-ABI: launch sequencing of the per-frame path
// (HandTracker::update / update_cnn_model / MultiStepSim / FitPointCloud, include/handtrack.h:642-785).
#include <string.h>
#include <stdlib.h>
#include "ht_device.hpp"
#include "ht_host.hpp"
#include "ht_launch.hpp"

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_); return HT_ERR_HIP; } } while (0)
#define CHECK_READY(ctx) if (!(ctx)) return HT_ERR_ARG; if (!(ctx)->ready) { (ctx)->err = "context not initialised (ht_create failed)"; return HT_ERR_STATE; } ht_device_guard dev_guard_((ctx)->device)
#define CHECK_MODEL(ctx) do { if ((ctx)->cnn_only) { (ctx)->err = "this context was created without a hand model (CNN only)"; return HT_ERR_STATE; } } while (0)
#define CHECK_BATCH(ctx, B) do { if ((B) < 1 || (B) > (ctx)->B) { (ctx)->err = "batch exceeds the capacity given to ht_create"; return HT_ERR_ARG; } } while (0)
#define CHECK_RANGE(ctx, first, n) do { if ((first) < 0 || (n) < 1 || (first) + (n) > (ctx)->B) { (ctx)->err = "slot range exceeds the capacity given to ht_create"; return HT_ERR_ARG; } } while (0)

static int scratch_stride(const ht_ctx *ctx) { return (int)ht_scratch_rows((size_t)ctx->model.pts_cap, (size_t)ctx->model.nb); }

// ---- building blocks ------------------------------------------------------------------------------------------------
static int dev_alloc_points(ht_ctx *ctx)      // the second cloud of a context that uses voxel sub-sampling
{
	void *a = nullptr, *b = nullptr;
	HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * ctx->model.pts_cap * sizeof(float4))); ctx->allocs.push_back(a); ctx->d_ptsv = (float4 *)a;
	if (!ctx->d_nptsv) { HIPCHK(ctx, hipMalloc(&b, (size_t)ctx->B * sizeof(int))); ctx->allocs.push_back(b); ctx->d_nptsv = (int *)b; }
	return HT_OK;
}
static cloud_records cloud_rec(ht_ctx *ctx) { cloud_records r = { ctx->d_scratch, scratch_stride(ctx), ctx->d_rowbody, ctx->phys.deltaT }; return r; }
// the exact-order instantiation of the solver (ht_debug_solver_build 5, tests only) takes the cloud rows in the reference's layout (ctx->d_rows) instead of records
static bool exact_solver(const ht_ctx *ctx) { return ctx->solver_build == 5 || ctx->solver_build == 8; }      // 8: the same sweeps with the product's forked launch sequence (5 takes the kernels in order on one stream)
static const cloud_records *rec_or_rows(const ht_ctx *ctx, const cloud_records *cr) { return exact_solver(ctx) ? nullptr : cr; }
static void exact_args(ht_ctx *ctx, solve_args &a, bool cloud)
{
	a.force_build = exact_solver(ctx) ? 5 : ctx->solver_build;
	a.exact_lin = ctx->d_exact_lin; a.exact_ang = ctx->d_exact_ang;
	if (exact_solver(ctx) && cloud) { a.rows_cloud = ctx->d_rows; a.cloud_body = nullptr; }
}
// Solves of an update that keep a history (the slots of the contact launches: MultiStepSim step st -> st, main pass i -> 8 + i), for batches that take several rounds
// per CU: k_solve notes what every frame took, the next update's launch of the same slot takes the frames longest first (contact_orders ranks them beside the net), so
// that the launch ends on short frames instead of waiting for a long one that started last.  Results do not depend on the order.
static bool solve_history_on(const ht_ctx *ctx, int B) { return ctx->d_swork && B > ctx->n_cu * 8 && B + 8 <= ctx->cstride; }
static void solve_step(ht_ctx *ctx, int which, const float *rows_pre, const int *n_pre, bool cloud, bool contacts, const int *active,
                       int apply_angles, float drive_force, int ray_rows, int arm_cone, int zero_momenta, int B, hipStream_t s, bool shared_gpu = false, float *poses_out = nullptr, const int *out_npts = nullptr,
                       int hist_slot = -1, bool tables = false)
{
	solve_args a;
	memset(&a, 0, sizeof a);
	a.sf_select = -1;
	a.caps = reinterpret_cast<int *>(ctx->d_epa_ws) + 2;
	a.rows_pre = rows_pre; a.n_pre = n_pre; a.pre_stride = 5 * ctx->model.nb;
	a.cloud_body = cloud ? ctx->d_rowbody : nullptr; a.n_cloud = ctx->d_nrows;      // k_cloud_rows wrote the rows' records into the scratch slots (cloud_rec below)
	a.contacts = contacts ? ctx->d_contacts : nullptr; a.ncontacts = ctx->d_ncontacts;
	a.analysis = ctx->d_analysis; a.cams = ctx->d_cams; a.active_flag = active;
	a.state = ctx->d_state[which]; a.scratch = ctx->d_scratch; a.scratch_stride = scratch_stride(ctx); a.batch = ctx->B;
	a.apply_angles = apply_angles; a.drive_force = drive_force; a.ray_rows = ray_rows; a.arm_cone = arm_cone; a.zero_momenta = zero_momenta;
	a.steps_keyangles = ctx->par.steps_keyangles; a.min_cray_prob = ctx->par.min_cray_prob;
	a.dbg = ht_tuning_flags();
	a.shared_gpu = shared_gpu ? 1 : 0;
	a.tables = tables ? ctx->d_tables : nullptr;      // solve_prep below has made them for exactly this solve
	exact_args(ctx, a, cloud);
	a.out_poses = poses_out; a.out_npts = out_npts; a.out_initializing = ctx->d_initializing; a.out_min_point_num = ctx->par.min_point_num;
	if (hist_slot >= 0 && hist_slot < HT_CONTACT_SLOTS && !active && !exact_solver(ctx) && solve_history_on(ctx, B) && B == ctx->swork_B)
	{
		a.cost_out = ctx->d_swork + (size_t)hist_slot * ctx->cstride;
		ctx->swork_mask |= 1u << hist_slot;
		if ((ctx->sorder_mask >> hist_slot) & 1u) a.frame_order = ctx->d_sorder + (size_t)hist_slot * ctx->cstride;
	}
	ht_launch_solve(ctx->model, ctx->phys, a, B, s);
}
// Round 6: the tables of a solve (ht_solve_shared.hpp) are made by k_solve_prep on whichever stream runs the solve's row producers, behind the cloud rows (it lists them per
// body) and beside the contact kernel; the solve_step that follows is told to start from them.  Off for the exact-order builds (their sweeps take the rows as the reference
// lays them out) and under ht_debug_solve_tables(0).
// 0: off; 1: every table (pose-only and chain tables: one launch behind the cloud rows); 2: the pose-only tables alone, on the side stream the cloud rows do not use
static int solve_tables_mode(const ht_ctx *ctx)
{
	static const int env = ht_tuning_int("HT_TABLES", -1);      // timing experiments (-DHT_TUNING builds only): tools/exp_tables.sh
	const int m = env >= 0 ? env : ctx->solve_tables;
	return (m && ctx->d_tables && !exact_solver(ctx)) ? m : 0;
}
static bool solve_tables_on(const ht_ctx *ctx) { return solve_tables_mode(ctx) == 1; }
static void solve_prep(ht_ctx *ctx, int which, bool cloud, bool chamber, const int *active, int apply_angles, float drive_force, int ray_rows, int arm_cone, int B, hipStream_t s, int parts = 3)
{
	prep_args a;
	memset(&a, 0, sizeof a);
	a.state = ctx->d_state[which]; a.analysis = ctx->d_analysis; a.cams = ctx->d_cams; a.active_flag = active;
	a.scratch = ctx->d_scratch; a.scratch_stride = scratch_stride(ctx); a.batch = ctx->B;
	a.cloud_body = cloud ? ctx->d_rowbody : nullptr; a.n_cloud = ctx->d_nrows;
	if (chamber) { a.ch_planes = ctx->d_chplanes; a.ch_on = ctx->d_chon; a.rows_pre = ctx->d_chamber; a.n_pre = ctx->d_nchamber; a.ch_maxforce = 10.0f; }
	a.apply_angles = apply_angles; a.drive_force = drive_force; a.ray_rows = ray_rows; a.arm_cone = arm_cone;
	a.steps_keyangles = ctx->par.steps_keyangles; a.min_cray_prob = ctx->par.min_cray_prob;
	a.tables = ctx->d_tables; a.dbg = ht_tuning_flags(); a.parts = parts;
	ht_launch_solve_prep(ctx->model, ctx->phys, a, B, s);
}
// Fork/join helpers: the row-producing kernels of one fit step only read the pose, so they run side by side on two extra streams
// and the solve waits for all of them (each of them is latency-bound on its slowest frame and leaves most of the chip idle).
// Tuning builds, HT_MARKS=1: events at named points of an update on whichever stream, printed (ms since the first) after the update -- the concurrent streams as
// they really ran (a kernel trace serialises them).
#ifdef HT_TUNING
#include <vector>
#include <string>
static std::vector<std::pair<std::string, hipEvent_t>> g_marks;
static bool marks_on() { static const bool on = getenv("HT_MARKS") != nullptr; return on; }
static void mark(const char *name, hipStream_t s) { if (!marks_on()) return; hipEvent_t e; (void)hipEventCreate(&e); (void)hipEventRecord(e, s); g_marks.emplace_back(name, e); }
static void marks_dump()
{
	if (!marks_on() || g_marks.empty()) return;
	(void)hipDeviceSynchronize();
	for (auto &m : g_marks) { float ms = 0; (void)hipEventElapsedTime(&ms, g_marks[0].second, m.second); fprintf(stderr, "  mark %-28s %8.3f ms\n", m.first.c_str(), ms); }
	for (auto &m : g_marks) (void)hipEventDestroy(m.second);
	g_marks.clear();
}
#else
static inline void mark(const char *, hipStream_t) {}
static inline void marks_dump() {}
#endif
// Contact launches of an update that keep a work history (ht_launch.hpp: HT_CONTACT_SLOTS).  slot < 0 or a launch on the reset frames alone: no history, the fixed assignment.
struct contact_slot { const int *order; int *work; };
static contact_slot contact_history(ht_ctx *ctx, int slot, const int *active, int B)
{
	contact_slot c = { nullptr, nullptr };
	if (slot < 0 || slot >= HT_CONTACT_SLOTS || !ctx->d_cwork || active == ctx->d_flags || B != ctx->cwork_B) return c;
	c.work = ctx->d_cwork + (size_t)slot * ctx->cstride;
	ctx->cwork_mask |= 1u << slot;
	if ((ctx->corder_mask >> slot) & 1u) c.order = ctx->d_corder + (size_t)slot * ctx->cstride;
	return c;
}
// At the head of an update, on a stream that has nothing to do while the net runs: the assignments of this update's contact launches from the works of the last one
static void contact_orders(ht_ctx *ctx, int B, hipStream_t t)
{
	const int nfr = ht_contacts_frames_per_block(ctx->model, B);
	ctx->corder_mask = 0;
	if (ctx->d_cwork && ctx->cwork_B == B && ctx->cwork_mask && nfr > 1 && B + 8 <= ctx->cstride && ctx->contact_kernel != 2)
	{
		// frames with polytope runs per block: every block a CU of its own (the slowest block is the launch's time) -> one; several rounds per CU (the sum counts) -> all.
		// tools/exp_contact_epb.sh at 1024 frames, 1 / 2 / 3 / 4 per block: slowest block 329 / 339 / 355 / 369 k cycles, mean frame 2.13 / 2.11 / 2.05 / 2.01 M cycles per step
		static const int epb_env = ht_tuning_int("HT_CONTACT_EPB", 0);      // -DHT_TUNING builds: pins it
		const int blocks = (B + nfr - 1) / nfr;
		ht_launch_contact_order(ctx->d_cwork, ctx->d_corder, B, nfr, ctx->cstride, ctx->cwork_mask, HT_CONTACT_SLOTS, epb_env > 0 ? epb_env : blocks > ctx->n_cu ? nfr : 1, t);
		ctx->corder_mask = ctx->cwork_mask;
	}
	ctx->cwork_mask = 0; ctx->cwork_B = B;
	ctx->sorder_mask = 0;
	if (solve_history_on(ctx, B) && ctx->swork_B == B && ctx->swork_mask)
	{
		ht_launch_rank_desc(ctx->d_swork, ctx->d_sorder, B, ctx->cstride, ctx->swork_mask, HT_CONTACT_SLOTS, t);
		ctx->sorder_mask = ctx->swork_mask;
	}
	ctx->swork_mask = 0; ctx->swork_B = B;
}
static void fork(ht_ctx *ctx, hipStream_t s) { (void)hipEventRecord(ctx->ev_fork, s); for (int i = 0; i < 2; i++) (void)hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0); }
static void fork1(ht_ctx *ctx, hipStream_t s, int i) { (void)hipEventRecord(ctx->ev_fork, s); (void)hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0); }
static void join1(ht_ctx *ctx, hipStream_t s, int i) { (void)hipEventRecord(ctx->ev_join[i], ctx->side[i]); (void)hipStreamWaitEvent(s, ctx->ev_join[i], 0); }
static void join(ht_ctx *ctx, hipStream_t s, int n) { for (int i = 0; i < n; i++) { (void)hipEventRecord(ctx->ev_join[i], ctx->side[i]); (void)hipStreamWaitEvent(s, ctx->ev_join[i], 0); } }

// HandTracker::MultiStepSim on othermodel (handtrack.h:642-690)
// `active`: when given, only the frames whose flag is set are touched.  `side`: index of the side stream the cloud rows of a step run on beside the
// contacts (-1: everything in order on s).  `prof`: bracket the solves for the profile (off for a concurrent second instance).
// `part`: 0 a whole step, 1 only what precedes the solve (cloud rows, contacts), 2 only the solve.
static void multistep(ht_ctx *ctx, int B, hipStream_t s, int from_step = 0, int to_step = 1 << 30, const int *active = nullptr, bool first_contacts_done = false, int side = 0, bool prof = true, bool shared_gpu = false,
                      int part = 0)
{
	const ht_params &p = ctx->par;
	for (int st = from_step; st < p.steps && st < to_step; st++)
	{
		const bool angles = (st < p.steps_keyangles) || p.angles_only;
		const bool rays = (st < p.steps_keypoints) && !p.angles_only;
		const bool cloud = (st >= p.steps_cloudstart) && !p.angles_only;
		const bool coll = ctx->phys.use_collision != 0;
		static const bool no_side = ht_tuning_env("HT_NO_SIDE");      // timing experiments (-DHT_TUNING builds only)
		const bool pose_only = solve_tables_mode(ctx) == 2 && side >= 0 && coll && !ctx->profile_phases && !no_side && part == 0 && !active;      // the pose-only tables: on the OTHER side stream, beside the cloud rows
		const bool tables = solve_tables_on(ctx) || pose_only;
		// beside the contact kernel: the step's cloud rows, and (round 6) the solve's tables behind them -- a step without cloud rows forks for the tables alone
		const bool par = side >= 0 && (cloud || tables) && coll && !ctx->profile_phases && !no_side;
		if (part != 2)
		{
			if (par && pose_only) fork(ctx, s); else if (par) fork1(ctx, s, side);
			const cloud_records cr = cloud_rec(ctx);
			if (cloud) { ht_prof_scope ps(ctx, prof ? "cloud_rows" : nullptr, s, true); ht_launch_cloud_rows(ctx->model, ctx->d_state[1], ctx->d_pts, ctx->d_npts, ctx->d_cams, active, 4, 1, 2, p, ctx->d_rows, ctx->d_nrows, B, par ? ctx->side[side] : s, 0.0f, 0.0f, rec_or_rows(ctx, &cr)); }
			if (pose_only) solve_prep(ctx, 1, false, false, active, angles, st < p.steps_palmangle ? 10000.0f : 0.0f, rays, 1, B, ctx->side[1 - side], 1);
			else if (tables) { ht_prof_scope ps(ctx, prof ? "solve_prep" : nullptr, s, true); solve_prep(ctx, 1, cloud, false, active, angles, st < p.steps_palmangle ? 10000.0f : 0.0f, rays, 1, B, par ? ctx->side[side] : s); }
			if (coll && !(first_contacts_done && st == from_step)) { ht_prof_scope ps(ctx, prof ? "contacts" : nullptr, s, true); const contact_slot ch = contact_history(ctx, st < 8 ? st : -1, active, B); ht_launch_contacts(ctx->model, ctx->d_state[1], ctx->phys.driftmax, ctx->phys.jiggle_sin, active, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, false, ctx->contact_kernel, active && active == ctx->d_flags && !ctx->many_reset, ch.order, ch.work); }
			if (par && pose_only) join(ctx, s, 2); else if (par) join1(ctx, s, side);
			if (part == 0 && !active) mark("  step: rows done", s);
		}
		if (part == 1) continue;
		ht_prof_scope ps(ctx, prof ? "solve" : nullptr, s);
		solve_step(ctx, 1, nullptr, nullptr, cloud, coll, active, angles, st < p.steps_palmangle ? 10000.0f : 0.0f, rays, 1, 1, B, s, shared_gpu, nullptr, nullptr, st < 8 ? st : -1, tables);
	}
}
// one main-thread pass of HandTracker::update (handtrack.h:769-780)
static void main_pass(ht_ctx *ctx, int B, hipStream_t s, float *poses_out = nullptr, int pass = -1)      // poses_out: the update's last pass also writes the user poses
{
	const ht_params &p = ctx->par;
	const float4 *pts = p.subsample_voxel ? ctx->d_ptsv : ctx->d_pts;      // handtrack.h:751: the main-thread cloud
	const int *npts = p.subsample_voxel ? ctx->d_nptsv : ctx->d_npts;
	const bool coll = ctx->phys.use_collision != 0;
	static const bool no_side = ht_tuning_env("HT_NO_SIDE");
	const bool par = !ctx->profile_phases && !no_side;
	const bool pose_only = solve_tables_mode(ctx) == 2 && par;
	const bool tables = solve_tables_on(ctx);
	// Round 6: the five boundary planes follow from the points alone, so an update makes them once (beside the net: run_update) and every pass only their rows (k_chamber,
	// or k_solve_prep with the solve tables).  A pass outside an update (ht_stage_fit) makes them here.
	if (!ctx->planes_valid) { ht_prof_scope ps(ctx, "chamber", s, true); ht_launch_chamber_planes(ctx->model, pts, npts, p.min_point_num, p.boundary_planes, ctx->d_chplanes, ctx->d_chon, B, s); }
	if (par) { if (tables) fork1(ctx, s, 0); else fork(ctx, s); }
	if (pose_only) solve_prep(ctx, 0, false, false, nullptr, 0, 0.0f, 0, 0, B, ctx->side[1], 1);      // the pose-only tables ahead of the boundary planes on their side stream: both beside the cloud rows
	if (!tables) { ht_prof_scope ps(ctx, "chamber", s, true); ht_launch_chamber(ctx->model, ctx->d_state[0], pts, npts, p.min_point_num, p.boundary_planes, 10.0f, ctx->d_chamber, ctx->d_nchamber, B, par ? ctx->side[1] : s, ctx->d_chplanes, ctx->d_chon); }
	const cloud_records cr = cloud_rec(ctx);
	{ ht_prof_scope ps(ctx, "cloud_rows", s, true); ht_launch_cloud_rows(ctx->model, ctx->d_state[0], pts, npts, ctx->d_cams, nullptr, 1, 0, 1, p, ctx->d_rows, ctx->d_nrows, B, par ? ctx->side[0] : s, 0.0f, 0.0f, rec_or_rows(ctx, &cr)); }
	if (tables) { ht_prof_scope ps(ctx, "solve_prep", s, true); solve_prep(ctx, 0, true, true, nullptr, 0, 0.0f, 0, 0, B, par ? ctx->side[0] : s); }
	if (coll) { ht_prof_scope ps(ctx, "contacts", s, true); const contact_slot ch = contact_history(ctx, pass >= 0 && pass < 8 ? 8 + pass : -1, nullptr, B); ht_launch_contacts(ctx->model, ctx->d_state[0], ctx->phys.driftmax, ctx->phys.jiggle_sin, nullptr, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, par, ctx->contact_kernel, 0, ch.order, ch.work); }
	mark("  pass: contacts done", s);
	if (par) { mark("  pass: cloud rows done", ctx->side[0]); if (!tables) mark("  pass: chamber done", ctx->side[1]); }
	if (par) join(ctx, s, tables ? 1 : 2);
	ht_prof_scope ps(ctx, "solve", s);
	solve_step(ctx, 0, ctx->d_chamber, ctx->d_nchamber, true, coll, nullptr, 0, 0.0f, 0, 0, 0, B, s, false, poses_out, npts, pass >= 0 && pass < 8 ? 8 + pass : -1, tables || pose_only);
}
// Behind the join of the side stream, so nothing of the step waits for it: the running counts of reset frames go to the host (ht_host.hpp: d_nreset).  The
// update's stream picks the copy up again at its very end (reset_tail_join: long finished by then) -- every stream of an update has to come back to the
// caller's, or the update could not be captured into a HIP graph.
static void reset_tail(ht_ctx *ctx)
{
	(void)hipMemcpyAsync(const_cast<unsigned *>(ctx->h_nreset), ctx->d_nreset, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->side[0]);
	ctx->tail_pending = true;
}
static void reset_tail_join(ht_ctx *ctx, hipStream_t s) { if (ctx->tail_pending) { join(ctx, s, 1); ctx->tail_pending = false; } }
static void reset_path(ht_ctx *ctx, bool listed, int n_unibody, int B, hipStream_t s, hipStream_t prof_stream, bool many_frames = false)      // listed: the frames of d_flist; otherwise all
{
	ht_prof_scope ps(ctx, "reset_path", prof_stream, true);
	if (listed) ctx->last_reset_many = many_frames ? 1 : 0;
	ht_launch_reset(ctx->model, ctx->phys, ctx->d_state[1], ctx->d_pts, ctx->d_npts, ctx->d_analysis, ctx->d_cams, listed ? ctx->d_flist : nullptr, listed ? ctx->d_nflist : nullptr, n_unibody, ctx->par,
	                ctx->d_rows, ctx->d_nrows, ctx->d_scratch, scratch_stride(ctx), ctx->B, B, s, many_frames, ctx->n_cu, exact_solver(ctx));
}

// the whole unit of work on device buffers
// `fs` != null: d_depth / d_cams are full-size frames (handtrack.h:693-785 with dim != 64x64).  The tracker segments them for the CNN
// (handtrack.h:697-698) and from then on ctx->d_cams holds the SEGMENT cameras (CNN decode, landmark rays, PoseFromScratch, UnibodyFit and
// MultiStepSim take segment.cam.pose); the point cloud and FitError keep the full frame and its camera.
// `direct` != 0: the frame (direct x direct pixels) is its own segment and feeds the net of that input size -- what HandSegmentVR returns for a frame of the net's
// size (handtrack.h:283-284), the stages of :693-729 called directly (BASELINE configs[4] end to end, SURVEY 8d config 5 i-iii); heat-map camera camsub(cam, direct / 16).
struct frame_src { int w, h; float segment_scale; int direct; };
// `mode`: UPD_FULL = HandTracker::update (handtrack.h:748-785); UPD_CNN_MODEL = update_cnn_model alone (:734-741): othermodel is NOT re-seeded from
// handmodel, no main-thread passes, no "initializing = 50" rule, the result is othermodel.GetPose() plus the accept decision, handmodel untouched;
// UPD_KICKSTART = kickstart (:743-746) = the same followed by handmodel.SetPose(pose) where the pose was accepted.
// UPD_PASSES = only the caller's part of update() (:751-753, 769-785): the cloud of the frame, the main-thread passes on handmodel, the user poses (the overlapped mode, ht_update_passes_sync)
enum { UPD_FULL = 0, UPD_CNN_MODEL = 1, UPD_KICKSTART = 2, UPD_PASSES = 3 };
static int run_update_(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, const float *d_start, int B, float *d_poses_out, float *d_cnn_out, hipStream_t s, const frame_src *fs, int mode);
static int run_update(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, const float *d_start, int B, float *d_poses_out, float *d_cnn_out, hipStream_t s, const frame_src *fs = nullptr, int mode = UPD_FULL)
{
	const int r = run_update_(ctx, d_depth, d_cams, d_start, B, d_poses_out, d_cnn_out, s, fs, mode);
	ctx->model.frame_order = nullptr;      // the launch order of the block-per-frame kernels belongs to the update that made it
	ctx->planes_valid = false;             // and so do the boundary planes of its cloud
	return r;
}
static int run_update_(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, const float *d_start, int B, float *d_poses_out, float *d_cnn_out, hipStream_t s, const frame_src *fs, int mode)
{
	if (mode == UPD_PASSES) {}      // no net in the caller's part
	else if (fs && fs->direct) { if (!ctx->have_weights128) { ctx->err = "weights of the 128x128 net not loaded (ht_cnn_load_weights_sized)"; return HT_ERR_STATE; } }
	else if (!ctx->have_weights) { ctx->err = "CNN weights not loaded (ht_cnn_load_weights)"; return HT_ERR_STATE; }
	const ht_params &p = ctx->par;
	const int nb = ctx->model.nb;
	const int iw = fs ? fs->w : 64, ih = fs ? fs->h : 64;      // the image FitError looks at
	{ const int npx = iw * ih, fr = p.subsample_fraction > 0 ? p.subsample_fraction : 1, n = ((mode == UPD_FULL || mode == UPD_PASSES) && p.subsample_voxel) ? npx : (npx + fr - 1) / fr; if (n > ctx->model.pts_cap) { const int r = ht_reserve_points_locked(ctx, n); if (r) return r; } ctx->model.pts_bound = (n + 63) & ~63; }
	const float *img_cams = ctx->d_cams;
	if (fs)
	{
		if (!ctx->d_seg_tiles)
		{
			void *a = nullptr, *b = nullptr, *c = nullptr;
			HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * 4096 * sizeof(uint16_t))); ctx->allocs.push_back(a); ctx->d_seg_tiles = (uint16_t *)a;
			HIPCHK(ctx, hipMalloc(&b, (size_t)ctx->B * HT_CAM * sizeof(float))); ctx->allocs.push_back(b); ctx->d_frame_cams = (float *)b;
			HIPCHK(ctx, hipMalloc(&c, sizeof(int))); ctx->allocs.push_back(c); ctx->d_overflow = (int *)c;
		}
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_cams, d_cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyDeviceToDevice, s));
		HIPCHK(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(int), s));
		if (fs->direct) { if (d_cams != ctx->d_cams) HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, d_cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyDeviceToDevice, s)); }      // segment.cam = the frame's camera
		else if (mode != UPD_PASSES) ht_launch_segment(d_depth, ctx->d_frame_cams, fs->w, fs->h, 0xF, p.drangey, fs->segment_scale, ctx->d_seg_tiles, ctx->d_cams, B, s);
		img_cams = ctx->d_frame_cams;
	}
	// 64x64 tiles: the camera copy and the re-seeding of the trackers ride on k_prepare (below); full-size frames keep their own small kernels
	ht_prepare_extra px = { (!fs && d_cams != ctx->d_cams) ? ctx->d_cams : nullptr, ctx->d_state[0], ctx->d_state[1], fs ? nullptr : d_start, ctx->d_prev_err, ctx->d_initializing, nb, ctx->d_nflist };
	if (d_start && fs)
	{
		ht_launch_set_pose(ctx->d_state[0], d_start, nb, B, 1, s);
		ht_launch_set_pose(ctx->d_state[1], d_start, nb, B, 1, s);
		ht_launch_clear_flags(ctx->d_prev_err, ctx->d_initializing, B, s);
	}
	{
		ht_prof_scope ps(ctx, "prepare", s, true);
		if (fs)
		{
			const ht_prepare_extra pz = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, ctx->d_nflist };
			if (fs->direct) { HIPCHK(ctx, hipMemsetAsync(ctx->d_nflist, 0, sizeof(int), s)); ht_launch_cnn_input(d_depth, ctx->d_cams, fs->w * fs->h, p.drangey, ctx->d_in128, B, s); }
			else if (mode != UPD_PASSES) ht_launch_prepare(ctx->d_seg_tiles, ctx->d_cams, p.drangey, p.subsample_fraction, ctx->d_cnn_in, nullptr, nullptr, ctx->model.pts_cap, B, s, &pz);      // (the caller's part of the overlapped update has no segment and no net)
			ht_launch_prepare_frame(d_depth, img_cams, fs->w, fs->h, p.drangey, p.subsample_fraction, ctx->d_pts, ctx->d_npts, ctx->d_overflow, ctx->model.pts_cap, B, s);
		}
		else ht_launch_prepare(d_depth, d_cams, p.drangey, p.subsample_fraction, ctx->d_cnn_in, ctx->d_pts, ctx->d_npts, ctx->model.pts_cap, B, s, &px);
		if ((mode == UPD_FULL || mode == UPD_PASSES) && p.subsample_voxel)      // UPD_PASSES: the caller's part of the overlapped update runs its passes on this very cloud (handtrack.h:753)
		{
			// the main-thread cloud of handtrack.h:751 with the voxel rule: ALL in-range points (taken once more, into the cloud-row array, which nothing
			// uses before the first fit step) go through the voxel table; the CNN job keeps the every-n-th cloud above (handtrack.h:703)
			if (!ctx->d_ptsv) { int r = dev_alloc_points(ctx); if (r) return r; }
			float4 *all = reinterpret_cast<float4 *>(ctx->d_rows);
			if (fs) ht_launch_prepare_frame(d_depth, img_cams, fs->w, fs->h, p.drangey, 1, all, ctx->d_nrows, ctx->d_overflow, ctx->model.pts_cap, B, s);
			else ht_launch_prepare(d_depth, ctx->d_cams, p.drangey, 1, nullptr, all, ctx->d_nrows, ctx->model.pts_cap, B, s);
			ht_launch_voxel(all, ctx->d_nrows, ctx->model.pts_cap, p.subsample_size, p.subsample_fraction, ctx->d_ptsv, ctx->d_nptsv, B, s);
		}
	}
	// batches of several rounds per CU: the block-per-frame kernels take the frames with the most points first, so that a launch ends on short blocks
	if (ctx->d_porder && B > ctx->n_cu * 8 && !exact_solver(ctx)) { ht_launch_order_by_points(ctx->d_npts, ctx->d_porder, B, s); ctx->model.frame_order = ctx->d_porder; }
	auto update_planes = [&](hipStream_t t) {      // the boundary planes of the main-thread cloud (handtrack.h:751, 774-778), once per update
		if (mode == UPD_CNN_MODEL || mode == UPD_KICKSTART || p.angles_only || p.mainthreadpasses < 1) return;
		ht_launch_chamber_planes(ctx->model, p.subsample_voxel ? ctx->d_ptsv : ctx->d_pts, p.subsample_voxel ? ctx->d_nptsv : ctx->d_npts, p.min_point_num, p.boundary_planes, ctx->d_chplanes, ctx->d_chon, B, t);
		ctx->planes_valid = true;
	};
	if (mode == UPD_PASSES)
	{
		const int passes = p.angles_only ? 0 : p.mainthreadpasses;
		update_planes(s);
		for (int i = 0; i < passes; i++) main_pass(ctx, B, s, i + 1 == passes ? d_poses_out : nullptr, i);
		if (passes < 1) ht_launch_output(ctx->model, ctx->d_state[0], p.subsample_voxel ? ctx->d_nptsv : ctx->d_npts, ctx->d_initializing, p.min_point_num, d_poses_out, B, s);
		return HT_OK;
	}
	float *cnn_out = d_cnn_out ? d_cnn_out : ctx->d_cnn_out;
	static const bool no_overlap = ht_tuning_env("HT_NO_OVERLAP");      // timing experiments (-DHT_TUNING builds only)
	const bool overlap = !no_overlap && !ctx->profile_phases && p.steps >= 1 && p.steps_cloudstart >= 1 && !p.angles_only && ctx->solver_build != 5;
	if (overlap)
	{
		// Nothing on this side branch needs the CNN: the error of the carried pose and the reset decision only read the point cloud and the
		// tracker state, so they run beside the CNN.  (The contacts of MultiStepSim's first step do not need it either, but the contact kernel
		// owns whole CUs and the FC layers want one block per CU: beside each other they took 0.78 ms, one after the other 0.46.)
		hipStream_t t = ctx->side[1];
		fork(ctx, s);
		update_planes(t);
		contact_orders(ctx, B, t);
		if (mode == UPD_FULL && !(d_start && !fs)) ht_launch_set_pose(ctx->d_state[1], ctx->d_state[0], nb, B, 2, t);     // othermodel.SetPose(handmodel.GetPose()) handtrack.h:757 (both were just seeded with the same pose otherwise)
		ht_fit_after dec; memset(&dec, 0, sizeof dec);
		dec.mode = 1; dec.reset_thr = p.full_reset_on_error; dec.angles_only = p.angles_only; dec.flags = ctx->d_flags; dec.nflags = ctx->d_nflags; dec.list = ctx->d_flist; dec.nlist = ctx->d_nflist; dec.nreset = ctx->d_nreset;
		ht_launch_fit_error(ctx->model, ctx->d_state[0], ctx->d_pts, ctx->d_npts, d_depth, img_cams, iw, ih, p.bone_sum_error_scale, ctx->d_err_old, B, t, &dec);      // with the reset decision (handtrack.h:706)
	}
	if (!overlap) { contact_orders(ctx, B, s); update_planes(s); }
	{
		ht_prof_scope ps(ctx, (fs && fs->direct) ? "cnn128" : "cnn", s, true);
		if (fs && fs->direct) ht_launch_cnn(ctx->cnnw128, ctx->d_in128, ctx->d_act1_128, ctx->d_act2_128, ctx->d_act3, ctx->d_logits, B, s, fs->direct, overlap);
		else ht_launch_cnn(ctx->cnnw, ctx->d_cnn_in, ctx->d_act1, ctx->d_act2, ctx->d_act3, ctx->d_logits, B, s, 64, overlap);      // overlap: the side branch's FitError runs beside the net
		ht_launch_softmax_decode(ctx->d_logits, cnn_out, ctx->d_cams, ctx->d_analysis, 1, B, s, (fs && fs->direct) ? fs->direct / 16 : 4);
	}
	if (overlap)
	{
		(void)hipEventRecord(ctx->ev_join[1], ctx->side[1]); (void)hipStreamWaitEvent(s, ctx->ev_join[1], 0);
		// the full-reset path touches few frames but is long (3 sequential single-body solves): it runs on a side stream while step 0 of
		// MultiStepSim (which uses no cloud rows) proceeds for all other frames; the reset frames then do their step 0 on their own.
		// (Taking the reset frames through ALL their steps on the side stream was measured: their five few-frame steps are pure latency and end
		// later than the main stream's full-batch steps plus this one extra step, 4.6 against 4.5 ms.)
		static const int join_step = ht_tuning_int("HT_RESET_JOIN", 0);      // experiment (-DHT_TUNING): the reset frames take steps [0, join_step) on the side stream
		fork(ctx, s);
		{
			// few or many reset frames (ht_host.hpp: d_nreset): the average over the updates whose counts have arrived since the last look
			const unsigned frames = ctx->h_nreset[0], updates = ctx->h_nreset[1];
			if (updates != ctx->nreset_seen[1])
			{
				ctx->many_reset = (frames - ctx->nreset_seen[0]) / (updates - ctx->nreset_seen[1]) > (unsigned)ctx->n_cu;
				ctx->nreset_seen[0] = frames; ctx->nreset_seen[1] = updates;
			}
		}
		static const bool no_lap = ht_tuning_env("HT_NO_STEP1_LAP");      // experiment (-DHT_TUNING)
		if (join_step > 0)
		{
			reset_path(ctx, true, p.steps_unibody, B, ctx->side[0], s, ctx->many_reset);
			multistep(ctx, B, ctx->side[0], 0, join_step, ctx->d_flags, false, -1, false, true);
			multistep(ctx, B, s, 0, join_step, ctx->d_nflags, false, 1, true, true);
			join(ctx, s, 1);
			reset_tail(ctx);
			multistep(ctx, B, s, join_step);
		}
		else if (p.steps >= 2 && !no_lap)
		{
			// The reset frames' chain -- the reset kernel, then their own first step (eight frames: pure latency) -- is what the batch ends up waiting for, and its
			// few long blocks must find CUs although the batch's kernels ask for all of them (a cooperative contact block for a whole CU's LDS).  So the chain
			// stays on THIS stream, where it is dispatched the moment the CNN ends, and the batch's first step goes to the side stream, which only starts
			// after a cross-queue wait: launched the other way round the reset blocks often found no CU until the batch's contact kernel had finished.
			// While the reset frames take their first step the batch prepares its second (cloud rows and contacts of the other frames: a frame's rows and
			// contacts are its own), after the reset frames' contact blocks are in (same reason); the reset frames' rows for step 1 follow their solve on this
			// stream, beside the batch's.  Then ONE solve for all frames.  (The reset frames any further behind the batch was measured and does not pay:
			// DESIGN.md section 4.)  What the reset frames' contact blocks wait for (event marks): the batch's first solve -- both contact kernels want more
			// registers than a SIMD has left beside a solver wave (256 and 410 against 512 - 168), so they start when that solve ends, 0.55 ms after the fork,
			// however early the reset kernel is through (0.25 ms); making the batch's solve wait for them instead was measured: 5.39 against 5.31 ms.
			hipStream_t u = ctx->side[0];
			mark("fork", s);
			reset_path(ctx, true, p.steps_unibody, B, s, s, ctx->many_reset);
			mark("reset kernel done", s);
			multistep(ctx, B, u, 0, 1, ctx->d_nflags, false, -1, false, true);
			mark("batch step 0 done", u);
			multistep(ctx, B, s, 0, 1, ctx->d_flags, false, -1, true, false, 1);
			mark("reset frames contacts done", s);
			(void)hipEventRecord(ctx->ev_lap, s); (void)hipStreamWaitEvent(u, ctx->ev_lap, 0);
			multistep(ctx, B, s, 0, 1, ctx->d_flags, false, -1, true, false, 2);
			mark("reset frames step 0 done", s);
			multistep(ctx, B, u, 1, 2, ctx->d_nflags, false, -1, false, false, 1);      // in order on the side stream: there is time (0.76 against 0.82 ms), and a fork out of a forked stream does not survive a HIP graph capture
			mark("batch step 1 rows done", u);
			multistep(ctx, B, s, 1, 2, ctx->d_flags, false, 1, true, false, 1);      // the reset frames' rows for step 1, beside the batch's
			mark("reset frames step 1 rows done", s);
			join(ctx, s, 1);
			reset_tail(ctx);
			multistep(ctx, B, s, 1, 2, nullptr, false, 0, true, false, 2);
			mark("step 1 done", s);
			multistep(ctx, B, s, 2);
			mark("MultiStepSim done", s);
		}
		else
		{
			reset_path(ctx, true, p.steps_unibody, B, ctx->side[0], s, ctx->many_reset);
			multistep(ctx, B, s, 0, 1, ctx->d_nflags, false, 0, true, true);
			join(ctx, s, 1);
			reset_tail(ctx);
			multistep(ctx, B, s, 0, 1, ctx->d_flags);
			multistep(ctx, B, s, 1);
		}
	}
	else
	{
		if (mode == UPD_FULL) ht_launch_set_pose(ctx->d_state[1], ctx->d_state[0], nb, B, 2, s);     // othermodel.SetPose(handmodel.GetPose()) handtrack.h:757
		ht_fit_after dec; memset(&dec, 0, sizeof dec);
		dec.mode = 1; dec.reset_thr = p.full_reset_on_error; dec.angles_only = p.angles_only; dec.flags = ctx->d_flags; dec.nflags = ctx->d_nflags; dec.list = ctx->d_flist; dec.nlist = ctx->d_nflist; dec.nreset = ctx->d_nreset;
		{ ht_prof_scope ps(ctx, "fit_error", s, true); ht_launch_fit_error(ctx->model, ctx->d_state[0], ctx->d_pts, ctx->d_npts, d_depth, img_cams, iw, ih, p.bone_sum_error_scale, ctx->d_err_old, B, s, &dec); }
		reset_path(ctx, true, p.steps_unibody, B, s, s);
		multistep(ctx, B, s);
	}
	{
		// FitError of the CNN-driven pose, and on its last thread the accept step (handtrack.h:713-731)
		ht_fit_after acc; memset(&acc, 0, sizeof acc);
		acc.mode = 2; acc.hand = mode == UPD_CNN_MODEL ? nullptr : ctx->d_state[0]; acc.other = ctx->d_state[1]; acc.err_old = ctx->d_err_old; acc.prev_err = ctx->d_prev_err;
		acc.initializing = ctx->d_initializing; acc.accepted = ctx->d_accepted; acc.nb = nb; acc.min_point_num = p.min_point_num; acc.always_take_cnn = p.always_take_cnn;
		acc.angles_only = p.angles_only; acc.accum_thr = p.accum_error_threshold;
		ht_prof_scope ps(ctx, "fit_error", s, true);
		ht_launch_fit_error(ctx->model, ctx->d_state[1], ctx->d_pts, ctx->d_npts, d_depth, img_cams, iw, ih, p.bone_sum_error_scale, ctx->d_err_new, B, s, &acc);
	}
	if (mode != UPD_FULL) { ht_launch_output(ctx->model, ctx->d_state[1], ctx->d_npts, ctx->d_initializing, p.min_point_num, d_poses_out, B, s, 1); reset_tail_join(ctx, s); return HT_OK; }      // othermodel.GetPose()
	const int passes = p.angles_only ? 0 : p.mainthreadpasses;
	mark("accept done", s);
	for (int i = 0; i < passes; i++) { main_pass(ctx, B, s, i + 1 == passes ? d_poses_out : nullptr, i); mark("pass done", s); }      // the last pass's solve writes the poses
	if (passes < 1) ht_launch_output(ctx->model, ctx->d_state[0], p.subsample_voxel ? ctx->d_nptsv : ctx->d_npts, ctx->d_initializing, p.min_point_num, d_poses_out, B, s);
	reset_tail_join(ctx, s);
	mark("update done", s);
	marks_dump();
	return HT_OK;
}

// ---- public entry points ----------------------------------------------------------------------------------------------
extern "C" int ht_tracker_reset(ht_ctx *ctx, int first, int n, const float *poses)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	if (!poses) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage, poses, (size_t)n * nb * HT_POSE * sizeof(float), hipMemcpyHostToDevice, s));
	for (int w = 0; w < 2; w++) ht_launch_set_pose(ctx->d_state[w] + (size_t)first * nb * HT_STATE_STRIDE, ctx->d_stage, nb, n, 1, s);
	ht_launch_clear_flags(ctx->d_prev_err + first, ctx->d_initializing + first, n, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_get_state(ht_ctx *ctx, int which, int first, int n, float *state)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	if (!state || which < 0 || which > 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	ht_launch_get_state(ctx->d_state[which] + (size_t)first * nb * HT_STATE_STRIDE, ctx->d_stage, nb, n, s);
	HIPCHK(ctx, hipMemcpyAsync(state, ctx->d_stage, (size_t)n * nb * HT_STATE * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_set_state(ht_ctx *ctx, int which, int first, int n, const float *state)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	if (!state || which < 0 || which > 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage, state, (size_t)n * nb * HT_STATE * sizeof(float), hipMemcpyHostToDevice, s));
	ht_launch_set_pose(ctx->d_state[which] + (size_t)first * nb * HT_STATE_STRIDE, ctx->d_stage, nb, n, 3, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_get_tracker_flags(ht_ctx *ctx, int first, int n, float *prev_frame_error, int *initializing)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	HIPCHK(ctx, ht_sync_all(ctx));
	if (prev_frame_error) HIPCHK(ctx, hipMemcpy(prev_frame_error, ctx->d_prev_err + first, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
	if (initializing) HIPCHK(ctx, hipMemcpy(initializing, ctx->d_initializing + first, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
	return HT_OK;
}
extern "C" int ht_set_tracker_flags(ht_ctx *ctx, int first, int n, const float *prev_frame_error, const int *initializing)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	if (prev_frame_error) HIPCHK(ctx, hipMemcpy(ctx->d_prev_err + first, prev_frame_error, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
	if (initializing) HIPCHK(ctx, hipMemcpy(ctx->d_initializing + first, initializing, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
	return HT_OK;
}
extern "C" int ht_update_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, const float *d_start_poses, int B, float *d_poses_out, void *stream)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!d_depth || !d_cams || !d_poses_out) return HT_ERR_ARG;
	int r = run_update(ctx, d_depth, d_cams, d_start_poses, B, d_poses_out, nullptr, ht_user_stream(ctx, stream));
	if (r) return r;
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_update_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int B, float *poses_out, float *cnn_out)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams || !poses_out) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_depth, depth, (size_t)B * 4096 * sizeof(uint16_t), hipMemcpyHostToDevice, s));
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
	int r = run_update(ctx, ctx->d_depth, ctx->d_cams, nullptr, B, ctx->d_poses_out, nullptr, s);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->d_poses_out, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyDeviceToHost, s));
	if (cnn_out) HIPCHK(ctx, hipMemcpyAsync(cnn_out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// HandTracker::update on frames of any size up to 320x240 (handtrack.h:693-785): segmentation for the CNN inside, the cloud of the full frame
static int frames_args_ok(ht_ctx *ctx, int w, int h)
{
	if (w == 64 && h == 64) return 1;
	if (!ht_segment_supported(w, h)) { ctx->err = "ht_update_frames: frame size must be a multiple of 4 and at most 320x240 pixels"; return 0; }
	return 1;
}
extern "C" int ht_update_frames_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int w, int h, float segment_scale, const float *d_start_poses, int B, float *d_poses_out, void *stream)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!d_depth || !d_cams || !d_poses_out) return HT_ERR_ARG;
	if (!frames_args_ok(ctx, w, h)) return HT_ERR_ARG;
	hipStream_t s = ht_user_stream(ctx, stream);
	const frame_src fs = { w, h, segment_scale, 0 };
	int r = run_update(ctx, d_depth, d_cams, d_start_poses, B, d_poses_out, nullptr, s, (w == 64 && h == 64) ? nullptr : &fs);
	if (r) return r;
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
// Capacities of the contact kernel that the reference does not have: expanding-polytope runs cut short (128 iterations, 96 vertices, 192
// triangles in LDS; hull.h:246 loops without bound), touching samples beyond the 192 of a frame's pool, and solves whose angular rows exceed the 126
// the solver keeps (a model with many ranged joints).  Counted since ht_create; 0 on every
// workload of the test suite and the benches, so no result there depends on them.
extern "C" int ht_capacity_events(ht_ctx *ctx, int *epa_cut_short, int *contacts_dropped, int *angular_rows_over)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx);
	int v[3] = { 0, 0, 0 };
	HIPCHK(ctx, ht_sync_all(ctx));
	HIPCHK(ctx, hipMemcpy(v, ctx->d_epa_ws, sizeof v, hipMemcpyDeviceToHost));
	if (epa_cut_short) *epa_cut_short = v[0];
	if (contacts_dropped) *contacts_dropped = v[1];
	if (angular_rows_over) *angular_rows_over = v[2];
	return HT_OK;
}
// What the contact kernel's per-frame pool can be asked to hold AT MOST, from the model alone: FindShapeShapeContacts (physics.h:451-462) visits every pair of colliding bodies
// that do not ignore each other, and ContactPatch (gjk.h:607-643) returns one sample for a pair -- five only when neither body is smaller than the 0.05 m of its proximity test
// (the kernel's exact shortcut: DESIGN.md section 4).  samples_bound <= pool and patches_bound <= patch_slots is a PROOF that no touching sample can be dropped for this model
// (the stock hand: 91 pairs, none of them between two large bodies that do not ignore each other: 91 <= 192, 0 <= 40); otherwise the pool is a counted capacity
// (ht_capacity_events).  Follows ht_scale (the diameters grow with the model).
extern "C" int ht_contact_capacity(ht_ctx *ctx, int *samples_bound, int *patches_bound, int *pool, int *patch_slots)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx);
	const int nb = ctx->model.nb;
	int pairs = 0, big = 0;
	for (int i = 0; i < nb; i++) for (int j = i + 1; j < nb; j++)
	{
		if (!(ctx->model.collide[i] & ctx->model.collide[j] & 2)) continue;
		if (ctx->model.ignore[i] & (1u << j)) continue;
		pairs++;
		const float di = ctx->h_bodyc[(size_t)i * HT_BC + HT_BC_DIAM], dj = ctx->h_bodyc[(size_t)j * HT_BC + HT_BC_DIAM];
		if (!((di < dj ? di : dj) < 0.049f)) big++;
	}
	if (samples_bound) *samples_bound = pairs + 4 * big;
	if (patches_bound) *patches_bound = big;
	if (pool) *pool = HT_MAXCONTACT;
	if (patch_slots) *patch_slots = 40;      // GJK_JMAX (csrc/ht_gjk.hip)
	return HT_OK;
}
extern "C" int ht_frames_overflow(ht_ctx *ctx, int *frames_over)
{
	CHECK_READY(ctx);
	if (!frames_over) return HT_ERR_ARG;
	*frames_over = 0;
	HIPCHK(ctx, ht_sync_all(ctx));
	if (ctx->d_overflow) HIPCHK(ctx, hipMemcpy(frames_over, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost));
	return HT_OK;
}
extern "C" int ht_update_frames_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B, float *poses_out, float *cnn_out)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams || !poses_out) return HT_ERR_ARG;
	if (w == 64 && h == 64) return ht_update_sync(ctx, depth, cams, B, poses_out, cnn_out);
	if (!frames_args_ok(ctx, w, h)) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	const size_t npx = (size_t)w * h;
	if (ctx->frames_cap < (size_t)B * npx)
	{
		void *a = nullptr;
		HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * npx * sizeof(uint16_t))); ctx->allocs.push_back(a); ctx->d_frames = (uint16_t *)a; ctx->frames_cap = (size_t)ctx->B * npx;
	}
	if (!ctx->d_frame_cams_in) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * HT_CAM * sizeof(float))); ctx->allocs.push_back(a); ctx->d_frame_cams_in = (float *)a; }
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_cams_in, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
	const frame_src fs = { w, h, segment_scale, 0 };
	int r = run_update(ctx, ctx->d_frames, ctx->d_frame_cams_in, nullptr, B, ctx->d_poses_out, nullptr, s, &fs);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->d_poses_out, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyDeviceToHost, s));
	if (cnn_out) HIPCHK(ctx, hipMemcpyAsync(cnn_out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	int over = 0;
	HIPCHK(ctx, hipMemcpy(&over, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost));
	if (over) { ctx->err = "ht_update_frames: " + std::to_string(over) + " frame(s) have more in-range points than the context's point capacity holds; their result is not the reference's"; return HT_ERR_ARG; }
	return HT_OK;
}

// BASELINE configs[4] end to end (SURVEY 8d "config 5 (i)-(iii)"): HandTracker::update on side x side frames that are their own segment, evaluated by the net of
// that input size (ht_cnn_load_weights_sized) -- no HandSegmentVR, heat-map camera camsub(cam, side / 16), everything else as handtrack.h:693-785
extern "C" int ht_update_direct_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int side, const float *d_start_poses, int B, float *d_poses_out, void *stream)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!d_depth || !d_cams || !d_poses_out) return HT_ERR_ARG;
	if (side == 64) return ht_update_dev(ctx, d_depth, d_cams, d_start_poses, B, d_poses_out, stream);
	if (side != 128) { ctx->err = "ht_update_direct: CNN input side must be 64 or 128"; return HT_ERR_ARG; }
	if (((uintptr_t)d_depth & 15) != 0) { ctx->err = "ht_update_direct_dev: d_depth must be 16-byte aligned (the input transform reads eight pixels per 128-bit load)"; return HT_ERR_ARG; }
	hipStream_t s = ht_user_stream(ctx, stream);
	const frame_src fs = { side, side, 0.0f, side };
	int r = run_update(ctx, d_depth, d_cams, d_start_poses, B, d_poses_out, nullptr, s, &fs);
	if (r) return r;
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_update_direct_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int side, int B, float *poses_out, float *cnn_out)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams || !poses_out) return HT_ERR_ARG;
	if (side == 64) return ht_update_sync(ctx, depth, cams, B, poses_out, cnn_out);
	if (side != 128) { ctx->err = "ht_update_direct: CNN input side must be 64 or 128"; return HT_ERR_ARG; }
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	const size_t npx = (size_t)side * side;
	if (ctx->frames_cap < (size_t)B * npx) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * npx * sizeof(uint16_t))); ctx->allocs.push_back(a); ctx->d_frames = (uint16_t *)a; ctx->frames_cap = (size_t)ctx->B * npx; }
	if (!ctx->d_frame_cams_in) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * HT_CAM * sizeof(float))); ctx->allocs.push_back(a); ctx->d_frame_cams_in = (float *)a; }
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_cams_in, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
	const frame_src fs = { side, side, 0.0f, side };
	int r = run_update(ctx, ctx->d_frames, ctx->d_frame_cams_in, nullptr, B, ctx->d_poses_out, nullptr, s, &fs);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->d_poses_out, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyDeviceToHost, s));
	if (cnn_out) HIPCHK(ctx, hipMemcpyAsync(cnn_out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// HandTracker::update_cnn_model (handtrack.h:734-741) and kickstart (:743-746) for B trackers, frames of any supported size
extern "C" int ht_update_cnn_model_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B, int apply_to_handmodel,
                                        float *poses_out, int *accepted_out, float *cnn_out)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams) return HT_ERR_ARG;
	const bool tile = (w == 64 && h == 64);
	if (!tile && !frames_args_ok(ctx, w, h)) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	const size_t npx = (size_t)w * h;
	const uint16_t *d_in; const float *d_cin;
	if (tile)
	{
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_depth, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
		d_in = ctx->d_depth; d_cin = ctx->d_cams;
	}
	else
	{
		if (ctx->frames_cap < (size_t)B * npx) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * npx * sizeof(uint16_t))); ctx->allocs.push_back(a); ctx->d_frames = (uint16_t *)a; ctx->frames_cap = (size_t)ctx->B * npx; }
		if (!ctx->d_frame_cams_in) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * HT_CAM * sizeof(float))); ctx->allocs.push_back(a); ctx->d_frame_cams_in = (float *)a; }
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_cams_in, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
		d_in = ctx->d_frames; d_cin = ctx->d_frame_cams_in;
	}
	const frame_src fs = { w, h, segment_scale, 0 };
	int r = run_update(ctx, d_in, d_cin, nullptr, B, ctx->d_poses_out, nullptr, s, tile ? nullptr : &fs, apply_to_handmodel ? UPD_KICKSTART : UPD_CNN_MODEL);
	if (r) return r;
	if (poses_out) HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->d_poses_out, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyDeviceToHost, s));
	if (accepted_out) HIPCHK(ctx, hipMemcpyAsync(accepted_out, ctx->d_accepted, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
	if (cnn_out) HIPCHK(ctx, hipMemcpyAsync(cnn_out, ctx->d_cnn_out, (size_t)B * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	if (accepted_out) for (int b = 0; b < B; b++) accepted_out[b] = accepted_out[b] != 0;
	if (!tile) { int over = 0; HIPCHK(ctx, hipMemcpy(&over, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost)); if (over) { ctx->err = "ht_update_cnn_model: frame(s) with more in-range points than the solver's capacity"; return HT_ERR_ARG; } }
	return HT_OK;
}
// ---- the reference's OVERLAPPED update() (handtrack.h:748-785) on two contexts of one device ----------------------------------------------------------------
// The reference runs the CNN job (update_cnn_model_threadsafe: net, decode, FitError, reset, MultiStepSim, accept decision) on a background thread and lets the caller go
// on with the main-thread passes; the job's pose is taken over by a later call, when the job is through (:760-768).  Here the job runs on a second context `job` (its own
// streams, buffers and othermodel) beside the caller's context `main`, whose update call does the cloud, the passes and the user poses only:
//   ht_job_start    othermodel.SetPose(handmodel.GetPose()) (:757) -- main's handmodel and tracker flags are copied to `job` --, the frame goes to pinned staging and the
//                   job is enqueued on job's stream; returns without waiting (what std::async does, :758)
//   ht_job_poll     pose_estimator.wait_for(...) == ready (:760) as a hipEventQuery
//   ht_job_collect  handmodel.SetPose(results.pose) (:767) where the job accepted its pose (an empty pose changes nothing); prev_frame_error as the job left it, and the
//                   job's `initializing = max(initializing - 1, 0)` (:726) applied to main's CURRENT value (the caller's own `initializing = 50` (:781) may have intervened:
//                   the reference shares the variable between the threads)
//   ht_update_passes_sync   the caller's part: points, mainthreadpasses x (HandModelEnhancements, cloud_chamber, FitPointCloud), the initializing rule, GetPoseUser
__global__ void k_job_collect(float *__restrict__ hand, const float *__restrict__ other, const int *__restrict__ accepted, float *__restrict__ prev_err, const float *__restrict__ job_prev_err, int *__restrict__ initializing, int nb, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * nb) return;
	const int f = i / nb;
	if (accepted[f]) { float *s = hand + (size_t)i * HT_STATE_STRIDE; const float *o = other + (size_t)i * HT_STATE_STRIDE; for (int k = 0; k < 7; k++) s[k] = o[k]; }
	if (i == f * nb) { prev_err[f] = job_prev_err[f]; const int v = initializing[f] - 1; initializing[f] = v < 0 ? 0 : v; }
}
extern "C" int ht_update_passes_sync(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, int B, float *poses_out)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!depth || !cams || !poses_out) return HT_ERR_ARG;
	const bool tile = (w == 64 && h == 64);
	if (!tile && !frames_args_ok(ctx, w, h)) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const int nb = ctx->model.nb;
	const size_t npx = (size_t)w * h;
	const uint16_t *d_in; const float *d_cin;
	if (tile)
	{
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_depth, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
		d_in = ctx->d_depth; d_cin = ctx->d_cams;
	}
	else
	{
		if (ctx->frames_cap < (size_t)B * npx) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * npx * sizeof(uint16_t))); ctx->allocs.push_back(a); ctx->d_frames = (uint16_t *)a; ctx->frames_cap = (size_t)ctx->B * npx; }
		if (!ctx->d_frame_cams_in) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * HT_CAM * sizeof(float))); ctx->allocs.push_back(a); ctx->d_frame_cams_in = (float *)a; }
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_frames, depth, (size_t)B * npx * sizeof(uint16_t), hipMemcpyHostToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(ctx->d_frame_cams_in, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s));
		d_in = ctx->d_frames; d_cin = ctx->d_frame_cams_in;
	}
	const frame_src fs = { w, h, 0.17f, 0 };
	int r = run_update(ctx, d_in, d_cin, nullptr, B, ctx->d_poses_out, nullptr, s, tile ? nullptr : &fs, UPD_PASSES);
	if (r) return r;
	HIPCHK(ctx, hipMemcpyAsync(poses_out, ctx->d_poses_out, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	if (!tile) { int over = 0; HIPCHK(ctx, hipMemcpy(&over, ctx->d_overflow, sizeof(int), hipMemcpyDeviceToHost)); if (over) { ctx->err = "ht_update_passes: frame(s) with more in-range points than the context's point capacity holds"; return HT_ERR_ARG; } }
	return HT_OK;
}
extern "C" int ht_job_start(ht_ctx *job, ht_ctx *main, const uint16_t *depth, const float *cams, int w, int h, float segment_scale, int B)
{
	CHECK_READY(job); CHECK_MODEL(job); CHECK_BATCH(job, B);
	if (!main || !depth || !cams || main == job || main->device != job->device || main->model.nb != job->model.nb || B > main->B) { job->err = "ht_job_start: the two contexts must be different ones of one device with the same model"; return HT_ERR_ARG; }
	if (job->job_pending) { job->err = "ht_job_start: a job is in flight on this context (ht_job_collect takes it over)"; return HT_ERR_STATE; }
	const bool tile = (w == 64 && h == 64);
	if (!tile && !frames_args_ok(job, w, h)) return HT_ERR_ARG;
	hipStream_t s = job->stream;
	const int nb = job->model.nb;
	const size_t npx = (size_t)w * h, in_bytes = (size_t)B * npx * sizeof(uint16_t), cam_bytes = (size_t)B * HT_CAM * sizeof(float);
	if (!job->ev_job) HIPCHK(job, hipEventCreateWithFlags(&job->ev_job, hipEventDisableTiming));
	if (job->h_job_cap < in_bytes + cam_bytes) { if (job->h_job_in) (void)hipHostFree(job->h_job_in); job->h_job_in = nullptr; HIPCHK(job, hipHostMalloc(&job->h_job_in, in_bytes + cam_bytes, hipHostMallocDefault)); job->h_job_cap = in_bytes + cam_bytes; }
	HIPCHK(job, hipStreamSynchronize(main->stream));      // the caller's context is between two of its (synchronous) calls: its handmodel is final
	memcpy(job->h_job_in, depth, in_bytes); memcpy((char *)job->h_job_in + in_bytes, cams, cam_bytes);      // the caller's image need not outlive this call (the reference moves it into the task)
	const uint16_t *d_in; const float *d_cin;
	if (tile)
	{
		HIPCHK(job, hipMemcpyAsync(job->d_depth, job->h_job_in, in_bytes, hipMemcpyHostToDevice, s));
		HIPCHK(job, hipMemcpyAsync(job->d_cams, (char *)job->h_job_in + in_bytes, cam_bytes, hipMemcpyHostToDevice, s));
		d_in = job->d_depth; d_cin = job->d_cams;
	}
	else
	{
		if (job->frames_cap < (size_t)B * npx) { void *a = nullptr; HIPCHK(job, hipMalloc(&a, (size_t)job->B * npx * sizeof(uint16_t))); job->allocs.push_back(a); job->d_frames = (uint16_t *)a; job->frames_cap = (size_t)job->B * npx; }
		if (!job->d_frame_cams_in) { void *a = nullptr; HIPCHK(job, hipMalloc(&a, (size_t)job->B * HT_CAM * sizeof(float))); job->allocs.push_back(a); job->d_frame_cams_in = (float *)a; }
		HIPCHK(job, hipMemcpyAsync(job->d_frames, job->h_job_in, in_bytes, hipMemcpyHostToDevice, s));
		HIPCHK(job, hipMemcpyAsync(job->d_frame_cams_in, (char *)job->h_job_in + in_bytes, cam_bytes, hipMemcpyHostToDevice, s));
		d_in = job->d_frames; d_cin = job->d_frame_cams_in;
	}
	// the job looks at handmodel (FitError of the carried pose, :704) and starts othermodel from its pose (:757); prev_frame_error and initializing as they stand
	HIPCHK(job, hipMemcpyAsync(job->d_state[0], main->d_state[0], (size_t)B * nb * HT_STATE_STRIDE * sizeof(float), hipMemcpyDeviceToDevice, s));
	ht_launch_set_pose(job->d_state[1], main->d_state[0], nb, B, 2, s);
	HIPCHK(job, hipMemcpyAsync(job->d_prev_err, main->d_prev_err, (size_t)B * sizeof(float), hipMemcpyDeviceToDevice, s));
	HIPCHK(job, hipMemcpyAsync(job->d_initializing, main->d_initializing, (size_t)B * sizeof(int), hipMemcpyDeviceToDevice, s));
	// othermodel's seed is taken BEFORE the caller goes on (the reference copies synchronously in front of std::async, :757): the caller's stream waits for these copies, so its
	// next passes cannot overwrite handmodel under them
	if (!job->ev_seed) HIPCHK(job, hipEventCreateWithFlags(&job->ev_seed, hipEventDisableTiming));
	HIPCHK(job, hipEventRecord(job->ev_seed, s));
	HIPCHK(job, hipStreamWaitEvent(main->stream, job->ev_seed, 0));
	const frame_src fs = { w, h, segment_scale, 0 };
	int r = run_update(job, d_in, d_cin, nullptr, B, job->d_poses_out, nullptr, s, tile ? nullptr : &fs, UPD_CNN_MODEL);
	if (r) { (void)hipStreamSynchronize(s); return r; }      // the queued uploads read h_job_in: nothing may be in flight when the caller frees or reuses it
	HIPCHK(job, hipEventRecord(job->ev_job, s));
	job->job_pending = true;
	return HT_OK;
}
extern "C" int ht_job_poll(ht_ctx *job, int *ready)
{
	if (!job || !ready) return HT_ERR_ARG;
	ht_device_guard dev_guard_(job->device);
	*ready = 0;
	if (!job->job_pending) return HT_OK;
	const hipError_t e = hipEventQuery(job->ev_job);
	if (e == hipSuccess) *ready = 1;
	else if (e != hipErrorNotReady) { job->err = std::string("ht_job_poll: ") + hipGetErrorString(e); return HT_ERR_HIP; }
	return HT_OK;
}
extern "C" int ht_job_wait(ht_ctx *job)
{
	if (!job) return HT_ERR_ARG;
	ht_device_guard dev_guard_(job->device);
	if (job->job_pending) HIPCHK(job, hipEventSynchronize(job->ev_job));
	return HT_OK;
}
extern "C" int ht_job_collect(ht_ctx *job, ht_ctx *main, int B, int *accepted_out)
{
	CHECK_READY(job); CHECK_MODEL(job); CHECK_BATCH(job, B);
	if (!main || main == job || main->device != job->device || main->model.nb != job->model.nb || B > main->B) return HT_ERR_ARG;
	if (!job->job_pending) { job->err = "ht_job_collect: no job in flight"; return HT_ERR_STATE; }
	HIPCHK(job, hipEventSynchronize(job->ev_job));
	const int nb = job->model.nb;
	hipLaunchKernelGGL(k_job_collect, dim3((B * nb + 255) / 256), dim3(256), 0, main->stream, main->d_state[0], job->d_state[1], job->d_accepted, main->d_prev_err, job->d_prev_err, main->d_initializing, nb, B);
	if (accepted_out) { HIPCHK(job, hipMemcpyAsync(accepted_out, job->d_accepted, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, main->stream)); }
	HIPCHK(job, hipStreamSynchronize(main->stream));
	if (accepted_out) for (int b = 0; b < B; b++) accepted_out[b] = accepted_out[b] != 0;
	job->job_pending = false;
	return HT_OK;
}
// Results of the CNN job of the latest update call of slots [first, first + n): HandTracker::cnn_input / cnn_output (handtrack.h:583-584) and the
// decoded CNNOutputAnalysis (:182-242; layout as ht_stage_decode).  Any pointer may be NULL.
extern "C" int ht_get_cnn_results(ht_ctx *ctx, int first, int n, float *cnn_input, float *cnn_output, float *analysis)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_RANGE(ctx, first, n);
	HIPCHK(ctx, ht_sync_all(ctx));
	if (cnn_input) HIPCHK(ctx, hipMemcpy(cnn_input, ctx->d_cnn_in + (size_t)first * HT_CNN_IN, (size_t)n * HT_CNN_IN * sizeof(float), hipMemcpyDeviceToHost));
	if (cnn_output) HIPCHK(ctx, hipMemcpy(cnn_output, ctx->d_cnn_out + (size_t)first * HT_CNN_OUT, (size_t)n * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost));
	if (analysis) HIPCHK(ctx, hipMemcpy(analysis, ctx->d_analysis + (size_t)first * HT_ANALYSIS, (size_t)n * HT_ANALYSIS * sizeof(float), hipMemcpyDeviceToHost));
	return HT_OK;
}

// The CNN's intermediate layers of the latest evaluation (ht_cnn_eval, an update call) of slots [first, first + n), for per-layer parity tests:
// act1 [n][3600] = layer 3 of the reference's list (conv 5x5, tanh, two 2x2 max-pools: 16 x 15 x 15), act2 [n][2304] = layer 6 (conv 4x4, tanh, pool: 64 x 6 x 6),
// act3 [n][2048] = layer 8 (first fully connected layer + tanh), logits [n][2304] = layer 9 (second fully connected layer, before the chunked soft-max).
extern "C" int ht_get_cnn_layers(ht_ctx *ctx, int first, int n, float *act1, float *act2, float *act3, float *logits)
{
	CHECK_READY(ctx); CHECK_RANGE(ctx, first, n);
	HIPCHK(ctx, ht_sync_all(ctx));
	if (act1) HIPCHK(ctx, hipMemcpy(act1, ctx->d_act1 + (size_t)first * 3600, (size_t)n * 3600 * sizeof(float), hipMemcpyDeviceToHost));
	if (act2) HIPCHK(ctx, hipMemcpy(act2, ctx->d_act2 + (size_t)first * 2304, (size_t)n * 2304 * sizeof(float), hipMemcpyDeviceToHost));
	if (act3)      // the device holds the layer in the column order the last layer's kernel reads (k_fc<PACK16>): position (c & ~15) | (c & 3) << 2 | (c >> 2) & 3 holds column c
	{
		std::vector<float> packed((size_t)n * 2048);
		HIPCHK(ctx, hipMemcpy(packed.data(), ctx->d_act3 + (size_t)first * 2048, (size_t)n * 2048 * sizeof(float), hipMemcpyDeviceToHost));
		for (int r = 0; r < n; r++) for (int c = 0; c < 2048; c++) act3[(size_t)r * 2048 + c] = packed[(size_t)r * 2048 + ((c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3))];
	}
	if (logits) HIPCHK(ctx, hipMemcpy(logits, ctx->d_logits + (size_t)first * HT_CNN_OUT, (size_t)n * HT_CNN_OUT * sizeof(float), hipMemcpyDeviceToHost));
	return HT_OK;
}

// ---- stage entry points (operate on the buffers ht_stage_prepare filled and on the tracker state of slots [0,B)) -------------
extern "C" int ht_stage_fit_error(ht_ctx *ctx, int which, int B, float *err)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!err || which < 0 || which > 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	ht_launch_fit_error(ctx->model, ctx->d_state[which], ctx->d_pts, ctx->d_npts, ctx->d_depth, ctx->d_cams, 64, 64, ctx->par.bone_sum_error_scale, ctx->d_err_old, B, s);
	HIPCHK(ctx, hipMemcpyAsync(err, ctx->d_err_old, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_stage_cloud_rows(ht_ctx *ctx, int which, int stride, int use_cam_origin, int B, float *rows, int *nrows)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!rows || !nrows || which < 0 || which > 1 || stride < 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemsetAsync(ctx->d_rows, 0, (size_t)B * ctx->model.pts_cap * HT_ROW * sizeof(float), s));
	ht_launch_cloud_rows(ctx->model, ctx->d_state[which], ctx->d_pts, ctx->d_npts, ctx->d_cams, nullptr, stride, use_cam_origin, 0, ctx->par, ctx->d_rows, ctx->d_nrows, B, s);
	HIPCHK(ctx, hipMemcpy2DAsync(rows, HT_MAXPTS * HT_ROW * sizeof(float), ctx->d_rows, (size_t)ctx->model.pts_cap * HT_ROW * sizeof(float), HT_MAXPTS * HT_ROW * sizeof(float), B, hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipMemcpyAsync(nrows, ctx->d_nrows, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
// cloud_chamber (physmodel.h:486-496) as HandTracker::update calls it (handtrack.h:774-778): rows [B][5*nb][16], nrows [B] (0 where the frame has no more than
// min_point_num points or boundary_planes is off)
extern "C" int ht_stage_chamber(ht_ctx *ctx, int which, int B, float *rows, int *nrows)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!rows || !nrows || which < 0 || which > 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	const size_t per = (size_t)5 * ctx->model.nb * HT_ROW;
	HIPCHK(ctx, hipMemsetAsync(ctx->d_chamber, 0, (size_t)B * per * sizeof(float), s));
	ht_launch_chamber(ctx->model, ctx->d_state[which], ctx->d_pts, ctx->d_npts, ctx->par.min_point_num, ctx->par.boundary_planes, 10.0f, ctx->d_chamber, ctx->d_nchamber, B, s);
	HIPCHK(ctx, hipMemcpyAsync(rows, ctx->d_chamber, (size_t)B * per * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipMemcpyAsync(nrows, ctx->d_nchamber, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_stage_contacts(ht_ctx *ctx, int which, int B, int cap, float *contacts, int *ncontacts)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!contacts || !ncontacts || which < 0 || which > 1 || cap < 1) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	ht_launch_contacts(ctx->model, ctx->d_state[which], ctx->phys.driftmax, ctx->phys.jiggle_sin, nullptr, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, false, ctx->contact_kernel);
	std::vector<float> tmp((size_t)B * HT_MAXCONTACT * HT_CONTACT);
	HIPCHK(ctx, hipMemcpyAsync(tmp.data(), ctx->d_contacts, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipMemcpyAsync(ncontacts, ctx->d_ncontacts, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	const int ncopy = cap < HT_MAXCONTACT ? cap : HT_MAXCONTACT;
	for (int b = 0; b < B; b++) memcpy(contacts + (size_t)b * cap * HT_CONTACT, tmp.data() + (size_t)b * HT_MAXCONTACT * HT_CONTACT, (size_t)ncopy * HT_CONTACT * sizeof(float));
	return HT_OK;
}
extern "C" int ht_stage_fit(ht_ctx *ctx, int B)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	main_pass(ctx, B, ctx->stream);
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_stage_multistep(ht_ctx *ctx, const float *analysis, int B)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!analysis) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_analysis, analysis, (size_t)B * HT_ANALYSIS * sizeof(float), hipMemcpyHostToDevice, s));
	multistep(ctx, B, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
// steps [from_step, to_step) of MultiStepSim alone (handtrack.h:660-688: every step has its own mix of CNN-driven angular rows, landmark rays and cloud rows): teacher-forced
// single-step tests start a step from a given state
extern "C" int ht_stage_multistep_range(ht_ctx *ctx, const float *analysis, int B, int from_step, int to_step)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!analysis || from_step < 0 || to_step < from_step) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_analysis, analysis, (size_t)B * HT_ANALYSIS * sizeof(float), hipMemcpyHostToDevice, s));
	multistep(ctx, B, s, from_step, to_step);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_stage_scratch_unibody(ht_ctx *ctx, const float *analysis, int B, int n_unibody)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!analysis) return HT_ERR_ARG;
	hipStream_t s = ctx->stream;
	HIPCHK(ctx, hipMemcpyAsync(ctx->d_analysis, analysis, (size_t)B * HT_ANALYSIS * sizeof(float), hipMemcpyHostToDevice, s));
	reset_path(ctx, false, n_unibody, B, s, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// Timing experiments only (HT_DEBUG_SKIP & 2048): per-frame k_solve statistics accumulated in the last scratch record of each frame:
// launches, cycles in chains / two-body linear / angular, cycles in all sweeps, steps (linear, angular), longest chain, row counts.
extern "C" int ht_debug_solve_stats(ht_ctx *ctx, int B, float *out, int reset)
{
	if (!ctx || !ctx->ready || B < 1 || B > ctx->B) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	const int stride = scratch_stride(ctx);
	if (ht_sync_all(ctx) != hipSuccess) return HT_ERR_HIP;
	for (int b = 0; b < B; b++)
	{
		float *src = ctx->d_scratch + ((size_t)b * stride + (stride - 1)) * HT_CREC;
		if (out && hipMemcpy(out + (size_t)b * 16, src, 16 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return HT_ERR_HIP;
		if (reset && hipMemset(src, 0, 16 * sizeof(float)) != hipSuccess) return HT_ERR_HIP;
	}
	return HT_OK;
}
// Tests only: pins the build of k_solve (0 = the launcher's choice by batch and frame size; 1 small, 2 only, 3 mid, 4 tiny = a build whose LDS arrays hold
// nothing, so that every frame takes the HBM placement of its two-body groups, impulse sums and angular records).  The builds differ in where a
// frame's arrays live, never in arithmetic: results must agree bit for bit (tests/test_gpu_solver.py).
// 5 = the EXACT-ORDER instantiation: the same rows, swept by the reference's own Iter functions in the reference's row order (physics.h:251-265, 289-307, 556-581) on one
// lane per frame, the single-body solves of UnibodyFit too.  With it an update equals the CPU restatement bit for bit (tests/test_gpu_exact_solver.py), which
// pins the Jacobian-form arithmetic of the product's sweeps as the solver's only difference from the reference.  Far slower; never chosen by a launcher.
extern "C" int ht_debug_solver_build(ht_ctx *ctx, int which)
{
	if (!ctx || which < 0 || which > 8) return HT_ERR_ARG;      // 6: the build with four angular-row slots per lane (up to 252 rows), otherwise chosen by the model's joint count; 7: the launcher's choice of build, with the two-body rows of EVERY frame taken by the level schedule (round 4's sweeps; otherwise only frames the blocked form of round 5 does not hold): another rounding of the same sweeps
	if ((which == 5 || which == 8) && !ctx->d_exact_lin)
	{
		ht_device_guard dev_guard_(ctx->device);
		void *a = nullptr, *b = nullptr;
		HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * HT_EX_LIN * HT_ROW * sizeof(float))); ctx->allocs.push_back(a); ctx->d_exact_lin = (float *)a;
		HIPCHK(ctx, hipMalloc(&b, (size_t)ctx->B * 256 * 8 * sizeof(float))); ctx->allocs.push_back(b); ctx->d_exact_ang = (float *)b;
	}
	ctx->solver_build = which;
	return HT_OK;
}
// Tests only: which organisation the latest update launched the full-reset branch in (0 = few frames: one k_reset block per CU, a contact block per reset frame;
// 1 = many frames: two blocks per CU, four frames per contact block; -1 = no update yet).  The choice follows the count of reset frames of earlier updates.
extern "C" int ht_debug_reset_organisation(ht_ctx *ctx, int *many)
{
	if (!ctx || !many) return HT_ERR_ARG;
	*many = ctx->last_reset_many;
	return HT_OK;
}
// Tests only: which frames of the latest update took the full-reset branch (handtrack.h:706-711): flags[i] = 1 for tracker slot i (the decision kernel's own array)
extern "C" int ht_debug_reset_flags(ht_ctx *ctx, int *flags, int n)
{
	if (!ctx || !flags || n < 0 || n > ctx->B) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipMemcpy(flags, ctx->d_flags, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
	return HT_OK;
}
// Tests only: pins the organisation of the contact kernel (0 = the launcher's choice; 1 cooperative, 2 lane-per-pair).  Same arithmetic, same contacts in the same order.
extern "C" int ht_debug_contact_kernel(ht_ctx *ctx, int which)
{
	if (!ctx || which < 0 || which > 2) return HT_ERR_ARG;
	ctx->contact_kernel = which;
	return HT_OK;
}
extern "C" int ht_debug_solve_tables(ht_ctx *ctx, int on)
{
	if (!ctx || !ctx->ready) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	if (on < 0 || on > 2) return HT_ERR_ARG;
	if (on) { const int r = ht_alloc_solve_tables(ctx); if (r) return r; }
	ctx->solve_tables = on;
	return HT_OK;
}
extern "C" int ht_debug_solve_tables_header(ht_ctx *ctx, int B, int *hdr)
{
	if (!ctx || !ctx->ready || !hdr || B < 1 || B > ctx->B || !ctx->d_tables) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	if (ht_sync_all(ctx) != hipSuccess) return HT_ERR_HIP;
	if (hipMemcpy2D(hdr, 32 * sizeof(int), ctx->d_tables, (size_t)TB_WORDS * sizeof(float), 32 * sizeof(int), B, hipMemcpyDeviceToHost) != hipSuccess) return HT_ERR_HIP;
	return HT_OK;
}
// Same for k_contacts (last contact slot of each frame): launches, cycles in GJK / polytope runs, polytope runs, polytope cycles in
// face scoring / support scans / mesh surgery, total cycles, candidate pairs, pairs that jiggle, contacts, polytope iterations.
extern "C" int ht_debug_contact_stats(ht_ctx *ctx, int B, float *out, int reset)
{
	if (!ctx || !ctx->ready || B < 1 || B > ctx->B) return HT_ERR_ARG;
	ht_device_guard dev_guard_(ctx->device);
	if (hipDeviceSynchronize() != hipSuccess) return HT_ERR_HIP;
	for (int b = 0; b < B; b++)
	{
		float *src = ctx->d_contacts + ((size_t)b * HT_MAXCONTACT + HT_MAXCONTACT - 2) * HT_CONTACT;      // the last two contact slots: [1] the frame's record, [0] its wave's polytope runs
		if (out && hipMemcpy(out + (size_t)b * 24, src, 24 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return HT_ERR_HIP;
		if (reset && hipMemset(src, 0, 24 * sizeof(float)) != hipSuccess) return HT_ERR_HIP;
	}
	return HT_OK;
}

// ------------------------------------------------------------------------------------------------- segmentation (before the tracker)
extern "C" int ht_segment_vr_dev(ht_ctx *ctx, const uint16_t *d_depth, const float *d_cams, int w, int h, int B, int entry_options, float wrange_lo, float wrange_hi, float diam,
                                 uint16_t *d_tiles, float *d_cams_out, void *stream)
{
	CHECK_READY(ctx);
	(void)wrange_lo;      // the reference's lower bound is commented out (handtrack.h:288)
	if (!d_depth || !d_cams || !d_tiles || !d_cams_out || B < 1) return HT_ERR_ARG;
	hipStream_t s = ht_user_stream(ctx, stream);
	if (w == 64 && h == 64)      // handtrack.h:283-284: a 64x64 frame is returned as it is
	{
		HIPCHK(ctx, hipMemcpyAsync(d_tiles, d_depth, (size_t)B * 4096 * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
		HIPCHK(ctx, hipMemcpyAsync(d_cams_out, d_cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyDeviceToDevice, s));
		return HT_OK;
	}
	if (!ht_segment_supported(w, h)) { ctx->err = "ht_segment_vr: frame size must be a multiple of 4 and at most 320x240 pixels"; return HT_ERR_ARG; }
	ht_launch_segment(d_depth, d_cams, w, h, entry_options, wrange_hi, diam, d_tiles, d_cams_out, B, s);
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
extern "C" int ht_segment_vr(ht_ctx *ctx, const uint16_t *depth, const float *cams, int w, int h, int B, int entry_options, float wrange_lo, float wrange_hi, float diam,
                             uint16_t *tiles, float *cams_out)
{
	CHECK_READY(ctx);
	if (!depth || !cams || !tiles || !cams_out || B < 1 || w < 1 || h < 1) return HT_ERR_ARG;
	uint16_t *d_in = nullptr, *d_tiles = nullptr; float *d_cams = nullptr, *d_co = nullptr;
	const size_t nin = (size_t)B * w * h;
	int rc = HT_OK;
	if (hipMalloc((void **)&d_in, nin * sizeof(uint16_t)) != hipSuccess || hipMalloc((void **)&d_tiles, (size_t)B * 4096 * sizeof(uint16_t)) != hipSuccess ||
	    hipMalloc((void **)&d_cams, (size_t)B * HT_CAM * sizeof(float)) != hipSuccess || hipMalloc((void **)&d_co, (size_t)B * HT_CAM * sizeof(float)) != hipSuccess)
	{ ctx->err = "ht_segment_vr: out of device memory"; rc = HT_ERR_HIP; }
	hipStream_t s = ctx->stream;
	if (rc == HT_OK && (hipMemcpyAsync(d_in, depth, nin * sizeof(uint16_t), hipMemcpyHostToDevice, s) != hipSuccess ||
	                    hipMemcpyAsync(d_cams, cams, (size_t)B * HT_CAM * sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess)) { ctx->err = "ht_segment_vr: upload failed"; rc = HT_ERR_HIP; }
	if (rc == HT_OK) rc = ht_segment_vr_dev(ctx, d_in, d_cams, w, h, B, entry_options, wrange_lo, wrange_hi, diam, d_tiles, d_co, s);
	if (rc == HT_OK && (hipMemcpyAsync(tiles, d_tiles, (size_t)B * 4096 * sizeof(uint16_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
	                    hipMemcpyAsync(cams_out, d_co, (size_t)B * HT_CAM * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
	                    hipStreamSynchronize(s) != hipSuccess)) { ctx->err = "ht_segment_vr: download failed"; rc = HT_ERR_HIP; }
	(void)hipFree(d_in); (void)hipFree(d_tiles); (void)hipFree(d_cams); (void)hipFree(d_co);
	return rc;
}

// ------------------------------------------------------------------------------------------------- caller-built constraint rows
// PhysModel::FitPointCloud(points, linears, angulars, microforce) (physmodel.h:345-356) and the free PhysicsUpdate(bodies, Linears, Angulars, wgeom)
// (physics.h:543-587) take rows the CALLER built.  Row layouts are those of the stage calls: a linear row is 16 floats (rb0 rb1 position0[3]
// position1[3] normal[3] targetdist targetspeednobias forcelimit.x forcelimit.y friction_master), an angular row 8 (rb0 rb1 axis[3] targetspin
// mintorque maxtorque); a body is its index in PhysModel::rigidbodies, -1 = NULL.
static void drop_alloc(ht_ctx *ctx, void *o) { if (!o) return; for (auto &q : ctx->allocs) if (q == o) { q = ctx->allocs.back(); ctx->allocs.pop_back(); break; } (void)hipFree(o); }
static int user_rows_reserve(ht_ctx *ctx, int lin_cap, int ang_cap)
{
	// the replacement is allocated FIRST: when an allocation fails the context keeps its old, valid arrays and capacities
	if (lin_cap > ctx->user_lin_cap)
	{
		HIPCHK(ctx, ht_sync_all(ctx));
		void *a = nullptr, *b = nullptr;
		HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * lin_cap * HT_ROW * sizeof(float)));
		if (hipMalloc(&b, (size_t)ctx->B * lin_cap * sizeof(unsigned short)) != hipSuccess) { (void)hipFree(a); ctx->err = "caller-built rows: out of device memory"; return HT_ERR_HIP; }
		drop_alloc(ctx, ctx->d_user_lin); drop_alloc(ctx, ctx->d_user_pos);
		ctx->allocs.push_back(a); ctx->d_user_lin = (float *)a; ctx->allocs.push_back(b); ctx->d_user_pos = (unsigned short *)b;
		ctx->user_lin_cap = lin_cap;
	}
	if (ang_cap > ctx->user_ang_cap)
	{
		HIPCHK(ctx, ht_sync_all(ctx));
		void *a = nullptr;
		HIPCHK(ctx, hipMalloc(&a, (size_t)ctx->B * ang_cap * HT_AROW * sizeof(float)));
		drop_alloc(ctx, ctx->d_user_ang);
		ctx->allocs.push_back(a); ctx->d_user_ang = (float *)a;
		ctx->user_ang_cap = ang_cap;
	}
	if (!ctx->d_user_n) { void *a = nullptr; HIPCHK(ctx, hipMalloc(&a, (size_t)4 * ctx->B * sizeof(int))); ctx->allocs.push_back(a); ctx->d_user_n = (int *)a; }
	return HT_OK;
}
static int upload_angulars(ht_ctx *ctx, int B, const float *angulars, int acap, const int *nangulars, std::vector<int> &na)
{
	na.assign(B, 0);
	for (int b = 0; b < B; b++)
	{
		na[b] = (angulars && nangulars) ? nangulars[b] : 0;
		if (na[b] < 0 || na[b] > acap || na[b] > 126 || 13 + 6 * ctx->model.nj + na[b] > 252) { ctx->err = "caller-built angular rows: a count is negative, exceeds the stride, exceeds 126 or leaves no room for the model's own rows among the 252 angular rows a solve keeps"; return HT_ERR_ARG; }
		for (int i = 0; i < na[b]; i++) { const float *r = angulars + ((size_t)b * acap + i) * HT_AROW; if ((int)r[0] >= ctx->model.nb || (int)r[1] >= ctx->model.nb) { ctx->err = "caller-built angular rows: body index out of range"; return HT_ERR_ARG; } }
	}
	if (acap > 0 && angulars) HIPCHK(ctx, hipMemcpy2D(ctx->d_user_ang, (size_t)ctx->user_ang_cap * HT_AROW * sizeof(float), angulars, (size_t)acap * HT_AROW * sizeof(float), (size_t)acap * HT_AROW * sizeof(float), B, hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_user_n + 3 * (size_t)ctx->B, na.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	return HT_OK;
}
// void PhysModel::FitPointCloud(const std::vector<float3> &points, std::vector<LimitLinear> linears, std::vector<LimitAngular> angulars, float microforce)
// on model `which` of slots [0,B): the caller's linear rows, then CloudConstraints(points) with the +-microforce limits (x physics_weak_force on
// bodies 0-2), then the joints' nailed rows; the caller's angular rows, then the joints' range rows; PhysicsUpdate with collision; SanityCheck.  As
// at every call site of the reference (handtrack.h:672-685, 773-779, 803-819) the joint ranges are first brought up to date from the pose
// (HandModelEnhancements, :417-420, 434-440).  The caller's linear rows must be anchored in the world (rb0 == NULL), as the rows of every such call
// site are (landmark rays, boundary planes, the annotator's nail): they join the per-body row chains ahead of the cloud rows.
extern "C" int ht_fit_rows(ht_ctx *ctx, int which, int B, const float *points, int pcap, const int *npoints, const float *linears, int lcap, const int *nlinears,
                           const float *angulars, int acap, const int *nangulars, float microforce)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (which < 0 || which > 1 || pcap < 0 || lcap < 0 || acap < 0) return HT_ERR_ARG;
	const int nb = ctx->model.nb;
	std::vector<int> nl(B, 0), na;
	for (int b = 0; b < B; b++)
	{
		nl[b] = (linears && nlinears) ? nlinears[b] : 0;
		if (nl[b] < 0 || nl[b] > lcap || nl[b] > 5 * nb + 128) { ctx->err = "ht_fit_rows: a linear row count is negative, exceeds the stride or exceeds the 5 * bodies + 128 caller rows a frame's scratch slot holds"; return HT_ERR_ARG; }
		for (int i = 0; i < nl[b]; i++)
		{
			const float *r = linears + ((size_t)b * lcap + i) * HT_ROW;
			if ((int)r[0] >= 0 || (int)r[1] < 0 || (int)r[1] >= nb || r[15] != 0.0f) { ctx->err = "ht_fit_rows: the caller's linear rows must act on one body from the world (rb0 == NULL, no friction master); use ht_physics_update for general rows"; return HT_ERR_ARG; }
		}
	}
	{ const int r = user_rows_reserve(ctx, lcap > 1 ? lcap : 1, acap > 1 ? acap : 1); if (r) return r; }
	{ const int r = upload_angulars(ctx, B, angulars, acap, nangulars, na); if (r) return r; }
	// the points (ht_set_points grows the point capacity when it has to)
	{
		static const float none[3] = { 0, 0, 0 }; std::vector<int> zero(B, 0);
		const int r = (points && npoints && pcap > 0) ? ht_set_points(ctx, B, points, pcap, npoints) : ht_set_points(ctx, 1, none, 1, zero.data());
		if (r) return r;
		if (!(points && npoints && pcap > 0)) HIPCHK(ctx, hipMemset(ctx->d_npts, 0, (size_t)B * sizeof(int)));
	}
	if (lcap > 0 && linears) HIPCHK(ctx, hipMemcpy2D(ctx->d_user_lin, (size_t)ctx->user_lin_cap * HT_ROW * sizeof(float), linears, (size_t)lcap * HT_ROW * sizeof(float), (size_t)lcap * HT_ROW * sizeof(float), B, hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_user_n, nl.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	hipStream_t s = ctx->stream;
	const bool coll = ctx->phys.use_collision != 0;
	ht_params par = ctx->par; par.microforce = microforce;
	const cloud_records cr = cloud_rec(ctx);
	ht_launch_cloud_rows(ctx->model, ctx->d_state[which], ctx->d_pts, ctx->d_npts, ctx->d_cams, nullptr, 1, 0, 1, par, ctx->d_rows, ctx->d_nrows, B, s, 0.0f, 0.0f, &cr);
	if (coll) ht_launch_contacts(ctx->model, ctx->d_state[which], ctx->phys.driftmax, ctx->phys.jiggle_sin, nullptr, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, false, ctx->contact_kernel);
	solve_args a;
	memset(&a, 0, sizeof a);
	a.sf_select = -1;
	a.caps = reinterpret_cast<int *>(ctx->d_epa_ws) + 2;
	a.rows_pre = ctx->d_user_lin; a.n_pre = ctx->d_user_n; a.pre_stride = ctx->user_lin_cap;
	a.cloud_body = ctx->d_rowbody; a.n_cloud = ctx->d_nrows;
	a.contacts = coll ? ctx->d_contacts : nullptr; a.ncontacts = ctx->d_ncontacts;
	a.ang_user = ctx->d_user_ang; a.n_ang_user = ctx->d_user_n + 3 * (size_t)ctx->B; a.ang_user_stride = ctx->user_ang_cap;
	{ int mx = 0; for (int v : na) mx = v > mx ? v : mx; a.ang_extra_bound = mx; }
	a.analysis = ctx->d_analysis; a.cams = ctx->d_cams;
	a.state = ctx->d_state[which]; a.scratch = ctx->d_scratch; a.scratch_stride = scratch_stride(ctx); a.batch = ctx->B;
	if (exact_solver(ctx)) { ctx->err = "the exact-order instantiation (ht_debug_solver_build 5) serves the update entry points only"; return HT_ERR_STATE; }
	a.force_build = ctx->solver_build;
	ctx->model.pts_bound = 0;
	ht_launch_solve(ctx->model, ctx->phys, a, B, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}
// void PhysicsUpdate(const std::vector<RigidBody*> &rigidbodies, std::vector<LimitLinear> &Linears, std::vector<LimitAngular> &Angulars, wgeom = {})
// (physics.h:543-587) on model `which` of slots [0,B): the caller's rows are all there is (plus the collision rows when physics_use_collision is set).
// Rows keep the caller's order: the leading rows that act on one body from the world (rb0 == NULL) run as per-body chains, the rest in groups of up
// to three consecutive rows on the same body pair; a friction row (friction_master -1 / -2) must follow its master directly, as ConstrainContacts
// builds them (physics.h:463-489).  At most 32 such groups per frame.
extern "C" int ht_physics_update(ht_ctx *ctx, int which, int B, const float *linears, int lcap, const int *nlinears, const float *angulars, int acap, const int *nangulars)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (which < 0 || which > 1 || lcap < 0 || acap < 0) return HT_ERR_ARG;
	const int nb = ctx->model.nb;
	std::vector<int> npre(B, 0), ntail(B, 0), ngrp(B, 0), na;
	std::vector<unsigned short> pos((size_t)B * (lcap > 1 ? lcap : 1), 0);
	int most_pre = 0;
	for (int b = 0; b < B; b++)
	{
		const int n = (linears && nlinears) ? nlinears[b] : 0;
		if (n < 0 || n > lcap) { ctx->err = "ht_physics_update: a linear row count is negative or exceeds the stride"; return HT_ERR_ARG; }
		const float *rows = linears + (size_t)b * lcap * HT_ROW;
		for (int i = 0; i < n; i++) if ((int)rows[i * HT_ROW] >= nb || (int)rows[i * HT_ROW + 1] >= nb || ((int)rows[i * HT_ROW] < 0 && (int)rows[i * HT_ROW + 1] < 0)) { ctx->err = "ht_physics_update: body index out of range"; return HT_ERR_ARG; }
		int p = 0;
		while (p < n && (int)rows[p * HT_ROW] < 0 && rows[p * HT_ROW + 15] == 0.0f && !(p + 1 < n && rows[(p + 1) * HT_ROW + 15] != 0.0f)) p++;
		npre[b] = p; ntail[b] = n - p; if (p > most_pre) most_pre = p;
		// groups of the tail: up to three consecutive rows on one body pair; a row that friction rows follow opens a group
		int g = -1, slot = 3, g0 = 0; int pa = -2, pb = -2;
		for (int i = p; i < n; i++)
		{
			const float *r = rows + (size_t)i * HT_ROW;
			const int fm = (int)r[15], ra = (int)r[0], rb = (int)r[1];
			const bool next_is_friction = i + 1 < n && rows[(size_t)(i + 1) * HT_ROW + 15] != 0.0f && fm == 0;
			if (fm != 0)
			{
				if (g < 0 || -fm != slot || slot > 2 || ra != pa || rb != pb) { ctx->err = "ht_physics_update: a friction row must directly follow its master row (friction_master -1, then -2) on the same bodies"; return HT_ERR_ARG; }
				for (int k = g0; k <= i; k++) pos[(size_t)b * lcap + (k - p)] |= 0x8000;
			}
			else if (g < 0 || slot > 2 || ra != pa || rb != pb || next_is_friction || (pos[(size_t)b * lcap + (g0 - p)] & 0x8000)) { g++; slot = 0; g0 = i; pa = ra; pb = rb; }
			pos[(size_t)b * lcap + (i - p)] |= (unsigned short)((g << 2) | slot);
			slot++;
		}
		ngrp[b] = g + 1;
		if (ngrp[b] > HT_MAXNJ) { ctx->err = "ht_physics_update: more than 32 groups of two-body rows in a frame"; return HT_ERR_ARG; }
	}
	if (most_pre > ctx->model.pts_cap) { const int r = ht_reserve_points_locked(ctx, most_pre); if (r) return r; }
	{ const int r = user_rows_reserve(ctx, lcap > 1 ? lcap : 1, acap > 1 ? acap : 1); if (r) return r; }
	{ const int r = upload_angulars(ctx, B, angulars, acap, nangulars, na); if (r) return r; }
	const size_t pc = (size_t)ctx->model.pts_cap;
	for (int b = 0; b < B; b++)
	{
		const float *rows = linears ? linears + (size_t)b * lcap * HT_ROW : nullptr;
		if (npre[b]) HIPCHK(ctx, hipMemcpy(ctx->d_rows + (size_t)b * pc * HT_ROW, rows, (size_t)npre[b] * HT_ROW * sizeof(float), hipMemcpyHostToDevice));
		if (ntail[b])
		{
			HIPCHK(ctx, hipMemcpy(ctx->d_user_lin + (size_t)b * ctx->user_lin_cap * HT_ROW, rows + (size_t)npre[b] * HT_ROW, (size_t)ntail[b] * HT_ROW * sizeof(float), hipMemcpyHostToDevice));
			HIPCHK(ctx, hipMemcpy(ctx->d_user_pos + (size_t)b * ctx->user_lin_cap, pos.data() + (size_t)b * lcap, (size_t)ntail[b] * sizeof(unsigned short), hipMemcpyHostToDevice));
		}
	}
	HIPCHK(ctx, hipMemcpy(ctx->d_nrows, npre.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_user_n + (size_t)ctx->B, ntail.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_user_n + 2 * (size_t)ctx->B, ngrp.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	hipStream_t s = ctx->stream;
	const bool coll = ctx->phys.use_collision != 0;
	if (coll) ht_launch_contacts(ctx->model, ctx->d_state[which], ctx->phys.driftmax, ctx->phys.jiggle_sin, nullptr, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, false, ctx->contact_kernel);
	solve_args a;
	memset(&a, 0, sizeof a);
	a.sf_select = -1;
	a.caps = reinterpret_cast<int *>(ctx->d_epa_ws) + 2;
	a.rows_cloud = ctx->d_rows; a.n_cloud = ctx->d_nrows;
	a.contacts = coll ? ctx->d_contacts : nullptr; a.ncontacts = ctx->d_ncontacts;
	a.ang_user = ctx->d_user_ang; a.n_ang_user = ctx->d_user_n + 3 * (size_t)ctx->B; a.ang_user_stride = ctx->user_ang_cap;
	{ int mx = 0; for (int v : na) mx = v > mx ? v : mx; a.ang_extra_bound = mx; }
	a.lin_tail = ctx->d_user_lin; a.n_lin_tail = ctx->d_user_n + (size_t)ctx->B; a.lin_tail_stride = ctx->user_lin_cap;
	a.lin_tail_pos = ctx->d_user_pos; a.n_tail_groups = ctx->d_user_n + 2 * (size_t)ctx->B;
	a.no_model_rows = 1;
	a.analysis = ctx->d_analysis; a.cams = ctx->d_cams;
	a.state = ctx->d_state[which]; a.scratch = ctx->d_scratch; a.scratch_stride = scratch_stride(ctx); a.batch = ctx->B;
	if (exact_solver(ctx)) { ctx->err = "the exact-order instantiation (ht_debug_solver_build 5) serves the update entry points only"; return HT_ERR_STATE; }
	a.force_build = ctx->solver_build;
	ctx->model.pts_bound = 0;
	ht_launch_solve(ctx->model, ctx->phys, a, B, s);
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// ------------------------------------------------------------------------------------------------- slowfit (annotation fit loop)
// HandTracker::slowfit (handtrack.h:786-821) on the handmodel of slots [0,B) against the points ht_stage_prepare left on the device.
extern "C" int ht_slowfit(ht_ctx *ctx, int B, int hold, const float *refpose, int steps, int select_rb, const float *spoint, const float *rbpoint, const float *crays, int ncray)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	const int nb = ctx->model.nb;
	if (steps < 1 || ncray < 0 || ncray > 8 || select_rb >= nb || (select_rb >= 0 && (!spoint || !rbpoint)) || (ncray > 0 && !crays)) { ctx->err = "ht_slowfit: bad arguments"; return HT_ERR_ARG; }
	hipStream_t s = ctx->stream;
	if (!ctx->d_sf_ref)
	{
		void *p = nullptr, *q = nullptr;
		HIPCHK(ctx, hipMalloc(&p, (size_t)ctx->B * nb * HT_POSE * sizeof(float))); ctx->allocs.push_back(p); ctx->d_sf_ref = (float *)p;
		HIPCHK(ctx, hipMalloc(&q, (size_t)ctx->B * 8 * 4 * sizeof(float))); ctx->allocs.push_back(q); ctx->d_sf_crays = (float *)q;
	}
	const bool rel = hold && refpose;
	if (rel) HIPCHK(ctx, hipMemcpyAsync(ctx->d_sf_ref, refpose, (size_t)B * nb * HT_POSE * sizeof(float), hipMemcpyHostToDevice, s));
	if (ncray) HIPCHK(ctx, hipMemcpyAsync(ctx->d_sf_crays, crays, (size_t)B * 8 * 4 * sizeof(float), hipMemcpyHostToDevice, s));
	const bool coll = ctx->phys.use_collision != 0;
	for (int st = 0; st < steps; st++)
	{
		const bool cloud = st < steps - 1;
		const cloud_records cr = cloud_rec(ctx);
		if (cloud) ht_launch_cloud_rows(ctx->model, ctx->d_state[0], ctx->d_pts, ctx->d_npts, ctx->d_cams, nullptr, 1, 0, 4, ctx->par, ctx->d_rows, ctx->d_nrows, B, s,
		                                1.0f * (float)(steps - st) / (float)steps, 0.1f * (float)(st < steps - 2), &cr);
		if (coll) ht_launch_contacts(ctx->model, ctx->d_state[0], ctx->phys.driftmax, ctx->phys.jiggle_sin, nullptr, ctx->d_epa_ws, ctx->d_contacts, ctx->d_ncontacts, B, s, false, ctx->contact_kernel);
		solve_args a;
		memset(&a, 0, sizeof a);
		a.caps = reinterpret_cast<int *>(ctx->d_epa_ws) + 2;
		a.cloud_body = cloud ? ctx->d_rowbody : nullptr; a.n_cloud = ctx->d_nrows;
		a.contacts = coll ? ctx->d_contacts : nullptr; a.ncontacts = ctx->d_ncontacts;
		a.analysis = ctx->d_analysis; a.cams = ctx->d_cams;
		a.state = ctx->d_state[0]; a.scratch = ctx->d_scratch; a.scratch_stride = scratch_stride(ctx); a.batch = ctx->B;
		a.sf_ncray = st < 5 ? ncray : 0; a.sf_crays = ctx->d_sf_crays; a.sf_select = select_rb;
		for (int i = 0; i < 3; i++) { a.sf_spoint[i] = spoint ? spoint[i] : 0.0f; a.sf_rbpoint[i] = rbpoint ? rbpoint[i] : 0.0f; }
		a.ray_rows = (a.sf_ncray > 0 || select_rb >= 0) ? 1 : 0;
		a.sf_refpose = rel ? ctx->d_sf_ref : nullptr; a.sf_hold = rel ? hold : 0;
		a.ang_extra_bound = rel ? 3 * ctx->model.nj : 0;      // RelativeAngularConstraints: one row per ranged axis of a joint at most
		a.steps_keyangles = ctx->par.steps_keyangles; a.min_cray_prob = ctx->par.min_cray_prob;
		if (exact_solver(ctx)) { ctx->err = "the exact-order instantiation (ht_debug_solver_build 5) serves the update entry points only"; return HT_ERR_STATE; }
		a.force_build = ctx->solver_build;
		ht_launch_solve(ctx->model, ctx->phys, a, B, s);
	}
	HIPCHK(ctx, hipStreamSynchronize(s));
	HIPCHK(ctx, hipGetLastError());
	return HT_OK;
}

// Caller-supplied point clouds for the stage functions / ht_slowfit (the reference's slowfit takes its points as an argument):
// points [B][cap][3], npoints [B] (<= cap; the context's point capacity grows to the largest).
extern "C" int ht_set_points(ht_ctx *ctx, int B, const float *points, int cap, const int *npoints)
{
	CHECK_READY(ctx); CHECK_MODEL(ctx); CHECK_BATCH(ctx, B);
	if (!points || !npoints || cap < 1) return HT_ERR_ARG;
	ctx->model.pts_bound = 0;
	{ int most = 0; for (int b = 0; b < B; b++) most = std::max(most, std::min(npoints[b], cap)); if (most > HT_POINTS_LIMIT) return HT_ERR_ARG; const int r = ht_reserve_points_locked(ctx, std::max(most, 1)); if (r) return r; }
	const size_t pc = (size_t)ctx->model.pts_cap;
	std::vector<float4> h((size_t)B * pc, make_float4(0, 0, 0, 0)); std::vector<int> n(B);
	for (int b = 0; b < B; b++)
	{
		n[b] = npoints[b] < 0 ? 0 : npoints[b] > cap ? cap : npoints[b]; if (n[b] > (int)pc) n[b] = (int)pc;
		for (int i = 0; i < n[b]; i++) { const float *p = points + ((size_t)b * cap + i) * 3; h[(size_t)b * pc + i] = make_float4(p[0], p[1], p[2], 0.0f); }
	}
	HIPCHK(ctx, hipMemcpy(ctx->d_pts, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
	HIPCHK(ctx, hipMemcpy(ctx->d_npts, n.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice));
	return HT_OK;
}
